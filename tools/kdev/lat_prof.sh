# kernel-level profile of a small SIPP proof (tail-round latency): bash tools/kdev/lat_prof.sh <n>
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
N=${1:-64}
mkdir -p gpurun_out/latprof
rocprofv3 --kernel-trace --stats -d gpurun_out/latprof -o lat --output-format csv -- python3 tools/kdev/lat16.py $N 20 > gpurun_out/latprof/run.log 2>&1
tail -3 gpurun_out/latprof/run.log
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/latprof/**/lat_kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:22]:
    print("%-60s calls %5s avg %9.1f us  total %8.2f ms" % (r["Name"].replace("ripp::","").replace("void ","")[:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
find gpurun_out/latprof -name "*_kernel_trace.csv" -delete; find gpurun_out/latprof -name "*.db" -delete
