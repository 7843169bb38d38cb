# per-kernel durations of bench.py (2 steps): bash tools/kdev/kstats.sh [ENV=VAL ...]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rm -rf gpurun_out/kst; mkdir -p gpurun_out/kst
rocprofv3 --kernel-trace --stats -d gpurun_out/kst -o k --output-format csv -- python3 bench.py --steps 2 --warmup 1 --cpu-log-n 0 > gpurun_out/kst/run.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/kst/**/k_kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:26]:
    print("%-62s calls %4s avg %9.1f us max %9.1f total/proof %7.2f ms" % (r["Name"].replace("ripp::","").replace("void ","")[:62], r["Calls"], float(r["AverageNs"])/1e3, float(r["MaxNs"])/1e3, float(r["TotalDurationNs"])/3e6))
PY
find gpurun_out/kst -name "*_kernel_trace.csv" -delete; find gpurun_out/kst -name "*.db" -delete
