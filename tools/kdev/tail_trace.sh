# GPU timeline of the pipelined tail rounds of a small proof: bash tools/kdev/tail_trace.sh <n>
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
N=${1:-64}
rm -rf gpurun_out/tt; mkdir -p gpurun_out/tt
rocprofv3 --kernel-trace -d gpurun_out/tt -o t --output-format csv -- python3 tools/kdev/lat16.py $N 2 > gpurun_out/tt/run.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/tt/**/t_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-90:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  +%7.1f  dur %7.1f  gap %7.1f  q%-3s %s" % ((s - t0) / 1e3, 0, (e - s) / 1e3, (s - prev_end) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0].replace("ripp::", "").replace("void ", "")[:44]))
    prev_end = max(prev_end, e)
PY
rm -rf gpurun_out/tt
