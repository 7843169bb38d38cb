# GPU timeline of the mid rounds (after the statement hash) of one n = 2^20 proof: bash tools/post_hash_timeline.sh [first_ms] [last_ms]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf gpurun_out/mt; mkdir -p gpurun_out/mt
rocprofv3 --kernel-trace -d gpurun_out/mt -o t --output-format csv -- python3 bench.py --steps 1 --warmup 0 --cpu-log-n 0 > gpurun_out/mt/run.log 2>&1
python3 - "$@" <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/mt/**/t_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last proof: starts at the last k_scale_g1_glv_q
starts = [i for i, r in enumerate(rows) if "k_scale_g1_glv" in r["Kernel_Name"]]
rows = rows[starts[-1]:]
t0 = int(rows[0]["Start_Timestamp"])
lo = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
hi = float(sys.argv[2]) if len(sys.argv) > 2 else 1e9
prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    ms = (s - t0) / 1e6
    if ms >= lo and ms <= hi and (e - s) > 20000:
        print("%8.2f ms  dur %8.3f  gap %7.3f  q%-3s %s" % (ms, (e - s) / 1e6, (s - prev_end) / 1e6, r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0].replace("ripp::", "").replace("void ", "")[:50]))
    prev_end = max(prev_end, e)
PY
rm -rf gpurun_out/mt
