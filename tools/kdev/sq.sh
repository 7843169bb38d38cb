cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf gpurun_out/sq; mkdir -p gpurun_out/sq
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d gpurun_out/sq -o s --output-format csv -- python3 bench.py --steps 1 --warmup 0 --cpu-log-n 0 > gpurun_out/sq/run.log 2>&1
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); 
for f in glob.glob("gpurun_out/sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ripp::", "")
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(tot, key=lambda k: -tot[k].get("SQ_WAVE_CYCLES",0))[:12]: print(k, {n: "%.3g" % v for n, v in tot[k].items()})
PY
tail -2 gpurun_out/sq/run.log | cut -c1-200
rm -rf gpurun_out/sq
