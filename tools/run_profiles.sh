# Collects every profile the bench line and DESIGN.md quote, on the CURRENT build.  Usage (GPU box): bash tools/run_profiles.sh <tag>
# (the microbenchmarks are built here, before the box: make -C tools/ubench)
set -x
TAG=${1:-r06}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
./tools/ubench/build/fpbench > $OUT/fpbench.txt 2>&1
./tools/ubench/build/fqbench > $OUT/fqbench.txt 2>&1
./tools/ubench/build/invbench > $OUT/invbench.txt 2>&1
# the group law in isolation at two waves per SIMD: products, G1 / G2 doublings and mixed additions, the cost of each piece of glue
./tools/ubench/build/fqgroup 200 > $OUT/fqgroup.txt 2>&1
# stage 2a of the pairing product in isolation: k_line_products_q (six-product sums) against k_line_products_k (Karatsuba), bit-compared
( ./tools/ubench/build/lpbench 17 2 5; ./tools/ubench/build/lpbench 17 6 3; ./tools/ubench/build/lpbench 15 2 3 ) > $OUT/lpbench.txt 2>&1
timeout 900 python3 bench.py --steps 5 --warmup 1 > $OUT/bench_n1.json 2> $OUT/bench_n1.err        # incl. the CPU baseline at n = 2^20 (~100 s)
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kstats -o k --output-format csv -- python3 bench.py --steps 3 --warmup 1 --cpu-log-n 0 > $OUT/kstats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o f --output-format csv -- python3 bench.py --steps 1 --warmup 0 --cpu-log-n 0 > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o w --output-format csv -- python3 bench.py --steps 1 --warmup 0 --cpu-log-n 0 > $OUT/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace -d $OUT/pmc_sq -o s --output-format csv -- python3 bench.py --steps 1 --warmup 0 --cpu-log-n 0 > $OUT/pmc_sq.log 2>&1
python3 tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/hbm_traffic.csv $OUT/traffic.json > $OUT/traffic_summary.txt
# config 3: MSM n = 2^20 (durations + HBM bytes)
timeout 300 python3 tools/msm_sweep.py 20 > $OUT/msm_sweep.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/msm_kstats -o k --output-format csv -- python3 tools/msm_sweep.py 20 > $OUT/msm_kstats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/msm_fetch -o f --output-format csv -- python3 tools/msm_sweep.py 20 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/msm_write -o w --output-format csv -- python3 tools/msm_sweep.py 20 > /dev/null 2>&1
python3 tools/pmc_traffic.py $OUT/msm_fetch $OUT/msm_write $OUT/msm_hbm_traffic.csv $OUT/msm_traffic.json > $OUT/msm_traffic_summary.txt
# SQ counters: summarise per kernel
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("$OUT/pmc_sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ripp::", "")
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] == "SQ_WAVES": cnt[k] += 1
names = ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_LDS_BANK_CONFLICT"]
with open("$OUT/sq_counters.csv", "w") as o:
    o.write("# rocprofv3 --pmc " + " ".join(names) + " over bench.py --steps 1 --warmup 0; sums over all launches of a kernel\n")
    o.write("kernel,launches," + ",".join(names) + "\n")
    for k in sorted(tot, key=lambda k: -tot[k]["SQ_INSTS_VALU"])[:25]:
        o.write(k + "," + str(cnt[k]) + "," + ",".join("%.0f" % tot[k][n] for n in names) + "\n")
PY
find $OUT -name "*_kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -delete
find $OUT -name "*agent_info.csv" -delete
ls -laR $OUT | head -60
cat $OUT/fpbench.txt; cat $OUT/fqbench.txt; cat $OUT/bench_n1.json; cat $OUT/traffic_summary.txt; cat $OUT/msm_sweep.txt; cat $OUT/msm_traffic_summary.txt
