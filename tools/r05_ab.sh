# build round 5 A/B runs on one box (same-box comparisons): bash tools/r05_ab.sh
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_ab; mkdir -p $O
b() { tag=$1; shift; env "$@" timeout 300 python3 bench.py --steps 8 --warmup 2 --cpu-log-n 0 > $O/bench_$tag.json 2> $O/bench_$tag.err; python3 - <<PY
import json
d = json.load(open("$O/bench_$tag.json"))
print("$tag", round(d["ms_per_step"], 1), "median", round(d["ms_per_step_median"], 1), "hash", d["statement_hash_ms"], "post", round(d["post_hash_ms"], 1), "wait", d["phase_ms"]["hash_wait_ms"], d["look_ahead"]["pairs"])
PY
}
b default RIPP_X=1
b tailpipe12 RIPP_TAIL_PIPE_MAX=4096
b tailpipe13 RIPP_TAIL_PIPE_MAX=8192
b glssplit15 RIPP_GLS_SPLIT_MAX=32768
b default2 RIPP_X=1
# item 8: achieved clock of the pairing kernels against an ALU-only kernel of the same instruction mix
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $O/clk_bench -o c --output-format csv -- python3 bench.py --steps 1 --warmup 0 --cpu-log-n 0 > $O/clk_bench.log 2>&1
timeout 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $O/clk_fq -o c --output-format csv -- ./tools/ubench/build/fqbench > $O/clk_fq.log 2>&1
python3 tools/pmc_clock.py $O/clk_bench $O/clk_fq > $O/clock_table.txt 2>&1; cat $O/clock_table.txt
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
