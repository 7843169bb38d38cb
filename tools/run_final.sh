# End-of-round measurement set on the CURRENT build (GPU box): bash tools/run_final.sh <tag>
set -x
TAG=${1:-r06}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
bash tools/run_profiles.sh $TAG > gpurun_out/prof_$TAG.log 2>&1
O=gpurun_out/final_$TAG; mkdir -p $O
timeout 900 python3 tools/scaling_ipp.py 1 20 $O --cpu-max 12 > $O/scaling_ipp.log 2>&1
timeout 300 python3 tools/aggregate_bench.py 14 > $O/aggregate_2p14.json 2> $O/aggregate_2p14.err
for G in 2 4 8; do RIPP_BENCH_SINGLE_DEVICE=1 timeout 600 python3 bench.py --gpus $G --steps 3 --warmup 1 --cpu-log-n 0 > $O/bench_n${G}_single_device_gloo.json 2> $O/bench_n$G.err; done
timeout 900 python3 tools/scaling_ipp.py 4 20 $O/c377 --cpu-max 12 --curve 377 > $O/scaling_ipp_377.log 2>&1
bash tools/post_hash_timeline.sh 0 450 > $O/post_hash_timeline.txt 2>&1
# the look-ahead planner priced for a slower device (ripp_config.plan_derate_pct): same box, alternating with the plain plan
( for D in 0 9 0 15 0 25; do RIPP_PLAN_DERATE_PCT=$D python3 bench.py --steps 5 --cpu-log-n 0 > $O/derate_$D.json 2>/dev/null; python3 - $O/derate_$D.json $D <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("plan_derate_pct %2s: step %.1f ms  hash %.1f  post-hash %.1f  hash_wait %.1f  look-ahead pairs %d  fold %.1f" % (sys.argv[2], d["ms_per_step"], d["statement_hash_ms"], d["post_hash_ms"], d["phase_ms"]["hash_wait_ms"], d["look_ahead"]["pairs"], d["phase_ms"]["fold_ms"]))
PY
done ) > $O/plan_derate_ab.txt 2>&1
cp $O/derate_9.json $O/bench_n1_derated.json
# randomized stress against the oracle: 4 and 8 ranks on this GPU, then the pipelined tail
timeout 400 python3 tools/stress_sharded.py 41 90 2 14 4 > $O/stress_world4.txt 2>&1
timeout 400 python3 tools/stress_sharded.py 81 90 3 14 8 > $O/stress_world8.txt 2>&1
timeout 200 python3 tools/stress_tail.py 5 60 1 12 > $O/stress_tail.txt 2>&1
timeout 300 python3 tools/poly_commit_bench.py 2 8 > $O/poly_commit_bench.csv 2> $O/poly_commit.err
# one rank of G alone on this GPU with recorded peers: rank{0,1}_of_{2,4,8}_{bench.json,timeline.txt}
for G in 2 4; do timeout 500 python3 tools/replay_ranks.py all --world $G --log-n 20 --out-dir $O/replay > $O/replay_w$G.log 2>&1; done
RIPP_HOT_WORKERS=1 timeout 900 python3 tools/replay_ranks.py all --world 8 --log-n 20 --out-dir $O/replay --sweep-latency-us 20,50,100,200 > $O/replay_w8.log 2>&1
timeout 500 python3 tools/replay_ranks.py all --world 8 --log-n 14 --workload aggregate --out-dir $O/replay > $O/replay_agg_w8.log 2>&1
timeout 900 python3 tools/sipp_2p24.py --out $O/sipp_2p24.txt > $O/sipp_2p24.log 2>&1
ls -la $O $O/c377; tail -3 $O/scaling_ipp.log; cat $O/aggregate_2p14.json | head -30; cat $O/bench_n2_single_device_gloo.json | cut -c1-300; for f in stress_world4 stress_world8 stress_tail; do tail -n 2 $O/$f.txt; done
