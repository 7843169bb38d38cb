# End-of-round measurement set on the CURRENT build (GPU box): bash tools/run_final.sh <tag>
set -x
TAG=${1:-r03}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
bash tools/run_profiles.sh $TAG > gpurun_out/prof_$TAG.log 2>&1
O=gpurun_out/final_$TAG; mkdir -p $O
timeout 900 python3 tools/scaling_ipp.py 1 20 $O --cpu-max 12 > $O/scaling_ipp.log 2>&1
timeout 300 python3 tools/aggregate_bench.py 14 > $O/aggregate_2p14.json 2> $O/aggregate_2p14.err
RIPP_BENCH_SINGLE_DEVICE=1 timeout 300 python3 bench.py --gpus 2 --steps 3 --warmup 1 --cpu-log-n 0 > $O/bench_n2_single_device_gloo.json 2> $O/bench_n2.err
timeout 600 python3 tools/scaling_ipp.py 4 18 $O/c377 --cpu-max 12 --curve 377 > $O/scaling_ipp_377.log 2>&1
timeout 300 python3 tools/poly_commit_bench.py 2 8 > $O/poly_commit_bench.csv 2> $O/poly_commit.err
ls -la $O $O/c377; tail -3 $O/scaling_ipp.log; cat $O/aggregate_2p14.json | head -30; cat $O/bench_n2_single_device_gloo.json | cut -c1-300
