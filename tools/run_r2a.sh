set -x
mkdir -p gpurun_out/r2a
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
(time timeout 1500 python -m pytest tests -m gpu -x -q --durations=15) > gpurun_out/r2a/gputest.log 2>&1
tail -30 gpurun_out/r2a/gputest.log
timeout 300 python bench.py --steps 5 --warmup 1 > gpurun_out/r2a/bench_n1.json 2> gpurun_out/r2a/bench_n1.err
cat gpurun_out/r2a/bench_n1.json
RIPP_BENCH_SINGLE_DEVICE=1 timeout 400 python bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r2a/bench_n2.json 2> gpurun_out/r2a/bench_n2.err; echo "n2 rc=$?"
cat gpurun_out/r2a/bench_n2.json; tail -5 gpurun_out/r2a/bench_n2.err
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r2a/kstats -o k --output-format csv -- python3 bench.py --steps 3 --warmup 1 --cpu-log-n 0 > gpurun_out/r2a/kstats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/r2a/pmc_fetch -o f --output-format csv -- python3 bench.py --steps 1 --warmup 0 --cpu-log-n 0 > gpurun_out/r2a/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/r2a/pmc_write -o w --output-format csv -- python3 bench.py --steps 1 --warmup 0 --cpu-log-n 0 > gpurun_out/r2a/pmc_write.log 2>&1
python3 tools/pmc_traffic.py gpurun_out/r2a/pmc_fetch gpurun_out/r2a/pmc_write gpurun_out/r2a/hbm_traffic.csv gpurun_out/r2a/traffic.json
find gpurun_out/r2a -name "*_kernel_trace.csv" -size +1M -delete
find gpurun_out/r2a -name "*counter_collection.csv" -size +8M -delete
ls -la gpurun_out/r2a gpurun_out/r2a/kstats
