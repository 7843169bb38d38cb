cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/full
(time timeout 2400 python -m pytest tests -m gpu -x -q --durations=8) > gpurun_out/full/gputest.log 2>&1
tail -16 gpurun_out/full/gputest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 300 python bench.py --steps 5 --warmup 1 2>/dev/null | tee gpurun_out/full/bench_n1.json | cut -c1-400
