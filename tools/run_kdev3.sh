set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_sharded_gloo.py -m gpu -x -q 2>&1 | tail -15
RIPP_BENCH_SINGLE_DEVICE=1 timeout 400 python bench.py --gpus 2 --steps 3 --warmup 1 2>&1 | tail -4 | cut -c1-600
