cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_sharded_gloo.py tests/test_gpu_tipa.py -m gpu -x -q 2>&1 | tail -30
