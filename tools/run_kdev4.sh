cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
NCCL_DEBUG=WARN timeout 600 python -m pytest tests/test_sharded_gloo.py -m gpu -x -q 2>&1 | tail -5
