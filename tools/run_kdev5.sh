cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bls12_377.py -m gpu -x -q 2>&1 | tail -30
