cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sipp" 2>&1 | tail -3
RIPP_TRACE=1 timeout 300 python bench.py --steps 3 --warmup 1 --cpu-log-n 0 2>&1 | grep "ripp\]\|metric" | tail -14 | cut -c1-330
