cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/scal377
timeout 600 python -m pytest tests/test_gpu_bls12_377.py -m gpu -x -q -k "verbatim" 2>&1 | tail -3
timeout 900 python tools/scaling_ipp.py 4 18 gpurun_out/scal377 --curve 377 --cpu-max 12 2>&1 | tail -20
timeout 600 python tools/scaling_ipp.py 10 10 gpurun_out/scal377 --curve 377 --repeated --cpu-max 10 2>&1 | tail -3
