set -x
mkdir -p gpurun_out/kdev
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pairing or sipp or golden" 2>&1 | tail -5
timeout 300 python bench.py --steps 3 --warmup 1 --cpu-log-n 0 2>/dev/null | tee gpurun_out/kdev/bench_new.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d['roofline']['int_alu']['achieved'], d['phase_ms'])"
RIPP_LP_ONE_LANE=1 timeout 300 python bench.py --steps 3 --warmup 1 --cpu-log-n 0 2>/dev/null | tee gpurun_out/kdev/bench_old.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d['roofline']['int_alu']['achieved'], d['phase_ms'])"
RIPP_TRACE=1 timeout 300 python bench.py --steps 1 --warmup 1 --cpu-log-n 0 2>&1 | grep "ripp\]" | tail -24
