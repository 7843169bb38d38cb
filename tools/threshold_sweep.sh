# A/B of a size threshold on one box: alternating runs of bench.py with and without the override.  Usage (GPU box): bash tools/threshold_sweep.sh NAME=VALUE [rounds]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/t/sweep
V=${1:-RIPP_TAIL_PIPE_MAX=4096}; N=${2:-4}
for i in $(seq 1 $N); do
  env X=1 timeout 200 python bench.py --steps 10 --warmup 2 --cpu-log-n 0 > gpurun_out/t/sweep/ab_base_$i.json 2> /dev/null
  env $V timeout 200 python bench.py --steps 10 --warmup 2 --cpu-log-n 0 > gpurun_out/t/sweep/ab_var_$i.json 2> /dev/null
done
