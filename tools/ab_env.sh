# A/B of one environment setting on the headline bench, alternating runs on the same box: bash tools/ab_env.sh NAME=VALUE [pairs]
mkdir -p gpurun_out/ab
n=${2:-3}
for i in $(seq 1 $n); do
  python bench.py --steps 5 --cpu-log-n 0 > gpurun_out/ab/base_$i.json 2>/dev/null
  env "$1" python bench.py --steps 5 --cpu-log-n 0 > gpurun_out/ab/with_$i.json 2>/dev/null
done
python - "$1" <<'P'
import json,glob,sys
print("A/B", sys.argv[1])
for f in sorted(glob.glob('gpurun_out/ab/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d['ms_per_step'],1), d['ms_per_step_all'], 'hash', d['statement_hash_ms'], 'post', round(d['post_hash_ms'],1), 'wait', d['phase_ms']['hash_wait_ms'], 'fold', d['phase_ms']['fold_ms'])
P
