#!/bin/bash
# crossover sweep of the latency-form (VM) thresholds: bench.py, 3 steps per setting
for cfg in "32768 2048 16384" "32768 2048 65536" "32768 2048 262144" "32768 2048 1048576" "32768 2048 16777216"; do
  set -- $cfg
  echo -n "lines_max=$1 fold_max=$2 tree_max=$3 : "
  RIPP_VM_LINES_MAX=$1 RIPP_VM_FOLD_MAX=$2 RIPP_VM_TREE_MAX=$3 python bench.py --steps 3 --warmup 1 --cpu-log-n 0 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f ms' % d['ms_per_step'], {k: v for k, v in d['phase_ms'].items() if k in ('miller_products_ms','fold_ms')})"
done
