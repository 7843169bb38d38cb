#!/bin/bash
# crossover sweep of the latency-form (VM) thresholds: two proofs per setting, report the second
for cfg in "8192 2048 16384 16384" "8192 4096 16384 16384" "8192 8192 16384 16384" "8192 16384 16384 16384" "8192 32768 16384 16384" "16384 8192 16384 16384" "16384 16384 32768 16384"; do
  set -- $cfg
  echo -n "lines_max=$1 fold_max=$2 tree_max=$3 split_max=$4 : "
  RIPP_VM_LINES_MAX=$1 RIPP_VM_FOLD_MAX=$2 RIPP_VM_TREE_MAX=$3 RIPP_GLS_SPLIT_MAX=$4 python tools/gputest2.py 20 20 2>&1 | grep "n=2" | tail -1 | sed 's/.*prove //'
done
