#!/bin/bash
# A/B of the 64-patches-per-GPU step (the share of one GPU at N = 8) by environment switch, interleaved on one box:
#   tools/ab_env64.sh OUT.log VAR VALUE_A VALUE_B ...
out=$1; var=$2; shift; shift
for rep in 1 2 3; do
  for v in "$@"; do
    env $var=$v timeout -k 10 200 python bench.py --global-batch 64 --steps 20 --warmup 3 --no-cpu-baseline --no-f32-leg --no-extra-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$var=$v', 'bs64', d['value'], 'ms', d['ms_per_step'], flush=True)" >> $out || exit 1
  done
done
