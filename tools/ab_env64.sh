#!/bin/bash
# A/B of the 64-patches-per-GPU step (the share of one GPU at N = 8) by environment switch, interleaved on one box:
#   tools/ab_env64.sh TAG VAR VALUE_A VALUE_B ...
. "$(dirname "$0")/outdir.sh"
out=$(new_outdir "$(basename "$1" .log)")/ab.log      # gpurun_out/TAG_<unix time>/ab.log: never an existing file
echo "-> $out"
var=$2; shift; shift
for rep in 1 2 3; do
  for v in "$@"; do
    env $var=$v timeout -k 10 200 python bench.py --global-batch 64 --steps 20 --warmup 3 --no-cpu-baseline --no-f32-leg --no-extra-legs 2>> "$out.err" | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$var=$v', 'bs64', d['value'], 'ms', d['ms_per_step'], flush=True)" | tee -a $out || exit 1
  done
done
