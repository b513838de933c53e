#!/bin/bash
# A/B of whole-step rates on ONE box: tools/ab_bench.sh TAG LIB_A LIB_B ...   ("default" = the in-tree library)
# Each library runs bench.py (bs=512 BASE + the extra legs incl. shard64), twice, interleaved.
. "$(dirname "$0")/outdir.sh"
out=$(new_outdir "$(basename "$1" .log)")/ab.log      # gpurun_out/TAG_<unix time>/ab.log: never an existing file
echo "-> $out"
shift
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = default ]; then unset MMLF_HIP_LIB; else export MMLF_HIP_LIB=variants/lib_$v.so; fi
    timeout -k 10 400 python bench.py --no-cpu-baseline --no-f32-leg --steps 5 2>> "$out.err" | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', 'bs512', d['value'], 'conv_ms', d['roofline']['avg_ms'], 'wgrad_ms', d['roofline_wgrad']['avg_ms'], 'upr', d['upr']['value'], 'dpp', d['dpp']['value'], 'shard64', d['shard64']['value'], d['shard64']['vs_bs512'], 'ese', d['ese']['value'], flush=True)" | tee -a $out || exit 1
  done
done
