#!/bin/bash
# Socket power and clocks sampled beside a running bench.py main leg (rocm-smi; 5 samples per second).
#   gpurun -- bash tools/power_trace.sh gpurun_out/power_trace.log
. "$(dirname "$0")/outdir.sh"
out=$(new_outdir "$(basename "${1:-power_trace}" .log)")/power_trace.log      # a directory of its own per call
echo "-> $out"
rocm-smi --showpower --showclocks --showmaxpower --showperflevel > "$out.idle" 2>&1
python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-f32-leg --no-extra-legs > "$out.bench.json" 2> "$out.bench.err" &
pid=$!
: > "$out"
while kill -0 $pid 2>/dev/null; do
  { date +%s.%N; rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Power|sclk|mclk|fclk'; } >> "$out"
  sleep 0.2
done
wait $pid
tail -c 400 "$out.bench.json"
