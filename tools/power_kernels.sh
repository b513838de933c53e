#!/bin/bash
# Socket power per launch kind: tools/kbench.py with long timed loops (KBENCH_STAMP=1 prints each loop's wall-clock window), rocm-smi
# sampled beside it, samples averaged per window.   gpurun -- bash tools/power_kernels.sh gpurun_out/power_kernels.log [reps]
# The environment the launches run under is part of the record (round 4's log was made with AMD_SERIALIZE_KERNEL=3 set by
# hand, which this script did not say): the script sets it itself now -- override by exporting another value -- and writes
# every AMD_* / HSA_* / HIP_* / MMLF_* variable in effect to $out.env, which the summary repeats in its first line.
. "$(dirname "$0")/outdir.sh"
out=$(new_outdir "$(basename "${1:-power_kernels}" .log)")/power_kernels.log      # a directory of its own per call
echo "-> $out"
reps=${2:-250}
export AMD_SERIALIZE_KERNEL=${AMD_SERIALIZE_KERNEL:-3}
env | grep -E '^(AMD_|HSA_|HIP_|MMLF_|KBENCH_)' | sort > "$out.env"
KBENCH_STAMP=1 python tools/kbench.py 512 $reps power > "$out.kbench" 2> "$out.err" &
pid=$!
: > "$out.smi"
while kill -0 $pid 2>/dev/null; do
  { date +%s.%N; rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Power|sclk'; } >> "$out.smi"
  sleep 0.15
done
wait $pid
python3 - "$out" <<'PY'
import re, sys
out = sys.argv[1]
samples = []
for b in re.split(r'\n(?=\d{10}\.\d+)', open(out + '.smi').read()):
    m = re.match(r'(\d+\.\d+)', b); p = re.search(r'Power \(W\): ([\d.]+)', b); s = re.search(r'sclk clock level: \S+ \((\d+)Mhz\)', b)
    if m and p and s:
        samples.append((float(m.group(1)), float(p.group(1)), int(s.group(1))))
with open(out, 'w') as f:
    f.write('# environment: ' + ' '.join(open(out + '.env').read().split()) + '\n')
    for line in open(out + '.kbench'):
        m = re.search(r'window (\d+\.\d+) (\d+\.\d+)', line)
        if not m:
            continue
        t0, t1 = float(m.group(1)), float(m.group(2))
        inside = [s for s in samples if t0 + 0.3 <= s[0] <= t1 - 0.1]       # (rocm-smi's reading lags by a sample)
        txt = line.split('  window')[0].rstrip()
        if inside:
            txt += '   power %4.0f W (%4.0f-%4.0f, %d samples)  sclk %4.0f MHz' % (
                sum(s[1] for s in inside) / len(inside), min(s[1] for s in inside), max(s[1] for s in inside), len(inside),
                sum(s[2] for s in inside) / len(inside))
        f.write(txt + '\n')
print(open(out).read())
PY
