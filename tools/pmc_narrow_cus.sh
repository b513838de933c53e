#!/bin/bash
# Memory-side bytes (gfx950's 32-byte-unit counters) and L2 hits / misses of the 70 -> 70 launches of tools/kbench.py against the
# number of workgroups that share an XCD's L2: the sixteen-wave tiled kernel on 256 / 128 / 64 CUs, the eight-wave tiled kernel
# with 512 / 256 workgroups, the register-streamed kernel on 256 / 128 CUs (profiles/r05_pmc_narrow_vs_cus.log).
#   gpurun -- bash tools/pmc_narrow_cus.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
. tools/outdir.sh
export OUT=$(new_outdir pmc_narrow_cus)      # a directory of its own per call: a retry never overwrites a failed run's logs
python3 -c "from mmlf_amd import _lib; print(_lib.build_info())" > $OUT/build.txt 2>&1
env | grep -E '^(AMD_|HSA_|HIP_|MMLF_|KBENCH_)' | sort > $OUT/env.txt
export KBENCH_ONLY70=1
i=0
for cfg in "MMLF_CONV_CUS=256" "MMLF_CONV_CUS=128" "MMLF_CONV_CUS=64" \
           "MMLF_CONV_NW16=0 MMLF_CONV_RS=0 MMLF_CONV_CUS=256" "MMLF_CONV_NW16=0 MMLF_CONV_RS=0 MMLF_CONV_CUS=128" \
           "MMLF_CONV_RS=1 MMLF_CONV_CUS=256" "MMLF_CONV_RS=1 MMLF_CONV_CUS=128"; do
  i=$((i+1))
  ( export $cfg; timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_32B_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $OUT/r$i -- python3 tools/kbench.py 512 3 pmc > $OUT/r$i.log 2>&1 ) || exit 1
  ( export $cfg; timeout -k 10 300 rocprofv3 --pmc TCC_EA0_WRREQ_WRITE_DRAM_32B_sum --kernel-trace --output-format csv -d $OUT/w$i -- python3 tools/kbench.py 512 3 pmc > $OUT/w$i.log 2>&1 ) || exit 1
  echo "cfg $i: $cfg"
done
python3 - <<'PY'
import csv, glob, collections
import os; OUT = os.environ['OUT']
for i in range(1, 8):
  for d in 'rw':
    rows = list(csv.DictReader(open(glob.glob(f'{OUT}/{d}{i}/*/*_counter_collection.csv')[0])))
    kt = {r['Dispatch_Id']: r for r in csv.DictReader(open(glob.glob(f'{OUT}/{d}{i}/*/*_kernel_trace.csv')[0]))}
    s = collections.defaultdict(float); n = collections.defaultdict(set); dur = collections.defaultdict(float)
    for r in rows:
        k = r['Kernel_Name'].split('(')[0]
        if 'conv4tap' not in k: continue
        s[(k, r['Counter_Name'])] += float(r['Counter_Value'])
        if r['Dispatch_Id'] not in n[k]:
            t = kt[r['Dispatch_Id']]; n[k].add(r['Dispatch_Id']); dur[k] += int(t['End_Timestamp']) - int(t['Start_Timestamp'])
    for k in sorted(n):
        vals = {c: round(v / len(n[k]) * (32 / 1e9 if '32B' in c else 1e-6), 3) for (kk, c), v in s.items() if kk == k}
        print(f'cfg={i} {k:48s} {dur[k] / len(n[k]) / 1e6:.3f} ms', vals, flush=True)
PY
