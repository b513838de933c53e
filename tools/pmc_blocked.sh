#!/bin/bash
# Memory-side bytes of the round-6 proxy (tools/kbench_blocked.py): the tiled 70 -> 70 kernel on an NHWC input against the same
# kernel on a chunk-blocked copy.  Both halves run the same kernel names; per name the dispatches come NHWC first, blocked second,
# in equal numbers -- the summary splits them by order.     gpurun -- bash tools/pmc_blocked.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
. tools/outdir.sh
export OUT=$(new_outdir pmc_blocked)
python3 -c "from mmlf_amd import _lib; print(_lib.build_info())" > $OUT/build.txt 2>&1
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/r -- python3 tools/kbench_blocked.py 512 3 > $OUT/r.log 2>&1 || { tail -5 $OUT/r.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_WRREQ_WRITE_DRAM_32B_sum --kernel-trace --output-format csv -d $OUT/w -- python3 tools/kbench_blocked.py 512 3 > $OUT/w.log 2>&1 || { tail -5 $OUT/w.log; exit 1; }
python3 - <<'PY' | tee $OUT/summary.txt
import csv, glob, collections, os
OUT = os.environ['OUT']
print('# tools/pmc_blocked.sh: 70->70 launches of tools/kbench_blocked.py, bs=512: memory-side GB per launch (32-byte-unit counters), L2 hits / misses and read requests (millions); first half of each kernel\'s dispatches = NHWC input, second half = chunk-blocked copy')
for d in 'rw':
    rows = list(csv.DictReader(open(glob.glob(f'{OUT}/{d}/*/*_counter_collection.csv')[0])))
    kt = {r['Dispatch_Id']: r for r in csv.DictReader(open(glob.glob(f'{OUT}/{d}/*/*_kernel_trace.csv')[0]))}
    per = collections.defaultdict(lambda: collections.defaultdict(dict))          # kernel -> dispatch -> counter -> value
    for r in rows:
        k = r['Kernel_Name'].split('(')[0]
        if 'conv4tap_x6s_kernel<5' not in k: continue
        per[k][int(r['Dispatch_Id'])][r['Counter_Name']] = per[k][int(r['Dispatch_Id'])].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    for k in sorted(per):
        ids = sorted(per[k])
        half = len(ids) // 2
        for tag, sel in (('NHWC   ', ids[:half]), ('blocked', ids[half:])):
            acc = collections.defaultdict(float); dur = 0.0
            for i in sel:
                for c, v in per[k][i].items(): acc[c] += v
                t = kt[str(i)]; dur += int(t['End_Timestamp']) - int(t['Start_Timestamp'])
            vals = {c: round(v / len(sel) * (32 / 1e9 if '32B' in c else 1e-6), 3) for c, v in acc.items()}
            print(f'{k:52s} {tag} {len(sel):2d} launches {dur / len(sel) / 1e6:.3f} ms', vals, flush=True)
PY
