#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
. tools/outdir.sh
export OUT=$(new_outdir pmc_traffic)      # a directory of its own per call: a retry never overwrites a failed run's logs
python3 -c "from mmlf_amd import _lib; print(_lib.build_info())" > $OUT/build.txt 2>&1
env | grep -E '^(AMD_|HSA_|HIP_|MMLF_|KBENCH_)' | sort > $OUT/env.txt
CMD="python bench.py --steps 1 --warmup 1 --no-cpu-baseline"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.log 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- $CMD > $OUT/write.log 2>&1
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $OUT/mfma -- $CMD > $OUT/mfma.log 2>&1
python - <<'PY'
import csv, glob, json, collections
import os; OUT = os.environ['OUT']
def agg(d, counter):
    rows = list(csv.DictReader(open(glob.glob(f'{OUT}/{d}/*/*_counter_collection.csv')[0])))
    kt = {r['Dispatch_Id']: r for r in csv.DictReader(open(glob.glob(f'{OUT}/{d}/*/*_kernel_trace.csv')[0]))}
    s = collections.defaultdict(float); n = collections.defaultdict(set); dur = collections.defaultdict(float)
    for r in rows:
        if r['Counter_Name'] != counter: continue
        k = r['Kernel_Name'].split('(')[0]
        s[k] += float(r['Counter_Value'])
        if r['Dispatch_Id'] not in n[k]:
            n[k].add(r['Dispatch_Id']); t = kt[r['Dispatch_Id']]; dur[k] += int(t['End_Timestamp']) - int(t['Start_Timestamp'])
    return {k: (s[k] / len(n[k]), len(n[k]), dur[k] / len(n[k])) for k in s}
f = agg('fetch', 'FETCH_SIZE'); w = agg('write', 'WRITE_SIZE')
m = agg('mfma', 'SQ_VALU_MFMA_BUSY_CYCLES'); g = agg('mfma', 'GRBM_GUI_ACTIVE')
out = {}
for k in f:
    if k in w:
        out[k] = {'launches': f[k][1], 'FETCH_SIZE_KB_raw': f[k][0], 'WRITE_SIZE_KB': w[k][0],
                  'hbm_bytes_per_launch': (2 * f[k][0] + w[k][0]) * 1024, 'avg_ns': f[k][2]}
        if k in m:
            out[k]['mfma_busy_frac'] = m[k][0] / (g[k][0] / 8 * 1024)
            out[k]['clock_ghz'] = g[k][0] / 8 / m[k][2]
json.dump(out, open(OUT + '/summary.json', 'w'), indent=1)
for k in sorted(out, key=lambda k: -out[k]['avg_ns'] * out[k]['launches'])[:6]:
    print(k, out[k])
PY
