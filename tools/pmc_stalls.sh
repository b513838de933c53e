#!/bin/bash
# wave-cycle breakdown of the conv / wgrad kernels (tools/bench_conv.py, B=512): where the waves' cycles go.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
. tools/outdir.sh
export OUT=$(new_outdir pmc_stalls)      # a directory of its own per call: a retry never overwrites a failed run's logs
python3 -c "from mmlf_amd import _lib; print(_lib.build_info())" > $OUT/build.txt 2>&1
env | grep -E '^(AMD_|HSA_|HIP_|MMLF_|KBENCH_)' | sort > $OUT/env.txt
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq -- python tools/bench_conv.py 512 > $OUT/sq.log 2>&1
python - <<'PY'
import csv, glob, json, collections
import os; OUT = os.environ['OUT']
rows = list(csv.DictReader(open(glob.glob(OUT + '/sq/*/*_counter_collection.csv')[0])))
kt = {r['Dispatch_Id']: r for r in csv.DictReader(open(glob.glob(OUT + '/sq/*/*_kernel_trace.csv')[0]))}
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set); dur = collections.defaultdict(float)
for r in rows:
    k = r['Kernel_Name'].split('(')[0] + ' grid=' + r.get('Grid_Size', '')
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Dispatch_Id'] not in n[k]:
        n[k].add(r['Dispatch_Id']); t = kt[r['Dispatch_Id']]; dur[k] += int(t['End_Timestamp']) - int(t['Start_Timestamp'])
out = {}
for k, c in acc.items():
    if 'conv4tap' not in k and 'wgrad4tap' not in k: continue
    wc = c['SQ_WAVE_CYCLES'] or 1.0
    L = len(n[k])
    out[k] = {'launches': L, 'avg_ms': dur[k] / L / 1e6,
              'wait_any': c['SQ_WAIT_ANY'] / wc, 'wait_inst_any': c['SQ_WAIT_INST_ANY'] / wc,
              'active_inst_any': c['SQ_ACTIVE_INST_ANY'] / wc, 'wait_inst_lds': c['SQ_WAIT_INST_LDS'] / wc,
              'mfma_busy_frac': c['SQ_VALU_MFMA_BUSY_CYCLES'] / (c['GRBM_GUI_ACTIVE'] / 8 * 1024),
              'clock_ghz': c['GRBM_GUI_ACTIVE'] / 8 / dur[k]}
json.dump(out, open(OUT + '/summary.json', 'w'), indent=1)
for k in sorted(out, key=lambda k: -out[k]['avg_ms']):
    print(k, {a: round(b, 3) for a, b in out[k].items()})
PY
