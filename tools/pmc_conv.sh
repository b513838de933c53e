#!/bin/bash
# deeper counters for the dominant kernels, on the micro-benchmark (tools/kbench.py): separate --pmc passes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
. tools/outdir.sh
export OUT=$(new_outdir pmc_conv)      # a directory of its own per call: a retry never overwrites a failed run's logs
python3 -c "from mmlf_amd import _lib; print(_lib.build_info())" > $OUT/build.txt 2>&1
env | grep -E '^(AMD_|HSA_|HIP_|MMLF_|KBENCH_)' | sort > $OUT/env.txt
export KBENCH_ONLY280=1
pass() {  # name counters...
  name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 tools/kbench.py 512 3 pmc > $OUT/$name.log 2>&1
}
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
pass b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE
pass c SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
pass d TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
python3 - <<'PY'
import csv, glob, json, collections
import os; OUT = os.environ['OUT']
out = collections.defaultdict(dict)
for d in 'abcd':
    try:
        rows = list(csv.DictReader(open(glob.glob(f'{OUT}/{d}/*/*_counter_collection.csv')[0])))
        kt = {r['Dispatch_Id']: r for r in csv.DictReader(open(glob.glob(f'{OUT}/{d}/*/*_kernel_trace.csv')[0]))}
    except Exception as e:
        print('pass', d, 'failed', e); continue
    s = collections.defaultdict(float); n = collections.defaultdict(set); dur = collections.defaultdict(float)
    for r in rows:
        k = r['Kernel_Name'].split('(')[0]
        if 'conv4tap' not in k and 'wgrad4tap' not in k: continue
        s[(k, r['Counter_Name'])] += float(r['Counter_Value'])
        if r['Dispatch_Id'] not in n[k]:
            n[k].add(r['Dispatch_Id']); t = kt[r['Dispatch_Id']]; dur[k] += int(t['End_Timestamp']) - int(t['Start_Timestamp'])
    for (k, c), v in s.items():
        out[k][c] = v / len(n[k])
        out[k][f'avg_ns_{d}'] = dur[k] / len(n[k])
json.dump(out, open(OUT + '/summary.json', 'w'), indent=1)
for k, v in out.items():
    print(k); print(json.dumps(v, indent=1))
PY
