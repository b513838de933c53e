#!/bin/bash
# counters for the 70 -> 70 launches of tools/kbench.py, tiled kernel (MMLF_CONV_RS=0) against the register-streamed one
# (MMLF_CONV_RS=1): separate --pmc passes; summary -> $OUT/summary.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
. tools/outdir.sh
export OUT=$(new_outdir pmc_narrow)      # a directory of its own per call: a retry never overwrites a failed run's logs
python3 -c "from mmlf_amd import _lib; print(_lib.build_info())" > $OUT/build.txt 2>&1
env | grep -E '^(AMD_|HSA_|HIP_|MMLF_|KBENCH_)' | sort > $OUT/env.txt
export KBENCH_ONLY70=1
pass() {  # rs name counters...
  rs=$1; name=$2; shift; shift
  MMLF_CONV_RS=$rs timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/rs$rs$name -- python3 tools/kbench.py 512 3 pmc > $OUT/rs$rs$name.log 2>&1
}
for rs in 0 1; do
  pass $rs a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
  pass $rs b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE
  pass $rs d TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
  pass $rs f FETCH_SIZE
  pass $rs w WRITE_SIZE
done
python3 - <<'PY'
import csv, glob, json, collections
import os; OUT = os.environ['OUT']
out = collections.defaultdict(dict)
for rs in '01':
  for d in 'abdfw':
    try:
        rows = list(csv.DictReader(open(glob.glob(f'{OUT}/rs{rs}{d}/*/*_counter_collection.csv')[0])))
        kt = {r['Dispatch_Id']: r for r in csv.DictReader(open(glob.glob(f'{OUT}/rs{rs}{d}/*/*_kernel_trace.csv')[0]))}
    except Exception as e:
        print('pass', rs, d, 'failed', e); continue
    s = collections.defaultdict(float); n = collections.defaultdict(set); dur = collections.defaultdict(float)
    for r in rows:
        k = r['Kernel_Name'].split('(')[0]
        if 'conv4tap' not in k: continue
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
