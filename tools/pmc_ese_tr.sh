#!/bin/bash
# The 288-column tiled kernel's ReLU launch on full frames (two-segment activation windows, tools/ese_bench.py 512), the
# transposed orientation forced (MMLF_CONV_TR=2) against the default rule (=1: the other orientation there): PMC passes that
# say where the 10 % go (wave cycles and waits, instruction mix, LDS conflicts, memory-side bytes).
#   gpurun -- bash tools/pmc_ese_tr.sh       -> $OUT/summary.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
. tools/outdir.sh
export OUT=$(new_outdir pmc_ese_tr)      # a directory of its own per call: a retry never overwrites a failed run's logs
python3 -c "from mmlf_amd import _lib; print(_lib.build_info())" > $OUT/build.txt 2>&1
env | grep -E '^(AMD_|HSA_|HIP_|MMLF_|KBENCH_)' | sort > $OUT/env.txt
pass() {  # tr name counters...
  tr=$1; name=$2; shift; shift
  MMLF_CONV_TR=$tr timeout -k 10 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/tr$tr$name -- python3 tools/ese_bench.py 512 > $OUT/tr$tr$name.log 2>&1 || exit 1
}
for tr in 1 2; do
  pass $tr a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
  pass $tr b SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM
  pass $tr c SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM
  pass $tr d TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum TCC_HIT_sum TCC_MISS_sum
  pass $tr e TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
done
python3 - <<'PY' > $OUT/summary.txt
import csv, glob, collections
import os; OUT = os.environ['OUT']
res = collections.defaultdict(dict)
for tr in '12':
  for d in 'abcde':
    try:
        rows = list(csv.DictReader(open(glob.glob(f'{OUT}/tr{tr}{d}/*/*_counter_collection.csv')[0])))
        kt = {r['Dispatch_Id']: r for r in csv.DictReader(open(glob.glob(f'{OUT}/tr{tr}{d}/*/*_kernel_trace.csv')[0]))}
    except Exception as e:
        print('pass', tr, d, 'failed', e); continue
    s = collections.defaultdict(float); n = set(); dur = 0
    for r in rows:
        if 'conv4tap_x6s_kernel<18' not in r['Kernel_Name']: continue
        s[r['Counter_Name']] += float(r['Counter_Value'])
        if r['Dispatch_Id'] not in n:
            n.add(r['Dispatch_Id']); t = kt[r['Dispatch_Id']]; dur += int(t['End_Timestamp']) - int(t['Start_Timestamp'])
            res[tr]['kernel'] = r['Kernel_Name'].split('(')[0]
    for c, v in s.items():
        res[tr][c] = v / len(n)
    res[tr][f'avg_ms_{d}'] = dur / len(n) / 1e6
keys = sorted(set(res['1']) | set(res['2']))
print(f"{'counter':40s} {'TR=1 (rule: other orientation)':>32s} {'TR=2 (transposed forced)':>28s}  ratio")
for k in keys:
    a, b = res['1'].get(k), res['2'].get(k)
    if isinstance(a, str) or isinstance(b, str):
        print(f'{k:40s} {a!s:>32s} {b!s:>28s}'); continue
    print(f"{k:40s} {a if a is not None else float('nan'):32.4g} {b if b is not None else float('nan'):28.4g}  {(b / a) if a and b else float('nan'):.3f}")
PY
cat $OUT/summary.txt
