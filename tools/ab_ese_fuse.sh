#!/bin/bash
# Round-6 timing proxy for a FUSED evaluation stream block: BASELINE.json configs[4] (one 512x512 light field,
# 70-member Ensamble) with the product library against a -DMMLF_ABL_RS_FUSE=1 build, in which the register-streamed narrow kernel
# stores nothing in a block's first convolution and loads no activations in its second (WRONG results by construction; the loader
# needs MMLF_ALLOW_ABLATION=1).  That is the LEAST a fused conv(p1)+ReLU+conv(p0) could cost -- the exchange of the intermediate
# through LDS and the second filter's streaming are not counted.  Interleaved, twice; then the per-launch times of the narrow
# kernel under rocprofv3 --kernel-trace for both builds.     gpurun -- bash tools/ab_ese_fuse.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
. tools/outdir.sh
export OUT=$(new_outdir ab_ese_fuse)
bash tools/build_variant.sh rsfuse -DMMLF_ABL_RS_FUSE=1 > $OUT/build.log 2>&1 || { cat $OUT/build.log; exit 1; }
for rep in 1 2; do
  for v in default rsfuse; do
    if [ $v = default ]; then unset MMLF_HIP_LIB MMLF_ALLOW_ABLATION; else export MMLF_HIP_LIB=variants/lib_rsfuse.so MMLF_ALLOW_ABLATION=1; fi
    echo "## $v (rep $rep)" | tee -a $OUT/ab.log
    timeout -k 10 300 python3 tools/ese_bench.py 512 2>> $OUT/ab.err | grep ESE | tee -a $OUT/ab.log || exit 1
  done
done
for v in default rsfuse; do
  if [ $v = default ]; then unset MMLF_HIP_LIB MMLF_ALLOW_ABLATION; else export MMLF_HIP_LIB=variants/lib_rsfuse.so MMLF_ALLOW_ABLATION=1; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$v -- python3 tools/ese_bench.py 512 > $OUT/trace_$v.log 2>&1 || { tail -5 $OUT/trace_$v.log; exit 1; }
done
python3 - <<'PY' | tee -a $OUT/ab.log
import csv, glob, os, collections
OUT = os.environ['OUT']
for v in ('default', 'rsfuse'):
    rows = list(csv.DictReader(open(glob.glob(f'{OUT}/trace_{v}/*/*_kernel_trace.csv')[0])))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    rs = [(r['Kernel_Name'].split('(')[0], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6) for r in rows if 'conv4tap_rs_kernel' in r['Kernel_Name']]
    n = len(rs) // 2
    rs = rs[n:]                     # the second (timed) scene: 4 streams x 6 launches
    per = collections.defaultdict(list)
    for i, (k, ms) in enumerate(rs):
        per[i % 6].append(ms)
    names = ['b0 conv1 27->70 (p1)', 'b0 conv2 (p0, folded BN)', 'b1 conv1 (p1)', 'b1 conv2 (p0)', 'b2 conv1 (p1)', 'b2 conv2 (p0)']
    print(f'## {v}: narrow-kernel launches of one scene, mean over the four streams (ms)')
    for i in range(6):
        print(f'  {names[i]:28s} {sum(per[i]) / len(per[i]):6.3f}   ({len(per[i])} launches)')
    print(f'  sum per scene {sum(ms for _, ms in rs):7.2f} ms')
    wide = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6 for r in rows if 'conv4tap_x6s_kernel<18' in r['Kernel_Name']]
    print(f'  wide launches: {len(wide)} at {sum(wide) / len(wide):.3f} ms')
PY
