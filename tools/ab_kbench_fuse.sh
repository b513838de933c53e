#!/bin/bash
# tools/kbench_fuse.py with the product library and the -DMMLF_ABL_RS_FUSE=1 build, interleaved, three times.
#   gpurun -- bash tools/ab_kbench_fuse.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
. tools/outdir.sh
export OUT=$(new_outdir ab_kbench_fuse)
bash tools/build_variant.sh rsfuse -DMMLF_ABL_RS_FUSE=1 > $OUT/build.log 2>&1 || { cat $OUT/build.log; exit 1; }
for rep in 1 2 3; do
  for v in default rsfuse; do
    if [ $v = default ]; then unset MMLF_HIP_LIB MMLF_ALLOW_ABLATION; else export MMLF_HIP_LIB=variants/lib_rsfuse.so MMLF_ALLOW_ABLATION=1; fi
    timeout -k 10 200 python3 tools/kbench_fuse.py 8 9 2>> $OUT/ab.err | tee -a $OUT/ab.log || exit 1
  done
done
