#!/bin/bash
# Run one command on the GPU box with everything it prints kept in a directory of its own:
#   gpurun -- bash tools/gpu_call.sh TAG 'command ...'
# -> gpurun_out/TAG_<unix time>/{log.txt (stdout + stderr), env.txt (AMD_* HSA_* HIP_* MMLF_* KBENCH_* in effect), build.txt
#    (mmlf_build_info of the library the command loads), rc.txt}; $OUT is exported to the command.  The directory is never reused.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
. tools/outdir.sh
tag=$1; shift
export OUT=$(new_outdir "$tag")
env | grep -E '^(AMD_|HSA_|HIP_|MMLF_|KBENCH_)' | sort > "$OUT/env.txt"
python3 -c "from mmlf_amd import _lib; print(_lib.build_info())" > "$OUT/build.txt" 2>&1
echo "[gpu_call] $OUT: $*"
bash -o pipefail -c "$*" 2>&1 | tee "$OUT/log.txt"
rc=${PIPESTATUS[0]}
echo $rc > "$OUT/rc.txt"
echo "[gpu_call] rc=$rc -> $OUT"
exit $rc
