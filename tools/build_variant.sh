#!/bin/bash
# Build another copy of the library for A/B runs:  tools/build_variant.sh NAME [-DFLAG ...]
# -> variants/lib_NAME.so (git-ignored AND gpurun-ignored: run this script on the GPU box, inside the gpurun command, in front of
#    the A/B; hipcc is there and a build takes 25 s); select it with MMLF_HIP_LIB=variants/lib_NAME.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p variants
# the variant says what it is built from (mmlf_build_info: git= is 'unknown' on the GPU box's snapshot, which has no .git; src= is
# the content hash of the kernel sources and always known)
GIT=$(python3 -m mmlf_amd.csrc.build --source-revision)
SRC=$(python3 -m mmlf_amd.csrc.build --source-hash)
FL="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-function -ffp-contract=off -DMMLF_GIT_HASH=\"$GIT\" -DMMLF_SRC_HASH=\"$SRC\""
/opt/rocm/bin/hipcc $FL "$@" -c mmlf_amd/csrc/conv.hip -o variants/conv_$name.o &
/opt/rocm/bin/hipcc $FL "$@" -c mmlf_amd/csrc/wgrad.hip -o variants/wgrad_$name.o &
/opt/rocm/bin/hipcc $FL "$@" -c mmlf_amd/csrc/elementwise.hip -o variants/ew_$name.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC variants/conv_$name.o variants/wgrad_$name.o variants/ew_$name.o -o variants/lib_$name.so
rm -f variants/conv_$name.o variants/wgrad_$name.o variants/ew_$name.o
echo variants/lib_$name.so
