#!/bin/bash
# Builds variant libraries for kernel ablation experiments: tools/ablate.sh NAME "-DABL_X=1 ..." -> abl_tmp/lib_NAME.so
# ABL_SRC=path/to/patched_copy.hip compiles a patched copy instead of the tree's kernels.hip (never patch the tree for ablations)
# (run a variant with SV_LIB_PATH=abl_tmp/lib_NAME.so; abl_tmp/ travels to the GPU box, gpurun_out/ does not)
set -e
cd "$(dirname "$0")/.."
PKG=low-cost-hardware-accelerated-vision-based-depth-perception-for-real-time-applications_amd
NAME=$1; shift
mkdir -p abl_tmp
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-rdc \
    -Iinclude -I$PKG/csrc $@ -c ${ABL_SRC:-$PKG/csrc/kernels.hip} -o abl_tmp/kernels_$NAME.o
OBJS=$(ls $PKG/build/*.o | grep -v "/kernels.hip.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o abl_tmp/lib_$NAME.so abl_tmp/kernels_$NAME.o $OBJS -lpthread
rm abl_tmp/kernels_$NAME.o
echo built abl_tmp/lib_$NAME.so
