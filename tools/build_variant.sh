#!/bin/bash
# A library variant for kernel experiments: the three .hip sources compiled with extra -D flags, linked with the current objects of the host sources.
#   bash tools/build_variant.sh NAME [-DFOO=1 ...]   ->  abl_tmp/lib_NAME.so   (then SV_LIB_PATH=abl_tmp/lib_NAME.so tools/ktime.py, tools/pmc_ab.sh ...)
set -e
cd "$(dirname "$0")/.."
P=$(ls -d *_amd)
NAME=$1; shift
mkdir -p abl_tmp
OBJS=""
for f in kernels delaunay_gpu legacy_kernels; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-value --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-rdc "$@" -c $P/csrc/$f.hip -o abl_tmp/${f}_$NAME.o &
    OBJS="$OBJS abl_tmp/${f}_$NAME.o"
done
wait
HOSTOBJS=$(ls $P/build/*.o | grep -v "\.hip\.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o abl_tmp/lib_$NAME.so $OBJS $HOSTOBJS -lpthread -ldl
rm -f $OBJS
echo abl_tmp/lib_$NAME.so
