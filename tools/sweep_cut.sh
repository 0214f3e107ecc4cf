#!/bin/bash
# KITTI headline, every triangulation on the GPU from host-prepared vertex orders (round-3 path, 14 pool threads prepare): whole sets in LDS
# (subtree size 4000) against the cut path (subtrees in LDS, the top merges on a mesh in global memory)
#   bash tools/sweep_cut.sh > gpurun_out/sweep_cut.txt
B="--steps 20 --warmup 3 --no-configs --host-share 0 --no-kernel-timing --no-host --no-latency --no-real --cpu-sample 0 --no-gate"
val() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['host_cpu_cores_busy'], d['config']['engine'])"; }
for sm in ${1:-4000 1100 600 300}; do
    SV_RESIDENT=0 SV_GPU_DELAUNAY=1 SV_DG_SUBMAX=$sm python3 bench.py $B 2>/dev/null | val "subtree size $sm:" || exit 1
done
