#!/bin/bash
# KITTI headline: pairs per launch x slots (default 64 x 8).   bash tools/sweep_chunk.sh > gpurun_out/sweep_chunk.txt
B="--steps 20 --warmup 3 --no-configs --host-share 0 --no-kernel-timing --no-host --no-latency --no-real --cpu-sample 0 --no-gate"
val() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['config']['engine'].get('gpu_triangulation_share'))"; }
for cs in "64 8" "128 6" "128 8" "96 8" "48 10" "32 12"; do
    set -- $cs
    python3 bench.py $B --chunk $1 --slots $2 --batch 384 2>/dev/null | val "chunk $1 slots $2:" || exit 1
done
