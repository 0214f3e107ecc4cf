#!/bin/bash
# Hardware queues per process (GPU_MAX_HW_QUEUES): the handle drives 12 streams (phase 1, 4 filter / triangulation, 4 phase 2, upload, 2 download);
# streams that share a hardware queue serialise.  bash tools/queue_sweep.sh "8 12 16" > gpurun_out/queues.txt
B="--no-configs --host-share 0 --no-kernel-timing --no-host --no-latency --no-real --cpu-sample 0 --no-gate"
val() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'])"; }
for q in ${1:-8 12 16}; do
    export GPU_MAX_HW_QUEUES=$q
    python3 bench.py --workload 4k_d192 --workers 2 --steps 4 --warmup 1 $B 2>/dev/null | val "queues $q: 4k, 2 host threads" || exit 1
    python3 bench.py --workers 1 --steps 20 --warmup 3 $B 2>/dev/null | val "queues $q: kitti, 1 host thread" || exit 1
    python3 bench.py --steps 20 --warmup 3 $B 2>/dev/null | val "queues $q: kitti, default" || exit 1
done
