#!/bin/bash
# 4K (3840 x 2160, D = 192), few host threads: resident (vertex preparation on the GPU) against the pool preparing the vertex orders, subtree sizes
#   bash tools/sweep_4k.sh > gpurun_out/sweep_4k.txt
B="--workload 4k_d192 --steps 4 --warmup 1 --no-configs --host-share 0 --no-kernel-timing --no-host --no-latency --no-real --cpu-sample 0 --no-gate"
val() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['host_cpu_cores_busy'], d['config']['engine'])"; }
run() {  # resident submax workers
    SV_RESIDENT=$1 SV_DG_SUBMAX=$2 python3 bench.py $B --workers $3 2>/dev/null | val "resident $1 subtree size $2 workers $3:" || exit 1
}
run 0 1024 2
run 0 700 2
run 0 512 2
run 0 512 1
run 0 512 3
run 0 512 4
run 1 1024 2
