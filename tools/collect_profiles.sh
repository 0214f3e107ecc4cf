# Collects the round's profiles on the GPU box (tools/collect_profiles.sh TAG -> gpurun_out/TAG_*.csv; copy what is to be judged into profiles/).
# Serial pass: one slot, one stream, 32 pairs per launch, 18 launches per kernel (medians in the summary); the PMC counters in their own passes.
set -e
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=profile_run.py
if [ "$2" != "configs-only" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p_kt -- python3 tools/$R --chunk 32 --batch 64 > gpurun_out/p_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/p_f -- python3 tools/$R --chunk 32 --batch 64 --reps 3 > gpurun_out/p_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/p_w -- python3 tools/$R --chunk 32 --batch 64 --reps 3 > gpurun_out/p_w.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/p_sq -- python3 tools/$R --chunk 32 --batch 64 --reps 3 > gpurun_out/p_sq.log 2>&1
python tools/summarize_rocprof.py --stats gpurun_out/p_kt --pmc gpurun_out/p_f gpurun_out/p_w gpurun_out/p_sq --pairs-per-launch 32 -o gpurun_out/${TAG}_serial_kernel_stats_pmc.csv
# the default pipelined configuration (what `value` is measured in), and the same with ONE host thread (every triangulation resident on the GPU)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p_b -- python3 bench.py --steps 20 --warmup 2 --min-seconds 0 --cpu-sample 0 --no-latency --no-host --no-real --no-kernel-timing --no-configs --no-gate --host-share 0 > gpurun_out/p_b.log 2>&1
python tools/summarize_rocprof.py --stats gpurun_out/p_b --pairs-per-launch 64 -o gpurun_out/${TAG}_bench_pipelined_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p_g -- python3 bench.py --workers 1 --steps 20 --warmup 2 --min-seconds 0 --cpu-sample 0 --no-latency --no-host --no-real --no-kernel-timing --no-configs --no-gate --host-share 0 > gpurun_out/p_g.log 2>&1
python tools/summarize_rocprof.py --stats gpurun_out/p_g --pairs-per-launch 64 -o gpurun_out/${TAG}_bench_pipelined_one_host_thread_kernel_stats.csv
rm -rf gpurun_out/p_kt gpurun_out/p_f gpurun_out/p_w gpurun_out/p_sq gpurun_out/p_b gpurun_out/p_g
tail -c 300 gpurun_out/p_b.log; tail -c 300 gpurun_out/p_g.log
fi
# BASELINE.json's other configurations, no-overlap passes (kernel trace only): D = 256 at KITTI size, and one 4K D = 192 chunk with the
# triangulations resident on the GPU (the cut path's kernels) - tools/collect_profiles.sh TAG configs | configs-only
if [ "$2" = "configs" ] || [ "$2" = "configs-only" ]; then
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p_d256 -- python3 tools/$R --chunk 32 --batch 64 --disp 256 > gpurun_out/p_d256.log 2>&1
    python tools/summarize_rocprof.py --stats gpurun_out/p_d256 --pairs-per-launch 32 -o gpurun_out/${TAG}_serial_d256_kernel_stats.csv
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p_4k -- python3 tools/profile_4k.py --reps 9 > gpurun_out/p_4k.log 2>&1
    python tools/summarize_rocprof.py --stats gpurun_out/p_4k --pairs-per-launch 16 -o gpurun_out/${TAG}_serial_4k_d192_resident_kernel_stats.csv
    rm -rf gpurun_out/p_d256 gpurun_out/p_4k
fi
