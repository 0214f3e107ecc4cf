set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=profile_run.py
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p_kt -- python3 tools/$R --chunk 32 --batch 64 > gpurun_out/p_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/p_f -- python3 tools/$R --chunk 32 --batch 64 > gpurun_out/p_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/p_w -- python3 tools/$R --chunk 32 --batch 64 > gpurun_out/p_w.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/p_sq -- python3 tools/$R --chunk 32 --batch 64 > gpurun_out/p_sq.log 2>&1
python tools/summarize_rocprof.py --stats gpurun_out/p_kt --pmc gpurun_out/p_f gpurun_out/p_w gpurun_out/p_sq --pairs-per-launch 32 -o gpurun_out/serial_kernel_stats_pmc.csv
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p_b -- python3 bench.py --steps 5 --warmup 2 --min-seconds 0 --cpu-sample 0 --no-latency --no-host --no-real --no-kernel-timing --no-configs --no-gate > gpurun_out/p_b.log 2>&1
python tools/summarize_rocprof.py --stats gpurun_out/p_b --pairs-per-launch 64 -o gpurun_out/bench_pipelined_kernel_stats.csv
rm -rf gpurun_out/p_kt gpurun_out/p_f gpurun_out/p_w gpurun_out/p_sq gpurun_out/p_b
tail -c 400 gpurun_out/p_b.log
