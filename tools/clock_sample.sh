#!/bin/bash
# Shader / memory clocks and power while the headline batch streams: rocm-smi sampled once a second beside tools/thread_cpu.py.
#   bash tools/clock_sample.sh [seconds] > gpurun_out/clocks.txt
S=${1:-8}
python3 tools/thread_cpu.py --seconds $S > /tmp/clock_sample_run.txt 2>&1 &
PID=$!
sleep 5   # import + warm-up
for i in $(seq 1 $((S - 2))); do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|power" | tr '\n' ' '
    echo
    sleep 1
done
wait $PID
cat /tmp/clock_sample_run.txt
echo "idle:"
sleep 2
rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|power" | tr '\n' ' '
echo
