#!/bin/bash
# Per-kernel totals of the pipelined 4K run, everything resident on the GPU (two host threads) against the pool doing the triangulations:
#   bash tools/compare_4k.sh > gpurun_out/compare_4k.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="--workload 4k_d192 --steps 6 --warmup 1 --min-seconds 0 --no-configs --host-share 0 --no-kernel-timing --no-host --no-latency --no-real --cpu-sample 0 --no-gate"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c4k_gpu -- python3 bench.py $B --workers 2 > gpurun_out/c4k_gpu.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c4k_host -- python3 bench.py $B > gpurun_out/c4k_host.log 2>&1
python3 - <<PY
import csv, glob, json
def load(d):
    f = glob.glob("gpurun_out/%s/*/*kernel_stats.csv" % d)[0]
    return {r["Name"].split("(")[0][-48:]: (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6) for r in csv.DictReader(open(f))}
g, h = load("c4k_gpu"), load("c4k_host")
for n in ("c4k_gpu", "c4k_host"):
    print(n, [json.loads(ln)["value"] for ln in open("gpurun_out/%s.log" % n) if ln.startswith("{")][-1])
print("%-50s %8s %10s %8s %10s" % ("kernel", "calls", "gpu ms", "calls", "host ms"))
for k in sorted(set(g) | set(h), key=lambda k: -(g.get(k, (0, 0))[1])):
    a, b = g.get(k, (0, 0.0)), h.get(k, (0, 0.0))
    if a[1] + b[1] > 1.0:
        print("%-50s %8d %10.1f %8d %10.1f" % (k, a[0], a[1], b[0], b[1]))
print("sum", sum(v[1] for v in g.values()), sum(v[1] for v in h.values()))
PY
rm -rf gpurun_out/c4k_gpu gpurun_out/c4k_host
