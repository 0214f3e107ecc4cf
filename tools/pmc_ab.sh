#!/bin/bash
# VALU wave-instructions per kernel of two library builds (no-overlap pass, 32 pairs per launch):  bash tools/pmc_ab.sh abl_tmp/lib_a.so abl_tmp/lib_b.so
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in "$@"; do
    tag=$(basename $lib .so)
    SV_LIB_PATH=$lib rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d gpurun_out/pmc_$tag -- python3 tools/profile_run.py --chunk 32 --batch 64 --reps 3 > gpurun_out/pmc_$tag.log 2>&1
done
python3 - "$@" <<'PY'
import csv, glob, os, sys, collections
tabs = []
for lib in sys.argv[1:]:
    tag = os.path.basename(lib)[:-3]
    f = glob.glob("gpurun_out/pmc_%s/*/*counter_collection.csv" % tag)[0]
    acc, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "SQ_INSTS_VALU":
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sv::", "")
            k = k[:k.index("<false")] if "<false" in k else k[:28]  # (instantiations of the product kernels under one name)
            acc[k] += float(r["Counter_Value"]); cnt[k] += 1
    tabs.append({k: acc[k] / cnt[k] / 32 for k in acc})
keys = sorted(tabs[0], key=lambda k: -tabs[0][k])
print("%-30s" % "wave-instructions per pair", *["%12s" % os.path.basename(l)[:-3] for l in sys.argv[1:]])
for k in keys:
    print("%-30s" % k, *["%12.0f" % t.get(k, 0) for t in tabs])
print("%-30s" % "sum", *["%12.0f" % sum(t.values()) for t in tabs])
PY
rm -rf gpurun_out/pmc_*
