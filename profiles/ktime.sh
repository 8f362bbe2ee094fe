#!/bin/bash
# per-kernel average duration of one bench configuration: bash profiles/ktime.sh <tag> [bench args]
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=$1; shift
rm -rf gpurun_out/kt_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_$tag -- python3 bench.py --no-cpu --no-secondary --steps 5 --warmup 2 "$@" > gpurun_out/kt_$tag.log 2>&1
python3 - <<PY
import csv,glob
for f in glob.glob("gpurun_out/kt_$tag/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "k_" in r["Name"]:
            print("$tag", r["Name"][:60].replace("(anonymous namespace)::",""), r["Calls"], "avg_us %.1f" % (float(r["AverageNs"])/1e3))
PY
