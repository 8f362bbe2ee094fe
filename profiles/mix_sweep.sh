#!/bin/bash
# Read-mix sweep of the d <= 2 matcher: kernel / pass time, VALU per tile and L2 requests per read on streams with
# 5 / 20 / 50 % reads that match nothing (bench.py --mixture 100+J: J % random reads, the others 0/1/2 substitutions 6:2:1)
# next to the standard stream (60/20/10/5 % with 0/1/2/3 substitutions + 5 % random: 10 % match nothing).
#   bash profiles/mix_sweep.sh [extra bench args]     -> stdout
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for mix in ${MIXES:-1 105 120 150}; do
  rm -rf gpurun_out/mx; mkdir -p gpurun_out/mx
  for pass in "SQ_INSTS_VALU SQ_INSTS_SALU" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
    p=$(echo $pass | cut -c1-12 | tr ' ' '_')
    timeout -k 10 200 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/mx/pmc_$p -- python3 bench.py --no-cpu --no-secondary --steps 2 --warmup 1 --mixture $mix "$@" > gpurun_out/mx/$p.log 2>&1 || echo "$p failed"
  done
  line=$(python3 bench.py --no-cpu --no-secondary --steps 10 --warmup 2 --mixture $mix "$@" 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("kernel_ms=%.4f pass_ms=%.4f value=%.0f matched=%.4f" % (r["kernel_ms"], r["pass_ms"], d["value"], d["config"].get("matched_fraction", -1)))')
  python3 profiles/summarize.py gpurun_out/mx k_match > gpurun_out/mx/summary.json
  python3 - <<PY
import json
d=json.load(open("gpurun_out/mx/summary.json"))
for name,e in d["pmc"].items():
    c={k: v["mean_per_dispatch"] for k,v in e.items() if k!="meta"}
    print("mixture=$mix $line", name[:40], "VALU/tile %.0f L2req/read %.3f L1acc/read %.3f" % (c.get("SQ_INSTS_VALU",0)/390625, c.get("TCP_TCC_READ_REQ_sum",0)/1e8, c.get("TCP_TOTAL_CACHE_ACCESSES_sum",0)/1e8))
PY
done
