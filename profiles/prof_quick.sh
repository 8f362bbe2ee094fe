#!/bin/bash
# quick PMC for fused kernel variants
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in "BC_SLAB4=1" "BC_SLAB4=0" "BC_SLAB4=1 BC_HASH_SLOTS_PCT=250"; do
  tag=$(echo $v | tr ' =' '__')
  for pass in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU"; do
    p=$(echo $pass | cut -c1-12)
    env $v timeout -k 10 200 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/pq_${tag}/pmc_$p -- python3 bench.py --no-cpu --steps 2 --warmup 1 > /dev/null 2>&1
  done
  python3 profiles/summarize.py gpurun_out/pq_${tag} k_match_fused > gpurun_out/pq_${tag}/summary.json
  echo "== $v"
  python3 - <<PY
import json
d=json.load(open("gpurun_out/pq_${tag}/summary.json"))
for name,e in d["pmc"].items():
    print(name, {c: float("%.4g"%v["mean_per_dispatch"]) for c,v in e.items() if c!="meta"})
PY
done
