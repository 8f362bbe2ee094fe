#!/bin/bash
# read-stream prefetch of k_match_wq (BC_PREFETCH, units of 128 tiles): L2 misses, fetched bytes and L1->L2 latency
# against the distance.  usage: bash profiles/pf_prof.sh "0 16 24 ..."
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pf
for pf in ${1:-0 16}; do
  for pass in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
    p=$(echo $pass | cut -c1-10 | tr ' ' '_')
    rm -rf gpurun_out/pf/pmc_${pf}_$p
    BC_PREFETCH=$pf timeout -k 10 200 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/pf/pmc_${pf}_$p -- python3 bench.py --no-cpu --no-secondary --steps 2 --warmup 1 > gpurun_out/pf/log_${pf}_$p.txt 2>&1 || echo "$p failed"
  done
  ms=$(BC_PREFETCH=$pf python3 bench.py --no-cpu --no-secondary --steps 10 --warmup 2 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4f %.0f" % (d["roofline"]["kernel_ms"], d["value"]))')
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("gpurun_out/pf/pmc_${pf}_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_match_wq" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
c={k: sum(v)/len(v) for k,v in acc.items()}
print("prefetch=$pf kernel_ms/value=$ms", "L2 misses %.2fe7 (of %.2fe8 requests)" % (c.get("TCC_MISS_sum",0)/1e7, c.get("TCC_REQ_sum",0)/1e8), "mean L1->L2 read latency %.0f cycles" % (c.get("TCP_TCC_READ_REQ_LATENCY_sum",0)/max(c.get("TCP_TCC_READ_REQ_sum",1),1)), "L1 pending-stall Mcycles %.0f" % (c.get("TCP_PENDING_STALL_CYCLES_sum",0)/1e6))
PY
done
