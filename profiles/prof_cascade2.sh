#!/bin/bash
# counters of the d = 2 cascade's kernels on a large library: usage  bash profiles/prof_cascade2.sh [guides]
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
G=${1:-500000}
rm -rf gpurun_out/pc2; mkdir -p gpurun_out/pc2
A="--no-cpu --no-secondary --guides $G --max-mismatch 2"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pc2/trace -- python3 bench.py $A --steps 5 --warmup 2 > gpurun_out/pc2/trace.log 2>&1 || echo "trace failed"
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU" "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "TCP_PENDING_STALL_CYCLES_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "FETCH_SIZE"; do
  p=$(echo $pass | cut -c1-14 | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/pc2/pmc_$p -- python3 bench.py $A --steps 2 --warmup 1 > gpurun_out/pc2/$p.log 2>&1 || echo "$p failed"
done
python3 profiles/summarize.py gpurun_out/pc2 k_ > gpurun_out/pc2/summary.json
python3 - <<PY
import json, csv, glob
d=json.load(open("gpurun_out/pc2/summary.json"))
for f in glob.glob("gpurun_out/pc2/trace/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "k_" in r["Name"]: print(r["Name"][:60].replace("(anonymous namespace)::",""), "calls", r["Calls"], "avg_us %.1f"%(float(r["AverageNs"])/1e3))
for name,e in d["pmc"].items():
    c={k: v["mean_per_dispatch"] for k,v in e.items() if k!="meta"}
    cyc=c.get("GRBM_GUI_ACTIVE",0)/8
    print(name[:40], e["meta"]["VGPR_Count"], "vgpr; cycles/XCD %.3gM"%(cyc/1e6), "VALU %.4gM (%.0f%% issue)"%(c.get("SQ_INSTS_VALU",0)/1e6, 100*c.get("SQ_INSTS_VALU",0)/1024*4/max(cyc,1)),
          "VMEM_RD %.4gM"%(c.get("SQ_INSTS_VMEM_RD",0)/1e6), "TCC req %.4gM hit %.4gM miss %.4gM"%(c.get("TCC_REQ_sum",0)/1e6, c.get("TCC_HIT_sum",0)/1e6, c.get("TCC_MISS_sum",0)/1e6),
          "TCP acc %.4gM"%(c.get("TCP_TOTAL_CACHE_ACCESSES_sum",0)/1e6), "TCP->TCC rd %.4gM lat %.0f"%(c.get("TCP_TCC_READ_REQ_sum",0)/1e6, c.get("TCP_TCC_READ_REQ_LATENCY_sum",0)/max(c.get("TCP_TCC_READ_REQ_sum",1),1)),
          "wave cyc %.4gM wait_any %.4gM"%(c.get("SQ_WAVE_CYCLES",0)/1e6, c.get("SQ_WAIT_INST_ANY",0)/1e6),
          "tcp pend %.4gM"%(c.get("TCP_PENDING_STALL_CYCLES_sum",0)/1e6), "utcl1 miss %.4gM hit %.4gM"%(c.get("TCP_UTCL1_TRANSLATION_MISS_sum",0)/1e6, c.get("TCP_UTCL1_TRANSLATION_HIT_sum",0)/1e6),
          "fetch %.4g MB"%(c.get("FETCH_SIZE",0)*64/1e6))
PY
