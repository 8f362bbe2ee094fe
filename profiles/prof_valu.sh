#!/bin/bash
# quick VALU / L2 accounting of the match kernel: usage  [KPREFIX=k_cascade] bash profiles/prof_valu.sh [bench args]
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pv; mkdir -p gpurun_out/pv
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU" "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "TCP_PENDING_STALL_CYCLES_sum SQ_INST_CYCLES_SALU"; do
  p=$(echo $pass | cut -c1-14 | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/pv/pmc_$p -- python3 bench.py --no-cpu --no-secondary --steps 2 --warmup 1 "$@" > gpurun_out/pv/$p.log 2>&1 || echo "$p failed"
done
python3 profiles/summarize.py gpurun_out/pv ${KPREFIX:-k_match} > gpurun_out/pv/summary.json
python3 - <<PY
import json
d=json.load(open("gpurun_out/pv/summary.json"))
for name,e in d["pmc"].items():
    c={k: v["mean_per_dispatch"] for k,v in e.items() if k!="meta"}
    cyc=c.get("GRBM_GUI_ACTIVE",0)/8
    print(name, e["meta"]["VGPR_Count"], "vgpr;", "cycles/XCD %.3gM"%(cyc/1e6), "VALU %.4gM (%.0f%% issue)"%(c.get("SQ_INSTS_VALU",0)/1e6, 100*c.get("SQ_INSTS_VALU",0)/1024*4/max(cyc,1)),
          "lanes/instr %.1f"%(c.get("SQ_THREAD_CYCLES_VALU",0)/max(c.get("SQ_INSTS_VALU",1),1)), "SALU %.4gM"%(c.get("SQ_INSTS_SALU",0)/1e6), "VMEM_RD %.4gM"%(c.get("SQ_INSTS_VMEM_RD",0)/1e6),
          "TCC req %.4gM miss %.4gM"%(c.get("TCC_REQ_sum",0)/1e6, c.get("TCC_MISS_sum",0)/1e6), "TCP acc %.4gM"%(c.get("TCP_TOTAL_CACHE_ACCESSES_sum",0)/1e6),
          "TCP->TCC rd %.4gM"%(c.get("TCP_TCC_READ_REQ_sum",0)/1e6), "wave cycles %.4gM wait_any %.4gM active_valu %.4gM"%(c.get("SQ_WAVE_CYCLES",0)/1e6, c.get("SQ_WAIT_INST_ANY",0)/1e6, c.get("SQ_ACTIVE_INST_VALU",0)/1e6),
          "tcp pending stall %.4gM"%(c.get("TCP_PENDING_STALL_CYCLES_sum",0)/1e6))
PY
