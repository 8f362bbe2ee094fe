#!/bin/bash
# instruction-issue accounting of the match kernels (VALU/SALU/LDS/VMEM wave-instructions per dispatch)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in ${VARIANTS:-"--max-mismatch 2"}; do
  tag=$(echo $v | tr ' -' '__')
  for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "FETCH_SIZE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
    p=$(echo $pass | cut -c1-14 | tr ' ' '_')
    echo "$tag $p" >> gpurun_out/pi_progress.log
    timeout -k 10 200 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/pi_${tag}/pmc_$p -- python3 bench.py --no-cpu --no-secondary --steps 2 --warmup 1 $v > gpurun_out/pi_${tag}_$p.log 2>&1 || echo "$tag $p failed" >> gpurun_out/pi_progress.log
  done
  python3 profiles/summarize.py gpurun_out/pi_${tag} k_ > gpurun_out/pi_${tag}/summary.json
  echo "== $v"
  python3 - <<PY
import json
d=json.load(open("gpurun_out/pi_${tag}/summary.json"))
for name,e in d["pmc"].items():
    if "match" in name or "hist" in name:
        print(name, {c: float("%.4g"%v["mean_per_dispatch"]) for c,v in e.items() if c!="meta"})
PY
done
