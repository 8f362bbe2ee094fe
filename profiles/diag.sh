#!/bin/bash
# stage attribution of k_match_wq: VALU wave-instructions and kernel time with stages switched off (BC_DIAG bits:
# 1 = no table-2 phase, 2 = no table-1 phase / item drain, 4 = no tie resolution).  Results are wrong with any bit set.
# Needs the diag build (`make -C barcoder_amd/csrc diag` before the gpurun call): the release library has no BC_DIAG.
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export BC_NATIVE_SO=$GRAFT_REPO_ROOT/barcoder_amd/libbarcoder_hip_diag.so
[ -f "$BC_NATIVE_SO" ] || { echo "diag.sh: $BC_NATIVE_SO missing (make -C barcoder_amd/csrc diag)"; exit 1; }
for d in ${DIAGS:-0 1 2 3 4 7}; do
  rm -rf gpurun_out/dg; mkdir -p gpurun_out/dg
  for pass in "SQ_INSTS_VALU SQ_INSTS_SALU" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
    p=$(echo $pass | cut -c1-12 | tr ' ' '_')
    BC_DIAG=$d timeout -k 10 200 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/dg/pmc_$p -- python3 bench.py --no-cpu --no-secondary --steps 2 --warmup 1 "$@" > gpurun_out/dg/$p.log 2>&1 || echo "$p failed"
  done
  ms=$(BC_DIAG=$d python3 bench.py --no-cpu --no-secondary --steps 10 --warmup 2 "$@" 2>/dev/null | python3 -c 'import sys,json; print("%.4f" % json.loads(sys.stdin.read())["roofline"]["kernel_ms"])')
  python3 profiles/summarize.py gpurun_out/dg k_match > gpurun_out/dg/summary.json
  python3 - <<PY
import json
d=json.load(open("gpurun_out/dg/summary.json"))
for name,e in d["pmc"].items():
    c={k: v["mean_per_dispatch"] for k,v in e.items() if k!="meta"}
    print("diag=$d kernel_ms=$ms", name, "VALU/tile %.0f SALU/tile %.0f L2req/read %.3f L1acc/read %.3f" % (c.get("SQ_INSTS_VALU",0)/390625, c.get("SQ_INSTS_SALU",0)/390625, c.get("TCP_TCC_READ_REQ_sum",0)/1e8, c.get("TCP_TOTAL_CACHE_ACCESSES_sum",0)/1e8))
PY
done
