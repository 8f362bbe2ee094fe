#!/bin/bash
# Profiling recipe (run on the GPU box via gpurun): kernel trace + separate PMC passes
# (never combined with sys/runtime tracing).  usage: bash profiles/run_prof.sh <tag> [bench args...]
set -e
TAG=$1; shift
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu --no-secondary "$@" > $OUT/bench_trace.log 2>&1  # (steady state: give --warmup at least 10)
i=0
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "GRBM_GUI_ACTIVE" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM" \
  "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
  "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum"; do
  i=$((i+1))
  echo "pass $i: $pass" >> $OUT/progress.log
  timeout -k 10 240 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc_$i -- python3 bench.py --no-cpu --no-secondary --steps 2 --warmup 1 "${@}" > $OUT/pmc_$i.log 2>&1 || echo "pass $i ($pass) failed" >> $OUT/errors.log
done
