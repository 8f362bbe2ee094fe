#!/bin/bash
# Profiling recipe (run on the GPU box via gpurun): kernel trace + separate PMC passes.
# usage: bash profiles/run_prof.sh <tag> [bench args...]
set -e
TAG=$1; shift
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu "$@" > $OUT/bench_trace.log 2>&1
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum" "GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 bench.py --no-cpu --steps 2 --warmup 1 "${@}" > $OUT/pmc_$name.log 2>&1 || echo "pass $pass failed" >> $OUT/errors.log
done
find $OUT -name "*.csv" | head -50 > $OUT/files.txt
