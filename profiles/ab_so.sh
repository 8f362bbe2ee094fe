#!/bin/bash
# Same-box A/B of two builds of the library: bash profiles/ab_so.sh <prev.so> <bench args...>
# (BC_NATIVE_SO selects the build; the previous build is kept as barcoder_amd/libbarcoder_hip_prev.so by hand)
cd $GRAFT_REPO_ROOT
prev=$1; shift
for round in 1 2; do
  for which in prev new; do
    if [ $which = prev ]; then export BC_NATIVE_SO=$GRAFT_REPO_ROOT/$prev; else unset BC_NATIVE_SO; fi
    python3 bench.py --no-cpu --no-secondary "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$which', 'value=%.0f median=%.0f kernel_ms=%.4f pass_ms=%.4f %s' % (d['value'], d['value_median'], r['kernel_ms'], r['pass_ms'], r['kernel']))"
  done
done
