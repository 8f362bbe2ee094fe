#!/bin/bash
# Same-box A/B of several builds of the library: bash profiles/ab3.sh "<a.so> <b.so> ..." <bench args...>
# ("-" = the default build, barcoder_amd/libbarcoder_hip.so); two rounds, interleaved
cd $GRAFT_REPO_ROOT
sos=$1; shift
for round in 1 2; do
  for so in $sos; do
    if [ "$so" = "-" ]; then unset BC_NATIVE_SO; else export BC_NATIVE_SO=$GRAFT_REPO_ROOT/$so; fi
    python3 bench.py --no-cpu --no-secondary "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$so', 'value=%.0f median=%.0f kernel_ms=%.4f pass_ms=%.4f %s' % (d['value'], d['value_median'], r['kernel_ms'], r['pass_ms'], r['kernel']))"
  done
done
unset BC_NATIVE_SO
