cd $GRAFT_REPO_ROOT
for g in 1024 2048 4096 8192 16384; do
  BC_PAIR_GRID=$g bash profiles/ktime.sh d3g_$g --max-mismatch 3 > gpurun_out/d3g_$g.txt 2>&1
  echo "== grid $g"; grep "cascade_pass" gpurun_out/d3g_$g.txt; grep -o '"value": [0-9.e+]*' gpurun_out/kt_d3g_$g.log | head -1
done
