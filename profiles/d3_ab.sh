#!/bin/bash
# d = 3 cascade: parity tests of the pair index, then per-kernel times against the table-group budget of the second stage
# (BC_PAIR_GROUP_KB; 0 = the one-kernel form).  gpurun -- 'bash profiles/d3_ab.sh'
cd $GRAFT_REPO_ROOT
timeout -k 10 700 python -m pytest tests/test_match_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu > gpurun_out/d3_tests.log 2>&1 || { tail -20 gpurun_out/d3_tests.log; exit 1; }
tail -2 gpurun_out/d3_tests.log
for kb in ${KBS:-0 2200 3300 5500}; do
  BC_PAIR_GROUP_KB=$kb bash profiles/ktime.sh d3_$kb --max-mismatch 3 > gpurun_out/d3_kt_$kb.txt 2>&1
  echo "== group_kb $kb"; grep "cascade\|k_match_wq" gpurun_out/d3_kt_$kb.txt
  grep -o '"value": [0-9.e+]*' gpurun_out/kt_d3_$kb.log | head -1
done
bash profiles/ktime.sh d2 > gpurun_out/d2_kt.txt 2>&1
echo "== d <= 2 default"; grep "k_match_wq\|k_hist_part" gpurun_out/d2_kt.txt; grep -o '"value": [0-9.e+]*' gpurun_out/kt_d2.log | head -1
