#!/bin/bash
# The default bench line and three driver-like invocations (other --steps / --warmup) on one box:
#   gpurun -- 'bash profiles/bench_invocations.sh'  ->  gpurun_out/bench_final.json, gpurun_out/bench_invocations.txt
cd $GRAFT_REPO_ROOT
python3 bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
for kw in "20 5" "5 1" "100 20"; do
  set -- $kw
  python3 bench.py --gpus 1 --steps $1 --warmup $2 --no-cpu --no-secondary 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('steps', d['steps'], 'warmup', d['warmup'], 'value=%.0f median=%.0f kernel_ms=%.4f frac=%.4f steady=%.4f pass_ms=%.4f' % (d['value'], d['value_median'], r['kernel_ms'], r['frac'], r['kernel_ms_steady'], r['pass_ms']))"
done > gpurun_out/bench_invocations.txt
python3 -c "
import json
d = json.load(open('gpurun_out/bench_final.json')); r = d['roofline']
print('default', d['steps'], d['warmup'], round(d['value']), round(d['value_median']), round(d['ms_per_step'], 4), round(r['kernel_ms'], 4), round(r['frac'], 4), round(r['kernel_ms_steady'], 4), round(r['pass_ms'], 4), r['traffic'], d['secondary']['from_file']['Mreads_per_s'], d['secondary']['exact_1e8']['kernel_ms'])" >> gpurun_out/bench_invocations.txt
cat gpurun_out/bench_invocations.txt
