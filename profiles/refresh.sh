#!/bin/bash
# Regenerate the round's measurement artefacts on the GPU box (one gpurun call):
#   bash profiles/refresh.sh r02       -> gpurun_out/refresh_r02/ ; copy what is judged into profiles/ afterwards
# Kernel trace and every PMC set run as separate rocprofv3 invocations (run_prof.sh); nothing here combines --pmc with
# system / runtime tracing.
set -u
TAG=${1:-r02}
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/refresh_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "rocprof" > $OUT/progress.log
bash profiles/run_prof.sh $TAG --steps 10 --warmup 2 > $OUT/run_prof.log 2>&1
python3 profiles/summarize.py gpurun_out/prof_$TAG k_ > $OUT/${TAG}_rocprof_summary.json
cp gpurun_out/prof_$TAG/trace/*/*kernel_stats.csv $OUT/${TAG}_kernel_stats.csv 2>/dev/null
python3 profiles/collect_traffic.py $OUT/${TAG}_rocprof_summary.json 1e8 seed 2 > $OUT/collect_traffic.log 2>&1
cp profiles/traffic.json $OUT/traffic.json 2>/dev/null
echo "default bench" >> $OUT/progress.log
python3 bench.py > $OUT/${TAG}_bench_n1.json 2> $OUT/bench_n1.err
for spec in "exact|--algo hash --max-mismatch 0 --mixture 0" "d1|--max-mismatch 1" "d3|--max-mismatch 3" "k32|--k 32" \
            "G2e4|--guides 20000" "G6e4|--guides 60000" "G8e4|--guides 80000" "G2e5|--guides 200000" "G5e5|--guides 500000"; do
  label=${spec%%|*}; args=${spec#*|}
  echo "bench $label" >> $OUT/progress.log
  python3 bench.py --no-cpu --no-secondary $args > $OUT/${TAG}_bench_${label}_n1.json 2>> $OUT/bench_variants.err
done
echo "gloo2" >> $OUT/progress.log
BC_DIST_BACKEND=gloo python3 bench.py --gpus 2 --no-cpu --no-secondary --reads 20000000 > $OUT/${TAG}_bench_gloo2_onegpu.json 2>> $OUT/bench_variants.err
echo "stage attribution" >> $OUT/progress.log
DIAGS="0 1 3" bash profiles/diag.sh > $OUT/${TAG}_stage_attribution.txt 2>&1
echo "d3 kernels" >> $OUT/progress.log
bash profiles/ktime.sh d3 --max-mismatch 3 > $OUT/${TAG}_d3_kernels.txt 2>&1
echo "e2e" >> $OUT/progress.log
python3 profiles/e2e_bench.py 1e8 > $OUT/${TAG}_e2e_bench.txt 2>&1
echo "targets" >> $OUT/progress.log
python3 profiles/targets_bench.py 20000 1 > $OUT/${TAG}_targets_bench.txt 2>&1
python3 profiles/targets_bench.py 100000 1 >> $OUT/${TAG}_targets_bench.txt 2>&1
echo done >> $OUT/progress.log
tail -c 400 $OUT/${TAG}_bench_n1.json
