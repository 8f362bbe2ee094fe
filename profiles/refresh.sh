#!/bin/bash
# Regenerate the round's measurement artefacts on the GPU box (one gpurun call):
#   make -C barcoder_amd/csrc diag          # here, before the call: the stage-attribution build travels with the tree
#   gpurun --timeout 1200 -- 'bash profiles/refresh.sh r04'   -> gpurun_out/refresh_r04/ ; copy what is judged into profiles/
# Kernel trace and every PMC set run as separate rocprofv3 invocations (run_prof.sh); nothing here combines --pmc with
# system / runtime tracing.  STEPS="prof bench variants exact pam diag mix d3 e2e targets sweep" selects parts.
set -u
TAG=${1:-r04}
STEPS=${STEPS:-prof bench variants exact pam diag mix d3 e2e targets sweep}
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/refresh_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
has() { [[ " $STEPS " == *" $1 "* ]]; }
kstats() {  # per-kernel averages of a rocprofv3 --kernel-trace --stats directory -> one line per kernel
  python3 - "$1" "$2" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "k_" in r["Name"]:
            print(sys.argv[2], r["Name"][:70].replace("(anonymous namespace)::", ""), "calls", r["Calls"],
                  "avg_us %.1f min_us %.1f max_us %.1f" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
}
if has prof; then
  echo "rocprof" > $OUT/progress.log
  bash profiles/run_prof.sh $TAG --steps 50 --warmup 20 > $OUT/run_prof.log 2>&1   # the default bench command (50 timed steps after 20)
  python3 profiles/summarize.py gpurun_out/prof_$TAG k_ > $OUT/${TAG}_rocprof_summary.json
  cp gpurun_out/prof_$TAG/trace/*/*kernel_stats.csv $OUT/${TAG}_kernel_stats.csv 2>/dev/null
  python3 profiles/collect_traffic.py $OUT/${TAG}_rocprof_summary.json 1e8 seed 2 > $OUT/collect_traffic.log 2>&1
  cp profiles/traffic.json $OUT/traffic.json 2>/dev/null
fi
if has bench; then
  echo "default bench" >> $OUT/progress.log
  python3 bench.py > $OUT/${TAG}_bench_n1.json 2> $OUT/bench_n1.err
fi
if has variants; then
  for spec in "exact|--algo hash --max-mismatch 0 --mixture 0" "d1|--max-mismatch 1" "d3|--max-mismatch 3" "k32|--k 32" \
              "G2e4|--guides 20000" "G2e5|--guides 200000" "G5e5|--guides 500000"; do
    label=${spec%%|*}; args=${spec#*|}
    echo "bench $label" >> $OUT/progress.log
    python3 bench.py --no-cpu --no-secondary $args > $OUT/${TAG}_bench_${label}_n1.json 2>> $OUT/bench_variants.err
  done
  echo "gloo2" >> $OUT/progress.log
  BC_DIST_BACKEND=gloo python3 bench.py --gpus 2 --no-cpu --no-secondary --reads 20000000 > $OUT/${TAG}_bench_gloo2_onegpu.json 2>> $OUT/bench_variants.err
fi
if has exact; then  # configs[1]: rocprofv3 rows of the exact-match run
  echo "exact kernels" >> $OUT/progress.log
  rm -rf gpurun_out/kt_exact
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_exact -- python3 bench.py --no-cpu --no-secondary --steps 5 --warmup 2 --algo hash --max-mismatch 0 --mixture 0 > $OUT/kt_exact.log 2>&1
  kstats gpurun_out/kt_exact exact > $OUT/${TAG}_exact_kernels.txt
fi
if has pam; then    # configs[3]: the scan at 4.6 Mbp and 3.1e9 bp, the fused site pipeline; rocprofv3 rows of the same script
  echo "pam" >> $OUT/progress.log
  python3 profiles/pam_bench.py > $OUT/${TAG}_pam_bench.json 2> $OUT/pam_bench.err
  rm -rf gpurun_out/kt_pam
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_pam -- python3 profiles/pam_bench.py > $OUT/kt_pam.log 2>&1
  kstats gpurun_out/kt_pam pam > $OUT/${TAG}_pam_kernels.txt
  python3 profiles/pam_big_row.py gpurun_out/kt_pam >> $OUT/${TAG}_pam_kernels.txt   # the 3.1e9-bp launches as rows of their own (by grid size)
fi
if has diag; then
  echo "stage attribution" >> $OUT/progress.log
  DIAGS="0 1 3" bash profiles/diag.sh > $OUT/${TAG}_stage_attribution.txt 2>&1
fi
if has mix; then   # read-mix sweep of the d <= 2 matcher: 5 / 10 (standard) / 20 / 50 % reads that match nothing
  echo "mix sweep" >> $OUT/progress.log
  bash profiles/mix_sweep.sh > $OUT/${TAG}_mix_sweep.txt 2>&1
fi
if has d3; then
  echo "d3 kernels" >> $OUT/progress.log
  bash profiles/ktime.sh d3 --max-mismatch 3 > $OUT/${TAG}_d3_kernels.txt 2>&1
fi
if has e2e; then
  echo "e2e" >> $OUT/progress.log
  BC_LIB_PROF=1 BC_SAMPLE_PROF=1 python3 profiles/e2e_bench.py 1e8 > $OUT/${TAG}_e2e_bench.txt 2>&1
  python3 profiles/e2e_bench.py 5e7 100000 paired > $OUT/${TAG}_e2e_paired.txt 2>&1
  BC_LIB_PROF=1 BC_SAMPLE_PROF=1 BC_STREAM_PROF=1 python3 profiles/e2e_bench.py 1e7 > $OUT/${TAG}_e2e_1e7.txt 2>&1   # a small job: the fixed costs
  python3 profiles/e2e_bench.py 2e7 100000 gz > $OUT/${TAG}_e2e_gz.txt 2>&1     # compressed input: the decoder alone, then the job
  python3 profiles/e2e_bench.py 2e7 100000 zst > $OUT/${TAG}_e2e_zst.txt 2>&1
fi
if has targets; then
  echo "targets" >> $OUT/progress.log
  python3 profiles/targets_bench.py 20000 1 > $OUT/${TAG}_targets_bench.txt 2>&1
  python3 profiles/targets_bench.py 100000 1 >> $OUT/${TAG}_targets_bench.txt 2>&1
fi
if has sweep; then
  echo "sweep" >> $OUT/progress.log
  TAG=$TAG GS="${GS:-20000 100000 170000 300000 500000 700000 1000000}" bash profiles/sweep.sh > /dev/null 2>&1
  cp gpurun_out/${TAG}_sweep.txt $OUT/${TAG}_sweep_G_d.txt 2>/dev/null
fi
echo done >> $OUT/progress.log
tail -c 600 $OUT/${TAG}_bench_n1.json 2>/dev/null
