# where the next longer seed starts to pay (d = 1: two 10-base segments, the seed is what build_seed_index picks)
cd $GRAFT_REPO_ROOT
for G in ${GS:-50000 60000 200000 250000 650000 800000}; do
  for b in 0 1; do
    line=$(BC_SEED_WANT_BIAS=$b python3 bench.py --no-cpu --no-secondary --steps 5 --warmup 2 --guides $G --max-mismatch 1 2>/dev/null)
    echo "G=$G d=1 bias=$b $(echo "$line" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("kernel=%s kernel_ms=%.3f pass_ms=%.3f" % (r["kernel"], r["kernel_ms"], r["pass_ms"]))' 2>&1 | tail -1)"
  done
done
