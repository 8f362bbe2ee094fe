# seed length of the segment tables (BC_SEED_WANT_BIAS: -1 / 0 / +1 base) on large libraries, d = 1 and 2.  gpurun -- 'bash profiles/seed_want_ab.sh'
cd $GRAFT_REPO_ROOT
for G in ${GS:-300000 500000 1000000}; do
  for D in 1 2; do
    for b in -1 0 1; do
      line=$(BC_SEED_WANT_BIAS=$b python3 bench.py --no-cpu --no-secondary --steps 5 --warmup 2 --guides $G --max-mismatch $D 2>/dev/null)
      echo "G=$G d=$D bias=$b $(echo "$line" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("kernel=%s kernel_ms=%.3f pass_ms=%.3f" % (r["kernel"], r["kernel_ms"], r["pass_ms"]))' 2>&1 | tail -1)"
    done
  done
done
