# the pair tables' key length (guides per bucket) against the cascades' time on large libraries.  gpurun -- 'bash profiles/pair_bucket_ab.sh'
cd $GRAFT_REPO_ROOT
out=gpurun_out/${TAG:-r04}_pair_bucket_ab.txt; : > $out
for G in ${GS:-200000 500000 1000000}; do
  for D in ${DS:-2 3}; do
    for pct in ${PCTS:-800 1600 3200 6400}; do
      line=$(BC_CASCADE2_MIN_G=1 BC_PAIR_BUCKET_PCT=$pct python3 bench.py --no-cpu --no-secondary --steps 5 --warmup 2 --guides $G --max-mismatch $D 2>>$out.err)
      echo "G=$G d=$D bucket_pct=$pct $(echo "$line" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("value=%.0f kernel=%s kernel_ms=%.3f pass_ms=%.3f" % (d["value"], r["kernel"], r["kernel_ms"], r["pass_ms"]))')" >> $out
    done
  done
done
cat $out
