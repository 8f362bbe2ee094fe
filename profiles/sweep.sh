cd $GRAFT_REPO_ROOT
out=gpurun_out/${TAG:-r03}_sweep.txt; : > $out
for G in ${GS:-1000 5000 20000 41000 45000 100000 125000 170000 300000 700000 1000000}; do
  for d in ${DS:-0 1 2 3}; do
    if [ $d = 0 ]; then a="--algo hash --max-mismatch 0 --mixture 0"; else a="--max-mismatch $d"; fi
    line=$(python3 bench.py --no-cpu --no-secondary --steps 5 --warmup 2 --guides $G $a 2>>$out.err)
    echo "G=$G d=$d $(echo "$line" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("value=%.0f kernel=%s kernel_ms=%.3f pass_ms=%.3f tally_ms=%.3f" % (d["value"], r["kernel"], r["kernel_ms"], r["pass_ms"], r["pass_ms"]-r["kernel_ms"]))')" >> $out
  done
done
cat $out
