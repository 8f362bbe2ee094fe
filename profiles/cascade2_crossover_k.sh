cd $GRAFT_REPO_ROOT
for spec in "16 40000" "16 60000" "16 75000" "18 100000" "18 150000" "24 1000000" "25 1000000"; do
  set -- $spec
  for env in "BC_CASCADE2_MIN_G=1" "BC_CASCADE2_MIN_G=0"; do
    line=$(env $env python3 bench.py --no-cpu --no-secondary --steps 5 --warmup 2 --k $1 --guides $2 --max-mismatch 2 2>/dev/null)
    echo "k=$1 G=$2 $env $(echo "$line" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("kernel=%s kernel_ms=%.3f pass_ms=%.3f" % (r["kernel"], r["kernel_ms"], r["pass_ms"]))')"
  done
done
