# which kernel the default rule picks, beside both forced choices (pass ms per 1e8 reads).  gpurun -- 'bash profiles/cascade2_rule_check.sh'
cd $GRAFT_REPO_ROOT
for spec in "20 300000" "20 350000" "20 370000" "20 400000" "16 60000" "16 75000" "18 100000" "18 150000" "17 150000" "19 150000" "22 400000" "23 600000"; do
  set -- $spec
  out="k=$1 G=$2"
  for env in "BC_NOP=1" "BC_CASCADE2_MIN_G=1" "BC_CASCADE2_MIN_G=0"; do
    line=$(env $env python3 bench.py --no-cpu --no-secondary --steps 5 --warmup 2 --k $1 --guides $2 --max-mismatch 2 2>/dev/null)
    out="$out | ${env#BC_} $(echo "$line" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("%s %.3f" % (r["kernel"][:8], r["pass_ms"]))')"
  done
  echo "$out"
done
