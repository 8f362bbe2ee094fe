# read-mix sweep on a large library (5e5 guides): the d = 2 cascade and the wave matcher, pass ms per 1e8 reads
cd $GRAFT_REPO_ROOT
for G in ${GS:-500000}; do
 for mix in 1 105 120 150; do
  out="G=$G mixture=$mix"
  for env in "BC_NOP=1" "BC_CASCADE2_MIN_G=0"; do
    line=$(env $env python3 bench.py --no-cpu --no-secondary --steps 5 --warmup 2 --guides $G --mixture $mix 2>/dev/null)
    out="$out | ${env#BC_} $(echo "$line" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("%s kernel %.3f pass %.3f" % (r["kernel"][:8], r["kernel_ms"], r["pass_ms"]))')"
  done
  echo "$out"
 done
done
