# d <= 2 on large libraries: the wave matcher against the cascade (d <= 1 pass + pair index), the latter with the four
# tables that cover three clean pieces (default) and with all ten (BC_PAIR_FOUR=0).  gpurun -- 'bash profiles/cascade2_ab.sh'
cd $GRAFT_REPO_ROOT
out=gpurun_out/${TAG:-r04}_cascade2_ab.txt; : > $out
one() {  # label, env..., G
  local label=$1 G=$2; shift 2
  line=$(env "$@" python3 bench.py --no-cpu --no-secondary --steps 5 --warmup 2 --guides $G --max-mismatch ${D:-2} 2>>$out.err)
  echo "G=$G d=${D:-2} $label $(echo "$line" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("value=%.0f kernel=%s kernel_ms=%.3f pass_ms=%.3f" % (d["value"], r["kernel"], r["kernel_ms"], r["pass_ms"]))')" >> $out
}
for G in ${GS:-200000 300000 400000 500000 650000 1000000}; do
  one default $G BC_NOP=1
  one wave $G BC_CASCADE2_MIN_G=0
  one cascade4 $G BC_CASCADE2_MIN_G=1
  one cascade10 $G BC_CASCADE2_MIN_G=1 BC_PAIR_FOUR=0
done
cat $out
