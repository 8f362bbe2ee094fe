#!/bin/bash
# quick A/B of bench variants on the GPU box: kernel_ms / pass_ms / value per labelled environment.
# usage: bash profiles/ab.sh <outfile> "<label>|<ENV=VAL ...>|<bench args>" ...
out=$1; shift
: > $out
for spec in "$@"; do
  label=${spec%%|*}; rest=${spec#*|}; envs=${rest%%|*}; args=${rest#*|}
  line=$(env $envs python3 bench.py --no-cpu --no-secondary --steps 10 --warmup 2 $args 2>>$out.err)
  echo "$label $(echo "$line" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("value=%.0f median=%.0f kernel_ms=%.4f pass_ms=%.4f" % (d["value"], d.get("value_median",0), r["kernel_ms"], r["pass_ms"]))')" >> $out
done
cat $out
