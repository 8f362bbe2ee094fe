cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for mode in 2 0; do
  rm -rf gpurun_out/kt_tally$mode
  BC_TALLY=$mode rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_tally$mode -- python3 bench.py --no-cpu --no-secondary --steps 40 --warmup 30 > gpurun_out/kt_tally$mode.log 2>&1
  python3 - gpurun_out/kt_tally$mode $mode <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+"/*/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f))); rows.sort(key=lambda r:int(r["Start_Timestamp"]))
seq=[]; cur=[]
names={}
for r in rows:
    n=r["Kernel_Name"]; d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    key="wq" if "k_match_wq" in n else "hist_part" if "k_hist_part" in n else "hist" if "k_hist(" in n or "k_hist<" in n else "refine" if "k_first_refine" in n else "export" if "k_tally_export" in n else None
    if key: names.setdefault(key,[]).append(d)
    if key=="wq": cur.append(round(d,1))
    elif key is None:
        if cur: seq.append(cur); cur=[]
if cur: seq.append(cur)
print("BC_TALLY=%s"%sys.argv[2], {k:(len(v), round(sum(v)/len(v),1)) for k,v in names.items()})
for i,c in enumerate(seq): print(" burst",i+1,len(c),"last 12 wq:",c[-12:])
PY
done
