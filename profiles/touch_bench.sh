export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tb; mkdir -p gpurun_out/tb
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d gpurun_out/tb -- ./profiles/_bin/touch_bench > gpurun_out/tb/log.txt 2>&1
python3 - <<'PY'
import csv,glob,collections
acc=collections.OrderedDict()
for f in glob.glob("gpurun_out/tb/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        acc.setdefault(r["Kernel_Name"],{})[r["Counter_Name"]]=float(r["Counter_Value"])
for k,v in acc.items(): print(k[:40], {c:round(x) for c,x in v.items()})
tr={}
for f in glob.glob("gpurun_out/tb/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)): tr[r["Kernel_Name"]]=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
print({k[:24]:v for k,v in tr.items()})
PY
