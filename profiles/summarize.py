#!/usr/bin/env python3
"""Summarise rocprofv3 csv output (kernel stats + per-kernel PMC means) as JSON.
usage: python profiles/summarize.py gpurun_out/prof_<tag> [kernel-substring]"""
import csv, glob, json, os, re, sys
from collections import defaultdict

root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "k_match"
out = {"kernel_stats": [], "pmc": {}}
for f in glob.glob(os.path.join(root, "trace", "*", "*kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        out["kernel_stats"].append({k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs") if k in r})
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
        acc = defaultdict(lambda: defaultdict(list))
        meta = {}
        for r in csv.DictReader(open(f)):
            if want not in r["Kernel_Name"]:
                continue
            m = re.search(r"(k_\w+(<[^>]*>)?)", r["Kernel_Name"])
            name = m.group(1) if m else r["Kernel_Name"][:60]
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[name] = {k: r.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size", "Workgroup_Size")}
        for name, cs in acc.items():
            e = out["pmc"].setdefault(name, {"meta": meta[name]})
            for c, v in cs.items():
                e[c] = {"mean_per_dispatch": sum(v) / len(v), "max_per_dispatch": max(v), "dispatches": len(v)}
print(json.dumps(out, indent=1))
