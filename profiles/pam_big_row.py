"""The k_pam_scan launches over the 3.1e9-bp genome as a row of their own: rocprofv3's --stats pools them with the
thousands of 4.6-Mbp launches of the same kernel (r04_pam_kernels.txt: 4047 calls, 7.5 us average).  Reads the
per-dispatch kernel trace of the same run (usage: python3 profiles/pam_big_row.py <rocprofv3 output directory>) and
prints the launches by grid size."""
import csv
import glob
import sys
from collections import defaultdict

rows = defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "k_pam_scan" in r["Kernel_Name"]:
            grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)))
            rows[grid].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for grid in sorted(rows):
    d = sorted(rows[grid])
    print("pam k_pam_scan grid_x %d launches %d avg_us %.1f min_us %.1f median_us %.1f max_us %.1f" %
          (grid, len(d), sum(d) / len(d), d[0], d[len(d) // 2], d[-1]))
