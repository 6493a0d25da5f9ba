"""Summarises a rocprofv3 kernel trace (…_kernel_trace.csv) per (kernel, launch geometry): calls, avg/min/max ns.
usage: python tools/trace_by_grid.py <kernel_trace.csv> > profiles/rNN_…_by_grid.csv"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
acc = defaultdict(list)
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"^void ", "", name)
    name = name.split("(")[0]
    key = (name, int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Workgroup_Size_X"]))
    acc[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
w = csv.writer(sys.stdout)
w.writerow(["kernel", "grid_x_threads", "grid_y", "wg_x", "calls", "avg_ns", "min_ns", "max_ns", "total_ns"])
for key, d in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    w.writerow([key[0], key[1], key[2], key[3], len(d), round(sum(d) / len(d), 1), min(d), max(d), sum(d)])
