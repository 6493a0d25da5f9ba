"""dev tool: prints the kernel sequence (name, duration, gap to the previous kernel's end) of a slice of a rocprofv3 kernel trace.
usage: trace_timeline.py <kernel_trace.csv> [first fraction 0..1] [count]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
count = int(sys.argv[3]) if len(sys.argv) > 3 else 80
i0 = int(len(rows) * frac)
prev_end = None
for r in rows[i0:i0 + count]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").replace("vsba::", "")
    name = name.split("(")[0].split("<")[0][:28]
    print("%-28s q%-3s dur %6.1f  gap %7.1f" % (name, r.get("Queue_Id", "?"), (e - s) / 1e3, (s - prev_end) / 1e3 if prev_end else 0.0))
    prev_end = max(prev_end or e, e)
