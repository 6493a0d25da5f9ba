"""dev tool: durations of one kernel's launches in launch order, and the gaps between them, from a rocprofv3 kernel trace.
usage: python tools/trace_sequence.py <kernel_trace.csv> <kernel name substring>"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev_end = None
for i, r in enumerate(rows):
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%3d  duration %7.2f us   gap before %8.2f us" % (i, (b - a) / 1e3, (a - prev_end) / 1e3 if prev_end else 0.0))
    prev_end = b
