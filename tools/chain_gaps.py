"""dev tool: the chain of back halves in a rocprofv3 kernel trace of tools/resident_prof.py -- per frame of the LAST pipelined
period: start-to-start distance of consecutive pnp_ransac_kernel launches and where it goes (kernel durations on that queue,
idle time between them)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1], r["Queue_Id"]) for r in rows))
pnp = [i for i, e in enumerate(ev) if e[2].startswith("pnp_ransac")]
last = pnp[-19:]
q = ev[last[0]][3]
chain = [e for e in ev[last[0]:] if e[3] == q]
print("frames %d, queue %s" % (len(last), q))
starts = [ev[i][0] for i in last]
d = [(b - a) / 1e3 for a, b in zip(starts, starts[1:])]
print("start-to-start of PnP launches: mean %.1f us, min %.1f, max %.1f" % (sum(d) / len(d), min(d), max(d)))
busy, idle, prev_end = {}, 0.0, None
t_end = starts[-1]
for s, e, n, _ in chain:
    if s >= t_end:
        break
    busy[n] = busy.get(n, 0.0) + (e - s) / 1e3
    if prev_end is not None:
        idle += max(0, s - prev_end) / 1e3
    prev_end = max(prev_end or 0, e)
nf = len(starts) - 1
print("per frame on that queue: " + ", ".join("%s %.1f" % (k, v / nf) for k, v in busy.items()) + ", idle %.1f us" % (idle / nf))
