"""dev tool: the chain of back halves in a rocprofv3 kernel trace of tools/resident_prof.py -- per frame of the LAST pipelined
period: start-to-start distance of consecutive pnp_ransac_kernel launches and where it goes (kernel durations on that queue,
idle time between them)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1], r["Queue_Id"]) for r in rows))
pnp = [i for i, e in enumerate(ev) if e[2].startswith("pnp_ransac")]
last = pnp[-19:]
# the back halves alternate between two queues (chained form): the chain is every PnP / motion-only solve / publish launch
names = ("pnp_ransac", "ba_motion", "track_publish", "__amd_rocclr_copyBuffer")
qs = sorted({ev[i][3] for i in last})
chain = [e for e in ev[last[0]:] if e[3] in qs and e[2].startswith(names)]
print("frames %d, queue(s) %s" % (len(last), ", ".join(qs)))
starts = [ev[i][0] for i in last]
d = [(b - a) / 1e3 for a, b in zip(starts, starts[1:])]
print("start-to-start of PnP launches: mean %.1f us, min %.1f, max %.1f" % (sum(d) / len(d), min(d), max(d)))
busy, idle, prev_end = {}, 0.0, None
t_end = starts[-1]
for s, e, n, _ in chain:
    if s >= t_end:
        break
    if n.startswith("pnp_ransac") and prev_end is not None and s < prev_end:
        s = prev_end  # resident early (other queue), waiting in-kernel for the previous solve: count from when that ended
    busy[n] = busy.get(n, 0.0) + (e - s) / 1e3
    if prev_end is not None:
        idle += max(0, s - prev_end) / 1e3
    prev_end = max(prev_end or 0, e)
ends = [e for s_, e, n, _ in chain if n.startswith("ba_motion") and e <= chain[-1][1]][-19:]
de = [(b - a) / 1e3 for a, b in zip(ends, ends[1:])]
if de:
    print("end-to-end of consecutive motion-only solves: mean %.1f us, min %.1f, max %.1f" % (sum(de) / len(de), min(de), max(de)))
nf = len(starts) - 1
print("per frame on the chain (a PnP launch resident early counts from the end of the previous solve): " + ", ".join("%s %.1f" % (k, v / nf) for k, v in busy.items()) + ", idle %.1f us" % (idle / nf))
