#!/usr/bin/env python3
"""dev tool: in-kernel phase stamps of hamming_knn2_kernel at cfg3 (10 000 x 10 000), per workgroup (vs_match_stamps).
env: TSTAGE (1 | 2), N warm launches before the stamped one."""
import _env  # noqa: F401
import ctypes as C
import os

import numpy as np
import torch

from visual_slam_amd import Context, _capi
from visual_slam_amd.workloads import match_workload

nq = int(os.environ.get("NQ", "10000"))
nt = int(os.environ.get("NT", "10000"))
ctx = Context(0)
lib = _capi.load()
stream = torch.cuda.ExternalStream(ctx.stream)
q_np, t_np = match_workload(nq, nt)
for ts in [int(v) for v in os.environ.get("TSTAGE", "1,2").split(",")]:
    ctx.tune_match(tstage=ts)
    with torch.cuda.stream(stream):
        q, t = torch.from_numpy(q_np).cuda(), torch.from_numpy(t_np).cuda()
        idx = torch.empty((nq, 2), dtype=torch.int32, device="cuda")
        dst = torch.empty((nq, 2), dtype=torch.int32, device="cuda")
        for _ in range(int(os.environ.get("N", "20"))):
            ctx.hamming_knn2_dev(q.data_ptr(), nq, t.data_ptr(), nt, idx.data_ptr(), dst.data_ptr())
        lib.vs_match_stamps(ctx.handle, 1)
        ctx.hamming_knn2_dev(q.data_ptr(), nq, t.data_ptr(), nt, idx.data_ptr(), dst.data_ptr())
        lib.vs_match_stamps(ctx.handle, 0)
        stream.synchronize()
    out = np.zeros((8192, 8))
    rows = lib.vs_match_stamps_read(ctx.handle, out.ctypes.data, len(out))
    s = out[:rows]
    hw = s[:, 7].astype(np.int64)
    xcc, cu, se = hw >> 16, (hw >> 8) & 15, (hw >> 13) & 7

    def q3(v):
        v = v[v > 0]
        return "n/a" if not len(v) else "min %.2f  p50 %.2f  p90 %.2f  max %.2f" % (v.min(), np.median(v), np.quantile(v, 0.9), v.max())
    pub = s[:, 4] > 0
    fold = s[:, 5] > 0
    cyc = -np.where(pub, s[:, 5], s[:, 4])
    s = np.maximum(s, 0.0)
    print("---- tstage %d: %d workgroups (%d publish, %d fold); microseconds since the first stamp of the launch" % (ts, rows, pub.sum(), fold.sum()))
    print("start                 :", q3(s[:, 0] + 1e-9))
    print("operands arrived      :", q3(s[:, 1]))
    print("  (after own start)   :", q3(s[:, 1] - s[:, 0]))
    print("wave 0 scan done      :", q3(s[:, 2]))
    print("  scan duration       :", q3(s[:, 2] - s[:, 1]))
    print("all waves scan done   :", q3(s[:, 3]))
    print("partial stored        :", q3(s[pub, 4]))
    print("fold: words arrived   :", q3(s[fold, 5]))
    print("  after own scan end  :", q3(s[fold, 5] - s[fold, 3]))
    print("results written / end :", q3(s[:, 6]))
    # shader-clock cycles from start to the end of wave 0's scan (raw count in the slot the role leaves free; the reader scaled
    # it like a time stamp: undo) against the wall-clock time of the same interval
    dt = s[:, 2] - s[:, 0]
    ok = (dt > 5) & (cyc > 0)
    print("cycle counter / wall clock over start..scan end: %s  (counts per microsecond)" % q3((cyc[ok] + 0.0) / dt[ok]))
    last_scan = s[:, 3].max()
    print("last scan end %.2f -> last result %.2f  (tail %.2f us)" % (last_scan, s[:, 6].max(), s[:, 6].max() - last_scan))
    # per compute unit: when did its last workgroup finish scanning?
    key = xcc * 1000 + se * 100 + cu
    ends = np.array([s[key == k, 3].max() for k in np.unique(key)])
    per = np.array([(key == k).sum() for k in np.unique(key)])
    print("compute units seen %d, workgroups per unit min %d max %d; last scan end per unit: %s" % (len(ends), per.min(), per.max(), q3(ends)))
    for x in np.unique(xcc):
        print("  XCC %d: workgroups %d, scan end %s" % (x, (xcc == x).sum(), q3(s[xcc == x, 3])))
ctx.close()
