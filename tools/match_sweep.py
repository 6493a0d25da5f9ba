#!/usr/bin/env python3
"""Interleaved timing of the Hamming-match kernel over launch geometries on one GPU (dev tool, not part of the product).
Usage: python tools/match_sweep.py [nq nt]      BLOCKS=0,1280,2560 (0 = the library's automatic plan)"""
import _env  # noqa: F401  (sys.path + VS_DATASET_DIR)
import os
import statistics
import sys

import numpy as np
import torch

from visual_slam_amd import Context  # noqa: E402
from visual_slam_amd.workloads import match_workload  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
blocks = [int(v) for v in os.environ.get("BLOCKS", "0,1024,1280,1536,2048,2560").split(",")]
stages = [int(v) for v in os.environ.get("TSTAGE", "0,1").split(",")]
# 0: train rows from SGPRs, 1: staged via LDS
ctx = Context(0)
stream = torch.cuda.ExternalStream(ctx.stream)
q_np, t_np = match_workload(nq, nt)
with torch.cuda.stream(stream):
    q = torch.from_numpy(q_np).cuda()
    t = torch.from_numpy(t_np).cuda()
    idx = torch.empty((nq, 2), dtype=torch.int32, device="cuda")
    dst = torch.empty((nq, 2), dtype=torch.int32, device="cuda")
    pk = torch.empty((nq, 4), dtype=torch.int32, device="cuda")
    packed = os.environ.get("PACKED", "0") == "1"

    def launch():
        if packed:
            ctx.hamming_knn2_packed_dev(q.data_ptr(), nq, t.data_ptr(), nt, pk.data_ptr())
        else:
            ctx.hamming_knn2_dev(q.data_ptr(), nq, t.data_ptr(), nt, idx.data_ptr(), dst.data_ptr())
    ref = None
    res = {}
    for rnd in range(5):
      for ts in stages:
        ctx.tune_match(tstage=ts)
        for b in blocks:
            ctx.tune_match(target_blocks=b)
            for _ in range(3):
                launch()
            if os.environ.get("FENCE", "0") == "1":
                stream.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            n = 20
            for _ in range(n):
                launch()
            e1.record(stream)
            stream.synchronize()
            res.setdefault((ts, b), []).append(e0.elapsed_time(e1) / n * 1e3)
            cur = (pk.cpu().numpy()[:, :2].copy(), pk.cpu().numpy()[:, 2:].copy()) if packed else (idx.cpu().numpy().copy(), dst.cpu().numpy().copy())
            if ref is None:
                ref = cur
            assert np.array_equal(ref[0], cur[0]) and np.array_equal(ref[1], cur[1]), (ts, b)
for (ts, b), v in sorted(res.items()):
    med = statistics.median(v)
    print("tstage %d blocks %5d : %8.1f us/call (min %.1f)  %7.0f Gmatches/s" % (ts, b, med, min(v), nq * nt / med / 1e3))
