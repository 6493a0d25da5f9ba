"""dev tool: does the number of live HIP streams in the process change the overlap of the pipelined tracking period?
(round 3: bench.py's plans left four torch streams alive and the pipelined leg fell from 7300 to 4800 frames/s)"""
import _env  # noqa: F401
import statistics
import time

import torch

from visual_slam_amd import Context
from visual_slam_amd.harness import load_sequence, track_sequence_resident

ctx = Context(0)
frames, depth0 = load_sequence(20)
frames = [ctx.pin(f) for f in frames]


def leg(tag):
    for pipelined in (False, True):
        track_sequence_resident(ctx, frames[:4], depth0, pipelined=pipelined)
        ts = []
        for _ in range(15):
            t0 = time.perf_counter()
            track_sequence_resident(ctx, frames, depth0, pipelined=pipelined)
            ts.append(time.perf_counter() - t0)
        print("%-34s pipelined=%d  median %.1f us/frame  min %.1f" % (tag, pipelined, statistics.median(ts) / 20 * 1e6, min(ts) / 20 * 1e6), flush=True)


import os
if os.environ.get("ORDER") == "before":  # streams that exist BEFORE the context creates its front-half stream (first track_begin)
    pre = [torch.cuda.Stream() for _ in range(int(os.environ.get("NPRE", "4")))]
    y = torch.zeros(16, device="cuda")
    for s in pre:
        with torch.cuda.stream(s):
            y += 1
    torch.cuda.synchronize()
    leg("%d streams created before" % len(pre))
else:
    leg("fresh process")
extra = [torch.cuda.Stream() for _ in range(4)]
x = torch.zeros(16, device="cuda")
for s in extra:
    with torch.cuda.stream(s):
        x += 1
torch.cuda.synchronize()
leg("4 extra torch streams alive")
extra2 = [torch.cuda.Stream() for _ in range(8)]
for s in extra2:
    with torch.cuda.stream(s):
        x += 1
torch.cuda.synchronize()
leg("12 extra torch streams alive")
del extra, extra2
import gc
gc.collect()
torch.cuda.synchronize()
leg("extra streams released")
ctx.close()
