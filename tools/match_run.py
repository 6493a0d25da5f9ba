"""dev tool: N back-to-back launches of the cfg3 match (10 000 x 10 000) for rocprofv3 counter passes.
env: TSTAGE=0|1 (train rows from SGPRs / staged through LDS), BLOCKS (0 = automatic plan), N (launches)."""
import _env  # noqa: F401
import os

import torch

from visual_slam_amd import Context
from visual_slam_amd.workloads import match_workload

nq = nt = 10000
ctx = Context(0)
ctx.tune_match(target_blocks=int(os.environ.get("BLOCKS", "0")), tstage=int(os.environ.get("TSTAGE", "1")))
stream = torch.cuda.ExternalStream(ctx.stream)
q_np, t_np = match_workload(nq, nt)
with torch.cuda.stream(stream):
    q, t = torch.from_numpy(q_np).cuda(), torch.from_numpy(t_np).cuda()
    idx = torch.empty((nq, 2), dtype=torch.int32, device="cuda")
    dst = torch.empty((nq, 2), dtype=torch.int32, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        ctx.hamming_knn2_dev(q.data_ptr(), nq, t.data_ptr(), nt, idx.data_ptr(), dst.data_ptr())
    n = int(os.environ.get("N", "30"))
    if os.environ.get("FENCE", "0") == "1":  # emulate bench.py's fence: the GPU goes idle before the burst
        stream.synchronize()
        torch.cuda.synchronize()
    e0.record(stream)
    for _ in range(n):
        ctx.hamming_knn2_dev(q.data_ptr(), nq, t.data_ptr(), nt, idx.data_ptr(), dst.data_ptr())
    e1.record(stream)
    stream.synchronize()
print("tstage %s: %.2f us per launch" % (os.environ.get("TSTAGE", "1"), e0.elapsed_time(e1) / n * 1e3))
ctx.close()
