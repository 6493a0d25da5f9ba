"""dev tool: one rank, 10k x 10k, with a (one-rank) RCCL all-gather in the step: wall time per step and host time of submit / collect
for the plan modes -- steps in flight, buffer sets, static inputs or not."""
import _env  # noqa: F401
import os
import time

import torch
import torch.distributed as dist

from visual_slam_amd import Context
from visual_slam_amd.sharded import ShardedMatcher
from visual_slam_amd.workloads import match_workload
import visual_slam_amd.context as vctx

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
ctx = Context(0)
vctx._DEFAULT = ctx
q_np, t_np = match_workload(10000, 10000)
for force in (True, False):
    m = ShardedMatcher(force_collective=force)
    st = m.torch_stream()
    torch.cuda.set_stream(st)
    q, t = torch.from_numpy(q_np).cuda(), torch.from_numpy(t_np).cuda()
    for in_flight, buffers, static in ((1, 4, True), (1, 8, True), (1, 4, False), (2, 4, True), (2, 2, True)):
        plan = m.plan(q, t, 10000, single_stream=in_flight == 1, in_flight=max(in_flight, 2), buffers=buffers, static_inputs=static)
        pend = []
        for _ in range(60):
            s = plan.submit()
            if pend:
                plan.collect(pend.pop())
            pend.append(s)
        torch.cuda.synchronize()
        n = 400
        t_sub = t_col = 0.0
        t0 = time.perf_counter()
        for _ in range(n):
            a = time.perf_counter()
            s = plan.submit()
            b = time.perf_counter()
            if pend:
                plan.collect(pend.pop())
            c = time.perf_counter()
            pend.append(s)
            t_sub += b - a
            t_col += c - b
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("collective=%-5s in_flight=%d buffers=%d static_inputs=%-5s: %.1f us per step wall (host loop alone %.1f); submit %.1f us, collect %.1f us; streams %s" % (
            force, in_flight, buffers, static, dt / n * 1e6, t_host / n * 1e6, t_sub / n * 1e6, t_col / n * 1e6,
            sorted({hex(s.cuda_stream)[-5:] for s in plan.streams})), flush=True)
        pend = []
    m.close()
dist.destroy_process_group()
ctx.close()
