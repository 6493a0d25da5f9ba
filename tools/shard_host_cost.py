"""dev tool: host time of one sharded step (ShardedMatcher.plan().submit() = vs_hamming_knn2_sharded_dev) with and without the
collective, world size 1 (run under torch.distributed.run --nproc-per-node 1 for the collective variant)."""
import _env  # noqa: F401
import os
import time

import torch
import torch.distributed as dist

from visual_slam_amd import Context
from visual_slam_amd.sharded import ShardedMatcher
from visual_slam_amd.workloads import match_workload
import visual_slam_amd.context as vctx

force = "WORLD_SIZE" in os.environ
torch.cuda.set_device(0)
if force:
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
ctx = Context(0)
vctx._DEFAULT = ctx
m = ShardedMatcher(force_collective=force)
st = m.torch_stream()
torch.cuda.set_stream(st)
q_np, t_np = match_workload(10000, 10000)
q, t = torch.from_numpy(q_np).cuda(), torch.from_numpy(t_np).cuda()
plan = m.plan(q, t, 10000, in_flight=2)
pend = []
for _ in range(50):
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
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("collective=%s path=%s: %.1f us per step wall; host: submit %.1f us, collect %.1f us" % (force, m.collective_path(), dt / n * 1e6, t_sub / n * 1e6, t_col / n * 1e6))
if force:
    m.close()
    dist.destroy_process_group()
ctx.close()
