import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
from visual_slam_amd import Context
import visual_slam_amd.context as vctx
from visual_slam_amd.sharded import ShardedMatcher
from visual_slam_amd.workloads import match_workload
ctx = Context(0); vctx._DEFAULT = ctx
m = ShardedMatcher(); st = m.torch_stream(); torch.cuda.set_stream(st)
qn, tn = match_workload(10000, 10000)
q = torch.from_numpy(qn).cuda(); t = torch.from_numpy(tn).cuda()
for _ in range(20): m.collect(m.submit(q, t, 10000))
torch.cuda.synchronize()
# host-only cost: time N submits+collects without waiting for the GPU
N=200
t0=time.perf_counter()
tk=None
for _ in range(N):
    t2 = m.submit(q,t,10000)
    if tk is not None: m.collect(tk)
    tk=t2
m.collect(tk)
t1=time.perf_counter()
torch.cuda.synchronize()
t2_=time.perf_counter()
print("host enqueue per step %.1f us ; incl drain %.1f us" % ((t1-t0)/N*1e6, (t2_-t0)/N*1e6))
t0=time.perf_counter()
for _ in range(N): m.knn2_local_shard(q,t)
t1=time.perf_counter(); torch.cuda.synchronize(); t2_=time.perf_counter()
print("local only: host enqueue per step %.1f us ; incl drain %.1f us" % ((t1-t0)/N*1e6, (t2_-t0)/N*1e6))
dist.destroy_process_group()
