"""dev tool: soak test of the matcher's in-launch fold (ticket hand-off, per-stream scratch): two different 10 000 x 10 000
workloads alternate on two streams, back to back without host synchronisation, for SECONDS (default 30); every 512 launches the
outputs are compared on the device with the results of the first launches (which tests/test_gpu_match.py checks against the
oracle)."""
import _env  # noqa: F401
import sys
import time

import torch

from visual_slam_amd import Context
from visual_slam_amd.workloads import match_workload

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
ctx = Context(0)
s0 = torch.cuda.ExternalStream(ctx.stream)
s1 = torch.cuda.Stream()
sets = []
with torch.cuda.stream(s0):
    for seed in (0, 1):
        q, t = match_workload(10000, 10000, seed=seed)
        dq, dt = torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda()
        ri = torch.empty((10000, 2), dtype=torch.int32, device="cuda")
        rd = torch.empty((10000, 2), dtype=torch.int32, device="cuda")
        ctx.hamming_knn2_dev(dq.data_ptr(), 10000, dt.data_ptr(), 10000, ri.data_ptr(), rd.data_ptr())
        sets.append((dq, dt, ri, rd))
    outs = [(torch.empty_like(sets[0][2]), torch.empty_like(sets[0][3])) for _ in range(2)]
s0.synchronize()
s1.wait_stream(s0)
t0 = last = time.time()
n = 0
while time.time() - t0 < seconds:
    for k in range(512):
        which = k & 1
        st = s0 if which == 0 else s1
        dq, dt, _, _ = sets[which]
        oi, od = outs[which]
        ctx.hamming_knn2_dev(dq.data_ptr(), 10000, dt.data_ptr(), 10000, oi.data_ptr(), od.data_ptr(), stream=st.cuda_stream)
    s0.synchronize()
    s1.synchronize()
    n += 512
    for which in (0, 1):
        assert torch.equal(outs[which][0], sets[which][2]) and torch.equal(outs[which][1], sets[which][3]), "after %d launches" % n
        outs[which][0].zero_()
        outs[which][1].zero_()
    torch.cuda.synchronize()
    if time.time() - last > 10:
        last = time.time()
        print("%d launches, outputs identical so far" % n, flush=True)
print("soak ok: %d launches of the 10k x 10k match on two streams, every checked output identical" % n)
ctx.close()
