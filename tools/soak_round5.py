"""dev tool: soak of what round 5 added, for SECONDS each (default 30):
  (1) the full driver with the explicit resident period, frames pipelined and the key-frame decision speculated, alternating key-frame
      rules (frames dropped with their period, vs_track_last_frame fetches): every run compared bit for bit with the first run of
      its rule;
  (2) the class-API driver (lazily materialised local-map copies, bulk write-back), likewise;
  (3) the sharded step in the single-launch mode (kernels on one auxiliary stream, exchange on the other, completed-event waits
      skipped) with a one-rank RCCL communicator: every result compared with the first."""
import _env  # noqa: F401
import os
import sys
import time

import numpy as np

from visual_slam_amd import Context, harness, slam
from visual_slam_amd.workloads import ICL_NUIM_K, match_workload

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
import torch  # noqa: E402  (before the first context: torch initialises the runtime its own way)
torch.cuda.set_device(0)
ctx = Context(0)
frames, depth0 = harness.load_sequence(20)
frames = [ctx.pin(f) for f in frames]
be = slam.Backends(context=ctx)
rules = ((4, 80), (2, 80), (100, 10 ** 6), (7, 80))
for resident in (True, False):
    ref, runs, t_end = {}, 0, time.time() + seconds
    while time.time() < t_end:
        gap, mt = rules[runs % len(rules)]
        r = slam.run_sequence(frames, depth0, ICL_NUIM_K, be, keyframe_gap=gap, min_tracked=mt, resident_ctx=ctx if resident else None)
        pts = np.array([p.location_3d for p in r["map"].points_3d.values()])
        key = (gap, mt)
        if key not in ref:
            ref[key] = (r["poses"].copy(), r["keyframes"], r["tracked"], pts)
        else:
            a = ref[key]
            assert np.array_equal(a[0], r["poses"]) and a[1] == r["keyframes"] and a[2] == r["tracked"] and np.array_equal(a[3], pts), (resident, key, runs)
        runs += 1
    print("soak ok: %d runs of the driver (%s), four key-frame rules in turn, every run bit-identical to the first of its rule; key frames %s" % (
        runs, "explicit resident period, pipelined" if resident else "class API only", {k: len(v[1]) for k, v in ref.items()}), flush=True)

import torch.distributed as dist  # noqa: E402
import visual_slam_amd.context as vctx  # noqa: E402
from visual_slam_amd.sharded import ShardedMatcher  # noqa: E402
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29537")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
vctx._DEFAULT = ctx
m = ShardedMatcher(force_collective=True)
st = m.torch_stream()
torch.cuda.set_stream(st)
q_np, t_np = match_workload(10000, 10000)
q, t = torch.from_numpy(q_np).cuda(), torch.from_numpy(t_np).cuda()
plan = m.plan(q, t, 10000, single_stream=True, buffers=4, static_inputs=True)
first, steps, pend, t_end = None, 0, [], time.time() + seconds
while time.time() < t_end:
    for _ in range(50):
        pend.append(plan.submit())
        if len(pend) > 2:
            i, d = plan.collect(pend.pop(0))
            if steps % 25 == 0:
                st.synchronize()
                got = (i.cpu().numpy().copy(), d.cpu().numpy().copy())
                if first is None:
                    first = got
                else:
                    assert np.array_equal(first[0], got[0]) and np.array_equal(first[1], got[1]), steps
            steps += 1
torch.cuda.synchronize()
m.close()
dist.destroy_process_group()
print("soak ok: %d sharded steps in the single-launch mode with a one-rank all-gather, every 25th result identical to the first" % steps, flush=True)
ctx.close()
