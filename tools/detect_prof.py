"""dev tool: kernel-only timing of the fused detect+describe path on a resident frame."""
import _env  # noqa: F401  (sys.path + VS_DATASET_DIR)
import os, sys, time
import numpy as np, torch
from visual_slam_amd import Context, harness
ctx = Context(0)
frames, _ = harness.load_sequence(4)
st = torch.cuda.ExternalStream(ctx.stream)
with torch.cuda.stream(st):
    img = torch.zeros((481, 640, 3), dtype=torch.uint8, device="cuda")
    img[:480] = torch.from_numpy(frames[1]).cuda()
    xy = torch.empty((3000, 2), dtype=torch.float32, device="cuda"); sc = torch.empty(3000, dtype=torch.uint8, device="cuda")
    desc = torch.empty((3000, 32), dtype=torch.uint8, device="cuda"); n = torch.zeros(1, dtype=torch.int32, device="cuda")
    import ctypes as C
    lib = ctx._lib
    def run():
        rc = lib.vs_detect_describe_bgr_dev(ctx._h, img.data_ptr(), 640, 480, 1920, 20, 3000, xy.data_ptr(), sc.data_ptr(), desc.data_ptr(), n.data_ptr(), None)
        assert rc == 0
    for _ in range(5): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(100): run()
    e1.record(st); st.synchronize()
    print("detect+describe (resident frame): %.1f us per frame, n=%d" % (e0.elapsed_time(e1) * 10, int(n.item())))
    ref = ctx.detect_describe_bgr(frames[1], 20, 3000)
    assert int(n.item()) == len(ref[0]) and np.array_equal(xy[:len(ref[0])].cpu().numpy(), ref[0]) and np.array_equal(desc[:len(ref[0])].cpu().numpy(), ref[2])
