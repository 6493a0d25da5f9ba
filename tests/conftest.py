import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN = os.path.join(ROOT, "tests", "golden")
os.environ.setdefault("VS_DATASET_DIR", os.path.join(GOLDEN, "icl_nuim"))  # the harness ships no data of its own


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Built on demand with gcc."""
    from oracle import oracle as o
    o.build()
    o.load()
    return o


@pytest.fixture(scope="session")
def brief_pattern():
    import re
    txt = open(os.path.join(ROOT, "include", "vs_brief_pattern.h")).read()
    rows = re.findall(r"\{\s*(-?\d+),\s*(-?\d+),\s*(-?\d+),\s*(-?\d+)\}", txt)
    pat = np.array(rows, dtype=np.int64)
    assert pat.shape == (256, 4)
    return pat


@pytest.fixture(scope="session")
def vs():
    """The product library on a GPU.  Fails loudly if the HIP library is missing; skips only when the machine has
    no GPU at all (the CPU container)."""
    from visual_slam_amd import _capi
    lib = _capi.load()  # raises if libvslam_hip.so is missing: no CPU fallback exists
    if _capi.device_count() == 0:
        pytest.skip("no GPU in this machine")
    from visual_slam_amd import Context
    ctx = Context(0)
    poison = os.environ.get("VS_TEST_POISON")  # e.g. 127: the whole GPU suite on a context whose every device buffer starts out as 0x7F
    if poison:
        ctx.debug_poison_alloc(int(poison, 0))
    return ctx


def icl_frame(i):
    from PIL import Image
    p = os.path.join(GOLDEN, "icl_nuim", "rgb", "%d.png" % i)
    return np.ascontiguousarray(np.asarray(Image.open(p).convert("RGB"))[:, :, ::-1])  # BGR as cv2.imread
