"""CPU: the C-ABI library loads and exports every symbol include/vslam_hip.h declares; the product fails loudly
without a GPU (no CPU fallback)."""
import os
import re
import subprocess

import pytest

from conftest import ROOT


def _declared():
    txt = open(os.path.join(ROOT, "include", "vslam_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vs_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from visual_slam_amd import _capi
    lib = _capi.load()
    assert lib._vs_missing == []
    names = _declared()
    assert len(names) >= 15
    out = subprocess.run(["nm", "-D", "--defined-only", _capi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (vs_\w+)", out))
    assert set(names) <= exported, sorted(set(names) - exported)
    assert set(names) == set(_capi.SIGNATURES), sorted(set(names) ^ set(_capi.SIGNATURES))
    assert lib.vs_abi_version() == 2


def test_struct_layouts_match_the_header():
    import ctypes as C
    from visual_slam_amd import _capi
    assert C.sizeof(_capi.BAProblem) == 4 * 4 + 11 * 8 + 6 * 8 + 2 * 4
    assert C.sizeof(_capi.BAResult) == 4 * 8 + 3 * 8 + 4 * 4 + 8 + 2 * 4
    from oracle import oracle
    assert C.sizeof(oracle.BAProblem) == C.sizeof(_capi.BAProblem) and C.sizeof(oracle.BAResult) == C.sizeof(_capi.BAResult)


def test_no_cpu_fallback():
    import visual_slam_amd
    from visual_slam_amd import _capi
    if _capi.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(visual_slam_amd.VsError):
        visual_slam_amd.Context(0)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "visual_slam_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "vs_oracle" not in txt.replace(
                    "oracle/vs_oracle.c", ""), os.path.join(dirpath, f)
