"""CPU: the C-ABI library loads and exports every symbol include/vslam_hip.h declares; the product fails loudly
without a GPU (no CPU fallback)."""
import os
import re
import subprocess

import pytest

from conftest import ROOT


def _declared(header="vslam_hip.h"):
    txt = open(os.path.join(ROOT, "include", header)).read()
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
    assert lib.vs_abi_version() == 5


def test_no_exported_symbol_is_undeclared():
    """Every vs_* symbol the library exports is declared in include/vslam_hip.h (the drop-in boundary) or in
    include/vslam_hip_dev.h (tuning and profiling switches of tests, tools and bench.py) -- and the developer header declares
    nothing the library lacks, with the binding's HOOKS table matching it name by name and parameter by parameter."""
    from visual_slam_amd import _capi
    out = subprocess.run(["nm", "-D", "--defined-only", _capi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (vs_\w+)", out))
    stable, dev = set(_declared()), set(_declared("vslam_hip_dev.h"))
    assert not (stable & dev)
    assert exported == stable | dev, sorted(exported ^ (stable | dev))
    assert dev == set(_capi.HOOKS), sorted(dev ^ set(_capi.HOOKS))
    protos = _header_prototypes("vslam_hip_dev.h")
    for name, (_, argtypes) in _capi.HOOKS.items():
        assert len(protos[name]) == len(argtypes), (name, protos[name], argtypes)


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


def _header_prototypes(header="vslam_hip.h"):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    protos = {}
    for ret, name, params in re.findall(r"\b(int|void\s*\*|const char\s*\*)\s*(vs_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", txt):
        protos[name] = [p.strip() for p in params.split(",") if p.strip() and p.strip() != "void"]
    return protos


def test_binding_pointee_types_match_the_header():
    """The ctypes binding passes plain addresses (void*) for speed; the element type of every pointer parameter is carried
    by a tag (visual_slam_amd/_capi.py) that _capi.ptr() enforces on the array it is given.  Here every tag -- and every
    scalar type and the parameter count -- is compared with the prototype in include/vslam_hip.h, so a drift between
    header and binding is caught without a GPU."""
    import ctypes as C
    from visual_slam_amd import _capi
    scalars = {"int": C.c_int, "double": C.c_double, "size_t": C.c_size_t, "uint64_t": C.c_uint64, "unsigned long long": C.c_uint64}
    protos = _header_prototypes()
    assert set(protos) == set(_capi.SIGNATURES)
    checked = 0
    for name, (_, argtypes) in _capi.SIGNATURES.items():
        params = protos[name]
        assert len(params) == len(argtypes), (name, params, argtypes)
        for p, a in zip(params, argtypes):
            if "*" in p:
                pointee = re.sub(r"\bconst\b", "", p.split("*")[0]).strip()
                if p.count("*") == 2 or pointee in ("vs_ba_problem", "vs_ba_result"):
                    assert not hasattr(a, "ctype"), (name, p)  # vs_ctx** / void** / struct pointers: ctypes POINTER types
                    continue
                assert getattr(a, "ctype", None) == pointee, (name, p, getattr(a, "ctype", a))
                checked += 1
            else:
                ctype = re.sub(r"\s+\w+$", "", p).strip()
                assert scalars[ctype] is a, (name, p, a)
    assert checked > 100


def test_ptr_refuses_a_wrong_element_type():
    import numpy as np
    from visual_slam_amd import _capi
    a = np.zeros((4, 2), np.float32)
    assert _capi.ptr(a, _capi.c_f32p) == a.ctypes.data
    with pytest.raises(TypeError):
        _capi.ptr(a, _capi.c_f64p)
    with pytest.raises(TypeError):
        _capi.ptr(np.zeros((4, 4), np.uint8)[:, ::2], _capi.c_u8p)
