#!/usr/bin/env python3
"""Resources of every kernel in a built libvslam_hip.so, read from the code objects' metadata notes (no GPU needed):
scratch bytes per lane (.private_segment_fixed_size), VGPRs / AGPRs / SGPRs, static LDS, dynamic stack.

    python tools/kernel_meta.py [path/to/libvslam_hip.so]

tests/test_kernel_resources.py uses kernel_meta() to assert the build properties chained tracking relies on (a kernel with
scratch memory may not be able to start while the kernel that waits for it in-kernel is resident: DESIGN.md 6b)."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIELDS = {"private_segment_fixed_size": "scratch", "vgpr_count": "vgpr", "agpr_count": "agpr", "sgpr_count": "sgpr",
          "group_segment_fixed_size": "lds", "uses_dynamic_stack": "dynamic_stack", "max_flat_workgroup_size": "max_threads",
          "vgpr_spill_count": "vgpr_spills", "sgpr_spill_count": "sgpr_spills"}


def kernel_meta(so_path=None):
    """{demangled kernel name: {scratch, vgpr, agpr, sgpr, lds, dynamic_stack, ...}} of every gfx950 kernel in the library."""
    so_path = so_path or os.path.join(ROOT, "visual_slam_amd", "libvslam_hip.so")
    tmp = tempfile.mkdtemp(prefix="vs_kmeta_")
    try:
        local = os.path.join(tmp, "lib.so")  # llvm-objdump writes the bundles next to its input: work on a copy
        shutil.copy(so_path, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, capture_output=True, cwd=tmp)
        out = {}
        for f in sorted(os.listdir(tmp)):
            if "gfx950" not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, f)], check=True,
                                   capture_output=True, text=True).stdout
            import yaml
            doc = notes[notes.index("amdhsa.kernels:"):]
            doc = doc[:doc.rindex("\n...")] if "\n..." in doc else doc
            for k in yaml.safe_load(doc)["amdhsa.kernels"]:
                out[k[".name"]] = {short: k[field] for field, short in ((".%s" % f, s_) for f, s_ in FIELDS.items()) if field in k}
        if not out:
            raise RuntimeError("no gfx950 kernel metadata found in %s" % so_path)
        names = list(out)
        dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
        return {d.replace("(anonymous namespace)::", "").replace("void ", "", 1) if d.startswith("void ") else d.replace(
            "(anonymous namespace)::", ""): out[n] for n, d in zip(names, dem)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def find(meta, prefix):
    """entries whose demangled name starts with `prefix` (e.g. 'hamming_knn2_kernel<true, true>(', 'vsba::pnp_ransac_kernel(')"""
    return {k: v for k, v in meta.items() if k.startswith(prefix)}


if __name__ == "__main__":
    meta = kernel_meta(sys.argv[1] if len(sys.argv) > 1 else None)
    print("%-58s %7s %5s %5s %5s %6s" % ("kernel", "scratch", "vgpr", "agpr", "sgpr", "lds"))
    for k in sorted(meta):
        r = meta[k]
        print("%-58s %7d %5d %5d %5d %6d" % (k.split("(")[0][:58], r.get("scratch", -1), r.get("vgpr", -1), r.get("agpr", 0),
                                             r.get("sgpr", -1), r.get("lds", -1)))
