"""Build properties of libvslam_hip.so that the product relies on, read from the code objects' metadata (CPU only, no GPU).

Chained tracking (vs_track_frame_pipelined) keeps a kernel resident that waits in-kernel for kernels on other
streams.  A kernel with scratch memory cannot start on a queue before the runtime has provided the scratch -- on a fresh process
that provision waited for the very kernels being waited for, and the first chained period timed out (round 3).  No kernel such a
period can launch may therefore have a private segment.  The library checks the same at run time (track_chain_scratch_free);
this test keeps the build from regressing without a GPU."""
import os
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def meta():
    from kernel_meta import kernel_meta
    return kernel_meta()


CHAIN = ["detect_band_kernel<true, true, false, false>(", "select_describe_kernel<true>(", "hamming_knn2_kernel<true>(",
         "hamming_knn2_kernel<false>(", "ratio_compact_kernel(", "track_append_kernel(", "track_publish_kernel(",
         "vsba::pnp_ransac_kernel(", "vsba::ba_motion_persistent<false>("]


@pytest.mark.parametrize("prefix", CHAIN)
def test_kernels_of_a_chained_tracking_period_use_no_scratch(meta, prefix):
    hits = {k: v for k, v in meta.items() if k.startswith(prefix)}
    assert len(hits) == 1, (prefix, sorted(meta))
    (name, r), = hits.items()
    assert r["scratch"] == 0 and not r["dynamic_stack"], (name, r)
    assert r.get("vgpr_spills", 0) == 0 and r.get("sgpr_spills", 0) == 0 or r["scratch"] == 0


def test_headline_kernel_resources(meta):
    """hamming_knn2_kernel<true>: four waves per SIMD (<= 128 VGPRs), LDS for five workgroups per compute unit, no scratch."""
    (r,) = [v for k, v in meta.items() if k.startswith("hamming_knn2_kernel<true>(")]
    assert r["vgpr"] <= 128 and r["scratch"] == 0 and r["lds"] * 5 <= 160 * 1024


def test_every_kernel_is_listed_and_none_allocates_a_dynamic_stack(meta):
    assert len(meta) >= 35
    assert not [k for k, v in meta.items() if v["dynamic_stack"]]
    # kernels known to spill (single-use set-up paths, never on a chain): listed so that a NEW one is noticed
    spills = sorted(k.split("(")[0] for k, v in meta.items() if v["scratch"] > 0)
    assert spills == ["ess_hypothesis_kernel", "vsba::ba_motion_persistent<true>", "vsba::ba_motion_step"], spills
