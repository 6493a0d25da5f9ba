/* vslam_hip_dev.h -- developer entry points of libvslam_hip.so: tuning knobs and profiling switches used by tests/, tools/
 * and bench.py.  They are NOT part of the drop-in boundary (include/vslam_hip.h) -- nothing in the reference corresponds to
 * them -- and may change between rounds without an ABI version bump.  All state they touch lives in the context (vs_tuning
 * in csrc/vs_internal.h): two contexts never see each other's knobs.
 *
 * tests/test_abi.py fails on any exported vs_* symbol that neither header declares. */
#ifndef VSLAM_HIP_DEV_H
#define VSLAM_HIP_DEV_H

#include "vslam_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Matcher (csrc/vs_match.hip).  target_blocks: workgroups per launch the chunk planner aims at (0 = automatic);
 * tstage: 1 = train rows staged through LDS (default), 0 = wave-uniform scalar loads (sweep only; neither instantiation has a
 * private segment -- tests/test_kernel_resources.py holds both at scratch 0 -- so both may run inside a device-chained tracking
 * period; the kernels that DO carry scratch on the tracking path are ba_motion_step and ba_motion_persistent<true>, see
 * track_redo / track_ba_batch in csrc/vs_track.hip).  A negative value leaves a knob as it is. */
int vs_tune_match(vs_ctx* ctx, int target_blocks, int tstage);
/* HIP-event pair around every launch of hamming_knn2_kernel on its launch stream while enabled. */
int vs_match_profile(vs_ctx* ctx, int enable);
/* Synchronises, stores the mean kernel duration [ms] of the profiled launches, clears them; returns their number. */
int vs_match_profile_read(vs_ctx* ctx, float* kernel_ms);

/* Per-workgroup phase stamps of hamming_knn2_kernel (wall clock, thread 0): while enabled every launch is stamped.
 * vs_match_stamps_read synchronises and returns the newest stamped launch as rows of 8 doubles, one per workgroup in
 * (chunk, tile) order: microseconds since the launch's first stamp of [0] start, [1] operands arrived, [2] wave 0 through its scan,
 * [3] all waves through, [4] partial stored, [5] all partials arrived (folding workgroups), [6] results written; [7] = XCC_ID *
 * 65536 + HW_ID.  Returns the number of rows (0: nothing stamped, or cap_rows too small). */
int vs_match_stamps(vs_ctx* ctx, int enable);
int vs_match_stamps_read(vs_ctx* ctx, double* out, int cap_rows);

/* Bundle adjustment (csrc/vs_ba.hip).  schur_variant: 0 automatic, 1 tile kernel, 2 ba_schur_small with a linearisation
 * launch per iteration, 3 banded windows on the tile kernel; points_per_workgroup (< 64: ba_schur_small, >= 64: slab size of
 * ba_schur_window); max_slabs: cap on ba_schur_small's slabs; motion_variant: 0 one-launch motion-only solve where it
 * applies, 1 one launch per LM step.  Values out of range leave a knob as it is. */
int vs_tune_ba(vs_ctx* ctx, int schur_variant, int points_per_workgroup, int max_slabs, int motion_variant);
/* Large problems (>= 400 000 observations, observation arrays in pinned memory, grouped by point) build their sparsity
 * structure on the device (csrc/vs_ba_build.hip); on_host = 1 keeps it on the host passes (tests compare the two). */
int vs_tune_ba_structure(vs_ctx* ctx, int on_host);
int vs_ba_structure_on_device(vs_ctx* ctx); /* 1: the newest vs_ba_solve of this context built its structure on the device */

/* Experiment (profiles/tried_and_dropped.md): on = 1 replays every batch of LM slots of vs_ba_solve (<= 10 slots of four launches,
 * the export kernel, the read-back) as ONE captured hipGraph; 0 = launch by launch (default); 2 = launch by launch with the
 * uploads waited for first, so that *last_batch_us covers the same interval as the graph form's; any other value leaves it.
 * *last_batch_us (may be NULL) = wall microseconds of the newest batch, graph launch (or first enqueue) to results on the host. */
int vs_tune_ba_graph(vs_ctx* ctx, int on, double* last_batch_us);
/* Dense solve of the reduced camera system in LDS (ba_solve_block): 0 automatic -- square storage up to 126 unknowns (21 free
 * cameras), the lower triangle packed from 127 to 198 (33) --, 1 never packed (systems beyond 126 take the blocked factorisation in
 * HBM), 2 packed at every size that fits.  Same operations in the same order either way; tests compare the bits. */
int vs_tune_ba_solve(vs_ctx* ctx, int packed_mode);

/* Which kernels the newest vs_ba_solve of this context took.  out6[0] Schur complement: 0 ba_schur (general, one slab of points per
 * workgroup), 1 ba_schur_tile (tiles of 10 x 10 camera blocks), 2 ba_schur_small (single tile), 3 ba_schur_window (banded, FP64
 * matrix cores), 4 none (motion-only: block diagonal); out6[1] reduced system: 0 ba_solve_block (LDS, <= 21 free cameras),
 * 1 ba_chol_band + ba_chol_finish, 2 ba_chol_panel / ba_chol_update per block column + ba_chol_finish, 3 element-wise last
 * resort, 4 none (motion-only); out6[2] unknowns of the reduced system; out6[3] band width handed to the factorisation (0: dense);
 * out6[4] camera tiles; out6[5] widest camera span of a banded window's slab (0: no window plan). */
int vs_ba_last_path(vs_ctx* ctx, int* out6);

/* Phase stamps of pnp_ransac_kernel (csrc/vs_pnp.hip).  vs_pnp_profile_read synchronises and returns the stamps of the
 * newest profiled launch as microseconds since the launch's first stamp: rows 0..H-1 = hypotheses, row H = the finishing
 * workgroup; 8 doubles per row, 0 = not reached.  Returns the number of rows (0 when nothing was profiled). */
int vs_pnp_profile(vs_ctx* ctx, int enable);
int vs_pnp_profile_read(vs_ctx* ctx, double* out, int cap_rows);

/* Per-step phase stamps of ba_motion_persistent inside a tracking period (camera 0's workgroup, thread 0).
 * vs_mo_profile_read synchronises and returns the newest stamped solve as rows of 8 doubles, one per LM step: columns 0..6 in
 * shader-clock cycles since the solve's first stamp ([0] step entered, [1] all cameras' partials arrived, [2] decision taken,
 * [3] linearised + reduced, [4] 6x6 solved + trial record, [5] trial chi2 summed, [6] partials posted; 0 = phase skipped), column
 * 7 the wall clock in microseconds since the first step.  Returns the number of steps (cap_rows must be >= 64). */
int vs_mo_profile(vs_ctx* ctx, int enable);
int vs_mo_profile_read(vs_ctx* ctx, double* out, int cap_rows);

/* Tracking period (csrc/vs_track.hip).  inject_fault = 1: the next CHAINED back half's PnP launch waits for a front-half tag
 * nobody publishes -- every workgroup's bounded wait runs out, the frame is then redone host-paced (track_redo); 0: nothing.
 * *recoveries_out (may be NULL) = back halves redone so far on this context. */
int vs_track_debug(vs_ctx* ctx, int inject_fault, int* recoveries_out);

/* Raises the pinned "a train chunk did not report" word of every match scratch set in use, as a folding workgroup of
 * hamming_knn2_kernel that ran out of its bounded wait does; returns how many.  Tests of where that report surfaces
 * (vs_match_status, the tracking period's hand-out, vs_track_end, a stream's next launch). */
int vs_match_debug_raise(vs_ctx* ctx);

/* Allocation poisoning (tests).  byte in 0..255: every device buffer the context allocates FROM NOW ON (vs_reserve) is filled
 * with that byte before first use; -1: off (default).  A kernel or protocol that relies on what hipMalloc happens to return --
 * flags assumed zero, tags assumed stale -- fails its parity test under a poison byte instead of once in a blue moon. */
int vs_debug_poison_alloc(vs_ctx* ctx, int byte);

#ifdef __cplusplus
}
#endif
#endif /* VSLAM_HIP_DEV_H */
