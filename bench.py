#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X tracking hot path (contract: see the task description / DESIGN.md 5).

A "step" is one pass of the brute-force Hamming 2-NN match (the reference's knnMatch, src/v2/frame.py:23) over one
batch of synthetic descriptors already resident in HBM:
  N = 1 : BASELINE.json configs[2] -- 10 000 x 10 000 x 256-bit, k = 2 (the configuration the metric is quoted on)
  N > 1 : weak scaling of the same -- every rank matches its own 10 000-query shard against the replicated 10 000-row
          train set, then one RCCL all-gather of the per-shard best matches (16 B per query) assembles N*10 000 results.
value = distance evaluations (Q*T) of all ranks / wall time of the K timed steps, in Gmatches/s.  ONE mode for `value`,
`ms_per_step` and `roofline.frac`: one match launch at a time on one stream (at N > 1 the all-gather of step k runs on a
second stream beside the kernel of step k + 1), so kernel time <= ms_per_step holds on the line's face and the kernel's
duration agrees with the rocprofv3 summaries under profiles/.  The throughput with two launches overlapped on two streams
is carried under its own key (`two_in_flight`), never as `value`.

Rank 0 prints ONE JSON line.  Extra objects on the line:
  roofline      dominant kernel (hamming_knn2_kernel), HIP-event timed in this run.  bound = "valu": the kernel is bound
                by integer VALU issue, not by HBM; the SURVEY 8d streamed-operand HBM model and the
                MEASURED HBM traffic are carried as labelled secondary entries; per-kernel entries for the detector and
                the local BA sit under roofline.kernels
  cpu_baseline  the CPU oracle timed on this box's host cores (1 thread and all threads)
  cfg5          BASELINE.json configs[4]: 100 000 x 100 000 query-split over the N ranks (strong scaling), with the
                all-gather, the compute-only time beside it
  cfg2 / local_ba / frames   the single-frame detector, the 10 x 2000 local BA and the 640x480 ICL-NUIM stream
"""
import argparse
import gc
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("VS_DATASET_DIR", os.path.join(ROOT, "tests", "golden", "icl_nuim"))  # fixture frames

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_PER_MATCH = 32           # SURVEY.md 8d streamed-operand model: one 32-byte train descriptor per evaluation
# integer VALU issue on gfx950 (tools/valu_probe*.hip, profiles/r02_valu_probe*.log): v_xor/v_and/v_or/v_add_u32 with
# VGPR-only sources issue one wave64 op per 2 cycles per SIMD, everything else in the loop (v_bcnt, v_min3, v_med3, v_min,
# v_lshl_or) one per 4 cycles.  Main loop of hamming_knn2_kernel (train rows staged through LDS), from its ISA: per 8
# distances 64 v_xor + 64 v_bcnt + 8 v_lshl_or + 4 v_med3 + 4 v_min3 + 4 v_min = 148 VALU instructions.
OPS_2CYCLE, OPS_4CYCLE = 64.0, 84.0
OPS_PER_MATCH = (OPS_2CYCLE + OPS_4CYCLE) / 8.0
CLOCK_HZ, N_SIMD = 2.4e9, 256 * 4
# uniform 4-cycle model (round 2's yard-stick): 16 lanes/clk/SIMD
VALU_PEAK_UNIFORM_4CYCLE = N_SIMD * 16 * CLOCK_HZ                                   # 3.93e13 lane-ops/s
# the guide's nominal figure: every VALU instruction at one wave64 op per 2 cycles per SIMD-32
VALU_PEAK_GUIDE_NOMINAL = N_SIMD * 32 * CLOCK_HZ                                    # 7.86e13 lane-ops/s
# mix-specific ceiling (the roofline `peak`): the 148 instructions of 8 distances need 64*2 + 84*4 = 464 issue cycles
VALU_PEAK_LANE_OPS = N_SIMD * 64 * (OPS_2CYCLE + OPS_4CYCLE) / (2 * OPS_2CYCLE + 4 * OPS_4CYCLE) * CLOCK_HZ   # 5.02e13
# what a dependent xor -> bcnt stream was MEASURED to issue at with 5 waves per SIMD (chain_grp_vv: 1.597 ns per instruction)
VALU_PROBE_MIX_LANE_OPS = N_SIMD * 64 / 1.597e-9                                     # 4.10e13
LAUNCH_BOUNDARY_US = 1.45      # dependent kernel boundary on MI355X (MI355X_MICROARCH.md price list, row "boundary")
PCIE_GBS = 63.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--nq", type=int, default=10000, help="queries per rank")
    ap.add_argument("--nt", type=int, default=10000, help="train descriptors (replicated)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-frames", action="store_true", help="skip the cfg2 / local BA / frames legs")
    ap.add_argument("--no-cfg5", action="store_true")
    ap.add_argument("--in-flight", type=int, default=2, help="steps in flight of the SECONDARY figure `two_in_flight` (2..4); "
                    "`value` is always one launch at a time on one stream")
    ap.add_argument("--launch-timeout", type=float, default=1500.0,
                    help="launcher mode (--gpus N started by hand): seconds after which ranks that have not exited are stopped")
    ap.add_argument("--buffers", type=int, default=0, help="rotating result buffer sets of a step (0: 2, or 4 with a collective)")
    ap.add_argument("--single-stream", action="store_true",
                    help="skip the secondary two-in-flight figure: every launch of the run is then one at a time on one stream "
                         "(the rocprofv3 kernel summaries under profiles/ are taken this way so that kernel durations do not overlap)")
    ap.add_argument("--target-blocks", type=int, default=0, help="tuning: workgroups per launch of the match kernel")
    ap.add_argument("--force-collective", action="store_true",
                    help="rehearsal: run the RCCL all-gather even at world size 1 (exercises the N>1 code path)")
    return ap.parse_args()


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launch_ranks(n, argv, worker=None, timeout=None):
    """`bench.py --gpus N` started by hand (no torchrun, WORLD_SIZE unset): this process becomes a launcher.  It starts N
    fresh rank processes of `worker` (default: this file) with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their
    environment, relays rank 0's standard output, and returns non-zero when any rank fails (the others are stopped: a rank
    alone in a collective would wait for ever).  It runs BEFORE torch or the HIP library is imported: the launcher never
    touches the GPU, so its children are ordinary child processes, not an exec out of a process that holds a device."""
    import subprocess
    assert "torch" not in sys.modules and "visual_slam_amd._capi" not in sys.modules, "the launcher must stay off the GPU"
    cmd = list(worker) if worker else [sys.executable, os.path.abspath(__file__)]
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    import tempfile
    # every rank's standard output goes to a file of its own (a pipe nobody drains could fill up and block the rank): rank 0's
    # is relayed, the others' are shown when a rank fails or the run times out
    outs = [tempfile.TemporaryFile(mode="w+") for _ in range(n)]
    out0 = outs[0]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0", VS_BENCH_LAUNCHED="1")
        procs.append(subprocess.Popen(cmd + list(argv), env=env, stdout=outs[r]))
    deadline = None if timeout is None else time.time() + timeout
    rcs = [None] * n
    failed = None
    while any(rc is None for rc in rcs):
        for r, p in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = p.poll()
                if rcs[r] not in (None, 0) and failed is None:
                    failed = r
        if failed is not None or (deadline is not None and time.time() > deadline):
            for r, p in enumerate(procs):       # exactly the processes started above, by handle
                if rcs[r] is None:
                    p.terminate()
            for r, p in enumerate(procs):
                if rcs[r] is None:
                    try:
                        rcs[r] = p.wait(10)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        rcs[r] = p.wait()
            if failed is None:
                failed = -1
            break
        time.sleep(0.05)
    texts = []
    for f in outs:
        f.seek(0)
        texts.append(f.read())
        f.close()
    out = texts[0]
    if failed is not None:
        sys.stderr.write("bench.py launcher: %s; ranks' exit codes %s\n"
                         % ("timed out after %.0f s (ranks still running were stopped)" % timeout if failed < 0
                            else "rank %d failed" % failed, rcs))
        for r in range(1, n):
            if texts[r].strip():
                sys.stderr.write("---- rank %d standard output\n%s\n" % (r, texts[r][-4000:]))
        sys.stdout.write(out)   # whatever rank 0 managed to say (diagnostics), but the exit code says failure
        return 1
    sys.stdout.write(out)
    sys.stdout.flush()
    return 0


def median_time(fn, reps, warm=3):
    """Median wall seconds of fn() over `reps` calls after `warm` untimed ones."""
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts), ts


def all_threads_rate(fn, threads, seconds=6.0):
    """Calls per second with `threads` host threads calling fn() concurrently (ctypes releases the GIL inside the C
    oracle): the all-cores figure of a sequential CPU algorithm = independent instances side by side."""
    import threading
    stop = time.perf_counter() + seconds
    counts = [0] * threads

    def work(i):
        while time.perf_counter() < stop:
            fn()
            counts[i] += 1

    t0 = time.perf_counter()
    ths = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    return sum(counts) / (time.perf_counter() - t0)


def host_cpu_share():
    """(logical CPUs the OS reports, CPU quota of this container in cores or None): the GPU boxes give one job a share of a
    larger host, so 'all cores' figures say what they ran on."""
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        pass
    return os.cpu_count() or 1, quota


def cpu_trackers_all_cores(cores, seconds=4.0):
    """frames/s of the CPU oracle with the host's cores busy: one sequential tracker per thread on the 20 fixture frames.
    The tracker's glue is Python, so the trackers are spread over worker PROCESSES (a few threads each; one interpreter
    lock would cap the figure); the workers never touch the GPU (no torch, no Context).  Each worker counts the frames it
    completes inside one common wall-clock window.  More threads is not always more frames/s on a shared host: three
    thread counts are run, all are reported, the best one is the figure."""
    import subprocess

    def run(total):
        procs_n = max(1, min(32, total // 4))
        per = max(1, total // procs_n)
        start = time.time() + 5.0 + 0.02 * procs_n   # imports + PNG decode + warm-up happen before the window opens
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-tracker-worker", str(per), repr(start), repr(seconds)]
        procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True) for _ in range(procs_n)]
        frames, late = 0.0, 0
        for p in procs:
            out, _ = p.communicate(timeout=120)
            last = [ln for ln in out.splitlines() if ln.startswith("frames ")]
            if last:
                frames += float(last[-1].split()[1])
                late += int(last[-1].split()[2])
        return {"threads": procs_n * per, "processes": procs_n, "threads_per_process": per, "frames_per_s": frames / seconds,
                "workers_late_for_the_window": late}

    runs = [run(t) for t in sorted({min(cores, 32), min(cores, 96), cores})]
    best = max(runs, key=lambda r: r["frames_per_s"])
    return dict(best, window_s=seconds, host_logical_cpus=cores, cpu_quota_cores=host_cpu_share()[1], sweep=runs,
                note="one sequential CPU-oracle tracker per thread on the same 20 frames (streams side by side), worker "
                     "processes x threads; best of the thread counts in `sweep`")


def cpu_tracker_worker(threads, start, seconds):
    """Worker of cpu_trackers_all_cores (runs in its own process, CPU only): `threads` trackers, frames completed inside
    [start, start + seconds) of the wall clock."""
    import threading
    from oracle import oracle
    from visual_slam_amd.harness import HUBER, load_sequence, track_sequence
    oracle.load()

    def detect(bgr):
        xy, _, desc = oracle.detect_describe_bgr(bgr, 20, 3000)
        return xy, desc

    def match(q, t):
        mq, mt, _ = oracle.match_ratio(q, t, 0.8)
        return mq, mt

    def ba(*problem):
        return oracle.ba_solve(*problem, huber_delta=HUBER, max_iterations=10)

    def pnp(obj, img, K4, pose0, seed=0):
        return oracle.pnp_ransac(obj, img, K4, pose0, seed=seed)

    frames, depth0 = load_sequence(20)
    track_sequence(detect, match, ba, frames[:3], depth0, pnp=pnp)
    late = int(time.time() > start)
    counts = [0.0] * threads

    def work(i):
        while time.time() < start:
            time.sleep(0.001)
        while True:
            t0 = time.time()
            if t0 >= start + seconds:
                break
            track_sequence(detect, match, ba, frames, depth0, pnp=pnp)
            t1 = time.time()
            counts[i] += len(frames) * (min(t1, start + seconds) - t0) / (t1 - t0)  # the run that crosses the end counts pro rata

    ths = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    print("frames %.3f %d" % (sum(counts), late))


def cpu_baseline(nq, nt):
    """Time the CPU oracle (kind 'port': the reference's cv2 path cannot run here) on this host."""
    import tempfile
    from oracle import oracle
    from visual_slam_amd.workloads import match_workload
    lib = None
    try:  # give the CPU its best shot: rebuild for this host's ISA into a scratch file
        tmp = os.path.join(tempfile.gettempdir(), "libvs_oracle_native_%d.so" % os.getpid())
        oracle.build(force=True, lib_path=tmp, extra_cflags=["-O3", "-march=native"])
        lib = oracle.load(tmp)
    except Exception:
        lib = oracle.load()
    q, t = match_workload(nq, nt)
    logical, quota = host_cpu_share()
    oracle.hamming_knn2(q[:512], t, threads=0, lib=lib)  # warm
    # the same thread sweep as the frames / local_ba legs (one machine, described once): the box reports its host's logical
    # CPUs but the job runs on a cgroup share of them, so three thread counts are timed and the best is the figure
    sweep = []
    for th in sorted({min(logical, 32), min(logical, 96), logical}):
        times = []
        t_end = time.time() + 5.0
        while len(times) < 3 or (time.time() < t_end and len(times) < 25):
            t0 = time.perf_counter()
            oracle.hamming_knn2(q, t, threads=th, lib=lib)
            times.append(time.perf_counter() - t0)
        sweep.append({"threads": th, "gmatches_per_s": nq * nt / statistics.median(times) / 1e9, "runs": len(times)})
    best = max(sweep, key=lambda r: r["gmatches_per_s"])
    t0 = time.perf_counter()
    oracle.hamming_knn2(q, t, threads=1, lib=lib)
    t1 = time.perf_counter() - t0
    return {"value": best["gmatches_per_s"], "unit": "Gmatches/s", "cores": best["threads"], "kind": "port",
            "host_logical_cpus": logical, "cpu_quota_cores": quota, "sweep": sweep,
            "sample": "full %dx%d cfg3 workload, CPU oracle (C, -O3 -march=native, OpenMP over queries), median of %d runs at "
                      "%d threads (best of the thread counts in `sweep`; `cores` = threads used)"
                      % (nq, nt, best["runs"], best["threads"]),
            "single_thread_value": nq * nt / t1 / 1e9}


def detector_leg(ctx, torch, stream, cpu=True):
    """(generator: yields its result dict after the GPU part and again after the CPU part, see `two_phase`)
    BASELINE.json configs[1] / SURVEY 8d cfg2: one 640x480 frame, detect + describe.  Kernel-only time from HIP events
    on the launch stream around the two launches (frame resident in HBM), end-to-end through the host ABI (H2D + D2H
    included), on the synthetic frame SURVEY prescribes and on ICL-NUIM frame 0."""
    from visual_slam_amd.harness import load_sequence
    from visual_slam_amd.workloads import synthetic_frame
    out = {}
    frames, _ = load_sequence(1)
    cases = {"synthetic_rng2_blur5": synthetic_frame(640, 480, 2), "icl_nuim_rgb0": frames[0]}
    max_kp = 3000
    for name, bgr in cases.items():
        h, w, _ = bgr.shape
        with torch.cuda.stream(stream):
            d_img = torch.from_numpy(np.ascontiguousarray(bgr)).cuda()
            d_xy = torch.empty((max_kp, 2), dtype=torch.float32, device="cuda")
            d_sc = torch.empty((max_kp + 16,), dtype=torch.uint8, device="cuda")
            d_desc = torch.empty((max_kp, 32), dtype=torch.uint8, device="cuda")
            d_n = torch.zeros((4,), dtype=torch.int32, device="cuda")

            def launch():
                ctx._chk(ctx._lib.vs_detect_describe_bgr_dev(ctx._h, d_img.data_ptr(), w, h, 3 * w, 20, max_kp,
                                                             d_xy.data_ptr(), d_sc.data_ptr(), d_desc.data_ptr(),
                                                             d_n.data_ptr(), None))
            for _ in range(20):
                launch()
            reps = []
            for _ in range(20):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(10):
                    launch()
                e1.record(stream)
                stream.synchronize()
                reps.append(e0.elapsed_time(e1) / 10 * 1e3)
            n_kp = int(d_n.cpu()[0])
        kernel_us = statistics.median(reps)
        pinned = ctx.pin(bgr)
        med, _ = median_time(lambda: ctx.detect_describe_bgr(pinned, 20, max_kp), 30, warm=5)
        # compulsory bytes: BGR in + box sums out and in again + results out
        alg_bytes = 3 * w * h + 2 * (2 * w * h) + n_kp * (8 + 1 + 32)
        entry = {"keypoints": n_kp, "kernel_only_us": kernel_us, "end_to_end_host_abi_us": med * 1e6,
                 "frames_per_s_kernel_only": 1e6 / kernel_us, "frames_per_s_host_abi": 1.0 / med,
                 "algorithmic_bytes": alg_bytes, "hbm_GBps": alg_bytes / (kernel_us * 1e-6) / 1e9,
                 "hbm_frac_of_8TBps": alg_bytes / (kernel_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                 "launch_floor_us": 2 * LAUNCH_BOUNDARY_US,
                 "host_floor_us": 2 * LAUNCH_BOUNDARY_US + 3 * w * h / (PCIE_GBS * 1e3)}
        out[name] = entry
    out["note"] = ("kernels: detect_band_kernel + select_describe_kernel (two launches per frame); latency-bound as SURVEY 8d "
                   "predicts -- bytes / time is a tiny HBM fraction, the floor is two kernel boundaries (+ the 921 KB H2D "
                   "at PCIe rate on the host path); the per-kernel split is in profiles/ (rocprofv3 --kernel-trace --stats)")
    yield out  # ---- the GPU part is complete; resumed for the CPU baseline once every GPU leg has been measured (main)
    if cpu:
        from oracle import oracle
        for name, bgr in cases.items():
            cm, _ = median_time(lambda: oracle.detect_describe_bgr(bgr, 20, max_kp), 20, warm=2)
            out[name]["cpu_us_1_thread"] = cm * 1e6
    yield out


def ba_leg(ctx, cpu=True):
    """(generator, as detector_leg)  BASELINE.json configs[3]: local BA of 10 key frames x 2000 points (20 000 residuals), 10 LM iterations."""
    from visual_slam_amd.workloads import ba_workload
    w = ba_workload()
    args = (w["poses"], w["pose_fixed"], w["points"], w["point_fixed"], w["obs_pose"], w["obs_point"], w["obs_uv"], w["K"])
    g = ctx.ba_solve(*args)
    med, ts = median_time(lambda: ctx.ba_solve(*args), 30, warm=5)
    n_res = len(w["obs_pose"])
    trials = int(g["trials"])
    out = {"workload": "BASELINE.json configs[3]: 10 cameras x 2000 points, 20000 residuals, Huber, 10 LM iterations",
           "ms_per_solve": med * 1e3, "ms_per_solve_min": min(ts) * 1e3, "repetitions": len(ts), "statistic": "median",
           "lm_trials": trials, "us_per_trial": med * 1e6 / max(trials, 1),
           # a trial of a single-tile window is four dependent launches (schur+cameras, reduce, solve, point_trial)
           "launches_per_trial": 4, "launch_floor_us_per_trial": 4 * LAUNCH_BOUNDARY_US,
           "chi2": [float(g["chi2_initial"]), float(g["chi2_final"])],
           # ~178 B and ~150 flop per residual per linearisation
           "algorithmic_bytes_per_trial": 178 * n_res,
           "hbm_GBps": 178.0 * n_res * trials / med / 1e9,
           "hbm_frac_of_8TBps": 178.0 * n_res * trials / med / 1e9 / HBM_PEAK_GBS,
           "note": "latency-bound (SURVEY 8d): 3.6 MB and 30 MFLOP per LM trial are ~1 us at either roof; the time is "
                   "dependent launches + one host structure pass + upload + read-back"}
    yield out  # ---- GPU part complete (see detector_leg)
    if cpu:
        from oracle import oracle
        c = oracle.ba_solve(*args)
        cmed, _ = median_time(lambda: oracle.ba_solve(*args), 5, warm=1)
        out["cpu_ms_per_solve_1_thread"] = cmed * 1e3
        cores = os.cpu_count() or 1
        sweep = [{"threads": t, "solves_per_s": all_threads_rate(lambda: oracle.ba_solve(*args), t, seconds=3.0)}
                 for t in sorted({min(cores, 32), min(cores, 96), cores})]
        bestr = max(sweep, key=lambda r: r["solves_per_s"])
        out["cpu_all_threads"] = {"threads": bestr["threads"], "host_logical_cpus": cores, "cpu_quota_cores": host_cpu_share()[1],
                                  "solves_per_s": bestr["solves_per_s"], "ms_per_solve_equivalent": 1e3 / bestr["solves_per_s"],
                                  "sweep": sweep,
                                  "note": "the LM solve is sequential; all-threads = independent solves side by side (the C "
                                          "solve releases the interpreter lock); best of the thread counts in `sweep`"}
        out["gpu_solves_per_s"] = 1.0 / med
        out["pose_rel_frobenius_vs_oracle"] = float(max(np.linalg.norm(a - b) / np.linalg.norm(b)
                                                         for a, b in zip(g["poses"], c["poses"])))
    yield out


def ba_scaled_leg(ctx):
    """SURVEY 8d's scaled bundle adjustment (not a BASELINE config): 100 cameras x 200 000 points, 2 000 000 residuals, a
    banded 594 x 594 reduced system, 3 LM iterations -- the BA size at which bytes move.  Observation arrays in pinned memory."""
    from visual_slam_amd.workloads import ba_sliding_window_workload
    w = ba_sliding_window_workload()
    for k in ("obs_pose", "obs_point", "obs_uv"):
        w[k] = ctx.pin(np.ascontiguousarray(w[k]))
    args = (w["poses"], w["pose_fixed"], w["points"], w["point_fixed"], w["obs_pose"], w["obs_point"], w["obs_uv"], w["K"])
    g = ctx.ba_solve(*args, max_iterations=3)
    med, ts = median_time(lambda: ctx.ba_solve(*args, max_iterations=3), 7, warm=1)
    n_res, trials = len(w["obs_pose"]), max(int(g["trials"]), 1)
    return {"workload": "SURVEY 8d: 100 cameras x 200000 points, sliding window of 10, 2000000 residuals, 3 LM iterations",
            "ms_per_solve": med * 1e3, "ms_per_solve_min": min(ts) * 1e3, "repetitions": len(ts), "statistic": "median",
            "lm_trials": trials, "mresiduals_per_s": n_res * trials / med / 1e6,
            "hbm_GBps_at_178B_per_residual": 178.0 * n_res * trials / med / 1e9,
            "chi2": [float(g["chi2_initial"]), float(g["chi2_final"])],
            "structure_built_on_device": bool(ctx.ba_structure_on_device()),
            "note": "whole vs_ba_solve call: DMA of the pinned observation arrays + sparsity structure built on the device "
                    "(csrc/vs_ba_build.hip) + kernels (banded Schur complement on the FP64 matrix cores, banded Cholesky with "
                    "look-ahead in one launch) + read-back; checked against the CPU oracle by tools/ba_scaled.py --check "
                    "(profiles/r04_ba_scaled.log); counter traffic in profiles/r04_ba_scaled_pmc.txt"}


def ba_growth_leg(ctx):
    """The reference's key-frame bundle adjustment is GLOBAL (every key frame, every point: LocalBA.py:143-172), so its reduced camera
    system grows with the sequence: scenes of 1 200 points seen from a random 30 % of N key frames (tools/ba_growth.py), and the three
    problems a real 420-frame run handed to the solver (tests/golden/real_ba_*.npz)."""
    from visual_slam_amd.workloads import ba_workload
    out = {"synthetic_1200_points_30pct_visibility": [], "real_sequence_fixtures": []}
    for n in (10, 22, 23, 52):
        w = ba_workload(n_cams=n, n_points=1200, visibility=0.3, seed=n)
        args = (w["poses"], w["pose_fixed"], w["points"], w["point_fixed"], w["obs_pose"], w["obs_point"], w["obs_uv"], w["K"])
        g = ctx.ba_solve(*args)
        med, _ = median_time(lambda: ctx.ba_solve(*args), 15, warm=3)
        p = ctx.ba_last_path()
        out["synthetic_1200_points_30pct_visibility"].append({
            "key_frames": n, "unknowns": p["unknowns"], "observations": int(len(w["obs_pose"])), "ms_per_solve": med * 1e3,
            "lm_trials": int(g["trials"]), "us_per_trial": med * 1e6 / max(int(g["trials"]), 1), "schur": p["schur"], "dense": p["dense"]})
    golden = os.path.join(ROOT, "tests", "golden")
    for name in ("early", "middle", "last", "guarded"):
        f = np.load(os.path.join(golden, "real_ba_%s.npz" % name))
        args = (f["poses"], f["pose_fixed"], f["points"], f["point_fixed"], f["obs_pose"], f["obs_point"], f["obs_uv"], tuple(f["K"]))
        kw = dict(huber_delta=float(f["huber_delta"]), dcs_phi=float(f["dcs_phi"]),
                  scale_edges=(f["scale_parent"].tolist(), f["scale_child"].tolist(), f["scale_meas"].tolist()))
        g = ctx.ba_solve(*args, **kw)
        med, _ = median_time(lambda: ctx.ba_solve(*args, **kw), 15, warm=3)
        p = ctx.ba_last_path()
        out["real_sequence_fixtures"].append({
            "fixture": "real_ba_%s" % name, "poses": int(len(f["poses"])), "points": int(len(f["points"])), "observations": int(len(f["obs_pose"])),
            "ms_per_solve": med * 1e3, "lm_trials": int(g["trials"]), "schur": p["schur"], "dense": p["dense"]})
    return out


def frames_leg(ctx, cpu=True):
    """(generator, as detector_leg)  frames/s of the 640x480 ICL-NUIM stream (detect+describe -> match -> PnP-RANSAC -> motion-only BA), GPU path and -- as the
    checker/baseline only -- the CPU oracle through the same harness.  GPU legs: medians of 20 repetitions (minima
    beside them); CPU: median of 5 on one core, and one tracker per host core."""
    from visual_slam_amd.harness import HUBER, bench_frames, dataset_dir, load_sequence, track_sequence
    # The class-API legs create a few thousand small Python objects per run (Point, DMatch, row views), which makes the interpreter's
    # cyclic collector run -- and its oldest generation scans EVERYTHING alive in this process, i.e. the other legs' data
    # (tools/driver_pinned_ab.py: 13.0 against 12.5 ms per class-API driver run with two million unrelated objects alive).  What is
    # alive now is moved out of the collector's sight for the GPU legs (gc.freeze, what a long-lived service does after start-up);
    # nothing is skipped inside the timed regions, and the collector stays enabled for what the legs allocate themselves.
    gc.collect()
    gc.freeze()
    try:
        out, poses = bench_frames(ctx, repeats=20)
    finally:
        gc.unfreeze()
    out["gc"] = "objects alive before the frames legs frozen (gc.freeze) for the GPU legs; collector enabled"
    # the full headless driver (main.py's tracking loop + key-frame insertion: triangulation + local BA), class API
    gc.collect()
    gc.freeze()  # (as above; the driver runs before the CPU legs, whose all-threads sweeps leave the host's clocks and caches in another state)
    try:
        from visual_slam_amd import dataset, slam
        from visual_slam_amd.workloads import ICL_NUIM_K
        frames, depth0 = load_sequence(20)
        frames = [ctx.pin(f) for f in frames]  # decoded frames live in pinned memory, as in the tracking-only legs (bench_frames)
        be = slam.Backends(context=ctx)
        res = {}
        med, _ = median_time(lambda: res.__setitem__("r", slam.run_sequence(frames, depth0, ICL_NUIM_K, be, keyframe_gap=4)),
                             20, warm=2)
        r = res["r"]
        _, gt = dataset.read_trajectory(os.path.join(dataset_dir(), "traj3.gt.freiburg.head20"))
        ate = dataset.ate_rmse(r["poses"], gt)
        med_res, _ = median_time(lambda: slam.run_sequence(frames, depth0, ICL_NUIM_K, be, keyframe_gap=4, resident_ctx=ctx),
                                 20, warm=2)
        out["driver"] = {"frames_per_s": len(frames) / med, "resident_frames_per_s": len(frames) / med_res,
                         "statistic": "median of 20", "keyframes": r["keyframes"], "map_points": r["n_points"],
                         "ate_rmse_m": ate["rmse"], "gt_path_length_m": ate["path_length"],
                         "note": "visual_slam_amd/slam.py: main.py:150-348 control flow, key frame every 5th frame, "
                                 "init from depth of frame 0"}
    except Exception as e:
        out["driver"] = {"error": repr(e)}
    finally:
        gc.unfreeze()
    yield out  # ---- GPU part complete (see detector_leg)
    if cpu:
        from oracle import oracle
        oracle.load()

        def detect(bgr):
            xy, _, desc = oracle.detect_describe_bgr(bgr, 20, 3000)
            return xy, desc

        def match(q, t):
            mq, mt, _ = oracle.match_ratio(q, t, 0.8)
            return mq, mt

        def ba(*problem):
            return oracle.ba_solve(*problem, huber_delta=HUBER, max_iterations=10)

        def pnp(obj, img, K4, pose0, seed=0):
            return oracle.pnp_ransac(obj, img, K4, pose0, seed=seed)

        frames, depth0 = load_sequence(20)
        cts, cposes, cstages = [], None, None
        track_sequence(detect, match, ba, frames[:3], depth0, pnp=pnp)
        for _ in range(5):
            t0 = time.perf_counter()
            cposes, cstages, _ = track_sequence(detect, match, ba, frames, depth0, pnp=pnp)
            cts.append(time.perf_counter() - t0)
        cdt = statistics.median(cts)
        out["cpu_frames_per_s"] = len(frames) / cdt
        out["cpu_frames_per_s_max"] = len(frames) / min(cts)
        out["cpu_statistic"] = "median of %d runs" % len(cts)
        out["cpu_stage_ms_per_frame"] = {k: v / len(frames) * 1e3 for k, v in cstages.items()}
        out["cpu_cores_used"] = 1
        cores = os.cpu_count() or 1
        try:
            out["cpu_all_threads"] = cpu_trackers_all_cores(cores)
        except Exception as e:
            out["cpu_all_threads"] = {"error": repr(e)}
        out["pose_rel_frobenius_vs_oracle"] = float(max(np.linalg.norm(a - b) / np.linalg.norm(b)
                                                         for a, b in zip(poses, cposes)))
    yield out


def two_phase(gen):
    """Runs a leg generator up to its first yield (the GPU part) and returns (result dict, finish), finish() running the rest
    (the CPU baseline of the leg, which fills the same dict).  main() measures every GPU leg first and the CPU baselines after them:
    the all-threads sweeps of the baselines leave the host in another state (the Python-heavy driver leg came out 5 - 8 % slower
    behind them), and a GPU figure should not depend on which baseline happened to run in front of it."""
    out = next(gen)

    def finish():
        for _ in gen:
            pass
        return out
    return out, finish


def frames_replicas(ctx, dist, world, dev, track=None, sequence=None, sync=None, reps=7):
    """frames/s with one independent replica of the tracker per GPU (north_star: detection and BA stay single-GPU, so
    N GPUs track N streams): every rank runs the device-resident tracking period on the 20 fixture frames between two
    barriers; the aggregate is ranks x frames / slowest rank (the MAX over ranks of every repetition, median over the
    repetitions), identical on every rank.  No data-path collective: the only communication is the barrier and the
    reduction of the timings.  track / sequence / sync: injected by tests/test_bench_launcher.py (a stub tracker on gloo)."""
    import torch
    if track is None:
        from visual_slam_amd.harness import load_sequence, track_sequence_resident
        frames, depth0 = load_sequence(20)
        frames = [ctx.pin(f) for f in frames]

        def track(fr):
            track_sequence_resident(ctx, fr, depth0, pipelined=True)
        sync = torch.cuda.synchronize
    else:
        frames = sequence
    track(frames[:4])
    times, own = [], []
    for _ in range(reps):
        if sync:
            sync()
        dist.barrier()
        t0 = time.perf_counter()
        track(frames)
        if sync:
            sync()
        dt = time.perf_counter() - t0
        own.append(dt)
        te = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        times.append(float(te.item()))
    med = statistics.median(times)
    per_rank = [None] * world
    dist.all_gather_object(per_rank, len(frames) / statistics.median(own))
    return {"replicas": world, "frames_per_s": world * len(frames) / med, "seconds_slowest_rank_median": med,
            "frames_per_s_per_rank": per_rank, "repetitions": reps,
            "note": "one tracker replica per GPU on the same 20 frames (device-resident tracking period, pipelined); "
                    "replicas only -- no data-path collective"}


def timed_steps(step, drain, fence, steps, warmup, torch, dist, use_dist, dev, collect=True, event_stream=None, events_out=None):
    """The contract's timing: W untimed steps, then EXACTLY K steps bracketed by barrier + synchronize, MAX over ranks.
    collect=False: the caller has already collected and disabled the interpreter's garbage collector (a collection takes
    tens of ms with torch loaded, during which the GPU would go idle and drop its clocks).
    event_stream / events_out: a pair of HIP events is recorded on that stream (the one the step's kernel is launched on) INSIDE the
    wall-clock bracket -- behind t0 and in front of the first launch, behind the last launch and in front of the closing
    synchronisation -- and events_out["ms"] receives their distance / K: the kernel's average launch duration over THE timed region
    itself, which cannot exceed ms_per_step (one event record of ~2 us of host time per region, not per step)."""
    import gc
    if collect:
        gc.collect()  # before the warm-up, not after it
    gc.disable()  # a generation-2 collection of the interpreter must not land in the K steps either
    for _ in range(warmup):
        step()
    drain()
    fence()
    ev = None
    if event_stream is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    t0 = time.perf_counter()
    if ev is not None:
        ev[0].record(event_stream)
    out = None
    for _ in range(steps):
        out = step()
    if ev is not None:
        ev[1].record(event_stream)
    out = drain() or out
    fence()
    elapsed = time.perf_counter() - t0
    if ev is not None and events_out is not None:
        events_out["ms"] = ev[0].elapsed_time(ev[1]) / steps
    if collect:
        gc.enable()
    if use_dist:
        te = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    return elapsed, out


def main():
    if len(sys.argv) >= 5 and sys.argv[1] == "--cpu-tracker-worker":  # CPU-only helper process of the frames leg
        cpu_tracker_worker(int(sys.argv[2]), float(sys.argv[3]), float(sys.argv[4]))
        return
    args = parse()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:   # started by hand: become the launcher (before any GPU call)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], timeout=args.launch_timeout if args.launch_timeout > 0 else None))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:   # a curve recorded with another rank count than the command line names would be meaningless
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: start it as `python -m torch.distributed.run --nproc-per-node "
                         "%d bench.py --gpus %d ...` (or by hand without WORLD_SIZE, and it launches its own ranks)\n"
                         % (args.gpus, world, args.gpus, args.gpus))
        sys.exit(2)
    # ONE JSON line on standard output, nothing else: libraries that print there on their own (RCCL's version banner at
    # communicator creation) are sent to standard error; the line itself goes through a private copy of the descriptor
    json_out = os.fdopen(os.dup(1), "w")
    sys.stdout.flush()
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the tracking hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_collective
    init_pg_s = None
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        t_pg = time.perf_counter()
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        init_pg_s = time.perf_counter() - t_pg

    from visual_slam_amd import Context, _capi
    from visual_slam_amd.sharded import ShardedMatcher, shard_bounds
    from visual_slam_amd.workloads import match_workload
    import visual_slam_amd.context as vctx
    ctx = Context(local_rank)
    vctx._DEFAULT = ctx
    lib = _capi.load()
    if args.target_blocks:
        ctx.tune_match(target_blocks=args.target_blocks)

    nq, nt = args.nq, args.nt
    q_np, t_np = match_workload(nq, nt)
    if world > 1:  # weak scaling: every rank owns a different 10k-query shard, same train set
        q_np, _ = match_workload(nq, nt, seed=100 + rank)
        _, t_np = match_workload(nq, nt)
    matcher = ShardedMatcher(force_collective=args.force_collective)
    stream = matcher.torch_stream()  # the library's stream, shared with torch copies and events
    torch.cuda.set_stream(stream)
    q = torch.from_numpy(q_np).to(dev)
    t = torch.from_numpy(t_np).to(dev)

    pending = []

    def make_step(qq, tt, n_total, single=False):
        """One pass of the match over this rank's batch: ONE C call (ShardedMatcher.plan).  With a collective, the
        all-gather of step k is started asynchronously and collected after the kernels of step k+1 are enqueued (two
        rotating buffer sets), so the exchange overlaps the next step's compute; drain() collects the last one inside
        the timed region."""
        # the bench's inputs are resident long before the first step and results are only read after collect(): the steps'
        # kernels need no ordering behind the library's stream (static_inputs); with a collective, four buffer sets on the two
        # compute streams keep the in-place all-gather of a set off the chain of launches (sharded.py, _Plan)
        one = args.single_stream or single
        # one launch at a time: with a collective the kernels run back to back on one auxiliary stream of the context and the
        # in-place all-gathers on another (sharded.py, _Plan)
        plan = matcher.plan(qq, tt, n_total, single_stream=one, in_flight=max(2, args.in_flight),
                            buffers=args.buffers or (4 if use_dist else None), static_inputs=True)
        plans.append(plan)

        def step():
            slot = plan.submit()
            out = plan.collect(pending.pop()) if pending else None
            pending.append(slot)
            return out
        step.plan = plan
        return step

    plans = []

    def drain():
        return plans[-1].collect(pending.pop()) if pending else None

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- BASELINE.json configs[4]: 100 000 x 100 000, query-split over the ranks (strong scaling).  Runs first: its
    # ~0.2 s of solid matching also brings the GPU to its sustained clocks before the short headline region (20 x 60 us).
    # Everything the headline needs is prepared before it (plan, buffers, one garbage collection), so that the W warm-up
    # steps of the headline follow the last cfg5 step without an idle gap.
    import gc
    duo_step = None if args.single_stream else make_step(q, t, nq * world)  # secondary figure: launches overlapped on two streams
    head_step = make_step(q, t, nq * world, single=True)  # THE step of `value`: one launch at a time on the library's stream
    gc.collect()
    gc.disable()
    cfg5 = None
    if not args.no_cfg5:
        try:
            Q5 = T5 = 100000
            b5, e5, per5 = shard_bounds(Q5, world, rank)
            q5_np, t5_np = match_workload(Q5, T5)
            q5 = torch.from_numpy(q5_np[b5:e5]).to(dev)
            t5 = torch.from_numpy(t5_np).to(dev)
            del q5_np, t5_np
            k5, w5 = max(3, min(args.steps, 20)), max(1, min(args.warmup, 3))
            el5, _ = timed_steps(make_step(q5, t5, Q5, single=True), drain, fence, k5, w5, torch, dist, use_dist, dev, collect=False)

            def local_only():
                return matcher.knn2_local_shard(q5, t5)
            el5c, _ = timed_steps(local_only, lambda: None, fence, k5, w5, torch, dist, use_dist, dev, collect=False)
            ms5, ms5c = el5 / k5 * 1e3, el5c / k5 * 1e3
            cfg5 = {"workload": "BASELINE.json configs[4]: 100000 x 100000 x 256-bit, k=2, query-split over %d rank(s): "
                                "rank r owns ceil(Q/N) queries, train replicated, one RCCL all-gather of 16 B/query" % world,
                    "scaling": "strong", "n_gpus": world, "steps": k5, "queries_per_rank": per5,
                    "ms_per_step": ms5, "gmatches_per_s": float(Q5) * T5 / (ms5 * 1e-3) / 1e9,
                    "ms_per_step_compute_only": ms5c, "collective_overhead_ms": ms5 - ms5c,
                    "valu_frac_compute_only": OPS_PER_MATCH * float(per5) * T5 / (ms5c * 1e-3) / VALU_PEAK_LANE_OPS,
                    "valu_frac_compute_only_uniform_4cycle": OPS_PER_MATCH * float(per5) * T5 / (ms5c * 1e-3) / VALU_PEAK_UNIFORM_4CYCLE,
                    "note": "collective_overhead_ms = step with the all-gather (overlapped with the next step's kernels) "
                            "minus the same step without it; max over ranks"}
        except Exception as e:  # all ranks take the same path: the collectives inside stay matched
            cfg5 = {"error": repr(e)}

    # secondary figure first (it also keeps the clocks up): the same K steps with two launches overlapped on two streams
    elapsed2 = None
    if duo_step is not None:
        plans.append(duo_step.plan)  # drain() collects from the newest plan
        elapsed2, _ = timed_steps(duo_step, drain, fence, args.steps, args.warmup, torch, dist, use_dist, dev, collect=False)
    # THE timed region of the contract: W warm-up steps, then exactly K steps, one launch at a time -- the mode of `value`,
    # `ms_per_step` and `roofline.frac` alike
    plans.append(head_step.plan)
    head_events = {}
    elapsed, _ = timed_steps(head_step, drain, fence, args.steps, args.warmup, torch, dist, use_dist, dev, collect=False,
                             event_stream=head_step.plan.streams[0], events_out=head_events)
    gc.enable()
    q5 = t5 = None
    ms_per_step = elapsed / args.steps * 1e3
    total_matches = float(nq) * nt * world
    value = total_matches / (ms_per_step * 1e-3) / 1e9
    ms_per_step_2 = None if elapsed2 is None else elapsed2 / args.steps * 1e3
    value_2 = None if elapsed2 is None else total_matches / (ms_per_step_2 * 1e-3) / 1e9

    # ---- dominant kernel (hamming_knn2_kernel): HIP events recorded by the library on the launch stream around each
    # launch of K more steps, in this process, right after the timed region
    roof = None
    if rank == 0:
        import ctypes as C
        lib.vs_match_profile(ctx.handle, 1)
        for _ in range(args.steps):
            matcher.knn2_local_shard(q, t)
        torch.cuda.synchronize()
        km = C.c_float(0)
        ncalls = lib.vs_match_profile_read(ctx.handle, C.byref(km))
        lib.vs_match_profile(ctx.handle, 0)
        # the same launches timed as ONE region (a single event pair around K back-to-back launches): the per-launch
        # pairs above put an event between any two kernels, which keeps them ~7 us apart
        solo = matcher.plan(q, t, nq * world, single_stream=True)  # one launch at a time: a launch's own duration
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(args.steps):
            solo.submit()
        e1.record(stream)
        torch.cuda.synchronize()
        region_ms = e0.elapsed_time(e1) / args.steps
        # and once more with the GPU already busy when the region starts (no idle start)
        for _ in range(5):
            solo.submit()
        e0.record(stream)
        for _ in range(args.steps):
            solo.submit()
        e1.record(stream)
        torch.cuda.synchronize()
        busy_ms = e0.elapsed_time(e1) / args.steps
        # the shader clock DURING the kernel: in-kernel stamps of one launch (vs_match_stamps: every workgroup's thread 0 reads the
        # shader-clock counter and the 100 MHz wall clock at its start and at the end of its scan)
        clock_ghz = None
        try:
            for _ in range(max(args.steps, 20)):   # the clock under SUSTAINED load: a burst, its last launch stamped
                matcher.knn2_local_shard(q, t)
            lib.vs_match_stamps(ctx.handle, 1)
            matcher.knn2_local_shard(q, t)
            lib.vs_match_stamps(ctx.handle, 0)
            torch.cuda.synchronize()
            st = np.zeros((8192, 8))
            rows = lib.vs_match_stamps_read(ctx.handle, st.ctypes.data, len(st))
            if rows > 0:
                st = st[:rows]
                cyc = -np.where(st[:, 4] > 0, st[:, 5], st[:, 4])
                dt_us = np.maximum(st[:, 2], 0) - np.maximum(st[:, 0], 0)
                ok = (dt_us > 5) & (cyc > 0)
                if ok.any():
                    clock_ghz = float(np.median(cyc[ok] / dt_us[ok])) / 1e3
        except Exception:
            clock_ghz = None
        kernel_ms_isolated = float(km.value)
        # the dominant kernel's average launch duration over THE timed region (events inside its wall-clock bracket); the
        # separate burst (region_ms) and the per-launch pairs are kept beside it
        in_region_ms = head_events.get("ms")
        kernel_ms = (in_region_ms if in_region_ms else region_ms) if not use_dist else kernel_ms_isolated
        pairs = float(nq) * nt
        lane_ops = OPS_PER_MATCH * pairs
        achieved = lane_ops / (kernel_ms * 1e-3)
        alg_bytes = BYTES_PER_MATCH * pairs
        traffic = None
        pmc_src = None
        for name in ("r05_pmc_match.json", "r04_pmc_match.json", "r03_pmc_match.json", "r02_pmc_match.json", "r01_pmc_match.json"):
            try:  # HBM bytes per launch from the committed PMC passes (profiles/), only for the workload they were taken on
                pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
                if (nq, nt) == (10000, 10000):
                    key = "hamming_knn2_kernel_per_launch" if "hamming_knn2_kernel_per_launch" in pmc else "hamming_partial_kernel_per_launch"
                    traffic = pmc[key]["hbm_bytes_corrected_upper"]
                    pmc_src = "profiles/" + name
                break
            except Exception:
                continue
        frac_of = lambda ms, peak: (lane_ops / (ms * 1e-3)) / peak  # noqa: E731
        roof = {"bound": "valu", "achieved": achieved / 1e12, "peak": VALU_PEAK_LANE_OPS / 1e12, "unit": "Tlane-op/s",
                "frac": achieved / VALU_PEAK_LANE_OPS, "traffic": traffic,
                "kernel": "hamming_knn2_kernel", "kernel_ms": kernel_ms, "profiled_calls": int(ncalls),
                "measured_in": "single_stream: one launch at a time on one stream -- the mode of `value` and `ms_per_step` "
                               "(kernel_ms <= ms_per_step; the difference is the host's share of a step)",
                "kernel_ms_source": "HIP events on the launch stream over THE timed region of `value`: one pair recorded inside the "
                                    "wall-clock bracket around its K launches (one at a time on ONE stream), divided by K; includes "
                                    "the ~1.5 us kernel boundary; `kernel_ms_separate_burst` is the same measurement on K more launches "
                                    "after the region",
                "kernel_ms_separate_burst": region_ms,
                "steps_in_flight_of_value": 1,
                "kernel_ms_event_pair_per_launch": kernel_ms_isolated, "kernel_ms_busy_start": busy_ms,
                "lane_ops_per_match": OPS_PER_MATCH,
                "lane_ops_model": "ISA of the main loop: per 8 distances 64 v_xor_b32 + 64 v_bcnt_u32_b32 + 8 v_lshl_or_b32 + "
                                  "4 v_med3_u32 + 4 v_min3_u32 + 4 v_min_u32; SQ_INSTS_VALU (profiles/) = this x 1e8 / 64 + 5 %",
                "peak_model": "mix-specific integer-VALU ceiling: the 64 VGPR-only v_xor of 8 distances issue at one wave64 op per "
                              "2 cycles per SIMD, the other 84 instructions (v_bcnt / v_lshl_or / v_med3 / v_min3 / v_min) at one "
                              "per 4 cycles: 148 x 64 lanes / 464 cycles x 1024 SIMDs x 2.4 GHz (tools/valu_probe2/3.hip, "
                              "profiles/r02_valu_probe*.log)",
                "valu_ceiling_gmatches": VALU_PEAK_LANE_OPS / OPS_PER_MATCH / 1e9,
                "frac_whole_step": frac_of(ms_per_step, VALU_PEAK_LANE_OPS) if world == 1 else None,
                "frac_two_in_flight": frac_of(ms_per_step_2, VALU_PEAK_LANE_OPS) if world == 1 and ms_per_step_2 else None,
                "against_guide_nominal": {
                    "peak": VALU_PEAK_GUIDE_NOMINAL / 1e12, "frac": achieved / VALU_PEAK_GUIDE_NOMINAL,
                    "note": "the guide's nominal VALU issue rate for every instruction alike: 256 CUs x 4 SIMD-32 x 32 lanes x "
                            "2.4 GHz = one wave64 op per 2 cycles per SIMD (MI355X_MICROARCH.md); `frac` above prices v_bcnt / "
                            "v_lshl_or / v_min3 / v_med3 at the one per 4 cycles they were measured to issue at"},
                "against_uniform_4cycle_peak": {
                    "peak": VALU_PEAK_UNIFORM_4CYCLE / 1e12, "frac": achieved / VALU_PEAK_UNIFORM_4CYCLE,
                    "note": "round 2's yard-stick: every op priced at one wave64 op per 4 cycles per SIMD (understates the "
                            "roof by 1.28x for this mix: the xor operands are staged into VGPRs precisely to issue faster)"},
                "shader_clock_ghz_measured": clock_ghz,
                "against_measured_clock": None if not clock_ghz else {
                    "peak": VALU_PEAK_LANE_OPS * clock_ghz / 2.4 / 1e12, "frac": achieved / (VALU_PEAK_LANE_OPS * clock_ghz / 2.4),
                    "note": "the same mix-specific ceiling at the shader clock measured inside this kernel (in-kernel cycle counter "
                            "against the 100 MHz wall clock, median over the workgroups of the last launch of a back-to-back burst) "
                            "instead of the nominal 2.4 GHz"},
                "against_measured_mix_rate": {
                    "peak": VALU_PROBE_MIX_LANE_OPS / 1e12, "frac": achieved / VALU_PROBE_MIX_LANE_OPS,
                    "note": "what a dependent xor -> bcnt stream was measured to issue at on this chip with 5 waves per SIMD "
                            "(tools/valu_probe3.hip chain_grp_vv: 1.597 ns per wave64 instruction per SIMD)"},
                "hbm_measured": None if traffic is None else {
                    "bytes_per_launch": traffic, "GBps": traffic / (kernel_ms * 1e-3) / 1e9,
                    "frac_of_8TBps": traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "compulsory_bytes": 32 * (nq + nt) + 16 * nq, "source": pmc_src,
                    "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH_SIZE doubled per the gfx950 "
                            "correction (upper bound); the excess over compulsory is the per-chunk partial words"},
                "hbm_model_secondary": {
                    "bound": "hbm", "model": "SURVEY.md 8d streamed-operand: 32 B per distance evaluation",
                    "algorithmic_bytes_per_launch": alg_bytes, "achieved_GBps": alg_bytes / (kernel_ms * 1e-3) / 1e9,
                    "peak_GBps": HBM_PEAK_GBS, "frac": alg_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "note": "> 1 by construction: train rows are reused from LDS, queries from VGPRs, so this model cannot "
                            "bound the kernel; kept because north_star words its target (>= 0.6) in it"},
                "kernels": {}}

    # ---- what ran the exchange: ranks of the matcher's own RCCL communicator (ncclCommCount) and every rank's device
    rccl_ranks, rccl_device, devices = None, None, [torch.cuda.current_device()]
    if use_dist:
        rccl_ranks = matcher.rccl_ranks()
        rccl_device = matcher._rccl.device() if matcher._rccl is not None else None
        dv = torch.tensor([torch.cuda.current_device()], dtype=torch.int32, device=dev)
        gl = [torch.zeros_like(dv) for _ in range(world)]
        dist.all_gather(gl, dv)
        devices = [int(x.item()) for x in gl]
        bad = (rccl_ranks is not None and rccl_ranks != world) or (world > 1 and len(set(devices)) != world)
        flag = torch.tensor([1 if bad else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()):
            sys.stderr.write("bench.py: rank %d: rccl_ranks=%r world=%d devices=%r -- ranks do not map one to one onto GPUs\n"
                             % (rank, rccl_ranks, world, devices))
            sys.exit(3)

    # per rank: which exchange it uses and why, seconds spent creating the direct communicator, seconds in init_process_group
    rank_diag = [dict(matcher.diagnostics(), init_process_group_s=init_pg_s, device=torch.cuda.current_device())]
    if use_dist and world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, rank_diag[0])
        rank_diag = gathered
    replicas = None
    if use_dist and not args.no_frames:
        try:
            replicas = frames_replicas(ctx, dist, world, dev)
        except Exception as e:
            replicas = {"error": repr(e)}
    host_abi = None
    if rank == 0 and world == 1:
        # the same workload through the HOST entry point (vs_hamming_knn2: descriptors arrive in pageable host memory,
        # results return to the host) with fresh contents every call, so both sets are uploaded: the PCIe-inclusive
        # rate.  Reported beside `value`, never as `value`.
        try:
            qh, th = q_np.copy(), t_np.copy()
            ctx.hamming_knn2(qh, th)
            state = {"i": 0}

            def call():
                state["i"] += 1
                qh[0, 0] ^= 1 + (state["i"] & 1)   # new content at the same address: both sets are uploaded again
                th[0, 0] ^= 1 + (state["i"] & 1)
                ctx.hamming_knn2(qh, th)
            dth, _ = median_time(call, 30, warm=3)
            host_abi = {"gmatches_per_s": float(nq) * nt / dth / 1e9, "ms_per_call": dth * 1e3, "statistic": "median of 30",
                        "note": "vs_hamming_knn2 on pageable host arrays, fresh contents per call: exact compare against the "
                                "resident copy + H2D 2 x 320 KB + kernel + D2H 160 KB + synchronisation"}
        except Exception as e:
            host_abi = {"error": repr(e)}
    if rank == 0:
        line = {
            "metric": "10k x 10k 256-bit Hamming 2-NN brute-force match throughput", "value": value,
            "unit": "Gmatches/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "value_mode": "one launch at a time on one stream" + ("" if not use_dist else
                          "; the all-gather of step k on a second stream beside the kernel of step k + 1"),
            "two_in_flight": None if value_2 is None else {
                "value": value_2, "ms_per_step": ms_per_step_2, "steps_in_flight": max(2, args.in_flight),
                "note": "the same K steps under the same contract with launches overlapped on as many streams (the tail of one "
                        "launch -- fold, kernel boundary -- runs beside the body of the next): throughput of a caller that keeps "
                        "two matches in flight; secondary, not the mode of value / ms_per_step / roofline"},
            "dtype": "u32 (xor + popcount on 256-bit descriptors)", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[2]: %d x %d x 256-bit descriptors, k=2, per GPU%s"
                                   % (nq, nt, "" if world == 1 else "; %d query shards + RCCL all-gather (16 B/query, overlapped with the next step)" % world),
                       "queries_per_gpu": nq, "train": nt, "parallelism": "query-shard x%d" % world},
            "roofline": roof,
            "rccl_ranks": rccl_ranks, "rank_devices": devices, "rccl_device_rank0": rccl_device,
            "collective_path": matcher.collective_path(), "collective_ranks": rank_diag, "launched_by": "bench.py launcher" if os.environ.get(
                "VS_BENCH_LAUNCHED") else "external launcher" if "WORLD_SIZE" in os.environ else "single process",
        }
        if cfg5 is not None:
            line["cfg5"] = cfg5
        if host_abi is not None:
            line["host_abi"] = host_abi
        cpu = world == 1 and not args.no_cpu_baseline
        finishers = []  # the CPU halves of the legs, run after every GPU leg (two_phase)
        if world == 1 and not args.no_frames:
            try:
                line["cfg2"], fin = two_phase(detector_leg(ctx, torch, stream, cpu=cpu))
                finishers.append(("cfg2", fin))
                k2 = line["cfg2"]["synthetic_rng2_blur5"]
                roof["kernels"]["detect_band_kernel+select_describe_kernel"] = {
                    "bound": "latency", "us_per_frame": k2["kernel_only_us"], "algorithmic_bytes": k2["algorithmic_bytes"],
                    "hbm_frac_of_8TBps": k2["hbm_frac_of_8TBps"], "launch_floor_us": k2["launch_floor_us"]}
            except Exception as e:
                line["cfg2"] = {"error": repr(e)}
            try:
                line["local_ba"], fin = two_phase(ba_leg(ctx, cpu=cpu))
                finishers.append(("local_ba", fin))
                lb = line["local_ba"]
                roof["kernels"]["local_ba_cfg4 (ba_* kernels of one solve)"] = {
                    "bound": "latency", "us_per_lm_trial": lb["us_per_trial"],
                    "algorithmic_bytes_per_trial": lb["algorithmic_bytes_per_trial"],
                    "hbm_frac_of_8TBps": lb["hbm_frac_of_8TBps"],
                    "launch_floor_us_per_trial": lb["launch_floor_us_per_trial"]}
            except Exception as e:
                line["local_ba"] = {"error": repr(e)}
            try:
                line["local_ba_scaled"] = ba_scaled_leg(ctx)
            except Exception as e:
                line["local_ba_scaled"] = {"error": repr(e)}
            try:
                line["local_ba_growth"] = ba_growth_leg(ctx)
            except Exception as e:
                line["local_ba_growth"] = {"error": repr(e)}
        if not args.no_frames:
            try:
                line["frames"], fin = two_phase(frames_leg(ctx, cpu=cpu))
                finishers.append(("frames", fin))
                if replicas is not None:
                    line["frames"]["replicas"] = replicas
                if world > 1:
                    line["frames"]["parallelism"] = ("replica: detection, the 600 x 600 per-frame match and BA run on "
                                                     "rank 0's GPU only (north_star: detection and BA stay single-GPU)")
            except Exception as e:
                line["frames"] = {"error": repr(e)}
        # ---- CPU baselines, after every GPU leg
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(nq, nt)
            except Exception as e:  # never lose the GPU numbers to a host-side problem
                line["cpu_baseline"] = {"error": repr(e)}
        for key, fin in finishers:
            try:
                fin()
            except Exception as e:
                if isinstance(line.get(key), dict):
                    line[key]["cpu_error"] = repr(e)
        line["leg_order"] = "every GPU leg first, then the CPU baselines (cpu_baseline, cfg2 / local_ba / frames CPU halves)"
        json_out.write(json.dumps(line) + "\n")
        json_out.flush()
    if use_dist:
        dist.barrier()  # the other ranks wait here while rank 0 runs the (replica) frames leg and prints
        matcher.close()
    ctx.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
