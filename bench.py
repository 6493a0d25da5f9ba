#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X tracking hot path (contract: see the task description / DESIGN.md).

A "step" is one pass of the brute-force Hamming 2-NN match (the reference's knnMatch, src/v2/frame.py:23) over one
batch of synthetic descriptors already resident in HBM:
  N = 1 : BASELINE.json configs[2] -- 10 000 x 10 000 x 256-bit, k = 2 (the configuration the metric is quoted on)
  N > 1 : weak scaling -- every rank matches its own 10 000-query shard against the replicated 10 000-row train set,
          then one RCCL all-gather of the per-shard best matches (16 B per query) assembles the N*10 000 results.
value = distance evaluations (Q*T) of all ranks / wall time of the K timed steps, in Gmatches/s.

Rank 0 prints ONE JSON line.  Extra objects on the line: roofline (dominant kernel, HIP-event timed in this run),
cpu_baseline (the CPU oracle timed on this box's host cores), frames (the 640x480 detect+describe -> match -> motion
BA stream, frames/s), config.
"""
import argparse
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_PER_MATCH = 32           # SURVEY.md 8d streamed-operand model: one 32-byte train descriptor per evaluation
VALU_LANE_OPS = 256 * 4 * 16 * 2.4e9  # 3.93e13: integer VALU issues a wave64 op every 4 cycles per SIMD (measured
                                      # 36-38e12 with tools/valu_probe.hip; only f32 FMA-class ops run at twice that)
OPS_PER_MATCH = 19             # 8 v_xor + 8 v_bcnt + v_lshl_or + v_med3 + v_min
VALU_LANE_OPS_MEASURED = 37.0e12  # tools/valu_probe.hip on this GPU (profiles/r01_valu_probe.log): 4.2-4.4 cycles per op


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--nq", type=int, default=10000, help="queries per rank")
    ap.add_argument("--nt", type=int, default=10000, help="train descriptors (replicated)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-frames", action="store_true")
    ap.add_argument("--target-blocks", type=int, default=0, help="tuning: workgroups per launch of the match kernel")
    ap.add_argument("--force-collective", action="store_true",
                    help="rehearsal: run the RCCL all-gather even at world size 1 (exercises the N>1 code path)")
    return ap.parse_args()


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
    cores = os.cpu_count() or 1
    oracle.hamming_knn2(q[:512], t, threads=0, lib=lib)  # warm
    times = []
    t_end = time.time() + 12.0
    while len(times) < 5 or (time.time() < t_end and len(times) < 40):
        t0 = time.perf_counter()
        oracle.hamming_knn2(q, t, threads=0, lib=lib)
        times.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    oracle.hamming_knn2(q, t, threads=1, lib=lib)
    t1 = time.perf_counter() - t0
    med = statistics.median(times)
    return {"value": nq * nt / med / 1e9, "unit": "Gmatches/s", "cores": cores, "kind": "port",
            "sample": "full %dx%d cfg3 workload, CPU oracle (C, -O3 -march=native, OpenMP %d threads), median of %d runs"
                      % (nq, nt, cores, len(times)),
            "single_thread_value": nq * nt / t1 / 1e9}


def frames_leg(ctx, cpu=True):
    """frames/s of the 640x480 ICL-NUIM stream (detect+describe -> match -> PnP-RANSAC -> motion-only BA), GPU path and -- as the
    checker/baseline only -- the CPU oracle through the same harness."""
    from visual_slam_amd.harness import HUBER, bench_frames, load_sequence, track_sequence
    out, poses = bench_frames(ctx)
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

        frames, depth0 = load_sequence(20)
        t0 = time.perf_counter()
        def pnp(obj, img, K4, pose0, seed=0):
            return oracle.pnp_ransac(obj, img, K4, pose0, seed=seed)

        cposes, cstages, _ = track_sequence(detect, match, ba, frames, depth0, pnp=pnp)
        cdt = time.perf_counter() - t0
        out["cpu_frames_per_s"] = len(frames) / cdt
        out["cpu_stage_ms_per_frame"] = {k: v / len(frames) * 1e3 for k, v in cstages.items()}
        out["cpu_cores_used"] = 1
        out["pose_rel_frobenius_vs_oracle"] = float(max(np.linalg.norm(a - b) / np.linalg.norm(b)
                                                         for a, b in zip(poses, cposes)))
    # the full headless driver (main.py's tracking loop + key-frame insertion: triangulation + local BA), class API
    try:
        from visual_slam_amd import slam
        from visual_slam_amd.workloads import ICL_NUIM_K
        frames, depth0 = load_sequence(20)
        be = slam.Backends(context=ctx)
        slam.run_sequence(frames[:7], depth0, ICL_NUIM_K, be, keyframe_gap=4)
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            r = slam.run_sequence(frames, depth0, ICL_NUIM_K, be, keyframe_gap=4)
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        from visual_slam_amd import dataset
        from visual_slam_amd.harness import ICL_DIR
        _, gt = dataset.read_trajectory(os.path.join(ICL_DIR, "traj3.gt.freiburg.head20"))
        ate = dataset.ate_rmse(r["poses"], gt)
        slam.run_sequence(frames[:7], depth0, ICL_NUIM_K, be, keyframe_gap=4, resident_ctx=ctx)
        best_res = None
        for _ in range(3):
            t0 = time.perf_counter()
            slam.run_sequence(frames, depth0, ICL_NUIM_K, be, keyframe_gap=4, resident_ctx=ctx)
            dt = time.perf_counter() - t0
            best_res = dt if best_res is None or dt < best_res else best_res
        out["driver"] = {"frames_per_s": len(frames) / best, "resident_frames_per_s": len(frames) / best_res,
                         "keyframes": r["keyframes"], "map_points": r["n_points"],
                         "ate_rmse_m": ate["rmse"], "gt_path_length_m": ate["path_length"],
                         "note": "visual_slam_amd/slam.py: main.py:150-348 control flow, key frame every 5th frame, "
                                 "init from depth of frame 0"}
    except Exception as e:
        out["driver"] = {"error": repr(e)}
    return out


def ba_leg(ctx, cpu=True):
    """BASELINE.json configs[3]: local BA of 10 key frames x 2000 points (20 000 residuals), 10 LM iterations."""
    from visual_slam_amd.workloads import ba_workload
    w = ba_workload()
    args = (w["poses"], w["pose_fixed"], w["points"], w["point_fixed"], w["obs_pose"], w["obs_point"], w["obs_uv"], w["K"])
    for _ in range(5):
        g = ctx.ba_solve(*args)
    t0 = time.perf_counter()
    for _ in range(10):
        g = ctx.ba_solve(*args)
    dt = (time.perf_counter() - t0) / 10
    out = {"workload": "BASELINE.json configs[3]: 10 cameras x 2000 points, 20000 residuals, Huber, 10 LM iterations",
           "ms_per_solve": dt * 1e3, "lm_trials": int(g["trials"]), "chi2": [float(g["chi2_initial"]), float(g["chi2_final"])]}
    if cpu:
        from oracle import oracle
        t0 = time.perf_counter()
        c = oracle.ba_solve(*args)
        out["cpu_ms_per_solve"] = (time.perf_counter() - t0) * 1e3
        out["cpu_cores_used"] = 1
        out["pose_rel_frobenius_vs_oracle"] = float(max(np.linalg.norm(a - b) / np.linalg.norm(b)
                                                         for a, b in zip(g["poses"], c["poses"])))
    return out


def frames_replicas(ctx, dist, world, dev):
    """frames/s with one independent replica of the tracker per GPU (north_star: detection and BA stay single-GPU, so
    N GPUs track N streams): every rank runs the device-resident tracking period on the 20 fixture frames between two
    barriers; the aggregate is ranks x frames / slowest rank."""
    import torch
    from visual_slam_amd.harness import load_sequence, track_sequence_resident
    frames, depth0 = load_sequence(20)
    frames = [ctx.pin(f) for f in frames]
    track_sequence_resident(ctx, frames[:4], depth0, pipelined=True)
    best = None
    for _ in range(3):
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        track_sequence_resident(ctx, frames, depth0, pipelined=True)
        dt = time.perf_counter() - t0
        te = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        dt = float(te.item())
        best = dt if best is None or dt < best else best
    return {"replicas": world, "frames_per_s": world * len(frames) / best, "seconds_slowest_rank": best,
            "note": "one tracker replica per GPU on the same 20 frames (device-resident tracking period, pipelined)"}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the tracking hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_collective
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from visual_slam_amd import Context, _capi
    from visual_slam_amd.sharded import ShardedMatcher
    from visual_slam_amd.workloads import match_workload
    import visual_slam_amd.context as vctx
    ctx = Context(local_rank)
    vctx._DEFAULT = ctx
    if args.target_blocks:
        _capi.load().vs_match_set_target_blocks(args.target_blocks)

    nq, nt = args.nq, args.nt
    q_np, t_np = match_workload(nq, nt)
    if world > 1:  # weak scaling: every rank owns a different 10k-query shard, same train set
        q_np, _ = match_workload(nq, nt, seed=100 + rank)
        _, t_np = match_workload(nq, nt)
    matcher = ShardedMatcher()
    stream = matcher.torch_stream()  # the library's stream, shared with torch copies, events and RCCL
    torch.cuda.set_stream(stream)
    q = torch.from_numpy(q_np).to(dev)
    t = torch.from_numpy(t_np).to(dev)

    pending = []

    def step():
        """One pass of the match over this rank's batch.  With a collective, the all-gather of step k is started
        asynchronously and collected after the kernels of step k+1 are enqueued (two rotating buffer sets), so the
        exchange overlaps the next step's compute; drain() collects the last one inside the timed region."""
        if not use_dist:
            return matcher.knn2_local_shard(q, t)
        ticket = matcher.submit(q, t, nq * world)
        out = matcher.collect(pending.pop()) if pending else None
        pending.append(ticket)
        return out

    def drain():
        return matcher.collect(pending.pop()) if pending else None

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    import gc
    for _ in range(args.warmup):
        step()
    drain()
    gc.collect()
    gc.disable()  # a generation-2 collection of the interpreter (tens of ms with torch loaded) must not land in K steps
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    out = drain() or out
    fence()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if use_dist:
        te = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    ms_per_step = elapsed / args.steps * 1e3
    total_matches = float(nq) * nt * world
    value = total_matches / (ms_per_step * 1e-3) / 1e9

    # ---- dominant kernel (hamming_partial_kernel): HIP events recorded by the library on the launch stream around
    # each kernel of the same K steps, in this process, right after the timed region
    roof = None
    if rank == 0:
        import ctypes as C
        lib = _capi.load()
        lib.vs_match_profile.argtypes = [C.c_int]
        lib.vs_match_profile_read.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float)]
        lib.vs_match_profile(1)
        for _ in range(args.steps):
            matcher.knn2_local_shard(q, t)
        torch.cuda.synchronize()
        pm, mm = C.c_float(0), C.c_float(0)
        ncalls = lib.vs_match_profile_read(C.byref(pm), C.byref(mm))
        lib.vs_match_profile(0)
        kernel_ms = float(pm.value)
        alg_bytes = BYTES_PER_MATCH * float(nq) * nt
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        traffic = None
        try:  # HBM bytes per launch from the committed PMC passes (profiles/), only for the workload they were taken on
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_match.json")))
            if (nq, nt) == (10000, 10000):
                traffic = pmc["hamming_partial_kernel_per_launch"]["hbm_bytes_corrected_upper"]
        except Exception:
            pass
        valu_ceiling = VALU_LANE_OPS / OPS_PER_MATCH / 1e9
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "kernel": "hamming_partial_kernel", "kernel_ms": kernel_ms, "merge_kernel_ms": float(mm.value),
                "profiled_calls": int(ncalls), "algorithmic_bytes_per_launch": alg_bytes,
                "algorithmic_model": "32 B per distance evaluation (one streamed train descriptor), SURVEY.md 8d",
                "traffic_source": "profiles/r01_pmc_match.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; "
                                  "FETCH_SIZE doubled per the gfx950 correction)",
                "valu_ceiling_gmatches": valu_ceiling,
                "valu_frac": (float(nq) * nt / (kernel_ms * 1e-3) / 1e9) / valu_ceiling,
                "valu_frac_of_measured_issue_rate": (float(nq) * nt / (kernel_ms * 1e-3) / 1e9) /
                                                    (VALU_LANE_OPS_MEASURED / OPS_PER_MATCH / 1e9),
                "note": "tiles are reused from SGPRs/VGPRs, so real HBM traffic is ~1000x below the streamed-operand "
                        "model and frac exceeds 1; the binding limit is integer VALU issue (valu_frac)"}
    replicas = None
    if use_dist and not args.no_frames:
        try:
            replicas = frames_replicas(ctx, dist, world, dev)
        except Exception as e:  # all ranks take the same path: the collectives inside stay matched
            replicas = {"error": repr(e)}
    host_abi = None
    if rank == 0 and world == 1:
        # the same workload through the HOST entry point (vs_hamming_knn2: descriptors arrive in pageable host memory,
        # results return to the host) with fresh contents every call, so both sets are uploaded: the PCIe-inclusive
        # rate.  Reported beside `value`, never as `value`.
        try:
            qh, th = q_np.copy(), t_np.copy()
            ctx.hamming_knn2(qh, th)
            reps = 20
            torch.cuda.synchronize()
            t0h = time.perf_counter()
            for i in range(reps):
                qh[0, 0] ^= 1 + (i & 1)   # new content at the same address: the descriptor cache must re-upload
                th[0, 0] ^= 1 + (i & 1)
                ctx.hamming_knn2(qh, th)
            dth = (time.perf_counter() - t0h) / reps
            host_abi = {"gmatches_per_s": float(nq) * nt / dth / 1e9, "ms_per_call": dth * 1e3,
                        "note": "vs_hamming_knn2 on pageable host arrays, fresh contents per call: content fingerprint + "
                                "H2D 2 x 320 KB + kernels + D2H 160 KB + synchronisation"}
        except Exception as e:
            host_abi = {"error": repr(e)}
    if rank == 0:
        line = {
            "metric": "10k x 10k 256-bit Hamming 2-NN brute-force match throughput", "value": value,
            "unit": "Gmatches/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32 (xor + popcount on 256-bit descriptors)", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[2]: %d x %d x 256-bit descriptors, k=2, per GPU%s"
                                   % (nq, nt, "" if world == 1 else "; %d query shards + RCCL all-gather (16 B/query, overlapped with the next step)" % world),
                       "queries_per_gpu": nq, "train": nt, "parallelism": "query-shard x%d" % world},
            "roofline": roof,
        }
        if host_abi is not None:
            line["host_abi"] = host_abi
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(nq, nt)
            except Exception as e:  # never lose the GPU numbers to a host-side problem
                line["cpu_baseline"] = {"error": repr(e)}
        if world == 1 and not args.no_frames:
            try:
                line["local_ba"] = ba_leg(ctx, cpu=not args.no_cpu_baseline)
            except Exception as e:
                line["local_ba"] = {"error": repr(e)}
        if not args.no_frames:
            try:
                line["frames"] = frames_leg(ctx, cpu=(world == 1 and not args.no_cpu_baseline))
                if replicas is not None:
                    line["frames"]["replicas"] = replicas
                if world > 1:
                    line["frames"]["parallelism"] = ("replica: detection, the 600 x 600 per-frame match and BA run on "
                                                     "rank 0's GPU only (north_star: detection and BA stay single-GPU)")
            except Exception as e:
                line["frames"] = {"error": repr(e)}
        print(json.dumps(line))
    if use_dist:
        dist.barrier()  # the other ranks wait here while rank 0 runs the (replica) frames leg and prints
        matcher.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
