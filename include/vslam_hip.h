/* vslam_hip.h -- C ABI of libvslam_hip.so: the MI355X (gfx950) per-frame tracking hot path.
 *
 * The reference (juuso-oskari/visual_slam, src/v2) has no FFI of its own: its native boundary is the pybind11
 * modules cv2 and g2o.  Each entry point below replaces one of those delegations and cites the reference call site
 * (file:line relative to the reference root) whose arithmetic it takes over.  The Python classes in
 * visual_slam_amd/ (FeatureExtractor, FeatureMatcher, BundleAdjustment ...) keep the reference's signatures and call
 * these functions through ctypes.
 *
 * Conventions
 *   - every function returns VS_OK (0) or a negative vs_status; the message is available from vs_last_error().
 *   - "host" entry points take caller-owned host buffers (C-contiguous unless a stride is given), copy in, run the
 *     kernels on the context's stream, copy out and return synchronously.
 *   - "_dev" entry points take device pointers that are already resident in HBM, enqueue on the given hipStream_t
 *     (passed as void*; NULL = the context's stream) and return WITHOUT synchronising.  They are what bench.py, the
 *     query-sharded matcher (device tensors handed to RCCL) and a resident map use.
 *   - one vs_ctx per process and GPU; a ctx is not thread-safe.
 *   - bit-exact contracts (integer work) and tolerance contracts (FP64 BA) are stated per function.
 */
#ifndef VSLAM_HIP_H
#define VSLAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vs_ctx vs_ctx;

typedef enum vs_status {
  VS_OK = 0,
  VS_EINVAL = -1, /* bad argument (null pointer, negative size, T < 2 for k=2 ...) */
  VS_ENOMEM = -2, /* host or device allocation failed */
  VS_EHIP = -3,   /* a HIP runtime call or kernel launch failed */
  VS_ENOTPD = -4, /* reserved: BA reports indefinite systems in vs_ba_result, it does not fail the call */
  VS_ECAP = -5,   /* an output capacity given by the caller is too small */
  VS_ENCCL = -6   /* RCCL is not loaded in the process, or the all-gather failed */
} vs_status;

#define VS_ABI_VERSION 5
#define VS_DESC_BYTES 32 /* BRIEF-256 */

/* ---- context ---------------------------------------------------------------------------------------------- */
int vs_abi_version(void);
int vs_create(vs_ctx** out, int device);
int vs_destroy(vs_ctx* ctx);
/* last error message of ctx (or of the failed vs_create when ctx == NULL); never NULL */
const char* vs_last_error(const vs_ctx* ctx);
/* the context's hipStream_t as void* (so torch / RCCL work can be ordered against it) */
void* vs_stream(vs_ctx* ctx);
/* auxiliary compute streams of the context (index 0 .. 1; NULL beyond): created together with the main stream so that
 * each sits on a hardware queue of its own -- for callers that keep several steps in flight (the query-sharded matcher,
 * vs_hamming_knn2_sharded_dev with compute_stream = one of these).  Owned by the context. */
void* vs_aux_stream(vs_ctx* ctx, int index);
int vs_synchronize(vs_ctx* ctx);

/* Pinned host memory (optional).  Host entry points accept any host pointer; when a frame lives in memory obtained
 * here (or otherwise pinned) it is DMA-ed directly instead of going through the context's staging buffer.  In the
 * reference the frame comes from cv2.imread (src/v2/frame.py:54); visual_slam_amd.frame.imread can decode into this. */
int vs_host_alloc(vs_ctx* ctx, size_t bytes, void** out);
int vs_host_free(vs_ctx* ctx, void* p);

/* ---- A2: gray conversion ------------------------------------------------------------------------------------
 * replaces np.mean(img, axis=2).astype(np.uint8)            (src/v2/frame.py:11)
 * gray[y][x] = (b + g + r) / 3 (integer division; exact for all 766 sums).  bgr rows are `stride` bytes apart. */
int vs_gray_mean3_u8(vs_ctx* ctx, const uint8_t* bgr, int w, int h, int stride, uint8_t* gray /*[h][w]*/);

/* ---- A3: keypoint detection ---------------------------------------------------------------------------------
 * replaces cv2.goodFeaturesToTrack(...) / the detector of cv2.ORB_create()   (src/v2/frame.py:8,11-12)
 * FAST-9 on the 16-pixel radius-3 circle with threshold `thr`; score = largest threshold at which the pixel is
 * still a corner; kept iff score is strictly greater than the score of all 8 neighbours (non-corners score 0);
 * pixels closer than `border` (>= 3) to the image edge are never keypoints.  If more than max_kp survive, the
 * max_kp with the highest score are kept (ties: lower y*w+x first).  Output order is row-major (y, then x).
 * xy[i] = (x, y) as float32 -- the layout cv2.KeyPoint_convert returns (src/v2/frame.py:14).  Bit-exact. */
int vs_fast9_detect(vs_ctx* ctx, const uint8_t* gray, int w, int h, int stride, int thr, int border, int max_kp,
                    float* xy /*[max_kp][2]*/, uint8_t* score /*[max_kp] or NULL*/, int* n_out);

/* ---- A4: description ----------------------------------------------------------------------------------------
 * replaces self.extractor.compute(img, kps)                 (src/v2/frame.py:7-8,13)
 * BRIEF-256 on 5x5 box sums with the committed pattern include/vs_brief_pattern.h.  Keypoints are rounded to the
 * nearest pixel (half to even); those whose 31x31 patch leaves the image are dropped, as extractor.compute drops
 * them: desc/keep_idx hold the survivors in input order, keep_idx[j] = index into xy.  Bit-exact. */
int vs_brief256(vs_ctx* ctx, const uint8_t* gray, int w, int h, int stride, const float* xy, int n,
                uint8_t* desc /*[n][32]*/, int32_t* keep_idx /*[n]*/, int* n_out);

/* ---- A2+A3+A4 fused: FeatureExtractor.compute_features ---------------------------------------------------------
 * replaces the whole body of compute_features(img)          (src/v2/frame.py:10-14)
 * == vs_gray_mean3_u8 -> vs_fast9_detect(border = 15) -> vs_brief256, in two kernel launches with one upload of
 * the BGR frame.  Because detection already excludes the 15-pixel border no keypoint is dropped by description. */
int vs_detect_describe_bgr(vs_ctx* ctx, const uint8_t* bgr, int w, int h, int stride, int thr, int max_kp,
                           float* xy /*[max_kp][2]*/, uint8_t* score /*[max_kp] or NULL*/,
                           uint8_t* desc /*[max_kp][32]*/, int* n_out);

/* device-resident variant: d_bgr rows are `pitch` bytes apart (pitch % 4 == 0, at least 4 bytes of slack after the
 * last row); outputs are device arrays xy float[max_kp][2], score u8[max_kp] (or NULL), desc u8[max_kp][32]
 * (8-byte aligned), n_out int32[1].  Enqueues two kernels on `stream` (NULL = the context's) without synchronising. */
int vs_detect_describe_bgr_dev(vs_ctx* ctx, const void* d_bgr, int w, int h, int pitch, int thr, int max_kp,
                               void* d_xy, void* d_score, void* d_desc, void* d_n_out, void* stream);

/* ---- A5: brute-force Hamming 2-NN ---------------------------------------------------------------------------
 * replaces cv2.BFMatcher(NORM_HAMMING).knnMatch(desc1, desc2, k=2)   (src/v2/frame.py:18,23)
 * For every query row the two train rows of smallest Hamming distance, ascending; ties -> lower train index
 * first.  idx[q] = {best, second}, dist[q] likewise.  nt >= 2 required (the reference raises on unpacking
 * otherwise, src/v2/frame.py:30); nq == 0 is legal.  Bit-exact. */
int vs_hamming_knn2(vs_ctx* ctx, const uint8_t* q /*[nq][32]*/, int nq, const uint8_t* t /*[nt][32]*/, int nt,
                    int32_t* idx /*[nq][2]*/, int32_t* dist /*[nq][2]*/);

/* ---- A5+A6: 2-NN + Lowe ratio test + ordered compaction -------------------------------------------------------
 * replaces knnMatch + the `m.distance < ratio * n.distance` loop     (src/v2/frame.py:23-47)
 * Keeps query i iff (double)d1 < ratio * (double)d2 -- the comparison Python performs.  Survivors are written in
 * query order: match_q[j], match_t[j], match_d[j].  Bit-exact. */
int vs_match_ratio(vs_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, double ratio,
                   int32_t* match_q /*[nq]*/, int32_t* match_t /*[nq]*/, int32_t* match_d /*[nq]*/, int* n_out);

/* device-resident variants (no copies, no synchronisation).  d_idx/d_dist: int32[nq][2] in HBM. */
int vs_hamming_knn2_dev(vs_ctx* ctx, const void* d_q, int nq, const void* d_t, int nt, void* d_idx, void* d_dist,
                        void* stream);
/* same, one 16-byte row per query: d_out int32[nq][4] = (idx0, idx1, dist0, dist1) -- the layout the query-sharded
 * matcher hands to the RCCL all-gather (north_star: "all-gather of per-shard best matches") */
int vs_hamming_knn2_packed_dev(vs_ctx* ctx, const void* d_q, int nq, const void* d_t, int nt, void* d_out,
                               void* stream);
/* Query-sharded match, one rank's step (north_star: "Shard only the brute-force descriptor match (query-split) across the
 * 8 GPUs of one node with an RCCL all-gather of per-shard best matches over xGMI"; the call it shards is knnMatch,
 * src/v2/frame.py:23).  Rank `rank` of `world` owns queries [rank*per, rank*per + nq_shard) of the full set, per =
 * ceil(Q / world); the train set is replicated, so rows carry GLOBAL train indices and no cross-rank tie-break exists.
 *   0. ordering of the compute stream, so that a caller may keep several steps in flight on several streams:
 *      - if `done_event` != NULL the compute stream first waits for it AS RECORDED BY THE PREVIOUS STEP ON THIS GATHER
 *        BUFFER (that step's all-gather sends from, and receives into, the rows the kernel overwrites); keep one event
 *        per gather buffer.  An event that was never recorded counts as complete;
 *      - if `after_stream` != NULL (and is not the compute stream) the compute stream is ordered behind everything
 *        enqueued on after_stream so far: the producers of q / t and the consumers of d_gathered's previous results.
 *      The inputs of a step must stay untouched until that step's done_event has completed.
 *   1. the match kernel writes this rank's packed rows straight into its slot of d_gathered (int32[world*per][4],
 *      16-byte aligned) on `compute_stream` (NULL: the context's stream);
 *   2. if nccl_comm != NULL (required when world > 1): an event orders `comm_stream` behind the kernel and ONE in-place
 *      ncclAllGather of per*4 int32 per rank is enqueued there; `done_event` (a hipEvent_t, may be NULL) is recorded
 *      on comm_stream after it.  nccl_comm is the caller's ncclComm_t; RCCL is resolved at run time from the library
 *      already loaded in the process (dlopen), libvslam_hip.so does not link it.  With nccl_comm == NULL no collective
 *      runs (same code path) and done_event is recorded on the compute stream after the kernel.
 * Nothing synchronises the host. */
int vs_hamming_knn2_sharded_dev(vs_ctx* ctx, const void* d_q_shard, int nq_shard, const void* d_t, int nt,
                                void* d_gathered, int per, int rank, int world, void* nccl_comm, void* compute_stream,
                                void* comm_stream, void* done_event, void* after_stream);
/* The device entry points only enqueue, so they cannot report what a launch finds out while it runs: a folding workgroup of
 * hamming_knn2_kernel that gave up its bounded wait for a train chunk (a workgroup that never ran; ~0.5 s) writes -1 rows
 * and raises a pinned flag.  vs_match_status returns VS_EHIP (once, and clears the flags) when that happened to any match
 * launch of the context on any stream since the last call.  Call it where results are consumed -- after the done event or
 * the synchronisation the caller waits on (ShardedMatcher does, in collect / close; the tracking period checks by itself
 * before it hands a frame out).  A stream's next launch still refuses to start on a raised flag, as before. */
int vs_match_status(vs_ctx* ctx);
/* d_n_out: one int32 in HBM receiving the match count */
int vs_match_ratio_dev(vs_ctx* ctx, const void* d_q, int nq, const void* d_t, int nt, double ratio, void* d_match_q,
                       void* d_match_t, void* d_match_d, void* d_n_out, void* stream);

/* ---- next row: two-view DLT triangulation --------------------------------------------------------
 * replaces helper_functions.triangulate(pose1, pose2, pts1, pts2)     (src/v2/helper_functions.py:281-291)
 * and the depth terms of the cheirality filter                        (src/v2/main.py:291-309)
 * P1, P2: 3x4 projection matrices K*[R|t] row-major (CameraProjectionMatrix2, helper_functions.py:376-377);
 * pts1/pts2: n rows of `stride` doubles whose first two entries are (u, v) (the reference passes homogeneous N x 3);
 * X4[i] = unit right singular vector of the smallest singular value of the 4x4 DLT matrix, sign chosen so w >= 0
 * (the reference returns LAPACK's arbitrary sign and divides by w at once).  If depth != NULL, depth[i] = (z of
 * T1*[X/w;1], z of T2*[X/w;1]) for the 3x4 (or top of 4x4) world-to-camera transforms T1, T2 -- what main.py's filter
 * `(proj1[2] > 0) & (proj2[2] > 0) & (proj2[2] < 1) & (proj1[2] < 1)` reads.  FP64; equals np.linalg.svd to ~1e-12. */
int vs_triangulate_dlt(vs_ctx* ctx, const double* P1, const double* P2, const double* pts1, const double* pts2, int n,
                       int stride, double* X4 /*[n][4]*/, const double* T1, const double* T2,
                       double* depth /*[n][2] or NULL*/);

/* ---- next row: PnP-RANSAC ------------------------------------------------------------------------
 * replaces cv2.solvePnPRansac(objectPoints, imagePoints, K, [], rvec, tvec, useExtrinsicGuess=True)  (src/v2/main.py:196)
 * Structure of OpenCV's routine with its defaults (ITERATIVE, 100 iterations, 8 px, confidence 0.99): hypothesis h =
 * LM refinement of the extrinsic guess on 5 sampled correspondences; inliers: squared reprojection error <= thr^2;
 * iteration budget updated by RANSACUpdateNumIters after every improvement; best model refined on its inliers.
 * Own specification where OpenCV cannot be pinned: samples from splitmix64(splitmix64(seed) ^ ((h << 20) + k)), LM = the g2o-style
 * LM of vs_ba_solve (one free camera, fixed points, no robust kernel, `refine_iters` iterations).
 * pose0 / pose_out: 4x4 camera-to-world row-major (invert for rvec/tvec); inliers: indices, ascending. */
int vs_pnp_ransac(vs_ctx* ctx, const double* obj /*[n][3]*/, const double* img /*[n][2]*/, int n, double fx, double fy,
                  double cx, double cy, const double* pose0, int iterations, double reproj_err, double confidence,
                  uint64_t seed, int refine_iters, double* pose_out, int32_t* inliers /*[n]*/, int* n_inliers, int* found);

/* ---- next row: two-view initialisation ------------------------------------------------------------
 * vs_essential_ransac replaces cv2.findEssentialMat(pts1, pts2, method=RANSAC, prob=0.999, threshold) on K-normalised
 * points (src/v2/helper_functions.py:47-52, called from main.py:102); vs_recover_pose replaces
 * cv2.recoverPose(E, pts1, pts2, cameraMatrix=K, distanceThresh=50) (helper_functions.py:175-176, main.py:109).
 * OpenCV's structure (Sampson error vs threshold^2, RANSACUpdateNumIters; decomposeEssentialMat's four
 * candidates and the cheirality vote z*w > 0, z < dist in both cameras) with an 8-point minimal solver, counter-based
 * sampling, Jacobi SVDs and one linear re-fit of the winner to its inliers, of this library's own specification.  x1/x2: K-normalised [n][2]; E row-major, x2^T E x1 = 0;
 * mask of vs_essential_ransac: 1/0; of vs_recover_pose: 255/0 (main.py checks == 255); X: homogeneous [n][4], w >= 0. */
int vs_essential_ransac(vs_ctx* ctx, const double* x1, const double* x2, int n, double threshold, double prob,
                        int max_iters, uint64_t seed, double* E /*[9]*/, uint8_t* mask /*[n]*/, int* n_inliers, int* found);
int vs_recover_pose(vs_ctx* ctx, const double* E, const double* x1, const double* x2, int n, double dist_thresh,
                    double* R /*[9]*/, double* t /*[3]*/, uint8_t* mask /*[n]*/, double* X /*[n][4]*/, int* n_good);

/* ---- one tracking period resident on the device ----------------------------------------------------
 * replaces the body of the per-frame tracking loop (src/v2/main.py:181-214): process_frame, GetImagePointsWithFrameID,
 * match_features, solvePnPRansac, AddPointToFrameCorrespondences, motionOnlyBundleAdjustement.
 * vs_track_begin uploads the last key frame's map points (xyz, descriptors, in Map.GetImagePointsWithFrameID order) and
 * its pose once; every vs_track_frame uploads only the image and runs detect+describe -> match (map = query, frame =
 * train, Lowe ratio) -> PnP-RANSAC from the previous pose -> append the observations -> motion-only BA over all poses of
 * the period (LocalBA.py:195-229 re-optimises all of them every frame), without the host rebuilding or re-uploading the
 * period's observations.  Same kernels and arithmetic as vs_detect_describe_bgr / vs_match_ratio / vs_pnp_ransac /
 * vs_ba_solve.  poses_out: [n_frames+1][16] camera-to-world, pose 0 = the key frame; *pnp_found = number of inliers of the
 * PnP model (0: none found, the previous pose was the start).  Optional outputs may be NULL. */
int vs_track_begin(vs_ctx* ctx, const double* xyz /*[n][3]*/, const uint8_t* desc /*[n][32]*/, int n_points,
                   const double* key_pose /*4x4*/, double fx, double fy, double cx, double cy, int max_frames, int max_kp,
                   int pnp_iterations /*0: start BA from the previous pose*/);
int vs_track_frame(vs_ctx* ctx, const uint8_t* bgr, int w, int h, int stride, int thr, double ratio,
                   double pnp_reproj_err, double pnp_confidence, uint64_t seed, int lm_iterations, double huber_delta,
                   double* poses_out, int* n_poses_out, int* n_matches, int* pnp_found, float* xy_out /*[max_kp][2]*/,
                   uint8_t* desc_out /*[max_kp][32]*/, int* n_kp_out, int32_t* match_q /*[n_points]*/,
                   int32_t* match_t /*[n_points]*/);
/* The optional per-frame arrays of the newest frame handed out by vs_track_frame / vs_track_frame_pipelined, fetched afterwards:
 * main.py needs a tracked frame's key points, descriptors and match lists only when the frame becomes a key frame (main.py:221-236,
 * one frame in twenty), so a caller passes NULL for them per frame and asks here for the one frame that needs them.  Valid until
 * the next frame is submitted (frame by frame) / the next but one (pipelined), or the period ends.  Any output may be NULL. */
int vs_track_last_frame(vs_ctx* ctx, float* xy_out /*[max_kp][2]*/, uint8_t* desc_out /*[max_kp][32]*/, int* n_kp_out,
                        int32_t* match_q /*[n_points]*/, int32_t* match_t /*[n_points]*/, int* n_matches_out);
/* Pipelined variant for recorded streams: the call for frame k+1 first enqueues the back half (append, PnP, BA) of frame
 * k on the context's stream, then prepares frame k+1's front half (upload, detect, match) on a second stream while that
 * runs, and finally returns frame k's results (*has_result = 1; 0 on the first call).  bgr == NULL flushes the pending
 * frame.  The parameters given with a frame are the ones used for it.  Results are identical to vs_track_frame.
 * With PnP and LM on and the motion-only solve in one launch, frame k+1's back half is enqueued too, behind frame k's and
 * before frame k's results are known (its kernels read what they need from frame k on the device and wait for their own
 * front half through a tagged word): the GPU passes from one back half to the next without the host in between.
 * Consecutive back halves alternate between the context's stream and vs_aux_stream(ctx, 1). */
int vs_track_frame_pipelined(vs_ctx* ctx, const uint8_t* bgr, int w, int h, int stride, int thr, double ratio,
                             double pnp_reproj_err, double pnp_confidence, uint64_t seed, int lm_iterations,
                             double huber_delta, int* has_result, double* poses_out, int* n_poses_out, int* n_matches,
                             int* pnp_found, float* xy_out, uint8_t* desc_out, int* n_kp_out, int32_t* match_q,
                             int32_t* match_t);
/* Host-fed variant for callers that match and estimate the start pose themselves, i.e. the reference's class API driven
 * call by call (FeatureMatcher.match_features, cv2.solvePnPRansac, Map.AddPointToFrameCorrespondences, then
 * BundleAdjustment.motionOnlyBundleAdjustement, src/v2/main.py:185-214): appends one frame -- observation i = (index of
 * the map point in the xyz array given to vs_track_begin, image point uv[i]) and the frame's start pose (4x4
 * camera-to-world) -- to the resident period and runs the motion-only BA over all its poses (LocalBA.py:195-229);
 * lm_iterations = 0 appends only.  Neither the period's observations nor its points are rebuilt or uploaded again. */
int vs_track_push_frame(vs_ctx* ctx, const int32_t* point_idx /*[m]*/, const double* uv /*[m][2]*/, int m,
                        const double* pose16, int lm_iterations, double huber_delta, double* poses_out, int* n_poses_out);
/* The same, for the class API driven call by call WITHOUT giving up the residency of the front half and the PnP: the three
 * calls below together are one vs_track_frame, cut where the reference's call sequence needs values on the host
 * (src/v2/main.py:181-214).  The key frame's descriptors given to vs_track_begin must be the map points' descriptors.
 *   vs_track_front       Frame.process_frame + FeatureMatcher.match_features (main.py:181,185): upload, detect + describe,
 *                        match the map points (queries) against the frame (train), append the matches as the new frame's
 *                        observations; ONE synchronisation.  Returns key points, descriptors and the matches
 *                        (match_q = map point index, match_t = key point index, match_d = Hamming distance).
 *   vs_track_back_begin  cv2.solvePnPRansac (main.py:196-197): PnP-RANSAC on those matches and, enqueued right behind it, the
 *                        motion-only BA over all poses (LocalBA.py:195-229); returns when the PnP outcome is in (pose16 =
 *                        camera-to-world 4x4, the inlier indices into the match list).  The extrinsic guess is the period's
 *                        previous pose, or guess_pose16 (camera-to-world 4x4) when the caller passes its own -- main.py:193-194
 *                        builds rvec / tvec from W_T_prev itself, i.e. hands OpenCV the camera-to-world transform where a
 *                        world-to-camera one is expected; a drop-in has to start from what the caller passed.  obj_as_f32 != 0:
 *                        the object points are rounded to float32 when read, as main.py:196's objectPoints.astype(np.float32)
 *                        does (the resident rows stay float64 for the bundle adjustment, as the reference's map does).
 *   vs_track_back_end    BundleAdjustment.motionOnlyBundleAdjustement (main.py:213-214): waits for that BA and returns all
 *                        poses of the period.
 * A front half that is not followed up (the caller went another way) costs nothing: the next frame overwrites its rows.
 * The caller is responsible for passing the same matches and start pose on to the rest of the API as the device used
 * (visual_slam_amd/map.py checks exactly that and otherwise restarts the period from the map). */
int vs_track_front(vs_ctx* ctx, const uint8_t* bgr, int w, int h, int stride, int thr, double ratio,
                   float* xy_out /*[max_kp][2]*/, uint8_t* desc_out /*[max_kp][32]*/, int* n_kp_out,
                   int32_t* match_q /*[n_points]*/, int32_t* match_t, int32_t* match_d, int* n_matches);
int vs_track_back_begin(vs_ctx* ctx, double pnp_reproj_err, double pnp_confidence, uint64_t seed, int lm_iterations,
                        double huber_delta, const double* guess_pose16 /*NULL: the period's previous pose*/, int obj_as_f32,
                        int* found, double* pose16, int32_t* inliers /*[n_matches]*/, int* n_inliers);
int vs_track_back_end(vs_ctx* ctx, double* poses_out /*[n_frames+1][16]*/, int* n_poses_out);
int vs_track_end(vs_ctx* ctx);

/* ---- A9-A16: bundle adjustment ------------------------------------------------------------------------------
 * replaces the g2o graph the reference builds and optimises          (src/v2/LocalBA.py:20-94,115-131,39-42)
 *   solver      : Levenberg-Marquardt( BlockSolverSE3( Cholesky ) ), points marginalised (Schur complement)
 *   add_pose    : poses[i] = 4x4 camera-to-world, row-major; internally (t, unit q) as g2o::SBACam (LocalBA.py:56-65)
 *   add_point   : points[j] (LocalBA.py:68-77)
 *   add_edge    : EdgeProjectP2MC residual pi(K * w2n * [X;1]) - uv, information I2 (or obs_info), Huber(huber_delta)
 *                 (LocalBA.py:79-94)
 *   AddScalingEdge: EdgeSBAScale residual m - |t_child - t_parent|, information 1, RobustKernelDCS (LocalBA.py:115-131)
 *   optimize    : initialize_optimization(); optimize(max_iterations) (LocalBA.py:39-42)
 * All arithmetic in FP64.  Contract: final poses within 1e-4 relative Frobenius norm of the CPU oracle. */
typedef struct vs_ba_problem {
  int32_t n_poses, n_points, n_obs, n_scale;
  const double* poses;        /* [n_poses][16] */
  const uint8_t* pose_fixed;  /* [n_poses] */
  const double* points;       /* [n_points][3] */
  const uint8_t* point_fixed; /* [n_points] */
  const int32_t* obs_pose;    /* [n_obs] index into poses */
  const int32_t* obs_point;   /* [n_obs] index into points */
  const double* obs_uv;       /* [n_obs][2] */
  const double* obs_info;     /* [n_obs][3] = (xx, xy, yy) or NULL for identity */
  const int32_t* scale_parent; /* [n_scale] index into poses */
  const int32_t* scale_child;  /* [n_scale] */
  const double* scale_meas;    /* [n_scale] */
  double fx, fy, cx, cy;
  double huber_delta; /* sqrt(5.991) in the reference; <= 0 disables the robust kernel */
  double dcs_phi;     /* RobustKernelDCS delta, 1.0 in g2o */
  int32_t max_iterations; /* 10 in the reference (LocalBA.py:39) */
  int32_t reserved;
} vs_ba_problem;

typedef struct vs_ba_result {
  double* poses_out;     /* [n_poses][16] optimised camera-to-world matrices (may alias nothing in the problem) */
  double* points_out;    /* [n_points][3] */
  double* chi2_trace;    /* [max_iterations] robust chi2 after each outer iteration, or NULL */
  double* lambda_trace;  /* [max_iterations] damping after each outer iteration, or NULL */
  double chi2_initial, chi2_final, lambda_final;
  int32_t iterations;    /* outer LM iterations executed */
  int32_t trials;        /* inner trials (linear solves) executed in total */
  int32_t not_pd;        /* trials whose reduced system was not positive definite (the reference's debug.txt case) */
  int32_t terminated;    /* 1 if LM stopped before max_iterations (10 rejected trials or zero gain) */
  /* optional per-TRIAL record (one row per linear solve, in execution order), or NULL:
   * row = (lambda the trial was solved with, robust chi2 of the trial state [DBL_MAX when the reduced system was not
   * positive definite], gain ratio rho, 1.0 if the Cholesky succeeded else 0.0).  At most trial_trace_cap rows are
   * written; `trials` tells how many exist. */
  double* trial_trace;   /* [trial_trace_cap][4] */
  int32_t trial_trace_cap;
  int32_t reserved;
} vs_ba_result;

int vs_ba_solve(vs_ctx* ctx, const vs_ba_problem* problem, vs_ba_result* result);

/* Test hook: the dense solver of the reduced camera system on its own -- what g2o's LinearSolverCholmod does inside
 * BlockSolver::solve.  Factorises the symmetric n x n matrix S (row-major, n a multiple of 6, lower triangle
 * read) with the SAME kernels vs_ba_solve uses for that size (one workgroup in LDS up to 126, blocked panels in HBM
 * beyond) and solves S x = b.  *ok = 0 if a pivot was not positive (the reference's debug.txt case: g2o dumps the
 * matrix and rejects the LM trial), x is then untouched. */
int vs_ba_debug_cholesky(vs_ctx* ctx, const double* S, int n, const double* b, double* x, int* ok);

#ifdef __cplusplus
}
#endif
#endif /* VSLAM_HIP_H */
