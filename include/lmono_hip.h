/*
 * include/lmono_hip.h -- C ABI of the MI355X-native lmono per-scan hot path.
 *
 * The reference (bobocode/lmono, ROS-1 C++) has no FFI / plugin registry; its seams are ROS topics,
 * the Estimator call surface and the Ceres cost-function ABI.  Each entry point below
 * names the reference interface it stands in for.  All functions return 0 on success or a negative
 * LMONO_E* code, never throw, and are re-entrant per context.  Pointers suffixed _d are device (HBM)
 * pointers, _h host pointers.  No torch / HIP types appear in signatures (streams travel as void*).
 */
#ifndef LMONO_HIP_H
#define LMONO_HIP_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LMONO_OK            0
#define LMONO_EINVAL       -1   /* bad argument                                            */
#define LMONO_ENODEV       -2   /* no usable gfx950 device / HIP runtime error              */
#define LMONO_ENOMEM       -3   /* device allocation failed                                 */
#define LMONO_ECAPACITY    -4   /* batch exceeds the capacity it was created with           */
#define LMONO_ESCAN        -5   /* a scan violated a kernel limit (see lmono_batch_status)  */

#define LMONO_MAX_RINGS     64
#define LMONO_RING_CAP      4096  /* max points of one ring after filtering                 */
#define LMONO_MAX_SHARP     (64 * 6 * 2)
#define LMONO_MAX_LESS_SHARP (64 * 6 * 20)
#define LMONO_MAX_FLAT      (64 * 6 * 4)

typedef struct lmono_ctx lmono_ctx;
typedef struct lmono_scan_batch lmono_scan_batch;

/* ---- context ---------------------------------------------------------------------------- */
lmono_ctx  *lmono_create(int device);           /* NULL when HIP / the device is unavailable */
void        lmono_destroy(lmono_ctx *);
const char *lmono_last_error(const lmono_ctx *);
int         lmono_set_stream(lmono_ctx *, void *hip_stream); /* hipStream_t; NULL = default  */
/* Run this context on a non-blocking stream the library creates (and destroys with the context).  No reference counterpart: the
 * reference runs marginalisation inline (Estimator.cc:1280-1470); the host mirror uses a second context on its own stream so that it
 * overlaps the next frame's solve.  A context is used by one host thread at a time; two contexts on two threads do not serialise
 * (the host-array entry points carve their scratch from a per-context arena and copy on the context stream only). */
int         lmono_use_own_stream(lmono_ctx *);
int         lmono_synchronize(lmono_ctx *);
/* Tuning / test switches of a context (no reference counterpart).  LMONO_OPT_CORR_TILE selects the laserOdometry correspondence
 * search: 3 (default) = flattened candidate sweeps over the (azimuth bin, scan line) index (k_corr_flat; needs no hash grid);
 * 0 = 32 lanes per feature point on a 1 m hash grid (k_correspond + k_grid_build: the round-1 search, compiled into the diagnostic
 * build liblmono_hip_diag.so only, where the two are checked against each other -- tests/diag_search_modes.py).  The product library
 * refuses every value but 3.  (Rounds 2-3 carried four more formulations -- LDS sector tiles, thread per feature, a sector-staged
 * flat search, one persistent workgroup per chain; all measured slower, profiles/r2/NOTES.md, profiles/r3/NOTES.md sections 2-3 --
 * they were removed in round 4 and live in the history at commit d914e8c.)                                                        */
#define LMONO_OPT_CORR_TILE 0
/* test hook: n > 0 makes the default search (mode 3) hand every n-th feature point to its fall-back kernel (k_correspond_list), which
 * then runs without hash grids; results must not change.  0 = off.                                                              */
#define LMONO_OPT_DEFER_EVERY 1
/* chain groups of lmono_odom_batch[_d]: G = 1 .. 8 groups of chains advance on G HIP streams side by side (results unchanged).
 * Default 4 (one per hardware queue); at most 3 for a context on a caller-created stream (the runtime keeps one queue for the null
 * stream); reduced until every group holds at least 32 chains, so a 1-chain call is ungrouped.                                */
#define LMONO_OPT_ODOM_STREAMS 2
/* lead-in of lmono_odom_batch[_d]'s chains: >= 0 = only the last N lead-in scan pairs of a chain use all feature points, the earlier
 * ones a quarter of them (a lead-in pair only produces the next pair's warm start); -1 (default) = every pair uses all of them.
 * An accuracy / speed knob like n_chains and lead: n_chains = 1 is unaffected.                                                  */
#define LMONO_OPT_LEAD_FULL 3
/* boundary validation of lmono_odom_batch[_d]'s chained schedule (n_chains > 1), in units of 1e-9: after the chains have run, the warm
 * start every chain used for its first owned scan pair (its own lead-in's estimate) is compared on the device with the increment the
 * strictly sequential schedule warm-starts from (the predecessor chain's last one): residual = max(|dq_i|, 0.1 |dt_i| / m).  Chains
 * above the tolerance are re-started from the sequential warm start and re-run until their increments agree with the stored ones
 * within it (everything behind stands), in rounds until no boundary is flagged; lmono_odom_boundary_report tells what happened.
 * Default 1000 (1e-6: 2e-6 rad, 1e-5 m); 0 = no validation (and no synchronisation inside lmono_odom_batch_d).                    */
#define LMONO_OPT_BOUNDARY_TOL 4
/* workgroups per window of lmono_ba_solve (the K = 1 / 2 / 4 / 8 workgroups of a window share its linearisations and candidate costs; the sums are formed
 * per 16-observation segment and added in segment order, so the result is the same bit for bit whatever K is): 0 (default) = as many as keep the
 * batch within half the device's compute units (hipDeviceProp_t::multiProcessorCount; on 256 CUs: 8 for up to 16 windows, 4 for up to 32, 2 for up to 64, else 1; at most 4
 * when the windows average fewer than 64 segments); 1, 2, 4, 8 = that many.  A cluster whose workgroups do not all become resident gives up (bounded polls) and
 * lmono_ba_batch_read solves the batch again with one workgroup per window: same bytes.  Takes effect at the next
 * lmono_ba_batch_create / _update of the context (the scratch is sized then).  (Slot 5 was round 3's LMONO_OPT_ODOM_PERSIST, removed in round 4.)   */
#define LMONO_OPT_BA_CLUSTER 5
/* lead-in seeding of lmono_odom_batch[_d]'s chains (slot 6: round 3's LMONO_OPT_CORR_SECT, removed since): 0 = every lead-in starts from the identity and
 * keeps its own estimates (the reference's initial para_q / para_t); 1 = after the first step of the pass the state of every chain that is still in its
 * lead-in becomes the component-wise median of the first results of chains c - 1, c, c + 1 (a constant-velocity prior across 1.8 s that rejects a first
 * pair gone wrong in clutter).  An accuracy / speed knob like lead: results are validated at the chain boundaries either way; n_chains = 1 is unaffected. */
#define LMONO_OPT_LEAD_SEED 6
#define LMONO_OPT_COUNT     7
int         lmono_set_option(lmono_ctx *, int key, int value);
int         lmono_get_option(lmono_ctx *, int key, int *value);     /* the configured value (option values may be negative) */
const char *lmono_version(void);

/* ---- LiDAR front end: A-LOAM scanRegistration::laserCloudHandler ------------------------ *
 * Reference interface: ROS node "ascanRegistration" subscribing /velodyne_points and
 * publishing /laser_cloud_{sharp,less_sharp,flat,less_flat} (source absent from the reference tree:
 * /root/reference/.gitmodules:1-3, README.md:54-60; behavioural spec SURVEY.md Appendix A.1).
 * A batch holds the device-resident working set of n independent scans.                      */
lmono_scan_batch *lmono_batch_create(lmono_ctx *, int n_scans_cap, int64_t total_points_cap);
void              lmono_batch_destroy(lmono_scan_batch *);

/* xyzi_d: [total][4] float32 (KITTI .bin layout) already resident in HBM; offsets_h: [n_scans+1]
 * point offsets of each scan.  n_lines in {16,32,64}; min_range = A-LOAM `minimum_range`.    */
int lmono_scanreg_batch(lmono_ctx *, lmono_scan_batch *, const float *xyzi_d, const int64_t *offsets_h,
                        int n_scans, int n_lines, float min_range);
/* Same with the scans in HOST memory (what the reference's node callback receives in a sensor_msgs/PointCloud2, or a
 * KITTI .bin file read from disk): the points are staged into a batch-owned HBM buffer on the context stream first.
 * This is the PCIe-inclusive entry; the resident-in-HBM one above is what bench.py times.                       */
int lmono_scanreg_batch_h(lmono_ctx *, lmono_scan_batch *, const float *xyzi_h, const int64_t *offsets_h,
                          int n_scans, int n_lines, float min_range);

/* Streamed input (PCIe-inclusive operation): lmono_batch_stage_h enqueues the H2D copy of a working set's points into the batch's own
 * staging buffer on the library's copy stream and returns at once (xyzi_h should be pinned: lmono_host_alloc, or hipHostRegister'ed by the
 * caller); lmono_scanreg_batch_staged makes the compute stream wait for that copy ON THE DEVICE and registers the staged scans.  With two
 * batches the copy of working set i + 1 runs under the compute of working set i; a batch's staging buffer is not overwritten before its
 * front end has read it.  No reference counterpart (the reference receives one message at a time in host memory).                    */
void *lmono_host_alloc(lmono_ctx *, size_t bytes);
void  lmono_host_free(lmono_ctx *, void *);
int lmono_batch_stage_h(lmono_ctx *, lmono_scan_batch *, const float *xyzi_h, int64_t total_points);
int lmono_scanreg_batch_staged(lmono_ctx *, lmono_scan_batch *, const int64_t *offsets_h, int n_scans, int n_lines, float min_range);

/* counts_h: [n_scans][6] = n_cloud, n_sharp, n_less_sharp, n_flat, n_less_flat, status.
 * status bits: 1 a ring holds more than LMONO_RING_CAP points (scan contributes no features); 2 a "last" cloud did not fit
 * its hash grid (too many points, a table page without a free slot, or a cell beyond +-1024 m: that table is empty, the
 * next scan finds no correspondences in it); 4 scan-line ids too disordered for the windowed walk and 16 a 1 m cell with
 * more than 32767 points (both: the generic, slower search path is used, results unchanged); 8 walk truncated.       */
int lmono_batch_counts(lmono_ctx *, lmono_scan_batch *, int32_t *counts_h);
/* which: 0 ring-sorted cloud, 1 sharp, 2 less_sharp, 3 flat, 4 less_flat.  out_h: [cap][4] float32.
 * Returns the number of points copied (>= 0) or a negative error.                               */
int lmono_batch_get_cloud(lmono_ctx *, lmono_scan_batch *, int scan, int which, float *out_h, int cap);
/* curvature (float32) and label (int32: 2 sharp, 1 less sharp, -1 flat, 0 other) of the sorted cloud */
int lmono_batch_get_curvature(lmono_ctx *, lmono_scan_batch *, int scan, float *curv_h, int32_t *label_h, int cap);

/* ---- LiDAR odometry: A-LOAM laserOdometry main loop + lidarFactor.hpp + ceres::Solve ------- *
 * Reference interface: ROS node "alaserOdometry" (feature clouds in, /laser_odom_to_init out; spec
 * SURVEY.md Appendix A.2/A.3).  Runs scan-to-scan odometry over the scans of a registered batch.
 * The sequence is cut into n_chains contiguous ranges processed concurrently; a range starting at
 * scan s > 0 starts `lead` scans early from an identity warm start and discards its lead-in
 * (n_chains = 1, lead = 0 = the strictly sequential reference behaviour).
 * incr_h / poses_h (either may be NULL): [n_scans][7] = q(x,y,z,w), t of T(k-1 -> k) and of the
 * accumulated pose t_w += q_w * t, q_w = q_w * q.                                              */
int lmono_odom_batch(lmono_ctx *, lmono_scan_batch *, int n_chains, int lead, double *incr_h, double *poses_h);
/* Same, results stay on the device: incr_d, poses_d [n_scans][7] float64 (may be NULL).       */
int lmono_odom_batch_d(lmono_ctx *, lmono_scan_batch *, int n_chains, int lead, double *incr_d, double *poses_d);

/* What the boundary validation of the last lmono_odom_batch[_d] / lmono_odom_shard_* call on this batch did (LMONO_OPT_BOUNDARY_TOL).
 * resid_h [n_chains] (may be NULL): residual of every chain's warm start at the first check (entry 0: 0, or the shard's external
 * boundary); rerun_h [n_chains] (may be NULL): scan pairs re-run per chain.  No reference counterpart (the reference is sequential). */
typedef struct {
    int n_chains;
    int flagged;        /* boundaries above the tolerance, summed over the rounds                                   */
    int chains_rerun;   /* distinct chains re-started                                                              */
    int pairs_rerun;    /* scan pairs re-run by them                                                               */
    int rounds;         /* check + repair rounds (a repair that reaches its chain's end can flag the next boundary) */
    int unresolved;     /* boundaries still above the tolerance behind the last repair round (counted by one more check when the
                         * round cap n_chains + 1 ends the loop; 0 otherwise)                                       */
    double tol, max_resid, repair_ms;
} lmono_boundary_report;
int lmono_odom_boundary_report(lmono_ctx *, lmono_scan_batch *, lmono_boundary_report *rep, double *resid_h, int32_t *rerun_h, int cap);

/* Scan-range sharding over the GPUs of a node: the batch holds scans [first_owned - lead', n) of a longer sequence, the
 * first `first_owned` of them are the previous rank's (chain 0's lead-in; their increments are not produced).  lmono_odom_shard_d runs
 * the chains over the owned scans; after ONE all-gather of every rank's last increment (incr_d[n - 1]), lmono_odom_shard_validate
 * checks chain 0's warm start against the previous rank's last increment prev_incr_h[7] exactly like an inner boundary and repairs;
 * *changed_last = 1 when this rank's own last increment changed (the next rank must validate again).  incr_d [n][7] as above.
 * lmono_odom_shard_main_d is lmono_odom_shard_d WITHOUT the validation of the boundaries inside the rank: the caller exchanges the last
 * increments right after the main pass and lmono_odom_shard_validate then validates every boundary of the rank -- the external one too --
 * in one set of repair rounds (prev_incr_h = NULL on the rank that owns the sequence's first scan: inner boundaries only).               */
int lmono_odom_shard_d(lmono_ctx *, lmono_scan_batch *, int n_chains, int lead, int first_owned, double *incr_d);
int lmono_odom_shard_main_d(lmono_ctx *, lmono_scan_batch *, int n_chains, int lead, int first_owned, double *incr_d);
int lmono_odom_shard_validate(lmono_ctx *, lmono_scan_batch *, const double *prev_incr_h, double *incr_d, int *changed_last);

/* Online form: ONE scan per call, as the reference's nodes run (ROS callbacks laserCloudHandler -> laserOdometry at sensor rate; lmono
 * consumes the result per frame: mono_lidar_mapping/src/image_process/MeasurementManager.cc:17-24, config/kitti_config_00.yaml:7-8).
 * A stream keeps the previous scan's feature clouds and search index on the device ("last") and A-LOAM's para_q / para_t between calls
 * (SURVEY.md A.2): lmono_odom_step registers the new scan (scanRegistration), runs the scan pair (2 x [correspondences -> <= 4 LM
 * iterations]) warm-started from the previous increment and returns q_last_curr / t_last_curr and the accumulated q_w_curr / t_w_curr
 * -- the same numbers lmono_odom_batch(n_chains 1, lead 0) gives for the same scans, bit for bit.  xyzi: [n_points][4] float32, host
 * (on_device 0) or HBM (1).  use_warm_start != 0: q_last_curr / t_last_curr are read as the pair's warm start instead of the stream's
 * own.  info [8] (may be NULL): n_cloud, n_sharp, n_less_sharp, n_flat, n_less_flat, status, LM iterations (outer 0 << 8 | outer 1),
 * residual blocks of the last solve.  history >= 1 slots of scans are kept (the slots are reused cyclically).  Synchronous.         */
typedef struct lmono_odom_stream lmono_odom_stream;
lmono_odom_stream *lmono_odom_stream_create(lmono_ctx *, int max_points_per_scan, int n_lines, float min_range, int history);
void lmono_odom_stream_destroy(lmono_odom_stream *);
int lmono_odom_step(lmono_ctx *, lmono_odom_stream *, const float *xyzi, int n_points, int on_device, int use_warm_start,
                    double *q_last_curr, double *t_last_curr, double *q_w_curr, double *t_w_curr, int32_t *info);
/* the batch / scan index holding the stream's newest scan: for lmono_batch_get_cloud and lmono_mapper_process (laserMapping behind it) */
int lmono_odom_stream_scan(lmono_odom_stream *, lmono_scan_batch **batch, int *scan);

/* Debug/parity view of one odometry step: correspondences of outer iteration `outer` (0/1) for the scan
 * pair (scan-1, scan) evaluated at pose q,t: corr_h [n_sharp + n_flat][4] = (a, b, c, kind).       */
int lmono_odom_correspond(lmono_ctx *, lmono_scan_batch *, int scan, const double q[4], const double t[3],
                          int32_t *corr_h, int cap);

/* ---- lmono BA factors: batched ceres::CostFunction::Evaluate -------------------------------------------- *
 * Reference interfaces (paths under /root/reference/mono_lidar_mapping), one residual block per index:
 *   kind 0 LASER   LASERFactor ctor + Evaluate            include/factor/LaserFactor.h:29-100
 *          params[14] = pose_i, pose_j (x y z qx qy qz qw); consts[24] = L0_Ri, L0_Rj (3x3 row-major), L0_Pi, L0_Pj;
 *          info[36] = LASERFactor::sqrt_info (Estimator.cc:95); r[6]; J[84] = J_i[6x7], J_j[6x7]
 *   kind 1 MONO    MonoProjectionFactor::Evaluate          src/factor/MonoProjectionFactor.cc:40-174
 *          params[22] = ex, pose_i, pose_j, inv_depth; consts[4] = pt_i.xy, pt_j.xy (normalised image points);
 *          info[4] = MonoProjectionFactor::sqrt_info (Estimator.cc:94); r[2]; J[44] = J_ex, J_i, J_j [2x7 each], J_depth[2]
 *   kind 2 PRIOR   PriorFactor ctor + Evaluate             include/factor/PriorFactor.h:29-69
 *          params[7] = ex; consts[16] = 4x4 transform row-major; info[2] = PRIOR_T, PRIOR_R; r[6]; J[42]
 *   kind 3 REPROJ  ReprojectionFactor ctor + Evaluate      include/factor/ReprojectionFactor.h:16-78
 *          params[1] = inv_depth; consts[44] = pt_i.xy, pt_j.xy, Ri, Pi, Rj, Pj, EX(4x4); info[1] = FACTOR_WEIGHT; r[2]; J[2]
 * Jacobians follow the Ceres layout (row-major, global block size, 7th pose column zero) and reproduce the reference's
 * formulae literally, including their known non-analytic blocks.  J may be NULL (residuals only).   */
#define LMONO_FACTOR_LASER  0
#define LMONO_FACTOR_MONO   1
#define LMONO_FACTOR_PRIOR  2
#define LMONO_FACTOR_REPROJ 3
int lmono_factor_eval(lmono_ctx *, int kind, int count, const double *params_h, const double *consts_h,
                      const double *info_h, double *r_h, double *J_h);
int lmono_factor_eval_d(lmono_ctx *, int kind, int count, const double *params_d, const double *consts_d,
                        const double *info_d, double *r_d, double *J_d);
/* The same with ceres::CostFunction::Evaluate's per-block contract (LaserFactor.h:45, MonoProjectionFactor.cc:40: any jacobians[k]
 * may be NULL): block_mask [count] holds one byte per residual block, bit k set = the Jacobian of parameter block k is wanted
 * (LASER: pose_i, pose_j; MONO: ex, pose_i, pose_j, inv_depth; PRIOR: ex; REPROJ: inv_depth).  Blocks whose bit is clear are left
 * untouched in J; a zero byte is `jacobians == NULL` for that residual block.  block_mask NULL = every block.                     */
int lmono_factor_eval_blocks(lmono_ctx *, int kind, int count, const double *params_h, const double *consts_h,
                             const double *info_h, double *r_h, double *J_h, const unsigned char *block_mask_h);
int lmono_factor_eval_blocks_d(lmono_ctx *, int kind, int count, const double *params_d, const double *consts_d,
                               const double *info_d, double *r_d, double *J_d, const unsigned char *block_mask_d);

/* ---- lmono sliding-window BA: Estimator::optimization()'s solve, batched over independent windows ------------ *
 * Reference interface: Estimator::optimization() -> ceres::Solve(DENSE_SCHUR, DOGLEG, max_num_iterations = NUM_ITERATIONS)
 * (mono_lidar_mapping/src/image_process/Estimator.cc:1124-1305) over para_pose[11][7], para_ex[1][7],
 * para_depth_inv[F][1] (Estimator.h:255-257) with PriorFactor / LASERFactor / MonoProjectionFactor+CauchyLoss(1).
 * A window depends on its predecessor, so a batch holds windows of independent sequences.  All arrays are host
 * pointers; lmono_ba_batch_create copies them into HBM once, lmono_ba_solve runs one workgroup per window.
 * Supported maximum: LMONO_BA_MAX_FEATURES = 1664 inverse-depth blocks per window (the reference sizes para_depth_inv[10000],
 * Estimator.h:256; its tracker caps a frame at MAX_CNT = 150 tracks, FeatureTracker.cc:21, so the 11 frames of a window can hold at
 * most 1650 tracks, of which the ones tracked >= TRACK_CNT frames enter a solve -- 100-300 on the S2 streams, SURVEY.md section 8:
 * the bound is above anything the reference's own front end can produce).  Up to 448 features the per-feature vectors of the dogleg step
 * (scale, D, gs, gn, va, vb, H_ff, g_f: 8 doubles per feature) live in LDS beside the 72 x 72 reduced system; a batch with a larger
 * window runs a second instantiation of the kernels that keeps them in an L2 scratch (same arithmetic in the same order, slower per
 * iteration).  A window above the bound is refused with LMONO_ECAPACITY by lmono_ba_batch_create / _update (and by the per-track
 * calls: lmono_triangulate, lmono_depth_refine, lmono_outlier_scores, lmono_shift_depth*) -- nothing is truncated.               */
#define LMONO_BA_MAX_FEATURES 1664
typedef struct lmono_ba_batch lmono_ba_batch;
typedef struct {
    int n_windows;
    const int *feat_off;        /* [n_windows+1] features (inverse-depth blocks) of each window                     */
    const int *obs_off;         /* [n_windows+1] MonoProjectionFactor blocks of each window                         */
    const int *flags;           /* [n_windows][4] n_poses (= frame_count+1 <= 11), use_prior, ex_constant, use_mono   */
    const double *poses;        /* [n_windows][11][7] para_pose, x y z qx qy qz qw                                  */
    const double *ex;           /* [n_windows][7] para_ex                                                           */
    const double *inv_depth;    /* [feat_off[n]] para_depth_inv                                                     */
    const int *obs_feat;        /* [obs_off[n]] window-local feature index; blocks grouped by feature, ascending     */
    const int *obs_i, *obs_j;   /* anchor frame (start_frame) and observing frame of each block                     */
    const double *obs_pts;      /* [obs_off[n]][4] pt_i.xy, pt_j.xy (normalised image coordinates)                 */
    const double *laser_consts; /* [n_windows][10][24] L0_Ri, L0_Rj, L0_Pi, L0_Pj of consecutive frames             */
    const double *prior_T;      /* [n_windows][16] prior_trans = TLC[0] at solve time (Estimator.cc:1155-1160)       */
    const double *laser_info;   /* [36] LASERFactor::sqrt_info; mono_info [4] MonoProjectionFactor::sqrt_info;      */
    const double *mono_info;
    const double *prior_w;      /* [2] PRIOR_T, PRIOR_R                                                             */
} lmono_ba_desc;
lmono_ba_batch *lmono_ba_batch_create(lmono_ctx *, const lmono_ba_desc *);
void            lmono_ba_batch_destroy(lmono_ba_batch *);
/* load another problem into an existing batch, reusing its device arrays (Estimator::optimization() of the next frame: the
 * reference rebuilds its ceres::Problem per call, Estimator.cc:1017; here the frame loop allocates nothing in steady state)  */
int             lmono_ba_batch_update(lmono_ctx *, lmono_ba_batch *, const lmono_ba_desc *);
int lmono_ba_solve(lmono_ctx *, lmono_ba_batch *, int max_iterations);   /* asynchronous on the context stream    */
int lmono_ba_batch_reset(lmono_ctx *, lmono_ba_batch *);                 /* restore the state given at creation   */
/* poses_h [n][11][7], ex_h [n][7], inv_depth_h [F total], summary_h [n][6] = initial_cost, final_cost, iterations,
 * termination (0 CONVERGENCE, 1 NO_CONVERGENCE, 2 FAILURE), successful steps, unsuccessful steps; any may be NULL */
int lmono_ba_batch_read(lmono_ctx *, lmono_ba_batch *, double *poses_h, double *ex_h, double *inv_depth_h, double *summary_h);
/* Diagnostic (no reference counterpart): a -DLMONO_BOUNDS build of the library checks every global access of k_ba_solve against the batch's allocation
 * and records the first one outside it instead of faulting: out4 = hits, source line of the first, its byte offset, its block.  The product build
 * answers LMONO_EINVAL.  tests/test_bounds_gpu.py builds the checked library into a scratch directory and runs the BA tests' problems through it.       */
int lmono_debug_bounds(lmono_ctx *, unsigned long long *out4);

/* ---- per-feature numerics of FeatureManager / Estimator (batched over windows; host arrays) ---------------- *
 * lmono_triangulate: FeatureManager::triangulate (src/image_process/FeatureManager.cc:75-255): linear multi-view
 *   triangulation of every track with estimated_depth <= 0 and >= track_cnt observations, then (refine_max_iter >= 0)
 *   the joint 1-D Ceres refinement with ReprojectionFactor + CauchyLoss(1) and setDepth()'s solve_flag (1 ok, 2 failed).
 *   Rs_h [n][11][9], Ps_h [n][11][3] (row-major), tlc_h [n][16]; track f of a window starts at start_frame[f] and owns the
 *   normalised points pts[obs_off[f] .. obs_off[f+1]) (first one = anchor); depth_h in/out.
 * lmono_outlier_scores: Estimator::outliersRejection's statistic FACTOR_WEIGHT * mean reprojection error
 *   (src/image_process/Estimator.cc:104-190); -1 for tracks shorter than track_cnt.
 * lmono_shift_depth: FeatureManager::removeBackShiftDepth as called by Estimator::slideWindowOld
 *   (FeatureManager.cc:540-590, Estimator.cc:744-763) for the tracks anchored at the dropped frame.               */
int lmono_triangulate(lmono_ctx *, int n_windows, const int *feat_off_h, const double *Rs_h, const double *Ps_h, const double *tlc_h,
                      const int *start_frame_h, const int *obs_off_h, const double *pts_h, double *depth_h, int *solve_flag_h,
                      int track_cnt, int window_size, double factor_weight, int refine_max_iter);
int lmono_outlier_scores(lmono_ctx *, int n_windows, const int *feat_off_h, const double *Rs_h, const double *Ps_h, const double *tlc_h,
                         const int *start_frame_h, const int *obs_off_h, const double *pts_h, const double *depth_h,
                         int track_cnt, double factor_weight, double *score_h);
int lmono_shift_depth(lmono_ctx *, const double *back_R0, const double *back_P0, const double *R1, const double *P1, const double *tlc,
                      int n, const double *pt_i_h, const double *depth_h, double *depth_out_h);
/* The same for n_windows independent Estimators in one call (EstimatorBatch: N sequences stepped in lock-step, SURVEY.md 8e "parallel only across
 * independent sequences"): frames_h [n_windows][40] = back_R0 (9), back_P0 (3), R1 (9), P1 (3), TLC (16, row-major 4 x 4) of each window;
 * track_off_h [n_windows + 1] offsets of the windows' tracks in pt_i_h / depth_h / depth_out_h.  A window's results are the single call's, bit for bit. */
int lmono_shift_depth_batch(lmono_ctx *, int n_windows, const double *frames_h, const int *track_off_h,
                            const double *pt_i_h, const double *depth_h, double *depth_out_h);

/* ---- marginalisation prior: Estimator::margin(), MARGIN_OLD branch first ------------------------------------ *
 * Reference interfaces: Estimator::margin (src/image_process/Estimator.cc:1307-1405), MarginalizationInfo::
 * {preMarginalize, marginalize} and Marginalization::Evaluate (src/factor/MarginalizationFactor.cc:109-131, :176-272,
 * :309-373).  Factors: LASERFactor(pose0, pose1) and one MonoProjectionFactor + CauchyLoss(1) per observation of the
 * tracks anchored at frame 0 (observations grouped by track; obs_j in 1..10; obs_pts = pt_i.xy, pt_j.xy -- the
 * reference passes the never-initialised right_pt here).  Kept blocks, in this order: ex, pose1 .. pose10 (n = 66).
 * lin_J_h [n_windows][66*66] = linearized_jacobians, lin_r_h [n_windows][66] = linearized_residuals (defined up to an
 * orthogonal row transform: compare J^T J and J^T r); status_h bit 0: H_mm needed the eps = 1e-8 cut; bit 1: the QL iteration of the eigen-decomposition hit its 60-sweep cap on an eigenvalue (never seen).
 * lmono_marg_evaluate: residual = r0 + J dx with x0_h / x_h [n_windows][11][7] (ex, pose1..pose10).               */
int lmono_marginalize(lmono_ctx *, int n_windows, const int *feat_off_h, const int *obs_off_h, const double *poses_h, const double *ex_h,
                      const double *inv_depth_h, const int *obs_feat_h, const int *obs_j_h, const double *obs_pts_h,
                      const double *laser01_h, const double *laser_info_h, const double *mono_info_h,
                      double *lin_J_h, double *lin_r_h, int *status_h);
int lmono_marg_evaluate(lmono_ctx *, int n_windows, const double *lin_J_h, const double *lin_r_h, const double *x0_h, const double *x_h,
                        double *residual_h);
/* The MARGIN_SECOND_NEW branch of Estimator::margin() (Estimator.cc:1406-1470): the only factor is the previous prior itself
 * (`Marginalization(last_marginalization_info)`, MarginalizationFactor.cc:300-373) over its n_blocks parameter blocks (7 doubles
 * each, 6 local; <= 11), evaluated at the current values x_h [n_windows][n_blocks][7] (linearisation point x0_h), and the block
 * drop_block (the one aliasing para_pose[WINDOW_SIZE - 1]) is eliminated: H = J^T J, b = J^T r, eigen pseudo-inverse of the 6x6
 * H_mm with the eps = 1e-8 cut, Schur complement, second eigen-decomposition.  lin_J_h [n_windows][n0*n0], lin_r_h [n_windows][n0],
 * n0 = 6 n_blocks; outputs over the kept blocks in their old order: lin_J_out_h [n_windows][n*n], lin_r_out_h [n_windows][n],
 * n = n0 - 6; the new linearisation point is x_h without the dropped block.  status_h bit 0: H_mm needed the eps cut.          */
int lmono_marg_second_new(lmono_ctx *, int n_windows, int n_blocks, int drop_block, const double *lin_J_h, const double *lin_r_h,
                          const double *x0_h, const double *x_h, double *lin_J_out_h, double *lin_r_out_h, int *status_h);

/* ---- laserMapping, optimisation step -------------------------------------------------------------------
 * Replaces the `for iterCount < 2 { 5-NN in the corner / surf map kd-trees; PCA line test -> LidarEdgeFactor; 5-point
 * plane fit -> LidarPlaneNormFactor; ceres::Solve }` block of A-LOAM laserMapping.cpp process() (source absent from the
 * reference tree; behavioural spec SURVEY.md Appendix A.4), batched over n_streams independent sequences.
 * *_map_h: the map clouds of each stream's 5 x 5 x 3 cube neighbourhood, *_stack_h: its voxel-filtered scan clouds
 * ([total][4] float32 x y z intensity, host memory; *_off: [n_streams + 1] point offsets).  pose_qt: [n_streams][7]
 * q_w_curr (x y z w), t_w_curr -- initial guess in, refined pose out.  stats (optional): [n_streams][8] = edge blocks of
 * the two outer iterations, plane blocks of the two, LM iterations of the two, then the device time of the whole batch in
 * microseconds: grid build, optimisation (2 x [correspond + solve]).  nn_out (optional):
 * [total stack points][5] neighbour indices of the last outer iteration (corner points first; -1 = no residual block).
 * The cube-map bookkeeping and the voxel filters of process() are not behind this ABI yet.                            */
int lmono_map_refine(lmono_ctx *, int n_streams,
                     const float *corner_map_h, const int64_t *corner_map_off, const float *surf_map_h, const int64_t *surf_map_off,
                     const float *corner_stack_h, const int64_t *corner_stack_off, const float *surf_stack_h, const int64_t *surf_stack_off,
                     double *pose_qt, int32_t *stats, int32_t *nn_out);

/* pcl::VoxelGrid (cubic leaf, every field averaged) on n_clouds independent clouds in one call: laserMapping's
 * downSizeFilterCorner / downSizeFilterSurf on the scan clouds and on the cubes of the neighbourhood (A-LOAM
 * laserMapping.cpp process()).  xyzi_h: [total][4] float32, off: [n_clouds + 1], leaf_h:
 * [n_clouds] leaf size in metres (<= 65536 points per cloud).  out_h: capacity total points; out_off: [n_clouds + 1]
 * offsets of the filtered clouds (centroids in ascending cell index, the points of a cell summed in index order).  */
int lmono_voxel_filter(lmono_ctx *, int n_clouds, const float *xyzi_h, const int64_t *off, const float *leaf_h,
                       float *out_h, int64_t *out_off);

/* laserMapping with a device-resident map: the 21 x 21 x 11 array of 50 m cubes of laserMapping.cpp (laserCloudCornerArray /
 * laserCloudSurfArray) lives in HBM, the library keeps only (offset, count) per cube on the host.  One call = one
 * process() of the node for scan `scan` of a registered batch (its less-sharp / less-flat clouds are read in place):
 * transformAssociateToMap, cube shifts, VoxelGrid of the scan clouds (mapping_line_resolution / _plane_resolution),
 * the optimisation block, transformUpdate, insertion of the scan into the cubes and re-filtering of the 5 x 5 x 3
 * neighbourhood.  q_wodom / t_wodom: laserOdometry's pose of the scan; q_w_curr / t_w_curr: aft_mapped_to_init.
 * stats (optional, [8]): edge blocks of the two outer iterations, plane blocks, LM iterations, then two sizes (for traffic
 * accounting): map points of the 5 x 5 x 3 neighbourhood handed to the optimisation, points of the cubes a map update rebuilt.
 * lmono_mapper_process keeps the (offset, count) table of the cubes ON THE DEVICE and plans the map update there: the call enqueues
 * the whole frame, waits once for the refined pose and returns; the scan joins the map behind the return (stream order: the next
 * call, lmono_mapper_cube and lmono_mapper_reset see the finished map).  Consequences for the caller: stats[7] is the PREVIOUS
 * frame's update, and an update the device had to refuse (LMONO_ECAPACITY: more than 65536 points in a cube, workspace or arena
 * exhausted) is reported by the following calls on the mapper until lmono_mapper_reset -- that update did not touch the map.
 * lmono_mapper_process_batch keeps the table on the host (one planning pass for all streams, two waits per frame); a mapper may be
 * used through both, the table is converted on entry.
 * A-LOAM laserMapping.cpp process(), source absent from the reference tree.          */
typedef struct lmono_mapper lmono_mapper;
lmono_mapper *lmono_mapper_create(lmono_ctx *, float line_res, float plane_res);      /* HDL-64 launch file: 0.4, 0.8 */
void          lmono_mapper_destroy(lmono_mapper *);
int           lmono_mapper_reset(lmono_ctx *, lmono_mapper *);                        /* empty map, identity correction */
int lmono_mapper_process(lmono_ctx *, lmono_mapper *, lmono_scan_batch *, int scan, const double q_wodom[4], const double t_wodom[3],
                         double q_w_curr[4], double t_w_curr[3], int32_t *stats);
/* n independent streams advanced by one frame each, every phase one launch for all of them: mappers[n] (distinct),
 * batches[n], scans[n], q_wodom [n][4], t_wodom [n][3] -> q_w_curr [n][4], t_w_curr [n][3], stats [n][8] (optional) */
int lmono_mapper_process_batch(lmono_ctx *, int n, lmono_mapper *const *mappers, lmono_scan_batch *const *batches, const int *scans,
                               const double *q_wodom, const double *t_wodom, double *q_w_curr, double *t_w_curr, int32_t *stats);
/* cube (i, j, k) of the corner (which = 0) / surf (1) array: returns its size; copies the points when out_h != NULL */
int lmono_mapper_cube(lmono_ctx *, lmono_mapper *, int which, int i, int j, int k, float *out_h, int cap);

/* ---- colour projection of the map builder ------------------------------------------------------------
 * Replaces MapBuilder::associateToMap + MapBuilder::depthFill (mono_lidar_mapping/src/map_builder/Map_Builder.cc:213-403)
 * and the accumulation of MapBuilder::processMapping (:8-90): a LiDAR scan is moved into the camera frame (the
 * pcl::transformPointCloud of map_build_node.cc:216-225, passed as `transform`), projected with camodocal's pinhole model
 * (camera_models/src/camera_models/PinholeCamera.cc:450-545), splatted into an 8-bit depth image (100 - z), hole-filled
 * (dilate / close / dilate 7 + fill / median 5 / bilateral or Gaussian), and every pixel with 0 < depth < 70 m is lifted
 * back to a coloured 3-D point, in row-major pixel order, in the camera frame (topic rgb_points) and in the world frame
 * (q_wc, t_wc = the Q, T arguments of associateToMap), the latter appended to the device-resident rgb_map.
 * The display products (HSV overlay :244, JET heat map :255) are not produced.                                          */
typedef struct {
    int width, height;                       /* image size (config image_width / image_height)                         */
    double fx, fy, cx, cy, k1, k2, p1, p2;   /* PINHOLE projection_parameters / distortion_parameters of the cam yaml   */
    int kernel_size;                         /* kernel_size (odd, <= 11)                                                */
    int kernel_type;                         /* kernel_type: 0 "FULL" (rect), 1 "CROSS", 2 anything else (ellipse)       */
    int blur_type;                           /* blur_type: 0 "bilateral", 1 anything else (Gaussian)                     */
} lmono_camera;
typedef struct { float x, y, z; uint32_t bgra; } lmono_point_rgb;   /* pcl::PointXYZRGB payload: b | g << 8 | r << 16 | 255 << 24 */
typedef struct lmono_map_builder lmono_map_builder;
/* max_cloud_points: largest scan of the host-buffer entry; map_capacity_points: capacity of rgb_map (>= width * height) */
lmono_map_builder *lmono_map_builder_create(lmono_ctx *, const lmono_camera *, int max_cloud_points, int64_t map_capacity_points);
void               lmono_map_builder_destroy(lmono_map_builder *);
/* One associateToMap: xyzi_h [n_points][4] float32 scan (LiDAR frame), transform: row-major 4 x 4 LiDAR -> camera,
 * bgr_h: [height][width][3] uint8 (cv::Mat BGR8), q_wc (x y z w) / t_wc: camera pose.  n_out: size of the coloured cloud. */
int lmono_associate_to_map(lmono_ctx *, lmono_map_builder *, const float *xyzi_h, int n_points, const double transform[16],
                           const uint8_t *bgr_h, const double q_wc[4], const double t_wc[3], int *n_out);
/* n_streams independent map builders advanced by one frame each, scans and images already resident in HBM (xyzi_d[s],
 * bgr_d[s] device pointers; transforms [n][16], q_wc [n][4], t_wc [n][3], n_out [n] host arrays); four launches in all. */
int lmono_associate_to_map_batch(lmono_ctx *, int n_streams, lmono_map_builder *const *mbs, const float *const *xyzi_d, const int *n_points,
                                 const double *transforms, const uint8_t *const *bgr_d, const double *q_wc, const double *t_wc, int *n_out);
int lmono_map_builder_depth(lmono_ctx *, lmono_map_builder *, uint8_t *depth_h);       /* filled depth map of the last frame */
/* coloured cloud of the last frame: which = 0 camera frame, 1 world frame; returns its size (copies when out_h != NULL)  */
int lmono_map_builder_cloud(lmono_ctx *, lmono_map_builder *, int which, lmono_point_rgb *out_h, int cap);
int64_t lmono_map_builder_map(lmono_ctx *, lmono_map_builder *, lmono_point_rgb *out_h, int64_t cap);   /* rgb_map; returns its size */
int lmono_map_builder_clear(lmono_ctx *, lmono_map_builder *);                         /* rgb_map->clear()                    */

/* ---- loop-closure pose graph -- NEW FEATURE, no counterpart in the reference --------------------------
 * The reference detects loops and publishes loop_info = relative_t, relative_q (w x y z), relative_yaw
 * (mono_lidar_mapping/src/loop_detection/KeyFrame.cc:570-633) and re-anchors the window rigidly (Estimator.cc:309-365); it never
 * optimises a graph and only carries the unused 4-DoF helpers of include/loop_detection/Loop_Detector.h:99-168.  This entry is
 * the 4-DoF (yaw, x, y, z; degrees) keyframe graph those helpers belong to: odometry edges to the four previous keyframes,
 * one Huber(0.1) edge per loop (yaw residual / 10), keyframe 0 fixed, Ceres-style Levenberg-Marquardt with a block-banded
 * Cholesky in reverse Cuthill-McKee order.
 * poses_tq_h: [n][7] keyframe poses t (x y z), q (x y z w); loops_h: [n_loops][2] (old keyframe, current keyframe);
 * loop_info_h: [n_loops][8] in the layout above.
 * Multi-GPU: every rank creates the same graph; per round each rank calls lmono_pose_graph_linearise(rank, world) -- the
 * normal equations of the edges it owns --, the fp64 buffer lmono_pose_graph_reduce_buffer (reduce_count doubles) is summed
 * over ranks with one all-reduce (RCCL), and lmono_pose_graph_step takes the identical trust-region step on every rank.   */
typedef struct lmono_pose_graph lmono_pose_graph;
lmono_pose_graph *lmono_pose_graph_create(lmono_ctx *, int n, const double *poses_tq_h, int n_loops, const int32_t *loops_h, const double *loop_info_h);
void              lmono_pose_graph_destroy(lmono_pose_graph *);
int   lmono_pose_graph_reset(lmono_ctx *, lmono_pose_graph *);                    /* back to the odometry poses, same graph */
int   lmono_pose_graph_info(lmono_pose_graph *, int64_t *reduce_count, int *bandwidth_blocks, int *n_edges);
void *lmono_pose_graph_reduce_buffer(lmono_pose_graph *);                         /* device pointer, reduce_count doubles */
/* use a caller-owned device buffer of reduce_count doubles instead (e.g. the storage of the tensor handed to the all-reduce) */
int   lmono_pose_graph_set_reduce_buffer(lmono_pose_graph *, void *buffer_d);
int   lmono_pose_graph_linearise(lmono_ctx *, lmono_pose_graph *, int rank, int world);
int   lmono_pose_graph_step(lmono_ctx *, lmono_pose_graph *, int max_iter, int *done);   /* done (optional) synchronises   */
int   lmono_pose_graph_optimize(lmono_ctx *, lmono_pose_graph *, int max_iter);          /* one GPU: the loop of the two   */
/* poses_tq_h: [n][7] optimised keyframes; stats (optional, [6]): LM iterations, initial cost, final cost, half bandwidth
 * (blocks), accepted steps, rejected steps */
int   lmono_pose_graph_result(lmono_ctx *, lmono_pose_graph *, double *poses_tq_h, double *stats);

/* ---- pose composition (laserOdometry: t_w_curr += q_w_curr * t_last_curr; q_w_curr *= q_last_curr) ------- *
 * lmono_pose_prefix_d: poses_d[k - first] = incr[first] (+) ... (+) incr[k] for k in [first, n) (incr[0] is the
 * identity: first = 0 gives poses relative to scan 0, first > 0 poses relative to scan first-1).  lmono_pose_rebase_d: poses[k] <- bases[0] (+) ... (+) bases[n_bases-1] (+) poses[k]; with scans
 * sharded over GPUs, bases are the cumulative transforms of the lower ranks (one RCCL all-gather of 56 B/rank). */
int lmono_pose_prefix_d(lmono_ctx *, const double *incr_d, int first, int n, double *poses_d);
int lmono_pose_rebase_d(lmono_ctx *, const double *bases_d, int n_bases, double *poses_d, int n);

/* Device-time accounting with hipEvents recorded on the context stream around every kernel group of every
 * lmono_scanreg_batch / lmono_odom_batch call since the last reset.  ms_out[0..8] = summed milliseconds of:
 * [0] front end total, [1] odometry total, [2] k_ring_sort, [3] k_curvature, [4] k_select, [5] k_voxel,
 * [6] k_compact, [7] k_grid_build, [8] k_line_index, [9] all k_correspond launches, [10] all k_lm_solve launches,
 * [11] number of k_correspond (= k_lm_solve) launches.  Both calls synchronise the stream.                 */
int lmono_timing_reset(lmono_ctx *);
int lmono_timing_read(lmono_ctx *, double *ms_out, int cap, int *n_scanreg_calls, int *n_odom_calls);

#ifdef __cplusplus
}
#endif
#endif
