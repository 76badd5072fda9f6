// lmono_amd/host/lmono_host.hpp -- host-side C++ mirror of the reference call surface above the C ABI
// (include/lmono_hip.h).  Same member names, argument meaning and state layout as the reference so that a maintainer
// can swap the bodies in place:
//   Estimator      /root/reference/mono_lidar_mapping/include/image_process/Estimator.h:110-173, .cc:1019-1305, :700-771
//   FeatureManager /root/reference/mono_lidar_mapping/src/image_process/FeatureManager.cc:38-73, :75-255, :497-590
//   ScanRegistration / LaserOdometry: the A-LOAM node handlers (source absent)
// No Eigen / Ceres / ROS: plain arrays (row-major 3x3, x y z qx qy qz qw parameter blocks).  All numerics run in the
// HIP library; this file only keeps the list / window bookkeeping that the reference keeps on the host.
#pragma once
#include <cstdint>
#include <functional>
#include <future>
#include <list>
#include <memory>
#include <map>
#include <set>
#include <stdexcept>
#include <string>
#include <array>
#include <vector>
#include "../../include/lmono_hip.h"

namespace lmono_host {

constexpr int WINDOW_SIZE = 10;          // include/parameter.h:51
constexpr double INIT_DEPTH = -1.0;

struct Mat3 { double m[9]; };
struct Vec3 { double v[3]; };

struct Params {                           // config values used on the hot path (kitti_config_05.yaml)
    double FACTOR_WEIGHT = 1500.0, LASER_W = 3.0, PRIOR_T = 1000.0, PRIOR_R = 1000.0, OUTLIER_T = 5.0;
    int TRACK_CNT = 3, FINE_TIMES = 1, NUM_ITERATIONS = 30, ESTIMATE_LASER = 1;
    double FEATURE_THRESHOLD = 10.0;      // feature_threshold (pixels of parallax that make a keyframe)
};

class HipContext {
public:
    explicit HipContext(int device = 0) : ctx_(lmono_create(device)), device_(device)
    {
        if (!ctx_) throw std::runtime_error("lmono_create failed: no usable gfx950 device (there is no CPU fallback)");
    }
    ~HipContext() { lmono_destroy(ctx_); }
    lmono_ctx *get() const { return ctx_; }
    int device() const { return device_; }
    void useOwnStream() { check(lmono_use_own_stream(ctx_), "lmono_use_own_stream"); }   // a non-blocking stream of the library's
    void check(int rc, const char *what) const
    {
        if (rc < 0) throw std::runtime_error(std::string(what) + ": " + lmono_last_error(ctx_));
    }
private:
    lmono_ctx *ctx_;
    int device_;
};

// ---- FeatureManager (track store) ---------------------------------------------------------------------------------
struct FeaturePerFrame { double pt[2]; double uv[2] = { 0, 0 }; };   // normalised point, pixel (FeatureManager.h FeaturePerFrame)
struct FeaturePerId {
    int feature_id, start_frame;
    std::vector<FeaturePerFrame> feature_per_frame;
    int used_num = 0, solve_flag = 0;
    double estimated_depth = INIT_DEPTH;
    int endFrame() const { return start_frame + (int)feature_per_frame.size() - 1; }
};

class FeatureManager {
public:
    std::list<FeaturePerId> feature;
    const Params *params = nullptr;
    HipContext *hip = nullptr;

    int getFeatureCount();                                   // FeatureManager.cc: tracks with used_num >= TRACK_CNT
    std::vector<double> getDepthVector();                    // :58-73  (inverse depths)
    void setDepth(const std::vector<double> &x);             // :38-56
    void removeFailures();                                   // erase solve_flag == 2
    void removeOutlier(const std::set<int> &ids);
    void triangulate(int frameCnt, const Mat3 Rs[], const Vec3 Ps[], const double tlc[16]);   // :75-255 -> lmono_triangulate
    void removeBackShiftDepth(const Mat3 &back_R0, const Vec3 &back_P0, const Mat3 &R1, const Vec3 &P1, const double tlc[16]);  // :540-590
    // the two halves of removeBackShiftDepth around its numeric call (EstimatorBatch makes ONE lmono_shift_depth_batch call for all its streams):
    // the tracks anchored at the dropped frame that survive it (points, depths); then the list surgery with the shifted depths
    void shiftDepthPack(std::vector<double> &pt, std::vector<double> &dep) const;
    void shiftDepthApply(const double *depth_out);
    // likewise triangulate(): what setDepth-style bookkeeping follows lmono_triangulate
    void triangulateApply(const double *depth, const int *solve_flag);
    void removeBack();                                       // :497-511
    void removeFront(int frame_count);                       // :513-538
    // one observation of the tracker per feature id: x_n, y_n, u, v (the first four of the reference's 6-vector)
    typedef std::map<int, std::array<double, 4>> Image;
    bool featureCheck(int frame_count, const Image &image, double td);   // :315-400: appends the observations, true = keyframe
    double computeParallax(const FeaturePerId &it_per_id, int frame_count) const;   // :279-313
    void clearDepth();                                       // :29-36
    void clearState() { feature.clear(); }
    int last_track_num = 0, long_track_num = 0, new_feature_num = 0;
    // packs tracks with used_num >= TRACK_CNT for the kernels: start, offsets, points (anchor first)
    void pack(std::vector<int> &start, std::vector<int> &off, std::vector<double> &pts, std::vector<double> &depth, bool all_tracks);
};

// ---- what one Estimator hands to a numeric C-ABI call.  Estimator makes the call for its own pack (n_windows = 1); EstimatorBatch concatenates
// the packs of N Estimators into one call (n_windows = N) -- the numbers a window gets back are the same bytes either way.
struct TrackPack {                         // lmono_triangulate / lmono_outlier_scores: every track of the window
    std::vector<int> start, off; std::vector<double> pts, depth;
    double R[99], P[33];
};
struct SolvePack {                         // lmono_ba_batch_create / _update: the residual blocks of Estimator.cc:1155-1215
    std::vector<int> obs_feat, obs_i, obs_j; std::vector<double> obs_pts;
    int F = 0, flags[4] = { 0, 0, 0, 0 };
    bool use_mono = false;
    double poses[77], laser[240];
};
struct MargPack {                          // lmono_marginalize (kind 1, MARGIN_OLD) / lmono_marg_second_new (kind 2); kind 0: nothing to do this frame
    int kind = 0;
    std::vector<int> obs_feat, obs_j; std::vector<double> obs_pts, invd;
    int f0 = 0;
    double poses[77], ex[7], laser01[24];
    int nb = 0, drop = 0; std::vector<double> x;      // kind 2: the prior's blocks at their current values
};
struct ShiftPack { std::vector<double> pt, dep; double frames[40]; };     // lmono_shift_depth[_batch]: back_R0, back_P0, R1, P1, TLC

class EstimatorBatch;
class MarginWorker;

// ---- Estimator ------------------------------------------------------------------------------------------------------
class Estimator {
public:
    enum MarginalizationFlag { MARGIN_OLD = 0, MARGIN_SECOND_NEW = 1 };
    Estimator(HipContext &hip, const Params &p);
    ~Estimator();
    Estimator(const Estimator &) = delete;
    Estimator &operator=(const Estimator &) = delete;

    // state, same names as Estimator.h:240-272
    Mat3 Rs[WINDOW_SIZE + 1];
    Vec3 Ps[WINDOW_SIZE + 1];
    double TLC[16];                        // laser <- camera 4x4 row-major (TLC[0])
    Mat3 L0_R[WINDOW_SIZE + 1];            // all_image_frame[i].second.L0_R / L0_T
    Vec3 L0_T[WINDOW_SIZE + 1];
    double para_pose[WINDOW_SIZE + 1][7], para_ex[1][7];
    std::vector<double> para_depth_inv;
    int frame_count = 0, first_refine = 0;
    bool static_status = false;
    MarginalizationFlag marginalization_flag = MARGIN_OLD;
    FeatureManager feature_manager;
    Mat3 back_R0; Vec3 back_P0;
    double final_cost = 0, initial_cost = 0; int iterations = 0, termination = 0;

    // ---- frame loop (Estimator.cc:236-273, :309-365, :367-499, :852-1017)
    enum StageFlag { NOT_INITED = 0, INITED = 1 };
    StageFlag stage_flag = NOT_INITED;
    double Header[WINDOW_SIZE + 1] = { 0 };
    Vec3 last_laser_t = { { 0, 0, 0 } };
    bool loop_closure = false;
    struct LoopFrame { double loop_time_stamp; double old_T[3], old_Q[4], correct_T[3], correct_Q[4]; };   // quaternions w x y z (Estimator.h:60-104)
    std::vector<LoopFrame> loop_buf;
    std::vector<std::array<double, 8>> new_odometry;     // trajectory of record (:634-645): header, P, q x y z w
    // L0_Pos of processCompactData: 4x4 row-major LiDAR pose (topic LASER_ODOM_TOPIC); sets static_status
    void processCompactData(const double L0_Pos[16]);
    // one pass of processEstimation without ROS / the image tracker: `image` is FeatureTracker::trackImage's output
    bool processImage(double header, const FeatureManager::Image &image, const double transform_to_init[16]);
    bool runInitialization();
    void loopCorrection();
    void setLoopFrame(const LoopFrame &f) { loop_buf.push_back(f); }

    void matrix2Double();                  // Estimator.cc:1019-1057
    void double2Matrix();                  // :1059-1122
    bool optimization();                   // :1124-1305 (solve through lmono_ba_*; margin() is not part of the solve path)
    void outliersRejection(std::set<int> &removeIndex, const double &error);   // :134-190
    void slideWindow();                    // :700-771
    // :1307-1470, both branches.  Like the reference, the prior is computed but never fed back into optimization()
    // (MarginalizationInfo::valid is never set).  The reference passes the mono pipeline's
    // never-initialised right_pt as the second observation; the mirror passes the tracked point.
    void margin();
    // Marginalisation overlapped with the next frame (off by default = the reference's inline order).  When on, margin() packs its
    // inputs and hands the kernel call to a second context on its own HIP stream and a host thread; processImage returns without
    // waiting and the next frame's solve runs beside it.  Results are the same bytes: nothing reads the prior before the next
    // margin() (MarginalizationInfo::valid is never set), which waits for the pending one first.  Call marginWait()
    // before reading last_marginalization_info from outside.
    void setAsyncMargin(bool on);
    void marginWait();
    struct MarginalizationInfo {           // include/factor/MarginalizationFactor.h:78-108 (fields used downstream)
        int m = 0, n = 0;
        std::vector<double> linearized_jacobians, linearized_residuals;   // [n*n], [n]
        std::vector<double> keep_block_data;                               // [blocks][7] at linearisation
        // last_marginalization_parameter_blocks as window slots: -1 = para_ex[0], i = para_pose[i] (after the address shift)
        std::vector<int> parameter_blocks;
        bool valid = false, present = false;
        int status = 0;
    } last_marginalization_info;
    // ---- the host halves of the numeric steps (what runs before and behind each C-ABI call); the methods above are made of them, EstimatorBatch
    //      runs them for N streams around ONE batched call per step
    bool packTracks(TrackPack &tp);                                       // false: the window holds no track
    void packSolve(SolvePack &sp);                                        // matrix2Double + the residual blocks
    void unpackSolve(const SolvePack &sp, const double *poses77, const double *ex7, const double *invd, const double *summary6);   // ... double2Matrix
    void packMargin(MargPack &mp);                                        // margin() up to its kernel call (both branches)
    void applyMarginOld(const MargPack &mp, const double *lin_J, const double *lin_r, int status);
    void applyMarginSecond(MargPack &mp, const double *lin_J, const double *lin_r, int status);
    void applyOutlierScores(const double *score, double error, std::set<int> &removeIndex);
    // slideWindow() in two halves: the pose / header shifts and the list surgery that needs no numerics; true = removeBackShiftDepth is due
    // (sp filled); slideWindowFinish applies its result
    bool slideWindowBegin(ShiftPack &sp);
    void slideWindowFinish(const double *depth_out);
    void preFrame(double header, const FeatureManager::Image &image, const double transform_to_init[16], bool *keyframe);   // processImage up to the stage switch
    void pushOdometryRow();                                               // new_odometry row, :634-645
    void initialPoses();                                                  // runInitialization :986-1003 (poses from the LiDAR trajectory, clearDepth)
    int margin_calls[2] = { 0, 0 };        // MARGIN_OLD priors built, MARGIN_SECOND_NEW eliminations done
    // algorithmic flops of the window solves so far, SURVEY.md 8d: iterations x (2000 per projection block + 72^3 / 3) -- bench.py's roofline figure
    double solve_flops = 0.0;
    long solve_obs = 0;

private:
    HipContext &hip_;
    lmono_ba_batch *ba_batch_ = nullptr;       // the window problem's device arrays, kept from frame to frame
    std::unique_ptr<HipContext> margin_hip_;   // setAsyncMargin: the context marginalisation runs on
    // the overlapped marginalisation's worker: ONE host thread for the Estimator's lifetime takes the jobs (a thread per frame -- std::async -- cost the
    // frame loop ~70 us per frame); at most one job is pending
    std::unique_ptr<MarginWorker> margin_worker_;
    friend class EstimatorBatch;
    void marginSubmit(std::function<void()> job);
    bool async_margin_ = false;
    Params p_;
};

// ---- N independent Estimators stepped in lock-step ("BA ... parallel only across independent sequences / streams") --------
// One sequence cannot batch (window k starts from window k - 1's result); N sequences can: every numeric step of a frame -- triangulation, the window
// solve, marginalisation, outlier scores, the depth shift of the slide -- is ONE C-ABI call over the N windows of the N streams (the entry points take
// n_windows), the host halves around the calls run on a small thread pool.  Every stream's output is the bytes of its own single-stream run: the kernels
// own one window per workgroup (or one track per thread) and form their sums in an order fixed by the window alone.
// The streams must be at the same frame of their sequences (same frame_count / stage_flag): they are created together and fed one frame each per call.
class HostPool;
class EstimatorBatch {
public:
    EstimatorBatch(HipContext &hip, const Params &p, int n_streams, int host_threads = 0);     // host_threads 0: min(hardware threads, 16) (LMONO_HOST_THREADS)
    ~EstimatorBatch();
    EstimatorBatch(const EstimatorBatch &) = delete;
    EstimatorBatch &operator=(const EstimatorBatch &) = delete;
    int size() const { return (int)est_.size(); }
    Estimator &stream(int s) { return *est_[(size_t)s]; }
    // one frame of every stream: headers[n], images[n], transform_to_init[n] (4 x 4 row-major LiDAR poses); keyframe[n] (optional) as processImage returns it
    void processImage(const double *headers, const FeatureManager::Image *images, const double (*transform_to_init)[16], bool *keyframe = nullptr);
    void processImage(const double *headers, const FeatureManager::Image *const *images, const double (*transform_to_init)[16], bool *keyframe = nullptr);   // images by pointer
    // the same frame in two halves around the (asynchronous) window solve: Begin returns with the solve in flight, Finish waits for it and runs the rest of the frame.
    // One thread can interleave several batches (own contexts) that way: a batch's host passes under another batch's solve
    void processImageBegin(const double *headers, const FeatureManager::Image *const *images, const double (*transform_to_init)[16], bool *keyframe = nullptr);
    void processImageFinish();
    void setAsyncMargin(bool on);          // marginalisation of frame k beside frame k + 1 (second context, own stream, one worker thread), as Estimator::setAsyncMargin
    void marginWait();
    // called once per stream at the end of every frame, inside the frame's last per-stream pass (on a pool thread: it may touch stream s's own data only) --
    // what a node does with a stream's result (publishing, logging) without a serial loop over the streams behind the frame
    void setFrameHook(std::function<void(int stream, const Estimator &)> hook) { frame_hook_ = std::move(hook); }
private:
    std::function<void(int, const Estimator &)> frame_hook_;
    struct Work;
    void concatTracks();
    void callTriangulate();
    void applyTriangulate(int s);
    void callOutliers();
    void applyOutliers(int s, double error);
    void callSolve();
    void readSolve();
    void applySolve(int s);
    void submitMargin(std::shared_ptr<std::vector<MargPack>> packs);
    void callShift();
    HipContext &hip_;
    Params p_;
    std::vector<std::unique_ptr<Estimator>> est_;
    lmono_ba_batch *ba_batch_ = nullptr;
    std::shared_ptr<HostPool> pool_;       // shared by the EstimatorBatches of a process that ask for the same number of threads (groups driven by one thread)
    std::unique_ptr<Work> work_;
    std::unique_ptr<HipContext> margin_hip_;
    std::unique_ptr<MarginWorker> margin_worker_;
    bool async_margin_ = false;
    int pending_ = 0;                      // a frame between Begin and Finish: 1 = an INITED frame, 2 = the initialisation frame
};

// ---- A-LOAM nodes ---------------------------------------------------------------------------------------------------
class ScanRegistration {                   // scanRegistration.cpp: laserCloudHandler
public:
    ScanRegistration(HipContext &hip, int n_scans_cap, int64_t points_cap, int N_SCANS = 64, float MINIMUM_RANGE = 5.0f);
    ~ScanRegistration();
    // xyzi_d: scans already staged in HBM; runs the batch (one call per queue flush)
    void laserCloudHandler(const float *xyzi_d, const int64_t *offsets_h, int n_scans);
    // same with the scans in host memory (a decoded PointCloud2 / a KITTI .bin): staged to HBM by the library
    void laserCloudHandlerHost(const float *xyzi_h, const int64_t *offsets_h, int n_scans);
    std::vector<float> cloud(int scan, int which);          // 0 velodyne_cloud_2, 1 sharp, 2 less_sharp, 3 flat, 4 less_flat
    lmono_scan_batch *batch() const { return batch_; }
    int n_scans() const { return n_; }
private:
    HipContext &hip_; lmono_scan_batch *batch_; int N_SCANS_; float MINIMUM_RANGE_; int n_ = 0; int64_t cap_;
};

class LaserOdometry {                      // laserOdometry.cpp main loop
public:
    explicit LaserOdometry(HipContext &hip) : hip_(hip) {}
    // q_w_curr / t_w_curr of every scan: [n][7] = qx qy qz qw tx ty tz
    std::vector<double> process(ScanRegistration &reg, int n_chains = 1, int lead = 0);
private:
    HipContext &hip_;
};

// The same two nodes as they run online: one /velodyne_points message per callback.  laserCloudHandler registers the scan, process()
// runs the scan pair against the previous scan's clouds (kept on the device) from para_q / para_t, which persist between callbacks
// like A-LOAM's globals; q_w_curr / t_w_curr accumulate.  Same numbers as LaserOdometry::process(reg, 1, 0).
class LaserOdometryNode {
public:
    LaserOdometryNode(HipContext &hip, int max_points_per_scan, int n_lines = 64, float minimum_range = 5.0f, int history = 8);
    ~LaserOdometryNode();
    LaserOdometryNode(const LaserOdometryNode &) = delete;
    LaserOdometryNode &operator=(const LaserOdometryNode &) = delete;
    // xyzi: [n][4] float32 in host memory (a sensor_msgs/PointCloud2 payload, a KITTI .bin file)
    void laserCloudHandler(const float *xyzi, int n_points);
    double q_last_curr[4] = { 0, 0, 0, 1 }, t_last_curr[3] = { 0, 0, 0 };     // para_q, para_t
    double q_w_curr[4] = { 0, 0, 0, 1 }, t_w_curr[3] = { 0, 0, 0 };
    int info[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };     // n_cloud, n_sharp, n_less_sharp, n_flat, n_less_flat, status, LM iterations, residual blocks
    // the device-resident scan for laserMapping behind this node
    lmono_scan_batch *batch() const;
    int scan() const;
private:
    HipContext &hip_;
    lmono_odom_stream *stream_;
};

// laserMapping.cpp process(): the 21 x 21 x 11 cube array (laserCloudCornerArray /
// laserCloudSurfArray) lives in HBM behind lmono_mapper_*; one call per scan of a registered batch.
class LaserMapping {
public:
    LaserMapping(HipContext &hip, float lineRes = 0.4f, float planeRes = 0.8f);   // mapping_line_resolution / _plane_resolution
    ~LaserMapping();
    LaserMapping(const LaserMapping &) = delete;
    LaserMapping &operator=(const LaserMapping &) = delete;
    // q_wodom_curr (x y z w), t_wodom_curr: laserOdometry's pose of scan `scan`.  Writes q_w_curr / t_w_curr
    // (aft_mapped_to_init) and updates the map.
    void process(ScanRegistration &reg, int scan, const double q_wodom_curr[4], const double t_wodom_curr[3], double q_w_curr[4], double t_w_curr[3]);
    // the same behind the online node: the scan it has just registered
    void process(LaserOdometryNode &node, double q_w_curr[4], double t_w_curr[3]);
    // cube (i, j, k) of the corner (which = 0) / surf (1) array: [n][4] float32
    std::vector<float> cube(int which, int i, int j, int k);
    int stats[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };     // of the last frame: edge / plane blocks and LM iterations per outer iteration
private:
    HipContext &hip_;
    lmono_mapper *mapper_;
};

// MapBuilder (include/map_builder/Map_Builder.h, src/map_builder/Map_Builder.cc): colour projection of a
// LiDAR scan into the camera image and accumulation of the coloured world-frame map, on the device behind lmono_map_builder_*.
class MapBuilder {
public:
    // camera: PINHOLE intrinsics + kernel_size / kernel_type / blur_type of the map config; save_map / map_dir: SAVE_MAP and the
    // directory of "rgb_map<index>.ply" (hard-coded "/home/bo/raw_data/map" in the reference, Map_Builder.cc:75)
    MapBuilder(HipContext &hip, const lmono_camera &camera, bool save_map = false, const std::string &map_dir = ".", int max_cloud_points = 1 << 18);
    ~MapBuilder();
    MapBuilder(const MapBuilder &) = delete;
    MapBuilder &operator=(const MapBuilder &) = delete;
    // Map_Builder.cc:213 with the cloud transform of map_build_node.cc:216-225 folded in: Q (x y z w), T = camera pose,
    // xyzi = the scan in the LiDAR frame, rlc / tlc = camera-from-LiDAR extrinsic (p_l = rlc p_c + tlc), frame = BGR8 image,
    // t = stamp.  Returns the size of the coloured cloud; the world-frame cloud is queued for processMapping().
    int associateToMap(const double Q[4], const double T[3], const float *xyzi, int n_points, const double rlc[9], const double tlc[3],
                       const uint8_t *frame, double t);
    // one pass of the body of MapBuilder::processMapping (Map_Builder.cc:12-86) for the frame queued last: rgb_map += cloud,
    // map_index++, and every 10th frame the map is written (when save_map) and cleared.  Returns the path written, or "".
    std::string processMapping();
    std::vector<lmono_point_rgb> rgbCloud(int which);        // last frame: 0 camera frame (topic rgb_points), 1 world frame
    std::vector<lmono_point_rgb> rgbMap();                   // rgb_map
    std::vector<uint8_t> depthMap();                         // filled depth map of the last frame (topic depth_map before COLORMAP_JET)
    int map_index = 0;
private:
    HipContext &hip_;
    lmono_map_builder *mb_;
    lmono_camera cam_;
    bool save_map_, pending_ = false;
    std::string map_dir_;
};

// Loop-closure pose graph -- NEW FEATURE: the reference's LoopDetector declares t_optimization / optimize_buf
// (include/loop_detection/Loop_Detector.h:85-86) but never optimises.  This is the 4-DoF graph its leftover helpers
// (Loop_Detector.h:99-168) belong to, behind lmono_pose_graph_*.
class PoseGraph {
public:
    struct Loop { int old_index, cur_index; double loop_info[8]; };   // loop_info as KeyFrame::findConnection fills it (KeyFrame.cc:630-633)
    explicit PoseGraph(HipContext &hip) : hip_(hip) {}
    // keyframe poses [n][7] = t (x y z), q (x y z w) in, optimised poses out; returns the LM iterations run
    int optimize4DoF(std::vector<double> &poses_tq, const std::vector<Loop> &loops, int max_iter = 5);
    double initial_cost = 0, final_cost = 0;
    int bandwidth = 0;
private:
    HipContext &hip_;
};

void estimator_print_phase_clock();      // LMONO_HOST_TIMING=1: host-side phase times of the INITED frame loop, to stderr

} // namespace lmono_host
