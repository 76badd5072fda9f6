/*
 * oracle/lo_oracle.h -- TEST INFRASTRUCTURE.  CPU restatement ("oracle") of the
 * lmono per-scan hot path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may build, link, import or call anything in oracle/.
 *
 * PARITY UNPINNED: the reference tree holds no source for the A-LOAM half
 * (Aloam/ is an empty, un-pinned submodule, /root/reference/.gitmodules:1-3)
 * and no tests / golden vectors for the lmono half.  The LiDAR functions here
 * restate the public HKUST-Aerial-Robotics/A-LOAM algorithm (SURVEY.md
 * Appendix A) that tops666/Aloam forks; the BA functions restate the in-tree
 * factors line by line (file:line cited at each function).
 */
#ifndef LO_ORACLE_H
#define LO_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y, z, i; } lo_pt; /* pcl::PointXYZI payload */

#define LO_MAX_RINGS 64

/* ---- scanRegistration (A-LOAM scanRegistration.cpp laserCloudHandler) ---- */
typedef struct {
    int n_cloud;                 /* ring-major cloud size after discards            */
    int ring_begin[LO_MAX_RINGS + 1]; /* ring r occupies [ring_begin[r], ring_begin[r+1]) */
    int scan_start[LO_MAX_RINGS];     /* ring_begin+5  (scanStartInd)                */
    int scan_end[LO_MAX_RINGS];       /* ring_end-6    (scanEndInd)                  */
    int n_sharp, n_less_sharp, n_flat, n_less_flat;
} lo_scanreg_info;

/* Buffers are caller-owned with capacity n (points) each:
 * cloud[n], curvature[n], label[n] (int32), sharp/less_sharp/flat/less_flat[n]. */
int lo_scanreg(const float *xyzi, int n, int n_scans, float min_range,
               lo_pt *cloud, float *curvature, int32_t *label,
               lo_pt *sharp, lo_pt *less_sharp, lo_pt *flat, lo_pt *less_flat,
               lo_scanreg_info *info);

/* ---- exact 1-NN (stands in for pcl::KdTreeFLANN::nearestKSearch(k=1)) ---- */
typedef struct lo_kdtree lo_kdtree;
lo_kdtree *lo_kdtree_build(const lo_pt *pts, int n);
void lo_kdtree_free(lo_kdtree *t);
/* returns index of nearest point (ties: lowest index), -1 if n==0; *d2 = float squared distance */
int lo_kdtree_nn(const lo_kdtree *t, float qx, float qy, float qz, float *d2);
int lo_brute_nn(const lo_pt *pts, int n, float qx, float qy, float qz, float *d2);

/* k nearest neighbours in ascending (distance, index) order; returns the number found (< k only when n < k) */
int lo_kdtree_knn(const lo_kdtree *t, float qx, float qy, float qz, int k, int *idx, float *d2);

/* pcl::VoxelGrid restated (cubic leaf, inv_leaf = 1.0f / leaf); out has capacity n; returns the output size */
int lo_voxel_filter(const lo_pt *in, int n, float inv_leaf, lo_pt *out);

/* residual block of the LiDAR solves: kind 1 edge (a, b = line points), 2 plane (a = point j, b = unit normal),
 * 3 plane-norm of laserMapping (b = unit normal, a[0] = negative_OA_dot_norm) */
typedef struct { int kind; float cp[3]; double a[3], b[3]; } lo_corr;
/* ceres::Solve restated (trust-region LM, <= 4 iterations, Huber 0.1); x = q(xyzw), t in/out; returns iterations */
int lo_lm_solve(const lo_corr *cs, int nc, double x[7], double *cost0, double *cost1);

/* ---- laserOdometry (A-LOAM laserOdometry.cpp main loop) ---- */
typedef struct {
    int n_corner_corr[2];   /* correspondences per outer iteration */
    int n_plane_corr[2];
    int lm_iters[2];        /* LM iterations executed (<=4)        */
    double initial_cost[2], final_cost[2];
} lo_odom_stats;

/* One scan-to-scan step.  q = (x,y,z,w) and t are in/out (warm start = last increment).
 * use_kdtree=0 selects brute force NN (validation of the kd-tree).
 * corr_out (optional, may be NULL): int32[2][(n_sharp + n_flat) * 4] correspondence indices
 *   per outer iteration: edge -> (a, b, -1, 1), plane -> (a, b, c, 2), none -> (-1,-1,-1,0). */
int lo_odom_step(const lo_pt *sharp, int n_sharp, const lo_pt *flat, int n_flat,
                 const lo_pt *corner_last, int n_corner_last,
                 const lo_pt *surf_last, int n_surf_last,
                 double q[4], double t[3], int use_kdtree,
                 lo_odom_stats *stats, int32_t *corr_out);

/* accumulate: t_w += q_w * t ; q_w = q_w * q  (quaternions xyzw) */
void lo_pose_accumulate(double q_w[4], double t_w[3], const double q[4], const double t[3]);

/* ---- synthetic HDL-64 generator S1 ---- */




/* ---- laserMapping (A-LOAM laserMapping.cpp process()) ---- */
typedef struct lo_map lo_map;
typedef struct {
    int n_corner_stack, n_surf_stack;     /* down-sampled scan clouds                         */
    int n_corner_map, n_surf_map;         /* map points in the 5 x 5 x 3 cube neighbourhood   */
    int n_edge[2], n_plane[2];            /* residual blocks per outer iteration              */
    int lm_iters[2];
    double final_cost[2];
} lo_map_stats;
lo_map *lo_map_create(float line_res, float plane_res);     /* HDL-64 launch: 0.4, 0.8 */
void lo_map_free(lo_map *);
/* One frame: corner_last / surf_last = the scan's less-sharp / less-flat clouds (sensor frame), q_wodom / t_wodom =
 * laserOdometry's pose of the scan.  Writes the refined pose (aft_mapped_to_init) and updates the map. */
int lo_map_process(lo_map *, const lo_pt *corner_last, int n_corner, const lo_pt *surf_last, int n_surf,
                   const double q_wodom[4], const double t_wodom[3], double q_w_curr[4], double t_w_curr[3], lo_map_stats *);
/* the optimisation part of one frame on explicit clouds (map clouds of the cube neighbourhood, down-sampled scan clouds);
 * x = q(xyzw), t in/out; corr_out / n_corr_out (optional): residual blocks of the last outer iteration */
int lo_map_refine(const lo_pt *cmap, int n_cmap, const lo_pt *smap, int n_smap, const lo_pt *cstack, int n_cs,
                  const lo_pt *sstack, int n_ss, double x[7], lo_map_stats *st, lo_corr *corr_out, int *n_corr_out);
/* introspection for the tests: cube (i, j, k) of the 21 x 21 x 11 arrays; which = 0 corner, 1 surf */
int lo_map_cube(const lo_map *, int which, int i, int j, int k, const lo_pt **pts);
void lo_map_centre(const lo_map *, int cen[3]);
int lo_run_mapping(const float *xyzi, const int64_t *offsets, int n_scans, int n_lines, float min_range,
                   float line_res, float plane_res, int threads, const double *poses_odom, double *poses_mapped,
                   lo_map_stats *stats, double *stage_ms);
/* building blocks (exposed for the tests): ascending eigen-decomposition of a symmetric 3x3, and the 5-point plane fit */
void lo_sym_eig3(const double A[9], double evals[3], double evecs[9] /* columns = eigenvectors */);
int lo_plane_fit5(const double pts[15], double norm[3], double *negative_OA_dot_norm);

/* ---- colour projection (MapBuilder::associateToMap + depthFill, mono_lidar_mapping/src/map_builder/Map_Builder.cc:213-403;
 * SURVEY row 8f-3).  The morphology / median / bilateral / Gaussian stages restate the documented OpenCV definitions
 * (OpenCV is an un-vendored dependency of the reference and absent here: those stages are UNPINNED). ---- */
typedef struct {
    int width, height;
    double fx, fy, cx, cy, k1, k2, p1, p2;   /* camodocal PINHOLE parameters (config/kitti00_cam.yaml)              */
    int kernel_size;                         /* KERNEL_SIZE (odd)                                                  */
    int kernel_type;                         /* KERNEL_TYPE: 0 FULL (rect), 1 CROSS, 2 anything else (ellipse)      */
    int blur_type;                           /* BLUR_TYPE: 0 bilateral, 1 anything else (Gaussian)                  */
} lo_cam;
typedef struct { float x, y, z; uint32_t bgra; } lo_pt_rgb;   /* pcl::PointXYZRGB payload: b | g << 8 | r << 16 | a << 24 */

void lo_structuring_element(int type, int k, uint8_t *mask /* k x k */);
/* op 0 dilate, 1 erode (cv::dilate / cv::erode with the default border: outside pixels do not take part) */
void lo_morph(const uint8_t *src, uint8_t *dst, int w, int h, const uint8_t *mask, int k, int op);
void lo_median5(const uint8_t *src, uint8_t *dst, int w, int h);                 /* cv::medianBlur(.., 5)               */
void lo_bilateral5(const uint8_t *src, uint8_t *dst, int w, int h, double sigma_color, double sigma_space); /* d = 5     */
void lo_gauss5(const uint8_t *src, uint8_t *dst, int w, int h);                  /* cv::GaussianBlur(.., (5,5), 0)      */
void lo_depth_fill(const lo_cam *, uint8_t *depth /* in/out */);
/* transform (row-major 4x4, pcl::transformPointCloud) + projection + 8-bit depth splat; depth must be zeroed by the caller */
void lo_depth_splat(const lo_cam *, const float *xyzi, int n, const double M[16], uint8_t *depth);
int lo_backproject(const lo_cam *, const uint8_t *depth, const uint8_t *bgr, lo_pt_rgb *out);
void lo_transform_rgb(const lo_pt_rgb *in, int n, const double q[4] /* xyzw */, const double t[3], lo_pt_rgb *out);
/* whole associateToMap: depth_out (w*h, optional) = the filled depth map, cam_out / world_out capacity w*h; returns the cloud size */
int lo_associate_to_map(const lo_cam *, const float *xyzi, int n, const double M[16], const uint8_t *bgr,
                        const double q[4], const double t[3], uint8_t *depth_out, lo_pt_rgb *cam_out, lo_pt_rgb *world_out);
void lo_lift_projective(const lo_cam *, double u, double v, double ray[3]);
int lo_space_to_plane(const lo_cam *, const double P[3], double p[2]);

/* ---- loop-closure pose graph: NEW FEATURE, no parity target -- see lo_posegraph.c ---- */
void lo_pg_q2ypr(const double q_xyzw[4], double ypr_deg[3]);     /* mathutils::R2ypr, math_utils.h:187-202 */
void lo_pg_ypr2q(const double ypr_deg[3], double q_xyzw[4]);     /* YawPitchRollToRotationMatrix, Loop_Detector.h:129-147 */
typedef struct lo_pg lo_pg;
lo_pg *lo_pg_create(int n, const double *poses_tq, int n_loops, const int32_t *loops, const double *loop_info, int ordering);
void lo_pg_free(lo_pg *);
int64_t lo_pg_reduce_count(const lo_pg *);
int lo_pg_bandwidth(const lo_pg *);
void lo_pg_linearise(lo_pg *, int rank, int world, double *buf);       /* this rank's [H band | g | cost] */
int lo_pg_step(lo_pg *, const double *summed_buf, int max_iter);        /* 1 when finished */
void lo_pg_result(const lo_pg *, double *out_tq, double *stats);
int lo_pose_graph_optimize(int n, const double *poses_tq, int n_loops, const int32_t *loops, const double *loop_info,
                           int max_iter, int ordering, double *out_tq, double *stats);

#ifdef __cplusplus
}
#endif
#endif
