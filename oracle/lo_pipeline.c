/*
 * oracle/lo_pipeline.c -- TEST INFRASTRUCTURE (CPU oracle).  PARITY UNPINNED.
 *
 * Drives lo_scanreg + lo_odom_step over a sequence of scans the way the A-LOAM nodes do
 * (scanRegistration -> laserOdometry): scan 0 only initialises the "last" clouds,
 * every later scan k yields the increment T(k-1 -> k) = (q_last_curr, t_last_curr) with the previous
 * increment as warm start, and the pose is accumulated as t_w += q_w * t ; q_w = q_w * q.
 *
 * Chain sharding: the sequence may be cut into n_chains contiguous ranges; a range that
 * starts at scan s > 0 begins `lead` scans earlier with an identity warm start and discards the increments
 * of its lead-in.  n_chains = 1, lead = 0 is the strictly sequential reference behaviour.
 */
#include "lo_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <omp.h>

static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return 1e3 * (double)ts.tv_sec + 1e-6 * (double)ts.tv_nsec;
}

typedef struct {
    lo_pt *sharp, *less_sharp, *flat, *less_flat;
    lo_scanreg_info info;
} scan_feats;

void lo_chain_bounds(int n_scans, int n_chains, int c, int *s, int *e)
{
    *s = (int)((int64_t)c * n_scans / n_chains);
    *e = (int)((int64_t)(c + 1) * n_scans / n_chains);
}

/* incr: [n_scans][7] = q(xyzw), t of T(k-1 -> k) (row 0 = identity);  poses: [n_scans][7] accumulated.
 * feat_counts: [n_scans][4] (sharp, less_sharp, flat, less_flat), may be NULL.
 * stage_ms: [2] wall-clock scanreg, odometry.  threads: OpenMP threads over scans / chains. */
int lo_run_sequence(const float *xyzi, const int64_t *offsets, int n_scans, int n_lines, float min_range,
                    int n_chains, int lead, int use_kdtree, int threads,
                    double *incr, double *poses, int32_t *feat_counts, double *stage_ms)
{
    if (n_scans <= 0) return 0;
    if (n_chains < 1) n_chains = 1;
    if (n_chains > n_scans) n_chains = n_scans;
    if (threads < 1) threads = 1;
    scan_feats *F = (scan_feats *)calloc((size_t)n_scans, sizeof(scan_feats));
    int rc = 0;
    double t0 = now_ms();
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (int s = 0; s < n_scans; s++) {
        int n = (int)(offsets[s + 1] - offsets[s]);
        size_t cap = (size_t)(n > 0 ? n : 1);
        lo_pt *cloud = (lo_pt *)malloc(sizeof(lo_pt) * cap);
        float *curv = (float *)malloc(sizeof(float) * cap);
        int32_t *label = (int32_t *)malloc(sizeof(int32_t) * cap);
        lo_pt *ls = (lo_pt *)malloc(sizeof(lo_pt) * cap), *lf = (lo_pt *)malloc(sizeof(lo_pt) * cap);
        lo_pt *sh = (lo_pt *)malloc(sizeof(lo_pt) * cap), *fl = (lo_pt *)malloc(sizeof(lo_pt) * cap);
        int r = lo_scanreg(xyzi + 4 * offsets[s], n, n_lines, min_range, cloud, curv, label, sh, ls, fl, lf, &F[s].info);
        if (r != 0) rc = r;
        F[s].sharp = sh; F[s].less_sharp = ls; F[s].flat = fl; F[s].less_flat = lf;
        free(cloud); free(curv); free(label);
        if (feat_counts) {
            feat_counts[4 * s] = F[s].info.n_sharp; feat_counts[4 * s + 1] = F[s].info.n_less_sharp;
            feat_counts[4 * s + 2] = F[s].info.n_flat; feat_counts[4 * s + 3] = F[s].info.n_less_flat;
        }
    }
    double t1 = now_ms();
    for (int k = 0; k < n_scans; k++) {
        double *r = incr + 7 * k;
        r[0] = r[1] = r[2] = 0.0; r[3] = 1.0; r[4] = r[5] = r[6] = 0.0;
    }
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (int c = 0; c < n_chains; c++) {
        int s, e;
        lo_chain_bounds(n_scans, n_chains, c, &s, &e);
        int b = s - lead; if (b < 0) b = 0;
        double q[4] = { 0, 0, 0, 1 }, t[3] = { 0, 0, 0 };
        for (int k = b + 1; k < e; k++) {
            lo_odom_step(F[k].sharp, F[k].info.n_sharp, F[k].flat, F[k].info.n_flat,
                         F[k - 1].less_sharp, F[k - 1].info.n_less_sharp,
                         F[k - 1].less_flat, F[k - 1].info.n_less_flat, q, t, use_kdtree, NULL, NULL);
            if (k >= s) { memcpy(incr + 7 * k, q, 4 * sizeof(double)); memcpy(incr + 7 * k + 4, t, 3 * sizeof(double)); }
        }
    }
    double t2 = now_ms();
    double qw[4] = { 0, 0, 0, 1 }, tw[3] = { 0, 0, 0 };
    for (int k = 0; k < n_scans; k++) {
        if (k > 0) lo_pose_accumulate(qw, tw, incr + 7 * k, incr + 7 * k + 4);
        memcpy(poses + 7 * k, qw, 4 * sizeof(double)); memcpy(poses + 7 * k + 4, tw, 3 * sizeof(double));
    }
    if (stage_ms) { stage_ms[0] = t1 - t0; stage_ms[1] = t2 - t1; }
    for (int s = 0; s < n_scans; s++) { free(F[s].sharp); free(F[s].less_sharp); free(F[s].flat); free(F[s].less_flat); }
    free(F);
    return rc;
}

/* laserMapping over a sequence: scanRegistration of every scan, then lo_map_process frame by frame with
 * the given laserOdometry poses poses_odom[n][7] (q xyzw, t).  poses_mapped: [n][7]; stats: [n] or NULL;
 * stage_ms: [2] wall-clock scanreg, mapping. */
int lo_run_mapping(const float *xyzi, const int64_t *offsets, int n_scans, int n_lines, float min_range,
                   float line_res, float plane_res, int threads, const double *poses_odom, double *poses_mapped,
                   lo_map_stats *stats, double *stage_ms)
{
    if (n_scans <= 0) return 0;
    if (threads < 1) threads = 1;
    scan_feats *F = (scan_feats *)calloc((size_t)n_scans, sizeof(scan_feats));
    int rc = 0;
    double t0 = now_ms();
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (int s = 0; s < n_scans; s++) {
        int n = (int)(offsets[s + 1] - offsets[s]);
        size_t cap = (size_t)(n > 0 ? n : 1);
        lo_pt *cloud = (lo_pt *)malloc(sizeof(lo_pt) * cap);
        float *curv = (float *)malloc(sizeof(float) * cap);
        int32_t *label = (int32_t *)malloc(sizeof(int32_t) * cap);
        lo_pt *ls = (lo_pt *)malloc(sizeof(lo_pt) * cap), *lf = (lo_pt *)malloc(sizeof(lo_pt) * cap);
        lo_pt *sh = (lo_pt *)malloc(sizeof(lo_pt) * cap), *fl = (lo_pt *)malloc(sizeof(lo_pt) * cap);
        int r = lo_scanreg(xyzi + 4 * offsets[s], n, n_lines, min_range, cloud, curv, label, sh, ls, fl, lf, &F[s].info);
        if (r != 0) rc = r;
        F[s].less_sharp = ls; F[s].less_flat = lf;
        free(cloud); free(curv); free(label); free(sh); free(fl);
    }
    double t1 = now_ms();
    lo_map *m = lo_map_create(line_res, plane_res);
    for (int k = 0; k < n_scans; k++)
        lo_map_process(m, F[k].less_sharp, F[k].info.n_less_sharp, F[k].less_flat, F[k].info.n_less_flat,
                       poses_odom + 7 * k, poses_odom + 7 * k + 4, poses_mapped + 7 * k, poses_mapped + 7 * k + 4, stats ? stats + k : NULL);
    lo_map_free(m);
    double t2 = now_ms();
    if (stage_ms) { stage_ms[0] = t1 - t0; stage_ms[1] = t2 - t1; }
    for (int s = 0; s < n_scans; s++) { free(F[s].less_sharp); free(F[s].less_flat); }
    free(F);
    return rc;
}
