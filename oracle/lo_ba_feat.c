/*
 * oracle/lo_ba_feat.c -- TEST INFRASTRUCTURE (CPU oracle).  PARITY UNPINNED.
 *
 * Per-feature numerics of lmono's FeatureManager / Estimator (paths under /root/reference/mono_lidar_mapping):
 *   src/image_process/FeatureManager.cc:75-195   triangulate(): linear multi-view triangulation   -> lo_triangulate_init
 *   src/image_process/FeatureManager.cc:197-251  triangulate(): joint 1-D Ceres depth refinement  -> lo_depth_refine
 *   src/image_process/FeatureManager.cc:38-56    setDepth(): solve_flag                            -> in lo_depth_refine
 *   src/image_process/Estimator.cc:104-190       reprojectionError / outliersRejection             -> lo_outlier_scores
 *   src/image_process/FeatureManager.cc:540-590  removeBackShiftDepth (called from Estimator.cc:744-763) -> lo_shift_depth
 * Eigen::JacobiSVD(ComputeThinV).matrixV().rightCols<1>() is restated as the eigenvector of A^T A (4x4) with the
 * smallest eigenvalue (cyclic Jacobi); only the ratio V[2]/V[3] is used, so the sign ambiguity is irrelevant.
 * The Ceres refinement (DENSE_SCHUR, default LEVENBERG_MARQUARDT, max 50 iterations, CauchyLoss(1)) is restated
 * without its 0.2 s wall-clock cap, which makes the reference's result machine dependent.
 */
#include "lo_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

void lo_reproj_factor(const double *params, const double *consts, const double *weight, double *r, double *J);
void lo_cauchy(double s, double a, double rho[3]);

static void mm3(const double *A, const double *B, double *C) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j]; }
static void mtm3(const double *A, const double *B, double *C) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) C[i * 3 + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j]; }   /* A^T B */
static void mv3(const double *A, const double *v, double *o) { for (int i = 0; i < 3; i++) o[i] = A[i * 3] * v[0] + A[i * 3 + 1] * v[1] + A[i * 3 + 2] * v[2]; }
static void mtv3(const double *A, const double *v, double *o) { for (int i = 0; i < 3; i++) o[i] = A[i] * v[0] + A[3 + i] * v[1] + A[6 + i] * v[2]; }

/* eigenvector of the symmetric 4x4 M for its smallest eigenvalue (cyclic Jacobi, fixed sweep order) */
static void smallest_eigvec4(const double *Min, double *v)
{
    double M[16], V[16];
    memcpy(M, Min, sizeof(M));
    for (int i = 0; i < 16; i++) V[i] = (i % 5 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; sweep++) {
        double off = 0;
        for (int p = 0; p < 4; p++) for (int q = p + 1; q < 4; q++) off += M[p * 4 + q] * M[p * 4 + q];
        if (off < 1e-300) break;
        for (int p = 0; p < 4; p++)
            for (int q = p + 1; q < 4; q++) {
                const double apq = M[p * 4 + q];
                if (apq == 0.0) continue;
                const double theta = (M[q * 4 + q] - M[p * 4 + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 4; k++) { const double a = M[k * 4 + p], b = M[k * 4 + q]; M[k * 4 + p] = c * a - s * b; M[k * 4 + q] = s * a + c * b; }
                for (int k = 0; k < 4; k++) { const double a = M[p * 4 + k], b = M[q * 4 + k]; M[p * 4 + k] = c * a - s * b; M[q * 4 + k] = s * a + c * b; }
                for (int k = 0; k < 4; k++) { const double a = V[k * 4 + p], b = V[k * 4 + q]; V[k * 4 + p] = c * a - s * b; V[k * 4 + q] = s * a + c * b; }
            }
    }
    int m = 0;
    for (int k = 1; k < 4; k++) if (M[k * 4 + k] < M[m * 4 + m]) m = k;
    for (int k = 0; k < 4; k++) v[k] = V[k * 4 + m];
}

/* Rs [n_frames][9], Ps [n_frames][3], tlc 4x4 row-major; feature f: start_frame[f], observations
 * pts[obs_off[f] .. obs_off[f+1]) in frames start_frame .. ; depth in/out (<= 0 = not initialised).           */
void lo_triangulate_init(const double *Rs, const double *Ps, const double *tlc, int n_feat, const int32_t *start_frame,
                         const int32_t *obs_off, const double *pts, double *depth, int track_cnt)
{
    double Rlc[9], Tlc[3];
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) Rlc[i * 3 + j] = tlc[i * 4 + j]; Tlc[i] = tlc[i * 4 + 3]; }
    for (int f = 0; f < n_feat; f++) {
        const int nobs = obs_off[f + 1] - obs_off[f];
        if (depth[f] > 0 || nobs < track_cnt) continue;
        const int i = start_frame[f];
        double t0[3], R0[9], tmp[3];
        mv3(Rs + 9 * i, Tlc, tmp); for (int k = 0; k < 3; k++) t0[k] = Ps[3 * i + k] + tmp[k];
        mm3(Rs + 9 * i, Rlc, R0);
        double AtA[16];
        memset(AtA, 0, sizeof(AtA));
        for (int o = 0; o < nobs; o++) {
            const int j = i + o;
            double t1[3], R1[9], d[3], t[3], R[9], P[12];
            mv3(Rs + 9 * j, Tlc, tmp); for (int k = 0; k < 3; k++) t1[k] = Ps[3 * j + k] + tmp[k];
            mm3(Rs + 9 * j, Rlc, R1);
            for (int k = 0; k < 3; k++) d[k] = t1[k] - t0[k];
            mtv3(R0, d, t); mtm3(R0, R1, R);
            /* P = [R^T | -R^T t] */
            double Rt_t[3];
            mtv3(R, t, Rt_t);
            for (int a = 0; a < 3; a++) { for (int b = 0; b < 3; b++) P[a * 4 + b] = R[b * 3 + a]; P[a * 4 + 3] = -Rt_t[a]; }
            const double px = pts[2 * (obs_off[f] + o)], py = pts[2 * (obs_off[f] + o) + 1];
            const double nn = sqrt(px * px + py * py + 1.0);
            const double fv[3] = { px / nn, py / nn, 1.0 / nn };
            for (int rr = 0; rr < 2; rr++) {
                double row[4];
                for (int k = 0; k < 4; k++) row[k] = fv[rr] * P[8 + k] - fv[2] * P[rr * 4 + k];
                for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) AtA[a * 4 + b] += row[a] * row[b];
            }
        }
        double v[4];
        smallest_eigvec4(AtA, v);
        const double z = v[2] / v[3];
        depth[f] = (z < 0.1) ? -1.0 : z;   /* INIT_DEPTH = -1 */
    }
}

/* 44-double constant block of ReprojectionFactor for (feature anchor i, frame j) */
static void reproj_consts(const double *Rs, const double *Ps, const double *tlc, int i, int j, const double *pt_i, const double *pt_j, double *c)
{
    c[0] = pt_i[0]; c[1] = pt_i[1]; c[2] = pt_j[0]; c[3] = pt_j[1];
    memcpy(c + 4, Rs + 9 * i, 9 * sizeof(double)); memcpy(c + 13, Ps + 3 * i, 3 * sizeof(double));
    memcpy(c + 16, Rs + 9 * j, 9 * sizeof(double)); memcpy(c + 25, Ps + 3 * j, 3 * sizeof(double));
    memcpy(c + 28, tlc, 16 * sizeof(double));
}

/* cost and per-feature (h, g) of the corrected problem; obs in frame == window_size and the anchor itself are skipped */
static double refine_eval(const double *Rs, const double *Ps, const double *tlc, int n_feat, const int32_t *start_frame, const int32_t *obs_off,
                          const double *pts, const double *x, int track_cnt, int window_size, double weight, double *h, double *g)
{
    double cost = 0;
    for (int f = 0; f < n_feat; f++) {
        if (h) { h[f] = 0; g[f] = 0; }
        const int nobs = obs_off[f + 1] - obs_off[f];
        if (nobs < track_cnt) continue;
        const int i = start_frame[f];
        for (int o = 1; o < nobs; o++) {
            const int j = i + o;
            if (j == window_size) continue;
            double c[44], r[2], J[2], rho[3];
            reproj_consts(Rs, Ps, tlc, i, j, pts + 2 * obs_off[f], pts + 2 * (obs_off[f] + o), c);
            lo_reproj_factor(x + f, c, &weight, r, h ? J : NULL);
            const double sq = r[0] * r[0] + r[1] * r[1];
            lo_cauchy(sq, 1.0, rho);
            cost += 0.5 * rho[0];
            if (h) { h[f] += rho[1] * (J[0] * J[0] + J[1] * J[1]); g[f] += rho[1] * (J[0] * r[0] + J[1] * r[1]); }
        }
    }
    return cost;
}

/* depth in/out (estimated_depth); solve_flag out (0 untouched, 1 ok, 2 failed), FeatureManager.cc:197-251 + setDepth */
void lo_depth_refine(const double *Rs, const double *Ps, const double *tlc, int n_feat, const int32_t *start_frame, const int32_t *obs_off,
                     const double *pts, double *depth, int32_t *solve_flag, int track_cnt, int window_size, double weight, int max_iter)
{
    double *x = (double *)calloc((size_t)(n_feat + 1), sizeof(double)), *cand = (double *)malloc(sizeof(double) * (size_t)(n_feat + 1));
    double *h = (double *)malloc(sizeof(double) * (size_t)(n_feat + 1)), *g = (double *)malloc(sizeof(double) * (size_t)(n_feat + 1));
    double *scale = (double *)malloc(sizeof(double) * (size_t)(n_feat + 1)), *diag = (double *)malloc(sizeof(double) * (size_t)(n_feat + 1));
    double *step = (double *)malloc(sizeof(double) * (size_t)(n_feat + 1));
    int *act = (int *)malloc(sizeof(int) * (size_t)(n_feat + 1));
    for (int f = 0; f < n_feat; f++) {
        x[f] = 1.0 / depth[f];
        /* a parameter block takes part only when at least one residual block references it */
        const int nobs = obs_off[f + 1] - obs_off[f];
        int nres = 0;
        if (nobs >= track_cnt) for (int o = 1; o < nobs; o++) if (start_frame[f] + o != window_size) nres++;
        act[f] = nres > 0;
    }
    const double function_tol = 1e-6, gradient_tol = 1e-10, parameter_tol = 1e-8, min_rel = 1e-3, min_diag = 1e-6, max_diag = 1e32;
    double radius = 1e4, dec = 2.0;
    int reuse = 0, invalid = 0, iter = 0;
    double x_cost = refine_eval(Rs, Ps, tlc, n_feat, start_frame, obs_off, pts, x, track_cnt, window_size, weight, h, g);
    double x_norm = 0, gmax = 0;
    for (int f = 0; f < n_feat; f++) if (act[f]) { x_norm += x[f] * x[f]; scale[f] = 1.0 / (1.0 + sqrt(h[f])); gmax = fmax(gmax, fabs(g[f])); }
    x_norm = sqrt(x_norm);
    if (gmax > gradient_tol) while (iter < max_iter) {
        iter++;
        double model = 0;
        int ok = 1;
        for (int f = 0; f < n_feat; f++) {
            if (!act[f]) { step[f] = 0; continue; }
            const double hs = h[f] * scale[f] * scale[f], gs = g[f] * scale[f];
            if (!reuse) { double d = hs; d = d < min_diag ? min_diag : (d > max_diag ? max_diag : d); diag[f] = d; }
            const double den = hs + diag[f] / radius;
            const double s = -gs / den;
            if (!isfinite(s)) ok = 0;
            step[f] = s;
            model += -(s * gs + 0.5 * s * hs * s);
        }
        if (!ok || !(model > 0.0)) { if (++invalid >= 5) break; radius *= 0.5; reuse = 1; continue; }
        invalid = 0;
        double sn = 0;
        for (int f = 0; f < n_feat; f++) { cand[f] = x[f] + step[f] * scale[f]; if (act[f]) sn += (cand[f] - x[f]) * (cand[f] - x[f]); }
        sn = sqrt(sn);
        const double cc = refine_eval(Rs, Ps, tlc, n_feat, start_frame, obs_off, pts, cand, track_cnt, window_size, weight, NULL, NULL);
        if (sn <= parameter_tol * (x_norm + parameter_tol)) break;
        if (fabs(x_cost - cc) <= function_tol * x_cost) break;
        const double rel = (x_cost - cc) / model;
        if (rel > min_rel) {
            memcpy(x, cand, sizeof(double) * (size_t)n_feat);
            x_norm = 0;
            for (int f = 0; f < n_feat; f++) if (act[f]) x_norm += x[f] * x[f];
            x_norm = sqrt(x_norm);
            x_cost = refine_eval(Rs, Ps, tlc, n_feat, start_frame, obs_off, pts, x, track_cnt, window_size, weight, h, g);
            const double t = 2.0 * rel - 1.0;
            double den = 1.0 - t * t * t; if (den < 1.0 / 3.0) den = 1.0 / 3.0;
            radius = radius / den; if (radius > 1e16) radius = 1e16;
            dec = 2.0; reuse = 0;
            gmax = 0;
            for (int f = 0; f < n_feat; f++) if (act[f]) gmax = fmax(gmax, fabs(g[f]));
            if (gmax <= gradient_tol) break;
        } else { radius /= dec; dec *= 2.0; reuse = 1; }
        if (radius <= 1e-32) break;
    }
    for (int f = 0; f < n_feat; f++) {
        const int nobs = obs_off[f + 1] - obs_off[f];
        solve_flag[f] = 0;
        if (nobs < track_cnt) continue;
        depth[f] = 1.0 / x[f];
        solve_flag[f] = (depth[f] < 0.1 || depth[f] > 300) ? 2 : 1;
    }
    free(x); free(cand); free(h); free(g); free(scale); free(diag); free(step); free(act);
}

/* score[f] = FACTOR_WEIGHT * mean_j ||reprojection error||, -1 for features with fewer than track_cnt observations */
void lo_outlier_scores(const double *Rs, const double *Ps, const double *tlc, int n_feat, const int32_t *start_frame, const int32_t *obs_off,
                       const double *pts, const double *depth, int track_cnt, double weight, double *score)
{
    double Rlc[9], Tlc[3];
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) Rlc[i * 3 + j] = tlc[i * 4 + j]; Tlc[i] = tlc[i * 4 + 3]; }
    for (int f = 0; f < n_feat; f++) {
        const int nobs = obs_off[f + 1] - obs_off[f];
        score[f] = -1.0;
        if (nobs < track_cnt) continue;
        const int i = start_frame[f];
        const double *pi = pts + 2 * obs_off[f];
        double err = 0; int cnt = 0;
        for (int o = 1; o < nobs; o++) {
            const int j = i + o;
            const double *pj = pts + 2 * (obs_off[f] + o);
            double pc[3] = { depth[f] * pi[0], depth[f] * pi[1], depth[f] }, pl[3], pw[3], plj[3], pcj[3], d[3];
            mv3(Rlc, pc, pl); for (int k = 0; k < 3; k++) pl[k] += Tlc[k];
            mv3(Rs + 9 * i, pl, pw); for (int k = 0; k < 3; k++) d[k] = pw[k] + Ps[3 * i + k] - Ps[3 * j + k];
            mtv3(Rs + 9 * j, d, plj); for (int k = 0; k < 3; k++) d[k] = plj[k] - Tlc[k];
            mtv3(Rlc, d, pcj);
            const double rx = pcj[0] / pcj[2] - pj[0], ry = pcj[1] / pcj[2] - pj[1];
            err += sqrt(rx * rx + ry * ry); cnt++;
        }
        score[f] = (err / cnt) * weight;
    }
}

/* removeBackShiftDepth for the features anchored at the dropped frame: marg/new camera poses built with TLC
 * (Estimator.cc:754-763); depth_out <= 0 results become INIT_DEPTH = -1 */
void lo_shift_depth(const double *back_R0, const double *back_P0, const double *R1s, const double *P1s, const double *tlc,
                    int n, const double *pt_i, const double *depth, double *depth_out)
{
    double Rlc[9], Tlc[3], R0[9], R1[9], P0[3], P1[3], tmp[3];
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) Rlc[i * 3 + j] = tlc[i * 4 + j]; Tlc[i] = tlc[i * 4 + 3]; }
    mm3(back_R0, Rlc, R0); mm3(R1s, Rlc, R1);
    mv3(back_R0, Tlc, tmp); for (int k = 0; k < 3; k++) P0[k] = back_P0[k] + tmp[k];
    mv3(R1s, Tlc, tmp); for (int k = 0; k < 3; k++) P1[k] = P1s[k] + tmp[k];
    for (int f = 0; f < n; f++) {
        const double pi[3] = { pt_i[2 * f] * depth[f], pt_i[2 * f + 1] * depth[f], depth[f] };
        double w[3], d[3], pj[3];
        mv3(R0, pi, w); for (int k = 0; k < 3; k++) d[k] = w[k] + P0[k] - P1[k];
        mtv3(R1, d, pj);
        depth_out[f] = pj[2] > 0 ? pj[2] : -1.0;
    }
}
