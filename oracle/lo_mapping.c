/*
 * oracle/lo_mapping.c -- TEST INFRASTRUCTURE (CPU oracle).  PARITY UNPINNED.
 *
 * Restatement of A-LOAM laserMapping.cpp process() (source absent from the reference
 * tree, /root/reference/.gitmodules:1-3): a 21 x 21 x 11 array of 50 m cubes holding the down-sampled corner / surf map,
 * shifted when the sensor nears its border; per frame 2 x [5-NN of every down-sampled scan point in the map of the
 * 5 x 5 x 3 cube neighbourhood -> PCA line test / 5-point plane fit -> LidarEdgeFactor / LidarPlaneNormFactor ->
 * ceres::Solve (DENSE_QR, 4 iterations, Huber 0.1)], then the scan is added to the cubes and those are re-filtered.
 * Eigen's SelfAdjointEigenSolver is replaced by cyclic Jacobi (same eigen-pairs up to rounding and the sign of the
 * vectors, which the edge factor does not see), colPivHouseholderQr().solve() by an explicit column-pivoted
 * Householder least-squares solve.
 */
#include "lo_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MW 21
#define MH 21
#define MD 11
#define MCUBES (MW * MH * MD)

typedef struct { lo_pt *p; int n, cap; } cloudbuf;
struct lo_map {
    cloudbuf corner[MCUBES], surf[MCUBES];
    int cen_w, cen_h, cen_d;
    double q_wmap_wodom[4], t_wmap_wodom[3];
    float inv_line, inv_plane;       /* 1.0f / leaf, as pcl::VoxelGrid::setLeafSize computes it */
};

static void cb_push(cloudbuf *c, const lo_pt *p)
{
    if (c->n == c->cap) { c->cap = c->cap ? 2 * c->cap : 64; c->p = (lo_pt *)realloc(c->p, sizeof(lo_pt) * (size_t)c->cap); }
    c->p[c->n++] = *p;
}

lo_map *lo_map_create(float line_res, float plane_res)
{
    lo_map *m = (lo_map *)calloc(1, sizeof(*m));
    m->cen_w = 10; m->cen_h = 10; m->cen_d = 5;
    m->q_wmap_wodom[3] = 1.0;
    m->inv_line = 1.0f / line_res; m->inv_plane = 1.0f / plane_res;
    return m;
}
void lo_map_free(lo_map *m)
{
    if (!m) return;
    for (int i = 0; i < MCUBES; i++) { free(m->corner[i].p); free(m->surf[i].p); }
    free(m);
}
int lo_map_cube(const lo_map *m, int which, int i, int j, int k, const lo_pt **pts)
{
    const cloudbuf *c = &(which ? m->surf : m->corner)[i + MW * j + MW * MH * k];
    if (pts) *pts = c->p;
    return c->n;
}
void lo_map_centre(const lo_map *m, int cen[3]) { cen[0] = m->cen_w; cen[1] = m->cen_h; cen[2] = m->cen_d; }

static void q_rot(const double q[4], const double v[3], double o[3])
{
    /* Eigen Quaternion * Vector3: v + 2 w (u x v) + 2 u x (u x v) */
    const double ux = q[0], uy = q[1], uz = q[2], w = q[3];
    const double uvx = 2.0 * (uy * v[2] - uz * v[1]), uvy = 2.0 * (uz * v[0] - ux * v[2]), uvz = 2.0 * (ux * v[1] - uy * v[0]);
    o[0] = v[0] + w * uvx + (uy * uvz - uz * uvy);
    o[1] = v[1] + w * uvy + (uz * uvx - ux * uvz);
    o[2] = v[2] + w * uvz + (ux * uvy - uy * uvx);
}
static void q_mul(const double a[4], const double b[4], double o[4])
{
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    o[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
}

/* ascending eigenvalues / eigenvectors (columns) of a symmetric 3x3: cyclic Jacobi to machine precision */
void lo_sym_eig3(const double A[9], double evals[3], double evecs[9])
{
    double a[9], v[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    memcpy(a, A, sizeof(a));
    for (int sweep = 0; sweep < 60; sweep++) {
        const double off = a[1] * a[1] + a[2] * a[2] + a[5] * a[5];
        const double dia = a[0] * a[0] + a[4] * a[4] + a[8] * a[8];
        if (off <= 1e-40 * (dia > 0 ? dia : 1.0) || off == 0.0) break;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                const double apq = a[p * 3 + q];
                if (apq == 0.0) continue;
                const double theta = (a[q * 3 + q] - a[p * 3 + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; k++) {   /* A <- A J */
                    const double akp = a[k * 3 + p], akq = a[k * 3 + q];
                    a[k * 3 + p] = c * akp - s * akq; a[k * 3 + q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; k++) {   /* A <- J^T A */
                    const double apk = a[p * 3 + k], aqk = a[q * 3 + k];
                    a[p * 3 + k] = c * apk - s * aqk; a[q * 3 + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; k++) {
                    const double vkp = v[k * 3 + p], vkq = v[k * 3 + q];
                    v[k * 3 + p] = c * vkp - s * vkq; v[k * 3 + q] = s * vkp + c * vkq;
                }
            }
    }
    int ord[3] = { 0, 1, 2 };
    const double d[3] = { a[0], a[4], a[8] };
    for (int i = 0; i < 2; i++) for (int j = i + 1; j < 3; j++) if (d[ord[j]] < d[ord[i]]) { int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
    for (int c = 0; c < 3; c++) { evals[c] = d[ord[c]]; for (int k = 0; k < 3; k++) evecs[k * 3 + c] = v[k * 3 + ord[c]]; }
}

/* norm = argmin |A norm + 1| over the 5 rows of A (matA0.colPivHouseholderQr().solve(matB0), matB0 = -1); returns
 * plane validity (every point within 0.2 of the plane).  pts: 5 x 3 row-major. */
int lo_plane_fit5(const double pts[15], double norm[3], double *negative_OA_dot_norm)
{
    double A[15], b[5] = { -1, -1, -1, -1, -1 };
    memcpy(A, pts, sizeof(A));
    int perm[3] = { 0, 1, 2 };
    for (int k = 0; k < 3; k++) {
        /* column pivoting: largest remaining column norm */
        int piv = k; double best = -1.0;
        for (int c = k; c < 3; c++) { double s = 0; for (int r = k; r < 5; r++) s += A[r * 3 + c] * A[r * 3 + c]; if (s > best) { best = s; piv = c; } }
        if (piv != k) { for (int r = 0; r < 5; r++) { double t = A[r * 3 + k]; A[r * 3 + k] = A[r * 3 + piv]; A[r * 3 + piv] = t; } int t = perm[k]; perm[k] = perm[piv]; perm[piv] = t; }
        /* Householder reflector for column k */
        double nrm = 0; for (int r = k; r < 5; r++) nrm += A[r * 3 + k] * A[r * 3 + k];
        nrm = sqrt(nrm);
        if (nrm == 0.0) continue;
        const double alpha = A[k * 3 + k] > 0 ? -nrm : nrm;
        double v[5] = { 0, 0, 0, 0, 0 };
        for (int r = k; r < 5; r++) v[r] = A[r * 3 + k];
        v[k] -= alpha;
        double vv = 0; for (int r = k; r < 5; r++) vv += v[r] * v[r];
        if (vv == 0.0) continue;
        for (int c = k; c < 3; c++) {
            double dot = 0; for (int r = k; r < 5; r++) dot += v[r] * A[r * 3 + c];
            const double f = 2.0 * dot / vv;
            for (int r = k; r < 5; r++) A[r * 3 + c] -= f * v[r];
        }
        double dot = 0; for (int r = k; r < 5; r++) dot += v[r] * b[r];
        const double f = 2.0 * dot / vv;
        for (int r = k; r < 5; r++) b[r] -= f * v[r];
    }
    double y[3];
    for (int k = 2; k >= 0; k--) {
        double s = b[k];
        for (int c = k + 1; c < 3; c++) s -= A[k * 3 + c] * y[c];
        y[k] = A[k * 3 + k] != 0.0 ? s / A[k * 3 + k] : 0.0;
    }
    for (int k = 0; k < 3; k++) norm[perm[k]] = y[k];
    const double nn = sqrt(norm[0] * norm[0] + norm[1] * norm[1] + norm[2] * norm[2]);
    *negative_OA_dot_norm = 1.0 / nn;
    for (int k = 0; k < 3; k++) norm[k] /= nn;
    for (int j = 0; j < 5; j++)
        if (fabs(norm[0] * pts[j * 3] + norm[1] * pts[j * 3 + 1] + norm[2] * pts[j * 3 + 2] + *negative_OA_dot_norm) > 0.2) return 0;
    return 1;
}

static int cube_of(double v, int cen)
{
    int c = (int)((v + 25.0) / 50.0) + cen;
    if (v + 25.0 < 0) c--;
    return c;
}

static void shift_axis(lo_map *m, int axis, int dir)
{
    /* dir = +1: every cube moves one index up along the axis (the last one is recycled, emptied, as the first);
     * dir = -1: the other way.  Pointer rotation like the upstream loops over laserCloudCornerArray / SurfArray. */
    const int n[3] = { MW, MH, MD };
    const int stride[3] = { 1, MW, MW * MH };
    const int a1 = (axis + 1) % 3, a2 = (axis + 2) % 3;
    for (int which = 0; which < 2; which++) {
        cloudbuf *arr = which ? m->surf : m->corner;
        for (int u = 0; u < n[a1]; u++)
            for (int v = 0; v < n[a2]; v++) {
                const int base = u * stride[a1] + v * stride[a2];
                if (dir > 0) {
                    cloudbuf hold = arr[base + (n[axis] - 1) * stride[axis]];
                    for (int i = n[axis] - 1; i >= 1; i--) arr[base + i * stride[axis]] = arr[base + (i - 1) * stride[axis]];
                    hold.n = 0;
                    arr[base] = hold;
                } else {
                    cloudbuf hold = arr[base];
                    for (int i = 0; i < n[axis] - 1; i++) arr[base + i * stride[axis]] = arr[base + (i + 1) * stride[axis]];
                    hold.n = 0;
                    arr[base + (n[axis] - 1) * stride[axis]] = hold;
                }
            }
    }
}

/* The optimisation part of one frame: 2 x [5-NN of every stack point in the map clouds -> line / plane tests -> factors ->
 * ceres::Solve].  x = q_w_curr (xyzw), t_w_curr in/out.  corr_out (optional): the residual blocks of the LAST outer
 * iteration (capacity n_cs + n_ss), *n_corr_out their number. */
int lo_map_refine(const lo_pt *cmap, int n_cmap, const lo_pt *smap, int n_smap, const lo_pt *cstack, int n_cs,
                  const lo_pt *sstack, int n_ss, double x[7], lo_map_stats *st, lo_corr *corr_out, int *n_corr_out)
{
    if (n_corr_out) *n_corr_out = 0;
    if (!(n_cmap > 10 && n_smap > 50)) return 0;
        lo_kdtree *kc = lo_kdtree_build(cmap, n_cmap), *ks = lo_kdtree_build(smap, n_smap);
        lo_corr *cs = (lo_corr *)malloc(sizeof(lo_corr) * (size_t)(n_cs + n_ss + 1));
        for (int it = 0; it < 2; it++) {
            int nc = 0, n_edge = 0, n_plane = 0;
            for (int i = 0; i < n_cs; i++) {
                /* pointAssociateToMap: double transform, stored in a float point */
                const double v[3] = { (double)cstack[i].x, (double)cstack[i].y, (double)cstack[i].z };
                double r[3];
                q_rot(x, v, r);
                const float sx = (float)(r[0] + x[4]), sy = (float)(r[1] + x[5]), sz = (float)(r[2] + x[6]);
                int idx[5]; float d2[5];
                if (lo_kdtree_knn(kc, sx, sy, sz, 5, idx, d2) < 5 || !((double)d2[4] < 1.0)) continue;
                double c[3] = { 0, 0, 0 };
                for (int j = 0; j < 5; j++) { c[0] += (double)cmap[idx[j]].x; c[1] += (double)cmap[idx[j]].y; c[2] += (double)cmap[idx[j]].z; }
                for (int k = 0; k < 3; k++) c[k] = c[k] / 5.0;
                double cov[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
                for (int j = 0; j < 5; j++) {
                    const double z[3] = { (double)cmap[idx[j]].x - c[0], (double)cmap[idx[j]].y - c[1], (double)cmap[idx[j]].z - c[2] };
                    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) cov[a * 3 + b] += z[a] * z[b];
                }
                double ev[3], evec[9];
                lo_sym_eig3(cov, ev, evec);
                if (!(ev[2] > 3.0 * ev[1])) continue;
                lo_corr *o = &cs[nc++];
                o->kind = 1; o->cp[0] = cstack[i].x; o->cp[1] = cstack[i].y; o->cp[2] = cstack[i].z;
                for (int k = 0; k < 3; k++) { o->a[k] = 0.1 * evec[k * 3 + 2] + c[k]; o->b[k] = -0.1 * evec[k * 3 + 2] + c[k]; }
                n_edge++;
            }
            for (int i = 0; i < n_ss; i++) {
                const double v[3] = { (double)sstack[i].x, (double)sstack[i].y, (double)sstack[i].z };
                double r[3];
                q_rot(x, v, r);
                const float sx = (float)(r[0] + x[4]), sy = (float)(r[1] + x[5]), sz = (float)(r[2] + x[6]);
                int idx[5]; float d2[5];
                if (lo_kdtree_knn(ks, sx, sy, sz, 5, idx, d2) < 5 || !((double)d2[4] < 1.0)) continue;
                double P[15], nrm[3], d;
                for (int j = 0; j < 5; j++) { P[j * 3] = (double)smap[idx[j]].x; P[j * 3 + 1] = (double)smap[idx[j]].y; P[j * 3 + 2] = (double)smap[idx[j]].z; }
                if (!lo_plane_fit5(P, nrm, &d)) continue;
                lo_corr *o = &cs[nc++];
                o->kind = 3; o->cp[0] = sstack[i].x; o->cp[1] = sstack[i].y; o->cp[2] = sstack[i].z;
                o->a[0] = d; o->a[1] = 0; o->a[2] = 0;
                for (int k = 0; k < 3; k++) o->b[k] = nrm[k];
                n_plane++;
            }
            double c0, c1;
            if (corr_out && it == 1) { memcpy(corr_out, cs, sizeof(lo_corr) * (size_t)nc); if (n_corr_out) *n_corr_out = nc; }
            const int iters = lo_lm_solve(cs, nc, x, &c0, &c1);
            if (st) { st->n_edge[it] = n_edge; st->n_plane[it] = n_plane; st->lm_iters[it] = iters; st->final_cost[it] = c1; }
        }
        free(cs); lo_kdtree_free(kc); lo_kdtree_free(ks);
    return 0;
}

int lo_map_process(lo_map *m, const lo_pt *corner_last, int n_corner, const lo_pt *surf_last, int n_surf,
                   const double q_wodom[4], const double t_wodom[3], double q_w_curr[4], double t_w_curr[3], lo_map_stats *st)
{
    if (st) memset(st, 0, sizeof(*st));
    /* transformAssociateToMap */
    double tmp[3];
    q_mul(m->q_wmap_wodom, q_wodom, q_w_curr);
    q_rot(m->q_wmap_wodom, t_wodom, tmp);
    for (int k = 0; k < 3; k++) t_w_curr[k] = tmp[k] + m->t_wmap_wodom[k];

    int ci = cube_of(t_w_curr[0], m->cen_w), cj = cube_of(t_w_curr[1], m->cen_h), ck = cube_of(t_w_curr[2], m->cen_d);
    while (ci < 3) { shift_axis(m, 0, +1); ci++; m->cen_w++; }
    while (ci >= MW - 3) { shift_axis(m, 0, -1); ci--; m->cen_w--; }
    while (cj < 3) { shift_axis(m, 1, +1); cj++; m->cen_h++; }
    while (cj >= MH - 3) { shift_axis(m, 1, -1); cj--; m->cen_h--; }
    while (ck < 3) { shift_axis(m, 2, +1); ck++; m->cen_d++; }
    while (ck >= MD - 3) { shift_axis(m, 2, -1); ck--; m->cen_d--; }

    int valid[125], n_valid = 0;
    for (int i = ci - 2; i <= ci + 2; i++)
        for (int j = cj - 2; j <= cj + 2; j++)
            for (int k = ck - 1; k <= ck + 1; k++)
                if (i >= 0 && i < MW && j >= 0 && j < MH && k >= 0 && k < MD) valid[n_valid++] = i + MW * j + MW * MH * k;
    int n_cmap = 0, n_smap = 0;
    for (int v = 0; v < n_valid; v++) { n_cmap += m->corner[valid[v]].n; n_smap += m->surf[valid[v]].n; }
    lo_pt *cmap = (lo_pt *)malloc(sizeof(lo_pt) * (size_t)(n_cmap + 1)), *smap = (lo_pt *)malloc(sizeof(lo_pt) * (size_t)(n_smap + 1));
    n_cmap = 0; n_smap = 0;
    for (int v = 0; v < n_valid; v++) {
        memcpy(cmap + n_cmap, m->corner[valid[v]].p, sizeof(lo_pt) * (size_t)m->corner[valid[v]].n); n_cmap += m->corner[valid[v]].n;
        memcpy(smap + n_smap, m->surf[valid[v]].p, sizeof(lo_pt) * (size_t)m->surf[valid[v]].n); n_smap += m->surf[valid[v]].n;
    }
    lo_pt *cstack = (lo_pt *)malloc(sizeof(lo_pt) * (size_t)(n_corner + 1)), *sstack = (lo_pt *)malloc(sizeof(lo_pt) * (size_t)(n_surf + 1));
    const int n_cs = lo_voxel_filter(corner_last, n_corner, m->inv_line, cstack);
    const int n_ss = lo_voxel_filter(surf_last, n_surf, m->inv_plane, sstack);
    if (st) { st->n_corner_stack = n_cs; st->n_surf_stack = n_ss; st->n_corner_map = n_cmap; st->n_surf_map = n_smap; }

    double x[7] = { q_w_curr[0], q_w_curr[1], q_w_curr[2], q_w_curr[3], t_w_curr[0], t_w_curr[1], t_w_curr[2] };
    lo_map_refine(cmap, n_cmap, smap, n_smap, cstack, n_cs, sstack, n_ss, x, st, NULL, NULL);
    for (int k = 0; k < 4; k++) q_w_curr[k] = x[k];
    for (int k = 0; k < 3; k++) t_w_curr[k] = x[4 + k];
    /* transformUpdate: q_wmap_wodom = q_w_curr * q_wodom^-1, t_wmap_wodom = t_w_curr - q_wmap_wodom * t_wodom */
    {
        const double n2 = q_wodom[0] * q_wodom[0] + q_wodom[1] * q_wodom[1] + q_wodom[2] * q_wodom[2] + q_wodom[3] * q_wodom[3];
        const double qi[4] = { -q_wodom[0] / n2, -q_wodom[1] / n2, -q_wodom[2] / n2, q_wodom[3] / n2 };
        q_mul(q_w_curr, qi, m->q_wmap_wodom);
        q_rot(m->q_wmap_wodom, t_wodom, tmp);
        for (int k = 0; k < 3; k++) m->t_wmap_wodom[k] = t_w_curr[k] - tmp[k];
    }
    /* the scan joins the map */
    for (int which = 0; which < 2; which++) {
        const lo_pt *stk = which ? sstack : cstack;
        const int ns = which ? n_ss : n_cs;
        cloudbuf *arr = which ? m->surf : m->corner;
        for (int i = 0; i < ns; i++) {
            const double v[3] = { (double)stk[i].x, (double)stk[i].y, (double)stk[i].z };
            double r[3];
            q_rot(x, v, r);
            const lo_pt sel = { (float)(r[0] + x[4]), (float)(r[1] + x[5]), (float)(r[2] + x[6]), stk[i].i };
            const int cI = cube_of((double)sel.x, m->cen_w), cJ = cube_of((double)sel.y, m->cen_h), cK = cube_of((double)sel.z, m->cen_d);
            if (cI >= 0 && cI < MW && cJ >= 0 && cJ < MH && cK >= 0 && cK < MD) cb_push(&arr[cI + MW * cJ + MW * MH * cK], &sel);
        }
    }
    for (int v = 0; v < n_valid; v++)
        for (int which = 0; which < 2; which++) {
            cloudbuf *c = &(which ? m->surf : m->corner)[valid[v]];
            if (c->n == 0) continue;
            lo_pt *out = (lo_pt *)malloc(sizeof(lo_pt) * (size_t)c->n);
            const int no = lo_voxel_filter(c->p, c->n, which ? m->inv_plane : m->inv_line, out);
            memcpy(c->p, out, sizeof(lo_pt) * (size_t)no);
            c->n = no;
            free(out);
        }
    free(cmap); free(smap); free(cstack); free(sstack);
    return 0;
}
