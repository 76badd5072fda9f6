/*
 * oracle/lo_odometry.c -- TEST INFRASTRUCTURE (CPU oracle).  PARITY UNPINNED.
 *
 * Restates one iteration of the A-LOAM laserOdometry.cpp main loop and lidarFactor.hpp
 * (LidarEdgeFactor / LidarPlaneFactor), source absent from /root/reference (SURVEY.md
 * Appendix A.2, A.3), plus the part of Ceres Solver (third-party, un-pinned, absent) that the
 * loop invokes: TrustRegionMinimizer + LevenbergMarquardtStrategy with Solver::Options
 * defaults except linear_solver_type = DENSE_QR and max_num_iterations = 4, HuberLoss(0.1),
 * EigenQuaternionParameterization, AutoDiffCostFunction.
 *
 * Deviations, stated: (1) the LM linear system is solved by Cholesky on the 6x6 normal
 * equations instead of Householder QR of the augmented Jacobian (same minimiser; ~1e-12
 * relative difference); (2) automatic differentiation is restated with a 7-wide dual number.
 */
#include "lo_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define DIST_SQ_THRESHOLD 25.0
#define NEARBY_SCAN 2.5

/* ---------- quaternion helpers (Eigen conventions, coefficient order x,y,z,w) ---------- */
static void quat_rotate(const double q[4], const double v[3], double out[3])
{
    /* Eigen QuaternionBase::_transformVector: v + w*uv + u x uv, uv = 2 (u x v) */
    double ux = q[0], uy = q[1], uz = q[2], w = q[3];
    double uvx = 2.0 * (uy * v[2] - uz * v[1]);
    double uvy = 2.0 * (uz * v[0] - ux * v[2]);
    double uvz = 2.0 * (ux * v[1] - uy * v[0]);
    out[0] = v[0] + w * uvx + (uy * uvz - uz * uvy);
    out[1] = v[1] + w * uvy + (uz * uvx - ux * uvz);
    out[2] = v[2] + w * uvz + (ux * uvy - uy * uvx);
}

static void quat_mul(const double a[4], const double b[4], double out[4])
{
    /* Eigen quaternion product a*b, xyzw */
    double ax = a[0], ay = a[1], az = a[2], aw = a[3];
    double bx = b[0], by = b[1], bz = b[2], bw = b[3];
    out[3] = aw * bw - ax * bx - ay * by - az * bz;
    out[0] = aw * bx + ax * bw + ay * bz - az * by;
    out[1] = aw * by + ay * bw + az * bx - ax * bz;
    out[2] = aw * bz + az * bw + ax * by - ay * bx;
}

void lo_pose_accumulate(double q_w[4], double t_w[3], const double q[4], const double t[3])
{
    double r[3], qn[4];
    quat_rotate(q_w, t, r);
    t_w[0] += r[0]; t_w[1] += r[1]; t_w[2] += r[2];
    quat_mul(q_w, q, qn);
    memcpy(q_w, qn, sizeof(qn));
}

/* Eigen slerp(t=1) of identity towards q returns +q or -q (scale0 = 0, scale1 = +-1); the rotation
 * and every derivative are unchanged by the sign, so with DISTORTION = 0 (s = 1) it is the identity map. */

/* ---------- 7-wide dual numbers (ceres::Jet<double,7>) ---------- */
typedef struct { double v; double d[7]; } jet;
static jet jc(double c) { jet r; r.v = c; memset(r.d, 0, sizeof(r.d)); return r; }
static jet jvar(double c, int k) { jet r = jc(c); r.d[k] = 1.0; return r; }
static jet jadd(jet a, jet b) { jet r; r.v = a.v + b.v; for (int k = 0; k < 7; k++) r.d[k] = a.d[k] + b.d[k]; return r; }
static jet jsub(jet a, jet b) { jet r; r.v = a.v - b.v; for (int k = 0; k < 7; k++) r.d[k] = a.d[k] - b.d[k]; return r; }
static jet jmul(jet a, jet b) { jet r; r.v = a.v * b.v; for (int k = 0; k < 7; k++) r.d[k] = a.v * b.d[k] + a.d[k] * b.v; return r; }
static jet jscale(jet a, double s) { jet r; r.v = a.v * s; for (int k = 0; k < 7; k++) r.d[k] = a.d[k] * s; return r; }

static void jet_transform(const double x[7], const float cp[3], jet lp[3])
{
    /* lp = q * cp + t with q = (x[0..3]) xyzw, t = x[4..6]; s = 1 */
    jet ux = jvar(x[0], 0), uy = jvar(x[1], 1), uz = jvar(x[2], 2), w = jvar(x[3], 3);
    jet tx = jvar(x[4], 4), ty = jvar(x[5], 5), tz = jvar(x[6], 6);
    jet vx = jc((double)cp[0]), vy = jc((double)cp[1]), vz = jc((double)cp[2]);
    jet uvx = jscale(jsub(jmul(uy, vz), jmul(uz, vy)), 2.0);
    jet uvy = jscale(jsub(jmul(uz, vx), jmul(ux, vz)), 2.0);
    jet uvz = jscale(jsub(jmul(ux, vy), jmul(uy, vx)), 2.0);
    lp[0] = jadd(jadd(jadd(vx, jmul(w, uvx)), jsub(jmul(uy, uvz), jmul(uz, uvy))), tx);
    lp[1] = jadd(jadd(jadd(vy, jmul(w, uvy)), jsub(jmul(uz, uvx), jmul(ux, uvz))), ty);
    lp[2] = jadd(jadd(jadd(vz, jmul(w, uvz)), jsub(jmul(ux, uvy), jmul(uy, uvx))), tz);
}

/* ---------- correspondences ---------- */
/* lo_corr (lo_oracle.h): kind 1 edge (a, b = the two line points), 2 plane (a = point j, b = unit normal),
 * 3 plane-norm of laserMapping (b = unit normal, a[0] = negative_OA_dot_norm: r = n . lp + d) */
typedef lo_corr corr;

/* EigenQuaternionParameterization::ComputeJacobian (4x3, rows x,y,z,w) */
static void quat_local_jac(const double q[4], double J[12])
{
    J[0] = q[3];  J[1] = q[2];   J[2] = -q[1];
    J[3] = -q[2]; J[4] = q[3];   J[5] = q[0];
    J[6] = q[1];  J[7] = -q[0];  J[8] = q[3];
    J[9] = -q[0]; J[10] = -q[1]; J[11] = -q[2];
}

/* EigenQuaternionParameterization::Plus + identity on t */
static void manifold_plus(const double x[7], const double delta[6], double out[7])
{
    double nd = sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
    if (nd > 0.0) {
        double s = sin(nd) / nd;
        double dq[4] = { s * delta[0], s * delta[1], s * delta[2], cos(nd) };
        quat_mul(dq, x, out); /* delta_q * x */
    } else {
        out[0] = x[0]; out[1] = x[1]; out[2] = x[2]; out[3] = x[3];
    }
    out[4] = x[4] + delta[3]; out[5] = x[5] + delta[4]; out[6] = x[6] + delta[5];
}

static void huber(double s, double rho[3])
{
    const double a = 0.1, b = 0.1 * 0.1;
    if (s > b) {
        double r = sqrt(s);
        rho[0] = 2.0 * a * r - b;
        rho[1] = a / r; if (rho[1] < DBL_MIN) rho[1] = DBL_MIN;
        rho[2] = -rho[1] / (2.0 * s);
    } else { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
}

/* Evaluate cost and (optionally) the 6x6 normal equations H = J^T J, g = J^T r of the corrected,
 * locally-parameterised problem.  H is full symmetric row-major. */
static double evaluate(const corr *cs, int nc, const double x[7], double *H, double *g)
{
    double cost = 0.0;
    double Jp[12];
    if (H) { memset(H, 0, 36 * sizeof(double)); memset(g, 0, 6 * sizeof(double)); quat_local_jac(x, Jp); }
    for (int c = 0; c < nc && !H; c++) {
        /* cost-only evaluation (T = double instantiation of the functors) */
        double v[3] = { (double)cs[c].cp[0], (double)cs[c].cp[1], (double)cs[c].cp[2] }, lp[3];
        quat_rotate(x, v, lp);
        lp[0] += x[4]; lp[1] += x[5]; lp[2] += x[6];
        double sq;
        if (cs[c].kind == 1) {
            double ax = lp[0] - cs[c].a[0], ay = lp[1] - cs[c].a[1], az = lp[2] - cs[c].a[2];
            double bx = lp[0] - cs[c].b[0], by = lp[1] - cs[c].b[1], bz = lp[2] - cs[c].b[2];
            double nux = ay * bz - az * by, nuy = az * bx - ax * bz, nuz = ax * by - ay * bx;
            double dex = cs[c].a[0] - cs[c].b[0], dey = cs[c].a[1] - cs[c].b[1], dez = cs[c].a[2] - cs[c].b[2];
            double den = sqrt(dex * dex + dey * dey + dez * dez);
            double r0 = nux / den, r1 = nuy / den, r2 = nuz / den;
            sq = r0 * r0 + r1 * r1 + r2 * r2;
        } else if (cs[c].kind == 2) {
            double r0 = (lp[0] - cs[c].a[0]) * cs[c].b[0] + (lp[1] - cs[c].a[1]) * cs[c].b[1] + (lp[2] - cs[c].a[2]) * cs[c].b[2];
            sq = r0 * r0;
        } else {
            /* LidarPlaneNormFactor: norm.dot(point_w) + negative_OA_dot_norm */
            double r0 = (cs[c].b[0] * lp[0] + cs[c].b[1] * lp[1] + cs[c].b[2] * lp[2]) + cs[c].a[0];
            sq = r0 * r0;
        }
        double rho[3];
        huber(sq, rho);
        cost += 0.5 * rho[0];
    }
    if (!H) return cost;
    for (int c = 0; c < nc; c++) {
        jet lp[3];
        jet_transform(x, cs[c].cp, lp);
        jet res[3];
        int nr;
        if (cs[c].kind == 1) {
            jet ax = jsub(lp[0], jc(cs[c].a[0])), ay = jsub(lp[1], jc(cs[c].a[1])), az = jsub(lp[2], jc(cs[c].a[2]));
            jet bx = jsub(lp[0], jc(cs[c].b[0])), by = jsub(lp[1], jc(cs[c].b[1])), bz = jsub(lp[2], jc(cs[c].b[2]));
            jet nux = jsub(jmul(ay, bz), jmul(az, by));
            jet nuy = jsub(jmul(az, bx), jmul(ax, bz));
            jet nuz = jsub(jmul(ax, by), jmul(ay, bx));
            double dex = cs[c].a[0] - cs[c].b[0], dey = cs[c].a[1] - cs[c].b[1], dez = cs[c].a[2] - cs[c].b[2];
            double den = sqrt(dex * dex + dey * dey + dez * dez);
            res[0] = jscale(nux, 1.0 / den); res[1] = jscale(nuy, 1.0 / den); res[2] = jscale(nuz, 1.0 / den);
            /* value path uses a true division like Jet / Jet */
            res[0].v = nux.v / den; res[1].v = nuy.v / den; res[2].v = nuz.v / den;
            nr = 3;
        } else if (cs[c].kind == 2) {
            jet dx = jsub(lp[0], jc(cs[c].a[0])), dy = jsub(lp[1], jc(cs[c].a[1])), dz = jsub(lp[2], jc(cs[c].a[2]));
            res[0] = jadd(jadd(jscale(dx, cs[c].b[0]), jscale(dy, cs[c].b[1])), jscale(dz, cs[c].b[2]));
            nr = 1;
        } else {
            res[0] = jadd(jadd(jadd(jscale(lp[0], cs[c].b[0]), jscale(lp[1], cs[c].b[1])), jscale(lp[2], cs[c].b[2])), jc(cs[c].a[0]));
            nr = 1;
        }
        double sq = 0.0;
        for (int r = 0; r < nr; r++) sq += res[r].v * res[r].v;
        double rho[3];
        huber(sq, rho);
        cost += 0.5 * rho[0];
        if (!H) continue;
        /* Corrector with rho[2] <= 0: scale residuals and Jacobians by sqrt(rho') */
        double sr = sqrt(rho[1]);
        for (int r = 0; r < nr; r++) {
            double Jl[6];
            for (int k = 0; k < 3; k++)
                Jl[k] = res[r].d[0] * Jp[0 + k] + res[r].d[1] * Jp[3 + k] + res[r].d[2] * Jp[6 + k] + res[r].d[3] * Jp[9 + k];
            Jl[3] = res[r].d[4]; Jl[4] = res[r].d[5]; Jl[5] = res[r].d[6];
            double rv = res[r].v * sr;
            for (int k = 0; k < 6; k++) Jl[k] *= sr;
            for (int i = 0; i < 6; i++) {
                g[i] += Jl[i] * rv;
                for (int j = 0; j < 6; j++) H[i * 6 + j] += Jl[i] * Jl[j];
            }
        }
    }
    return cost;
}

/* Cholesky solve of a 6x6 SPD system; returns 0 on success */
static int chol_solve6(const double *A, const double *b, double *x)
{
    double L[36];
    memset(L, 0, sizeof(L));
    for (int i = 0; i < 6; i++) {
        for (int j = 0; j <= i; j++) {
            double s = A[i * 6 + j];
            for (int k = 0; k < j; k++) s -= L[i * 6 + k] * L[j * 6 + k];
            if (i == j) {
                if (!(s > 0.0)) return -1;
                L[i * 6 + i] = sqrt(s);
            } else L[i * 6 + j] = s / L[j * 6 + j];
        }
    }
    double y[6];
    for (int i = 0; i < 6; i++) {
        double s = b[i];
        for (int k = 0; k < i; k++) s -= L[i * 6 + k] * y[k];
        y[i] = s / L[i * 6 + i];
    }
    for (int i = 5; i >= 0; i--) {
        double s = y[i];
        for (int k = i + 1; k < 6; k++) s -= L[k * 6 + i] * x[k];
        x[i] = s / L[i * 6 + i];
    }
    return 0;
}

static double norm7(const double *x) { double s = 0; for (int i = 0; i < 7; i++) s += x[i] * x[i]; return sqrt(s); }

/* ceres::Solve restated (TrustRegionMinimizer, LEVENBERG_MARQUARDT, max_num_iterations = 4) */
int lo_lm_solve(const lo_corr *cs, int nc, double x[7], double *cost0, double *cost1)
{
    const int max_iter = 4;
    const double function_tol = 1e-6, gradient_tol = 1e-10, parameter_tol = 1e-8;
    const double min_rel_decrease = 1e-3, min_diag = 1e-6, max_diag = 1e32, max_radius = 1e16, min_radius = 1e-32;
    double radius = 1e4, decrease_factor = 2.0;
    int reuse_diagonal = 0, invalid_steps = 0;
    double H[36], g[6], scale[6], diag[6];
    double x_cost = evaluate(cs, nc, x, H, g);
    *cost0 = x_cost; *cost1 = x_cost;
    if (nc == 0) return 0;
    double x_norm = norm7(x);
    for (int i = 0; i < 6; i++) scale[i] = 1.0 / (1.0 + sqrt(H[i * 6 + i]));
    double gmax = 0; for (int i = 0; i < 6; i++) if (fabs(g[i]) > gmax) gmax = fabs(g[i]);
    if (gmax <= gradient_tol) return 0;
    int iter = 0;
    while (iter < max_iter) {
        iter++;
        /* scaled normal equations */
        double Hs[36], gs[6];
        for (int i = 0; i < 6; i++) { gs[i] = g[i] * scale[i]; for (int j = 0; j < 6; j++) Hs[i * 6 + j] = H[i * 6 + j] * scale[i] * scale[j]; }
        if (!reuse_diagonal)
            for (int i = 0; i < 6; i++) { double d = Hs[i * 6 + i]; d = d < min_diag ? min_diag : d; d = d > max_diag ? max_diag : d; diag[i] = d; }
        double A[36];
        memcpy(A, Hs, sizeof(A));
        for (int i = 0; i < 6; i++) A[i * 6 + i] += diag[i] / radius;
        double step[6];
        int ok = chol_solve6(A, gs, step) == 0;
        for (int i = 0; i < 6 && ok; i++) if (!isfinite(step[i])) ok = 0;
        double model_change = 0.0;
        if (ok) {
            for (int i = 0; i < 6; i++) step[i] = -step[i];
            /* model_cost_change = -(J d)^T (r + J d / 2) = -(d^T g + d^T H d / 2) */
            double dg = 0, dHd = 0;
            for (int i = 0; i < 6; i++) { dg += step[i] * gs[i]; for (int j = 0; j < 6; j++) dHd += step[i] * Hs[i * 6 + j] * step[j]; }
            model_change = -(dg + 0.5 * dHd);
        }
        if (!ok || !(model_change > 0.0)) {
            if (++invalid_steps >= 5) break;
            radius *= 0.5; reuse_diagonal = 1;
            continue;
        }
        invalid_steps = 0;
        double delta[6], cand[7];
        for (int i = 0; i < 6; i++) delta[i] = step[i] * scale[i];
        manifold_plus(x, delta, cand);
        double cand_cost = evaluate(cs, nc, cand, NULL, NULL);
        /* parameter tolerance */
        double sn = 0; for (int i = 0; i < 7; i++) sn += (x[i] - cand[i]) * (x[i] - cand[i]);
        sn = sqrt(sn);
        if (sn <= parameter_tol * (x_norm + parameter_tol)) break;
        /* function tolerance */
        if (fabs(x_cost - cand_cost) <= function_tol * x_cost) break;
        double rel = (x_cost - cand_cost) / model_change;
        if (rel > min_rel_decrease) {
            memcpy(x, cand, sizeof(cand));
            x_norm = norm7(x);
            x_cost = evaluate(cs, nc, x, H, g);
            double t = 2.0 * rel - 1.0;
            double den = 1.0 - t * t * t; if (den < 1.0 / 3.0) den = 1.0 / 3.0;
            radius = radius / den; if (radius > max_radius) radius = max_radius;
            decrease_factor = 2.0; reuse_diagonal = 0;
            gmax = 0; for (int i = 0; i < 6; i++) if (fabs(g[i]) > gmax) gmax = fabs(g[i]);
            if (gmax <= gradient_tol) break;
        } else {
            radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diagonal = 1;
        }
        if (radius <= min_radius) break;
    }
    *cost1 = x_cost;
    return iter;
}

static inline float fdist2(const lo_pt *p, const lo_pt *q)
{
    float dx = p->x - q->x, dy = p->y - q->y, dz = p->z - q->z;
    return dx * dx + dy * dy + dz * dz;
}

int lo_odom_step(const lo_pt *sharp, int n_sharp, const lo_pt *flat, int n_flat,
                 const lo_pt *corner_last, int n_corner_last,
                 const lo_pt *surf_last, int n_surf_last,
                 double q[4], double t[3], int use_kdtree,
                 lo_odom_stats *stats, int32_t *corr_out)
{
    lo_kdtree *kc = NULL, *ks = NULL;
    if (use_kdtree) { kc = lo_kdtree_build(corner_last, n_corner_last); ks = lo_kdtree_build(surf_last, n_surf_last); }
    corr *cs = (corr *)malloc(sizeof(corr) * (size_t)(n_sharp + n_flat + 1));
    double x[7] = { q[0], q[1], q[2], q[3], t[0], t[1], t[2] };
    if (stats) memset(stats, 0, sizeof(*stats));

    for (int opti = 0; opti < 2; opti++) {
        int nc = 0, n_edge = 0, n_plane = 0;
        int32_t *co = corr_out ? corr_out + (size_t)opti * (size_t)(n_sharp + n_flat) * 4 : NULL;
        for (int i = 0; i < n_sharp; i++) {
            /* TransformToStart: double transform, result stored in a float point */
            double v[3] = { (double)sharp[i].x, (double)sharp[i].y, (double)sharp[i].z }, r[3];
            quat_rotate(x, v, r);
            lo_pt sel = { (float)(r[0] + x[4]), (float)(r[1] + x[5]), (float)(r[2] + x[6]), sharp[i].i };
            float d2; int closest = use_kdtree ? lo_kdtree_nn(kc, sel.x, sel.y, sel.z, &d2) : lo_brute_nn(corner_last, n_corner_last, sel.x, sel.y, sel.z, &d2);
            int ind2 = -1;
            if (closest >= 0 && (double)d2 < DIST_SQ_THRESHOLD) {
                int ring = (int)corner_last[closest].i;
                double min2 = DIST_SQ_THRESHOLD;
                for (int j = closest + 1; j < n_corner_last; j++) {
                    if ((int)corner_last[j].i <= ring) continue;
                    if ((double)(int)corner_last[j].i > (double)ring + NEARBY_SCAN) break;
                    double d = (double)fdist2(&corner_last[j], &sel);
                    if (d < min2) { min2 = d; ind2 = j; }
                }
                for (int j = closest - 1; j >= 0; j--) {
                    if ((int)corner_last[j].i >= ring) continue;
                    if ((double)(int)corner_last[j].i < (double)ring - NEARBY_SCAN) break;
                    double d = (double)fdist2(&corner_last[j], &sel);
                    if (d < min2) { min2 = d; ind2 = j; }
                }
            } else closest = -1;
            if (co) { co[4 * i] = -1; co[4 * i + 1] = -1; co[4 * i + 2] = -1; co[4 * i + 3] = 0; }
            if (ind2 >= 0) {
                corr *c = &cs[nc++];
                c->kind = 1;
                c->cp[0] = sharp[i].x; c->cp[1] = sharp[i].y; c->cp[2] = sharp[i].z;
                c->a[0] = corner_last[closest].x; c->a[1] = corner_last[closest].y; c->a[2] = corner_last[closest].z;
                c->b[0] = corner_last[ind2].x; c->b[1] = corner_last[ind2].y; c->b[2] = corner_last[ind2].z;
                n_edge++;
                if (co) { co[4 * i] = closest; co[4 * i + 1] = ind2; co[4 * i + 3] = 1; }
            }
        }
        for (int i = 0; i < n_flat; i++) {
            double v[3] = { (double)flat[i].x, (double)flat[i].y, (double)flat[i].z }, r[3];
            quat_rotate(x, v, r);
            lo_pt sel = { (float)(r[0] + x[4]), (float)(r[1] + x[5]), (float)(r[2] + x[6]), flat[i].i };
            float d2; int closest = use_kdtree ? lo_kdtree_nn(ks, sel.x, sel.y, sel.z, &d2) : lo_brute_nn(surf_last, n_surf_last, sel.x, sel.y, sel.z, &d2);
            int ind2 = -1, ind3 = -1;
            if (closest >= 0 && (double)d2 < DIST_SQ_THRESHOLD) {
                int ring = (int)surf_last[closest].i;
                double min2 = DIST_SQ_THRESHOLD, min3 = DIST_SQ_THRESHOLD;
                for (int j = closest + 1; j < n_surf_last; j++) {
                    int rj = (int)surf_last[j].i;
                    if ((double)rj > (double)ring + NEARBY_SCAN) break;
                    double d = (double)fdist2(&surf_last[j], &sel);
                    if (rj <= ring && d < min2) { min2 = d; ind2 = j; }
                    else if (rj > ring && d < min3) { min3 = d; ind3 = j; }
                }
                for (int j = closest - 1; j >= 0; j--) {
                    int rj = (int)surf_last[j].i;
                    if ((double)rj < (double)ring - NEARBY_SCAN) break;
                    double d = (double)fdist2(&surf_last[j], &sel);
                    if (rj >= ring && d < min2) { min2 = d; ind2 = j; }
                    else if (rj < ring && d < min3) { min3 = d; ind3 = j; }
                }
            } else closest = -1;
            int32_t *cf = co ? co + 4 * (size_t)(n_sharp + i) : NULL;
            if (cf) { cf[0] = -1; cf[1] = -1; cf[2] = -1; cf[3] = 0; }
            if (ind2 >= 0 && ind3 >= 0) {
                corr *c = &cs[nc++];
                c->kind = 2;
                c->cp[0] = flat[i].x; c->cp[1] = flat[i].y; c->cp[2] = flat[i].z;
                double pj[3] = { surf_last[closest].x, surf_last[closest].y, surf_last[closest].z };
                double pl[3] = { surf_last[ind2].x, surf_last[ind2].y, surf_last[ind2].z };
                double pm[3] = { surf_last[ind3].x, surf_last[ind3].y, surf_last[ind3].z };
                double u[3] = { pj[0] - pl[0], pj[1] - pl[1], pj[2] - pl[2] };
                double w[3] = { pj[0] - pm[0], pj[1] - pm[1], pj[2] - pm[2] };
                double n[3] = { u[1] * w[2] - u[2] * w[1], u[2] * w[0] - u[0] * w[2], u[0] * w[1] - u[1] * w[0] };
                double nn = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
                /* Eigen normalize(): divides by the norm when it is > 0 */
                if (nn > 0.0) { n[0] /= nn; n[1] /= nn; n[2] /= nn; }
                memcpy(c->a, pj, sizeof(pj)); memcpy(c->b, n, sizeof(n));
                n_plane++;
                if (cf) { cf[0] = closest; cf[1] = ind2; cf[2] = ind3; cf[3] = 2; }
            }
        }
        double c0, c1;
        int it = lo_lm_solve(cs, nc, x, &c0, &c1);
        if (stats) {
            stats->n_corner_corr[opti] = n_edge; stats->n_plane_corr[opti] = n_plane;
            stats->lm_iters[opti] = it; stats->initial_cost[opti] = c0; stats->final_cost[opti] = c1;
        }
    }
    q[0] = x[0]; q[1] = x[1]; q[2] = x[2]; q[3] = x[3];
    t[0] = x[4]; t[1] = x[5]; t[2] = x[6];
    free(cs);
    lo_kdtree_free(kc); lo_kdtree_free(ks);
    return 0;
}
