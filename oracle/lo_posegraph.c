/*
 * oracle/lo_posegraph.c -- TEST INFRASTRUCTURE (see lo_oracle.h).  CPU statement of the loop-closure pose graph
 * (SURVEY row 8f-2).
 *
 * NEW FEATURE, NO PARITY TARGET: the reference publishes loop constraints (KeyFrame::findConnection,
 * mono_lidar_mapping/src/loop_detection/KeyFrame.cc:570-684: loop_info = relative_t, relative_q (w x y z), relative_yaw)
 * and re-anchors the window rigidly (Estimator.cc:309-365) but never optimises a graph; it only carries the unused
 * 4-DoF leftovers AngleLocalParameterization / NormalizeAngle / YawPitchRollToRotationMatrix
 * (include/loop_detection/Loop_Detector.h:99-168) and mathutils::R2ypr (include/utils/math_utils.h:187-202).  This file
 * states the 4-DoF (yaw + translation, degrees) keyframe graph those helpers were written for: every keyframe is tied to
 * its (up to) four predecessors by its odometry, every loop adds one Huber(0.1)-robustified edge with the yaw residual
 * down-weighted by 10, the first keyframe is held fixed, and the solve is a Ceres-style Levenberg-Marquardt (the loop
 * of lo_lm_solve) on the sparse normal equations.  It is the checker of lmono_pose_graph_* only.
 */
#include "lo_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define PG_PI 3.14159265358979323846

static double normalize_angle(double a)          /* Loop_Detector.h:99-107 */
{
    if (a > 180.0) return a - 360.0;
    if (a < -180.0) return a + 360.0;
    return a;
}

/* mathutils::R2ypr (math_utils.h:187-202) of the rotation of quaternion q (x y z w), degrees */
void lo_pg_q2ypr(const double q[4], double ypr[3])
{
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double R00 = 1 - 2 * (y * y + z * z), R01 = 2 * (x * y - z * w), R02 = 2 * (x * z + y * w);
    const double R10 = 2 * (x * y + z * w), R11 = 1 - 2 * (x * x + z * z), R12 = 2 * (y * z - x * w);
    const double R20 = 2 * (x * z - y * w);
    const double yy = atan2(R10, R00);
    const double pp = atan2(-R20, R00 * cos(yy) + R10 * sin(yy));
    const double rr = atan2(R02 * sin(yy) - R12 * cos(yy), -R01 * sin(yy) + R11 * cos(yy));
    ypr[0] = yy / PG_PI * 180.0; ypr[1] = pp / PG_PI * 180.0; ypr[2] = rr / PG_PI * 180.0;
}

/* YawPitchRollToRotationMatrix (Loop_Detector.h:129-147) and its derivative with respect to yaw (per radian) */
static void ypr_to_R(double yaw, double pitch, double roll, double R[9], double dR[9])
{
    const double y = yaw / 180.0 * PG_PI, p = pitch / 180.0 * PG_PI, r = roll / 180.0 * PG_PI;
    const double cy = cos(y), sy = sin(y), cp = cos(p), sp = sin(p), cr = cos(r), sr = sin(r);
    R[0] = cy * cp; R[1] = -sy * cr + cy * sp * sr; R[2] = sy * sr + cy * sp * cr;
    R[3] = sy * cp; R[4] = cy * cr + sy * sp * sr;  R[5] = -cy * sr + sy * sp * cr;
    R[6] = -sp;     R[7] = cp * sr;                 R[8] = cp * cr;
    if (dR) {
        dR[0] = -sy * cp; dR[1] = -cy * cr - sy * sp * sr; dR[2] = cy * sr - sy * sp * cr;
        dR[3] = cy * cp;  dR[4] = -sy * cr + cy * sp * sr; dR[5] = sy * sr + cy * sp * cr;
        dR[6] = 0; dR[7] = 0; dR[8] = 0;
    }
}

void lo_pg_ypr2q(const double ypr[3], double q[4])
{
    double R[9];
    ypr_to_R(ypr[0], ypr[1], ypr[2], R, NULL);
    /* Eigen::Quaterniond(Matrix3d) */
    const double tr = R[0] + R[4] + R[8];
    if (tr > 0) {
        double t = sqrt(tr + 1.0);
        q[3] = 0.5 * t; t = 0.5 / t;
        q[0] = (R[7] - R[5]) * t; q[1] = (R[2] - R[6]) * t; q[2] = (R[3] - R[1]) * t;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[4 * i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        double t = sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
        q[i] = 0.5 * t; t = 0.5 / t;
        q[3] = (R[3 * k + j] - R[3 * j + k]) * t;
        q[j] = (R[3 * j + i] + R[3 * i + j]) * t;
        q[k] = (R[3 * k + i] + R[3 * i + k]) * t;
    }
}

typedef struct { int a, b, loop; double mt[3], myaw; } pg_edge;

typedef struct {
    int n, n_edges, w;           /* nodes, edges, half bandwidth in blocks */
    pg_edge *e;
    double *pitch, *roll;        /* fixed (degrees) */
    int *pos;                    /* elimination position of every node */
} pg_graph;

/* residual (4) and Jacobians (4 x 4 each, columns yaw tx ty tz) of one edge; returns 1/2 rho(|r|^2) */
static double edge_eval(const pg_graph *g, const pg_edge *e, const double *x /* [n][4] */, double r[4], double Ja[16], double Jb[16])
{
    const double *xa = x + 4 * e->a, *xb = x + 4 * e->b;
    double R[9], dR[9];
    ypr_to_R(xa[0], g->pitch[e->a], g->roll[e->a], R, dR);
    const double d[3] = { xb[1] - xa[1], xb[2] - xa[2], xb[3] - xa[3] };
    const double wt = 1.0, wy = e->loop ? 0.1 : 1.0;
    for (int k = 0; k < 3; k++) {
        r[k] = (R[k] * d[0] + R[3 + k] * d[1] + R[6 + k] * d[2] - e->mt[k]) * wt;
        if (Ja) {
            Ja[4 * k] = wt * (PG_PI / 180.0) * (dR[k] * d[0] + dR[3 + k] * d[1] + dR[6 + k] * d[2]);
            Jb[4 * k] = 0.0;
            for (int c = 0; c < 3; c++) { Ja[4 * k + 1 + c] = -wt * R[3 * c + k]; Jb[4 * k + 1 + c] = wt * R[3 * c + k]; }
        }
    }
    r[3] = normalize_angle(xb[0] - xa[0] - e->myaw) * wy;
    if (Ja) { Ja[12] = -wy; Jb[12] = wy; for (int c = 1; c < 4; c++) { Ja[12 + c] = 0.0; Jb[12 + c] = 0.0; } }
    const double s = r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3];
    if (!e->loop) return 0.5 * s;
    const double a = 0.1;                                   /* ceres::HuberLoss(0.1) with Ceres' corrector (rho'' <= 0) */
    if (s <= a * a) return 0.5 * s;
    const double sq = sqrt(s), wr = sqrt(a / sq);
    for (int k = 0; k < 4; k++) r[k] *= wr;
    if (Ja) for (int k = 0; k < 16; k++) { Ja[k] *= wr; Jb[k] *= wr; }
    return 0.5 * (2.0 * a * sq - a * a);
}

/* band storage: scalar row i = 4 pos + c holds columns 4 (pos - w) .. 4 pos + 3 in band[i * bw + (col - 4 (pos - w))], bw = 4 (w + 1).
 * Only the edges whose newer keyframe b lies in [lo, hi) are added (the share of one rank); the identity rows of the constant
 * first keyframe are written by the rank that owns keyframe 0. */
static double linearise(const pg_graph *g, const double *x, double *band, double *grad, int lo, int hi)
{
    const int n = g->n, w = g->w, bw = 4 * (w + 1);
    const int p0 = g->pos[0];
    double cost = 0.0;
    memset(band, 0, sizeof(double) * (size_t)n * 4 * bw); memset(grad, 0, sizeof(double) * (size_t)n * 4);
    for (int k = 0; k < g->n_edges; k++) {
        const pg_edge *e = &g->e[k];
        if (e->b < lo || e->b >= hi) continue;
        double r[4], J[2][16];
        cost += edge_eval(g, e, x, r, J[0], J[1]);
        const int node[2] = { e->a, e->b };
        for (int s = 0; s < 2; s++) {
            const int ps = g->pos[node[s]];
            if (node[s] == 0) continue;                      /* the first keyframe is constant: its rows / columns leave the system */
            for (int c = 0; c < 4; c++) {
                double acc = 0;
                for (int q = 0; q < 4; q++) acc += J[s][4 * q + c] * r[q];
                grad[4 * ps + c] += acc;
            }
            for (int u = 0; u < 2; u++) {
                const int pu = g->pos[node[u]];
                if (pu > ps || node[u] == 0) continue;       /* lower triangle only */
                for (int c = 0; c < 4; c++)
                    for (int c2 = 0; c2 < 4; c2++) {
                        if (pu == ps && c2 > c) continue;
                        double acc = 0;
                        for (int q = 0; q < 4; q++) acc += J[s][4 * q + c] * J[u][4 * q + c2];
                        band[(size_t)(4 * ps + c) * bw + (4 * pu + c2 - 4 * (ps - w))] += acc;
                    }
            }
        }
    }
    if (lo == 0) for (int c = 0; c < 4; c++) band[(size_t)(4 * p0 + c) * bw + (4 * w + c)] = 1.0;
    return cost;
}

/* in-place banded Cholesky of the scaled, damped system and solve; returns 0 on success */
static int band_solve(int n4, int w, double *A, const double *rhs, double *out)
{
    const int bw = 4 * (w + 1);
#define AT(i, j) A[(size_t)(i) * bw + ((j) - 4 * ((i) / 4 - w))]
    for (int i = 0; i < n4; i++) {
        const int ci = 4 * (i / 4 - w) > 0 ? 4 * (i / 4 - w) : 0;
        for (int j = ci; j <= i; j++) {
            const int cj = 4 * (j / 4 - w) > 0 ? 4 * (j / 4 - w) : 0;
            double s = AT(i, j);
            for (int k = ci > cj ? ci : cj; k < j; k++) s -= AT(i, k) * AT(j, k);
            if (i == j) { if (!(s > 0.0)) return 1; AT(i, i) = sqrt(s); }
            else AT(i, j) = s / AT(j, j);
        }
    }
    for (int i = 0; i < n4; i++) {
        const int ci = 4 * (i / 4 - w) > 0 ? 4 * (i / 4 - w) : 0;
        double s = rhs[i];
        for (int k = ci; k < i; k++) s -= AT(i, k) * out[k];
        out[i] = s / AT(i, i);
    }
    for (int i = n4 - 1; i >= 0; i--) {
        double s = out[i];
        const int hi = 4 * (i / 4 + w) + 3 < n4 - 1 ? 4 * (i / 4 + w) + 3 : n4 - 1;
        for (int k = i + 1; k <= hi; k++) s -= AT(k, i) * out[k];
        out[i] = s / AT(i, i);
    }
#undef AT
    return 0;
}

/* reverse Cuthill-McKee over the keyframe graph; returns the half bandwidth in blocks */
static int rcm_order(int n, int n_edges, const pg_edge *e, int *pos)
{
    int *deg = calloc((size_t)n + 1, sizeof(int)), *start = calloc((size_t)n + 2, sizeof(int));
    for (int k = 0; k < n_edges; k++) { deg[e[k].a]++; deg[e[k].b]++; }
    for (int v = 0; v < n; v++) start[v + 1] = start[v] + deg[v];
    int *adj = malloc(sizeof(int) * (size_t)(start[n] + 1)), *fill = calloc((size_t)n + 1, sizeof(int));
    for (int k = 0; k < n_edges; k++) { adj[start[e[k].a] + fill[e[k].a]++] = e[k].b; adj[start[e[k].b] + fill[e[k].b]++] = e[k].a; }
    int *order = malloc(sizeof(int) * (size_t)n), *seen = calloc((size_t)n, sizeof(int));
    int cnt = 0;
    for (int root = 0; root < n; root++) {
        if (seen[root]) continue;
        int head = cnt;
        order[cnt++] = root; seen[root] = 1;
        while (head < cnt) {
            const int v = order[head++];
            const int first = cnt;
            for (int k = start[v]; k < start[v + 1]; k++) { const int u = adj[k]; if (!seen[u]) { seen[u] = 1; order[cnt++] = u; } }
            for (int i = first + 1; i < cnt; i++) {          /* neighbours by increasing degree (insertion sort, ties by index) */
                const int u = order[i]; int j = i - 1;
                while (j >= first && (deg[order[j]] > deg[u] || (deg[order[j]] == deg[u] && order[j] > u))) { order[j + 1] = order[j]; j--; }
                order[j + 1] = u;
            }
        }
    }
    for (int i = 0; i < n; i++) pos[order[n - 1 - i]] = i;
    int w = 0;
    for (int k = 0; k < n_edges; k++) { const int d = abs(pos[e[k].a] - pos[e[k].b]); if (d > w) w = d; }
    free(deg); free(start); free(adj); free(fill); free(order); free(seen);
    return w;
}

/* ---- the solver as the same rounds the product runs: linearise (a rank's share) -> sum over ranks -> step ---- */
struct lo_pg {
    pg_graph g;
    int n4, bw;
    double *x, *cand;                 /* [n][4] yaw, t */
    double *H, *gr, *A, *scale, *diag, *gs, *step, *sol;
    double radius, decrease_factor, x_cost, cost0, model_change, x_norm, gmax;
    int iter, invalid_steps, reuse_diagonal, done, started, have_cand, accepted, rejected;
};

void lo_pg_free(lo_pg *s)
{
    if (!s) return;
    free(s->g.e); free(s->g.pitch); free(s->g.roll); free(s->g.pos); free(s->x); free(s->cand); free(s->H); free(s->gr); free(s->A);
    free(s);
}

/* poses_tq: [n][7] t (x y z), q (x y z w) of the keyframes (odometry).  loops: [n_loops] (old index, current index),
 * loop_info [n_loops][8] in the layout of KeyFrame.cc:630-633.  ordering: 0 reverse Cuthill-McKee band, 1 natural order with a
 * full band (dense: cross-check of the band logic). */
lo_pg *lo_pg_create(int n, const double *poses_tq, int n_loops, const int32_t *loops, const double *loop_info, int ordering)
{
    if (n < 2) return NULL;
    lo_pg *s = calloc(1, sizeof(lo_pg));
    pg_graph *g = &s->g;
    g->n = n;
    g->e = malloc(sizeof(pg_edge) * (size_t)(4 * n + n_loops));
    g->pitch = malloc(sizeof(double) * (size_t)n); g->roll = malloc(sizeof(double) * (size_t)n);
    g->pos = malloc(sizeof(int) * (size_t)n);
    s->x = malloc(sizeof(double) * (size_t)n * 4); s->cand = malloc(sizeof(double) * (size_t)n * 4);
    double *x = s->x;
    for (int i = 0; i < n; i++) {
        double ypr[3];
        lo_pg_q2ypr(poses_tq + 7 * i + 3, ypr);
        x[4 * i] = ypr[0]; g->pitch[i] = ypr[1]; g->roll[i] = ypr[2];
        for (int c = 0; c < 3; c++) x[4 * i + 1 + c] = poses_tq[7 * i + c];
    }
    int ne = 0;
    for (int i = 1; i < n; i++)
        for (int j = 1; j <= 4; j++) {
            if (i - j < 0) continue;
            const int a = i - j;
            pg_edge *e = &g->e[ne++];
            e->a = a; e->b = i; e->loop = 0;
            /* relative_t = q_a^-1 (t_i - t_a) with the full odometry rotation, relative_yaw = yaw_i - yaw_a */
            const double *q = poses_tq + 7 * a + 3;
            const double d[3] = { poses_tq[7 * i] - poses_tq[7 * a], poses_tq[7 * i + 1] - poses_tq[7 * a + 1], poses_tq[7 * i + 2] - poses_tq[7 * a + 2] };
            const double qx = -q[0], qy = -q[1], qz = -q[2], qw = q[3];
            const double uvx = 2.0 * (qy * d[2] - qz * d[1]), uvy = 2.0 * (qz * d[0] - qx * d[2]), uvz = 2.0 * (qx * d[1] - qy * d[0]);
            e->mt[0] = d[0] + qw * uvx + (qy * uvz - qz * uvy);
            e->mt[1] = d[1] + qw * uvy + (qz * uvx - qx * uvz);
            e->mt[2] = d[2] + qw * uvz + (qx * uvy - qy * uvx);
            e->myaw = x[4 * i] - x[4 * a];
        }
    for (int k = 0; k < n_loops; k++) {
        pg_edge *e = &g->e[ne++];
        e->a = loops[2 * k]; e->b = loops[2 * k + 1]; e->loop = 1;
        if (e->a < 0 || e->a >= n || e->b < 0 || e->b >= n || e->a == e->b) { lo_pg_free(s); return NULL; }
        for (int c = 0; c < 3; c++) e->mt[c] = loop_info[8 * k + c];
        e->myaw = loop_info[8 * k + 7];
    }
    g->n_edges = ne;
    if (ordering == 0) g->w = rcm_order(n, ne, g->e, g->pos);
    else { for (int i = 0; i < n; i++) g->pos[i] = i; g->w = n - 1; }
    s->n4 = 4 * n; s->bw = 4 * (g->w + 1);
    s->H = malloc(sizeof(double) * (size_t)s->n4 * s->bw); s->A = malloc(sizeof(double) * (size_t)s->n4 * s->bw);
    s->gr = malloc(sizeof(double) * (size_t)s->n4 * 6);
    s->scale = s->gr + s->n4; s->diag = s->scale + s->n4; s->gs = s->diag + s->n4; s->step = s->gs + s->n4; s->sol = s->step + s->n4;
    return s;
}

int64_t lo_pg_reduce_count(const lo_pg *s) { return (int64_t)s->n4 * s->bw + s->n4 + 1; }
int lo_pg_bandwidth(const lo_pg *s) { return s->g.w; }

/* buf: [H band | g | cost] of the edges owned by `rank` at the current linearisation point */
void lo_pg_linearise(lo_pg *s, int rank, int world, double *buf)
{
    const int n = s->g.n, lo = (int)((int64_t)n * rank / world), hi = (int)((int64_t)n * (rank + 1) / world);
    const size_t hsz = (size_t)s->n4 * s->bw;
    buf[hsz + s->n4] = linearise(&s->g, s->started ? s->cand : s->x, buf, buf + hsz, lo, hi);
}

/* one trust-region round on the summed buffer (the loop of lo_lm_solve, Ceres defaults): decide on the last candidate, solve for
 * the next one.  Returns 1 when finished. */
int lo_pg_step(lo_pg *s, const double *buf, int max_iter)
{
    const pg_graph *g = &s->g;
    const int n = g->n, n4 = s->n4, bw = s->bw;
    const size_t hsz = (size_t)n4 * bw;
    const double function_tol = 1e-6, gradient_tol = 1e-10, parameter_tol = 1e-8;
    const double min_rel_decrease = 1e-3, min_diag = 1e-6, max_diag = 1e32, max_radius = 1e16, min_radius = 1e-32;
    if (s->done) return 1;
    const double lin_cost = buf[hsz + n4];
    int take = 0;
    if (!s->started) {
        s->started = 1; take = 1;
        s->radius = 1e4; s->decrease_factor = 2.0; s->x_cost = lin_cost; s->cost0 = lin_cost;
        double xn = 0;
        for (int i = 0; i < n4; i++) xn += s->x[i] * s->x[i];
        s->x_norm = sqrt(xn);
        for (int i = 0; i < n4; i++) s->scale[i] = 1.0 / (1.0 + sqrt(buf[(size_t)i * bw + (i - 4 * (i / 4 - g->w))]));
    } else if (s->have_cand) {
        const double rel = (s->x_cost - lin_cost) / s->model_change;
        if (rel > min_rel_decrease) {
            take = 1;
            const int converged = fabs(s->x_cost - lin_cost) <= function_tol * s->x_cost;
            memcpy(s->x, s->cand, sizeof(double) * (size_t)n4);
            double xn = 0;
            for (int i = 0; i < n4; i++) xn += s->x[i] * s->x[i];
            s->x_norm = sqrt(xn); s->x_cost = lin_cost;
            const double t = 2.0 * rel - 1.0;
            double den = 1.0 - t * t * t; if (den < 1.0 / 3.0) den = 1.0 / 3.0;
            s->radius = s->radius / den; if (s->radius > max_radius) s->radius = max_radius;
            s->decrease_factor = 2.0; s->reuse_diagonal = 0; s->accepted++;
            if (converged) s->done = 1;                      /* function tolerance: the step is kept */
        } else {
            s->radius = s->radius / s->decrease_factor; s->decrease_factor *= 2.0; s->reuse_diagonal = 1; s->rejected++;
        }
        s->have_cand = 0;
        if (s->radius <= min_radius) s->done = 1;
    }
    if (take) {
        memcpy(s->H, buf, sizeof(double) * hsz); memcpy(s->gr, buf + hsz, sizeof(double) * (size_t)n4);
        s->gmax = 0;
        for (int i = 0; i < n4; i++) if (fabs(s->gr[i]) > s->gmax) s->gmax = fabs(s->gr[i]);
        if (s->gmax <= gradient_tol) s->done = 1;
    }
    while (!s->done) {
        if (s->iter >= max_iter) { s->done = 1; break; }
        s->iter++;
        double *A = s->A;
        for (int i = 0; i < n4; i++) {
            s->gs[i] = s->gr[i] * s->scale[i];
            const int c0 = 4 * (i / 4 - g->w);
            for (int j = 0; j < bw; j++) { const int col = c0 + j; A[(size_t)i * bw + j] = (col >= 0 && col <= i) ? s->H[(size_t)i * bw + j] * s->scale[i] * s->scale[col] : 0.0; }
            if (!s->reuse_diagonal) { double d = A[(size_t)i * bw + (i - c0)]; d = d < min_diag ? min_diag : d; d = d > max_diag ? max_diag : d; s->diag[i] = d; }
            A[(size_t)i * bw + (i - c0)] += s->diag[i] / s->radius;
        }
        int ok = band_solve(n4, g->w, A, s->gs, s->sol) == 0;
        for (int i = 0; i < n4 && ok; i++) if (!isfinite(s->sol[i])) ok = 0;
        double model_change = 0.0;
        if (ok) {
            double dg = 0, dd = 0;
            for (int i = 0; i < n4; i++) { s->step[i] = -s->sol[i]; dg += s->step[i] * s->gs[i]; dd += s->diag[i] / s->radius * s->step[i] * s->step[i]; }
            model_change = -0.5 * dg + 0.5 * dd;             /* -(d^T g + d^T H d / 2) with (H + D) d = -g */
        }
        if (!ok || !(model_change > 0.0)) {
            if (++s->invalid_steps >= 5) s->done = 1;
            s->radius *= 0.5; s->reuse_diagonal = 1;
            continue;
        }
        s->invalid_steps = 0;
        double sn = 0;
        for (int v = 0; v < n; v++) {
            const int p = g->pos[v];
            s->cand[4 * v] = normalize_angle(s->x[4 * v] + s->step[4 * p] * s->scale[4 * p]);       /* AngleLocalParameterization */
            for (int c = 1; c < 4; c++) s->cand[4 * v + c] = s->x[4 * v + c] + s->step[4 * p + c] * s->scale[4 * p + c];
            for (int c = 0; c < 4; c++) sn += (s->cand[4 * v + c] - s->x[4 * v + c]) * (s->cand[4 * v + c] - s->x[4 * v + c]);
        }
        sn = sqrt(sn);
        s->model_change = model_change; s->have_cand = 1;
        if (sn <= parameter_tol * (s->x_norm + parameter_tol)) { s->done = 1; s->have_cand = 0; }
        break;
    }
    return s->done;
}

/* stats: iterations, initial cost, final cost, half bandwidth (blocks), accepted, rejected steps */
void lo_pg_result(const lo_pg *s, double *out_tq, double *stats)
{
    for (int i = 0; i < s->g.n; i++) {
        const double ypr[3] = { s->x[4 * i], s->g.pitch[i], s->g.roll[i] };
        for (int c = 0; c < 3; c++) out_tq[7 * i + c] = s->x[4 * i + 1 + c];
        lo_pg_ypr2q(ypr, out_tq + 7 * i + 3);
    }
    if (stats) { stats[0] = s->iter; stats[1] = s->cost0; stats[2] = s->x_cost; stats[3] = s->g.w; stats[4] = s->accepted; stats[5] = s->rejected; }
}

int lo_pose_graph_optimize(int n, const double *poses_tq, int n_loops, const int32_t *loops, const double *loop_info,
                           int max_iter, int ordering, double *out_tq, double *stats)
{
    lo_pg *s = lo_pg_create(n, poses_tq, n_loops, loops, loop_info, ordering);
    if (!s) return -1;
    double *buf = malloc(sizeof(double) * (size_t)lo_pg_reduce_count(s));
    for (int round = 0; round <= max_iter; round++) {
        lo_pg_linearise(s, 0, 1, buf);
        if (lo_pg_step(s, buf, max_iter)) break;
    }
    lo_pg_result(s, out_tq, stats);
    free(buf);
    lo_pg_free(s);
    return 0;
}
