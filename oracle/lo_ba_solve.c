/*
 * oracle/lo_ba_solve.c -- TEST INFRASTRUCTURE (CPU oracle).  PARITY UNPINNED.
 *
 * Restates the solve inside Estimator::optimization() (/root/reference/mono_lidar_mapping/src/image_process/
 * Estimator.cc:1124-1305): problem assembly (:1128-1215: extrinsic + window poses with PoseLocalParameterization,
 * optional PriorFactor on the extrinsic, LASERFactor between consecutive poses without loss, MonoProjectionFactor
 * per (feature, observation j != i) with CauchyLoss(1.0)), and ceres::Solve with DENSE_SCHUR + DOGLEG,
 * max_num_iterations = NUM_ITERATIONS, everything else Ceres defaults (:1260-1277).
 *
 * Ceres itself is third-party and absent (version un-pinned, LocalParameterization API => < 2.2); its trust-region
 * loop, TRADITIONAL_DOGLEG strategy, Jacobi scaling, Schur elimination of the 1-D inverse-depth blocks and the
 * robust-loss corrector are restated from its published algorithm.  The elimination order does
 * not change the solution; compare converged states, never per-iteration traces.
 */
#include "lo_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

void lo_laser_factor(const double *params, const double *consts, const double *sqrt_info, double *r, double *J);
void lo_mono_factor(const double *params, const double *consts, const double *sqrt_info, double *r, double *J);
void lo_prior_factor(const double *params, const double *consts, const double *weights, double *r, double *J);
void lo_pose_plus(const double *x, const double *delta, double *out);
void lo_cauchy(double s, double a, double rho[3]);

typedef struct {
    int n_poses, n_feat, n_obs;
    int use_prior, ex_constant, use_mono, max_iter;
    double *poses;              /* [n_poses][7] in/out */
    double *ex;                 /* [7] in/out          */
    double *inv_depth;          /* [n_feat] in/out     */
    const int32_t *obs_feat, *obs_i, *obs_j;   /* [n_obs] */
    const double *obs_pts;      /* [n_obs][4] pt_i.xy, pt_j.xy */
    const double *laser_consts; /* [n_poses-1][24] */
    const double *laser_info;   /* [36] */
    const double *mono_info;    /* [4]  */
    const double *prior_T;      /* [16] */
    const double *prior_w;      /* [2]  */
} lo_ba_problem;

typedef struct {
    double initial_cost, final_cost;
    int iterations, termination;   /* 0 CONVERGENCE, 1 NO_CONVERGENCE, 2 FAILURE */
    int n_successful, n_unsuccessful;
} lo_ba_summary;

typedef struct {
    int P, F;          /* camera-side local size (6 per pose, +6 for the extrinsic when free), depth count */
    int ex_off;        /* -1 when the extrinsic is constant */
    double *Hpp, *Hpd, *Hdd, *gp, *gd;
} ba_sys;

static int pose_off(const ba_sys *s, int i) { return (s->ex_off < 0 ? 0 : 6) + 6 * i; }

static void add_block(ba_sys *s, const double *Ja, int oa, const double *Jb, int ob, int nr)
{
    /* H[oa.., ob..] += Ja^T Jb for 7-wide row-major Jacobians (first 6 columns) */
    for (int a = 0; a < 6; a++)
        for (int b = 0; b < 6; b++) {
            double v = 0;
            for (int r = 0; r < nr; r++) v += Ja[r * 7 + a] * Jb[r * 7 + b];
            s->Hpp[(size_t)(oa + a) * s->P + ob + b] += v;
            if (oa != ob) s->Hpp[(size_t)(ob + b) * s->P + oa + a] += v;
        }
}

/* cost and (when sys != NULL) the unscaled normal equations of the corrected, locally parameterised problem */
static double evaluate(const lo_ba_problem *p, const double *poses, const double *ex, const double *invd, ba_sys *s)
{
    double cost = 0.0;
    if (s) {
        memset(s->Hpp, 0, sizeof(double) * (size_t)s->P * s->P);
        memset(s->Hpd, 0, sizeof(double) * (size_t)s->P * (s->F > 0 ? s->F : 1));
        memset(s->Hdd, 0, sizeof(double) * (size_t)(s->F > 0 ? s->F : 1));
        memset(s->gp, 0, sizeof(double) * (size_t)s->P);
        memset(s->gd, 0, sizeof(double) * (size_t)(s->F > 0 ? s->F : 1));
    }
    if (p->use_prior && !p->ex_constant) {
        double r[6], J[42];
        lo_prior_factor(ex, p->prior_T, p->prior_w, r, s ? J : NULL);
        for (int k = 0; k < 6; k++) cost += 0.5 * r[k] * r[k];
        if (s) {
            add_block(s, J, s->ex_off, J, s->ex_off, 6);
            for (int a = 0; a < 6; a++) for (int k = 0; k < 6; k++) s->gp[s->ex_off + a] += J[k * 7 + a] * r[k];
        }
    }
    for (int i = 0; i + 1 < p->n_poses; i++) {
        double prm[14], r[6], J[84];
        memcpy(prm, poses + 7 * i, 7 * sizeof(double)); memcpy(prm + 7, poses + 7 * (i + 1), 7 * sizeof(double));
        lo_laser_factor(prm, p->laser_consts + 24 * i, p->laser_info, r, s ? J : NULL);
        for (int k = 0; k < 6; k++) cost += 0.5 * r[k] * r[k];
        if (s) {
            const int oi = pose_off(s, i), oj = pose_off(s, i + 1);
            add_block(s, J, oi, J, oi, 6); add_block(s, J, oi, J + 42, oj, 6); add_block(s, J + 42, oj, J + 42, oj, 6);
            for (int a = 0; a < 6; a++) for (int k = 0; k < 6; k++) { s->gp[oi + a] += J[k * 7 + a] * r[k]; s->gp[oj + a] += J[42 + k * 7 + a] * r[k]; }
        }
    }
    if (p->use_mono) {
        for (int o = 0; o < p->n_obs; o++) {
            const int f = p->obs_feat[o], i = p->obs_i[o], j = p->obs_j[o];
            double prm[22], r[2], J[44];
            memcpy(prm, ex, 7 * sizeof(double)); memcpy(prm + 7, poses + 7 * i, 7 * sizeof(double));
            memcpy(prm + 14, poses + 7 * j, 7 * sizeof(double)); prm[21] = invd[f];
            lo_mono_factor(prm, p->obs_pts + 4 * o, p->mono_info, r, s ? J : NULL);
            double rho[3];
            const double sq = r[0] * r[0] + r[1] * r[1];
            lo_cauchy(sq, 1.0, rho);
            cost += 0.5 * rho[0];
            if (!s) continue;
            /* rho'' < 0 for Cauchy: the corrector only scales by sqrt(rho') */
            const double sr = sqrt(rho[1]);
            for (int k = 0; k < 44; k++) J[k] *= sr;
            r[0] *= sr; r[1] *= sr;
            const int oi = pose_off(s, i), oj = pose_off(s, j);
            const double *Jx = J, *Ji = J + 14, *Jj = J + 28, *Jd = J + 42;
            if (s->ex_off >= 0) {
                add_block(s, Jx, s->ex_off, Jx, s->ex_off, 2); add_block(s, Jx, s->ex_off, Ji, oi, 2); add_block(s, Jx, s->ex_off, Jj, oj, 2);
            }
            add_block(s, Ji, oi, Ji, oi, 2); add_block(s, Ji, oi, Jj, oj, 2); add_block(s, Jj, oj, Jj, oj, 2);
            for (int a = 0; a < 6; a++) {
                if (s->ex_off >= 0) {
                    s->gp[s->ex_off + a] += Jx[a] * r[0] + Jx[7 + a] * r[1];
                    s->Hpd[(size_t)(s->ex_off + a) * s->F + f] += Jx[a] * Jd[0] + Jx[7 + a] * Jd[1];
                }
                s->gp[oi + a] += Ji[a] * r[0] + Ji[7 + a] * r[1];
                s->gp[oj + a] += Jj[a] * r[0] + Jj[7 + a] * r[1];
                s->Hpd[(size_t)(oi + a) * s->F + f] += Ji[a] * Jd[0] + Ji[7 + a] * Jd[1];
                s->Hpd[(size_t)(oj + a) * s->F + f] += Jj[a] * Jd[0] + Jj[7 + a] * Jd[1];
            }
            s->Hdd[f] += Jd[0] * Jd[0] + Jd[1] * Jd[1];
            s->gd[f] += Jd[0] * r[0] + Jd[1] * r[1];
        }
    }
    return cost;
}

static int cholesky(double *A, int n)   /* in-place lower Cholesky, 0 on success */
{
    for (int i = 0; i < n; i++) {
        for (int j = 0; j <= i; j++) {
            double s = A[(size_t)i * n + j];
            for (int k = 0; k < j; k++) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
            if (i == j) { if (!(s > 0.0)) return -1; A[(size_t)i * n + i] = sqrt(s); }
            else A[(size_t)i * n + j] = s / A[(size_t)j * n + j];
        }
    }
    return 0;
}
static void chol_solve(const double *L, int n, double *b)
{
    for (int i = 0; i < n; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= L[(size_t)i * n + k] * b[k]; b[i] = s / L[(size_t)i * n + i]; }
    for (int i = n - 1; i >= 0; i--) { double s = b[i]; for (int k = i + 1; k < n; k++) s -= L[(size_t)k * n + i] * b[k]; b[i] = s / L[(size_t)i * n + i]; }
}

/* y = Hs v for the Jacobi-scaled system; v, y of length P + F; scale of length P + F */
static void hs_mul(const ba_sys *s, const double *scale, const double *v, double *y)
{
    const int P = s->P, F = s->F;
    for (int a = 0; a < P; a++) {
        double acc = 0;
        for (int b = 0; b < P; b++) acc += s->Hpp[(size_t)a * P + b] * scale[b] * v[b];
        for (int f = 0; f < F; f++) acc += s->Hpd[(size_t)a * F + f] * scale[P + f] * v[P + f];
        y[a] = acc * scale[a];
    }
    for (int f = 0; f < F; f++) {
        double acc = s->Hdd[f] * scale[P + f] * v[P + f];
        for (int a = 0; a < P; a++) acc += s->Hpd[(size_t)a * F + f] * scale[a] * v[a];
        y[P + f] = acc * scale[P + f];
    }
}

/* solve (Hs + mu diag(D2)) x = gs by Schur elimination of the depth columns; returns 0 on success */
static int schur_solve(const ba_sys *s, const double *scale, const double *D2, double mu, const double *gs, double *x, double *work)
{
    const int P = s->P, F = s->F;
    double *S = work, *rhs = work + (size_t)P * P;
    for (int a = 0; a < P; a++) {
        for (int b = 0; b < P; b++) S[(size_t)a * P + b] = s->Hpp[(size_t)a * P + b] * scale[a] * scale[b];
        S[(size_t)a * P + a] += mu * D2[a];
        rhs[a] = gs[a];
    }
    for (int f = 0; f < F; f++) {
        const double hff = s->Hdd[f] * scale[P + f] * scale[P + f] + mu * D2[P + f];
        if (!(hff > 0.0)) return -1;
        const double w = 1.0 / hff;
        for (int a = 0; a < P; a++) {
            const double ha = s->Hpd[(size_t)a * F + f] * scale[a] * scale[P + f];
            if (ha == 0.0) continue;
            rhs[a] -= ha * w * gs[P + f];
            for (int b = 0; b < P; b++) {
                const double hb = s->Hpd[(size_t)b * F + f] * scale[b] * scale[P + f];
                if (hb != 0.0) S[(size_t)a * P + b] -= ha * w * hb;
            }
        }
    }
    if (cholesky(S, P) != 0) return -1;
    chol_solve(S, P, rhs);
    for (int a = 0; a < P; a++) x[a] = rhs[a];
    for (int f = 0; f < F; f++) {
        const double hff = s->Hdd[f] * scale[P + f] * scale[P + f] + mu * D2[P + f];
        double acc = gs[P + f];
        for (int a = 0; a < P; a++) acc -= s->Hpd[(size_t)a * F + f] * scale[a] * scale[P + f] * x[a];
        x[P + f] = acc / hff;
    }
    for (int k = 0; k < P + F; k++) if (!isfinite(x[k])) return -1;
    return 0;
}

static void apply_plus(const lo_ba_problem *p, const ba_sys *s, const double *poses, const double *ex, const double *invd,
                       const double *delta, double *poses_o, double *ex_o, double *invd_o)
{
    if (s->ex_off >= 0) lo_pose_plus(ex, delta + s->ex_off, ex_o); else memcpy(ex_o, ex, 7 * sizeof(double));
    for (int i = 0; i < p->n_poses; i++) lo_pose_plus(poses + 7 * i, delta + pose_off(s, i), poses_o + 7 * i);
    for (int f = 0; f < s->F; f++) invd_o[f] = invd[f] + delta[s->P + f];
}

static double global_norm(const lo_ba_problem *p, const ba_sys *s, const double *poses, const double *ex, const double *invd)
{
    double q = 0;
    if (s->ex_off >= 0) for (int k = 0; k < 7; k++) q += ex[k] * ex[k];
    for (int k = 0; k < 7 * p->n_poses; k++) q += poses[k] * poses[k];
    for (int f = 0; f < s->F; f++) q += invd[f] * invd[f];
    return sqrt(q);
}
static double global_diff(const lo_ba_problem *p, const ba_sys *s, const double *pa, const double *ea, const double *da,
                          const double *pb, const double *eb, const double *db)
{
    double q = 0;
    if (s->ex_off >= 0) for (int k = 0; k < 7; k++) q += (ea[k] - eb[k]) * (ea[k] - eb[k]);
    for (int k = 0; k < 7 * p->n_poses; k++) q += (pa[k] - pb[k]) * (pa[k] - pb[k]);
    for (int f = 0; f < s->F; f++) q += (da[f] - db[f]) * (da[f] - db[f]);
    return sqrt(q);
}

int lo_ba_solve(lo_ba_problem *p, lo_ba_summary *sum)
{
    ba_sys s;
    s.F = p->use_mono ? p->n_feat : 0;
    s.ex_off = p->ex_constant ? -1 : 0;
    s.P = 6 * p->n_poses + (p->ex_constant ? 0 : 6);
    const int P = s.P, F = s.F, N = P + F;
    const size_t Fa = (size_t)(F > 0 ? F : 1);
    s.Hpp = (double *)malloc(sizeof(double) * (size_t)P * P); s.Hpd = (double *)malloc(sizeof(double) * (size_t)P * Fa);
    s.Hdd = (double *)malloc(sizeof(double) * Fa); s.gp = (double *)malloc(sizeof(double) * (size_t)P); s.gd = (double *)malloc(sizeof(double) * Fa);
    double *scale = (double *)malloc(sizeof(double) * N), *D = (double *)malloc(sizeof(double) * N), *D2 = (double *)malloc(sizeof(double) * N);
    double *gs = (double *)malloc(sizeof(double) * N), *gd = (double *)malloc(sizeof(double) * N), *gn = (double *)malloc(sizeof(double) * N);
    double *step = (double *)malloc(sizeof(double) * N), *tmp = (double *)malloc(sizeof(double) * N), *tmp2 = (double *)malloc(sizeof(double) * N);
    double *delta = (double *)malloc(sizeof(double) * N), *work = (double *)malloc(sizeof(double) * ((size_t)P * P + P));
    double *cp = (double *)malloc(sizeof(double) * 7 * (size_t)p->n_poses), *cd = (double *)malloc(sizeof(double) * Fa), ce[7];
    double *poses = p->poses, *ex = p->ex, *invd = p->inv_depth;

    const double function_tol = 1e-6, gradient_tol = 1e-10, parameter_tol = 1e-8, min_rel_decrease = 1e-3;
    const double min_diag = 1e-6, max_diag = 1e32, max_radius = 1e16, min_radius = 1e-32;
    const double min_mu = 1e-8, max_mu = 1.0, mu_inc = 10.0;
    double radius = 1e4, mu = min_mu, alpha = 0.0, dogleg_norm = 0.0;
    int reuse = 0, invalid = 0, iter = 0, termination = 1;
    memset(sum, 0, sizeof(*sum));

    double x_cost = evaluate(p, poses, ex, invd, &s);
    sum->initial_cost = x_cost;
    double x_norm = global_norm(p, &s, poses, ex, invd);
    for (int a = 0; a < P; a++) scale[a] = 1.0 / (1.0 + sqrt(s.Hpp[(size_t)a * P + a]));
    for (int f = 0; f < F; f++) scale[P + f] = 1.0 / (1.0 + sqrt(s.Hdd[f]));
    double gmax = 0;
    for (int a = 0; a < P; a++) gmax = fmax(gmax, fabs(s.gp[a]));
    for (int f = 0; f < F; f++) gmax = fmax(gmax, fabs(s.gd[f]));
    if (gmax <= gradient_tol) termination = 0;
    else while (iter < p->max_iter) {
        iter++;
        int ok = 1;
        if (!reuse) {
            for (int a = 0; a < P; a++) { gs[a] = s.gp[a] * scale[a]; double d = s.Hpp[(size_t)a * P + a] * scale[a] * scale[a]; d = d < min_diag ? min_diag : (d > max_diag ? max_diag : d); D2[a] = d; D[a] = sqrt(d); }
            for (int f = 0; f < F; f++) { gs[P + f] = s.gd[f] * scale[P + f]; double d = s.Hdd[f] * scale[P + f] * scale[P + f]; d = d < min_diag ? min_diag : (d > max_diag ? max_diag : d); D2[P + f] = d; D[P + f] = sqrt(d); }
            /* gradient in the dogleg's scaled space and the Cauchy point */
            double g2 = 0;
            for (int k = 0; k < N; k++) { gd[k] = gs[k] / D[k]; g2 += gd[k] * gd[k]; tmp[k] = gd[k] / D[k]; }
            hs_mul(&s, scale, tmp, tmp2);
            double jg2 = 0;
            for (int k = 0; k < N; k++) jg2 += tmp[k] * tmp2[k];
            alpha = g2 / jg2;
            /* Gauss-Newton step with increasing regularisation */
            ok = 0;
            while (mu < max_mu) {
                if (schur_solve(&s, scale, D2, mu, gs, gn, work) == 0) { ok = 1; break; }
                mu *= mu_inc;
            }
            if (ok) {
                mu = fmax(min_mu, 2.0 * mu / mu_inc);
                for (int k = 0; k < N; k++) gn[k] *= -D[k];
            }
        }
        if (ok) {
            /* traditional dogleg interpolation in the scaled space */
            double gn_norm = 0, g_norm = 0;
            for (int k = 0; k < N; k++) { gn_norm += gn[k] * gn[k]; g_norm += gd[k] * gd[k]; }
            gn_norm = sqrt(gn_norm); g_norm = sqrt(g_norm);
            if (gn_norm <= radius) { memcpy(step, gn, sizeof(double) * N); dogleg_norm = gn_norm; }
            else if (alpha * g_norm >= radius) { for (int k = 0; k < N; k++) step[k] = -(radius / g_norm) * gd[k]; dogleg_norm = radius; }
            else {
                double b_dot_a = 0;
                for (int k = 0; k < N; k++) b_dot_a += gd[k] * gn[k];
                b_dot_a *= -alpha;
                const double a2 = alpha * alpha * g_norm * g_norm;
                const double bma2 = a2 - 2 * b_dot_a + gn_norm * gn_norm;
                const double c = b_dot_a - a2;
                const double d = sqrt(c * c + bma2 * (radius * radius - a2));
                const double beta = (c <= 0) ? (d - c) / bma2 : (radius * radius - a2) / (d + c);
                for (int k = 0; k < N; k++) step[k] = (-alpha * (1.0 - beta)) * gd[k] + beta * gn[k];
                dogleg_norm = radius;
            }
            for (int k = 0; k < N; k++) step[k] /= D[k];
        }
        double model_change = 0;
        if (ok) {
            hs_mul(&s, scale, step, tmp);
            double dg = 0, dHd = 0;
            for (int k = 0; k < N; k++) { dg += step[k] * gs[k]; dHd += step[k] * tmp[k]; }
            model_change = -(dg + 0.5 * dHd);
        }
        if (!ok || !(model_change > 0.0)) {
            if (++invalid >= 5) { termination = 2; break; }
            mu *= mu_inc; reuse = 0;
            continue;
        }
        invalid = 0;
        for (int k = 0; k < N; k++) delta[k] = step[k] * scale[k];
        apply_plus(p, &s, poses, ex, invd, delta, cp, ce, cd);
        const double cand_cost = evaluate(p, cp, ce, cd, NULL);
        const double sn = global_diff(p, &s, poses, ex, invd, cp, ce, cd);
        if (sn <= parameter_tol * (x_norm + parameter_tol)) { termination = 0; break; }
        if (fabs(x_cost - cand_cost) <= function_tol * x_cost) { termination = 0; break; }
        const double rel = (x_cost - cand_cost) / model_change;
        if (rel > min_rel_decrease) {
            memcpy(poses, cp, sizeof(double) * 7 * (size_t)p->n_poses); memcpy(ex, ce, sizeof(ce)); memcpy(invd, cd, sizeof(double) * (size_t)F);
            x_norm = global_norm(p, &s, poses, ex, invd);
            x_cost = evaluate(p, poses, ex, invd, &s);
            sum->n_successful++;
            if (rel < 0.25) radius *= 0.5;
            if (rel > 0.75) radius = fmax(radius, 3.0 * dogleg_norm);
            if (radius > max_radius) radius = max_radius;
            reuse = 0;
            gmax = 0;
            for (int a = 0; a < P; a++) gmax = fmax(gmax, fabs(s.gp[a]));
            for (int f = 0; f < F; f++) gmax = fmax(gmax, fabs(s.gd[f]));
            if (gmax <= gradient_tol) { termination = 0; break; }
        } else {
            radius *= 0.5; reuse = 1;
            sum->n_unsuccessful++;
        }
        if (radius <= min_radius) { termination = 0; break; }
    }
    sum->final_cost = x_cost; sum->iterations = iter; sum->termination = termination;
    free(s.Hpp); free(s.Hpd); free(s.Hdd); free(s.gp); free(s.gd); free(scale); free(D); free(D2); free(gs); free(gd); free(gn);
    free(step); free(tmp); free(tmp2); free(delta); free(work); free(cp); free(cd);
    return 0;
}

/* Estimator::double2Matrix re-anchoring (Estimator.cc:1059-1087): rot_diff = Rs0 * R(para_pose[0])^T,
 * Ps[i] = rot_diff (p_i - p_0) + Ps0, Rs[i] = rot_diff R(q_i).  R0_before: 3x3 row-major, P0_before: 3.
 * out_R: [n][9], out_P: [n][3]. */
void lo_ba_reanchor(const double *poses, int n, const double *R0_before, const double *P0_before, double *out_R, double *out_P)
{
    double R0[9], rd[9];
    {
        const double *q = poses + 3;
        const double nn = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        const double x = q[0] / nn, y = q[1] / nn, z = q[2] / nn, w = q[3] / nn;
        const double tx = 2 * x, ty = 2 * y, tz = 2 * z, twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
        R0[0] = 1 - (tyy + tzz); R0[1] = txy - twz; R0[2] = txz + twy; R0[3] = txy + twz; R0[4] = 1 - (txx + tzz); R0[5] = tyz - twx; R0[6] = txz - twy; R0[7] = tyz + twx; R0[8] = 1 - (txx + tyy);
    }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) rd[i * 3 + j] = R0_before[i * 3] * R0[j * 3] + R0_before[i * 3 + 1] * R0[j * 3 + 1] + R0_before[i * 3 + 2] * R0[j * 3 + 2];
    for (int k = 0; k < n; k++) {
        const double *pk = poses + 7 * k, *q = pk + 3;
        const double t[3] = { pk[0] - poses[0], pk[1] - poses[1], pk[2] - poses[2] };
        const double nn = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        const double x = q[0] / nn, y = q[1] / nn, z = q[2] / nn, w = q[3] / nn;
        const double tx = 2 * x, ty = 2 * y, tz = 2 * z, twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
        const double R[9] = { 1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1 - (txx + tyy) };
        for (int i = 0; i < 3; i++) {
            out_P[3 * k + i] = rd[i * 3] * t[0] + rd[i * 3 + 1] * t[1] + rd[i * 3 + 2] * t[2] + P0_before[i];
            for (int j = 0; j < 3; j++) out_R[9 * k + i * 3 + j] = rd[i * 3] * R[j] + rd[i * 3 + 1] * R[3 + j] + rd[i * 3 + 2] * R[6 + j];
        }
    }
}
