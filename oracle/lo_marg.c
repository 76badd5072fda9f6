/*
 * oracle/lo_marg.c -- TEST INFRASTRUCTURE (CPU oracle).  PARITY UNPINNED.
 *
 * Restates the MARGIN_OLD branch of Estimator::margin() (/root/reference/mono_lidar_mapping/src/image_process/
 * Estimator.cc:1307-1405) with MarginalizationInfo::{preMarginalize, marginalize} and Marginalization::Evaluate
 * (src/factor/MarginalizationFactor.cc:109-131, :176-272, :309-373; ResidualBlockInfo::Evaluate :18-68).
 * Factors: LASERFactor(pose0, pose1) dropping pose0, and for every track anchored at frame 0 one MonoProjectionFactor
 * (ex, pose0, pose_j, depth) with CauchyLoss(1) dropping pose0 and the depth.  The reference orders the blocks by
 * std::unordered_map iteration over their addresses (implementation defined); here: m = [pose0, depths...],
 * n = [ex, pose1 .. pose10].  Any order gives the same prior up to a permutation, so tests compare J^T J and J^T r.
 * Notes: the reference never adds this prior to the solve (MarginalizationInfo::valid stays false)
 * and feeds it_per_frame.right_pt, which is never set in the mono pipeline; pt_j is therefore an input here.
 * Eigen::SelfAdjointEigenSolver is restated as a cyclic Jacobi eigen-solver.
 */
#include "lo_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

void lo_laser_factor(const double *params, const double *consts, const double *sqrt_info, double *r, double *J);
void lo_mono_factor(const double *params, const double *consts, const double *sqrt_info, double *r, double *J);
void lo_cauchy(double s, double a, double rho[3]);
void lo_corrector(double *r, int nr, double *J, int nc, const double rho[3]);

/* symmetric eigen-decomposition A = V diag(w) V^T (cyclic Jacobi); A is destroyed, V row-major n x n (columns = vectors) */
static void jacobi_eig(double *A, int n, double *w, double *V)
{
    for (int i = 0; i < n * n; i++) V[i] = 0.0;
    for (int i = 0; i < n; i++) V[i * n + i] = 1.0;
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0, diag = 0;
        for (int p = 0; p < n; p++) { diag += A[p * n + p] * A[p * n + p]; for (int q = p + 1; q < n; q++) off += A[p * n + q] * A[p * n + q]; }
        if (off <= 1e-30 * diag || off == 0.0) break;
        for (int p = 0; p < n; p++)
            for (int q = p + 1; q < n; q++) {
                const double apq = A[p * n + q];
                if (apq == 0.0) continue;
                const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; k++) { const double a = A[k * n + p], b = A[k * n + q]; A[k * n + p] = c * a - s * b; A[k * n + q] = s * a + c * b; }
                for (int k = 0; k < n; k++) { const double a = A[p * n + k], b = A[q * n + k]; A[p * n + k] = c * a - s * b; A[q * n + k] = s * a + c * b; }
                for (int k = 0; k < n; k++) { const double a = V[k * n + p], b = V[k * n + q]; V[k * n + p] = c * a - s * b; V[k * n + q] = s * a + c * b; }
            }
    }
    for (int i = 0; i < n; i++) w[i] = A[i * n + i];
}

/* poses [11][7], ex [7]; n_f0 tracks anchored at frame 0 with inverse depths inv_depth[n_f0]; observation o belongs to
 * track obs_feat[o], frame obs_j[o] (1..10), points obs_pts[o] = pt_i.xy, pt_j.xy.  Outputs: lin_J [66*66] row-major,
 * lin_r [66] (MarginalizationInfo::linearized_jacobians / linearized_residuals), m_out = 6 + n_f0. */
int lo_marginalize(const double *poses, const double *ex, int n_f0, const double *inv_depth, int n_obs, const int32_t *obs_feat,
                   const int32_t *obs_j, const double *obs_pts, const double *laser_consts01, const double *laser_info, const double *mono_info,
                   double *lin_J, double *lin_r, int *m_out)
{
    const int m = 6 + n_f0, n = 66, pos = m + n;
    const double eps = 1e-8;
    double *H = (double *)calloc((size_t)pos * pos, sizeof(double)), *b = (double *)calloc((size_t)pos, sizeof(double));
#define IDX_POSE(i) ((i) == 0 ? 0 : m + 6 + 6 * ((i) - 1))
#define IDX_EX (m)
#define IDX_DEP(f) (6 + (f))
    {   /* LASERFactor(pose0, pose1), no loss */
        double prm[14], r[6], J[84];
        memcpy(prm, poses, 14 * sizeof(double));
        lo_laser_factor(prm, laser_consts01, laser_info, r, J);
        const int idx[2] = { IDX_POSE(0), IDX_POSE(1) };
        for (int a = 0; a < 2; a++) {
            for (int c = a; c < 2; c++)
                for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) {
                    double v = 0; for (int k = 0; k < 6; k++) v += J[42 * a + k * 7 + i] * J[42 * c + k * 7 + j];
                    H[(size_t)(idx[a] + i) * pos + idx[c] + j] += v;
                    if (a != c) H[(size_t)(idx[c] + j) * pos + idx[a] + i] = H[(size_t)(idx[a] + i) * pos + idx[c] + j];
                }
            for (int i = 0; i < 6; i++) { double v = 0; for (int k = 0; k < 6; k++) v += J[42 * a + k * 7 + i] * r[k]; b[idx[a] + i] += v; }
        }
    }
    for (int o = 0; o < n_obs; o++) {
        const int f = obs_feat[o], j = obs_j[o];
        double prm[22], r[2], J[44], rho[3];
        memcpy(prm, ex, 7 * sizeof(double)); memcpy(prm + 7, poses, 7 * sizeof(double)); memcpy(prm + 14, poses + 7 * j, 7 * sizeof(double)); prm[21] = inv_depth[f];
        lo_mono_factor(prm, obs_pts + 4 * o, mono_info, r, J);
        lo_cauchy(r[0] * r[0] + r[1] * r[1], 1.0, rho);
        /* ResidualBlockInfo::Evaluate: robust correction of every Jacobian block, then of the residual */
        double r0[2] = { r[0], r[1] }, rr[2];
        for (int blk = 0; blk < 3; blk++) { rr[0] = r0[0]; rr[1] = r0[1]; lo_corrector(rr, 2, J + 14 * blk, 7, rho); }
        rr[0] = r0[0]; rr[1] = r0[1]; lo_corrector(rr, 2, J + 42, 1, rho);
        r[0] = rr[0]; r[1] = rr[1];
        const int idx[4] = { IDX_EX, IDX_POSE(0), IDX_POSE(j), IDX_DEP(f) }, sz[4] = { 6, 6, 6, 1 };
        const double *Jb[4] = { J, J + 14, J + 28, J + 42 };
        const int ld[4] = { 7, 7, 7, 1 };
        for (int a = 0; a < 4; a++) {
            for (int c = a; c < 4; c++)
                for (int i = 0; i < sz[a]; i++) for (int jj = 0; jj < sz[c]; jj++) {
                    const double v = Jb[a][i] * Jb[c][jj] + Jb[a][ld[a] + i] * Jb[c][ld[c] + jj];
                    H[(size_t)(idx[a] + i) * pos + idx[c] + jj] += v;
                    if (a != c) H[(size_t)(idx[c] + jj) * pos + idx[a] + i] = H[(size_t)(idx[a] + i) * pos + idx[c] + jj];
                }
            for (int i = 0; i < sz[a]; i++) b[idx[a] + i] += Jb[a][i] * r[0] + Jb[a][ld[a] + i] * r[1];
        }
    }
    /* Hmm^-1 by eigen-decomposition with the eps cut */
    double *Hmm = (double *)malloc(sizeof(double) * (size_t)m * m), *w = (double *)malloc(sizeof(double) * (size_t)(m > n ? m : n));
    double *V = (double *)malloc(sizeof(double) * (size_t)(m > n ? m * m : n * n)), *Hinv = (double *)calloc((size_t)m * m, sizeof(double));
    for (int i = 0; i < m; i++) for (int j = 0; j < m; j++) Hmm[i * m + j] = 0.5 * (H[(size_t)i * pos + j] + H[(size_t)j * pos + i]);
    jacobi_eig(Hmm, m, w, V);
    for (int k = 0; k < m; k++) {
        if (!(w[k] > eps)) continue;
        const double iw = 1.0 / w[k];
        for (int i = 0; i < m; i++) for (int j = 0; j < m; j++) Hinv[i * m + j] += V[i * m + k] * iw * V[j * m + k];
    }
    /* Schur: H' = Hrr - Hrm Hmm^-1 Hmr ; b' = brr - Hrm Hmm^-1 bmm */
    double *T = (double *)malloc(sizeof(double) * (size_t)n * m), *Hr = (double *)malloc(sizeof(double) * (size_t)n * n), br[66];
    for (int i = 0; i < n; i++) for (int j = 0; j < m; j++) { double v = 0; for (int k = 0; k < m; k++) v += H[(size_t)(m + i) * pos + k] * Hinv[k * m + j]; T[i * m + j] = v; }
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) { double v = 0; for (int k = 0; k < m; k++) v += T[i * m + k] * H[(size_t)k * pos + m + j]; Hr[i * n + j] = H[(size_t)(m + i) * pos + m + j] - v; }
        double v = 0; for (int k = 0; k < m; k++) v += T[i * m + k] * b[k];
        br[i] = b[m + i] - v;
    }
    jacobi_eig(Hr, n, w, V);
    for (int k = 0; k < n; k++) {
        const double S = w[k] > eps ? w[k] : 0.0, Si = w[k] > eps ? 1.0 / w[k] : 0.0;
        double vb = 0;
        for (int i = 0; i < n; i++) { lin_J[k * n + i] = sqrt(S) * V[i * n + k]; vb += V[i * n + k] * br[i]; }
        lin_r[k] = sqrt(Si) * vb;
    }
    *m_out = m;
    free(H); free(b); free(Hmm); free(w); free(V); free(Hinv); free(T); free(Hr);
    return 0;
}

/* Marginalization::Evaluate: x0 / x = kept blocks [ex, pose1..pose10] (11 x 7) at linearisation / now.  residual [66],
 * jac (optional) [11][66*7] row-major per block with the 7th column zero. */
void lo_marg_evaluate(const double *lin_J, const double *lin_r, const double *x0, const double *x, double *residual, double *jac)
{
    double dx[66];
    for (int bk = 0; bk < 11; bk++) {
        const double *a = x + 7 * bk, *a0 = x0 + 7 * bk;
        for (int k = 0; k < 3; k++) dx[6 * bk + k] = a[k] - a0[k];
        /* 2 * vec(q0^-1 * q), negated when the scalar part is negative */
        const double n2 = a0[3] * a0[3] + a0[4] * a0[4] + a0[5] * a0[5] + a0[6] * a0[6];
        const double ix = -a0[3] / n2, iy = -a0[4] / n2, iz = -a0[5] / n2, iw = a0[6] / n2;
        const double qx = a[3], qy = a[4], qz = a[5], qw = a[6];
        const double rw = iw * qw - ix * qx - iy * qy - iz * qz;
        double rx = iw * qx + ix * qw + iy * qz - iz * qy, ry = iw * qy + iy * qw + iz * qx - ix * qz, rz = iw * qz + iz * qw + ix * qy - iy * qx;
        if (!(rw >= 0)) { rx = -rx; ry = -ry; rz = -rz; }
        dx[6 * bk + 3] = 2.0 * rx; dx[6 * bk + 4] = 2.0 * ry; dx[6 * bk + 5] = 2.0 * rz;
    }
    for (int i = 0; i < 66; i++) { double v = lin_r[i]; for (int k = 0; k < 66; k++) v += lin_J[i * 66 + k] * dx[k]; residual[i] = v; }
    if (jac)
        for (int bk = 0; bk < 11; bk++)
            for (int i = 0; i < 66; i++) { for (int c = 0; c < 6; c++) jac[(size_t)bk * 462 + i * 7 + c] = lin_J[i * 66 + 6 * bk + c]; jac[(size_t)bk * 462 + i * 7 + 6] = 0.0; }
}

/* MARGIN_SECOND_NEW branch of Estimator::margin() (Estimator.cc:1406-1470): the only factor is the previous prior itself,
 * `Marginalization(last_marginalization_info)` over last_marginalization_parameter_blocks (MarginalizationFactor.cc:300-373),
 * with the block that aliases para_pose[WINDOW_SIZE - 1] in the drop set.  preMarginalize evaluates it at the CURRENT parameter
 * values (residual = r0 + J0 dx, Jacobian blocks = columns of J0) and keeps those values as the new linearisation point;
 * marginalize then eliminates the dropped block's 6 local dimensions (H = J^T J, b = J^T r, eigen pseudo-inverse of H_mm with the
 * eps cut, Schur complement, second eigen-decomposition -> linearized_jacobians / linearized_residuals).
 *   nb blocks of the previous prior (7 doubles each, 6 local), x0 / x [nb][7] at its linearisation point / now, drop = block index.
 *   lin_J [6nb x 6nb] row-major, lin_r [6nb].  Outputs over the nb - 1 kept blocks in their old order: out_J [n x n], out_r [n],
 *   n = 6 (nb - 1).  Block order is implementation defined in the reference (unordered_map iteration); any order gives the same
 *   prior up to a permutation. */
int lo_marg_second_new(int nb, int drop, const double *lin_J, const double *lin_r, const double *x0, const double *x, double *out_J, double *out_r)
{
    const int n0 = 6 * nb, n = n0 - 6, m = 6;
    const double eps = 1e-8;
    if (nb < 2 || nb > 11 || drop < 0 || drop >= nb) return -1;
    double dx[66], r[66];
    for (int bk = 0; bk < nb; bk++) {
        const double *a = x + 7 * bk, *a0 = x0 + 7 * bk;
        for (int k = 0; k < 3; k++) dx[6 * bk + k] = a[k] - a0[k];
        const double n2 = a0[3] * a0[3] + a0[4] * a0[4] + a0[5] * a0[5] + a0[6] * a0[6];
        const double ix = -a0[3] / n2, iy = -a0[4] / n2, iz = -a0[5] / n2, iw = a0[6] / n2;
        const double qx = a[3], qy = a[4], qz = a[5], qw = a[6];
        const double rw = iw * qw - ix * qx - iy * qy - iz * qz;
        double rx = iw * qx + ix * qw + iy * qz - iz * qy, ry = iw * qy + iy * qw + iz * qx - ix * qz, rz = iw * qz + iz * qw + ix * qy - iy * qx;
        if (!(rw >= 0)) { rx = -rx; ry = -ry; rz = -rz; }
        dx[6 * bk + 3] = 2.0 * rx; dx[6 * bk + 4] = 2.0 * ry; dx[6 * bk + 5] = 2.0 * rz;
    }
    for (int i = 0; i < n0; i++) { double v = lin_r[i]; for (int k = 0; k < n0; k++) v += lin_J[i * n0 + k] * dx[k]; r[i] = v; }
    /* permutation: dropped block first, then the kept blocks in order */
    int perm[66];
    for (int c = 0; c < 6; c++) perm[c] = 6 * drop + c;
    for (int bk = 0, p = 6; bk < nb; bk++) { if (bk == drop) continue; for (int c = 0; c < 6; c++) perm[p++] = 6 * bk + c; }
    double *H = (double *)calloc((size_t)n0 * n0, sizeof(double)), b[66];
    for (int i = 0; i < n0; i++) {
        for (int j = 0; j < n0; j++) { double v = 0; for (int k = 0; k < n0; k++) v += lin_J[k * n0 + perm[i]] * lin_J[k * n0 + perm[j]]; H[i * n0 + j] = v; }
        double v = 0; for (int k = 0; k < n0; k++) v += lin_J[k * n0 + perm[i]] * r[k];
        b[i] = v;
    }
    double Hmm[36], w[66], Vm[36], Hinv[36] = { 0 };
    for (int i = 0; i < m; i++) for (int j = 0; j < m; j++) Hmm[i * m + j] = 0.5 * (H[i * n0 + j] + H[j * n0 + i]);
    jacobi_eig(Hmm, m, w, Vm);
    for (int k = 0; k < m; k++) {
        if (!(w[k] > eps)) continue;
        for (int i = 0; i < m; i++) for (int j = 0; j < m; j++) Hinv[i * m + j] += Vm[i * m + k] / w[k] * Vm[j * m + k];
    }
    double *T = (double *)malloc(sizeof(double) * (size_t)n * m), *Hr = (double *)malloc(sizeof(double) * (size_t)n * n), *V = (double *)malloc(sizeof(double) * (size_t)n * n), br[66];
    for (int i = 0; i < n; i++) for (int j = 0; j < m; j++) { double v = 0; for (int k = 0; k < m; k++) v += H[(m + i) * n0 + k] * Hinv[k * m + j]; T[i * m + j] = v; }
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) { double v = 0; for (int k = 0; k < m; k++) v += T[i * m + k] * H[k * n0 + m + j]; Hr[i * n + j] = H[(m + i) * n0 + m + j] - v; }
        double v = 0; for (int k = 0; k < m; k++) v += T[i * m + k] * b[k];
        br[i] = b[m + i] - v;
    }
    jacobi_eig(Hr, n, w, V);
    for (int k = 0; k < n; k++) {
        const double S = w[k] > eps ? w[k] : 0.0, Si = w[k] > eps ? 1.0 / w[k] : 0.0;
        double vb = 0;
        for (int i = 0; i < n; i++) { out_J[k * n + i] = sqrt(S) * V[i * n + k]; vb += V[i * n + k] * br[i]; }
        out_r[k] = sqrt(Si) * vb;
    }
    free(H); free(T); free(Hr); free(V);
    return 0;
}
