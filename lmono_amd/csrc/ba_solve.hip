// lmono_amd/csrc/ba_solve.hip -- sliding-window BA solve on gfx950 (fp64): one 256-thread workgroup per window,
// windows batched across the grid (independent sequences / streams; a window depends on its predecessor, so one
// sequence alone cannot batch -- SURVEY.md 8e).
//
// Restates the solve inside Estimator::optimization() (/root/reference/mono_lidar_mapping/src/image_process/
// Estimator.cc:1124-1305): PriorFactor on the extrinsic, LASERFactor chain, MonoProjectionFactor + CauchyLoss(1) per
// (feature, observation), ceres::Solve(DENSE_SCHUR, DOGLEG, max_num_iterations) with Ceres defaults (Appendix B of
// SURVEY.md): Jacobi scaling, traditional dogleg on the elliptical trust region, Schur elimination of the 1-D inverse
// depth blocks into the 72x72 reduced system, Cholesky, step acceptance and radius update.
//
// Data flow per iteration: one thread per feature evaluates its observations (ba::mono_factor) and keeps the depth
// block (H_ff, g_f) private, camera blocks go into the LDS-resident 72x72 H_pp with LDS double atomics, the coupling
// columns H_pf into an HBM scratch (72 x F, L2-resident); the Schur complement is accumulated from 32-feature tiles
// staged through LDS; Cholesky and the triangular solves run column-parallel in LDS.  The 72x72 system is far too
// small for MFMA to matter at one window per workgroup: plain fp64 FMA.
#include "common.hpp"

namespace lmono {

constexpr int kBaMaxPoses = 11;
constexpr int kBaP = 6 * (kBaMaxPoses + 1);      // 72
constexpr int kBaMaxFeat = 448;
constexpr int kBaN = kBaP + kBaMaxFeat;          // 520
constexpr int kBaTile = 32;

struct BaBatch {
    int n_windows;
    int max_iter;
    const int *feat_off;        // [W+1]
    const int *obs_off;         // [W+1]
    const int *flags;           // [W][4] n_poses, use_prior, ex_constant, use_mono
    double *poses;              // [W][11][7]
    double *ex;                 // [W][7]
    double *inv_depth;          // [total F]
    const int *obs_feat;        // [total O] window-local feature index, grouped by feature
    const int *obs_i, *obs_j;   // [total O]
    const double *obs_pts;      // [total O][4]
    const int *feat_obs_off;    // [total F + 1] first observation of each feature (global index)
    const double *laser_consts; // [W][10][24]
    const double *prior_T;      // [W][16]
    const double *info;         // laser_info[36], mono_info[4], prior_w[2]
    double *hpd;                // scratch [W][72][kBaMaxFeat]
    double *cand;               // scratch [W][ (11+1)*7 + kBaMaxFeat ]
    double *summary;            // [W][6] initial_cost, final_cost, iterations, termination, successful, unsuccessful
};

struct BaLds {
    double Hpp[kBaP * kBaP];
    double S[kBaP * kBaP];
    double tile[kBaP * kBaTile];
    double Hdd[kBaMaxFeat], gdd[kBaMaxFeat];
    double gp[kBaP];
    double scale[kBaN], D[kBaN], D2[kBaN], gs[kBaN], gdv[kBaN], gn[kBaN], step[kBaN], tmp[kBaN], tmp2[kBaN];
    double rhs[kBaP];
    double red[8];
    double poses[kBaMaxPoses * 7], ex[7];
    double cposes[kBaMaxPoses * 7], cex[7];
    int ok;
};

__device__ __forceinline__ double block_sum(double v, double *red)
{
    v = wave_sum_d(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((red[0] + red[1]) + red[2]) + red[3];
}
__device__ __forceinline__ double block_max(double v, double *red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}

__device__ __forceinline__ void lds_add_block(double *H, const double *Ja, int oa, const double *Jb, int ob, int nr)
{
    for (int a = 0; a < 6; a++)
        for (int b = 0; b < 6; b++) {
            double v = 0;
            for (int r = 0; r < nr; r++) v += Ja[r * 7 + a] * Jb[r * 7 + b];
            atomicAdd(&H[(oa + a) * kBaP + ob + b], v);
            if (oa != ob) atomicAdd(&H[(ob + b) * kBaP + oa + a], v);
        }
}

struct BaCtx {
    int n_poses, use_prior, ex_constant, use_mono, F, P, ex_off;
    int f0, o0;
};
__device__ __forceinline__ int ba_pose_off(const BaCtx &c, int i) { return (c.ex_off < 0 ? 0 : 6) + 6 * i; }

// cost (returned to every thread) and, when kJac, the unscaled normal equations: Hpp, gp, Hdd, gdd in LDS, Hpd in HBM
template <bool kJac>
__device__ double ba_evaluate(const BaBatch &B, const BaCtx &c, BaLds &L, const double *poses, const double *ex, const double *invd, double *hpd)
{
    const int tid = threadIdx.x;
    if (kJac) {
        for (int k = tid; k < kBaP * kBaP; k += 256) L.Hpp[k] = 0.0;
        for (int k = tid; k < kBaP; k += 256) L.gp[k] = 0.0;
        for (int k = tid; k < c.F; k += 256) { L.Hdd[k] = 0.0; L.gdd[k] = 0.0; }
        for (int k = tid; k < c.P * c.F; k += 256) hpd[(k / c.F) * kBaMaxFeat + (k % c.F)] = 0.0;
        __syncthreads();
    }
    double cost = 0.0;
    const double *laser_info = B.info, *mono_info = B.info + 36, *prior_w = B.info + 40;
    if (tid < c.n_poses - 1) {
        double prm[14], r[6], J[84];
        for (int k = 0; k < 7; k++) { prm[k] = poses[7 * tid + k]; prm[7 + k] = poses[7 * (tid + 1) + k]; }
        ba::laser_factor(prm, B.laser_consts + ((size_t)blockIdx.x * 10 + tid) * 24, laser_info, r, kJac ? J : nullptr);
        for (int k = 0; k < 6; k++) cost += 0.5 * r[k] * r[k];
        if (kJac) {
            const int oi = ba_pose_off(c, tid), oj = ba_pose_off(c, tid + 1);
            lds_add_block(L.Hpp, J, oi, J, oi, 6); lds_add_block(L.Hpp, J, oi, J + 42, oj, 6); lds_add_block(L.Hpp, J + 42, oj, J + 42, oj, 6);
            for (int a = 0; a < 6; a++) {
                double gi = 0, gj = 0;
                for (int k = 0; k < 6; k++) { gi += J[k * 7 + a] * r[k]; gj += J[42 + k * 7 + a] * r[k]; }
                atomicAdd(&L.gp[oi + a], gi); atomicAdd(&L.gp[oj + a], gj);
            }
        }
    }
    if (tid == 32 && c.use_prior && !c.ex_constant) {
        double r[6], J[42];
        ba::prior_factor(ex, B.prior_T + (size_t)blockIdx.x * 16, prior_w, r, kJac ? J : nullptr);
        for (int k = 0; k < 6; k++) cost += 0.5 * r[k] * r[k];
        if (kJac) {
            lds_add_block(L.Hpp, J, c.ex_off, J, c.ex_off, 6);
            for (int a = 0; a < 6; a++) { double g = 0; for (int k = 0; k < 6; k++) g += J[k * 7 + a] * r[k]; atomicAdd(&L.gp[c.ex_off + a], g); }
        }
    }
    // one thread per feature: its observations are contiguous
    for (int f = tid; f < c.F; f += 256) {
        const int ob = B.feat_obs_off[c.f0 + f], oe = B.feat_obs_off[c.f0 + f + 1];
        double hff = 0.0, gf = 0.0, hx[6] = { 0, 0, 0, 0, 0, 0 }, hi[6] = { 0, 0, 0, 0, 0, 0 };
        int anchor = -1;
        for (int o = ob; o < oe; o++) {
            const int i = B.obs_i[o], j = B.obs_j[o];
            anchor = i;
            double prm[22], r[2], J[44];
            for (int k = 0; k < 7; k++) { prm[k] = ex[k]; prm[7 + k] = poses[7 * i + k]; prm[14 + k] = poses[7 * j + k]; }
            prm[21] = invd[f];
            ba::mono_factor(prm, B.obs_pts + (size_t)o * 4, mono_info, r, kJac ? J : nullptr);
            const double sq = r[0] * r[0] + r[1] * r[1];
            // ceres::CauchyLoss(1): rho = log(1 + s), rho' = 1 / (1 + s), rho'' < 0 -> corrector scales by sqrt(rho')
            cost += 0.5 * log(1.0 + sq);
            if (!kJac) continue;
            const double inv = 1.0 / (1.0 + sq);
            const double sr = sqrt(inv > DBL_MIN ? inv : DBL_MIN);
            for (int k = 0; k < 44; k++) J[k] *= sr;
            r[0] *= sr; r[1] *= sr;
            const int oi = ba_pose_off(c, i), oj = ba_pose_off(c, j);
            const double *Jx = J, *Ji = J + 14, *Jj = J + 28, *Jd = J + 42;
            if (c.ex_off >= 0) { lds_add_block(L.Hpp, Jx, c.ex_off, Jx, c.ex_off, 2); lds_add_block(L.Hpp, Jx, c.ex_off, Ji, oi, 2); lds_add_block(L.Hpp, Jx, c.ex_off, Jj, oj, 2); }
            lds_add_block(L.Hpp, Ji, oi, Ji, oi, 2); lds_add_block(L.Hpp, Ji, oi, Jj, oj, 2); lds_add_block(L.Hpp, Jj, oj, Jj, oj, 2);
            for (int a = 0; a < 6; a++) {
                if (c.ex_off >= 0) { atomicAdd(&L.gp[c.ex_off + a], Jx[a] * r[0] + Jx[7 + a] * r[1]); hx[a] += Jx[a] * Jd[0] + Jx[7 + a] * Jd[1]; }
                atomicAdd(&L.gp[oi + a], Ji[a] * r[0] + Ji[7 + a] * r[1]);
                atomicAdd(&L.gp[oj + a], Jj[a] * r[0] + Jj[7 + a] * r[1]);
                hi[a] += Ji[a] * Jd[0] + Ji[7 + a] * Jd[1];
                hpd[(size_t)(oj + a) * kBaMaxFeat + f] = Jj[a] * Jd[0] + Jj[7 + a] * Jd[1];   // frame j is observed once per feature
            }
            hff += Jd[0] * Jd[0] + Jd[1] * Jd[1];
            gf += Jd[0] * r[0] + Jd[1] * r[1];
        }
        if (kJac) {
            L.Hdd[f] = hff; L.gdd[f] = gf;
            if (anchor >= 0) {
                const int oi = ba_pose_off(c, anchor);
                for (int a = 0; a < 6; a++) {
                    if (c.ex_off >= 0) hpd[(size_t)(c.ex_off + a) * kBaMaxFeat + f] = hx[a];
                    hpd[(size_t)(oi + a) * kBaMaxFeat + f] = hi[a];
                }
            }
        }
    }
    return block_sum(cost, L.red);
}

// y = Hs v (Jacobi-scaled), v and y in LDS arrays of length N
__device__ void ba_hs_mul(const BaCtx &c, BaLds &L, const double *hpd, const double *v, double *y)
{
    const int tid = threadIdx.x;
    __syncthreads();
    for (int a = tid; a < c.P; a += 256) {
        double acc = 0;
        for (int b = 0; b < c.P; b++) acc += L.Hpp[a * kBaP + b] * L.scale[b] * v[b];
        for (int f = 0; f < c.F; f++) acc += hpd[(size_t)a * kBaMaxFeat + f] * L.scale[c.P + f] * v[c.P + f];
        y[a] = acc * L.scale[a];
    }
    for (int f = tid; f < c.F; f += 256) {
        double acc = L.Hdd[f] * L.scale[c.P + f] * v[c.P + f];
        for (int a = 0; a < c.P; a++) acc += hpd[(size_t)a * kBaMaxFeat + f] * L.scale[a] * v[a];
        y[c.P + f] = acc * L.scale[c.P + f];
    }
    __syncthreads();
}

// solve (Hs + mu diag(D2)) x = gs by Schur elimination of the depth columns; result in L.gn; returns success to all
__device__ bool ba_schur_solve(const BaCtx &c, BaLds &L, const double *hpd, double mu)
{
    const int tid = threadIdx.x, P = c.P, F = c.F;
    __syncthreads();
    for (int k = tid; k < P * P; k += 256) {
        const int a = k / P, b = k % P;
        L.S[a * kBaP + b] = L.Hpp[a * kBaP + b] * L.scale[a] * L.scale[b] + (a == b ? mu * L.D2[a] : 0.0);
    }
    for (int a = tid; a < P; a += 256) L.rhs[a] = L.gs[a];
    if (tid == 0) L.ok = 1;
    __syncthreads();
    for (int f0 = 0; f0 < F; f0 += kBaTile) {
        const int nf = min(kBaTile, F - f0);
        // tile[a][t] = scaled coupling of camera parameter a with feature f0 + t
        for (int k = tid; k < P * kBaTile; k += 256) {
            const int a = k / kBaTile, t = k % kBaTile;
            L.tile[k] = t < nf ? hpd[(size_t)a * kBaMaxFeat + f0 + t] * L.scale[a] * L.scale[P + f0 + t] : 0.0;
        }
        __syncthreads();
        for (int k = tid; k < P * P; k += 256) {
            const int a = k / P, b = k % P;
            double acc = 0;
            for (int t = 0; t < nf; t++) {
                const double hff = L.Hdd[f0 + t] * L.scale[P + f0 + t] * L.scale[P + f0 + t] + mu * L.D2[P + f0 + t];
                acc += L.tile[a * kBaTile + t] * L.tile[b * kBaTile + t] / hff;
            }
            L.S[a * kBaP + b] -= acc;
        }
        for (int a = tid; a < P; a += 256) {
            double acc = 0;
            for (int t = 0; t < nf; t++) {
                const double hff = L.Hdd[f0 + t] * L.scale[P + f0 + t] * L.scale[P + f0 + t] + mu * L.D2[P + f0 + t];
                if (!(hff > 0.0)) L.ok = 0;
                acc += L.tile[a * kBaTile + t] * L.gs[P + f0 + t] / hff;
            }
            L.rhs[a] -= acc;
        }
        __syncthreads();
    }
    // in-place lower Cholesky of S (column by column), then forward / backward substitution, all in LDS
    for (int j = 0; j < P; j++) {
        if (tid == 0) {
            double s = L.S[j * kBaP + j];
            for (int k = 0; k < j; k++) s -= L.S[j * kBaP + k] * L.S[j * kBaP + k];
            if (!(s > 0.0)) { L.ok = 0; s = 1.0; }
            L.S[j * kBaP + j] = sqrt(s);
        }
        __syncthreads();
        for (int i = j + 1 + tid; i < P; i += 256) {
            double s = L.S[i * kBaP + j];
            for (int k = 0; k < j; k++) s -= L.S[i * kBaP + k] * L.S[j * kBaP + k];
            L.S[i * kBaP + j] = s / L.S[j * kBaP + j];
        }
        __syncthreads();
    }
    for (int k = 0; k < P; k++) {
        if (tid == 0) L.rhs[k] = L.rhs[k] / L.S[k * kBaP + k];
        __syncthreads();
        for (int i = k + 1 + tid; i < P; i += 256) L.rhs[i] -= L.S[i * kBaP + k] * L.rhs[k];
        __syncthreads();
    }
    for (int k = P - 1; k >= 0; k--) {
        if (tid == 0) L.rhs[k] = L.rhs[k] / L.S[k * kBaP + k];
        __syncthreads();
        for (int i = tid; i < k; i += 256) L.rhs[i] -= L.S[k * kBaP + i] * L.rhs[k];
        __syncthreads();
    }
    for (int a = tid; a < P; a += 256) { L.gn[a] = L.rhs[a]; if (!isfinite(L.rhs[a])) L.ok = 0; }
    __syncthreads();
    for (int f = tid; f < F; f += 256) {
        const double hff = L.Hdd[f] * L.scale[P + f] * L.scale[P + f] + mu * L.D2[P + f];
        double acc = L.gs[P + f];
        for (int a = 0; a < P; a++) acc -= hpd[(size_t)a * kBaMaxFeat + f] * L.scale[a] * L.scale[P + f] * L.gn[a];
        const double x = acc / hff;
        L.gn[P + f] = x;
        if (!isfinite(x)) L.ok = 0;
    }
    __syncthreads();
    return L.ok != 0;
}

__global__ __launch_bounds__(256) void k_ba_solve(BaBatch B)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    BaLds &L = *reinterpret_cast<BaLds *>(smem_raw);
    const int w = blockIdx.x, tid = threadIdx.x;
    BaCtx c;
    c.n_poses = B.flags[w * 4 + 0]; c.use_prior = B.flags[w * 4 + 1]; c.ex_constant = B.flags[w * 4 + 2]; c.use_mono = B.flags[w * 4 + 3];
    c.f0 = B.feat_off[w]; c.o0 = B.obs_off[w];
    c.F = c.use_mono ? B.feat_off[w + 1] - c.f0 : 0;
    c.ex_off = c.ex_constant ? -1 : 0;
    c.P = 6 * c.n_poses + (c.ex_constant ? 0 : 6);
    const int P = c.P, F = c.F, N = P + F;
    double *gposes = B.poses + (size_t)w * kBaMaxPoses * 7, *gex = B.ex + (size_t)w * 7, *ginvd = B.inv_depth + c.f0;
    double *hpd = B.hpd + (size_t)w * kBaP * kBaMaxFeat;
    double *cinvd = B.cand + (size_t)w * kBaMaxFeat;
    for (int k = tid; k < c.n_poses * 7; k += 256) L.poses[k] = gposes[k];
    if (tid < 7) L.ex[tid] = gex[tid];
    __syncthreads();

    const double function_tol = 1e-6, gradient_tol = 1e-10, parameter_tol = 1e-8, min_rel_decrease = 1e-3;
    const double min_diag = 1e-6, max_diag = 1e32, max_radius = 1e16, min_radius = 1e-32;
    const double min_mu = 1e-8, max_mu = 1.0, mu_inc = 10.0;
    double radius = 1e4, mu = min_mu, alpha = 0.0, dogleg_norm = 0.0;
    bool reuse = false;
    int invalid = 0, iter = 0, termination = 1, n_succ = 0, n_unsucc = 0;

    double x_cost = ba_evaluate<true>(B, c, L, L.poses, L.ex, ginvd, hpd);
    const double initial_cost = x_cost;
    auto global_norm = [&](const double *poses, const double *ex, const double *invd) {
        double q = 0;
        if (c.ex_off >= 0 && tid < 7) q += ex[tid] * ex[tid];
        for (int k = tid; k < 7 * c.n_poses; k += 256) q += poses[k] * poses[k];
        for (int f = tid; f < F; f += 256) q += invd[f] * invd[f];
        return sqrt(block_sum(q, L.red));
    };
    double x_norm = global_norm(L.poses, L.ex, ginvd);
    for (int a = tid; a < P; a += 256) L.scale[a] = 1.0 / (1.0 + sqrt(L.Hpp[a * kBaP + a]));
    for (int f = tid; f < F; f += 256) L.scale[P + f] = 1.0 / (1.0 + sqrt(L.Hdd[f]));
    auto grad_max = [&]() {
        double g = 0;
        for (int a = tid; a < P; a += 256) g = fmax(g, fabs(L.gp[a]));
        for (int f = tid; f < F; f += 256) g = fmax(g, fabs(L.gdd[f]));
        return block_max(g, L.red);
    };
    double gmax = grad_max();
    if (gmax <= gradient_tol) termination = 0;
    else while (iter < B.max_iter) {
        iter++;
        bool ok = true;
        if (!reuse) {
            __syncthreads();
            for (int k = tid; k < N; k += 256) {
                const double h = k < P ? L.Hpp[k * kBaP + k] : L.Hdd[k - P];
                const double g = k < P ? L.gp[k] : L.gdd[k - P];
                L.gs[k] = g * L.scale[k];
                double d = h * L.scale[k] * L.scale[k];
                d = d < min_diag ? min_diag : (d > max_diag ? max_diag : d);
                L.D2[k] = d; L.D[k] = sqrt(d);
                L.gdv[k] = L.gs[k] / L.D[k];
                L.tmp[k] = L.gdv[k] / L.D[k];
            }
            ba_hs_mul(c, L, hpd, L.tmp, L.tmp2);
            double g2 = 0, jg2 = 0;
            for (int k = tid; k < N; k += 256) { g2 += L.gdv[k] * L.gdv[k]; jg2 += L.tmp[k] * L.tmp2[k]; }
            g2 = block_sum(g2, L.red); jg2 = block_sum(jg2, L.red);
            alpha = g2 / jg2;
            ok = false;
            while (mu < max_mu) {
                if (ba_schur_solve(c, L, hpd, mu)) { ok = true; break; }
                mu *= mu_inc;
            }
            if (ok) {
                mu = fmax(min_mu, 2.0 * mu / mu_inc);
                for (int k = tid; k < N; k += 256) L.gn[k] *= -L.D[k];
            }
            __syncthreads();
        }
        double model_change = 0.0;
        if (ok) {
            double a2 = 0, b2 = 0, ab = 0;
            for (int k = tid; k < N; k += 256) { a2 += L.gn[k] * L.gn[k]; b2 += L.gdv[k] * L.gdv[k]; ab += L.gdv[k] * L.gn[k]; }
            const double gn_norm = sqrt(block_sum(a2, L.red)), g_norm = sqrt(block_sum(b2, L.red));
            ab = block_sum(ab, L.red);
            double ca, cb;   // step = ca * gdv + cb * gn
            if (gn_norm <= radius) { ca = 0.0; cb = 1.0; dogleg_norm = gn_norm; }
            else if (alpha * g_norm >= radius) { ca = -(radius / g_norm); cb = 0.0; dogleg_norm = radius; }
            else {
                const double b_dot_a = -alpha * ab;
                const double aa = alpha * alpha * g_norm * g_norm;
                const double bma2 = aa - 2 * b_dot_a + gn_norm * gn_norm;
                const double cc = b_dot_a - aa;
                const double d = sqrt(cc * cc + bma2 * (radius * radius - aa));
                const double beta = (cc <= 0) ? (d - cc) / bma2 : (radius * radius - aa) / (d + cc);
                ca = -alpha * (1.0 - beta); cb = beta; dogleg_norm = radius;
            }
            __syncthreads();
            for (int k = tid; k < N; k += 256) L.step[k] = (ca * L.gdv[k] + cb * L.gn[k]) / L.D[k];
            ba_hs_mul(c, L, hpd, L.step, L.tmp);
            double dg = 0, dHd = 0;
            for (int k = tid; k < N; k += 256) { dg += L.step[k] * L.gs[k]; dHd += L.step[k] * L.tmp[k]; }
            dg = block_sum(dg, L.red); dHd = block_sum(dHd, L.red);
            model_change = -(dg + 0.5 * dHd);
        }
        if (!ok || !(model_change > 0.0)) {
            if (++invalid >= 5) { termination = 2; break; }
            mu *= mu_inc; reuse = false;
            continue;
        }
        invalid = 0;
        // candidate = Plus(x, step * scale)
        __syncthreads();
        if (tid <= c.n_poses) {
            const bool is_ex = tid == c.n_poses;
            if (!is_ex || c.ex_off >= 0) {
                const int off = is_ex ? c.ex_off : ba_pose_off(c, tid);
                double d6[6];
                for (int a = 0; a < 6; a++) d6[a] = L.step[off + a] * L.scale[off + a];
                ba::pose_plus(is_ex ? L.ex : L.poses + 7 * tid, d6, is_ex ? L.cex : L.cposes + 7 * tid);
            } else {
                for (int k = 0; k < 7; k++) L.cex[k] = L.ex[k];
            }
        }
        for (int f = tid; f < F; f += 256) cinvd[f] = ginvd[f] + L.step[P + f] * L.scale[P + f];
        __syncthreads();
        const double cand_cost = ba_evaluate<false>(B, c, L, L.cposes, L.cex, cinvd, hpd);
        double dq = 0;
        if (c.ex_off >= 0 && tid < 7) dq += (L.ex[tid] - L.cex[tid]) * (L.ex[tid] - L.cex[tid]);
        for (int k = tid; k < 7 * c.n_poses; k += 256) dq += (L.poses[k] - L.cposes[k]) * (L.poses[k] - L.cposes[k]);
        for (int f = tid; f < F; f += 256) dq += (ginvd[f] - cinvd[f]) * (ginvd[f] - cinvd[f]);
        const double sn = sqrt(block_sum(dq, L.red));
        if (sn <= parameter_tol * (x_norm + parameter_tol)) { termination = 0; break; }
        if (fabs(x_cost - cand_cost) <= function_tol * x_cost) { termination = 0; break; }
        const double rel = (x_cost - cand_cost) / model_change;
        if (rel > min_rel_decrease) {
            __syncthreads();
            for (int k = tid; k < 7 * c.n_poses; k += 256) L.poses[k] = L.cposes[k];
            if (tid < 7) L.ex[tid] = L.cex[tid];
            for (int f = tid; f < F; f += 256) ginvd[f] = cinvd[f];
            __syncthreads();
            x_norm = global_norm(L.poses, L.ex, ginvd);
            x_cost = ba_evaluate<true>(B, c, L, L.poses, L.ex, ginvd, hpd);
            n_succ++;
            if (rel < 0.25) radius *= 0.5;
            if (rel > 0.75) radius = fmax(radius, 3.0 * dogleg_norm);
            if (radius > max_radius) radius = max_radius;
            reuse = false;
            gmax = grad_max();
            if (gmax <= gradient_tol) { termination = 0; break; }
        } else {
            radius *= 0.5; reuse = true;
            n_unsucc++;
        }
        if (radius <= min_radius) { termination = 0; break; }
    }
    __syncthreads();
    for (int k = tid; k < c.n_poses * 7; k += 256) gposes[k] = L.poses[k];
    if (tid < 7) gex[tid] = L.ex[tid];
    if (tid == 0) {
        double *sm = B.summary + (size_t)w * 6;
        sm[0] = initial_cost; sm[1] = x_cost; sm[2] = iter; sm[3] = termination; sm[4] = n_succ; sm[5] = n_unsucc;
    }
}

} // namespace lmono
