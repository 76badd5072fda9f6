// lmono_amd/csrc/marg.hip -- marginalisation prior of lmono's Estimator on gfx950 (fp64), one workgroup per window.
// Restates the MARGIN_OLD branch of Estimator::margin() (/root/reference/mono_lidar_mapping/src/image_process/
// Estimator.cc:1307-1405) with ResidualBlockInfo::Evaluate, MarginalizationInfo::marginalize and
// Marginalization::Evaluate (src/factor/MarginalizationFactor.cc:18-68, :176-272, :309-373).
//
// Blocks: m = [pose0 (6), inverse depths of the F0 tracks anchored at frame 0], n = [ex, pose1..pose10] = 66.
// H_mm = [[A, B], [B^T, D]] with D diagonal (each MonoProjectionFactor touches one depth), so
//   H' = H_rr - G^T S_A^+ G - W_d^T D^+ W_d,   G = W_p - B D^+ W_d,   S_A = A - B D^+ B^T   (6x6)
// (W_p / W_d = pose0 / depth rows of H_mr).  The reference inverts the full H_mm through an eigen-decomposition with
// an eps = 1e-8 cut; the structured form applies the same cut to D and to S_A and agrees with it to rounding whenever
// H_mm has no eigenvalue near eps (every depth observed, pose0 held by the LASERFactor) -- the well-posed case the
// parity tests cover; a degenerate H_mm is flagged in the status word.  The 66x66 eigen-decomposition of H' that
// yields linearized_jacobians = sqrt(S) V^T and linearized_residuals = sqrt(S^-1) V^T b' is a Householder tridiagonalisation +
// implicit QL in LDS (marg_eig_ql; rounds 2-5: a parallel round-robin Jacobi iteration).  As in the reference the result is a prior that Estimator::optimization never consumes
// (MarginalizationInfo::valid stays false).
#include "common.hpp"

namespace lmono {

constexpr int kMargN = 66;
#ifndef LMONO_MG_T
#define LMONO_MG_T 512      // 256 / 512 / 1024: Estimator loop 283 / 287 / 287 frames/s with overlapped, 201 / 220 / 213 with inline marginalisation
#endif
constexpr int kMgT = LMONO_MG_T;       // threads of the two marginalisation kernels
constexpr int kMargMaxF0 = 160;        // >= the tracker's MAX_CNT = 150 features per frame (FeatureTracker.cc:21), all of which can be anchored at frame 0

struct MargBatch {
    int n_windows;
    const int *feat_off;        // [W+1] tracks anchored at frame 0
    const int *obs_off;         // [W+1]
    const double *poses;        // [W][11][7]
    const double *ex;           // [W][7]
    const double *inv_depth;    // [total F0]
    const int *feat_obs_off;    // [total F0 + 1]
    const int *obs_j;           // [total O]
    const double *obs_pts;      // [total O][4]
    const double *laser01;      // [W][24]
    const double *info;         // laser_info[36], mono_info[4]
    double *lin_J;              // [W][66*66]
    double *lin_r;              // [W][66]
    int *status;                // [W] bit 0: H_mm degenerate (eps cut applied)
};

// Factor pass (round 5): every sum that many tracks share is reduced inside the wave (butterfly: deterministic) and added by ONE lane to the wave's own
// slot array; the slots of the waves are added in wave order afterwards.  Until then those sums were LDS atomics on a few dozen addresses that every
// lane of every wave hit at once: 3.5 M cycles for 43 tracks (`scripts/prof_marg.sh`), and an order of additions that depended on timing.
// Slot layout per wave: common [90] = bp[6] | br_ex[6] | A (upper triangle, 21) | Wp_ex[36] | H_ex,ex (21); then per frame j = 1..10 [99] =
// br_j[6] | Wp_j[36] | H_ex,j[36] | H_jj (21).
constexpr int kMgAccC = 90, kMgAccJ = 99, kMgAcc = kMgAccC + 10 * kMgAccJ;
constexpr int kMgAccWaves = (kMargMaxF0 + 63) / 64;

struct MargEigWork {
    double d[kMargN], e[kMargN], p[kMargN], u[kMargN];
    double cs[2][2 * kMargN];        // rotation lists of two sweeps in flight: c_i, s_i for i = hi - 1 .. lo
    double part[(kMgT / 64) * kMargN];   // per-wave partial dot products of the accumulation phase
    double sc[4];                    // [0] h of the current reflector, [1] K
    int ctl[2][4];                   // per list: lo, hi (rotations i = hi - 1 .. lo; hi <= lo: none), done
    int fail;
#ifdef LMONO_MG_PROF
    unsigned long long prof[5]; int n_sweeps, n_rot;     // cycles of the three phases, QL sweeps and rotations; [3] wave 0's rotation chains, [4] its barrier waits
#endif
};
struct MargLds {
    double Hrr[kMargN * kMargN];
    union {                             // the eigenvectors are only formed after the last use of W_d
        double Wd[kMargMaxF0 * kMargN]; // depth rows of H_mr
        double V[kMargN * kMargN];
    };
    double Wp[6 * kMargN];              // pose0 rows of H_mr, later G
    double B[6 * kMargMaxF0];
    double D[kMargMaxF0], bd[kMargMaxF0];
    double A[36], bp[6], br[kMargN];
    union {
        struct {                        // Schur complement and eigen-decomposition
            double Y[6 * kMargN];
            double SAi[36], u[6];
            MargEigWork eig;
        };
        struct {                        // factor pass
            double acc[kMgAccWaves * kMgAcc];
            double lz[120 + 90];        // LASERFactor(0, 1): g0[6] | g1[6] | J0^T J0 [36] | J0^T J1 [36] | J1^T J1 [36]; then its r[6] | J[84] while they are formed
        };
    };
    int flag;
};
static_assert(sizeof(MargLds) <= 160 * 1024, "k_marginalize: LDS");

__device__ __forceinline__ int tri6(int a, int b) { const int lo = a < b ? a : b, hi = a < b ? b : a; return lo * (11 - lo) / 2 + hi; }
// the wave's sum of v goes to slot idx of the wave's array (one writer per array: the additions to a slot happen in program order)
__device__ __forceinline__ void mg_acc(double *acc, int idx, double v, int lane)
{
    const double t = wave_sum_d_lane63(v);
    if (lane == 63) atomicAdd(acc + idx, t);          // ds_add_f64 without a return value: nothing waits for it
}
// symmetric 6x6 pseudo-inverse with the eps cut (serial Jacobi, one thread; every index is a compile-time constant after unrolling, so the two
// matrices live in registers -- with run-time indices they lived in scratch memory: ~0.4 M cycles per call)
__device__ void pinv6(const double *Ain, double *out, double eps, int *degenerate)
{
    double M[36], Vv[36];
#pragma unroll
    for (int i = 0; i < 36; i++) { M[i] = Ain[i]; Vv[i] = (i % 7 == 0) ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 40; sweep++) {
        double offn = 0;
#pragma unroll
        for (int p = 0; p < 6; p++)
#pragma unroll
            for (int q = p + 1; q < 6; q++) offn += M[p * 6 + q] * M[p * 6 + q];
        if (offn < 1e-300) break;
#pragma unroll
        for (int p = 0; p < 6; p++)
#pragma unroll
            for (int q = p + 1; q < 6; q++) {
                const double apq = M[p * 6 + q];
                if (apq != 0.0) {
                    double c, s;
                    jacobi_cs(M[p * 6 + p], M[q * 6 + q], apq, c, s);
#pragma unroll
                    for (int k = 0; k < 6; k++) { const double a = M[k * 6 + p], b = M[k * 6 + q]; M[k * 6 + p] = c * a - s * b; M[k * 6 + q] = s * a + c * b; }
#pragma unroll
                    for (int k = 0; k < 6; k++) { const double a = M[p * 6 + k], b = M[q * 6 + k]; M[p * 6 + k] = c * a - s * b; M[q * 6 + k] = s * a + c * b; }
#pragma unroll
                    for (int k = 0; k < 6; k++) { const double a = Vv[k * 6 + p], b = Vv[k * 6 + q]; Vv[k * 6 + p] = c * a - s * b; Vv[k * 6 + q] = s * a + c * b; }
                }
            }
    }
#pragma unroll
    for (int i = 0; i < 36; i++) out[i] = 0.0;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const double w = M[k * 6 + k];
        if (!(w > eps)) { *degenerate = 1; continue; }
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
            for (int j = 0; j < 6; j++) out[i * 6 + j] += Vv[i * 6 + k] * Vv[j * 6 + k] / w;
    }
}

// Symmetric eigen-decomposition of the n x n matrix H (LDS, leading dimension n, BOTH triangles valid; n <= kMargN) by Householder tridiagonalisation and
// implicit-shift QL (round 6; EISPACK tred2 / tql2 in the form of Numerical Recipes' tred2 / tqli, laid out for one workgroup of kMgT threads).
// On return W.d[e] is eigenvalue e (no particular order) and H[e * n + i] component i of its eigenvector -- the eigenvectors are the ROWS of H.
// Rounds 2-5 ran a round-robin parallel Jacobi here: ~20 sweeps x 65 rounds x (33 rotations applied to H and V through the LDS) = 2.2 M cycles of the
// kernel's 2.6 M (profiles/r5/marg_phase_cycles.txt), bound by LDS bandwidth.  This form moves ~30 x fewer bytes:
//   1. tred2: 65 reflectors, each a matrix-vector product and a rank-2 update of the leading block (all threads), three barriers per reflector;
//   2. the product of the reflectors accumulated in place (two barriers per step), then transposed in place so that an eigenvector is a row;
//   3. QL: wave 0 runs the scalar recurrence of a sweep (rotations c_i, s_i into a list) while the other waves apply the PREVIOUS sweep's list to the
//      eigenvector rows -- one thread per component, a rotation touches rows i and i + 1, consecutive threads consecutive addresses; one barrier per sweep.
// Deterministic (no atomics, fixed association); the eps cut and everything downstream see eigen-pairs accurate to rounding like the Jacobi's.
__device__ __forceinline__ double mg_readlane_d(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
// the wave's sum in every lane: DPP row steps + one readlane (the ds_bpermute butterfly of wave_sum_d is ~1 k cycles of LDS-crossbar round trips)
__device__ __forceinline__ double mg_wave_sum(double v) { return mg_readlane_d(wave_sum_d_lane63(v), 63); }
// sum over the 8 lanes of this lane's group of eight, in every lane of the group: quad_perm [1,0,3,2], [2,3,0,1], then row_half_mirror
__device__ __forceinline__ double mg_sum8(double v)
{
    v += dpp_mov_f64<0xB1, 0xf>(v);
    v += dpp_mov_f64<0x4E, 0xf>(v);
    v += dpp_mov_f64<0x141, 0xf>(v);
    return v;
}
// entry i (wave-uniform) of a vector kept one entry per lane in two registers (v0: entries 0..63, v1: 64..127)
__device__ __forceinline__ double mg_get2(double v0, double v1, int i) { const double a = mg_readlane_d(v0, i & 63), b = mg_readlane_d(v1, i & 63); return i < 64 ? a : b; }     // (both lanes read, a scalar select: no branch)
__device__ __forceinline__ void mg_set2(double &v0, double &v1, int i, double x, int lane) { if (lane == i) v0 = x; if (64 + lane == i) v1 = x; }

__device__ __noinline__ void marg_eig_ql(double *A, int n, MargEigWork &W, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
    constexpr int kWv = kMgT / 64;
#ifdef LMONO_MG_PROF
    unsigned long long pt0 = __builtin_readcyclecounter(), p_chain = 0, p_bar = 0;
    int p_sweeps = 0, p_rot = 0;
#endif
    // ---- 1. Householder reduction to tridiagonal form: d, e; reflector i keeps u in row i and u / h in column i
    for (int i = n - 1; i >= 1; i--) {
        const int l = i - 1;
        if (wave == 0) {
            // (i <= 65 entries of row i: lanes k and 64 + k)
            const double a0 = lane < i ? A[i * n + lane] : 0.0, a1 = 64 + lane < i ? A[i * n + 64 + lane] : 0.0;
            double h = 0.0, ei = 0.0;
            if (l > 0) {
                const double scale = mg_wave_sum(fabs(a0) + fabs(a1));
                if (scale == 0.0) ei = mg_get2(a0, a1, l);
                else {
                    const double inv = 1.0 / scale;
                    double u0 = a0 * inv, u1 = a1 * inv;
                    h = mg_wave_sum(u0 * u0 + u1 * u1);
                    const double f = mg_get2(u0, u1, l);
                    const double g = f >= 0.0 ? -sqrt(h) : sqrt(h);
                    ei = scale * g;
                    h -= f * g;
                    mg_set2(u0, u1, l, f - g, lane);
                    const double ih = 1.0 / h;
                    if (lane < i) { A[i * n + lane] = u0; A[lane * n + i] = u0 * ih; W.u[lane] = u0; }
                    if (64 + lane < i) { A[i * n + 64 + lane] = u1; A[(64 + lane) * n + i] = u1 * ih; W.u[64 + lane] = u1; }
                }
            } else ei = mg_readlane_d(a0, 0);
            if (lane == 0) { W.e[i] = ei; W.d[i] = h; W.sc[0] = h; }
        }
        __syncthreads();
        const double h = W.sc[0];
        if (h != 0.0) {                                  // (uniform: every thread reads the same word)
            // p = A u / h over the leading i x i block: eight threads per row, four products of a thread in flight
            const double ih = 1.0 / h;
            for (int j = tid >> 3; j < i; j += kMgT / 8) {
                double acc = 0.0;
                for (int k = tid & 7; k < i; k += 32) {
                    const int k1 = k + 8, k2 = k + 16, k3 = k + 24;
                    const double x0 = A[j * n + k] * W.u[k], x1 = k1 < i ? A[j * n + k1] * W.u[k1] : 0.0, x2 = k2 < i ? A[j * n + k2] * W.u[k2] : 0.0, x3 = k3 < i ? A[j * n + k3] * W.u[k3] : 0.0;
                    acc += (x0 + x1) + (x2 + x3);
                }
                acc = mg_sum8(acc);
                if ((tid & 7) == 0) W.p[j] = acc * ih;
            }
            __syncthreads();
            // K = u . p / (2 h) (every wave for itself), A -= u q^T + q u^T with q = p - K u
            const double pu = (lane < i ? W.p[lane] * W.u[lane] : 0.0) + (64 + lane < i ? W.p[64 + lane] * W.u[64 + lane] : 0.0);
            const double Kk = mg_wave_sum(pu) / (h + h);
            const double uk0 = lane < i ? W.u[lane] : 0.0, uk1 = 64 + lane < i ? W.u[64 + lane] : 0.0;
            const double qk0 = lane < i ? W.p[lane] - Kk * uk0 : 0.0, qk1 = 64 + lane < i ? W.p[64 + lane] - Kk * uk1 : 0.0;
            for (int j = wave; j < i; j += kWv) {
                const double uj = W.u[j], qj = W.p[j] - Kk * uj;
                if (lane < i) A[j * n + lane] -= uj * qk0 + qj * uk0;
                if (64 + lane < i) A[j * n + 64 + lane] -= uj * qk1 + qj * uk1;
            }
        }
        __syncthreads();
    }
    if (tid == 0) { W.d[0] = 0.0; W.e[0] = 0.0; }
    __syncthreads();
#ifdef LMONO_MG_PROF
    if (tid == 0) { const unsigned long long t = __builtin_readcyclecounter(); W.prof[0] = t - pt0; pt0 = t; }
#endif
    // ---- 2. accumulate the transformation: A becomes Q (columns)
    for (int i = 0; i < n; i++) {
        const bool refl = W.d[i] != 0.0;                 // (uniform)
        if (refl) {
            // g_j = sum_k A[i][k] A[k][j], eight threads per j (k = part, part + 8, ..); the column i (u / h) is saved: the update below rewrites row and
            // column i beside it
            for (int j = tid & 63; j < i; j += 64) {     // lanes: consecutive j (consecutive addresses of a row k); the eight waves: the parts
                double acc = 0.0;
                for (int k = wave; k < i; k += 4 * kWv) {
                    const int k1 = k + kWv, k2 = k + 2 * kWv, k3 = k + 3 * kWv;
                    const double x0 = A[i * n + k] * A[k * n + j], x1 = k1 < i ? A[i * n + k1] * A[k1 * n + j] : 0.0, x2 = k2 < i ? A[i * n + k2] * A[k2 * n + j] : 0.0,
                                 x3 = k3 < i ? A[i * n + k3] * A[k3 * n + j] : 0.0;
                    acc += (x0 + x1) + (x2 + x3);
                }
                W.part[wave * kMargN + j] = acc;
            }
            if (tid < i) W.u[tid] = A[tid * n + i];
        }
        __syncthreads();
        if (refl && tid < i) {
            double g = 0.0;
#pragma unroll
            for (int wv = 0; wv < kWv; wv++) g += W.part[wv * kMargN + tid];
            W.p[tid] = g;
        }
        __syncthreads();
        for (int k = wave; k <= i; k += kWv)
            for (int j = lane; j <= i; j += 64) {
                if (k == i && j == i) { W.d[i] = A[i * n + i]; A[i * n + i] = 1.0; }
                else if (k == i || j == i) A[k * n + j] = 0.0;
                else if (refl) A[k * n + j] -= W.p[j] * W.u[k];
            }
        __syncthreads();
    }
    // transpose in place: row e becomes eigenvector e's storage
    for (int k = wave; k < n; k += kWv)
        for (int j = lane; j < k; j += 64) { const double x = A[k * n + j], y = A[j * n + k]; A[k * n + j] = y; A[j * n + k] = x; }
    if (tid < n) W.u[tid] = tid + 1 < n ? W.e[tid + 1] : 0.0;       // e shifted down by one
    if (tid == 0) { W.ctl[0][0] = W.ctl[0][1] = W.ctl[0][2] = 0; W.ctl[1][0] = W.ctl[1][1] = W.ctl[1][2] = 0; W.fail = 0; }
    __syncthreads();
    if (tid < n) W.e[tid] = W.u[tid];
    __syncthreads();
#ifdef LMONO_MG_PROF
    if (tid == 0) { const unsigned long long t = __builtin_readcyclecounter(); W.prof[1] = t - pt0; pt0 = t; }
#endif
    // ---- 3. QL with implicit shifts.  Wave 0: the scalar recurrences (every lane computes the same values; d, e and the rotation list in LDS: a step's
    //         operands are requested one step ahead, its results stored by lane 0 and never waited for -- the version with d and e spread over the lanes'
    //         registers spent 2/3 of a step on v_readlane / v_cndmask: 627 cycles per rotation).  Waves 1..: the eigenvector rows.
    int l = 0, iter = 0, buf = 0;
    for (;;) {
        if (wave == 0) {
            int lo = 0, hi = 0, done = 0;
            for (;;) {
                if (l >= n) { done = 1; break; }
                // the first negligible sub-diagonal element at or behind l: lanes test m = l + lane, l + 64 + lane
                int m;
                {
                    const int m0 = l + lane, m1 = l + 64 + lane;
                    bool s0 = m0 >= n - 1, s1 = m1 >= n - 1;
                    if (!s0) { const double dd = fabs(W.d[m0]) + fabs(W.d[m0 + 1]); s0 = fabs(W.e[m0]) <= 2.220446049250313e-16 * dd; }
                    if (!s1) { const double dd = fabs(W.d[m1]) + fabs(W.d[m1 + 1]); s1 = fabs(W.e[m1]) <= 2.220446049250313e-16 * dd; }
                    const unsigned long long b0 = __ballot(s0), b1 = __ballot(s1);
                    m = b0 ? l + (int)__builtin_ctzll(b0) : l + 64 + (int)__builtin_ctzll(b1 | (1ull << 63));
                    if (m > n - 1) m = n - 1;
                    m = __builtin_amdgcn_readfirstlane(m);
                }
                // (l, m, i and every branch on them are wave-uniform; readfirstlane says so to the compiler, which otherwise masks lanes around each of them)
                if (m == l) { l = __builtin_amdgcn_readfirstlane(l + 1); iter = 0; continue; }
                if (++iter > 60) { if (lane == 0) W.fail = 1; l = __builtin_amdgcn_readfirstlane(l + 1); iter = 0; continue; }      // (never seen; the pair is left as it is)
#ifdef LMONO_MG_PROF
                p_sweeps++; p_rot += m - l;
#endif
                l = __builtin_amdgcn_readfirstlane(l);
                const double dl = W.d[l], el = W.e[l];
                double g = (W.d[l + 1] - dl) / (2.0 * el);
                double r = sqrt(__builtin_fma(g, g, 1.0));
                g = W.d[m] - dl + el / (g + (g >= 0.0 ? r : -r));
                double sn = 1.0, cn = 1.0, pp = 0.0;
                double *cs = W.cs[buf];
                int i = __builtin_amdgcn_readfirstlane(m - 1);
                bool zero_r = false;
                double e_i = W.e[i], d_i = W.d[i], d_i1 = W.d[i + 1];
#ifdef LMONO_MG_PROF
                const unsigned long long pc0 = __builtin_readcyclecounter();
#endif
                for (; i >= l; i = __builtin_amdgcn_readfirstlane(i - 1)) {
                    const int inx = i > l ? i - 1 : l;                                    // the next step's operands (e[i - 1], d[i - 1]: untouched by this sweep so far)
                    const double e_nx = W.e[inx], d_nx = W.d[inx];
                    const double f = sn * e_i, bb = cn * e_i;
                    const double x = __builtin_fma(f, f, g * g);
                    if (__builtin_amdgcn_readfirstlane((int)(x == 0.0))) {
                        if (lane == 0) { W.e[i + 1] = 0.0; W.d[i + 1] = d_i1 - pp; W.e[m] = 0.0; }
                        zero_r = true;
                        break;
                    }
                    const double ir = mg_rsqrt(x);
                    sn = f * ir; cn = g * ir;
                    g = d_i1 - pp;
                    r = __builtin_fma(d_i - g, sn, 2.0 * cn * bb);
                    pp = sn * r;
                    if (lane == 0) { W.e[i + 1] = x * ir; W.d[i + 1] = g + pp; cs[2 * (i - l)] = cn; cs[2 * (i - l) + 1] = sn; }
                    g = __builtin_fma(cn, r, -bb);
                    d_i1 = d_i; e_i = e_nx; d_i = d_nx;
                }
#ifdef LMONO_MG_PROF
                p_chain += __builtin_readcyclecounter() - pc0;
#endif
                if (zero_r) {
                    // (NR: r == 0 -> recover from underflow; the rotations i = m - 1 .. i + 1 computed so far stand.  Their list positions were written
                    // relative to l: move them down to start at 0 -- one wave, program order; a lane's read is ahead of every write that could reach it)
                    lo = i + 1; hi = m;
                    for (int q = lane; q < 2 * (hi - lo); q += 64) { const double v = cs[2 * (lo - l) + q]; cs[q] = v; }
                } else {
                    if (lane == 0) { W.d[l] = dl - pp; W.e[l] = g; W.e[m] = 0.0; }
                    lo = l; hi = m;
                }
                break;
            }
            if (lane == 0) { W.ctl[buf][0] = lo; W.ctl[buf][1] = hi; W.ctl[buf][2] = done; }
        } else {
            // the previous sweep's rotations on the eigenvector rows: thread per component k
            const int lo = W.ctl[buf ^ 1][0], hi = W.ctl[buf ^ 1][1];
            const int k = tid - 64;
            if (hi > lo && k < n) {
                const double *cs = W.cs[buf ^ 1];
                double z1 = A[hi * n + k];
                double zi = A[(hi - 1) * n + k], cn = cs[2 * (hi - 1 - lo)], sn = cs[2 * (hi - 1 - lo) + 1];
                for (int i = hi - 1; i >= lo; i--) {
                    const double z_nx = i > lo ? A[(i - 1) * n + k] : 0.0, c_nx = i > lo ? cs[2 * (i - 1 - lo)] : 0.0, s_nx = i > lo ? cs[2 * (i - 1 - lo) + 1] : 0.0;
                    A[(i + 1) * n + k] = __builtin_fma(sn, zi, cn * z1);
                    z1 = __builtin_fma(cn, zi, -sn * z1);
                    zi = z_nx; cn = c_nx; sn = s_nx;
                }
                A[lo * n + k] = z1;
            }
        }
#ifdef LMONO_MG_PROF
        const unsigned long long pb0 = __builtin_readcyclecounter();
#endif
        __syncthreads();
#ifdef LMONO_MG_PROF
        p_bar += __builtin_readcyclecounter() - pb0;
#endif
        const int done = W.ctl[buf][2];
        buf ^= 1;
        if (done) break;
    }
    // (the list written in the last productive sweep was applied during the sweep that found nothing left)
#ifdef LMONO_MG_PROF
    if (tid == 0) { W.prof[2] = __builtin_readcyclecounter() - pt0; W.n_sweeps = p_sweeps; W.n_rot = p_rot; W.prof[3] = p_chain; W.prof[4] = p_bar; }
#endif
    __syncthreads();
}

__global__ __launch_bounds__(kMgT) void k_marginalize(MargBatch Bt)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    MargLds &L = *reinterpret_cast<MargLds *>(smem_raw);
    const int w = blockIdx.x, tid = threadIdx.x;
    const int f0 = Bt.feat_off[w], F0 = Bt.feat_off[w + 1] - f0;
    const double *poses = Bt.poses + (size_t)w * 77, *ex = Bt.ex + (size_t)w * 7;
    const double *laser_info = Bt.info, *mono_info = Bt.info + 36;
    const double eps = 1e-8;
    for (int k = tid; k < F0 * kMargN; k += kMgT) L.Wd[k] = 0.0;
    for (int k = tid; k < 6 * F0; k += kMgT) L.B[k] = 0.0;
    for (int k = tid; k < F0; k += kMgT) { L.D[k] = 0.0; L.bd[k] = 0.0; }
    for (int k = tid; k < kMgAccWaves * kMgAcc; k += kMgT) L.acc[k] = 0.0;
    if (tid == 0) L.flag = 0;
    __syncthreads();
#ifdef LMONO_MG_PROF
    const unsigned long long mg_t0 = __builtin_readcyclecounter();
#endif
    // kept-block column of pose j (1..10) and of the extrinsic inside n
    auto col_pose = [](int j) { return 6 + 6 * (j - 1); };
    const int lane = tid & 63;
    if (tid >= kMgT - 64) {          // LASERFactor(pose0, pose1) on a wave that has no tracks: one lane evaluates it, the wave forms J^T J and J^T r
        double *stage = L.lz + 120;  // r[6] | J[84]
        if (tid == kMgT - 64) {
            double prm[14], r[6], J[84];
            for (int k = 0; k < 14; k++) prm[k] = poses[k];
            ba::laser_factor(prm, Bt.laser01 + (size_t)w * 24, laser_info, r, J);
            for (int k = 0; k < 6; k++) stage[k] = r[k];
            for (int k = 0; k < 84; k++) stage[6 + k] = J[k];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const double *r = stage, *J = stage + 6;
        for (int e = lane; e < 120; e += 64) {
            double v = 0;
            if (e < 12) { const int a = e % 6, o = e < 6 ? 0 : 42; for (int k = 0; k < 6; k++) v += J[o + k * 7 + a] * r[k]; }
            else { const int blk = (e - 12) / 36, a = ((e - 12) % 36) / 6, bb = (e - 12) % 6, oa = blk == 2 ? 42 : 0, ob = blk == 0 ? 0 : 42;
                   for (int k = 0; k < 6; k++) v += J[oa + k * 7 + a] * J[ob + k * 7 + bb]; }
            L.lz[e] = v;
        }
    }
    if (tid < kMgAccWaves * 64) {    // a track per lane; the o-th observations of a wave's tracks are evaluated together
        double *acc = L.acc + (tid >> 6) * kMgAcc;
        const int f = tid;
        const bool has = f < F0;
        const int o0 = has ? Bt.feat_obs_off[f0 + f] : 0, no = has ? Bt.feat_obs_off[f0 + f + 1] - o0 : 0;
        const int nmax = wave_max_i(no);
        double Df = 0, bdf = 0, Bf[6] = { 0, 0, 0, 0, 0, 0 }, Wx[6] = { 0, 0, 0, 0, 0, 0 };
        for (int it = 0; it < nmax; it++) {
            const bool act = it < no;
            int j = 0;
            double r[2] = { 0, 0 }, J[44];
#pragma unroll
            for (int k = 0; k < 44; k++) J[k] = 0.0;
            if (act) {
                const int o = o0 + it;
                j = Bt.obs_j[o];
                double prm[22];
                for (int k = 0; k < 7; k++) { prm[k] = ex[k]; prm[7 + k] = poses[k]; prm[14 + k] = poses[7 * j + k]; }
                prm[21] = Bt.inv_depth[f0 + f];
                ba::mono_factor(prm, Bt.obs_pts + (size_t)o * 4, mono_info, r, J);
                const double sq = r[0] * r[0] + r[1] * r[1];
                const double inv = 1.0 / (1.0 + sq);
                const double rho1 = inv > DBL_MIN ? inv : DBL_MIN, rho2 = -(inv * inv);
                marg_corrector(r, J, 7, rho1, rho2); marg_corrector(r, J + 14, 7, rho1, rho2); marg_corrector(r, J + 28, 7, rho1, rho2); marg_corrector(r, J + 42, 1, rho1, rho2);
                {   // residual scaling
                    const double sr = sqrt(rho1);
                    double scaling = sr;
                    if (!(sq == 0.0 || rho2 <= 0.0)) { const double Dd = 1.0 + 2.0 * sq * rho2 / rho1; scaling = sr / (1.0 - (1.0 - sqrt(Dd))); }
                    r[0] *= scaling; r[1] *= scaling;
                }
            }
            const double *Jx = J, *J0 = J + 14, *Jj = J + 28, *Jd = J + 42;
            // the track's own sums (one writer)
            if (act) {
                const int cj = col_pose(j);
#pragma unroll
                for (int a = 0; a < 6; a++) {
                    Bf[a] += J0[a] * Jd[0] + J0[7 + a] * Jd[1];
                    Wx[a] += Jd[0] * Jx[a] + Jd[1] * Jx[7 + a];
                    L.Wd[f * kMargN + cj + a] = Jd[0] * Jj[a] + Jd[1] * Jj[7 + a];     // frame j is observed once per track
                }
                Df += Jd[0] * Jd[0] + Jd[1] * Jd[1];
                bdf += Jd[0] * r[0] + Jd[1] * r[1];
            }
            // sums every track adds to: pose 0 and the extrinsic (an inactive lane's J and r are zero)
#pragma unroll
            for (int a = 0; a < 6; a++) {
                mg_acc(acc, a, J0[a] * r[0] + J0[7 + a] * r[1], lane);
                mg_acc(acc, 6 + a, Jx[a] * r[0] + Jx[7 + a] * r[1], lane);
#pragma unroll
                for (int bb = 0; bb < 6; bb++) {
                    if (bb >= a) {
                        mg_acc(acc, 12 + tri6(a, bb), J0[a] * J0[bb] + J0[7 + a] * J0[7 + bb], lane);
                        mg_acc(acc, 69 + tri6(a, bb), Jx[a] * Jx[bb] + Jx[7 + a] * Jx[7 + bb], lane);
                    }
                    mg_acc(acc, 33 + a * 6 + bb, J0[a] * Jx[bb] + J0[7 + a] * Jx[7 + bb], lane);
                }
            }
            // sums per observing frame: the frames the wave's tracks see in this round, one after the other (almost always one: a track anchored at
            // frame 0 is seen in frames 1, 2, .. without a gap)
            unsigned long long todo = __ballot(act);
            while (todo != 0ull) {
                const int jj = __shfl(j, (int)__ffsll((long long)todo) - 1);
                const bool mine = act && j == jj;
                todo &= ~__ballot(mine);
                double *aj = acc + kMgAccC + (jj - 1) * kMgAccJ;
                const double m = mine ? 1.0 : 0.0;
#pragma unroll
                for (int a = 0; a < 6; a++) {
                    mg_acc(aj, a, m * (Jj[a] * r[0] + Jj[7 + a] * r[1]), lane);
#pragma unroll
                    for (int bb = 0; bb < 6; bb++) {
                        mg_acc(aj, 6 + a * 6 + bb, m * (J0[a] * Jj[bb] + J0[7 + a] * Jj[7 + bb]), lane);
                        mg_acc(aj, 42 + a * 6 + bb, m * (Jx[a] * Jj[bb] + Jx[7 + a] * Jj[7 + bb]), lane);
                        if (bb >= a) mg_acc(aj, 78 + tri6(a, bb), m * (Jj[a] * Jj[bb] + Jj[7 + a] * Jj[7 + bb]), lane);
                    }
                }
            }
        }
        if (has) {
            L.D[f] = Df; L.bd[f] = bdf;
            for (int a = 0; a < 6; a++) { L.B[a * kMargMaxF0 + f] = Bf[a]; L.Wd[f * kMargN + a] = Wx[a]; }
        }
    }
    __syncthreads();
    // the waves' slots, added in wave order, become A, bp, W_p, b_r and H_rr (+ the LASERFactor's blocks)
    {
        auto S = [&](int idx) { double v = L.acc[idx]; for (int wv = 1; wv < kMgAccWaves; wv++) v += L.acc[wv * kMgAcc + idx]; return v; };
        for (int k = tid; k < kMargN * kMargN; k += kMgT) {
            const int rr = k / kMargN, cc = k % kMargN;
            double v = 0.0;
            if (rr < 6 && cc < 6) v = S(69 + tri6(rr, cc));
            else if (rr < 6) v = S(kMgAccC + ((cc - 6) / 6) * kMgAccJ + 42 + rr * 6 + (cc - 6) % 6);
            else if (cc < 6) v = S(kMgAccC + ((rr - 6) / 6) * kMgAccJ + 42 + cc * 6 + (rr - 6) % 6);
            else if ((rr - 6) / 6 == (cc - 6) / 6) {
                const int jb = (rr - 6) / 6, a = (rr - 6) % 6, bb = (cc - 6) % 6;
                v = S(kMgAccC + jb * kMgAccJ + 78 + tri6(a, bb));
                if (jb == 0) v += L.lz[84 + a * 6 + bb];
            }
            L.Hrr[k] = v;
        }
        for (int k = tid; k < 6 * kMargN; k += kMgT) {
            const int a = k / kMargN, cc = k % kMargN;
            double v = cc < 6 ? S(33 + a * 6 + cc) : S(kMgAccC + ((cc - 6) / 6) * kMgAccJ + 6 + a * 6 + (cc - 6) % 6);
            if (cc >= 6 && cc < 12) v += L.lz[48 + a * 6 + (cc - 6)];
            L.Wp[k] = v;
        }
        if (tid >= 64 && tid < 64 + kMargN) {
            const int cc = tid - 64;
            double v = cc < 6 ? S(6 + cc) : S(kMgAccC + ((cc - 6) / 6) * kMgAccJ + (cc - 6) % 6);
            if (cc >= 6 && cc < 12) v += L.lz[6 + (cc - 6)];
            L.br[cc] = v;
        }
        if (tid < 36) L.A[tid] = S(12 + tri6(tid / 6, tid % 6)) + L.lz[12 + tid];
        if (tid >= 36 && tid < 42) L.bp[tid - 36] = S(tid - 36) + L.lz[tid - 36];
    }
    __syncthreads();
#ifdef LMONO_MG_PROF
    const unsigned long long mg_t1 = __builtin_readcyclecounter();
#endif
    // D^+ (eps cut), S_A = A - B D^+ B^T, G = W_p - B D^+ W_d, u = bp - B D^+ bd
    for (int f = tid; f < F0; f += kMgT) { const double d = L.D[f]; if (!(d > eps)) L.flag = 1; L.D[f] = d > eps ? 1.0 / d : 0.0; }
    __syncthreads();
    if (tid < 36) {
        const int a = tid / 6, bb = tid % 6;
        double acc = 0.5 * (L.A[a * 6 + bb] + L.A[bb * 6 + a]);
        for (int f = 0; f < F0; f++) acc -= L.B[a * kMargMaxF0 + f] * L.D[f] * L.B[bb * kMargMaxF0 + f];
        L.SAi[tid] = acc;
    }
    if (tid >= 64 && tid < 70) {
        const int a = tid - 64;
        double acc = L.bp[a];
        for (int f = 0; f < F0; f++) acc -= L.B[a * kMargMaxF0 + f] * L.D[f] * L.bd[f];
        L.u[a] = acc;
    }
    for (int k = tid; k < 6 * kMargN; k += kMgT) {
        const int a = k / kMargN, cidx = k % kMargN;
        double acc = L.Wp[k];
        for (int f = 0; f < F0; f++) acc -= L.B[a * kMargMaxF0 + f] * L.D[f] * L.Wd[f * kMargN + cidx];
        L.Y[k] = acc;      // G, staged in Y
    }
    __syncthreads();
    for (int k = tid; k < 6 * kMargN; k += kMgT) L.Wp[k] = L.Y[k];
    if (tid == 0) { double tmp[36]; int deg = 0; pinv6(L.SAi, tmp, eps, &deg); for (int i = 0; i < 36; i++) L.SAi[i] = tmp[i]; if (deg) L.flag = 1; }
    __syncthreads();
    for (int k = tid; k < 6 * kMargN; k += kMgT) {
        const int a = k / kMargN, cidx = k % kMargN;
        double acc = 0;
        for (int q = 0; q < 6; q++) acc += L.SAi[a * 6 + q] * L.Wp[q * kMargN + cidx];
        L.Y[k] = acc;
    }
    __syncthreads();
    // H' = Hrr - G^T Y - Wd^T D^+ Wd ; b' = br - G^T SA^+ u - Wd^T D^+ bd   (H' in place: every element only reads itself of Hrr)
    for (int k = tid; k < kMargN * kMargN; k += kMgT) {
        const int a = k / kMargN, bb = k % kMargN;
        double acc = L.Hrr[k];
        for (int q = 0; q < 6; q++) acc -= L.Wp[q * kMargN + a] * L.Y[q * kMargN + bb];
        for (int f = 0; f < F0; f++) acc -= L.Wd[f * kMargN + a] * L.D[f] * L.Wd[f * kMargN + bb];
        L.Hrr[k] = acc;
    }
    for (int a = tid; a < kMargN; a += kMgT) {
        double acc = L.br[a];
        for (int q = 0; q < 6; q++) { double su = 0; for (int p = 0; p < 6; p++) su += L.SAi[q * 6 + p] * L.u[p]; acc -= L.Wp[q * kMargN + a] * su; }
        for (int f = 0; f < F0; f++) acc -= L.Wd[f * kMargN + a] * L.D[f] * L.bd[f];
        L.br[a] = acc;
    }
    __syncthreads();
    // symmetrise in place (one thread per unordered pair), then V (which shares W_d's memory) becomes the identity
    for (int k = tid; k < kMargN * kMargN; k += kMgT) {
        const int a = k / kMargN, bb = k % kMargN;
        if (a < bb) { const double v = 0.5 * (L.Hrr[k] + L.Hrr[bb * kMargN + a]); L.Hrr[k] = v; L.Hrr[bb * kMargN + a] = v; }
    }
    __syncthreads();
    // ---- eigen-decomposition of H' (66 x 66): Householder + QL; the eigenvectors come back as the rows of Hrr
#ifdef LMONO_MG_PROF
    const unsigned long long mg_t2 = __builtin_readcyclecounter();
#endif
    marg_eig_ql(L.Hrr, kMargN, L.eig, tid);
#ifdef LMONO_MG_PROF
    if (tid == 0) printf("MGPROF F0 %d obs %d factor %llu schur %llu jacobi %llu | tred2 %llu accumulate %llu ql %llu sweeps %d rotations %d chain %llu barrier %llu\n", F0, Bt.feat_obs_off[f0 + F0] - Bt.feat_obs_off[f0], mg_t1 - mg_t0, mg_t2 - mg_t1, (unsigned long long)__builtin_readcyclecounter() - mg_t2,
                         L.eig.prof[0], L.eig.prof[1], L.eig.prof[2], L.eig.n_sweeps, L.eig.n_rot, L.eig.prof[3], L.eig.prof[4]);
#endif
    // linearized_jacobians = sqrt(S) V^T, linearized_residuals = sqrt(S^-1) V^T b'
    for (int k = tid; k < kMargN * kMargN; k += kMgT) {
        const int e = k / kMargN, i = k % kMargN;
        const double wv = L.eig.d[e];
        Bt.lin_J[(size_t)w * kMargN * kMargN + k] = (wv > eps ? sqrt(wv) : 0.0) * L.Hrr[k];
    }
    for (int e = tid; e < kMargN; e += kMgT) {
        const double wv = L.eig.d[e];
        double vb = 0;
        for (int i = 0; i < kMargN; i++) vb += L.Hrr[e * kMargN + i] * L.br[i];
        Bt.lin_r[(size_t)w * kMargN + e] = (wv > eps ? sqrt(1.0 / wv) : 0.0) * vb;
    }
    if (tid == 0) Bt.status[w] = L.flag | (L.eig.fail ? 2 : 0);
}

// Marginalization::Evaluate: residual = r0 + J dx for the kept blocks x (11 x 7: ex, pose1..pose10); one thread per row
__global__ __launch_bounds__(128) void k_marg_evaluate(int n_windows, const double *lin_J, const double *lin_r, const double *x0, const double *x, double *residual)
{
    const int w = blockIdx.x, e = threadIdx.x;
    if (w >= n_windows) return;
    __shared__ double dx[kMargN];
    if (e < 11) {
        const double *a = x + (size_t)w * 77 + 7 * e, *a0 = x0 + (size_t)w * 77 + 7 * e;
        for (int k = 0; k < 3; k++) dx[6 * e + k] = a[k] - a0[k];
        const double n2 = a0[3] * a0[3] + a0[4] * a0[4] + a0[5] * a0[5] + a0[6] * a0[6];
        const double ix = -a0[3] / n2, iy = -a0[4] / n2, iz = -a0[5] / n2, iw = a0[6] / n2;
        const double qx = a[3], qy = a[4], qz = a[5], qw = a[6];
        const double rw = iw * qw - ix * qx - iy * qy - iz * qz;
        double rx = iw * qx + ix * qw + iy * qz - iz * qy, ry = iw * qy + iy * qw + iz * qx - ix * qz, rz = iw * qz + iz * qw + ix * qy - iy * qx;
        if (!(rw >= 0)) { rx = -rx; ry = -ry; rz = -rz; }
        dx[6 * e + 3] = 2.0 * rx; dx[6 * e + 4] = 2.0 * ry; dx[6 * e + 5] = 2.0 * rz;
    }
    __syncthreads();
    if (e < kMargN) {
        double v = lin_r[(size_t)w * kMargN + e];
        for (int k = 0; k < kMargN; k++) v += lin_J[(size_t)w * kMargN * kMargN + e * kMargN + k] * dx[k];
        residual[(size_t)w * kMargN + e] = v;
    }
}


// ---- MARGIN_SECOND_NEW (Estimator.cc:1406-1470): the previous prior, as the only factor (Marginalization::Evaluate at the current
// parameter values, MarginalizationFactor.cc:309-373), loses the block that aliases para_pose[WINDOW_SIZE - 1].
// One workgroup per window: dx, r = r0 + J0 dx, H = J0^T J0 and b = J0^T r with the dropped block's six columns first, eigen
// pseudo-inverse of the 6x6 H_mm (eps cut), Schur complement, eigen-decomposition (marg_eig_ql) of the (n0 - 6)-square H',
// linearized_jacobians = sqrt(S) V^T, linearized_residuals = sqrt(S^-1) V^T b'.  Kept blocks stay in their old order.
struct Marg2Batch {
    int n_windows, nb, drop;     // blocks of the previous prior (<= 11), index of the dropped one
    const double *lin_J;         // [W][n0*n0], n0 = 6 nb
    const double *lin_r;         // [W][n0]
    const double *x0;            // [W][nb][7] linearisation point of the previous prior
    const double *x;             // [W][nb][7] current values
    double *out_J;               // [W][n*n], n = n0 - 6
    double *out_r;               // [W][n]
    int *status;                 // [W] bit 0: H_mm degenerate (eps cut applied)
};
struct Marg2Lds {
    double A[kMargN * kMargN];   // J0, later the eigenvectors V
    double H[kMargN * kMargN];
    double T[kMargN * 6];
    double r[kMargN], b[kMargN], dx[kMargN], br[kMargN];
    double Hmm[36], Hinv[36];
    MargEigWork eig;
    int perm[kMargN], flag;
};

__global__ __launch_bounds__(kMgT) void k_marg_second_new(Marg2Batch Bt)
{
    extern __shared__ __align__(16) unsigned char smem_raw2[];
    Marg2Lds &L = *reinterpret_cast<Marg2Lds *>(smem_raw2);
    const int w = blockIdx.x, tid = threadIdx.x;
    const int nb = Bt.nb, n0 = 6 * nb, n = n0 - 6, drop = Bt.drop;
    const double eps = 1e-8;
    const double *J0 = Bt.lin_J + (size_t)w * n0 * n0;
    for (int k = tid; k < n0 * n0; k += kMgT) L.A[k] = J0[k];
    if (tid < nb) {
        const double *a = Bt.x + ((size_t)w * nb + tid) * 7, *a0 = Bt.x0 + ((size_t)w * nb + tid) * 7;
        for (int k = 0; k < 3; k++) L.dx[6 * tid + k] = a[k] - a0[k];
        const double n2 = a0[3] * a0[3] + a0[4] * a0[4] + a0[5] * a0[5] + a0[6] * a0[6];
        const double ix = -a0[3] / n2, iy = -a0[4] / n2, iz = -a0[5] / n2, iw = a0[6] / n2;
        const double qx = a[3], qy = a[4], qz = a[5], qw = a[6];
        const double rw = iw * qw - ix * qx - iy * qy - iz * qz;
        double rx = iw * qx + ix * qw + iy * qz - iz * qy, ry = iw * qy + iy * qw + iz * qx - ix * qz, rz = iw * qz + iz * qw + ix * qy - iy * qx;
        if (!(rw >= 0)) { rx = -rx; ry = -ry; rz = -rz; }
        L.dx[6 * tid + 3] = 2.0 * rx; L.dx[6 * tid + 4] = 2.0 * ry; L.dx[6 * tid + 5] = 2.0 * rz;
    }
    if (tid < n0) {
        // dropped block first, then the kept blocks in order
        int src;
        if (tid < 6) src = 6 * drop + tid;
        else { const int kb = (tid - 6) / 6, c = (tid - 6) % 6; src = 6 * (kb < drop ? kb : kb + 1) + c; }
        L.perm[tid] = src;
    }
    if (tid == 0) L.flag = 0;
    __syncthreads();
    if (tid < n0) { double v = Bt.lin_r[(size_t)w * n0 + tid]; for (int k = 0; k < n0; k++) v += L.A[tid * n0 + k] * L.dx[k]; L.r[tid] = v; }
    __syncthreads();
    for (int k = tid; k < n0 * n0; k += kMgT) {
        const int i = k / n0, j = k % n0, pi = L.perm[i], pj = L.perm[j];
        double v = 0;
        for (int q = 0; q < n0; q++) v += L.A[q * n0 + pi] * L.A[q * n0 + pj];
        L.H[k] = v;
    }
    if (tid < n0) { const int pi = L.perm[tid]; double v = 0; for (int q = 0; q < n0; q++) v += L.A[q * n0 + pi] * L.r[q]; L.b[tid] = v; }
    __syncthreads();
    if (tid < 36) { const int i = tid / 6, j = tid % 6; L.Hmm[tid] = 0.5 * (L.H[i * n0 + j] + L.H[j * n0 + i]); }
    __syncthreads();
    if (tid == 0) { int deg = 0; pinv6(L.Hmm, L.Hinv, eps, &deg); if (deg) L.flag = 1; }
    __syncthreads();
    for (int k = tid; k < n * 6; k += kMgT) {
        const int i = k / 6, j = k % 6;
        double v = 0;
        for (int q = 0; q < 6; q++) v += L.H[(6 + i) * n0 + q] * L.Hinv[q * 6 + j];
        L.T[k] = v;
    }
    __syncthreads();
    // H' (n x n, leading dimension n) into A; b'
    for (int k = tid; k < n * n; k += kMgT) {
        const int i = k / n, j = k % n;
        double v = 0;
        for (int q = 0; q < 6; q++) v += L.T[i * 6 + q] * L.H[q * n0 + 6 + j];
        L.A[k] = L.H[(6 + i) * n0 + 6 + j] - v;
    }
    if (tid < n) { double v = 0; for (int q = 0; q < 6; q++) v += L.T[tid * 6 + q] * L.b[q]; L.br[tid] = L.b[6 + tid] - v; }
    __syncthreads();
    // H' symmetrised into H (ld n): the eigen-solver reads both triangles
    for (int k = tid; k < n * n; k += kMgT) { const int i = k / n, j = k % n; L.H[k] = 0.5 * (L.A[k] + L.A[j * n + i]); }
    __syncthreads();
    // eigen-decomposition (Householder + QL): the eigenvectors come back as the rows of H
    marg_eig_ql(L.H, n, L.eig, tid);
    for (int k = tid; k < n * n; k += kMgT) {
        const int e = k / n;
        const double wv = L.eig.d[e];
        Bt.out_J[(size_t)w * n * n + k] = (wv > eps ? sqrt(wv) : 0.0) * L.H[k];
    }
    for (int e = tid; e < n; e += kMgT) {
        const double wv = L.eig.d[e];
        double vb = 0;
        for (int i = 0; i < n; i++) vb += L.H[e * n + i] * L.br[i];
        Bt.out_r[(size_t)w * n + e] = (wv > eps ? sqrt(1.0 / wv) : 0.0) * vb;
    }
    if (tid == 0) Bt.status[w] = L.flag | (L.eig.fail ? 2 : 0);
}

} // namespace lmono
