// posegraph.hip -- loop-closure pose graph on gfx950.
//
// NEW FEATURE: the reference detects loops and publishes loop_info (relative_t, relative_q, relative_yaw;
// mono_lidar_mapping/src/loop_detection/KeyFrame.cc:570-633) but has no graph optimisation; it only carries the unused 4-DoF
// helpers NormalizeAngle / AngleLocalParameterization / YawPitchRollToRotationMatrix (include/loop_detection/Loop_Detector.h:
// 99-168).  This is the 4-DoF (yaw + translation, degrees) keyframe graph those helpers belong to: odometry edges to the four
// previous keyframes, one Huber(0.1) edge per loop with the yaw residual down-weighted by 10, keyframe 0 fixed, solved by a
// Ceres-style Levenberg-Marquardt.
//
//   k_pg_linearise  one thread per keyframe gathers its incident edges (residual, 4 x 4 Jacobians, robust weight) into its own
//                   block row of the banded normal equations, its gradient entries and its cost share -- no atomics, bit-identical
//                   from run to run.  With `world` ranks each rank only adds the edges it owns (edges follow their newer keyframe,
//                   keyframes are split into contiguous ranges like the scans): the buffer [H | g | cost] is then summed over
//                   ranks with ONE all-reduce (RCCL over xGMI) and every rank takes the same step.
//   k_pg_step       one workgroup: accept / reject the last candidate, trust-region bookkeeping, Jacobi scaling + damping,
//                   in-place right-looking block-banded Cholesky (keyframes in reverse Cuthill-McKee order, half bandwidth w
//                   blocks of 4 x 4), both substitutions, the next candidate.  The band lives in HBM / L2 (w reaches ~80 for
//                   multi-lap graphs: the (4w)^2 window does not fit LDS), the current block column is staged in LDS.
#pragma once
#include "common.hpp"

namespace lmono {

constexpr int kPgT = 1024;
constexpr int kPgMaxW = 255;          // half bandwidth (blocks) the LDS staging of one block column is sized for
constexpr double kPgPi = 3.14159265358979323846;

struct PgState {
    double radius, decrease_factor, x_cost, cand_cost, model_change, x_norm, cost0, gmax;
    int iter, invalid_steps, reuse_diag, done, started, have_cand, accepted, rejected;
};

struct PgView {
    int n, w, n_edges;
    const int *ea, *eb, *eloop;
    const double *emeas;               // [n_edges][4] relative_t, relative_yaw
    const int *inc_start, *inc_edge;   // edges incident to every keyframe
    const int *pos, *node_at;          // elimination position of a keyframe and its inverse
    const double *pitch, *roll;        // fixed, degrees
    double *x, *cand;                  // [n][4] yaw, t (keyframe order)
    double *lin;                       // reduce buffer: H [n][w+1][16] | g [4n] | cost [n]   (position order)
    double *cur;                       // accepted linearisation, same layout
    double *Aw;                        // working band [n][w+1][16]
    double *scale, *diag, *gs, *sol;   // [4n] position order
    PgState *st;
};

__device__ __forceinline__ double pg_normalize_angle(double a)      // Loop_Detector.h:99-107
{
    if (a > 180.0) return a - 360.0;
    if (a < -180.0) return a + 360.0;
    return a;
}

// residual (4) and Jacobians (4 x 4, columns yaw tx ty tz) of one edge; returns 1/2 rho(|r|^2).  YawPitchRollToRotationMatrix:
// Loop_Detector.h:129-147.
__device__ __forceinline__ double pg_edge_eval(const PgView &v, int e, const double *x, double r[4], double Ja[16], double Jb[16])
{
    const int a = v.ea[e], b = v.eb[e];
    const bool loop = v.eloop[e] != 0;
    const double ya = x[4 * a], y = ya / 180.0 * kPgPi, p = v.pitch[a] / 180.0 * kPgPi, rr = v.roll[a] / 180.0 * kPgPi;
    const double cy = cos(y), sy = sin(y), cp = cos(p), sp = sin(p), cr = cos(rr), sr = sin(rr);
    const double R[9] = { cy * cp, -sy * cr + cy * sp * sr, sy * sr + cy * sp * cr, sy * cp, cy * cr + sy * sp * sr, -cy * sr + sy * sp * cr, -sp, cp * sr, cp * cr };
    const double dR[6] = { -sy * cp, -cy * cr - sy * sp * sr, cy * sr - sy * sp * cr, cy * cp, -sy * cr + cy * sp * sr, sy * sr + cy * sp * cr };
    const double d0 = x[4 * b + 1] - x[4 * a + 1], d1 = x[4 * b + 2] - x[4 * a + 2], d2 = x[4 * b + 3] - x[4 * a + 3];
    const double wy = loop ? 0.1 : 1.0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        r[k] = R[k] * d0 + R[3 + k] * d1 + R[6 + k] * d2 - v.emeas[4 * e + k];
        Ja[4 * k] = (kPgPi / 180.0) * (dR[k] * d0 + dR[3 + k] * d1);
        Jb[4 * k] = 0.0;
#pragma unroll
        for (int c = 0; c < 3; c++) { Ja[4 * k + 1 + c] = -R[3 * c + k]; Jb[4 * k + 1 + c] = R[3 * c + k]; }
    }
    r[3] = pg_normalize_angle(x[4 * b] - ya - v.emeas[4 * e + 3]) * wy;
    Ja[12] = -wy; Jb[12] = wy;
#pragma unroll
    for (int c = 1; c < 4; c++) { Ja[12 + c] = 0.0; Jb[12 + c] = 0.0; }
    const double s = r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3];
    if (!loop || s <= 0.01) return 0.5 * s;
    const double sq = sqrt(s), wr = sqrt(0.1 / sq);          // ceres::HuberLoss(0.1), corrector with rho'' <= 0
#pragma unroll
    for (int k = 0; k < 4; k++) r[k] *= wr;
#pragma unroll
    for (int k = 0; k < 16; k++) { Ja[k] *= wr; Jb[k] *= wr; }
    return 0.5 * (0.2 * sq - 0.01);
}

// lin must be zero on entry.  lo..hi: the keyframe range of this rank (an edge belongs to the rank that owns its newer keyframe b).
__global__ __launch_bounds__(256) void k_pg_linearise(PgView v, int lo, int hi)
{
    const int node = blockIdx.x * 256 + threadIdx.x;
    if (node >= v.n || v.st->done) return;
    const double *x = v.st->started ? v.cand : v.x;
    const int bs = (v.w + 1) * 16, p = v.pos[node];
    double *Hrow = v.lin + (size_t)p * bs, *g = v.lin + (size_t)v.n * bs + 4 * p, *cost = v.lin + (size_t)v.n * bs + 4 * (size_t)v.n + p;
    double Hd[16], gv[4] = { 0, 0, 0, 0 }, cs = 0.0;
#pragma unroll
    for (int k = 0; k < 16; k++) Hd[k] = 0.0;
    bool any = false;
    for (int k = v.inc_start[node]; k < v.inc_start[node + 1]; k++) {
        const int e = v.inc_edge[k], a = v.ea[e], b = v.eb[e];
        if (b < lo || b >= hi) continue;
        double r[4], J[2][16];
        const double c = pg_edge_eval(v, e, x, r, J[0], J[1]);
        const int s = node == a ? 0 : 1, other = s ? a : b;
        if (s == 1) cs += c;
        if (node == 0) continue;                         // keyframe 0 is constant: identity row below
        any = true;
#pragma unroll
        for (int c1 = 0; c1 < 4; c1++) {
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < 4; q++) acc += J[s][4 * q + c1] * r[q];
            gv[c1] += acc;
#pragma unroll
            for (int c2 = 0; c2 < 4; c2++) {
                double h = 0.0;
#pragma unroll
                for (int q = 0; q < 4; q++) h += J[s][4 * q + c1] * J[s][4 * q + c2];
                Hd[4 * c1 + c2] += h;
            }
        }
        const int po = v.pos[other];
        if (other != 0 && po < p) {
            double *blk = Hrow + (size_t)(p - po) * 16;
#pragma unroll
            for (int c1 = 0; c1 < 4; c1++)
#pragma unroll
                for (int c2 = 0; c2 < 4; c2++) {
                    double h = 0.0;
#pragma unroll
                    for (int q = 0; q < 4; q++) h += J[s][4 * q + c1] * J[1 - s][4 * q + c2];
                    blk[4 * c1 + c2] += h;
                }
        }
    }
    *cost = cs;
    if (node == 0) {
        if (lo == 0) { Hrow[0] = 1.0; Hrow[5] = 1.0; Hrow[10] = 1.0; Hrow[15] = 1.0; }     // once over all ranks
        return;
    }
    if (!any) return;
#pragma unroll
    for (int k = 0; k < 16; k++) Hrow[k] = Hd[k];
#pragma unroll
    for (int k = 0; k < 4; k++) g[k] = gv[k];
}

// ---- single-workgroup helpers -----------------------------------------------------------------------------------------
__device__ __forceinline__ double pg_block_sum(double v, double *s_red)
{
    v = wave_sum_d(v);
    __syncthreads();
    if (lane_id() == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < kPgT / kWave; k++) t += s_red[k];
    return t;
}
__device__ __forceinline__ double pg_block_max(double v, double *s_red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    __syncthreads();
    if (lane_id() == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < kPgT / kWave; k++) t = fmax(t, s_red[k]);
    return t;
}

// ---- panelled band solver ------------------------------------------------------------------------------------------------
// In-place right-looking Cholesky of the block band Aw (lower triangle, block (p, p - d) at Aw[(p (w+1) + d) 16], row-major) fused
// with the forward substitution of sol, then the backward substitution.  Returns false when a pivot is not positive.
//
// Round 1 factored one block column at a time with the (w+1)^2-block trailing window read and written through L2 for EVERY
// keyframe (54.8 ms at 4541 keyframes / w = 67).  Here P block columns form a panel that lives in LDS while it is factored
// (the same three barriers per keyframe, but between them only LDS is touched), and the trailing window is then updated ONCE
// per panel as a rank-4P product on the matrix cores (v_mfma_f64_16x16x4_f64: one instruction per 16 x 16 output tile and panel
// column, operands read from the LDS panel, where blocks outside the band are stored as zeros so that no operand needs a mask).
// The window traffic falls by P; the backward substitution is panelled the same way.  Measured: 37.5 ms per round -- the round is
// now bound by the dependent chains inside a panel (a 32 x 32 dense factorisation is 32 sequential pivots whatever executes it;
// a one-wave register version with lane-to-lane broadcasts and independent row solves was tried: 50 ms).
typedef double pg_d4 __attribute__((ext_vector_type(4)));
__host__ __device__ inline int pg_panel_width(int w) { return w <= 120 ? 8 : 4; }     // the panel must fit LDS
__host__ __device__ inline size_t pg_panel_doubles(int w) { const int P = pg_panel_width(w); return (size_t)(w + P + 4) * P * 16 + (size_t)(w + P + 4) * 4; }

__device__ bool pg_band_solve(const PgView &v, double *s_pan, double *s_d, int *s_flag)
{
    const int n = v.n, w = v.w, bs = (w + 1) * 16, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = pg_panel_width(w), rows_cap = w + P + 4;
    double *A = v.Aw, *y = v.sol;
    double *s_y = s_pan + (size_t)rows_cap * P * 16;          // right-hand side rows of the panel's window
    if (tid == 0) *s_flag = 1;
    __syncthreads();
    for (int p0 = 0; p0 < n; p0 += P) {
        const int Pw = min(P, n - p0);
        const int R = min(Pw - 1 + w, n - 1 - p0) + 1;        // block rows p0 .. p0 + R - 1 are touched by this panel
        // ---- load: block (p0 + row, p0 + k) -> s_pan[(row P + k) 16]; zeros outside the band and below row R
        for (int t = tid; t < rows_cap * P * 16; t += kPgT) {
            const int e = t & 15, k = (t >> 4) % P, row = (t >> 4) / P;
            double a = 0.0;
            if (row < R && k < Pw && row >= k && row - k <= w) a = A[(size_t)(p0 + row) * bs + (size_t)(row - k) * 16 + e];
            s_pan[t] = a;
        }
        for (int t = tid; t < rows_cap * 4; t += kPgT) s_y[t] = t < R * 4 ? y[4 * p0 + t] : 0.0;
        __syncthreads();
        // ---- factor the panel inside LDS, one block column after the other
        for (int k = 0; k < Pw; k++) {
            const int m = min(w, n - 1 - (p0 + k));            // block rows below the pivot
            if (tid == 0) {
                double *D = s_pan + (size_t)(k * P + k) * 16;
                double L[16];
                bool ok = true;
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j <= i; j++) {
                        double s = D[4 * i + j];
#pragma unroll
                        for (int q = 0; q < j; q++) s -= L[4 * i + q] * L[4 * j + q];
                        if (i == j) { ok = ok && s > 0.0; L[4 * i + i] = sqrt(s); }
                        else L[4 * i + j] = s / L[4 * j + j];
                    }
                if (!ok) *s_flag = 0;
                double yy[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    double s = s_y[4 * k + i];
#pragma unroll
                    for (int q = 0; q < i; q++) s -= L[4 * i + q] * yy[q];
                    yy[i] = s / L[4 * i + i];
                    s_y[4 * k + i] = yy[i];
                    s_d[16 + i] = yy[i];
                }
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 4; j++) { const double l = j <= i ? L[4 * i + j] : 0.0; D[4 * i + j] = l; s_d[4 * i + j] = l; }
            }
            __syncthreads();
            if (!*s_flag) return false;
            // block column k below the pivot: L_ik = A_ik L_kk^-T (one scalar row per thread), b_i -= L_ik y_k
            for (int t = tid; t < 4 * m; t += kPgT) {
                const int io = t >> 2, r = t & 3, row = k + 1 + io;
                double *blk = s_pan + (size_t)(row * P + k) * 16 + 4 * r;
                double l[4];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    double s = blk[c];
#pragma unroll
                    for (int q = 0; q < c; q++) s -= l[q] * s_d[4 * c + q];
                    l[c] = s / s_d[4 * c + c];
                }
                double dot = 0.0;
#pragma unroll
                for (int c = 0; c < 4; c++) { blk[c] = l[c]; dot += l[c] * s_d[16 + c]; }
                s_y[4 * row + r] -= dot;
            }
            __syncthreads();
            // the later columns of the panel: A_(row, k2) -= L_(row, k) L_(k2, k)^T for k < k2 < Pw, k2 <= row <= k + m
            const int nk2 = Pw - 1 - k;
            for (int t = tid; t < nk2 * m * 4; t += kPgT) {
                const int r = t & 3, io = (t >> 2) % m, k2 = k + 1 + (t >> 2) / m, row = k + 1 + io;
                if (row < k2) continue;
                const double *li = s_pan + (size_t)(row * P + k) * 16 + 4 * r, *lj = s_pan + (size_t)(k2 * P + k) * 16;
                double *blk = s_pan + (size_t)(row * P + k2) * 16 + 4 * r;
#pragma unroll
                for (int c = 0; c < 4; c++) blk[c] -= li[0] * lj[4 * c] + li[1] * lj[4 * c + 1] + li[2] * lj[4 * c + 2] + li[3] * lj[4 * c + 3];
            }
            __syncthreads();
        }
        // ---- the factored panel and the window's right-hand side go back
        for (int t = tid; t < R * P * 16; t += kPgT) {
            const int e = t & 15, k = (t >> 4) % P, row = (t >> 4) / P;
            if (k < Pw && row >= k && row - k <= w) A[(size_t)(p0 + row) * bs + (size_t)(row - k) * 16 + e] = s_pan[t];
        }
        for (int t = tid; t < R * 4; t += kPgT) y[4 * p0 + t] = s_y[t];
        // ---- trailing window: A_(row, col) -= sum_k L_(row, k) L_(col, k)^T for Pw <= col <= row < R, 16 x 16 tiles on the matrix cores
        const int Tn = (R - Pw + 3) >> 2;
        for (int tile = wave; tile < Tn * (Tn + 1) / 2; tile += kPgT / kWave) {
            int a = 0;
            while ((a + 1) * (a + 2) / 2 <= tile) a++;           // tile (a, b), b <= a, in row-major order of the lower triangle
            const int b = tile - a * (a + 1) / 2;
            const int cq = lane >> 4, cc = lane & 15;            // this lane: tile rows 4 v + cq (v = register), tile column cc
            const int col = Pw + 4 * b + (cc >> 2), c = cc & 3;
            pg_d4 acc;
            bool valid[4];
#pragma unroll
            for (int vv = 0; vv < 4; vv++) {
                const int row = Pw + 4 * a + vv;
                valid[vv] = row < R && col <= row;
                acc[vv] = valid[vv] ? A[(size_t)(p0 + row) * bs + (size_t)(row - col) * 16 + 4 * cq + c] : 0.0;
            }
            // operands: A[i][kk] = -L[(row_i, r_i), (k, kk)], B[kk][j] = L[(col_j, c_j), (k, kk)]; lane = (i or j) + 16 kk
            const int orow = Pw + 4 * a + (cc >> 2), orr = cc & 3, kk = cq;
            const double *pa = s_pan + (size_t)orow * P * 16 + 4 * orr + kk, *pb = s_pan + (size_t)col * P * 16 + 4 * c + kk;
            for (int k = 0; k < Pw; k++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[k * 16], pb[k * 16], acc, 0, 0, 0);
#pragma unroll
            for (int vv = 0; vv < 4; vv++) {
                const int row = Pw + 4 * a + vv;
                if (valid[vv]) A[(size_t)(p0 + row) * bs + (size_t)(row - col) * 16 + 4 * cq + c] = acc[vv];
            }
        }
        __syncthreads();
    }
    // ---- backward substitution, panel by panel from the bottom: x_i = L_ii^-T (y_i - sum_(j > i) L_ji^T x_j)
    for (int i0 = ((n - 1) / P) * P; i0 >= 0; i0 -= P) {
        const int Pw = min(P, n - i0);
        // block rows i0 .. i0 + Pw - 1 of L: s_pan[(ii (w+1) + d) 16] = block (i0 + ii, i0 + ii - d)
        for (int t = tid; t < Pw * bs; t += kPgT) {
            const int ii = t / bs, d = (t - ii * bs) >> 4;
            s_pan[t] = i0 + ii - d >= 0 ? A[(size_t)(i0 + ii) * bs + (t - ii * bs)] : 0.0;
        }
        for (int t = tid; t < Pw * 4; t += kPgT) s_y[t] = y[4 * i0 + t];
        __syncthreads();
        if (tid == 0) {
            for (int ii = Pw - 1; ii >= 0; ii--) {
                const double *L = s_pan + (size_t)ii * bs;
                double xx[4];
#pragma unroll
                for (int c = 3; c >= 0; c--) {
                    double s = s_y[4 * ii + c];
#pragma unroll
                    for (int q = c + 1; q < 4; q++) s -= L[4 * q + c] * xx[q];
                    xx[c] = s / L[4 * c + c];
                    s_y[4 * ii + c] = xx[c];
                }
                for (int d = 1; d <= ii && d <= w; d++) {       // the rows above it inside the panel
                    const double *blk = L + d * 16;
#pragma unroll
                    for (int c = 0; c < 4; c++) s_y[4 * (ii - d) + c] -= blk[c] * xx[0] + blk[4 + c] * xx[1] + blk[8 + c] * xx[2] + blk[12 + c] * xx[3];
                }
            }
        }
        __syncthreads();
        for (int t = tid; t < Pw * 4; t += kPgT) y[4 * i0 + t] = s_y[t];
        // rows above the panel: y_J -= sum_ii L_(i0 + ii, J)^T x_ii for J = i0 - dd, dd = 1 .. w
        for (int t = tid; t < 4 * min(w, i0); t += kPgT) {
            const int dd = (t >> 2) + 1, c = t & 3;
            double s = 0.0;
            for (int ii = 0; ii < Pw && ii + dd <= w; ii++) {
                const double *blk = s_pan + (size_t)ii * bs + (size_t)(ii + dd) * 16;
                s += blk[c] * s_y[4 * ii] + blk[4 + c] * s_y[4 * ii + 1] + blk[8 + c] * s_y[4 * ii + 2] + blk[12 + c] * s_y[4 * ii + 3];
            }
            y[4 * (i0 - dd) + c] -= s;
        }
        __syncthreads();
    }
    return true;
}

// One trust-region round (the loop body of the oracle's lo_pose_graph_optimize / lo_lm_solve): consume the linearisation in
// v.lin (initial point or last candidate), decide, solve for the next candidate.
__global__ __launch_bounds__(kPgT) void k_pg_step(PgView v, int max_iter)
{
    extern __shared__ __align__(16) double s_pan[];          // pg_panel_doubles(w): the panel of the band solver
    __shared__ double s_d[20];
    __shared__ double s_red[kPgT / kWave];
    __shared__ int s_flag;
    __shared__ PgState s;
    const int tid = threadIdx.x, n = v.n, n4 = 4 * n, bs = (v.w + 1) * 16;
    const size_t hsz = (size_t)n * bs;
    const double function_tol = 1e-6, gradient_tol = 1e-10, parameter_tol = 1e-8;
    const double min_rel_decrease = 1e-3, min_diag = 1e-6, max_diag = 1e32, max_radius = 1e16, min_radius = 1e-32;
    if (tid == 0) s = *v.st;
    __syncthreads();
    if (s.done) return;
    // cost of the linearisation point, summed in position order
    double part = 0.0;
    for (int i = tid; i < n; i += kPgT) part += v.lin[hsz + (size_t)n4 + i];
    const double lin_cost = pg_block_sum(part, s_red);
    bool take = false;           // v.lin becomes the accepted linearisation
    if (!s.started) {
        take = true;
        double xn = 0.0;
        for (int i = tid; i < n4; i += kPgT) xn += v.x[i] * v.x[i];
        xn = pg_block_sum(xn, s_red);
        for (int i = tid; i < n4; i += kPgT) v.scale[i] = 1.0 / (1.0 + sqrt(v.lin[(size_t)(i >> 2) * bs + 5 * (i & 3)]));    // Jacobi scaling, fixed at the start
        if (tid == 0) {
            s.started = 1; s.radius = 1e4; s.decrease_factor = 2.0; s.x_cost = lin_cost; s.cost0 = lin_cost; s.x_norm = sqrt(xn);
            s.iter = 0; s.invalid_steps = 0; s.reuse_diag = 0; s.have_cand = 0; s.accepted = 0; s.rejected = 0;
        }
    } else if (s.have_cand) {
        const double rel = (s.x_cost - lin_cost) / s.model_change;
        take = rel > min_rel_decrease;
        if (take) {
            double xn = 0.0;
            for (int i = tid; i < n4; i += kPgT) { const double c = v.cand[i]; v.x[i] = c; xn += c * c; }
            xn = pg_block_sum(xn, s_red);
            if (tid == 0) {
                const bool converged = fabs(s.x_cost - lin_cost) <= function_tol * s.x_cost;
                s.x_norm = sqrt(xn); s.x_cost = lin_cost;
                const double t = 2.0 * rel - 1.0;
                double den = 1.0 - t * t * t;
                den = den < 1.0 / 3.0 ? 1.0 / 3.0 : den;
                s.radius = fmin(s.radius / den, max_radius);
                s.decrease_factor = 2.0; s.reuse_diag = 0; s.accepted++;
                if (converged) s.done = 1;
            }
        } else if (tid == 0) {
            s.radius = s.radius / s.decrease_factor; s.decrease_factor *= 2.0; s.reuse_diag = 1; s.rejected++;
        }
        if (tid == 0) { s.have_cand = 0; if (s.radius <= min_radius) s.done = 1; }
    }
    __syncthreads();
    if (take) {
        for (size_t i = tid; i < hsz + (size_t)n4; i += kPgT) v.cur[i] = v.lin[i];
        double gm = 0.0;
        for (int i = tid; i < n4; i += kPgT) gm = fmax(gm, fabs(v.lin[hsz + i]));
        gm = pg_block_max(gm, s_red);
        if (tid == 0) { s.gmax = gm; if (gm <= gradient_tol) s.done = 1; }
    }
    __syncthreads();
    // next step: scaled, damped band -> Cholesky -> candidate
    while (!s.done) {
        if (s.iter >= max_iter) { if (tid == 0) s.done = 1; __syncthreads(); break; }
        __syncthreads();
        if (tid == 0) s.iter++;
        const double radius = s.radius;
        const bool reuse = s.reuse_diag != 0;
        // one wave per block row at a time (no 64-bit division per element)
        for (int p = tid >> 6; p < n; p += kPgT / kWave) {
            const double *src = v.cur + (size_t)p * bs;
            double *dst = v.Aw + (size_t)p * bs;
            for (int rem = tid & 63; rem < bs; rem += kWave) {
                const int d = rem >> 4, r = (rem >> 2) & 3, c = rem & 3;
                const int col = 4 * (p - d) + c;
                double a = 0.0;
                if (p - d >= 0 && !(d == 0 && c > r)) a = src[rem] * v.scale[4 * p + r] * v.scale[col];
                if (d == 0 && c == r) {
                    if (!reuse) v.diag[4 * p + r] = fmin(fmax(a, min_diag), max_diag);
                    a += v.diag[4 * p + r] / radius;
                }
                dst[rem] = a;
            }
        }
        for (int i = tid; i < n4; i += kPgT) { const double g = v.cur[hsz + i] * v.scale[i]; v.gs[i] = g; v.sol[i] = g; }
        __syncthreads();
        bool ok = pg_band_solve(v, s_pan, s_d, &s_flag);
        double dg = 0.0, dd = 0.0, bad = 0.0;
        if (ok)
            for (int i = tid; i < n4; i += kPgT) {
                const double st = -v.sol[i];
                if (!isfinite(st)) bad = 1.0;
                dg += st * v.gs[i]; dd += v.diag[i] / radius * st * st;
            }
        dg = pg_block_sum(dg, s_red); dd = pg_block_sum(dd, s_red); bad = pg_block_sum(bad, s_red);
        const double model_change = -0.5 * dg + 0.5 * dd;
        ok = ok && bad == 0.0 && model_change > 0.0;
        if (!ok) {
            __syncthreads();
            if (tid == 0) { s.invalid_steps++; if (s.invalid_steps >= 5) s.done = 1; s.radius *= 0.5; s.reuse_diag = 1; }
            __syncthreads();
            continue;
        }
        double sn = 0.0;
        for (int i = tid; i < n4; i += kPgT) {
            const int node = i >> 2, c = i & 3, q = 4 * v.pos[node] + c;
            const double xo = v.x[i], delta = -v.sol[q] * v.scale[q];
            const double xc = c == 0 ? pg_normalize_angle(xo + delta) : xo + delta;        // AngleLocalParameterization
            v.cand[i] = xc;
            sn += (xc - xo) * (xc - xo);
        }
        sn = sqrt(pg_block_sum(sn, s_red));
        if (tid == 0) {
            s.invalid_steps = 0; s.model_change = model_change; s.have_cand = 1;
            if (sn <= parameter_tol * (s.x_norm + parameter_tol)) { s.done = 1; s.have_cand = 0; }
        }
        __syncthreads();
        break;
    }
    __syncthreads();
    if (tid == 0) *v.st = s;
}

}  // namespace lmono
