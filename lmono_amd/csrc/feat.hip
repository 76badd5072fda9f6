// lmono_amd/csrc/feat.hip -- per-feature kernels of lmono's FeatureManager / Estimator (fp64), paths under
// /root/reference/mono_lidar_mapping:
//   k_triangulate_init  FeatureManager::triangulate linear part            src/image_process/FeatureManager.cc:75-195
//   k_depth_refine      FeatureManager::triangulate Ceres refinement + setDepth  :197-251, :38-56
//   k_outlier_scores    Estimator::reprojectionError / outliersRejection  src/image_process/Estimator.cc:104-190
//   k_shift_depth       FeatureManager::removeBackShiftDepth               src/image_process/FeatureManager.cc:540-590
// Every track is independent: one thread per feature; the refinement shares one trust region per window (one
// workgroup per window, block reductions for cost / model decrease / norms).
#include "common.hpp"

namespace lmono {

struct FeatBatch {
    int n_windows, track_cnt, window_size, max_iter;
    double weight;              // FACTOR_WEIGHT
    const int *feat_off;        // [W+1]
    const double *Rs, *Ps, *tlc;  // [W][11][9], [W][11][3], [W][16]
    const int *start_frame;     // [F]
    const int *obs_off;         // [F+1]
    const double *pts;          // [total obs][2]
    double *depth;              // [F] estimated_depth in/out
    int *solve_flag;            // [F]
    double *score;              // [F]
    double *x, *cand;           // [F] scratch (inverse depths)
};

namespace feat {

__device__ __forceinline__ void mm3(const double *A, const double *B, double *C)
{
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}
__device__ __forceinline__ void mtm3(const double *A, const double *B, double *C)
{
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) C[i * 3 + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
}
__device__ __forceinline__ void mv3(const double *A, const double *v, double *o)
{
    const double a = A[0] * v[0] + A[1] * v[1] + A[2] * v[2], b = A[3] * v[0] + A[4] * v[1] + A[5] * v[2], c = A[6] * v[0] + A[7] * v[1] + A[8] * v[2];
    o[0] = a; o[1] = b; o[2] = c;
}
__device__ __forceinline__ void mtv3(const double *A, const double *v, double *o)
{
    const double a = A[0] * v[0] + A[3] * v[1] + A[6] * v[2], b = A[1] * v[0] + A[4] * v[1] + A[7] * v[2], c = A[2] * v[0] + A[5] * v[1] + A[8] * v[2];
    o[0] = a; o[1] = b; o[2] = c;
}

// eigenvector of the symmetric 4x4 M for its smallest eigenvalue (cyclic Jacobi, fixed sweep order)
__device__ void smallest_eigvec4(const double *Min, double *v)
{
    double M[16], V[16];
    for (int i = 0; i < 16; i++) { M[i] = Min[i]; V[i] = (i % 5 == 0) ? 1.0 : 0.0; }
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

__device__ __forceinline__ int window_of(const int *feat_off, int n_windows, int f)
{
    int lo = 0, hi = n_windows;      // largest w with feat_off[w] <= f
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (feat_off[mid] <= f) lo = mid; else hi = mid; }
    return lo;
}

} // namespace feat

__global__ __launch_bounds__(128) void k_triangulate_init(FeatBatch B)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= B.feat_off[B.n_windows]) return;
    const int nobs = B.obs_off[f + 1] - B.obs_off[f];
    if (B.depth[f] > 0 || nobs < B.track_cnt) return;
    const int w = feat::window_of(B.feat_off, B.n_windows, f);
    const double *Rs = B.Rs + (size_t)w * 99, *Ps = B.Ps + (size_t)w * 33, *tlc = B.tlc + (size_t)w * 16;
    double Rlc[9], Tlc[3];
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) Rlc[i * 3 + j] = tlc[i * 4 + j]; Tlc[i] = tlc[i * 4 + 3]; }
    const int i = B.start_frame[f];
    double t0[3], R0[9], tmp[3];
    feat::mv3(Rs + 9 * i, Tlc, tmp); for (int k = 0; k < 3; k++) t0[k] = Ps[3 * i + k] + tmp[k];
    feat::mm3(Rs + 9 * i, Rlc, R0);
    double AtA[16];
    for (int k = 0; k < 16; k++) AtA[k] = 0.0;
    for (int o = 0; o < nobs; o++) {
        const int j = i + o;
        double t1[3], R1[9], d[3], t[3], R[9], P[12], Rt_t[3];
        feat::mv3(Rs + 9 * j, Tlc, tmp); for (int k = 0; k < 3; k++) t1[k] = Ps[3 * j + k] + tmp[k];
        feat::mm3(Rs + 9 * j, Rlc, R1);
        for (int k = 0; k < 3; k++) d[k] = t1[k] - t0[k];
        feat::mtv3(R0, d, t); feat::mtm3(R0, R1, R);
        feat::mtv3(R, t, Rt_t);
        for (int a = 0; a < 3; a++) { for (int b = 0; b < 3; b++) P[a * 4 + b] = R[b * 3 + a]; P[a * 4 + 3] = -Rt_t[a]; }
        const double px = B.pts[2 * (size_t)(B.obs_off[f] + o)], py = B.pts[2 * (size_t)(B.obs_off[f] + o) + 1];
        const double nn = sqrt(px * px + py * py + 1.0);
        const double fv[3] = { px / nn, py / nn, 1.0 / nn };
        for (int rr = 0; rr < 2; rr++) {
            double row[4];
            for (int k = 0; k < 4; k++) row[k] = fv[rr] * P[8 + k] - fv[2] * P[rr * 4 + k];
            for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) AtA[a * 4 + b] += row[a] * row[b];
        }
    }
    double v[4];
    feat::smallest_eigvec4(AtA, v);
    const double z = v[2] / v[3];
    B.depth[f] = (z < 0.1) ? -1.0 : z;
}

// ReprojectionFactor::Evaluate (ba::reproj_factor, ba.hip) with the product T = Rlc^T Rj^T Ri Rlc of its Jacobian handed in: T depends on the frame
// pair only, and the refinement evaluates every observation of every track in every iteration.  Same expressions in the same order: same bits.
__device__ __forceinline__ void reproj_factor_T(double xf, const double *p_i2, const double *p_j2, const double *Ri, const double *Pi, const double *Rj, const double *Pj,
                                                const double *Rlc, const double *Tlc, const double *T, double w, double *r, double *J)
{
    const double p_i[3] = { p_i2[0], p_i2[1], 1.0 }, p_j[3] = { p_j2[0], p_j2[1], 1.0 };
    const double dep = 1.0 / xf;
    const double pc[3] = { dep * p_i[0], dep * p_i[1], dep * p_i[2] };
    double pl[3], pw[3], plj[3], pcj[3], tmp[3];
    feat::mv3(Rlc, pc, pl); for (int k = 0; k < 3; k++) pl[k] += Tlc[k];
    feat::mv3(Ri, pl, pw); for (int k = 0; k < 3; k++) pw[k] += Pi[k];
    for (int k = 0; k < 3; k++) tmp[k] = pw[k] - Pj[k];
    feat::mtv3(Rj, tmp, plj);
    for (int k = 0; k < 3; k++) tmp[k] = plj[k] - Tlc[k];
    feat::mtv3(Rlc, tmp, pcj);
    const double d = pcj[2];
    r[0] = w * (pcj[0] / d - p_j[0]);
    r[1] = w * (pcj[1] / d - p_j[1]);
    const double red[6] = { w * (1.0 / d), 0, w * (-pcj[0] / (d * d)), 0, w * (1.0 / d), w * (-pcj[1] / (d * d)) };
    double v[3];
    feat::mv3(T, p_i, v);
    for (int a = 0; a < 2; a++) J[a] = -(red[a * 3] * v[0] + red[a * 3 + 1] * v[1] + red[a * 3 + 2] * v[2]) * dep * dep;
}

// cost and (h, g) of one feature's ReprojectionFactor blocks at inverse depth xf; sT: the window's T(i, j) = Rlc^T Rj^T Ri Rlc, [11][11][9] in LDS
template <bool kJac>
__device__ __forceinline__ double refine_feature(const FeatBatch &B, int f, const double *Rs, const double *Ps, const double *Rlc, const double *Tlc, const double *sT, double xf, double &h, double &g)
{
    double cost = 0;
    h = 0; g = 0;
    const int nobs = B.obs_off[f + 1] - B.obs_off[f];
    if (nobs < B.track_cnt) return 0.0;
    const int i = B.start_frame[f];
    const double *pi = B.pts + 2 * (size_t)B.obs_off[f];
    for (int o = 1; o < nobs; o++) {
        const int j = i + o;
        if (j == B.window_size) continue;
        double r[2], J[2];
        const double *pj = B.pts + 2 * (size_t)(B.obs_off[f] + o);
        reproj_factor_T(xf, pi, pj, Rs + 9 * i, Ps + 3 * i, Rs + 9 * j, Ps + 3 * j, Rlc, Tlc, sT + (i * 11 + j) * 9, B.weight, r, J);
        const double sq = r[0] * r[0] + r[1] * r[1];
        cost += 0.5 * log(1.0 + sq);
        if (kJac) { double rho1 = 1.0 / (1.0 + sq); rho1 = rho1 > DBL_MIN ? rho1 : DBL_MIN; h += rho1 * (J[0] * J[0] + J[1] * J[1]); g += rho1 * (J[0] * r[0] + J[1] * r[1]); }
    }
    return cost;
}

__global__ __launch_bounds__(256) void k_depth_refine(FeatBatch B)
{
    const int w = blockIdx.x, tid = threadIdx.x;
    const int f0 = B.feat_off[w], f1 = B.feat_off[w + 1];
    const double *gRs = B.Rs + (size_t)w * 99, *gPs = B.Ps + (size_t)w * 33, *tlc = B.tlc + (size_t)w * 16;
    __shared__ double red[8];
    // the window's poses, the extrinsic and the Jacobians' frame-pair products T(i, j) = Rlc^T Rj^T Ri Rlc in LDS (round 5: every observation of every
    // iteration recomputed T with three 3 x 3 products -- 60 % of an evaluation; the products are formed exactly as ba::reproj_factor forms them)
    __shared__ double Rs[99], Ps[33], sRlc[9], sTlc[3], sT[121 * 9];
    if (tid < 99) Rs[tid] = gRs[tid];
    if (tid < 33) Ps[tid] = gPs[tid];
    if (tid < 9) sRlc[tid] = tlc[(tid / 3) * 4 + tid % 3];
    if (tid < 3) sTlc[tid] = tlc[tid * 4 + 3];
    __syncthreads();
    if (tid < 121) {
        const int i = tid / 11, j = tid % 11;
        double RjT[9], RlcT[9], T[9];
        ba::mT(Rs + 9 * j, RjT); ba::mT(sRlc, RlcT);
        ba::mm(RlcT, RjT, T); ba::mm(T, Rs + 9 * i, T); ba::mm(T, sRlc, T);
        for (int k = 0; k < 9; k++) sT[tid * 9 + k] = T[k];
    }
    __syncthreads();
    const double *Rlc = sRlc, *Tlc = sTlc;
    // per-thread feature slots: features f0 + tid + 256 m, at most kSlots per thread
    constexpr int kSlots = (LMONO_BA_MAX_FEATURES + 255) / 256;   // tracks per thread: a window holds at most LMONO_BA_MAX_FEATURES
    double x[kSlots], h[kSlots], g[kSlots], scale[kSlots], diag[kSlots], step[kSlots], cand[kSlots];
    bool act[kSlots];
    for (int m = 0; m < kSlots; m++) {
        const int f = f0 + tid + 256 * m;
        act[m] = false; x[m] = 0; h[m] = 0; g[m] = 0; scale[m] = 1; diag[m] = 1; step[m] = 0; cand[m] = 0;
        if (f < f1) {
            x[m] = 1.0 / B.depth[f];
            const int nobs = B.obs_off[f + 1] - B.obs_off[f];
            int nres = 0;
            if (nobs >= B.track_cnt) for (int o = 1; o < nobs; o++) if (B.start_frame[f] + o != B.window_size) nres++;
            act[m] = nres > 0;
        }
    }
    auto bsum = [&](double v) {
        v = wave_sum_d(v);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = v;
        __syncthreads();
        return ((red[0] + red[1]) + red[2]) + red[3];
    };
    auto bmax = [&](double v) {
        for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = v;
        __syncthreads();
        return fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    };
    const double function_tol = 1e-6, gradient_tol = 1e-10, parameter_tol = 1e-8, min_rel = 1e-3, min_diag = 1e-6, max_diag = 1e32;
    double radius = 1e4, dec = 2.0;
    bool reuse = false;
    int invalid = 0, iter = 0;
    double c0 = 0, xn = 0, gm = 0;
    for (int m = 0; m < kSlots; m++) {
        const int f = f0 + tid + 256 * m;
        if (f < f1) c0 += refine_feature<true>(B, f, Rs, Ps, Rlc, Tlc, sT, x[m], h[m], g[m]);
        if (act[m]) { xn += x[m] * x[m]; scale[m] = 1.0 / (1.0 + sqrt(h[m])); gm = fmax(gm, fabs(g[m])); }
    }
    double x_cost = bsum(c0), x_norm = sqrt(bsum(xn)), gmax = bmax(gm);
    if (gmax > gradient_tol) while (iter < B.max_iter) {
        iter++;
        double model = 0, bad = 0;
        for (int m = 0; m < kSlots; m++) {
            step[m] = 0;
            if (!act[m]) continue;
            const double hs = h[m] * scale[m] * scale[m], gs = g[m] * scale[m];
            if (!reuse) { double d = hs; d = d < min_diag ? min_diag : (d > max_diag ? max_diag : d); diag[m] = d; }
            const double s = -gs / (hs + diag[m] / radius);
            if (!isfinite(s)) bad = 1;
            step[m] = s;
            model += -(s * gs + 0.5 * s * hs * s);
        }
        model = bsum(model); bad = bmax(bad);
        if (bad > 0 || !(model > 0.0)) { if (++invalid >= 5) break; radius *= 0.5; reuse = true; continue; }
        invalid = 0;
        // the candidate is evaluated WITH its Jacobian sums (round 4): an accepted step -- the usual case -- is the next linearisation, which was a
        // second pass over the same observations computing the same residuals (same arithmetic: the cost, and with it every decision, is unchanged)
        double sn = 0, cc = 0, hh[kSlots], gg[kSlots];
        for (int m = 0; m < kSlots; m++) {
            const int f = f0 + tid + 256 * m;
            cand[m] = x[m] + step[m] * scale[m];
            hh[m] = 0; gg[m] = 0;
            if (act[m]) sn += (cand[m] - x[m]) * (cand[m] - x[m]);
            if (f < f1) cc += refine_feature<true>(B, f, Rs, Ps, Rlc, Tlc, sT, cand[m], hh[m], gg[m]);
        }
        sn = sqrt(bsum(sn)); cc = bsum(cc);
        if (sn <= parameter_tol * (x_norm + parameter_tol)) break;
        if (fabs(x_cost - cc) <= function_tol * x_cost) break;
        const double rel = (x_cost - cc) / model;
        if (rel > min_rel) {
            xn = 0; gm = 0;
            for (int m = 0; m < kSlots; m++) {
                x[m] = cand[m]; h[m] = hh[m]; g[m] = gg[m];
                if (act[m]) { xn += x[m] * x[m]; gm = fmax(gm, fabs(g[m])); }
            }
            x_cost = cc; x_norm = sqrt(bsum(xn)); gmax = bmax(gm);
            const double t = 2.0 * rel - 1.0;
            double den = 1.0 - t * t * t; if (den < 1.0 / 3.0) den = 1.0 / 3.0;
            radius = radius / den; if (radius > 1e16) radius = 1e16;
            dec = 2.0; reuse = false;
            if (gmax <= gradient_tol) break;
        } else { radius /= dec; dec *= 2.0; reuse = true; }
        if (radius <= 1e-32) break;
    }
    for (int m = 0; m < kSlots; m++) {
        const int f = f0 + tid + 256 * m;
        if (f >= f1) continue;
        const int nobs = B.obs_off[f + 1] - B.obs_off[f];
        B.solve_flag[f] = 0;
        if (nobs < B.track_cnt) continue;
        const double d = 1.0 / x[m];
        B.depth[f] = d;
        B.solve_flag[f] = (d < 0.1 || d > 300) ? 2 : 1;
    }
}

// ---- k_depth_refine, one observation per thread (round 5).  The kernel above gives a track to a thread, which walks its <= 10 observations one after the
// other in every evaluation: ~110 us per Estimator frame for ~150 tracks, a chain of ~3000 dependent fp64 instructions per thread and evaluation on a
// fraction of one compute unit.  Here every (track, observation) pair of the window is an ITEM; an evaluation computes the items side by side on 1024
// threads, leaves (cost, h, g) of each in LDS, and the track's thread adds its items in observation order -- the same terms in the same order as the walk,
// so every sum, every decision and every result is the same bit for bit.  Windows above kDrItems items or 1024 tracks take the kernel above.
constexpr int kDrT = 1024, kDrItems = 3072;
__global__ __launch_bounds__(kDrT) void k_depth_refine_items(FeatBatch B)
{
    const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int f0 = B.feat_off[w], f1 = B.feat_off[w + 1];
    const double *gRs = B.Rs + (size_t)w * 99, *gPs = B.Ps + (size_t)w * 33, *tlc = B.tlc + (size_t)w * 16;
    __shared__ double red[kDrT / 64];
    __shared__ double Rs[99], Ps[33], sRlc[9], sTlc[3], sT[121 * 9];
    __shared__ double s_x[kDrT];                       // the inverse depths an evaluation is made at
    __shared__ double s_res[kDrItems][3];              // (cost, h, g) of every item
    __shared__ unsigned short s_if[kDrItems];          // item -> track (thread index)
    __shared__ unsigned char s_io[kDrItems];           // item -> observation
    __shared__ int s_wtot[kDrT / 64], s_nitems;
    if (tid < 99) Rs[tid] = gRs[tid];
    if (tid < 33) Ps[tid] = gPs[tid];
    if (tid < 9) sRlc[tid] = tlc[(tid / 3) * 4 + tid % 3];
    if (tid < 3) sTlc[tid] = tlc[tid * 4 + 3];
    __syncthreads();
    if (tid < 121) {
        const int i = tid / 11, j = tid % 11;
        double RjT[9], RlcT[9], T[9];
        ba::mT(Rs + 9 * j, RjT); ba::mT(sRlc, RlcT);
        ba::mm(RlcT, RjT, T); ba::mm(T, Rs + 9 * i, T); ba::mm(T, sRlc, T);
        for (int k = 0; k < 9; k++) sT[tid * 9 + k] = T[k];
    }
    // the track of this thread and its items
    const int f = f0 + tid;
    const bool has = f < f1;
    const int ob0 = has ? B.obs_off[f] : 0, nobs = has ? B.obs_off[f + 1] - ob0 : 0, sf = has ? B.start_frame[f] : 0;
    const bool cnt_ok = has && nobs >= B.track_cnt;
    int nres = 0;
    if (cnt_ok) for (int o = 1; o < nobs; o++) if (sf + o != B.window_size) nres++;
    const bool act = nres > 0;
    double x = has ? 1.0 / B.depth[f] : 0.0, h = 0, g = 0, scale = 1, diag = 1, step = 0, cand = 0;
    int it0;
    {
        const int incl = wave_scan_incl(nres);
        if (lane == 63) s_wtot[wave] = incl;
        __syncthreads();
        it0 = incl - nres;
        for (int q = 0; q < wave; q++) it0 += s_wtot[q];
        if (tid == kDrT - 1) s_nitems = it0 + nres;
        if (cnt_ok) { int k = it0; for (int o = 1; o < nobs; o++) if (sf + o != B.window_size) { s_if[k] = (unsigned short)tid; s_io[k] = (unsigned char)o; k++; } }
    }
    __syncthreads();
    const int n_items = s_nitems;
    auto bsum = [&](double v) {
        v = wave_sum_d(v);
        __syncthreads();
        if (lane == 0) red[wave] = v;
        __syncthreads();
        // (the tracks of a window sit in the first waves: the order of this sum is the 256-thread kernel's -- ((w0 + w1) + w2) + w3 -- with zeros behind it)
        double t = red[0];
        for (int q = 1; q < kDrT / 64; q++) t += red[q];
        return t;
    };
    auto bmax = [&](double v) {
        for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
        __syncthreads();
        if (lane == 0) red[wave] = v;
        __syncthreads();
        double t = red[0];
        for (int q = 1; q < kDrT / 64; q++) t = fmax(t, red[q]);
        return t;
    };
    // cost of the track at xv, with (h, g): the items side by side, then the track's own items added in observation order
    auto evaluate = [&](double xv, double &ho, double &go) -> double {
        s_x[tid] = xv;
        __syncthreads();
        for (int k = tid; k < n_items; k += kDrT) {
            const int ft = (int)s_if[k], o = (int)s_io[k], fg = f0 + ft;
            const int i = B.start_frame[fg], j = i + o;
            const double *pi = B.pts + 2 * (size_t)B.obs_off[fg], *pj = pi + 2 * (size_t)o;
            double r[2], J[2];
            reproj_factor_T(s_x[ft], pi, pj, Rs + 9 * i, Ps + 3 * i, Rs + 9 * j, Ps + 3 * j, sRlc, sTlc, sT + (i * 11 + j) * 9, B.weight, r, J);
            const double sq = r[0] * r[0] + r[1] * r[1];
            double rho1 = 1.0 / (1.0 + sq); rho1 = rho1 > DBL_MIN ? rho1 : DBL_MIN;
            s_res[k][0] = 0.5 * log(1.0 + sq); s_res[k][1] = rho1 * (J[0] * J[0] + J[1] * J[1]); s_res[k][2] = rho1 * (J[0] * r[0] + J[1] * r[1]);
        }
        __syncthreads();
        double cost = 0;
        ho = 0; go = 0;
        for (int k = it0; k < it0 + nres; k++) { cost += s_res[k][0]; ho += s_res[k][1]; go += s_res[k][2]; }
        return cost;
    };
    const double function_tol = 1e-6, gradient_tol = 1e-10, parameter_tol = 1e-8, min_rel = 1e-3, min_diag = 1e-6, max_diag = 1e32;
    double radius = 1e4, dec = 2.0;
    bool reuse = false;
    int invalid = 0, iter = 0;
    double c0 = evaluate(x, h, g), xn = 0, gm = 0;
    if (act) { xn = x * x; scale = 1.0 / (1.0 + sqrt(h)); gm = fabs(g); }
    double x_cost = bsum(c0), x_norm = sqrt(bsum(xn)), gmax = bmax(gm);
    if (gmax > gradient_tol) while (iter < B.max_iter) {
        iter++;
        double model = 0, bad = 0;
        step = 0;
        if (act) {
            const double hs = h * scale * scale, gs = g * scale;
            if (!reuse) { double d = hs; d = d < min_diag ? min_diag : (d > max_diag ? max_diag : d); diag = d; }
            const double sv = -gs / (hs + diag / radius);
            if (!isfinite(sv)) bad = 1;
            step = sv;
            model = -(sv * gs + 0.5 * sv * hs * sv);
        }
        model = bsum(model); bad = bmax(bad);
        if (bad > 0 || !(model > 0.0)) { if (++invalid >= 5) break; radius *= 0.5; reuse = true; continue; }
        invalid = 0;
        cand = x + step * scale;
        double hh = 0, gg = 0;
        double sn = act ? (cand - x) * (cand - x) : 0.0;
        double cc = evaluate(cand, hh, gg);
        sn = sqrt(bsum(sn)); cc = bsum(cc);
        if (sn <= parameter_tol * (x_norm + parameter_tol)) break;
        if (fabs(x_cost - cc) <= function_tol * x_cost) break;
        const double rel = (x_cost - cc) / model;
        if (rel > min_rel) {
            x = cand; h = hh; g = gg;
            xn = act ? x * x : 0.0; gm = act ? fabs(g) : 0.0;
            x_cost = cc; x_norm = sqrt(bsum(xn)); gmax = bmax(gm);
            const double t = 2.0 * rel - 1.0;
            double den = 1.0 - t * t * t; if (den < 1.0 / 3.0) den = 1.0 / 3.0;
            radius = radius / den; if (radius > 1e16) radius = 1e16;
            dec = 2.0; reuse = false;
            if (gmax <= gradient_tol) break;
        } else { radius /= dec; dec *= 2.0; reuse = true; }
        if (radius <= 1e-32) break;
    }
    if (has) {
        B.solve_flag[f] = 0;
        if (nobs >= B.track_cnt) {
            const double d = 1.0 / x;
            B.depth[f] = d;
            B.solve_flag[f] = (d < 0.1 || d > 300) ? 2 : 1;
        }
    }
}

__global__ __launch_bounds__(128) void k_outlier_scores(FeatBatch B)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= B.feat_off[B.n_windows]) return;
    const int nobs = B.obs_off[f + 1] - B.obs_off[f];
    B.score[f] = -1.0;
    if (nobs < B.track_cnt) return;
    const int w = feat::window_of(B.feat_off, B.n_windows, f);
    const double *Rs = B.Rs + (size_t)w * 99, *Ps = B.Ps + (size_t)w * 33, *tlc = B.tlc + (size_t)w * 16;
    double Rlc[9], Tlc[3];
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) Rlc[i * 3 + j] = tlc[i * 4 + j]; Tlc[i] = tlc[i * 4 + 3]; }
    const int i = B.start_frame[f];
    const double *pi = B.pts + 2 * (size_t)B.obs_off[f];
    const double dep = B.depth[f];
    double err = 0; int cnt = 0;
    for (int o = 1; o < nobs; o++) {
        const int j = i + o;
        const double *pj = B.pts + 2 * (size_t)(B.obs_off[f] + o);
        const double pc[3] = { dep * pi[0], dep * pi[1], dep };
        double pl[3], pw[3], plj[3], pcj[3], d[3];
        feat::mv3(Rlc, pc, pl); for (int k = 0; k < 3; k++) pl[k] += Tlc[k];
        feat::mv3(Rs + 9 * i, pl, pw); for (int k = 0; k < 3; k++) d[k] = pw[k] + Ps[3 * i + k] - Ps[3 * j + k];
        feat::mtv3(Rs + 9 * j, d, plj); for (int k = 0; k < 3; k++) d[k] = plj[k] - Tlc[k];
        feat::mtv3(Rlc, d, pcj);
        const double rx = pcj[0] / pcj[2] - pj[0], ry = pcj[1] / pcj[2] - pj[1];
        err += sqrt(rx * rx + ry * ry); cnt++;
    }
    B.score[f] = (err / cnt) * B.weight;
}

// poses: [0..8] back_R0, [9..11] back_P0, [12..20] Rs[0] after the slide, [21..23] Ps[0], [24..39] TLC -- one such record per window, win[f] = the
// window of track f (nullptr: one window)
__global__ __launch_bounds__(128) void k_shift_depth(const double *poses, int n, const double *pt_i, const double *depth, double *depth_out, const int *win)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n) return;
    if (win) poses += (size_t)40 * win[f];
    const double *tlc = poses + 24;
    double Rlc[9], Tlc[3], R0[9], R1[9], P0[3], P1[3], tmp[3];
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) Rlc[i * 3 + j] = tlc[i * 4 + j]; Tlc[i] = tlc[i * 4 + 3]; }
    feat::mm3(poses, Rlc, R0); feat::mm3(poses + 12, Rlc, R1);
    feat::mv3(poses, Tlc, tmp); for (int k = 0; k < 3; k++) P0[k] = poses[9 + k] + tmp[k];
    feat::mv3(poses + 12, Tlc, tmp); for (int k = 0; k < 3; k++) P1[k] = poses[21 + k] + tmp[k];
    const double pi[3] = { pt_i[2 * f] * depth[f], pt_i[2 * f + 1] * depth[f], depth[f] };
    double w[3], d[3], pj[3];
    feat::mv3(R0, pi, w); for (int k = 0; k < 3; k++) d[k] = w[k] + P0[k] - P1[k];
    feat::mtv3(R1, d, pj);
    depth_out[f] = pj[2] > 0 ? pj[2] : -1.0;
}

} // namespace lmono
