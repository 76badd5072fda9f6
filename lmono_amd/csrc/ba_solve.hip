// lmono_amd/csrc/ba_solve.hip -- sliding-window BA solve on gfx950 (fp64): one 512-thread workgroup per window,
// windows batched across the grid (independent sequences / streams; a window depends on its predecessor, so one
// sequence alone cannot batch -- SURVEY.md 8e).
//
// Restates the solve inside Estimator::optimization() (/root/reference/mono_lidar_mapping/src/image_process/
// Estimator.cc:1124-1305): PriorFactor on the extrinsic, LASERFactor chain, MonoProjectionFactor + CauchyLoss(1) per
// (feature, observation), ceres::Solve(DENSE_SCHUR, DOGLEG, max_num_iterations) with Ceres defaults (Appendix B of
// SURVEY.md): Jacobi scaling, traditional dogleg on the elliptical trust region, Schur elimination of the 1-D inverse
// depth blocks into the 72x72 reduced system, Cholesky, step acceptance and radius update.
//
// Data flow of one linearisation (ba_evaluate<true>):
//   * every 3x3 product of MonoProjectionFactor::Evaluate (MonoProjectionFactor.cc:40-174) that depends only on the
//     frame pair (i, j) and the extrinsic is computed once per pair (<= 110 pairs) instead of once per observation;
//   * observations are visited in (i, j)-pair order, two threads each (one per residual row); the robustified Jacobian
//     rows [J_i | J_j | J_ex | r] go to an LDS stage (192 observations per pass);
//   * the camera part of J^T J is a sum of small GEMMs, one per frame pair: the 8 waves walk the staged rows four at a
//     time with v_mfma_f64_16x16x4_f64 (A = B^T = the same staged value) and add the finished 16x16 blocks into the
//     LDS-resident 72x72 H_pp once per pair -- no per-observation atomics;
//   * depth blocks H_ff, g_f and the coupling rows H_fp (feature-major, 80 doubles, in an HBM/L2 scratch): every observation
//     leaves its shares in its own 128-B record, one pass per feature sums them in observation order.
// Every sum of the linearisation is formed in an order fixed by the problem alone (static pair schedule, turn-ordered block
// additions, per-feature passes): two runs of the same problem give the same bits.
// Schur complement S = H_pp - sum_f e_f e_f^T / h_ff: 64-feature tiles staged in LDS, 15 upper 16x16 tiles of S
// accumulated in registers with the same f64 MFMA; blocked Cholesky (8-column panels) and the triangular solves in
// LDS.  fp64 throughout; contraction to FMA is allowed in this file (the BA parity bar is a tolerance).
#include "common.hpp"
#include <type_traits>

#pragma clang fp contract(fast)

namespace lmono {

constexpr int kBaMaxPoses = 11;
constexpr int kBaP = 6 * (kBaMaxPoses + 1);      // 72
constexpr int kBaPS = 80;                        // padded row of the coupling matrix: 5 MFMA tiles of 16
constexpr int kBaMaxFeat = LMONO_BA_MAX_FEATURES;      // include/lmono_hip.h states the bound and why
constexpr int kBaLdsFeat = 448;                  // features whose dogleg vectors live in LDS; a window above it runs the kBig instantiation (vectors in an L2 scratch)
constexpr int kBaN = kBaP + kBaLdsFeat;          // 520
constexpr int kBaMaxPairs = kBaMaxPoses * (kBaMaxPoses - 1);
constexpr int kBaRound = 32;                     // observations a wave stages per round (two lanes each)
constexpr int kBaRow = 19;                       // staged row: J_i(6) J_j(6) J_ex(6) r (odd stride: few bank conflicts)
constexpr int kBaFT = 64;                        // features per Schur tile
constexpr int kBaSS = 73;                        // row stride of S in LDS (odd: conflict-free column walks)
constexpr int kBaPairRec = 52;                   // Tres(9) tres(3) Cm(9) A(9) B(9) Tn(9) tn(3) pad
constexpr int kBaPairTile = 320;                 // 16 x 16 tile + 16 x 4 side tile of a SEGMENT of a frame pair (column 3 of the side tile: the segment's scalar sums)
constexpr int kBaSeg = 16;                       // observations of a segment: a frame pair's slots in runs of 16 (two lanes each: half a wave)
constexpr int kBaMaxSeg = kBaMaxPairs + kBaLdsFeat * (kBaMaxPoses - 1) / kBaSeg;      // 110 + 280 (segments whose table lives in LDS; kBig reads the table from HBM)
constexpr int kBaMbox = 96 + kBaMaxFeat;         // leader -> followers: [0] command, [1] re-use the records, [8..84] poses, [88..94] extrinsic, [96..] inverse depths
constexpr int kBaMaxK = 8;                       // workgroups per window
constexpr int kBaBar = 32;                       // flag words per window: [0] go, [r] follower r's "segments done", [8] the leader's, [8 + r] follower r's "sums done", [16 + r] its XCD + 1
// ba_reduce_pairs' gather program (built once per solve: which tile cells an entry of H_pp / g_p is the sum of does not change between iterations):
// part A = the 36 entries of every lower frame-block pair (two sources, written twice: the tiles are symmetric), part B = per frame the 21 lower
// entries of its diagonal block, its 36 entries against the extrinsic and its 6 gradient entries (22 sources: the pairs (f, k), then the pairs (k, f))
constexpr int kBaGaN = 2048;                     // part A items (55 block pairs x 36 = 1980, padded): fields source 1, source 2, destination, mirror
constexpr int kBaGbN = 704;                      // part B items (11 frames x 63 = 693, padded): fields 22 sources, destination, mirror
constexpr int kBaGprog = 4 * kBaGaN + 24 * kBaGbN;
constexpr int kBaObsRec = 24;                    // per-observation record: hdd, gd, hx[6], hi[6], hj[6], oj
constexpr int kBaT = 512;                        // threads per workgroup (2 waves per SIMD)
#ifndef LMONO_BA_HS_FUSED
#define LMONO_BA_HS_FUSED 1                      // the H s product reads the coupling rows once (1) or once for the camera rows and once for the depth rows (0: rounds 1-5)
#endif
constexpr int kBaW = kBaT / 64;

typedef double ba_d4 __attribute__((ext_vector_type(4)));

struct BaBatch {
    int n_windows;
    int max_iter;
    const int *feat_off;        // [W+1]
    const int *obs_off;         // [W+1]
    const int *flags;           // [W][4] n_poses, use_prior, ex_constant, use_mono
    double *poses;              // [W][11][7]
    double *ex;                 // [W][7]
    double *inv_depth;          // [total F]
    const int *feat_anchor;     // [total F] frame the feature is anchored in (-1: no observation)
    const int *pair_off;        // [W+1] first (i, j) frame pair of each window
    const int *pair_ij;         // [total pairs] i | j << 8
    const int *pobs_off;        // [W+1] first slot of each window in the pair-ordered observation list
    const int *pair_slot;       // [total pairs + W] per window n_pairs + 1 window-local slot offsets; pairs sorted by descending size
    const int *seg_off;         // [W+1] first segment of each window
    const unsigned short *seg_tab;  // [total segments] pair (window-local, 7 bits) | index of the segment inside its pair << 7
    const int *pair_seg;        // [total pairs + W] per window n_pairs + 1 window-local first segments of the pairs
    const int *n_multi;         // [W] pairs of more than one segment (they come first: the pairs are sorted by descending size)
    const int *slot_info;       // [total slots] feature | pair << 16 (both window-local), slots ordered by pair
    const double *slot_pts;     // [total slots][4] the observation's two normalised image points, in slot order
    const double *laser_consts; // [W][10][24]
    const double *prior_T;      // [W][16]
    const double *info;         // laser_info[36], mono_info[4], prior_w[2]
    double *hpd;                // scratch [W][kBaMaxFeat][kBaPS] coupling rows, feature-major
    double *pairdat;            // scratch [total pairs][kBaPairRec]
    const int *feat_obs_off;    // [total F + 1] first observation of every feature (observations grouped by feature, host order)
    const int *slot_obs;        // [total slots] observation (host order, global index) behind every slot
    double *obsc;               // scratch [total obs][kBaObsRec]: per-observation depth / coupling contributions (hdd, gd, hx[6], hi[6], hj[6], oj),
                                // in host observation order = grouped by feature: a feature's records are contiguous
    double *pairH;              // scratch [total segments + total pairs + W][kBaPairTile], per window [its segments | its pairs | one tile of zeros]: every segment's J^T [J r] tile
                                // (16 x 16 + 16 x 4) and scalar sums; a pair of several segments has their sum (segment order) in its own tile (ba_reduce_pairs)
    double *cpart;              // scratch [total segments]: every segment's share of a candidate's cost
    // several workgroups per window (k_ba_solve<true>): the leader's mail box [W][kBaMbox] (command, state to evaluate), flag words [W][16]
    // (go, done of every follower; zeroed before every launch), a failure flag; pairdat then holds one copy per workgroup of a window
    int *gprog;                 // scratch [W][kBaGprog]: every window's gather program (see kBaGaN)
    double *hred;               // scratch [W][kBaHred]: H_pp | g_p of a linearisation whose ordered sums the cluster's workgroups share (ba_reduce_pairs)
    double *fdg;                // scratch [W][2][feat_cap]: H_ff | g_f likewise (LDS-resident windows; a kBig window's live in bigv anyway)
    int feat_cap;               // stride of the per-window feature arrays hpd / cand / mbox: kBaLdsFeat, or kBaMaxFeat when a window of the batch is larger
    double *bigv;               // scratch [W][8][kBaMaxFeat] (kBig only): H_ff, g_f and the feature part of scale, D, gs, gn, va, vb
    double *mbox;
    unsigned int *bar;
    int *fail;
    int n_pairs_total;          // stride of the per-rank copies of pairdat
    double *cand;               // scratch [W][kBaMaxFeat]
    double *summary;            // [W][6] initial_cost, final_cost, iterations, termination, successful, unsuccessful
    int lds_ok;                 // every observation's anchor frame precedes its observer (i < j: what the Estimator produces): the one-workgroup solve may add a
                                // pair's tiles into H_pp in pair order straight from LDS (ba_linearise_lds); 0: it goes through the scratch like a cluster
    const char *blob_lo, *blob_hi;  // the batch's one device allocation (every pointer above points into it): the bounds-checked build's limits
};

struct BaSchurStage { double et[kBaFT * kBaPS]; double ic[kBaFT]; double gi[kBaFT]; double part[4][kBaPS]; };
struct BaFactor { double S[(kBaP + 1) * kBaSS]; double idiag[kBaP]; };   // reduced system / its Cholesky factor (row P = right-hand side), 1 / diagonal

struct BaLds {
    double Hpp[kBaP * kBaP];
    union {
        double stage[kBaW * 2 * kBaRound * kBaRow];   // evaluate: every wave's staged Jacobian rows
        double hs_part[kBaW][kBaPS];                  // Hs v: per-wave partial sums of the camera rows
        BaSchurStage sch;                         // Schur: scaled coupling tile, 1 / h_ff
        BaFactor fac;
    } u;
    double Hdd[kBaLdsFeat], gdd[kBaLdsFeat];
    double gp[kBaP];
    double scale[kBaN], D[kBaN], gs[kBaN], gn[kBaN], va[kBaN], vb[kBaN];
    double rhs[kBaPS];
    double red[3 * kBaW];
    double poses[kBaMaxPoses * 7], ex[7];
    double cposes[kBaMaxPoses * 7], cex[7];
    double Rp[(kBaMaxPoses + 1) * 9];            // rotation matrices of the window poses, then R_lc (normalised quaternions: Jacobians)
    double Mq[(kBaMaxPoses + 1) * 18 + 9];       // per pose and for the extrinsic: M(q), M(q^-1) of the RAW quaternion (residual path); then M(qx^-1)^-1
    double vinv[kBaLdsFeat];                     // inverse depths of the state being evaluated
    int pair_ij[kBaMaxPairs];
    short pair_slot[kBaMaxPairs + 1];            // first slot of every pair (pairs in descending size)
    // The linearisation's unit of work is a SEGMENT: up to 16 consecutive slots of one frame pair (round 5).  Every sum that crosses observations is
    // formed per segment (its J^T [J r] tile, its share of the cost and of the extrinsic corner) and the segments' results are added in segment
    // order, so the bits depend neither on which wave takes a segment nor on how many workgroups share a window.
#ifdef LMONO_BA_PROF
    unsigned short seg[kBaMaxSeg - 64];          // (the profiling build's clock cells need the room; its windows hold fewer segments)
#else
    unsigned short seg[kBaMaxSeg];               // pair | index inside the pair << 7
#endif
    short pair_seg[kBaMaxPairs + 1];             // first segment of every pair
    short pair_of[kBaMaxPoses * kBaMaxPoses];    // pair (anchor i, observer j) -> index in the window's pair list, -1: none
    unsigned short fobs[kBaLdsFeat + 1];         // first observation of every feature, relative to the window's first (host order: grouped by feature)
    signed char fanchor[kBaLdsFeat];             // the frame a feature is anchored in (-1: no observation)
    int ok;
    int eval_no, failed;        // several workgroups per window: evaluations handed out so far; a workgroup did not arrive
    int coloc;                  // the leader's: -1 unknown, 1 every workgroup of the cluster runs on the leader's XCD (then they share the ordered sums), 0 not
#ifdef LMONO_BA_PROF
    unsigned long long prof[24];
#endif
};
static_assert(sizeof(BaLds) <= 160 * 1024, "BaLds exceeds the 160 KB LDS of a gfx950 CU");
// The workgroup's state is a STATIC LDS object (gfx950 takes 160 KB of it).  As dynamic LDS, every out-of-line phase of k_ba_solve found its base through
// llvm.amdgcn.dynlds.offset.table -- a global_load + s_waitcnt vmcnt(0) that the compiler, short of registers, repeated in front of LDS stores all over
// the linearisation; a module-scope __shared__ object sits at a link-time address and every access is a ds_ instruction with an immediate.
__shared__ BaLds g_ba_lds;
#define BA_BIND_LDS(arg) (void)arg; BaLds &L = g_ba_lds;

__device__ __forceinline__ double block_sum(double v, double *red)
{
    v = wave_sum_d(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0;
#pragma unroll
    for (int w = 0; w < kBaW; w++) s += red[w];
    return s;
}
__device__ __forceinline__ void block_sum3(double &a, double &b, double &c, double *red)
{
    a = wave_sum_d(a); b = wave_sum_d(b); c = wave_sum_d(c);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { const int w = threadIdx.x >> 6; red[w] = a; red[kBaW + w] = b; red[2 * kBaW + w] = c; }
    __syncthreads();
    a = 0; b = 0; c = 0;
#pragma unroll
    for (int w = 0; w < kBaW; w++) { a += red[w]; b += red[kBaW + w]; c += red[2 * kBaW + w]; }
}
__device__ __forceinline__ double block_max(double v, double *red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double m = red[0];
#pragma unroll
    for (int w = 1; w < kBaW; w++) m = fmax(m, red[w]);
    return m;
}

#ifdef LMONO_BA_PROF
// phase clocks of window 0's leader: thread 0 adds to cells in LDS with no-return ds_add / ds_sub (a read-modify-write of a global cell made every probe
// a ~500-cycle stall of wave 0: the phases with many probes looked twice as long as they are); copied out at the end of the solve
__device__ long long g_prof[24];
#define BA_TICK(i) { if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_fetch_sub(&g_ba_lds.prof[i], (unsigned long long)clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
#define BA_TOCK(i) { if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_fetch_add(&g_ba_lds.prof[i], (unsigned long long)clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
#else
#define BA_TICK(i)
#define BA_TOCK(i)
#endif

struct BaCtx {
    int n_poses, use_prior, ex_constant, use_mono, F, P, ex_off;
    int f0, o0;
    int pp0, n_pairs;      // first pair / number of pairs of this window
    int ps0, n_slots;      // first slot / number of slots of the pair-ordered observation list
    int sg0, n_seg, n_multi;   // first segment / number of segments of this window; pairs of more than one segment
    int K, rank, GW;           // workgroups of this window, this workgroup among them (0 = the leader), waves that share the window's segments
    int win;                   // the window
};
__device__ __forceinline__ int ba_pose_off(const BaCtx &c, int i) { return (c.ex_off < 0 ? 0 : 6) + 6 * i; }

// row vector times 3x3 (row-major)
__device__ __forceinline__ void rowmul(const double *u, const double *M, double *o)
{
    o[0] = u[0] * M[0] + u[1] * M[3] + u[2] * M[6];
    o[1] = u[0] * M[1] + u[1] * M[4] + u[2] * M[7];
    o[2] = u[0] * M[2] + u[1] * M[5] + u[2] * M[8];
}
__device__ __forceinline__ void cross3(const double *a, const double *b, double *o)
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

// everything of MonoProjectionFactor::Evaluate that depends only on (pose_i, pose_j, extrinsic).  The reference forms the RESIDUAL with
// the parameter quaternions as they are (Eigen's q * v and q.inverse() * v, MonoProjectionFactor.cc:64-69: for |q| != 1 the map
// v -> v + 2 w (u x v) + 2 u x (u x v) is linear but not a rotation) and the JACOBIANS with normalised rotation matrices (:58-60).  The
// window's quaternions come out of matrix2Double un-normalised in the 9th digit, so the two are kept apart:
//   Tres, tres   p_cj = Tres (depth p_i) + tres, from M(q) = q_to_R of the raw quaternion and M(q^-1)
//   Tn, tn, Cm, A, Bm   the normalised products of the Jacobian blocks
__device__ __forceinline__ void ba_pair_record(const double *Ri, const double *Rj, const double *Rlc, const double *Mi, const double *Mjinv,
                                               const double *Mx, const double *Mxinv, const double *ti, const double *tj, const double *tx, double *rec)
{
    double RlcT[9], RjT[9], G[9], A[9], Bm[9], T[9], Cm[9], v[3], v2[3];
    ba::mT(Rlc, RlcT); ba::mT(Rj, RjT);
    // (Rj^T Ri).normalized() is a Frobenius normalisation (MonoProjectionFactor.cc:125)
    ba::mm(RjT, Ri, G);
    double fro = 0;
    for (int k = 0; k < 9; k++) fro += G[k] * G[k];
    fro = sqrt(fro);
    for (int k = 0; k < 9; k++) G[k] = G[k] / fro - ((k % 4 == 0) ? 1.0 : 0.0);
    ba::mm(RlcT, G, Cm);
    ba::mm(RlcT, RjT, A);
    ba::mm(A, Ri, Bm);
    ba::mm(Bm, Rlc, T);
    ba::mv(Ri, tx, v);
    for (int k = 0; k < 3; k++) v[k] = v[k] + ti[k] - tj[k];
    ba::mv(RjT, v, v2);
    for (int k = 0; k < 3; k++) v2[k] -= tx[k];
    ba::mv(RlcT, v2, v);
    for (int k = 0; k < 9; k++) { rec[39 + k] = T[k]; rec[12 + k] = Cm[k]; rec[21 + k] = A[k]; rec[30 + k] = Bm[k]; }
    for (int k = 0; k < 3; k++) rec[48 + k] = v[k];
    // residual path: p_cj = Mxinv (Mjinv (Mi (Mx pc + tx) + ti - tj) - tx)
    double P1[9], P2[9];
    ba::mm(Mxinv, Mjinv, P1); ba::mm(P1, Mi, P2); ba::mm(P2, Mx, T);
    ba::mv(Mi, tx, v);
    for (int k = 0; k < 3; k++) v[k] = v[k] + ti[k] - tj[k];
    ba::mv(Mjinv, v, v2);
    for (int k = 0; k < 3; k++) v2[k] -= tx[k];
    ba::mv(Mxinv, v2, v);
    for (int k = 0; k < 9; k++) rec[k] = T[k];
    for (int k = 0; k < 3; k++) rec[9 + k] = v[k];
    rec[51] = 0.0;
}

// inverse of a 3x3 matrix (row-major) by the adjugate
__device__ __forceinline__ void inv3(const double *m, double *o)
{
    const double c0 = m[4] * m[8] - m[5] * m[7], c1 = m[5] * m[6] - m[3] * m[8], c2 = m[3] * m[7] - m[4] * m[6];
    const double id = 1.0 / (m[0] * c0 + m[1] * c1 + m[2] * c2);
    o[0] = c0 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = c1 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = c2 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

// v + the value of the neighbouring lane (lane ^ 1): DPP quad_perm [1, 0, 3, 2], no LDS round trip
__device__ __forceinline__ double pair_sum(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0xB1, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0xB1, 0xF, 0xF, false);
    return v + __hiloint2double(hi, lo);
}
// the scratch / index arrays of a window live in HBM: say so (a pointer loaded from the argument struct is generic,
// and generic loads / atomics tie the LDS and the vector-memory counters together)
typedef __attribute__((address_space(1))) double ba_gd;
typedef __attribute__((address_space(1))) const double ba_gcd;
typedef __attribute__((address_space(1))) const int ba_gci;
#ifdef LMONO_BOUNDS
// Bounds-checked build (tests/test_bounds_gpu.py): every global access of k_ba_solve is checked against the batch's ONE device allocation
// (BaBatch::blob_lo / blob_hi: every array of a batch -- inputs, results, scratch -- lives in it).  An access outside it -- what the GPU reports as a
// "memory access fault" without saying where -- is counted, the first one recorded as (source line, kernel phase tag, byte offset from the blob's start) in
// g_ba_oob[], and redirected to the blob's first 8 bytes, so the run survives and says which line it was.  LDS accesses need no check: out-of-range LDS
// reads return zero and writes are dropped (no fault); they show as wrong results, which the parity tests cover.
__device__ unsigned long long g_ba_oob[4];      // [0] hits, [1] line of the first, [2] its byte offset (two's complement), [3] block index
__shared__ const char *g_ba_lo_hi[2];
template <typename T> __device__ __forceinline__ T *ba_chk(T *p, int line)
{
    const char *q = (const char *)p;
    if (q >= g_ba_lo_hi[0] && q + sizeof(T) <= g_ba_lo_hi[1]) return p;
    if (atomicAdd(&g_ba_oob[0], 1ull) == 0ull) { g_ba_oob[1] = (unsigned long long)line; g_ba_oob[2] = (unsigned long long)(q - g_ba_lo_hi[0]); g_ba_oob[3] = blockIdx.x; }
    return (T *)g_ba_lo_hi[0];
}
#define BA_P(p) ba_chk((p), __LINE__)
#define gld(p) (*(ba_gcd *)BA_P((const double *)(p)))
#define gldi(p) (*(ba_gci *)BA_P((const int *)(p)))
#define gst(p, v) (*(ba_gd *)BA_P((double *)(p)) = (v))
#else
#define BA_P(p) (p)
__device__ __forceinline__ double gld(const double *p) { return *(ba_gcd *)p; }
__device__ __forceinline__ int gldi(const int *p) { return *(ba_gci *)p; }
__device__ __forceinline__ void gst(double *p, double v) { *(ba_gd *)p = v; }
#endif

// this wave among the GW waves that take segments (-1: none).  One workgroup: its eight waves.  Several: the leader's wave 1 evaluates the LASERFactor
// chain and the prior meanwhile and takes none.
__device__ __forceinline__ int ba_worker(const BaCtx &c, int wave)
{
    if (c.K == 1) return wave;
    // (the low ids get one segment more when the count does not divide: they go to the followers; the leader has the barrier, the sums and the
    // LASERFactor chain on its hands)
    if (c.rank == 0) return wave == 1 ? -1 : c.GW - (kBaW - 1) + (wave == 0 ? 0 : wave - 1);
    return (c.rank - 1) * kBaW + wave;
}
// data that crosses workgroups (segment tiles, observation records, cost cells, the mail box): device-coherent stores and loads (sc1: they bypass the
// CU's L1 and are ordered by the arrival counters; MI355X_MICROARCH.md, hand-off table) when several workgroups share a window, plain ones otherwise
template <bool kCl> __device__ __forceinline__ double ld_sh(const double *p)
{
    p = BA_P(p);
    if (kCl) return __hip_atomic_load((ba_gcd *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return gld(p);
}
template <bool kCl> __device__ __forceinline__ void st_sh(double *p, double v)
{
    p = BA_P(p);
    if (kCl) __hip_atomic_store((ba_gd *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else gst(p, v);
}
// Per-feature vectors.  Up to kBaLdsFeat = 448 features they live in LDS (BaLds); a larger window (kBig: up to LMONO_BA_MAX_FEATURES = 1024, the
// reference sizes para_depth_inv[10000]) keeps them in an L2 scratch of its own -- same arithmetic in the same order, every access a memory round trip
// instead of an LDS read.  BA_F(name, f): feature f of H_ff / g_f; BA_V(name, k): entry k of a dogleg vector (k < P: pose part, always LDS).
struct BaBig {
    double *Hdd, *gdd, *scale, *D, *gs, *gn, *va, *vb;      // feature parts, [kBaMaxFeat] each
    const double *vinv;                                     // inverse depths of the state being evaluated (the leader's array / the mail box)
    const int *fobs;                                        // feat_obs_off of the window (global observation indices)
    const int *fanchor;
    const unsigned short *seg;
    int o0;
};
template <bool kBig> __device__ __forceinline__ double &ba_fref(double *lds, double *glob, int f) { return kBig ? glob[f] : lds[f]; }
template <bool kBig> __device__ __forceinline__ double &ba_vref(double *lds, double *glob, int k, int P) { return (kBig && k >= P) ? glob[k - P] : lds[k]; }
#define BA_F(name, f) ba_fref<kBig>(L.name, G.name, (f))
#define BA_V(name, k) ba_vref<kBig>(L.name, G.name, (k), c.P)
// inverse depth of feature f of the state being evaluated; first observation of feature f (window-relative); anchor frame; segment word
// (kBig followers read them straight from the leader's mail box, which the leader rewrites with sc1 stores at every publish inside the same launch: the
// read must bypass this CU's L1 on ANY placement -- kFol -- or a line cached during the previous linearisation answers with the old state)
#define BA_VINV(f) (kBig ? ld_sh<kCl || kFol>(G.vinv + (f)) : L.vinv[(f)])
#define BA_FOBS(f) (kBig ? *BA_P(G.fobs + (f)) - G.o0 : (int)L.fobs[(f)])
#define BA_FANCHOR(f) (kBig ? *BA_P(G.fanchor + (f)) : (int)L.fanchor[(f)])
#define BA_SEG(sg) (kBig ? (unsigned int)*BA_P(G.seg + (sg)) : (unsigned int)L.seg[(sg)])

// Hand-offs between the workgroups of a window are FLAG WORDS with one writer each (device-coherent stores, polled with device-coherent loads; no
// read-modify-write: an agent-scope atomic add is resolved beyond the XCD's L2 and costs a microsecond before anybody can see it): the leader's "go"
// word holds the number of evaluations it has published, follower r's "done" word the number it has answered.  B.bar: [W][16] words, [0] = go,
// [r] = done of follower r (rank r >= 1).  Zeroed before every launch.
__device__ __forceinline__ unsigned int ba_flag_load(const unsigned int *p) { return __hip_atomic_load(BA_P(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ba_flag_store(unsigned int *p, unsigned int v) { __hip_atomic_store(BA_P(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// the XCD this workgroup runs on (HW_REG_XCC_ID, bits 3:0).  Used for SPEED only: a follower on the leader's XCD shares its L2, so its plain
// (write-through) stores are where the leader's device-coherent loads look first; a follower elsewhere stores device-coherently (sc1), which is right
// on any placement and slower to read back
__device__ __forceinline__ int ba_xcc_id() { return (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u); }
// lanes 0 .. n - 1 of ONE wave wait until flags[lane] >= target (bounded: a wave that never finishes can take the GPU down); false = gave up
__device__ __forceinline__ bool ba_wait_flags(const unsigned int *flags, int n, unsigned int target, int *fail)
{
    const int lane = threadIdx.x & 63;
    int spins = 0;
    for (;;) {
        const bool there = lane >= n || ba_flag_load(flags + (lane < n ? lane : 0)) >= target;
        if (__all(there)) return true;
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 1023) == 0 && (spins > (1 << 19) || __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
            __hip_atomic_store(fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
    }
}

// LASERFactor chain and the extrinsic prior: a handful of residual blocks, one thread each.  The thread leaves its
// residuals and Jacobians in LDS ([r(6) | J(84)] per block, block 10 = prior); the J^T J products are then spread over
// one wave (ba_small_accumulate_wave).  Kept out of line: its register-resident 6x14 Jacobian must not shape the
// register allocation of the hot loops.
constexpr int kBaSmallRec = 96;
template <bool kJac>
__device__ __noinline__ double ba_small_factors(const BaBatch &B, const BaCtx c, double *out, const double *poses, const double *ex, int t)
{
    double cost = 0.0;
    const double *laser_info = B.info, *prior_w = B.info + 40;
    if (t >= 0 && t < c.n_poses - 1) {
        double prm[14], cst[24], inf[36], r[6], J[84];
#pragma unroll
        for (int k = 0; k < 7; k++) { prm[k] = poses[7 * t + k]; prm[7 + k] = poses[7 * (t + 1) + k]; }
#pragma unroll
        for (int k = 0; k < 24; k++) cst[k] = gld(B.laser_consts + ((size_t)c.win * 10 + t) * 24 + k);      // (c.win, not blockIdx.x: with several workgroups per window a block is not a window)
#pragma unroll
        for (int k = 0; k < 36; k++) inf[k] = gld(laser_info + k);
        ba::laser_factor(prm, cst, inf, r, kJac ? J : nullptr);
#pragma unroll
        for (int k = 0; k < 6; k++) cost += 0.5 * r[k] * r[k];
        if (kJac) {
            double *o = out + t * kBaSmallRec;
#pragma unroll
            for (int k = 0; k < 6; k++) o[k] = r[k];
#pragma unroll
            for (int k = 0; k < 84; k++) o[6 + k] = J[k];
        }
    }
    if (t == 16 && c.use_prior && !c.ex_constant) {
        double prm[7], cst[16], r[6], J[42];
        const double pw[2] = { gld(prior_w), gld(prior_w + 1) };
#pragma unroll
        for (int k = 0; k < 7; k++) prm[k] = ex[k];
#pragma unroll
        for (int k = 0; k < 16; k++) cst[k] = gld(B.prior_T + (size_t)c.win * 16 + k);
        ba::prior_factor(prm, cst, pw, r, kJac ? J : nullptr);
#pragma unroll
        for (int k = 0; k < 6; k++) cost += 0.5 * r[k] * r[k];
        if (kJac) {
            double *o = out + 10 * kBaSmallRec;
#pragma unroll
            for (int k = 0; k < 6; k++) o[k] = r[k];
#pragma unroll
            for (int k = 0; k < 42; k++) o[6 + k] = J[k];
        }
    }
    return cost;
}

// J^T J and J^T r of the blocks left in LDS by ba_small_factors, by one wave (one output entry per lane and round).  Neighbouring
// LASER blocks share a pose: the even blocks are added first, then the odd ones (within a phase every entry has one writer), so the
// sums are formed in a fixed order with plain LDS read-modify-writes.  Nothing else touches H_pp meanwhile (L.turn == -1).
// Round 4: the LASER blocks are spread over the WHOLE workgroup (a barrier between the even and the odd blocks) -- one wave walking all 1560 entries
// was the critical path of the phase (the other seven waves were done with the per-feature sums in half its time); same sums in the same order.
__device__ __forceinline__ void ba_small_accumulate_block(const BaCtx &c, BaLds &L, const double *in, int tid)
{
    const int nl = c.n_poses - 1;
    for (int par = 0; par < 2; par++) {
        const int nb = (nl - par + 1) / 2;                       // blocks par, par + 2, ...
        for (int idx = tid; idx < nb * 156; idx += kBaT) {
            const int b = 2 * (idx / 156) + par, e = idx % 156;
            const double *r = in + b * kBaSmallRec, *J = r + 6;
            const int a = e < 144 ? e / 12 : e - 144, bb = e < 144 ? e % 12 : 0;
            const double *Ja = J + (a < 6 ? a : 42 + a - 6);
            const int ga = ba_pose_off(c, a < 6 ? b : b + 1) + (a < 6 ? a : a - 6);
            double v = 0;
            if (e < 144) {
                const double *Jb = J + (bb < 6 ? bb : 42 + bb - 6);
#pragma unroll
                for (int k = 0; k < 6; k++) v += Ja[7 * k] * Jb[7 * k];
                L.Hpp[ga * kBaP + ba_pose_off(c, bb < 6 ? b : b + 1) + (bb < 6 ? bb : bb - 6)] += v;
            } else {
#pragma unroll
                for (int k = 0; k < 6; k++) v += Ja[7 * k] * r[k];
                L.gp[ga] += v;
            }
        }
        __syncthreads();
    }
}
// the extrinsic prior's block (entries nothing else has touched yet), by one wave
__device__ __forceinline__ void ba_prior_accumulate_wave(const BaCtx &c, BaLds &L, const double *in, int lane)
{
    if (c.use_prior && !c.ex_constant && lane < 42) {
        const double *r = in + 10 * kBaSmallRec, *J = r + 6;
        const int e = lane, a = e < 36 ? e / 6 : e - 36, bb = e < 36 ? e % 6 : 0;
        double v = 0;
        if (e < 36) {
#pragma unroll
            for (int k = 0; k < 6; k++) v += J[7 * k + a] * J[7 * k + bb];
            L.Hpp[(c.ex_off + a) * kBaP + c.ex_off + bb] += v;
        } else {
#pragma unroll
            for (int k = 0; k < 6; k++) v += J[7 * k + a] * r[k];
            L.gp[c.ex_off + a] += v;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();    // the slice becomes the wave's Jacobian stage
}

// H_pp and g_p as ordered sums of the segments' tiles (ba_evaluate leaves one record per segment in the L2 scratch): every entry is owned
// by one thread, which adds the tiles that touch it in a fixed order -- a pair's segments in segment order, then the pairs (f, j) by ascending j and
// the pairs (i, f) by ascending i for an entry of frame f's rows / columns; all segments by index for the extrinsic block -- so the sums are the same
// bits in every run whatever the waves' timing was and whoever computed a segment.  Entries nothing touches become zero: the pass replaces clearing H_pp.
// Round 6: WHO forms an entry does not change it either, so the entries CAN be dealt to the workgroups of a window's cluster (part of nparts: the gather
// program's items in turn, the extrinsic block to the last part; VERDICT r5 #2a) instead of the leader forming all of them while the followers wait.
// Out: where an entry goes -- the leader's LDS when one workgroup forms them all, the window's row of B.hred (72 x 72 + 72 doubles, L2) when the cluster
// shares them; the leader then loads the row in one sweep.  Built, byte-identical (the K = 1 / 2 / 4 / 8 tests ran on it), and MEASURED SLOWER: single
// window 3.44 -> 3.51 ms, Estimator loop 393 -> 372 frames/s.  The pass is a chain of three dependent L2 round trips (program words -> tile cells -> the
// other segments of a pair) whatever the share is, so a smaller share does not end sooner, and sharing adds two flag hand-offs and the leader's read-back
// of 5.3 k doubles per linearisation.  Kept behind LMONO_BA_SHARE_SUMS=1 (default off: the leader forms every entry).
struct BaOutLds {
    BaLds &L;
    __device__ __forceinline__ void put(int dst, double v) const { if (dst >= kBaP * kBaP) L.gp[dst - kBaP * kBaP] = v; else L.Hpp[dst] = v; }
};
template <bool kSc> struct BaOutGlob {
    double *h;
    __device__ __forceinline__ void put(int dst, double v) const { st_sh<kSc>(h + dst, v); }
};
constexpr int kBaHred = kBaP * kBaP + kBaP;
template <bool kCl, class Out>
__device__ __forceinline__ void ba_reduce_pairs(const BaBatch &B, const BaCtx c, BaLds &L, const Out out, int part, int nparts)
{
    const int tid = threadIdx.x;
    // the window's tiles: one per segment
    double *tiles = B.pairH + (size_t)(c.sg0 + c.pp0 + c.win) * kBaPairTile;
    // (A), (B): the entries of H_pp and g_p that are sums over frame pairs, by the window's gather program (ba_setup: the cells an entry is the sum of are
    // the same in every iteration; looking them up anew -- pair tables in LDS, 64-bit address arithmetic, a branch per source -- was 400 instructions
    // per entry).  A source is a frame PAIR: (position in its first segment's tile) | (segments << 20); the pair's value is the sum of its segments' tiles in
    // segment order, formed here, inline (round 6: a separate pass used to pre-sum the pairs of several segments into tiles of their own -- one more trip
    // through the L2 and a workgroup barrier per linearisation; same sums in the same order); an absent source (0 segments) counts as zero.
    // Part A: the two pairs that can hold a block of two different frames; part B: the pairs (f, k) by ascending k, then the pairs (k, f), for the entries
    // of ONE frame f.
    const int x0 = c.ex_off;                                   // first extrinsic index (-1: extrinsic constant)
    const int *gp = B.gprog + (size_t)c.win * kBaGprog;
    auto pair_value = [&](int src, double first) {             // first = the entry of the pair's first segment (requested with everybody's); the rest in order
        const int cnt = src >> 20;
        const double *t0 = tiles + (src & 0xfffff);
        for (int sgi = 1; sgi < cnt; sgi++) first += ld_sh<kCl>(t0 + (size_t)sgi * kBaPairTile);
        return first;
    };
    for (int a0 = part * kBaT * 4; a0 < kBaGaN; a0 += nparts * kBaT * 4) {
        int o1[4], o2[4], d1[4], d2[4];
        double v1[4], v2[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int a = a0 + tid + u * kBaT;
            const bool in = a < kBaGaN;
            o1[u] = in ? gldi(gp + a) : 0; o2[u] = in ? gldi(gp + kBaGaN + a) : 0; d1[u] = in ? gldi(gp + 2 * kBaGaN + a) : -1; d2[u] = in ? gldi(gp + 3 * kBaGaN + a) : -1;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) { v1[u] = (o1[u] >> 20) ? ld_sh<kCl>(tiles + (o1[u] & 0xfffff)) : 0.0; v2[u] = (o2[u] >> 20) ? ld_sh<kCl>(tiles + (o2[u] & 0xfffff)) : 0.0; }
#pragma unroll
        for (int u = 0; u < 4; u++) if (d1[u] >= 0) { const double sum = pair_value(o1[u], v1[u]) + pair_value(o2[u], v2[u]); out.put(d1[u], sum); out.put(d2[u], sum); }
    }
    const int *gb = gp + 4 * kBaGaN;
    for (int t = part * kBaT + tid; t < kBaGbN; t += nparts * kBaT) {
        int off[22];
        double v[22];
#pragma unroll
        for (int k = 0; k < 22; k++) off[k] = gldi(gb + k * kBaGbN + t);
        const int dst = gldi(gb + 22 * kBaGbN + t), dst2 = gldi(gb + 23 * kBaGbN + t);
#pragma unroll
        for (int k = 0; k < 22; k++) v[k] = (off[k] >> 20) ? ld_sh<kCl>(tiles + (off[k] & 0xfffff)) : 0.0;
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < 22; k++) acc += pair_value(off[k], v[k]);
        if (dst >= 0) { out.put(dst, acc); if (dst2 >= 0) out.put(dst2, acc); }
    }
    // (C) the extrinsic block and gradient get a share from EVERY segment: 33 values per segment (rows 12..15 of the tile and of the side tile, and
    //     the five scalar sums of the (4..5, 4..5) corner); 15 threads per value add the segments s = g (mod 15) in ascending order into L.D (dead during
    //     a linearisation; every workgroup of a cluster has its own), one thread per value then adds the 15 in order.  The last part's.
    if (x0 >= 0 && part == nparts - 1) {
        constexpr int kG = 15;
        if (tid < 33 * kG) {
            const int e = tid % 33, g = tid / 33;
            int off;
            if (e < 16) off = (12 + e / 4) * 16 + 12 + e % 4;
            else if (e < 28) off = 256 + (12 + (e - 16) / 3) * 4 + (e - 16) % 3;
            else off = 256 + (e - 27) * 4 + 3;
            double acc = 0.0;
            for (int sg = g; sg < c.n_seg; sg += 4 * kG) {
                double v[4];
#pragma unroll
                for (int d = 0; d < 4; d++) v[d] = sg + d * kG < c.n_seg ? ld_sh<kCl>(tiles + (size_t)(sg + d * kG) * kBaPairTile + off) : 0.0;
#pragma unroll
                for (int d = 0; d < 4; d++) acc += v[d];
            }
            L.D[g * 33 + e] = acc;
        }
        __syncthreads();
        if (tid < 33) {
            double tot = 0.0;
#pragma unroll
            for (int g = 0; g < kG; g++) tot += L.D[g * 33 + tid];
            const int e = tid, x4 = x0 + 4, x5 = x0 + 5, G0 = kBaP * kBaP;
            if (e < 16) out.put((x0 + e / 4) * kBaP + x0 + e % 4, tot);
            else if (e < 28) {
                const int om = (e - 16) / 3, cc = (e - 16) % 3;
                if (cc < 2) { out.put((x0 + om) * kBaP + x4 + cc, tot); out.put((x4 + cc) * kBaP + x0 + om, tot); }
                else out.put(G0 + x0 + om, tot);
            }
            else if (e == 28) out.put(x4 * kBaP + x4, tot);
            else if (e == 29) { out.put(x4 * kBaP + x5, tot); out.put(x5 * kBaP + x4, tot); }
            else if (e == 30) out.put(x5 * kBaP + x5, tot);
            else if (e == 31) out.put(G0 + x4, tot);
            else out.put(G0 + x5, tot);
        }
    }
    __syncthreads();
}

// the segments' shares of a cost (one double per segment, `stride` apart), added in segment order by wave 0; the result is left in L.red[3 * kBaW - 1]
// for everybody (read it behind the next workgroup barrier)
template <bool kCl>
__device__ __forceinline__ void ba_segment_cost(const BaCtx &c, BaLds &L, const double *part, int stride)
{
    const int tid = threadIdx.x;
    if (tid < 64) {
        double acc = 0.0;
        for (int sg = tid; sg < c.n_seg; sg += 4 * 64) {
            double v[4];
#pragma unroll
            for (int d = 0; d < 4; d++) v[d] = sg + 64 * d < c.n_seg ? ld_sh<kCl>(part + (size_t)(sg + 64 * d) * stride) : 0.0;
#pragma unroll
            for (int d = 0; d < 4; d++) acc += v[d];
        }
        acc = wave_sum_d(acc);
        if (tid == 0) L.red[3 * kBaW - 1] = acc;
    }
}

// sum over the 32 lanes of this lane's half of the wave (lanes 0..31 / 32..63), in every lane of the half
__device__ __forceinline__ double half_sum_d(double v)
{
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// sum over the 16 lanes of this lane's quarter of the wave
__device__ __forceinline__ double quarter_sum_d(double v)
{
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// pose matrices and inverse depths of the state (poses, ex, invd) in LDS (ba_state), then its pair records in HBM (ba_pair_records) -- every workgroup
// of a window computes its own (the followers read the inverse depths from the leader's mail box)
template <bool kSharedInvd, bool kBig>      // kSharedInvd: a follower (the inverse depths are in LDS already); kBig: they are read where they lie
__device__ __forceinline__ void ba_state(const BaCtx &c, BaLds &L, const double *poses, const double *ex, const double *invd)
{
    const int tid = threadIdx.x;
    if (tid <= c.n_poses) {
        double qn[4];
        const double *qraw = tid == c.n_poses ? ex + 3 : poses + 7 * tid + 3;
        ba::q_norm(qraw, qn);
        ba::q_to_R(qn, L.Rp + 9 * tid);
        double qi[4];
        ba::q_to_R(qraw, L.Mq + 18 * tid);
        ba::q_inv(qraw, qi);
        ba::q_to_R(qi, L.Mq + 18 * tid + 9);
        if (tid == c.n_poses) inv3(L.Mq + 18 * tid + 9, L.Mq + 18 * (kBaMaxPoses + 1));
    }
    if (!kSharedInvd && !kBig) for (int f = tid; f < c.F; f += kBaT) L.vinv[f] = gld(invd + f);       // (a follower has read them from the mail box)
    __syncthreads();
}
__device__ __forceinline__ void ba_pair_records(const BaCtx &c, BaLds &L, const double *poses, const double *ex, double *pairdat)
{
    const int tid = threadIdx.x;
    const double *Rlc = L.Rp + 9 * c.n_poses;
    for (int p = tid; p < c.n_pairs; p += kBaT) {
        const int ij = L.pair_ij[p], i = ij & 255, j = ij >> 8;
        double rec[kBaPairRec];
        ba_pair_record(L.Rp + 9 * i, L.Rp + 9 * j, Rlc, L.Mq + 18 * i, L.Mq + 18 * j + 9, L.Mq + 18 * c.n_poses, L.Mq + 18 * c.n_poses + 9,
                       poses + 7 * i, poses + 7 * j, ex, rec);
        double *dst = pairdat + (size_t)p * kBaPairRec;
#pragma unroll
        for (int k = 0; k < kBaPairRec; k++) gst(dst + k, rec[k]);
    }
}

// The leader hands the state to evaluate to the other workgroups of its window: mail box, then its go word.  cmd 1 = linearise (the followers take their
// share of the segments and report), 2 = a candidate is being evaluated (the leader does that alone -- the shares are small and a hand-off costs more
// than they save -- but the followers compute the candidate's records meanwhile: if the step is accepted the linearisation behind it re-uses them),
// 3 = leave.  A follower acknowledges a mail box as soon as it has read it; the leader waits for those acknowledgements before it writes the next.
__device__ __forceinline__ void ba_publish(const BaBatch &B, const BaCtx &c, BaLds &L, int cmd, const double *poses, const double *ex, const double *invd, bool reuse)
{
    const int tid = threadIdx.x;
    unsigned int *flags = B.bar + (size_t)c.win * kBaBar;
    if (tid < 64 && L.eval_no > 0 && !L.failed) { if (!ba_wait_flags(flags + 1, c.K - 1, (unsigned int)L.eval_no, B.fail) && tid == 0) L.failed = 1; }
    __syncthreads();
    double *mb = B.mbox + (size_t)c.win * kBaMbox;
    if (tid == 0) { st_sh<true>(mb, (double)cmd); st_sh<true>(mb + 1, reuse ? 1.0 : 0.0); st_sh<true>(mb + 2, (double)ba_xcc_id()); st_sh<true>(mb + 3, (cmd == 1 && L.coloc == 1) ? 1.0 : 0.0); }
    if (!reuse && cmd != 3) {
        for (int k = tid; k < 7 * c.n_poses; k += kBaT) st_sh<true>(mb + 8 + k, poses[k]);
        if (tid < 7) st_sh<true>(mb + 88 + tid, ex[tid]);
        for (int f = tid; f < c.F; f += kBaT) st_sh<true>(mb + 96 + f, gld(invd + f));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) ba_flag_store(flags, (unsigned int)(L.eval_no + 1));
}
// everybody's stores of a linearisation are out: the leader waits for the followers' done words, a follower sets its own
template <bool kLeader>
__device__ __forceinline__ void ba_done(const BaBatch &B, const BaCtx &c, BaLds &L)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned int *flags = B.bar + (size_t)c.win * kBaBar;
    if (kLeader) {
        if (threadIdx.x < 64 && !L.failed) { if (!ba_wait_flags(flags + 1, c.K - 1, (unsigned int)(L.eval_no + 1), B.fail) && threadIdx.x == 0) L.failed = 1; }
        __syncthreads();
    } else if (threadIdx.x == 0) ba_flag_store(flags + c.rank, (unsigned int)(L.eval_no + 1));
}

// one workgroup's share of an evaluation: the segments gw, gw + GW, ... of the window (a linearisation takes two per round and wave, a cost evaluation
// four).  Every result goes to the segment's / the observation's own record.
// kLds (one workgroup per window, linearisation): ONE round -- the segments 16 round + wave and 16 round + 8 + wave -- and the two tiles stay in the wave's own
// slice of the LDS stage (ba_linearise_lds adds them into H_pp from there) instead of going to the L2 scratch
template <bool kJac, bool kCl, bool kBig, bool kFol = false, bool kLds = false>      // kFol: a follower workgroup (its inverse depths are the leader's mail box)
__device__ __forceinline__ void ba_segments(const BaBatch &B, const BaCtx &c, BaLds &L, const BaBig &G, const double *ex, const double *pairdat, int round = 0,
                                            const double *minfo = nullptr)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double *Rlc = L.Rp + 9 * c.n_poses;
    const double *mono_info = B.info + 36;
    // (sqrt_info of the projection factor: from the caller's registers when it calls once per round -- four L2 round trips per call otherwise)
    const double m00 = minfo ? minfo[0] : gld(mono_info), m01 = minfo ? minfo[1] : gld(mono_info + 1), m10 = minfo ? minfo[2] : gld(mono_info + 2), m11 = minfo ? minfo[3] : gld(mono_info + 3);
    const int *sinfo = B.slot_info + c.ps0;
    const double *spts = B.slot_pts + (size_t)c.ps0 * 4;
    const int gw = ba_worker(c, wave), GW = c.GW;        // this wave among the waves that share the window's segments
    if (gw < 0) return;
    if (!kJac) {
        // one lane per slot, 16 lanes per segment, four segments per wave and round: a segment's share of the cost is the sum over its 16 lanes and
        // goes to the segment's own cell; the cells are added in segment order by the leader
        double *cpart = B.cpart + c.sg0;
        for (int s4 = 4 * gw; s4 < c.n_seg; s4 += 4 * GW) {
            const int sg = s4 + (lane >> 4);
            const bool seg_ok = sg < c.n_seg;
            const unsigned int sv = BA_SEG(seg_ok ? sg : s4);
            const int pr = sv & 127;
            const int so = L.pair_slot[pr] + kBaSeg * (int)(sv >> 7) + (lane & 15);
            double cst = 0.0;
            if (seg_ok && so < L.pair_slot[pr + 1]) {
                const int f = gldi(sinfo + so) & 0xffff;
                const double pax = gld(spts + (size_t)so * 4), pay = gld(spts + (size_t)so * 4 + 1);
                const double pbx = gld(spts + (size_t)so * 4 + 2), pby = gld(spts + (size_t)so * 4 + 3);
                const double *rec = pairdat + (size_t)pr * kBaPairRec;
                double Tt[12], pcj[3];
#pragma unroll
                for (int k = 0; k < 12; k++) Tt[k] = gld(rec + k);
                const double depth = 1.0 / BA_VINV(f);
                const double pc[3] = { depth * pax, depth * pay, depth };
                ba::mv(Tt, pc, pcj);
                for (int k = 0; k < 3; k++) pcj[k] += Tt[9 + k];
                const double inv = 1.0 / pcj[2];
                const double e0 = pcj[0] * inv - pbx, e1 = pcj[1] * inv - pby;
                const double r0 = m00 * e0 + m01 * e1, r1 = m10 * e0 + m11 * e1;
                cst = 0.5 * log(1.0 + (r0 * r0 + r1 * r1));   // ceres::CauchyLoss(1)
            }
            cst = quarter_sum_d(cst);
            if (seg_ok && (lane & 15) == 0) st_sh<kCl>(cpart + sg, cst);
        }
        return;
    }
    // Every wave takes SEGMENTS (up to 16 slots of one frame pair), two per round: lanes 0..31 the first, lanes 32..63 the second, two lanes per
    // observation (lane q of the pair owns residual row q); the 64 rows are staged in the wave's own LDS slice and multiplied right away (MFMA steps
    // 0..7 -> the first segment's tile, 8..15 -> the second's) -- no workgroup barrier inside the loop.  A segment's tile, its share of the cost and
    // of the extrinsic (4..5, 4..5) corner go to the segment's own record: nothing is carried from one round to the next, so which wave (of which
    // workgroup) takes a segment does not change a bit of the result.
    double *wstage = L.u.stage + (size_t)wave * 2 * kBaRound * kBaRow;
    double *tiles = B.pairH + (size_t)(c.sg0 + c.pp0 + c.win) * kBaPairTile;
    const int q = lane & 1, half = lane >> 5, lo = (lane & 31) >> 1, col = lane & 15, kq = lane >> 4;
    const double *Mx = L.Mq + 18 * c.n_poses, *Mxi = L.Mq + 18 * (kBaMaxPoses + 1);
    const int s_lo = kLds ? 2 * kBaW * round + gw : gw, s_hi = kLds ? min(c.n_seg, 2 * kBaW * round + kBaW) : c.n_seg;
    for (int sA = s_lo; sA < s_hi; sA += 2 * GW) {
        const int sB = sA + GW;
        const int sg = half ? sB : sA;                       // this lane's segment
        const bool seg_ok = sg < c.n_seg;
        const unsigned int sv = BA_SEG(seg_ok ? sg : sA);
        const int pr = sv & 127;
        const int s_begin = L.pair_slot[pr] + kBaSeg * (int)(sv >> 7), s_end = min(s_begin + kBaSeg, (int)L.pair_slot[pr + 1]);
        const int so = s_begin + lo;
        const int ij = L.pair_ij[pr], fj = ij >> 8;
        const int oj = ba_pose_off(c, fj);
        // the pair record: one request per value (at most two addresses per wave), kept in registers for the round
        double rec[kBaPairRec - 1];
        {
            const double *rp = pairdat + (size_t)pr * kBaPairRec;
#pragma unroll
            for (int k = 0; k < kBaPairRec - 1; k++) rec[k] = gld(rp + k);
        }
        const double *T = rec, *Cm = rec + 12, *A = rec + 21, *Bm = rec + 30, *Tn = rec + 39;
        double cst = 0, xx44 = 0, xx45 = 0, xx55 = 0, gx4 = 0, gx5 = 0;       // this lane's share of its segment's scalar sums
        double *row = wstage + (size_t)lane * kBaRow;
        if (seg_ok && so < s_end) {
            const int f = gldi(sinfo + so) & 0xffff;
            const int ob = gldi(B.slot_obs + c.ps0 + so);
            const double pax = gld(spts + (size_t)so * 4), pay = gld(spts + (size_t)so * 4 + 1);
            const double pbx = gld(spts + (size_t)so * 4 + 2), pby = gld(spts + (size_t)so * 4 + 3);
            const double depth = 1.0 / BA_VINV(f);
            const double pc[3] = { depth * pax, depth * pay, depth };
            double Tp[3], pcj[3], pcn[3], uT[3], u[3];
            ba::mv(T, pc, Tp);
            for (int k = 0; k < 3; k++) pcj[k] = Tp[k] + rec[9 + k];     // the residual's p_cj (raw quaternions)
            ba::mv(Tn, pc, Tp);                                          // from here on Tp = Tn pc: the Jacobians' normalised product
            for (int k = 0; k < 3; k++) pcn[k] = Tp[k] + rec[48 + k];
            const double inv = 1.0 / pcj[2];
            const double e0 = pcj[0] * inv - pbx, e1 = pcj[1] * inv - pby;
            const double r0 = m00 * e0 + m01 * e1, r1 = m10 * e0 + m11 * e1;
            const double sq = r0 * r0 + r1 * r1;
            if (q == 0) cst = 0.5 * log(1.0 + sq);
            // ceres::CauchyLoss(1): rho' = 1 / (1 + s), rho'' < 0 -> the corrector scales rows by sqrt(rho')
            const double rho1 = 1.0 / (1.0 + sq);
            const double sr = sqrt(rho1 > DBL_MIN ? rho1 : DBL_MIN);
            const double rq = (q ? r1 : r0) * sr;
            // row q of sqrt_info * [[1/z, 0, -x/z^2], [0, 1/z, -y/z^2]], robustified
            const double ma = q ? m10 : m00, mb = q ? m11 : m01;
            u[0] = sr * ma * inv; u[1] = sr * mb * inv; u[2] = -sr * (ma * pcj[0] + mb * pcj[1]) * inv * inv;
            rowmul(u, Tn, uT);
            // inverse depth: -u (Tn p_i) depth^2 = -u (Tn pc) depth
            const double Jd = -(u[0] * Tp[0] + u[1] * Tp[1] + u[2] * Tp[2]) * depth;
            double Jx[6], Ji[6], Jj[6], c1[3], c2[3];
            // extrinsic block: position u Cm, rotation -(uT) x pc + u x (Tn pc + tn)   (u skew(v) = u x v; :127-130: all normalised)
            if (c.ex_off >= 0) {
                rowmul(u, Cm, Jx);
                cross3(uT, pc, c1); cross3(u, pcn, c2);
                for (int k = 0; k < 3; k++) Jx[3 + k] = c2[k] - c1[k];
                xx44 = Jx[4] * Jx[4]; xx45 = Jx[4] * Jx[5]; xx55 = Jx[5] * Jx[5];
                gx4 = Jx[4] * rq; gx5 = Jx[5] * rq;
            } else {
#pragma unroll
                for (int k = 0; k < 6; k++) Jx[k] = 0.0;
            }
            // pose i: position u A, rotation -(u B) x pl;  pose j: position -u A, rotation (u Rlc^T) x plj, with the residual
            // path's own points pl = pts_laser_i = Qx pc + tx and plj = pt_l_j = M(qx^-1)^-1 p_cj + tx (:66-68, :145, :158)
            {
                double uB[3], uR[3], pl[3], plj[3];
                rowmul(u, A, Ji);
                for (int k = 0; k < 3; k++) Jj[k] = -Ji[k];
                ba::mv(Mx, pc, pl);
                for (int k = 0; k < 3; k++) pl[k] += ex[k];
                rowmul(u, Bm, uB); cross3(uB, pl, c1);
                for (int k = 0; k < 3; k++) Ji[3 + k] = -c1[k];
                ba::mv(Mxi, pcj, plj);
                for (int k = 0; k < 3; k++) plj[k] += ex[k];
                uR[0] = u[0] * Rlc[0] + u[1] * Rlc[1] + u[2] * Rlc[2];
                uR[1] = u[0] * Rlc[3] + u[1] * Rlc[4] + u[2] * Rlc[5];
                uR[2] = u[0] * Rlc[6] + u[1] * Rlc[7] + u[2] * Rlc[8];
                cross3(uR, plj, c1);
                for (int k = 0; k < 3; k++) Jj[3 + k] = c1[k];
            }
#pragma unroll
            for (int k = 0; k < 6; k++) { row[k] = Ji[k]; row[6 + k] = Jj[k]; row[12 + k] = Jx[k]; }
            row[18] = rq;
            // depth block and coupling row: the two rows of the observation are summed across the lane pair.  The observation's shares of H_ff, g_f and
            // of its feature's coupling row go to the observation's own scratch record (host observation order: a feature's records are contiguous)
            // and are summed / placed per feature afterwards -- no atomics, the same bits every run, and whoever evaluates the observation.
            double hx[6], hi[6], hj[6];
#pragma unroll
            for (int k = 0; k < 6; k++) { hx[k] = pair_sum(Jx[k] * Jd); hi[k] = pair_sum(Ji[k] * Jd); hj[k] = pair_sum(Jj[k] * Jd); }
            const double hdd = pair_sum(Jd * Jd), gd = pair_sum(Jd * rq);
            double *sc = B.obsc + (size_t)ob * kBaObsRec;
            if (q == 0) {
                st_sh<kCl>(sc, hdd); st_sh<kCl>(sc + 1, gd);
#pragma unroll
                for (int k = 0; k < 6; k++) st_sh<kCl>(sc + 2 + k, hx[k]);
                st_sh<kCl>(sc + 20, (double)(oj + 1)); st_sh<kCl>(sc + 21, (double)f);   // where frame j's share goes: column + 1 (0: no record), row
            } else {
#pragma unroll
                for (int k = 0; k < 6; k++) { st_sh<kCl>(sc + 8 + k, hi[k]); st_sh<kCl>(sc + 14 + k, hj[k]); }
            }
        } else {
#pragma unroll
            for (int k = 0; k < kBaRow; k++) row[k] = 0.0;
        }
        // the segment's scalar sums: over the 32 lanes of this half
        cst = half_sum_d(cst);
        if (c.ex_off >= 0) { xx44 = half_sum_d(xx44); xx45 = half_sum_d(xx45); xx55 = half_sum_d(xx55); gx4 = half_sum_d(gx4); gx5 = half_sum_d(gx5); }
        // the wave's rows are in LDS (its LDS operations execute in order): J^T [J r], four rows per MFMA step
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int nA = __builtin_amdgcn_readlane(s_end - s_begin, 0), nB = sB < c.n_seg ? __builtin_amdgcn_readlane(s_end - s_begin, 32) : 0;
        ba_d4 keep_a[2], keep_b[2];              // kLds: both tiles wait in registers until both products have read the staged rows (the tiles take the rows' place)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int n = h ? nB : nA;
            if (n > 0) {
                ba_d4 aa = { 0, 0, 0, 0 }, ab = { 0, 0, 0, 0 };
                for (int ks = 0; 4 * ks < 2 * n; ks++) {
                    const double *rowp = wstage + (size_t)(2 * kBaSeg * h + 4 * ks + kq) * kBaRow;
                    const double a = rowp[col];
                    const double bq = col < 3 ? rowp[16 + col] : 0.0;
                    aa = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, aa, 0, 0, 0);
                    ab = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bq, ab, 0, 0, 0);
                }
                if (kLds) { keep_a[h] = aa; keep_b[h] = ab; continue; }
                // the segment's tile [J_i J_j J_x0..3]^T [J_i J_j J_x0..3 | J_x4 J_x5 r] and its scalar sums go to the segment's own record in the
                // L2 scratch (plain stores, nobody else touches it); ba_reduce_pairs adds the records in a fixed order afterwards
                double *tile = tiles + (size_t)(h ? sB : sA) * kBaPairTile;
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    st_sh<kCl>(tile + (kq + 4 * v) * 16 + col, aa[v]);
                    if (col < 3) st_sh<kCl>(tile + 256 + (kq + 4 * v) * 4 + col, ab[v]);
                }
                if (lane == 32 * h) {
                    st_sh<kCl>(tile + 256 + 3, cst); st_sh<kCl>(tile + 260 + 3, xx44); st_sh<kCl>(tile + 264 + 3, xx45); st_sh<kCl>(tile + 268 + 3, xx55);
                    st_sh<kCl>(tile + 272 + 3, gx4); st_sh<kCl>(tile + 276 + 3, gx5);
                }
            }
        }
        if (kLds) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();                     // every lane's products have read their rows
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int n = h ? nB : nA;
                if (n > 0) {
                    double *tile = wstage + h * kBaPairTile;
#pragma unroll
                    for (int v = 0; v < 4; v++) {
                        tile[(kq + 4 * v) * 16 + col] = keep_a[h][v];
                        if (col < 3) tile[256 + (kq + 4 * v) * 4 + col] = keep_b[h][v];
                    }
                    if (lane == 32 * h) {
                        tile[256 + 3] = cst; tile[260 + 3] = xx44; tile[264 + 3] = xx45; tile[268 + 3] = xx55; tile[272 + 3] = gx4; tile[276 + 3] = gx5;
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();   // the slice is rewritten by the next round
    }
}

// H_ff, g_f and the coupling rows from the observations' records, the features / observations f = part (mod nparts) of them: 16 lanes per feature, lane k
// sums entry k of the feature's observation records (contiguous, at most 10: one per other frame of the window) in observation order; the loads are
// independent and requested together.  kGlob: H_ff / g_f of an LDS-resident window go to the window's row of B.fdg (the cluster shares the pass; the leader
// loads them), else to the leader's LDS (kBig: the L2 scratch either way).
template <bool kCl, bool kBig, bool kGlob>
__device__ __forceinline__ void ba_feature_pass(const BaBatch &B, const BaCtx &c, BaLds &L, const BaBig &G, double *hpd, int part, int nparts)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double *fd = B.fdg + (size_t)c.win * 2 * B.feat_cap;
    {
        const int k = lane & 15, t8 = wave * 4 + (lane >> 4);
        for (int f = part * 4 * kBaW + t8; f < c.F; f += nparts * 4 * kBaW) {
            const int o0 = c.o0 + BA_FOBS(f), o1 = c.o0 + BA_FOBS(f + 1);
            double acc = 0.0;
            for (int ob = o0; ob < o1; ob += 10) {
                double v[10];
#pragma unroll
                for (int u = 0; u < 10; u++) v[u] = ob + u < o1 ? ld_sh<kCl>(B.obsc + (size_t)(ob + u) * kBaObsRec + k) : 0.0;
#pragma unroll
                for (int u = 0; u < 10; u++) acc += v[u];
            }
            if (k == 0) { if (kGlob && !kBig) st_sh<true>(fd + f, acc); else BA_F(Hdd, f) = acc; }
            else if (k == 1) { if (kGlob && !kBig) st_sh<true>(fd + B.feat_cap + f, acc); else BA_F(gdd, f) = acc; }
            else if (k < 14) {
                double *hrow = hpd + (size_t)f * kBaPS;
                if (k < 8) { if (c.ex_off >= 0) gst(hrow + c.ex_off + k - 2, acc); }
                else { const int anchor = BA_FANCHOR(f); if (anchor >= 0) gst(hrow + ba_pose_off(c, anchor) + k - 8, acc); }
            }
        }
        // frame j's share of a coupling row (frame j sees a feature once) moves from the observation's record to its place: 8 lanes per observation, four
        // observations of a lane in flight
        const int n_obs = c.use_mono ? BA_FOBS(c.F) : 0;
        const int kk = tid & 7;
        for (int ob0 = part * 4 * (kBaT / 8) + (tid >> 3); ob0 < n_obs; ob0 += nparts * 4 * (kBaT / 8)) {
            double vj[4], vo[4], vf[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int ob = ob0 + u * (kBaT / 8);
                const double *rc = B.obsc + (size_t)(c.o0 + (ob < n_obs ? ob : ob0)) * kBaObsRec;
                vj[u] = ld_sh<kCl>(rc + 14 + (kk < 6 ? kk : 0)); vo[u] = ld_sh<kCl>(rc + 20); vf[u] = ld_sh<kCl>(rc + 21);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int ob = ob0 + u * (kBaT / 8);
                // (an observation outside the problem -- a feature below the track count -- has no record: its cells hold zeros and oj = 0 marks it)
                if (ob < n_obs && kk < 6 && vo[u] > 0.0) gst(hpd + (size_t)(int)vf[u] * kBaPS + (int)vo[u] - 1 + kk, vj[u]);
            }
        }
    }
}

// Linearisation of a window that has ONE workgroup (round 6; VERDICT r5 #1a: the batched solve moved 27 x its algorithmic bytes through HBM, a third of it
// the segments' tiles -- written to the L2 scratch by ba_segments and gathered back by ba_reduce_pairs).  Alone in its workgroup a window needs neither trip:
// the eight waves evaluate the segments in ROUNDS of sixteen (wave w: segments 16 r + w and 16 r + 8 + w, as before), leave the sixteen tiles in their slices of
// the LDS stage, and every entry of H_pp / g_p -- owned by one thread, as in the gather -- adds the round's tiles that touch it, straight from LDS, carrying its
// running sums in registers.  THE SAME SUMS IN THE SAME ORDER as ba_reduce_pairs forms from the scratch, so the bytes equal the cluster's:
//   * an entry of frame f adds the window's pairs that touch f in ascending pair order (the gather program's source order: ba_setup), a pair's value being
//     the sum of its segments in segment order, started from zero -- the pairs are sorted by (observer, anchor), segments follow pairs, so a round delivers
//     an entry's sources in exactly that order and a cursor over the frame's pair list (flist, built once per solve) replaces the gather program;
//   * an entry between two frames adds the two pairs that can hold it (either order: a sum of two);
//   * the extrinsic block's 33 values: 15 partial sums over the segments s = g (mod 15), then the 15 in order;
//   * the cost: lane t of wave 0 adds the segments s = t (mod 64) in ascending order, then the wave's butterfly.
// The per-observation records (depth blocks, coupling rows) still go through the scratch: their sums follow the observation order, not the pair order.
// MEASURED (scripts/r6_ba_ab.sh, one box, A / B): byte-identical to the scratch path (the K = 1 vs K = 2 / 4 / 8 tests ran on it) and SLOWER -- 1024 windows
// 33.0 k instead of 41.5 k windows/s, one window with one workgroup 5.7 instead of 4.2 ms.  A round ends in a workgroup barrier, so every round pays the
// full latency of its segments' loads (pair record, slot tables, points, inverse depths: ~3 us) with nothing to overlap it; the scratch path lets each wave
// run through its segments on its own, eight waves' latencies overlapping, and pays the L2 once in the gather.  The tiles' trip through the scratch is
// traffic (a third of the solve's), not time.  Default OFF; kept as a compile-time variant (-DLMONO_BA_LDS_TILES=1, tests/test_bounds_gpu.py builds and
// checks it) because it is the form a solve with a second stage buffer (rounds overlapped two deep) would start from.
#ifndef LMONO_BA_LDS_TILES
#define LMONO_BA_LDS_TILES 0
#endif
template <bool kBig>
__device__ __forceinline__ void ba_linearise_lds(const BaBatch &B, const BaCtx &c, BaLds &L, const BaBig &G, const double *ex, const double *pairdat)
{
    const int tid = threadIdx.x, np_ = c.n_poses, x0 = c.ex_off;
    constexpr int kWS = 2 * kBaRound * kBaRow;                   // doubles of a wave's stage slice
    auto tile_of = [&](int q) { return L.u.stage + (size_t)(q % kBaW) * kWS + (q / kBaW) * kBaPairTile; };      // segment q of the round
    auto pos = [](int r, int q) { return q < 16 ? (r < 16 ? r * 16 + q : 256 + q * 4 + (r - 16)) : 256 + r * 4 + (q - 16); };
    // ---- this thread's entries
    // part A: items a = tid + 512 u of the 55 x 36 frame-block entries (as the gather program numbers them)
    int a_p1[4], a_p2[4], a_o1[4], a_o2[4], a_d1[4], a_d2[4], a_s1[4][2], a_s2[4][2];      // a_s*: the pair's segments [first, end) (none: an empty range)
    double a_v1[4], a_v2[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int a = tid + u * kBaT;
        a_p1[u] = a_p2[u] = -1; a_o1[u] = a_o2[u] = 0; a_d1[u] = a_d2[u] = -1; a_v1[u] = a_v2[u] = 0.0;
        a_s1[u][0] = a_s1[u][1] = a_s2[u][0] = a_s2[u][1] = 0;
        if (a < 55 * 36) {
            const int pb = a / 36, e = a % 36, om = e / 6, on = e % 6;
            int bm = 1;
            while (bm * (bm + 1) / 2 <= pb) bm++;
            const int bn = pb - bm * (bm - 1) / 2;
            if (bm < np_ && bm < kBaMaxPoses) {
                a_p1[u] = L.pair_of[bn * kBaMaxPoses + bm]; a_p2[u] = L.pair_of[bm * kBaMaxPoses + bn];
                a_o1[u] = pos(6 + om, on); a_o2[u] = pos(om, 6 + on);
                a_d1[u] = (ba_pose_off(c, bm) + om) * kBaP + ba_pose_off(c, bn) + on;
                a_d2[u] = (ba_pose_off(c, bn) + on) * kBaP + ba_pose_off(c, bm) + om;
                if (a_p1[u] >= 0) { a_s1[u][0] = L.pair_seg[a_p1[u]]; a_s1[u][1] = L.pair_seg[a_p1[u] + 1]; }
                if (a_p2[u] >= 0) { a_s2[u][0] = L.pair_seg[a_p2[u]]; a_s2[u][1] = L.pair_seg[a_p2[u] + 1]; }
            }
        }
    }
    // part B: items t = tid, tid + 512 of the 11 x 63 entries of one frame (diagonal block, frame x extrinsic, gradient)
    int b_f[2], b_oi[2], b_oj[2], b_dst[2], b_dst2[2], b_cur[2];
    double b_acc[2], b_pacc[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int t = tid + u * kBaT;
        b_f[u] = -1; b_oi[u] = b_oj[u] = 0; b_dst[u] = b_dst2[u] = -1; b_cur[u] = 0; b_acc[u] = 0.0; b_pacc[u] = 0.0;
        if (t < kBaMaxPoses * 63) {
            const int f = t / 63, r = t % 63;
            bool on_ = f < np_;
            int oi_, oj_, dst, dst2 = -1;
            if (r < 21) {
                int om = 0;
                while ((om + 1) * (om + 2) / 2 <= r) om++;
                const int on = r - om * (om + 1) / 2;
                oi_ = pos(om, on); oj_ = pos(6 + om, 6 + on);
                dst = (ba_pose_off(c, f) + om) * kBaP + ba_pose_off(c, f) + on; dst2 = om != on ? (ba_pose_off(c, f) + on) * kBaP + ba_pose_off(c, f) + om : -1;
            } else if (r < 57) {
                const int q = r - 21, om = q / 6, ox = q % 6;
                oi_ = pos(om, 12 + ox); oj_ = pos(6 + om, 12 + ox);
                dst = (ba_pose_off(c, f) + om) * kBaP + x0 + ox; dst2 = (x0 + ox) * kBaP + ba_pose_off(c, f) + om;
                if (x0 < 0) on_ = false;
            } else {
                const int om = r - 57;
                oi_ = pos(om, 18); oj_ = pos(6 + om, 18);
                dst = kBaP * kBaP + ba_pose_off(c, f) + om;
            }
            if (on_) { b_f[u] = f; b_oi[u] = oi_; b_oj[u] = oj_; b_dst[u] = dst; b_dst2[u] = dst2; }
        }
    }
    // part C: value e of the extrinsic block, partial sum g
    constexpr int kG = 15;
    const int c_e = tid % 33, c_g = tid / 33;
    int c_off = 0;
    if (c_e < 16) c_off = (12 + c_e / 4) * 16 + 12 + c_e % 4;
    else if (c_e < 28) c_off = 256 + (12 + (c_e - 16) / 3) * 4 + (c_e - 16) % 3;
    else c_off = 256 + (c_e - 27) * 4 + 3;
    double c_acc = 0.0, cost_acc = 0.0;
    int c_rb15 = 0;                                              // (first segment of the round) mod 15
    // ---- every frame's pairs in pair order: flist[f] = pair | (1 << 7 when f is the pair's observer), in L.gs (dead during a linearisation)
    int *flist = (int *)L.gs;                                    // [kBaMaxPoses][24]: [0] count, [1 ..] entries
    static_assert(sizeof(double) * kBaN >= sizeof(int) * kBaMaxPoses * 24, "flist fits gs");
    __syncthreads();
    if (tid < kBaMaxPoses) {
        int n = 0;
        for (int p = 0; p < c.n_pairs; p++) {
            const int ij = L.pair_ij[p], i = ij & 255, j = ij >> 8;
            if (i == tid) flist[tid * 24 + 1 + n++] = p;
            else if (j == tid) flist[tid * 24 + 1 + n++] = p | 128;
        }
        flist[tid * 24] = n;
    }
    __syncthreads();
    const int n_rounds = (c.n_seg + 2 * kBaW - 1) / (2 * kBaW);
    const double minfo[4] = { gld(B.info + 36), gld(B.info + 37), gld(B.info + 38), gld(B.info + 39) };
    for (int round = 0; round < n_rounds; round++) {
        ba_segments<true, false, kBig, false, true>(B, c, L, G, ex, pairdat, round, minfo);
        __syncthreads();
        const int rb = 2 * kBaW * round, re = min(rb + 2 * kBaW, c.n_seg);      // the round's segments [rb, re)
        // the value a pair adds to an entry: its segments' cells in order, from zero
        auto take = [&](int p, int at, double &pacc) -> bool {              // adds the pair's segments of this round; true = the pair is complete
            const int s0 = L.pair_seg[p], s1 = L.pair_seg[p + 1];
            for (int sg = max(s0, rb); sg < min(s1, re); sg++) { const double v = tile_of(sg - rb)[at]; pacc = sg == s0 ? v : pacc + v; }
            return s1 <= re;
        };
#pragma unroll
        for (int u = 0; u < 4; u++) {
            for (int sg = max(a_s1[u][0], rb); sg < min(a_s1[u][1], re); sg++) { const double v = tile_of(sg - rb)[a_o1[u]]; a_v1[u] = sg == a_s1[u][0] ? v : a_v1[u] + v; }
            for (int sg = max(a_s2[u][0], rb); sg < min(a_s2[u][1], re); sg++) { const double v = tile_of(sg - rb)[a_o2[u]]; a_v2[u] = sg == a_s2[u][0] ? v : a_v2[u] + v; }
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            if (b_f[u] < 0) continue;
            const int *fl = flist + b_f[u] * 24;
            const int n = fl[0];
            while (b_cur[u] < n) {
                const int e = fl[1 + b_cur[u]], p = e & 127;
                if (L.pair_seg[p] >= re) break;                             // the next source comes in a later round
                if (take(p, (e & 128) ? b_oj[u] : b_oi[u], b_pacc[u])) { b_acc[u] += b_pacc[u]; b_cur[u]++; }
                else break;                                                 // the pair goes on in the next round
            }
        }
        if (x0 >= 0 && tid < 33 * kG) {
            // the round's segments s = g (mod 15): at most two of sixteen
            int sg = rb + (c_g - c_rb15 + kG) % kG;
            if (sg < re) { c_acc += tile_of(sg - rb)[c_off]; sg += kG; if (sg < re) c_acc += tile_of(sg - rb)[c_off]; }
        }
        c_rb15 = (c_rb15 + 2 * kBaW) % kG;
        if (tid < 64) { const int sg = rb + ((tid - rb) & 63); if (sg < re) cost_acc += tile_of(sg - rb)[256 + 3]; }
        __syncthreads();                                                    // the stage is rewritten by the next round
    }
    // ---- the sums go to their places (entries nothing touches: zero)
#pragma unroll
    for (int u = 0; u < 4; u++) if (a_d1[u] >= 0) { const double sum = a_v1[u] + a_v2[u]; L.Hpp[a_d1[u]] = sum; L.Hpp[a_d2[u]] = sum; }
#pragma unroll
    for (int u = 0; u < 2; u++)
        if (b_dst[u] >= 0) {
            if (b_dst[u] >= kBaP * kBaP) L.gp[b_dst[u] - kBaP * kBaP] = b_acc[u];
            else { L.Hpp[b_dst[u]] = b_acc[u]; if (b_dst2[u] >= 0) L.Hpp[b_dst2[u]] = b_acc[u]; }
        }
    if (x0 >= 0) {
        if (tid < 33 * kG) L.D[c_g * 33 + c_e] = c_acc;
        __syncthreads();
        if (tid < 33) {
            double tot = 0.0;
#pragma unroll
            for (int g = 0; g < kG; g++) tot += L.D[g * 33 + tid];
            const int e = tid, x4 = x0 + 4, x5 = x0 + 5;
            if (e < 16) L.Hpp[(x0 + e / 4) * kBaP + x0 + e % 4] = tot;
            else if (e < 28) {
                const int om = (e - 16) / 3, cc = (e - 16) % 3;
                if (cc < 2) { L.Hpp[(x0 + om) * kBaP + x4 + cc] = tot; L.Hpp[(x4 + cc) * kBaP + x0 + om] = tot; }
                else L.gp[x0 + om] = tot;
            }
            else if (e == 28) L.Hpp[x4 * kBaP + x4] = tot;
            else if (e == 29) { L.Hpp[x4 * kBaP + x5] = tot; L.Hpp[x5 * kBaP + x4] = tot; }
            else if (e == 30) L.Hpp[x5 * kBaP + x5] = tot;
            else if (e == 31) L.gp[x4] = tot;
            else L.gp[x5] = tot;
        }
    }
    if (tid < 64) { cost_acc = wave_sum_d(cost_acc); if (tid == 0) L.red[3 * kBaW - 1] = cost_acc; }
    __syncthreads();
}

// cost (returned to every thread) and, when kJac, the unscaled normal equations: Hpp, gp, Hdd, gdd in LDS, Hpd in HBM -- the LEADER's view of an
// evaluation (with several workgroups per window the others run ba_follow).
// records_valid: the pose matrices, the inverse depths in LDS and the pair records in HBM were computed by the previous call for the SAME parameter
// values (the candidate evaluation of a step that was then accepted): a linearisation right behind it re-uses them instead of computing the same
// numbers again
template <bool kJac, bool kCl, bool kBig>
__device__ __noinline__ double ba_evaluate(const BaBatch &B, const BaCtx c, BaLds &L_arg, BaBig G, const double *poses, const double *ex, const double *invd,
                                              double *hpd, double *pairdat, bool records_valid = false)
{
    BA_BIND_LDS(L_arg)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __syncthreads();
    BA_TICK(kJac ? 0 : 3)
    if (kCl) ba_publish(B, c, L, kJac ? 1 : 2, poses, ex, invd, records_valid);
    G.vinv = invd;
    if (!records_valid) ba_state<false, kBig>(c, L, poses, ex, invd);
    // wave 1: LASERFactor chain and prior, next to the pair records of wave 0
    // (their residuals / Jacobians wait in gn | va | vb, which are dead during a linearisation, until the pair blocks are in)
    static_assert(3 * kBaN >= 11 * kBaSmallRec, "small-factor records must fit gn | va | vb");
    // (linearisation: wave 1 evaluates them at the head of its pair loop instead -- a single lane per block walks ~1000 double-precision instructions, and
    // behind an accepted step, when the pair records are re-used, the whole workgroup used to wait for it here)
    double small_cost = 0.0;
    // (cost evaluation: wave 1's LASERFactor chain runs beside wave 0's pair records)
    if (!kJac && tid >= 64 && tid < 96) small_cost = ba_small_factors<kJac>(B, c, L.gn, poses, ex, tid - 64);
    if (!records_valid) ba_pair_records(c, L, poses, ex, pairdat);
    double *tiles = B.pairH + (size_t)(c.sg0 + c.pp0 + c.win) * kBaPairTile;
    if (!kJac) {
        __syncthreads();   // pair records are visible
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // (the leader's eight waves take every segment of a candidate's cost, whatever K is)
        BaCtx c1 = c;
        c1.K = 1; c1.GW = kBaW;
        ba_segments<false, false, kBig>(B, c1, L, G, ex, pairdat);
        double *cpart = B.cpart + c.sg0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        ba_segment_cost<false>(c, L, cpart, 1);
        // the LASERFactor chain's and the prior's share: wave 1's lanes 0..31
        if (wave == 1) { small_cost = wave_sum_d(small_cost); if (lane == 0) L.red[3 * kBaW - 2] = small_cost; }
        __syncthreads();
        const double cost = L.red[3 * kBaW - 1] + L.red[3 * kBaW - 2];
        __syncthreads();
        if (kCl && tid == 0) L.eval_no++;
        BA_TOCK(3)
        return cost;
    }

    __syncthreads();   // pair records and zeroed coupling rows are visible
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    BA_TOCK(0)
    BA_TICK(1)
    if (kJac && tid >= 64 && tid < 96) small_cost = ba_small_factors<kJac>(B, c, L.gn, poses, ex, tid - 64);
    // one workgroup per window: the tiles never leave the CU (ba_linearise_lds); a cluster's leader: its share of the segments, records in the L2 scratch
    const bool lds_path = LMONO_BA_LDS_TILES && !kCl && B.lds_ok;
    if (lds_path) ba_linearise_lds<kBig>(B, c, L, G, ex, pairdat);
    else ba_segments<true, false, kBig>(B, c, L, G, ex, pairdat);          // (the leader's own records: plain stores; it reads them back from the L2 like everybody's)
    BA_TOCK(1)
    BA_TICK(11)
    // the cluster shares the ordered sums of this linearisation when all its workgroups sit on the leader's XCD (known from the second linearisation on)
    const bool shared = kCl && L.coloc == 1;
    if (kCl) {
        if (shared) {
            // the leader's own records are out: its "segments done" word lets the followers start on their shares of the sums
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) ba_flag_store(B.bar + (size_t)c.win * kBaBar + 8, (unsigned int)(L.eval_no + 1));
        }
        ba_done<true>(B, c, L);                                  // every workgroup's segment and observation records are written
        if (L.coloc < 0) {
            // where the followers run: each left its XCD + 1 in its word before it answered the first linearisation
            __syncthreads();
            if (tid == 0) {
                const unsigned int *fl = B.bar + (size_t)c.win * kBaBar;
                const unsigned int mine = (unsigned int)ba_xcc_id() + 1u;
                int same = 1;
                for (int r = 1; r < c.K; r++) if (ba_flag_load(fl + 16 + r) != mine) same = 0;
                static_assert(kBaMaxK <= 8, "flag words 16 + r");
                L.coloc = same;
            }
            __syncthreads();
        }
    }
    else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");       // the segment and observation records are written
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");       // ... and read below by other waves: drop this CU's L1 copies
    }
    BA_TICK(12)
    if (lds_path) { }                                                  // (H_pp, g_p and the segments' cost are in LDS already)
    else if (shared) ba_reduce_pairs<true>(B, c, L, BaOutGlob<true>{ B.hred + (size_t)c.win * kBaHred }, 0, c.K);
    else ba_reduce_pairs<kCl>(B, c, L, BaOutLds{ L }, 0, 1);
    BA_TOCK(12)
    BA_TICK(13)
    if (shared) {
        ba_feature_pass<true, kBig, true>(B, c, L, G, hpd, 0, c.K);
        // the other workgroups' shares: wait for their "sums done" words, drop this CU's L1 (their stores went through the shared L2; the coupling rows are
        // read with plain loads from here on), load H_pp | g_p and H_ff | g_f
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned int *flags = B.bar + (size_t)c.win * kBaBar;
        if (tid < 64 && !L.failed) { if (!ba_wait_flags(flags + 9, c.K - 1, (unsigned int)(L.eval_no + 1), B.fail) && tid == 0) L.failed = 1; }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const double *hr = B.hred + (size_t)c.win * kBaHred;
        for (int k = tid; k < kBaHred; k += kBaT) { const double v = ld_sh<true>(hr + k); if (k >= kBaP * kBaP) L.gp[k - kBaP * kBaP] = v; else L.Hpp[k] = v; }
        if (!kBig) {
            const double *fd = B.fdg + (size_t)c.win * 2 * B.feat_cap;
            for (int f = tid; f < c.F; f += kBaT) { L.Hdd[f] = ld_sh<true>(fd + f); L.gdd[f] = ld_sh<true>(fd + B.feat_cap + f); }
        }
        __syncthreads();
    } else
        ba_feature_pass<kCl, kBig, false>(B, c, L, G, hpd, 0, 1);
    BA_TOCK(13)
    // every segment's block is in H_pp: the LASERFactor chain left in gn | va | vb is added by everybody, the prior by wave 1 -- after the pair
    // blocks in every run
    ba_small_accumulate_block(c, L, L.gn, tid);
    if (wave == 1) {
        ba_prior_accumulate_wave(c, L, L.gn, lane);
        small_cost = wave_sum_d(small_cost);
        if (lane == 0) L.red[3 * kBaW - 2] = small_cost;
    }
    if (!lds_path) ba_segment_cost<kCl>(c, L, tiles + 256 + 3, kBaPairTile);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    const double cost = L.red[3 * kBaW - 1] + L.red[3 * kBaW - 2];
    __syncthreads();
    if (kCl && tid == 0) L.eval_no++;
    BA_TOCK(11)
    // the coupling rows were written through L2: drop this CU's L1 copies before they are read with plain loads
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return cost;
}

// vb = Hs va (Jacobi-scaled).  gn is used as scratch (it is dead until the next solve).
template <bool kBig>
__device__ __noinline__ void ba_hs_mul(const BaCtx c, BaLds &L_arg, const BaBig G, const double *hpd)
{
    BA_BIND_LDS(L_arg)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = c.P, F = c.F, N = P + F;
    __syncthreads();
    BA_TICK(4)
    for (int k = tid; k < N; k += kBaT) BA_V(gn, k) = BA_V(scale, k) * BA_V(va, k);
    __syncthreads();
#if LMONO_BA_HS_FUSED
    // ONE pass over the coupling rows (round 6: the camera rows and the depth rows each read all of them -- 2 x 275 KB per product for a 430-feature
    // window, a quarter of what the batched solve moves through HBM): wave w takes the features f = w (mod 8), lanes own the parameter columns
    // (coalesced 640-B rows), eight rows in flight.  A row's share of the camera rows is row x (feature's entry of the vector); its own depth row is the
    // dot product of the row with the pose part of the vector, summed over the wave (DPP row steps; the eight sums of a round are independent chains).
    {
        const double gp0 = lane < P ? BA_V(gn, lane) : 0.0, gp1 = 64 + lane < P ? BA_V(gn, 64 + lane) : 0.0;
        double a0 = 0, a1 = 0;
        for (int f = wave; f < F; f += 8 * kBaW) {
            double r0[8], r1[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int fu = f + kBaW * u;
                const double *row = hpd + (size_t)fu * kBaPS;
                r0[u] = (fu < F && lane < P) ? gld(row + lane) : 0.0;
                r1[u] = (fu < F && 64 + lane < P) ? gld(row + 64 + lane) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int fu = f + kBaW * u;
                const double w = fu < F ? BA_V(gn, P + fu) : 0.0;
                a0 += r0[u] * w; a1 += r1[u] * w;
                const double dot = wave_sum_d_lane63(r0[u] * gp0 + r1[u] * gp1);
                if (lane == 63 && fu < F) BA_V(vb, P + fu) = (dot + BA_F(Hdd, fu) * w) * BA_V(scale, P + fu);
            }
        }
        L.u.hs_part[wave][lane] = a0;
        if (lane < kBaPS - 64) L.u.hs_part[wave][64 + lane] = a1;
    }
#else
    // camera rows: wave w sums the features f = w (mod 8); lanes own the parameter columns (coalesced 640-B rows),
    // eight feature rows in flight
    {
        double a0 = 0, a1 = 0;
        for (int f = wave; f < F; f += 8 * kBaW) {
            double r0[8], r1[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int fu = f + kBaW * u;
                const double *row = hpd + (size_t)fu * kBaPS;
                r0[u] = fu < F ? gld(row + lane) : 0.0;
                r1[u] = (fu < F && lane < kBaPS - 64) ? gld(row + 64 + lane) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int fu = f + kBaW * u;
                const double w = fu < F ? BA_V(gn, P + fu) : 0.0;
                a0 += r0[u] * w; a1 += r1[u] * w;
            }
        }
        L.u.hs_part[wave][lane] = a0;
        if (lane < kBaPS - 64) L.u.hs_part[wave][64 + lane] = a1;
    }
    // depth rows: four lanes per feature, the 18 entries of a lane requested together
    for (int f = tid >> 2; f < F; f += kBaT / 4) {
        const double *row = hpd + (size_t)f * kBaPS;
        double rv[18];
#pragma unroll
        for (int k = 0; k < 18; k++) { const int a = (tid & 3) + 4 * k; rv[k] = a < P ? gld(row + a) : 0.0; }
        double acc = 0;
#pragma unroll
        for (int k = 0; k < 18; k++) { const int a = (tid & 3) + 4 * k; acc += rv[k] * (a < P ? BA_V(gn, a) : 0.0); }
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        if ((tid & 3) == 0) BA_V(vb, P + f) = (acc + BA_F(Hdd, f) * BA_V(gn, P + f)) * BA_V(scale, P + f);
    }
#endif
    __syncthreads();
    if (tid < P) {
        double acc = 0;
#pragma unroll
        for (int w = 0; w < kBaW; w++) acc += L.u.hs_part[w][tid];
        for (int b = 0; b < P; b++) acc += L.Hpp[b * kBaP + tid] * BA_V(gn, b);   // H_pp is symmetric: column walk, no bank conflicts
        BA_V(vb, tid) = acc * BA_V(scale, tid);
    }
    __syncthreads();
    BA_TOCK(4)
}

__constant__ int kBaTileM[16] = { 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 3, 3, 4, -1 };
__constant__ int kBaTileN[16] = { 0, 1, 2, 3, 4, 1, 2, 3, 4, 2, 3, 4, 3, 4, 4, -1 };

__device__ __forceinline__ double readlane_d(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// solve (Hs + mu diag(D2)) x = gs by Schur elimination of the depth columns; result in L.gn; returns success to all
template <bool kBig>
__device__ __noinline__ bool ba_schur_solve(const BaCtx c, BaLds &L_arg, const BaBig G, const double *hpd, double mu)
{
    BA_BIND_LDS(L_arg)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, P = c.P, F = c.F;
    const int col = lane & 15, kq = lane >> 4;
    __syncthreads();
    BA_TICK(5)
    for (int a = tid; a < kBaPS; a += kBaT) L.rhs[a] = a < P ? BA_V(gs, a) : 0.0;
    if (tid == 0) L.ok = 1;
    // upper 16x16 tiles of E diag(1 / h_ff) E^T: tiles 2 wave and 2 wave + 1
    ba_d4 acc[2];
    int tm[2], tn[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        acc[u] = ba_d4{ 0, 0, 0, 0 };
        tm[u] = kBaTileM[2 * wave + u]; tn[u] = kBaTileN[2 * wave + u];
    }
    for (int f0 = 0; f0 < F; f0 += kBaFT) {
        const int nf = min(kBaFT, F - f0);
        __syncthreads();
        {
            // 64 x 80 coupling entries, 10 per thread, all requested before the first is used
            double v[10];
#pragma unroll
            for (int qq = 0; qq < 10; qq++) {
                const int k = tid + kBaT * qq, t = k / kBaPS, a = k - t * kBaPS;
                v[qq] = (t < nf && a < P) ? gld(hpd + (size_t)(f0 + t) * kBaPS + a) : 0.0;
            }
#pragma unroll
            for (int qq = 0; qq < 10; qq++) {
                const int k = tid + kBaT * qq, t = k / kBaPS, a = k - t * kBaPS;
                L.u.sch.et[k] = (t < nf && a < P) ? v[qq] * BA_V(scale, a) * BA_V(scale, P + f0 + t) : 0.0;
            }
        }
        if (tid < kBaFT) {
            double ic = 0.0, gi = 0.0;
            if (tid < nf) {
                const double s = BA_V(scale, P + f0 + tid);
                const double hff = BA_F(Hdd, f0 + tid) * s * s + mu * BA_V(D, P + f0 + tid) * BA_V(D, P + f0 + tid);
                if (!(hff > 0.0)) L.ok = 0;
                ic = 1.0 / hff;
                gi = BA_V(gs, P + f0 + tid) * ic;
            }
            L.u.sch.ic[tid] = ic; L.u.sch.gi[tid] = gi;
        }
        __syncthreads();
        // right-hand side: thread (a, part) sums a quarter of the tile's features
        if (tid < 4 * kBaPS) {
            const int a = tid % kBaPS, part = tid / kBaPS;
            double s = 0;
#pragma unroll 4
            for (int t = part * (kBaFT / 4); t < (part + 1) * (kBaFT / 4); t++) s += L.u.sch.et[t * kBaPS + a] * L.u.sch.gi[t];
            L.u.sch.part[part][a] = s;
        }
        const int nks = (nf + 3) >> 2;
        for (int ks = 0; ks < nks; ks++) {
            const int t = 4 * ks + kq;
            const double *er = L.u.sch.et + t * kBaPS;
            const double icv = L.u.sch.ic[t];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                if (tm[u] < 0) continue;
                const double a = er[16 * tm[u] + col];
                const double b = er[16 * tn[u] + col] * icv;
                acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u], 0, 0, 0);
            }
        }
        __syncthreads();
        if (tid < P) L.rhs[tid] -= ((L.u.sch.part[0][tid] + L.u.sch.part[1][tid]) + L.u.sch.part[2][tid]) + L.u.sch.part[3][tid];
    }
    __syncthreads();   // the staged tile is dead: S takes its place
    double *S = L.u.fac.S;
#pragma unroll
    for (int u = 0; u < 2; u++) {
        if (tm[u] < 0) continue;
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const int m = 16 * tm[u] + kq + 4 * v, n = 16 * tn[u] + col;
            if (m < P && n < P) {
                const double val = L.Hpp[m * kBaP + n] * BA_V(scale, m) * BA_V(scale, n) + (m == n ? mu * BA_V(D, m) * BA_V(D, m) : 0.0) - acc[u][v];
                S[m * kBaSS + n] = val;
                if (tm[u] != tn[u]) S[n * kBaSS + m] = val;
            }
        }
    }
    __syncthreads();
    // the right-hand side rides along as row P of the matrix: its factor entries are the forward substitution L y = b
    if (tid < P) S[P * kBaSS + tid] = L.rhs[tid];
    __syncthreads();
    BA_TOCK(5)
    BA_TICK(6)
    // Blocked in-place lower Cholesky (round 5).  Wave 0 owns the 8-column panels: it holds the panel in registers (lane = row), factors it
    // right-looking (the pivot row is broadcast with v_readlane; only  readlane -> rsqrt -> scale -> readlane -> fma  is on the chain of a column, the
    // LDS is not) and stores it.  The trailing update S -= L_panel L_panel^T runs on the matrix cores as 16 x 16 tiles (the panel's 8 columns = two
    // k-steps of v_mfma_f64_16x16x4_f64) in two parts: the first tile COLUMN -- it holds the next panel -- by waves 1..5 at once, one tile each, between
    // two workgroup barriers; the other tiles by waves 1..7 while wave 0 already factors the next panel.  One wave alone issues a double-precision
    // instruction every ~8 cycles, so the chain is priced in wave 0's instructions: ~270 per panel for the factorisation; a first version of this
    // round applied the panel to the next one in wave 0's registers as well (+150 instructions per panel: 49 k cycles per factorisation; round 4,
    // with the whole trailing update between two panels and an LDS write + wait in every column step: 52 k).
    // Panels 0 and 1 reach past 64 rows (rows p0 .. P, P = 72): lane r carries rows r AND 64 + r (a, b).  From panel 2 on the remaining rows fit one
    // register set: lane l carries row p0 + l, the pivot of column k sits in lane k.  A tile entry receives the panels in ascending order.
    auto chol_tile = [&](int r0, int ti, int tk, int pc0) {
        const int ra = r0 + 16 * ti + col, rb = r0 + 16 * tk + col;       // operand rows of this lane
        ba_d4 acc4 = { 0, 0, 0, 0 };
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            const int q = 4 * ks + kq;
            const double va = S[min(ra, P) * kBaSS + pc0 + q], vb = S[min(rb, P) * kBaSS + pc0 + q];
            const double a = ra <= P ? va : 0.0;
            const double bq = rb <= P ? vb : 0.0;
            acc4 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bq, acc4, 0, 0, 0);
        }
        // read-modify-write of the tile: the four reads go out together (unconditional, clamped), the writes are masked
        const int nn = r0 + 16 * tk + col, nc = min(nn, kBaSS - 1);
        double cur[4];
#pragma unroll
        for (int v = 0; v < 4; v++) cur[v] = S[min(r0 + 16 * ti + kq + 4 * v, P) * kBaSS + nc];
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const int m = r0 + 16 * ti + kq + 4 * v;
            if (m <= P && nn < P && nn <= m) S[m * kBaSS + nn] = cur[v] - acc4[v];
        }
    };
    double cid0 = 0.0, cid1 = 0.0;                  // wave 0: 1 / l_rr of rows lane and 64 + lane (the backward substitution reads them)
    {
        double a[8], b[8];                          // wave 0: the current panel; single-set panels use a[] only
        bool bad = false;
        const int ra_ = min(lane, P), rb_ = min(64 + lane, P);      // this lane's rows in the two-set map, clamped into the matrix
        // 1 / sqrt(d): v_rsq_f64 and one third-order correction, y (1 + e / 2 + 3 e^2 / 8) with e = 1 - d y^2
        auto rsqrt3 = [](double d) { double y = __builtin_amdgcn_rsq(d); const double e = __builtin_fma(-d * y, y, 1.0); return __builtin_fma(y * e, __builtin_fma(e, 0.375, 0.5), y); };
        // the panel that starts at column q0: unconditional LDS reads at clamped addresses + a select (a masked read is a branch with its own s_waitcnt)
        auto load_panel = [&](int q0) {
            if (q0 < 16) {
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const double va = S[ra_ * kBaSS + q0 + k], vb = S[rb_ * kBaSS + q0 + k];
                    a[k] = lane <= P ? va : 0.0;
                    b[k] = 64 + lane <= P ? vb : 0.0;
                }
            } else {
                const int row = q0 + lane, rc = min(row, P);
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const double va = S[rc * kBaSS + min(q0 + k, P - 1)];
                    a[k] = (row <= P && q0 + k < P) ? va : 0.0;
                }
            }
        };
        if (wave == 0) { __builtin_amdgcn_s_setprio(3); load_panel(0); }
        for (int p0 = 0; p0 < P; p0 += 8) {
            const int pw = min(8, P - p0);
            if (wave == 0) {
                BA_TICK(14)
                if (p0 < 16) {
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        double d = readlane_d(a[k], p0 + k);
                        if (!(d > 0.0)) { bad = true; d = 1.0; }
                        const double y = rsqrt3(d);
                        if (lane == p0 + k) cid0 = y;
                        a[k] *= y; b[k] *= y;                      // the pivot row's own entry becomes d / sqrt(d)
#pragma unroll
                        for (int jj = k + 1; jj < 8; jj++) {
                            const double ljk = readlane_d(a[k], p0 + jj);
                            a[jj] -= a[k] * ljk; b[jj] -= b[k] * ljk;
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        if (lane <= P && lane >= p0 + k) S[lane * kBaSS + p0 + k] = a[k];
                        if (64 + lane <= P) S[(64 + lane) * kBaSS + p0 + k] = b[k];
                    }
                } else {
                    const int row = p0 + lane;
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        double d = readlane_d(a[k], k);
                        if (k >= pw) d = 1.0;                      // a column past the matrix (last panel of P = 66): zeros, never stored
                        if (!(d > 0.0)) { bad = true; d = 1.0; }
                        const double y = rsqrt3(d);
                        if (lane == ((p0 + k) & 63)) { if (p0 + k >= 64) cid1 = y; else cid0 = y; }
                        a[k] *= y;
#pragma unroll
                        for (int jj = k + 1; jj < 8; jj++) a[jj] -= a[k] * readlane_d(a[k], jj);
                    }
#pragma unroll
                    for (int k = 0; k < 8; k++) if (k < pw && row <= P && lane >= k) S[row * kBaSS + p0 + k] = a[k];
                }
                BA_TOCK(14)
            }
            BA_TICK(15)
            __syncthreads();                                                   // the panel is in LDS
            const int r0 = p0 + 8;
            if (r0 >= P) { BA_TOCK(15) break; }
            const int nt = (P + 1 - r0 + 15) >> 4;                            // tile rows of the trailing matrix (<= 5)
            if (wave >= 1 && wave <= nt) chol_tile(r0, wave - 1, 0, p0);       // the first tile column: the next panel's (and the one after's) columns
            __syncthreads();
            BA_TOCK(15)
            if (wave == 0) {
                BA_TICK(10)
                load_panel(r0);
                BA_TOCK(10)
            } else {
                // the other tile columns, while wave 0 factors (wave 4 shares its SIMD with wave 0, whose column chain is the critical path: it takes none)
                for (int t = wave < 4 ? wave - 1 : wave - 2; wave != 4 && t < nt * (nt - 1) / 2; t += kBaW - 2) {
                    int ti = 1, rem = t;                                          // tiles (ti, tk) with 1 <= tk <= ti
                    while (rem >= ti) { rem -= ti; ti++; }
                    chol_tile(r0, ti, rem + 1, p0);
                }
            }
        }
        if (wave == 0 && bad && lane == 0) L.ok = 0;
    }
    BA_TOCK(6)
    BA_TICK(7)
    // Backward substitution L^T x = y by wave 0 (y = row P of the factor).  Lane j carries z_j = y_j / l_jj for rows j and 64 + j and the rows of
    // L it reads are scaled by 1 / l_jj on the way in, so a step is  x_k = readlane(z, k);  z_j -= (l_kj / l_jj) x_k  -- the division and the LDS
    // reads are off the dependent chain (round 4: a loop with a vector loop counter, masked LDS reads and an s_waitcnt per step: 360 cycles per step).
    // Rows are requested eight at a time; every index is a compile-time constant after unrolling.
    if (wave == 0) {
        const double id0 = lane < P ? cid0 : 0.0, id1 = 64 + lane < P ? cid1 : 0.0;
        // (unconditional reads of the 73-row LDS array + selects, see the panel loads)
        const double y0 = S[P * kBaSS + lane], y1 = S[P * kBaSS + min(64 + lane, kBaSS - 1)];
        double z0 = (lane < P ? y0 : 0.0) * id0, z1 = (64 + lane < P ? y1 : 0.0) * id1;
        if (P > 64) {
            double l0[8], l1[8];
#pragma unroll
            for (int kk = 0; kk < 8; kk++) {
                const int k = 64 + kk;
                const double v0 = S[k * kBaSS + lane], v1 = S[k * kBaSS + 64 + (lane & 7)];
                l0[kk] = k < P ? v0 * id0 : 0.0;
                l1[kk] = (k < P && lane < kk) ? v1 * id1 : 0.0;
            }
#pragma unroll
            for (int kk = 7; kk >= 0; kk--) {
                const double xk = readlane_d(z1, kk);          // rows >= P carry z = 0 and l = 0: no effect
                z0 -= l0[kk] * xk; z1 -= l1[kk] * xk;
            }
        }
#pragma unroll
        for (int kb = 56; kb >= 0; kb -= 8) {
            if (kb < P) {
                double l0[8];
#pragma unroll
                for (int kk = 0; kk < 8; kk++) {
                    const int k = kb + kk;
                    const double v0 = S[k * kBaSS + lane];
                    l0[kk] = (k < P && lane < k) ? v0 * id0 : 0.0;
                }
#pragma unroll
                for (int kk = 7; kk >= 0; kk--) {
                    const double xk = readlane_d(z0, kb + kk);
                    z0 -= l0[kk] * xk;
                }
            }
        }
        if (lane < P) { BA_V(gn, lane) = z0; if (!isfinite(z0)) L.ok = 0; }
        if (64 + lane < P) { BA_V(gn, 64 + lane) = z1; if (!isfinite(z1)) L.ok = 0; }
        __builtin_amdgcn_s_setprio(0);
    }
    __syncthreads();
    BA_TOCK(7)
    BA_TICK(8)
    for (int k = tid; k < P; k += kBaT) L.rhs[k] = BA_V(scale, k) * BA_V(gn, k);
    __syncthreads();
    for (int f = tid >> 2; f < F; f += kBaT / 4) {
        const double *row = hpd + (size_t)f * kBaPS;
        double rv[18];
#pragma unroll
        for (int k = 0; k < 18; k++) { const int a = (tid & 3) + 4 * k; rv[k] = a < P ? gld(row + a) : 0.0; }
        double acc = 0;
#pragma unroll
        for (int k = 0; k < 18; k++) { const int a = (tid & 3) + 4 * k; acc += rv[k] * (a < P ? L.rhs[a] : 0.0); }
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        if ((tid & 3) == 0) {
            const double s = BA_V(scale, P + f);
            const double hff = BA_F(Hdd, f) * s * s + mu * BA_V(D, P + f) * BA_V(D, P + f);
            const double x = (BA_V(gs, P + f) - acc * s) / hff;
            BA_V(gn, P + f) = x;
            if (!isfinite(x)) L.ok = 0;
        }
    }
    __syncthreads();
    BA_TOCK(8)
    return L.ok != 0;
}

// Window tables that live in LDS for the whole solve: (anchor, observer) -> pair, the first observation and the anchor frame of every feature
template <bool kBig>
__device__ __noinline__ void ba_setup(const BaBatch &B, const BaCtx c, BaLds &L_arg)
{
    BA_BIND_LDS(L_arg)
    const int tid = threadIdx.x;
    __syncthreads();
    for (int k = tid; k < kBaMaxPoses * kBaMaxPoses; k += kBaT) L.pair_of[k] = -1;
    if (!kBig) for (int f = tid; f <= c.F; f += kBaT) L.fobs[f] = (unsigned short)(*BA_P(B.feat_obs_off + c.f0 + f) - c.o0);      // <= 448 x 10 observations per window
    if (!kBig) for (int f = tid; f < c.F; f += kBaT) L.fanchor[f] = (signed char)*BA_P(B.feat_anchor + c.f0 + f);
    __syncthreads();
    // (anchor, observer) -> the pair's index in the window's pair list
    for (int p = tid; p < c.n_pairs; p += kBaT) {
        const int ij = L.pair_ij[p];
        L.pair_of[(ij & 255) * kBaMaxPoses + (ij >> 8)] = (short)p;
    }
    __syncthreads();
    if (c.rank != 0) return;
    // the gather program of ba_reduce_pairs (see kBaGaN); position of (row r, column q) of a tile: rows / columns 0..5 = frame i, 6..11 = frame j,
    // 12..15 = extrinsic 0..3 (the 16 x 16 tile), column 16 / 17 = extrinsic 4 / 5 and column 18 = the residual (the 16 x 4 side tile)
    auto pos = [](int r, int q) { return q < 16 ? (r < 16 ? r * 16 + q : 256 + q * 4 + (r - 16)) : 256 + r * 4 + (q - 16); };
    // a source = a frame pair: (position of the entry in the tile of the pair's FIRST segment) | (its segments << 20); 0 = no such pair
    auto source = [&](int p, int at) { const int s0 = L.pair_seg[p], cnt = L.pair_seg[p + 1] - s0; return (s0 * kBaPairTile + at) | (cnt << 20); };
    static_assert((kBaMaxSeg + 64 * kBaMaxPoses) * kBaPairTile < (1 << 20) && kBaMaxFeat / kBaSeg + 1 < (1 << 11), "gather source packing");
    const int zero = 0;
    int *gp = B.gprog + (size_t)c.win * kBaGprog;
    const int np_ = c.n_poses, x0 = c.ex_off;
    for (int a = tid; a < kBaGaN; a += kBaT) {
        int o1 = zero, o2 = zero, d1 = -1, d2 = -1;
        const int pb = a / 36, e = a % 36, om = e / 6, on = e % 6;
        int bm = 1;
        while (bm * (bm + 1) / 2 <= pb) bm++;                      // pb = bm (bm - 1) / 2 + bn, bn < bm
        const int bn = pb - bm * (bm - 1) / 2;
        if (bm < np_ && bm < kBaMaxPoses) {
            // frame bn = i is the smaller one: the pair (i, j = bm) holds the block in rows 6.. / columns 0.., the pair (j, i) in rows 0.. / columns 6..
            const int p1 = L.pair_of[bn * kBaMaxPoses + bm], p2 = L.pair_of[bm * kBaMaxPoses + bn];
            if (p1 >= 0) o1 = source(p1, pos(6 + om, on));
            if (p2 >= 0) o2 = source(p2, pos(om, 6 + on));
            d1 = (ba_pose_off(c, bm) + om) * kBaP + ba_pose_off(c, bn) + on;
            d2 = (ba_pose_off(c, bn) + on) * kBaP + ba_pose_off(c, bm) + om;
        }
        *BA_P(gp + a) = o1; *BA_P(gp + kBaGaN + a) = o2; *BA_P(gp + 2 * kBaGaN + a) = d1; *BA_P(gp + 3 * kBaGaN + a) = d2;
    }
    int *gb = gp + 4 * kBaGaN;
    for (int t = tid; t < kBaGbN; t += kBaT) {
        const int f = t / 63, r = t % 63;
        int oi_ = 0, oj_ = 0, dst = -1, dst2 = -1;
        bool on_ = f < np_ && f < kBaMaxPoses;
        if (r < 21) {                                              // lower triangle of the frame's diagonal block, mirrored
            int om = 0;
            while ((om + 1) * (om + 2) / 2 <= r) om++;
            const int on = r - om * (om + 1) / 2;
            oi_ = pos(om, on); oj_ = pos(6 + om, 6 + on);
            dst = (ba_pose_off(c, f) + om) * kBaP + ba_pose_off(c, f) + on; dst2 = om != on ? (ba_pose_off(c, f) + on) * kBaP + ba_pose_off(c, f) + om : -1;
        } else if (r < 57) {                                       // (frame row om, extrinsic column ox) and its mirror
            const int q = r - 21, om = q / 6, ox = q % 6;
            oi_ = pos(om, 12 + ox); oj_ = pos(6 + om, 12 + ox);
            dst = (ba_pose_off(c, f) + om) * kBaP + x0 + ox; dst2 = (x0 + ox) * kBaP + ba_pose_off(c, f) + om;
            if (x0 < 0) on_ = false;
        } else {                                                   // gradient
            const int om = r - 57;
            oi_ = pos(om, 18); oj_ = pos(6 + om, 18);
            dst = kBaP * kBaP + ba_pose_off(c, f) + om;
        }
        for (int k = 0; k < kBaMaxPoses; k++) {
            const int pa = (on_ && k < np_ && k != f) ? L.pair_of[f * kBaMaxPoses + k] : -1;
            const int pb = (on_ && k < np_ && k != f) ? L.pair_of[k * kBaMaxPoses + f] : -1;
            // (sources 0..10: the pairs (k, f) that OBSERVE in f, by ascending k; 11..21: the pairs (f, k) anchored in f, by ascending k -- with the pairs
            // sorted by (observer, anchor) and anchors preceding their observers that is ascending pair order: the order a one-workgroup solve meets them in)
            *BA_P(gb + k * kBaGbN + t) = pb >= 0 ? source(pb, oj_) : zero;
            *BA_P(gb + (kBaMaxPoses + k) * kBaGbN + t) = pa >= 0 ? source(pa, oi_) : zero;
        }
        *BA_P(gb + 22 * kBaGbN + t) = on_ ? dst : -1; *BA_P(gb + 23 * kBaGbN + t) = on_ ? dst2 : -1;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// A follower workgroup of a window (rank > 0): waits for the leader's mail, computes its own copy of the state's records and, for a linearisation,
// takes its share of the segments and reports; until the leader sends command 3.
template <bool kBig>
__device__ __noinline__ void ba_follow(const BaBatch &B, const BaCtx c, BaLds &L_arg, BaBig G, double *pairdat)
{
    BA_BIND_LDS(L_arg)
    const int tid = threadIdx.x;
    const double *mb = B.mbox + (size_t)c.win * kBaMbox;
    unsigned int *flags = B.bar + (size_t)c.win * kBaBar;
    G.vinv = mb + 96;                                   // (kBig: a follower reads the inverse depths straight from the mail box)
    if (tid == 0) ba_flag_store(flags + 16 + c.rank, (unsigned int)ba_xcc_id() + 1u);
    double *hpd = B.hpd + (size_t)c.win * B.feat_cap * kBaPS;
    for (;;) {
        if (tid < 64) { if (!ba_wait_flags(flags, 1, (unsigned int)(L.eval_no + 1), B.fail) && tid == 0) L.failed = 1; }
        __syncthreads();
        if (L.failed) return;
        const int cmd = (int)ld_sh<true>(mb), reuse = (int)ld_sh<true>(mb + 1), shared = (int)ld_sh<true>(mb + 3);
        if (cmd == 3) return;
        if (!reuse) {
            for (int k = tid; k < 7 * c.n_poses; k += kBaT) L.cposes[k] = ld_sh<true>(mb + 8 + k);
            if (tid < 7) L.cex[tid] = ld_sh<true>(mb + 88 + tid);
            if (!kBig) for (int f = tid; f < c.F; f += kBaT) L.vinv[f] = ld_sh<true>(mb + 96 + f);
        }
        __syncthreads();
        // a candidate: the mail box is read -- acknowledge, then compute the records in the leader's shadow
        if (cmd == 2 && tid == 0) ba_flag_store(flags + c.rank, (unsigned int)(L.eval_no + 1));
        if (!reuse) { ba_state<true, kBig>(c, L, L.cposes, L.cex, nullptr); ba_pair_records(c, L, L.cposes, L.cex, pairdat); }
        __syncthreads();   // pair records are visible
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (cmd == 1) {
            if ((int)ld_sh<true>(mb + 2) == ba_xcc_id()) ba_segments<true, false, kBig, true>(B, c, L, G, L.cex, pairdat);      // the leader's XCD: plain stores stay in the shared L2
            else ba_segments<true, true, kBig, true>(B, c, L, G, L.cex, pairdat);
            ba_done<false>(B, c, L);
            if (shared) {
                // every workgroup's records (the other followers' and the leader's): then this workgroup's share of the ordered sums and of the per-feature
                // pass -- an entry is formed by one thread from the same records in the same order whoever that thread is
                if (tid < 64) {
                    bool ok = ba_wait_flags(flags + 1, c.K - 1, (unsigned int)(L.eval_no + 1), B.fail);
                    ok = ok && ba_wait_flags(flags + 8, 1, (unsigned int)(L.eval_no + 1), B.fail);
                    if (!ok && tid == 0) L.failed = 1;
                }
                __syncthreads();
                if (L.failed) return;
                ba_reduce_pairs<true>(B, c, L, BaOutGlob<true>{ B.hred + (size_t)c.win * kBaHred }, c.rank, c.K);
                ba_feature_pass<true, kBig, true>(B, c, L, G, hpd, c.rank, c.K);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) ba_flag_store(flags + 8 + c.rank, (unsigned int)(L.eval_no + 1));
            }
        }
        if (tid == 0) L.eval_no++;
        __syncthreads();
    }
}

// K workgroups per window (kCl: K > 1): block b runs on XCD b mod 8 as dispatched today, so the K workgroups of a window are given the same b mod 8
// (a shared L2 makes their exchange cheaper; nothing depends on it: every hand-off is device-coherent and ordered by the arrival counters)
// spread (test hook, LMONO_BA_SPREAD=1): workgroup r of a window is given the residue (x + r) mod 8 instead, i.e. the K workgroups of a window land on K
// different XCDs -- the placement the exchange must also be right on
template <bool kCl, bool kBig>
__global__ __launch_bounds__(kBaT) void k_ba_solve(BaBatch B, int K, int spread)
{
    BaLds &L = g_ba_lds;
    const int tid = threadIdx.x;
    BaCtx c;
    c.K = kCl ? K : 1;
    c.rank = kCl ? ((int)blockIdx.x / 8) % K : 0;
    const int w = kCl ? ((int)blockIdx.x / 8 / K) * 8 + (((int)blockIdx.x % 8) + ((spread & 1) ? 8 - c.rank : 0)) % 8 : (int)blockIdx.x;
    c.GW = c.K == 1 ? kBaW : c.K * kBaW - 1;
    c.win = w;
    if (w >= B.n_windows) return;
#ifdef LMONO_BOUNDS
    if (tid == 0) { g_ba_lo_hi[0] = B.blob_lo; g_ba_lo_hi[1] = B.blob_hi; }
    __syncthreads();
#endif
    c.n_poses = *BA_P(B.flags + w * 4 + 0); c.use_prior = *BA_P(B.flags + w * 4 + 1); c.ex_constant = *BA_P(B.flags + w * 4 + 2); c.use_mono = *BA_P(B.flags + w * 4 + 3);
    c.f0 = *BA_P(B.feat_off + w); c.o0 = *BA_P(B.obs_off + w);
    c.F = c.use_mono ? *BA_P(B.feat_off + w + 1) - c.f0 : 0;
    c.ex_off = c.ex_constant ? -1 : 0;
    c.P = 6 * c.n_poses + (c.ex_constant ? 0 : 6);
    c.pp0 = *BA_P(B.pair_off + w); c.n_pairs = c.use_mono ? *BA_P(B.pair_off + w + 1) - c.pp0 : 0;
    c.ps0 = *BA_P(B.pobs_off + w); c.n_slots = c.use_mono ? *BA_P(B.pobs_off + w + 1) - c.ps0 : 0;
    c.sg0 = *BA_P(B.seg_off + w); c.n_seg = c.use_mono ? *BA_P(B.seg_off + w + 1) - c.sg0 : 0; c.n_multi = c.use_mono ? *BA_P(B.n_multi + w) : 0;
    const int P = c.P, F = c.F, N = P + F;
    double *gposes = B.poses + (size_t)w * kBaMaxPoses * 7, *gex = B.ex + (size_t)w * 7, *ginvd = B.inv_depth + c.f0;
    double *hpd = B.hpd + (size_t)w * B.feat_cap * kBaPS;
    double *pairdat = B.pairdat + ((size_t)c.rank * B.n_pairs_total + c.pp0) * kBaPairRec;       // (one copy per workgroup of a window)
    double *cinvd = B.cand + (size_t)w * B.feat_cap;
    BaBig G;
    {
        double *gv = kBig ? B.bigv + (size_t)w * 8 * kBaMaxFeat : nullptr;
        G.Hdd = gv; G.gdd = gv + kBaMaxFeat; G.scale = gv + 2 * kBaMaxFeat; G.D = gv + 3 * kBaMaxFeat; G.gs = gv + 4 * kBaMaxFeat; G.gn = gv + 5 * kBaMaxFeat;
        G.va = gv + 6 * kBaMaxFeat; G.vb = gv + 7 * kBaMaxFeat; G.vinv = ginvd; G.fobs = B.feat_obs_off + c.f0; G.fanchor = B.feat_anchor + c.f0; G.seg = B.seg_tab + c.sg0; G.o0 = c.o0;
    }
    // (spread bit 1 = LMONO_BA_SHARE_SUMS: the cluster shares the ordered sums; off by default -- measured slower, see ba_reduce_pairs)
    if (tid == 0) { L.eval_no = 0; L.failed = 0; L.coloc = (spread & 2) ? -1 : 0; }
#ifdef LMONO_BA_PROF
    if (tid < 24) L.prof[tid] = 0;
#endif
    for (int k = tid; k < c.n_poses * 7; k += kBaT) L.poses[k] = *BA_P(gposes + k);
    if (tid < 7) L.ex[tid] = *BA_P(gex + tid);
    for (int k = tid; k < c.n_pairs; k += kBaT) L.pair_ij[k] = *BA_P(B.pair_ij + c.pp0 + k);
    for (int k = tid; k <= c.n_pairs; k += kBaT) { L.pair_slot[k] = (short)*BA_P(B.pair_slot + c.pp0 + w + k); L.pair_seg[k] = (short)*BA_P(B.pair_seg + c.pp0 + w + k); }
    if (!kBig) for (int k = tid; k < c.n_seg; k += kBaT) L.seg[k] = *BA_P(B.seg_tab + c.sg0 + k);
    __syncthreads();
    ba_setup<kBig>(B, c, L);
    if (kCl && c.rank > 0) { ba_follow<kBig>(B, c, L, G, pairdat); return; }

    const double function_tol = 1e-6, gradient_tol = 1e-10, parameter_tol = 1e-8, min_rel_decrease = 1e-3;
    const double min_diag = 1e-6, max_diag = 1e32, max_radius = 1e16, min_radius = 1e-32;
    const double min_mu = 1e-8, max_mu = 1.0, mu_inc = 10.0;
    double radius = 1e4, mu = min_mu, mu_used = min_mu, alpha = 0.0, dogleg_norm = 0.0;
    bool reuse = false;
    int invalid = 0, iter = 0, termination = 1, n_succ = 0, n_unsucc = 0;

    double x_cost = 0.0, initial_cost = 0.0, x_norm = 0.0;
    bool linearise = true, first = true, records_valid = false;
    BA_TICK(9)
    // one call site per phase (the phases are inlined: LDS addressing, no register-file round trips through a call)
    for (;;) {
        if (linearise) {
            x_cost = ba_evaluate<true, kCl, kBig>(B, c, L, G, L.poses, L.ex, ginvd, hpd, pairdat, records_valid);
            records_valid = false;
            BA_TICK(16)
            double q = 0, g = 0;
            if (c.ex_off >= 0 && tid < 7) q += L.ex[tid] * L.ex[tid];
            for (int k = tid; k < 7 * c.n_poses; k += kBaT) q += L.poses[k] * L.poses[k];
            for (int f = tid; f < F; f += kBaT) { { const double xv = kBig ? gld(ginvd + f) : L.vinv[f]; q += xv * xv; } g = fmax(g, fabs(BA_F(gdd, f))); }
            for (int a = tid; a < P; a += kBaT) g = fmax(g, fabs(L.gp[a]));
            x_norm = sqrt(block_sum(q, L.red));
            const double gmax = block_max(g, L.red);
            if (first) {
                initial_cost = x_cost;
                // Jacobi scaling from the first linearisation (ceres jacobi_scaling)
                for (int a = tid; a < P; a += kBaT) BA_V(scale, a) = 1.0 / (1.0 + sqrt(L.Hpp[a * kBaP + a]));
                for (int f = tid; f < F; f += kBaT) BA_V(scale, P + f) = 1.0 / (1.0 + sqrt(BA_F(Hdd, f)));
                first = false;
            }
            linearise = false; reuse = false;
            if (gmax <= gradient_tol) { termination = 0; break; }
        }
        if (radius <= min_radius) { termination = 0; break; }
        if (iter >= B.max_iter) break;
        iter++;
        bool ok = true;
        double model_change = 0.0;
        if (!reuse) {
            __syncthreads();
            for (int k = tid; k < N; k += kBaT) {
                const double h = k < P ? L.Hpp[k * kBaP + k] : BA_F(Hdd, k - P);
                const double g = k < P ? L.gp[k] : BA_F(gdd, k - P);
                BA_V(gs, k) = g * BA_V(scale, k);
                double d = h * BA_V(scale, k) * BA_V(scale, k);
                d = d < min_diag ? min_diag : (d > max_diag ? max_diag : d);
                BA_V(D, k) = sqrt(d);
                BA_V(va, k) = BA_V(gs, k) / d;
            }
            BA_TOCK(16)
            ba_hs_mul<kBig>(c, L, G, hpd);      // vb = Hs va = Hs (gs / D^2): kept until the next linearisation
            BA_TICK(17)
            double g2 = 0, jg2 = 0, zero = 0;
            for (int k = tid; k < N; k += kBaT) { const double gd = BA_V(gs, k) / BA_V(D, k); g2 += gd * gd; jg2 += BA_V(va, k) * BA_V(vb, k); }
            block_sum3(g2, jg2, zero, L.red);
            alpha = g2 / jg2;
            ok = false;
            BA_TOCK(17)
            while (mu < max_mu) {
                if (ba_schur_solve<kBig>(c, L, G, hpd, mu)) { ok = true; break; }
                mu *= mu_inc;
            }
            if (ok) {
                mu_used = mu;
                mu = fmax(min_mu, 2.0 * mu / mu_inc);
                for (int k = tid; k < N; k += kBaT) BA_V(gn, k) *= -BA_V(D, k);
            }
            __syncthreads();
        }
        BA_TICK(18)
        if (ok) {
            double a2 = 0, b2 = 0, ab = 0;
            for (int k = tid; k < N; k += kBaT) { const double gd = BA_V(gs, k) / BA_V(D, k); a2 += BA_V(gn, k) * BA_V(gn, k); b2 += gd * gd; ab += gd * BA_V(gn, k); }
            block_sum3(a2, b2, ab, L.red);
            const double gn_norm = sqrt(a2), g_norm = sqrt(b2);
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
            // step = ca gs / D^2 - cb x with (Hs + mu D2) x = gs (x = -gn / D), hence
            // Hs step = ca vb - cb (gs + mu D2 gn / D): the model decrease needs no second product with Hs
            double dg = 0, dHd = 0, zero = 0;
            for (int k = tid; k < N; k += kBaT) {
                const double st = (ca * (BA_V(gs, k) / BA_V(D, k)) + cb * BA_V(gn, k)) / BA_V(D, k);
                const double hs = ca * BA_V(vb, k) - cb * (BA_V(gs, k) + mu_used * BA_V(D, k) * BA_V(gn, k));
                BA_V(va, k) = st;
                dg += st * BA_V(gs, k); dHd += st * hs;
            }
            block_sum3(dg, dHd, zero, L.red);
            model_change = -(dg + 0.5 * dHd);
        }
        if (!ok || !(model_change > 0.0)) {
            BA_TOCK(18)
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
                for (int a = 0; a < 6; a++) d6[a] = BA_V(va, off + a) * BA_V(scale, off + a);
                ba::pose_plus(is_ex ? L.ex : L.poses + 7 * tid, d6, is_ex ? L.cex : L.cposes + 7 * tid);
            } else {
                for (int k = 0; k < 7; k++) L.cex[k] = L.ex[k];
            }
        }
        for (int f = tid; f < F; f += kBaT) *BA_P(cinvd + f) = *BA_P(ginvd + f) + BA_V(va, P + f) * BA_V(scale, P + f);
        __syncthreads();
        BA_TOCK(18)
        const double cand_cost = ba_evaluate<false, kCl, kBig>(B, c, L, G, L.cposes, L.cex, cinvd, hpd, pairdat);
        BA_TICK(19)
        double dq = 0;
        if (c.ex_off >= 0 && tid < 7) dq += (L.ex[tid] - L.cex[tid]) * (L.ex[tid] - L.cex[tid]);
        for (int k = tid; k < 7 * c.n_poses; k += kBaT) dq += (L.poses[k] - L.cposes[k]) * (L.poses[k] - L.cposes[k]);
        for (int f = tid; f < F; f += kBaT) { const double dv = *BA_P(ginvd + f) - *BA_P(cinvd + f); dq += dv * dv; }
        const double sn = sqrt(block_sum(dq, L.red));
        if (sn <= parameter_tol * (x_norm + parameter_tol)) { termination = 0; break; }
        if (fabs(x_cost - cand_cost) <= function_tol * x_cost) { termination = 0; break; }
        const double rel = (x_cost - cand_cost) / model_change;
        if (rel > min_rel_decrease) {
            __syncthreads();
            for (int k = tid; k < 7 * c.n_poses; k += kBaT) L.poses[k] = L.cposes[k];
            if (tid < 7) L.ex[tid] = L.cex[tid];
            for (int f = tid; f < F; f += kBaT) *BA_P(ginvd + f) = *BA_P(cinvd + f);
            __syncthreads();
            n_succ++;
            records_valid = true;            // the candidate's records are the accepted state's
            if (rel < 0.25) radius *= 0.5;
            if (rel > 0.75) radius = fmax(radius, 3.0 * dogleg_norm);
            if (radius > max_radius) radius = max_radius;
            linearise = true;
        } else {
            radius *= 0.5; reuse = true;
            n_unsucc++;
        }
        BA_TOCK(19)
    }
    __syncthreads();
    if (kCl) ba_publish(B, c, L, 3, L.poses, L.ex, ginvd, true);        // the followers leave
    BA_TOCK(9)
#ifdef LMONO_BA_PROF
    if (w == 0 && tid < 24) g_prof[tid] += (long long)L.prof[tid];
    __syncthreads();
    if (w == 0 && tid == 0) printf("PROF total %lld | lin: prologue %lld eval %lld mfma %lld | cost %lld | hs %lld | schur: stage+mfma %lld chol %lld subst %lld depth %lld | iters %d | wave-0 turn wait %lld, feature pass + small factors %lld (reduce pairs %lld, wave 0's features %lld) | chol: factor %lld barrier %lld next-panel %lld | glue: norms+D %lld alpha %lld dogleg+candidate %lld decision %lld\n", g_prof[9], g_prof[0], g_prof[1], g_prof[2], g_prof[3], g_prof[4], g_prof[5], g_prof[6], g_prof[7], g_prof[8], iter, g_prof[10] * 0, g_prof[11], g_prof[12], g_prof[13], g_prof[14], g_prof[15], g_prof[10], g_prof[16], g_prof[17], g_prof[18], g_prof[19]);
#endif
    for (int k = tid; k < c.n_poses * 7; k += kBaT) *BA_P(gposes + k) = L.poses[k];
    if (tid < 7) *BA_P(gex + tid) = L.ex[tid];
    if (tid == 0) {
        double *sm = BA_P(B.summary + (size_t)w * 6 + 5) - 5;
        sm[0] = initial_cost; sm[1] = x_cost; sm[2] = iter; sm[3] = (kCl && L.failed) ? 3 : termination; sm[4] = n_succ; sm[5] = n_unsucc;
    }
}

} // namespace lmono

#pragma clang fp contract(off)
