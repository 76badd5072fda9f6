// lmono_amd/csrc/odometry.hip -- gfx950 kernels of scan-to-scan LiDAR odometry (A-LOAM laserOdometry main
// loop + lidarFactor.hpp + the ceres::Solve call it makes; source absent from the reference tree,
// behavioural spec SURVEY.md Appendix A.2/A.3/B).
//
//   k_grid_build  one workgroup per (scan, table page): 1 m hash grid over less_sharp / less_flat (the "last" clouds
//                 of the next scan), built page by page in LDS.  Exact 1-NN needs only neighbours closer than 5 m
//                 (DISTANCE_SQ_THRESHOLD).
//   k_line_index  one workgroup per (scan, cloud): copy of the cloud counting-sorted by (scan line, azimuth bin)
//   k_correspond  32 lanes per feature point: de-skew transform (fp64), exact 1-NN over the 27 cells around the point
//                 (candidates as u64 keys, four cells per round in 8-lane sub-groups; arc sweep of the line index for the
//                 rare far cases), then the reference's scan-line walk restricted to the lines ra-2..ra+2 and the azimuth
//                 arc within 5 m (the five lines side by side in 6-lane sub-groups)
//   k_lm_solve    one workgroup per chain: <= 4 Levenberg-Marquardt iterations restating Ceres' trust-region loop,
//                 closed-form edge/plane Jacobians, reduction into the 6x6 normal equations (fp64)
//   k_pose_prefix sequential pose accumulation
#include "batch.hpp"

namespace lmono {

constexpr float kCell = 1.0f;          // grid edge (m)
constexpr float kInvCell = 1.0f;
constexpr int kMaxShell = 6;           // shells 1..5 cover 5 m up to float slop; shell 6 makes d2 < 25 exact
constexpr int kCoordOff = 1 << 20;
// k_correspond moves a run of points between lanes as ONE register: start (17 bits: a cloud holds < 2^17 points) | count << 17
constexpr int kRunMaxCount = 32767;
__device__ __forceinline__ int pack_run(int start, int count) { return start | (count << 17); }

__device__ __forceinline__ unsigned long long cell_key(int cx, int cy, int cz)
{
    return ((unsigned long long)(unsigned int)(cx + kCoordOff) << 42) |
           ((unsigned long long)(unsigned int)(cy + kCoordOff) << 21) |
           (unsigned long long)(unsigned int)(cz + kCoordOff);
}
__device__ __forceinline__ unsigned int hash_key(unsigned long long k)
{
    // three 32-bit multiplies on the packed 21-bit cell coordinates (Teschner et al. primes) and a final fold
    const unsigned int cz = (unsigned int)(k & 0x1fffffull), cy = (unsigned int)((k >> 21) & 0x1fffffull), cx = (unsigned int)(k >> 42);
    unsigned int h = (cx * 73856093u) ^ (cy * 19349663u) ^ (cz * 83492791u);
    h ^= h >> 15;
    return h;
}

struct GridRef {
    const GridCell *cell;
    const float4 *pts;
    int mask;
};

// ------------------------------------------------------------------------------------------------
// k_grid_build: 1 m hash grid of a "last" cloud, built page by page in LDS.
//
// The table of T = 2^k > n slots is cut into pages of kGridPage slots; a cell lives in the page its hash selects
// (initial slot = hash & (T - 1)) and linear probing wraps INSIDE the page (grid_next), so a page is a closed open-addressing
// table that fits LDS: keys (32-bit cell code) and one counter per slot (count, then write cursor), 128 KB.  One workgroup builds one page at a time:
//   pass 1  every point of the cloud: cell, hash, page; points of other pages are only counted (lower pages = this page's first
//           output position); runs of consecutive points in the same cell (feature clouds are ring / voxel ordered) are
//           merged by shift + ballot and their head lane does ONE LDS atomicCAS on the key and ONE counted LDS atomicAdd
//   scan    exclusive prefix of the page's counts, in place -> cursors
//   pass 2  the page's points again: read-only probe, counted atomicAdd on the cursor, coalesced-source scatter of the
//           cell-sorted copy (.w = index << 7 | line); the order of the points inside a cell is arbitrary, the searches
//           take minima over (distance, index) keys
//   write   the finished page streams out as 16-B GridCell records (every slot written: no clear pass over the table);
//           after pass 2 a cursor is the end of its cell = the start of the next slot's, so start and count follow from neighbours
// HBM sees the cloud (read), the copy and the table (written once); the random read-modify-write traffic of the old
// global-memory table (8.6 MB per scan) stays in LDS.  Surf pages of one scan are built by kGridPar workgroups in parallel.
constexpr int kGridPage = 16384;
constexpr int kGridPar = 2;
constexpr int kGridLds = kGridPage * 8;
constexpr unsigned int kEmptyKey32 = 0xffffffffu;

__device__ __forceinline__ unsigned int grid_next(unsigned int sl, unsigned int pmask) { return (sl & ~pmask) | ((sl + 1) & pmask); }

// 32-bit cell code of the LDS table: 11 + 11 + 10 bits (|cx|, |cy| < 1024 cells, -512 <= cz < 511); never kEmptyKey32
__device__ __forceinline__ bool cell_key32(int cx, int cy, int cz, unsigned int &k)
{
    const unsigned int ux = (unsigned int)(cx + 1024), uy = (unsigned int)(cy + 1024), uz = (unsigned int)(cz + 512);
    k = ux | (uy << 11) | (uz << 22);
    return ux < 2048u && uy < 2048u && uz < 1023u;
}
// hash_key(cell_key(cx, cy, cz)) without building the 64-bit key (coordinates inside the 32-bit code's range)
__device__ __forceinline__ unsigned int hash_cell(int cx, int cy, int cz)
{
    unsigned int h = ((unsigned int)(cx + kCoordOff) * 73856093u) ^ ((unsigned int)(cy + kCoordOff) * 19349663u) ^ ((unsigned int)(cz + kCoordOff) * 83492791u);
    h ^= h >> 15;
    return h;
}
__device__ __forceinline__ unsigned long long key32_to_64(unsigned int k)
{
    return cell_key((int)(k & 2047u) - 1024, (int)((k >> 11) & 2047u) - 1024, (int)(k >> 22) - 512);
}

#ifdef LMONO_DIAG_SEARCH
__global__ __launch_bounds__(1024) void k_grid_build(BatchView b)
{
    const int s = b.scan0 + blockIdx.x;
    const bool surf = blockIdx.y > 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = b.feat_n[s * 4 + (surf ? 3 : 1)];
    const int Tcap = surf ? kSurfTable : kCornerTable;
    // occupied slots = distinct 1 m cells, typically n / 3: a table of >= n + 1 slots keeps the load near 0.3
    int T = next_pow2(n + 1);
    if (T < 1024) T = 1024;
    GridCell *cell = surf ? b.sg_cell + (size_t)s * kSurfTable : b.cg_cell + (size_t)s * kCornerTable;
    const int p0 = surf ? (int)blockIdx.y - 1 : 0, pstep = surf ? kGridPar : 1;
    if (T > Tcap) {
        // the cloud does not fit its table: empty table, no correspondences for the next scan, flagged in status
        if (p0 == 0) {
            for (int i = tid; i < 1024; i += 1024) { GridCell e; e.key = kEmptyKey; e.start = 0; e.cnt = 0; cell[i] = e; }
            if (tid == 0) { b.grid_mask[s * 2 + (surf ? 1 : 0)] = 1023; atomicOr(&b.status[s], kStatusGridOverflow); }
        }
        return;
    }
    const int P = T < kGridPage ? T : kGridPage, n_pages = T / P, pshift = __ffs(P) - 1;
    if (p0 >= n_pages) return;
    if (tid == 0 && p0 == 0) b.grid_mask[s * 2 + (surf ? 1 : 0)] = T - 1;
    const float4 *src = surf ? b.less_flat + b.off[s] : b.less_sharp + (size_t)s * kMaxLessSharp;
    float4 *dst = surf ? b.sg_pts + b.off[s] : b.cg_pts + (size_t)s * kMaxLessSharp;
    extern __shared__ __align__(16) unsigned int g_lds[];
    unsigned int *keys = g_lds;
    int *cnt = (int *)(g_lds + kGridPage);      // pass 1: points per cell; after the scan: write cursor
    __shared__ int s_wsum[16], s_before, s_fail, s_occ;
    const unsigned int pm = (unsigned int)(P - 1);

    for (int p = p0; p < n_pages; p += pstep) {
        for (int i = tid; i < P; i += 1024) { keys[i] = kEmptyKey32; cnt[i] = 0; }
        if (tid == 0) { s_before = 0; s_fail = 0; s_occ = 0; }
        __syncthreads();
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {       // unrolled: two specialised copies (insert / look up and scatter)
            int before = 0;
            bool fail = false;
            // 256 consecutive points per wave and round (lane l owns points t0 + 64 q + l, q = 0..3): four independent chains
            float4 nxt[4];     // the next round's points are requested before this round's atomics are waited for
#pragma unroll
            for (int q = 0; q < 4; q++) { const int i = wave * 256 + 64 * q + lane; nxt[q] = i < n ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f); }
            for (int t0 = wave * 256; t0 < n; t0 += 16 * 256) {
                unsigned int k32[4], sl[4];
                float4 pt[4];
                bool mine[4], head[4];
                int hl[4], len[4], base[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    pt[q] = nxt[q];
                    const int i = t0 + 16 * 256 + 64 * q + lane;
                    nxt[q] = i < n ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int i = t0 + 64 * q + lane;
                    const int cx = (int)floorf(pt[q].x * kInvCell), cy = (int)floorf(pt[q].y * kInvCell), cz = (int)floorf(pt[q].z * kInvCell);
                    const bool ok = cell_key32(cx, cy, cz, k32[q]);
                    const unsigned int h = hash_cell(cx, cy, cz) & (unsigned int)(T - 1);
                    const int pg = (int)(h >> pshift);
                    sl[q] = h & pm;
                    mine[q] = i < n && pg == p;
                    if (i < n && pg < p) before++;
                    if (mine[q] && !ok) { fail = true; mine[q] = false; }
                }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const unsigned int prev = (unsigned int)__shfl_up((int)k32[q], 1);
                    const unsigned long long mm = __ballot(mine[q]);
                    head[q] = mine[q] && (lane == 0 || !((mm >> (lane - 1)) & 1ull) || k32[q] != prev);
                    const unsigned long long hm = __ballot(head[q]);
                    hl[q] = 63 - __clzll((long long)(hm & ((2ull << lane) - 1ull)));          // head lane of this lane's run
                    const unsigned long long stop = hm | ~mm;                                  // a run ends before the next head / foreign point
                    const unsigned long long rest = lane < 63 ? stop >> (lane + 1) : 0ull;
                    len[q] = rest ? __ffsll((long long)rest) : 64 - lane;                      // run length (meaningful on head lanes)
                }
                // first probe of the four sub-tiles back to back (four LDS operations in flight), then the (rare) collisions;
                // pass 0 inserts, pass 1 only looks the slot up
                unsigned int o4[4];
#pragma unroll
                for (int q = 0; q < 4; q++)
                    o4[q] = head[q] ? (pass == 0 ? atomicCAS(&keys[sl[q]], kEmptyKey32, k32[q]) : keys[sl[q]]) : k32[q];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if (head[q]) {
                        unsigned int o = o4[q];
                        int tries = 0;
                        while (!(o == k32[q] || (pass == 0 && o == kEmptyKey32))) {
                            if (++tries >= P || (pass == 1 && o == kEmptyKey32)) { fail = true; head[q] = false; break; }
                            sl[q] = (sl[q] + 1) & pm;
                            o = pass == 0 ? atomicCAS(&keys[sl[q]], kEmptyKey32, k32[q]) : keys[sl[q]];
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; q++) base[q] = head[q] ? atomicAdd(&cnt[sl[q]], len[q]) : -1;
                if (pass == 1) {
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int i = t0 + 64 * q + lane;
                        const int h = max(hl[q], 0);
                        const int b_run = __shfl(base[q], h);
                        if (mine[q] && b_run >= 0) {
                            const int ln = (int)pt[q].w;
                            dst[b_run + (lane - h)] = make_float4(pt[q].x, pt[q].y, pt[q].z, __int_as_float((i << 7) | (ln < 0 ? 0 : (ln > 65 ? 65 : ln))));
                        }
                    }
                }
            }
            if (pass == 0) {
                before = wave_sum_i(before);
                if (lane == 0 && before) atomicAdd(&s_before, before);
            }
            if (fail) s_fail = 1;
            __syncthreads();
            if (pass == 0) {
                // exclusive prefix of the page's counts -> cursors (first output position of every cell)
                const int per = P >> 10;                       // 1 .. 16 consecutive slots per thread
                int c[16], sum = 0;
                int occ = 0;
#pragma unroll
                for (int q = 0; q < 16; q++) { c[q] = q < per ? cnt[tid * per + q] : 0; sum += c[q]; occ += c[q] > 0 ? 1 : 0; }
                const int incl = wave_scan_incl(sum);
                if (lane == 63) s_wsum[wave] = incl;
                // a page without a single empty slot would never end the probe sequence of an absent cell: treated as full
                occ = wave_sum_i(occ);
                if (lane == 0 && occ) atomicAdd(&s_occ, occ);
                __syncthreads();
                if (s_occ >= P) s_fail = 1;
                int run = s_before + incl - sum;
                for (int w = 0; w < wave; w++) run += s_wsum[w];
#pragma unroll
                for (int q = 0; q < 16; q++) { if (q < per) { cnt[tid * per + q] = run; run += c[q]; } }
                __syncthreads();
            }
        }
        // the finished page streams out; a failed page (cell code out of range, page full) is written empty and flagged
        const bool bad = s_fail != 0;
        bool dense = false;
        for (int j = tid; j < P; j += 1024) {
            const unsigned int k = keys[j];
            const int end = cnt[j], start = j > 0 ? cnt[j - 1] : s_before;
            const int c = end - start;
            GridCell e;
            e.key = (k == kEmptyKey32 || bad) ? kEmptyKey : key32_to_64(k);
            e.cnt = bad ? 0 : c;
            e.start = bad ? 0 : start;
            cell[(size_t)p * P + j] = e;
            dense = dense || c > kRunMaxCount;
        }
        if (tid == 0 && bad) atomicOr(&b.status[s], kStatusGridOverflow);
        if (dense) atomicOr(&b.status[s], kStatusDenseCell);
        __syncthreads();
    }
}
#endif // LMONO_DIAG_SEARCH
// ------------------------------------------------------------------------------------------------
struct OdomView {
    int n_scans, n_chains, lead;
    int first;            // first owned scan of the batch (> 0: the scans before it are the lead-in of chain 0, results discarded)
    int chain0, chain1;   // the chains this launch advances: [chain0, chain1) (groups of chains run on their own streams)
    int fixed_k;          // >= 0: single-pair debug view, every chain works on this scan
    int lead_full;        // >= 0: only the last lead_full lead-in scan pairs of a chain use all feature points, the earlier ones every
                          // kThinStride-th workgroup's share (their result is only the next pair's warm start); -1: all pairs use all
    double *state;        // [n_chains][8]  q(xyzw), t, pad
    int *corr;            // [n_chains][kMaxQueries][4]
    float4 *crec;         // [n_chains][kMaxQueries][4] residual-block records: (cp, kind), a, b, c
    double *incr;         // [n_scans][7]
    int *lm_info;         // [n_chains][4]
    int *seed;            // [n_chains][kMaxQueries] nearest point found by the previous outer iteration of the same scan pair (-1: none)
    // ---- boundary validation of the chained schedule (k_boundary_check, repair launches; lmono_hip.hip: odom_run)
    double *ws;           // [n_chains][8] the warm start chain c used for its first owned scan pair: the result of its last lead-in pair
                          // (identity without a lead-in); after a repair, the increment it was re-started from
    int repair;           // 1: repair launch -- chain slot c re-runs its OWN pairs from scan s_c on (no lead-in: state = incr[s_c - 1]),
                          // compares every new increment with the stored one and stops at the first that agrees within tol
    int step0;            // repair: pairs of the chain already re-run by earlier launches of this repair round
    const int *clist;     // repair: the launch's chains are clist[chain0 .. chain1) (flagged boundaries); nullptr: chains chain0 .. chain1
    int *rstat;           // [n_chains][4] repair state: [0] stopped (agreement or end of chain), [1] pairs re-run in total, [2] flagged in the current round, [3] times flagged
    unsigned int *rcount; // [0] chains flagged by k_boundary_check, [1] repair chains still running, then the flagged chain ids (clist)
    double tol;           // agreement bound of boundary_residual()
};

// scans [first, n_scans) are cut into n_chains ranges; scans before `first` are an external lead-in (the previous rank's scans)
__device__ __host__ __forceinline__ void chain_bounds(int first, int n_scans, int n_chains, int c, int &s, int &e)
{
    s = first + (int)((long long)c * (n_scans - first) / n_chains);
    e = first + (int)((long long)(c + 1) * (n_scans - first) / n_chains);
}

__device__ __forceinline__ void quat_rotate(const double *q, double vx, double vy, double vz, double &ox, double &oy, double &oz)
{
    const double ux = q[0], uy = q[1], uz = q[2], w = q[3];
    const double uvx = 2.0 * (uy * vz - uz * vy);
    const double uvy = 2.0 * (uz * vx - ux * vz);
    const double uvz = 2.0 * (ux * vy - uy * vx);
    ox = vx + w * uvx + (uy * uvz - uz * uvy);
    oy = vy + w * uvy + (uz * uvx - ux * uvz);
    oz = vz + w * uvz + (ux * uvy - uy * uvx);
}

// exact 1-NN of (qx,qy,qz) in a hash-gridded cloud; all 64 lanes cooperate.  Returns packed (d2 bits, index),
// ~0 when the cloud holds no point closer than kMaxShell cells.  Ties resolve to the lowest original index.
__device__ __forceinline__ unsigned long long wave_nn(const GridRef &g, float qx, float qy, float qz, int lane)
{
    const int cqx = (int)floorf(qx * kInvCell), cqy = (int)floorf(qy * kInvCell), cqz = (int)floorf(qz * kInvCell);
    unsigned long long best = ~0ull;
    for (int sh = 1; sh <= kMaxShell; sh++) {
        const int side = 2 * sh + 1, ncell = side * side * side;
        for (int base = 0; base < ncell; base += 64) {
            // every lane probes one cell of the shell (one 16-B load per probe step) ...
            const int ci = base + lane;
            int st = 0, cn = 0;
            if (ci < ncell) {
                const int dx = ci % side - sh, dy = (ci / side) % side - sh, dz = ci / (side * side) - sh;
                if (sh == 1 || max(max(abs(dx), abs(dy)), abs(dz)) == sh) {
                    const unsigned long long k = cell_key(cqx + dx, cqy + dy, cqz + dz);
                    unsigned int sl = hash_key(k) & g.mask;
                    const unsigned int pmask = min((unsigned int)g.mask, (unsigned int)(kGridPage - 1));
                    while (true) {
                        const GridCell e = g.cell[sl];
                        if (e.key == k) { st = e.start; cn = e.cnt; break; }
                        if (e.key == kEmptyKey) break;
                        sl = grid_next(sl, pmask);
                    }
                }
            }
            // ... then the whole wave sweeps each non-empty cell's contiguous point run
            unsigned long long m = __ballot(cn > 0);
            while (m) {
                const int src = __ffsll((long long)m) - 1;
                m &= m - 1;
                const int s0 = __shfl(st, src), n0 = __shfl(cn, src);
                for (int i = lane; i < n0; i += 64) {
                    const float4 p = g.pts[s0 + i];
                    const unsigned long long cand = pack_fu(dist2f(p.x, p.y, p.z, qx, qy, qz), (unsigned int)__float_as_int(p.w) >> 7);
                    best = cand < best ? cand : best;
                }
            }
        }
        best = wave_min_u64(best);
        if (best != ~0ull) {
            const float bound = (float)sh * kCell * 0.9999f;
            if (__uint_as_float((unsigned int)(best >> 32)) <= bound * bound) break;
        }
    }
    return best;
}

// the same without a grid: every lane sweeps the whole cloud (only the generic path of flagged scans can get here, and only when the
// grids were not built because the default search does not use them)
__device__ __forceinline__ unsigned long long wave_nn_brute(const float4 *cloud, int n, float qx, float qy, float qz, int lane)
{
    unsigned long long best = ~0ull;
    for (int i = lane; i < n; i += 64) {
        const float4 p = cloud[i];
        const unsigned long long cand = pack_fu(dist2f(p.x, p.y, p.z, qx, qy, qz), (unsigned int)i);
        best = cand < best ? cand : best;
    }
    return wave_min_u64(best);
}

constexpr unsigned int kSeqBack = 1u << 24;   // backward candidates rank after every forward candidate

// Wave-parallel restatement of the reference's two index walks around the nearest point `closest` of a feature
// cloud whose int(intensity) is the scan line.  Forward: j = closest+1.. until the first line > ra + 2.5; backward:
// j = closest-1.. until the first line < ra - 2.5.  Points on the near side of the closest line ("same": line <= ra
// going forward, >= ra going backward) and on the far side ("other") are minimised separately; edges use only
// "other".  Candidates are ordered like the reference visits them (forward ascending, then backward descending) so
// that exact distance ties resolve identically.  No monotonicity of the line ids is assumed.
__device__ __forceinline__ void line_walk(const float4 *cloud, int n_last, int closest, int ra, float qx, float qy, float qz,
                                          int lane, unsigned long long &same, unsigned long long &other)
{
    for (int base = closest + 1; base < n_last; base += 64) {
        const int j = base + lane;
        float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
        int v = INT_MAX;
        if (j < n_last) { c = cloud[j]; v = (int)c.w; }
        const bool brk = (j < n_last) && ((double)v > (double)ra + 2.5);
        const unsigned long long mb = __ballot(brk);
        const int stop = mb ? (__ffsll((long long)mb) - 1) : 64;
        if (j < n_last && lane < stop) {
            const unsigned long long cand = pack_fu(dist2f(c.x, c.y, c.z, qx, qy, qz), (unsigned int)(j - closest - 1));
            if (v <= ra) same = cand < same ? cand : same;
            else other = cand < other ? cand : other;
        }
        if (mb) break;
    }
    for (int base = closest - 1; base >= 0; base -= 64) {
        const int j = base - lane;
        float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
        int v = INT_MIN;
        if (j >= 0) { c = cloud[j]; v = (int)c.w; }
        const bool brk = (j >= 0) && ((double)v < (double)ra - 2.5);
        const unsigned long long mb = __ballot(brk);
        const int stop = mb ? (__ffsll((long long)mb) - 1) : 64;
        if (j >= 0 && lane < stop) {
            const unsigned long long cand = pack_fu(dist2f(c.x, c.y, c.z, qx, qy, qz), kSeqBack + (unsigned int)(closest - 1 - j));
            if (v >= ra) same = cand < same ? cand : same;
            else other = cand < other ? cand : other;
        }
        if (mb) break;
    }
    same = wave_min_u64(same);
    other = wave_min_u64(other);
}

__device__ __forceinline__ int seq_to_index(unsigned int seq, int closest)
{
    return seq >= kSeqBack ? (closest - 1 - (int)(seq - kSeqBack)) : closest + 1 + (int)seq;
}

// Correspondence search for one feature point of scan k against scan k-1.  kind 1 = edge (a, b), 2 = plane (a, b, c).
__device__ __forceinline__ int4 correspond_one(const BatchView &b, int k, int qi, const double *x, int lane)
{
    const int n_sharp = b.feat_n[k * 4 + 0];
    const bool edge = qi < n_sharp;
    const float4 p = edge ? b.sharp[(size_t)k * kMaxSharp + qi] : b.flat[(size_t)k * kMaxFlat + (qi - n_sharp)];
    double rx, ry, rz;
    quat_rotate(x, (double)p.x, (double)p.y, (double)p.z, rx, ry, rz);
    const float qx = (float)(rx + x[4]), qy = (float)(ry + x[5]), qz = (float)(rz + x[6]);
    const int l = k - 1;
    GridRef g;
    const float4 *cloud;
    int n_last;
    if (edge) {
        g.cell = b.cg_cell + (size_t)l * kCornerTable; g.pts = b.cg_pts + (size_t)l * kMaxLessSharp; g.mask = b.grid_mask[l * 2 + 0];
        cloud = b.less_sharp + (size_t)l * kMaxLessSharp; n_last = b.feat_n[l * 4 + 1];
    } else {
        g.cell = b.sg_cell + (size_t)l * kSurfTable; g.pts = b.sg_pts + b.off[l]; g.mask = b.grid_mask[l * 2 + 1];
        cloud = b.less_flat + b.off[l]; n_last = b.feat_n[l * 4 + 3];
    }
    int4 out = make_int4(-1, -1, -1, 0);
    if (n_last == 0) return out;
    const unsigned long long nn = b.has_grid ? wave_nn(g, qx, qy, qz, lane) : wave_nn_brute(cloud, n_last, qx, qy, qz, lane);
    if (nn == ~0ull) return out;
    const float d2 = __uint_as_float((unsigned int)(nn >> 32));
    if (!((double)d2 < 25.0)) return out;
    const int closest = (int)(unsigned int)(nn & 0xffffffffull);
    const int ra = (int)cloud[closest].w;
    const unsigned long long thr = pack_fu(25.0f, 0u);   // candidates need d2 < 25
    unsigned long long same = thr, other = thr;
    line_walk(cloud, n_last, closest, ra, qx, qy, qz, lane, same, other);
    const int i_other = other < thr ? seq_to_index((unsigned int)(other & 0xffffffffull), closest) : -1;
    if (edge) {
        if (i_other >= 0) out = make_int4(closest, i_other, -1, 1);
        return out;
    }
    const int i_same = same < thr ? seq_to_index((unsigned int)(same & 0xffffffffull), closest) : -1;
    if (i_same >= 0 && i_other >= 0) out = make_int4(closest, i_same, i_other, 2);
    return out;
}

constexpr int kThinStride = 4;
static_assert(kMaxQueries <= 4096, "the work lists pack (chain << 12 | feature index)");

// lead-in scan pair k of a chain owning scans from own_begin on that runs on a thinned feature set
__device__ __forceinline__ bool lead_in_thinned(const OdomView &o, int k, int own_begin)
{
    return o.lead_full >= 0 && k < own_begin && own_begin - k > o.lead_full;
}

// which scan does chain c work on at this step (-1: chain finished)
__device__ __forceinline__ int chain_scan(const OdomView &o, int c, int step, int &own_begin)
{
    if (o.fixed_k >= 0) { own_begin = 0; return o.fixed_k; }
    int s, e;
    chain_bounds(o.first, o.n_scans, o.n_chains, c, s, e);
    own_begin = s;
    if (o.repair) {
        const int k = s + o.step0 + step;
        return (k < e && !o.rstat[c * 4]) ? k : -1;
    }
    const int begin = max(s - o.lead, 0);
    const int k = begin + 1 + step;
    return k < e ? k : -1;
}

constexpr int kRepairAgree = 2;      // consecutive scan pairs a repair chain must reproduce (within tol) before the rest of its chain stands

// Distance of two scan-pair increments (q xyzw, t): max(|dq_i| with the signs aligned, 0.1 |dt_i| / m) -- 1e-6 = 2e-6 rad, 1e-5 m.
__device__ __forceinline__ double boundary_residual(const double *a, const double *b)
{
    const double sg = (a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]) < 0.0 ? -1.0 : 1.0;
    double r = 0.0;
    for (int i = 0; i < 4; i++) r = fmax(r, fabs(a[i] - sg * b[i]));
    for (int i = 4; i < 7; i++) r = fmax(r, 0.1 * fabs(a[i] - b[i]));
    return r == r ? r : 1e300;            // a NaN never agrees
}

// ---- (line, azimuth-bin) index of a feature cloud -------------------------------------------------------------
// Every candidate the searches ever accept lies closer than 5 m to the (transformed) feature point, i.e. within +-asin(r / rho_xy)
// of its azimuth and within a few scan lines of its elevation.  k_line_index sorts a copy of every "last" cloud by (scan line,
// azimuth bin) once per scan (counting sort in LDS) and keeps the start of every (line, bin) bucket: the candidates of a ball are ONE
// contiguous run per scan line -- a handful of short coalesced sweeps instead of a pass over whole scan lines.  A point of the copy carries
// (cloud index << 7 | line) in .w, the low word of the searches' u64 keys.
// Round 4 measured the (bin, line)-major order beside this one (commit 9c24d9d holds both behind LMONO_LB_ORDER; profiles/r4/NOTES.md section 1):
// equal for the chunked search (0.250 against 0.236 ms per launch), 3.0 instead of 2.2 ms per pass to build (a ring's consecutive points land 66
// buckets apart: scattered 16-B writes).  Removed again.
constexpr int kAzBins = 384;           // 0.9375 deg: about one point of a 0.2 m-voxelised cloud per bin and line at 12 m
constexpr int kLineKeys = 66 * kAzBins;
__device__ __forceinline__ int lb_key(int bin, int line) { return line * kAzBins + bin; }
constexpr int kLbPad = 4;              // entries behind the index copies: k_corr_flat's 64-B chunks may read up to three points past a run
// the runs of the copy that hold the lines va .. vb of the bins [b_lo, b_lo + nbins) (wrapping past the last bin): f(first point, count).
// Fall-back kernels only: simple beats fast.
template <class F>
__device__ __forceinline__ void lb_for_runs(const int *table, int b_lo, int nbins, int va, int vb, F f)
{
    const int b_end = b_lo + nbins;
    for (int part = 0; part < 2; part++) {
        if (part == 1 && b_end <= kAzBins) break;
        const int p0 = part ? 0 : b_lo, p1 = part ? b_end - kAzBins : min(b_end, kAzBins);
        for (int v = va; v <= vb; v++) { const int s0 = table[v * kAzBins + p0]; f(s0, table[v * kAzBins + p1] - s0); }
    }
}

__device__ __forceinline__ int az_bin(float x, float y)
{
    const float t = (atan2f(y, x) + 3.14159265f) * (kAzBins / 6.28318531f);
    const int bi = (int)t;
    return bi < 0 ? 0 : (bi >= kAzBins ? kAzBins - 1 : bi);
}
__device__ __forceinline__ int line_of(float w)
{
    const int v = (int)w;
    return v < 0 ? 0 : (v > 65 ? 65 : v);
}
// elevation angle of a point as the index sees it (only ever compared with margins: no exactness requirement on atan2f)
__device__ __forceinline__ float elev_of(float x, float y, float z) { return atan2f(z, sqrtf(x * x + y * y)); }
// order-preserving float <-> int map for LDS atomicMin / atomicMax
__device__ __forceinline__ int f2ord(float f) { const int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ __forceinline__ float ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

constexpr int kLiT = 1024;     // threads of k_line_index
constexpr int kLiLdsHalf = kLineKeys * 2, kLiLdsFull = kLineKeys * 4;   // dynamic LDS: one 16-bit / 32-bit counter per (line, bin)
constexpr int kLiBigGrid = 64;

// One (scan, cloud): copy of the cloud counting-sorted by (line, azimuth bin), the (line, bin) start table, per-line elevation bounds.
// kHalf: the counters are 16-bit halves of 32-bit LDS words (50 KB instead of 101 KB: three workgroups per CU instead of one --
// the kernel is bound by its own dependent rounds, not by bytes); a cloud of more than 65535 points cannot be counted in 16 bits
// and goes through the work list `li_todo` to the full-width launch (small fixed grid, normally empty).
template <bool kHalf>
__device__ __forceinline__ void line_index_cloud(const BatchView &b, int s, bool surf, int n, int *s_cnt, int *s_wsum, int *s_emin, int *s_emax)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float4 *src = surf ? b.less_flat + b.off[s] : b.less_sharp + (size_t)s * kMaxLessSharp;
    float4 *dst = surf ? b.lbs_pts + b.off[s] : b.lbc_pts + (size_t)s * kMaxLessSharp;
    int *table = b.lb_start + (size_t)(s * 2 + (surf ? 1 : 0)) * (kLineKeys + 1);
    unsigned int *s_c32 = (unsigned int *)s_cnt;
    constexpr int kWords = kHalf ? kLineKeys / 2 : kLineKeys;
    static_assert(kLineKeys % 2 == 0, "two 16-bit counters per LDS word");
    // counter `key`: add one, return the old value
    auto bump = [&](int key) -> int {
        if (kHalf) { const int sh = (key & 1) * 16; return (int)((atomicAdd(&s_c32[key >> 1], 1u << sh) >> sh) & 0xffffu); }
        return atomicAdd(&s_cnt[key], 1);
    };
    for (int i = tid; i < kWords; i += kLiT) s_cnt[i] = 0;
    if (tid < 66) { s_emin[tid] = INT_MAX; s_emax[tid] = INT_MIN; }
    __syncthreads();
    // four points per thread and round, their loads in flight together.  A wave holds 64 consecutive points, which nearly always
    // share one scan line: their elevation bounds are reduced in the wave (DPP) and leave as ONE pair of LDS atomics
    for (int base = 0; base < n; base += 4 * kLiT) {      // uniform trip count: the wave reductions need every lane
        float4 p[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { const int i = base + tid + kLiT * q; p[q] = src[i < n ? i : 0]; }      // unconditional (clamped): really four in flight
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const bool ok = base + tid + kLiT * q < n;
            const int ln = line_of(p[q].w);
            if (ok) (void)bump(lb_key(az_bin(p[q].x, p[q].y), ln));
            const unsigned long long act = __ballot(ok);
            if (act == 0ull) continue;
            const int eo = f2ord(elev_of(p[q].x, p[q].y, p[q].z));
            const int ln0 = __shfl(ln, __ffsll((long long)act) - 1);
            if (__ballot(ok && ln != ln0) == 0ull) {
                const unsigned int lo = wave_min_u32_uniform(ok ? (unsigned int)eo ^ 0x80000000u : ~0u);
                const unsigned int hi = wave_max_u32_uniform(ok ? (unsigned int)eo ^ 0x80000000u : 0u);
                if (lane == 0) { atomicMin(&s_emin[ln0], (int)(lo ^ 0x80000000u)); atomicMax(&s_emax[ln0], (int)(hi ^ 0x80000000u)); }
            } else if (ok) {
                atomicMin(&s_emin[ln], eo);
                atomicMax(&s_emax[ln], eo);
            }
        }
    }
    __syncthreads();
    // per-line elevation bounds of the cloud (empty line: lo > hi), used by the searches to bound the lines a ball can meet:
    // (lo, hi, A = min of lo over lines <= v, B = max of hi over lines >= v).  A and B are monotone (non-increasing in v) whatever
    // the sensor, so the lines that can meet an elevation window [a, b] lie between the first v with A_v <= b and the last with B_v >= a
    if (tid < 66) {
        float4 *el = b.lb_elev + (size_t)(s * 2 + (surf ? 1 : 0)) * 66;
        float A = 1e30f, B = -1e30f;
        for (int v = 0; v <= tid; v++) if (s_emin[v] != INT_MAX) A = fminf(A, ord2f(s_emin[v]));
        for (int v = tid; v < 66; v++) if (s_emin[v] != INT_MAX) B = fmaxf(B, ord2f(s_emax[v]));
        el[tid] = s_emin[tid] == INT_MAX ? make_float4(1e30f, -1e30f, A, B) : make_float4(ord2f(s_emin[tid]), ord2f(s_emax[tid]), A, B);
    }
    // exclusive prefix over the kLineKeys counters: kPer consecutive counters per thread (kPer even: whole LDS words; last threads padded)
    constexpr int kPer = (((kLineKeys + kLiT - 1) / kLiT) + 1) & ~1;
    auto get = [&](int idx) -> int { return kHalf ? (int)((s_c32[idx >> 1] >> ((idx & 1) * 16)) & 0xffffu) : s_cnt[idx]; };
    int local = 0;
    for (int i = 0; i < kPer; i++) { const int idx = tid * kPer + i; if (idx < kLineKeys) local += get(idx); }
    const int incl = wave_scan_incl(local);
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    int run = incl - local;
    for (int w = 0; w < wave; w++) run += s_wsum[w];
    if (kHalf) {
        // a thread owns whole words (kPer is even): no other thread touches them here.  Cursors are < n <= 65535.
        for (int i = 0; i < kPer; i += 2) {
            const int idx = tid * kPer + i;
            if (idx < kLineKeys) {
                const unsigned int wv = s_c32[idx >> 1];
                const int c0 = (int)(wv & 0xffffu), c1 = (int)(wv >> 16);
                table[idx] = run; table[idx + 1] = run + c0;
                s_c32[idx >> 1] = (unsigned int)run | ((unsigned int)(run + c0) << 16);
                run += c0 + c1;
            }
        }
    } else {
        for (int i = 0; i < kPer; i++) {
            const int idx = tid * kPer + i;
            if (idx < kLineKeys) { const int cnt = s_cnt[idx]; s_cnt[idx] = run; table[idx] = run; run += cnt; }
        }
    }
    if (tid == kLiT - 1) table[kLineKeys] = n;
    __syncthreads();
    for (int i0 = tid; i0 < n; i0 += 4 * kLiT) {
        float4 p[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { const int i = i0 + kLiT * q; p[q] = src[i < n ? i : 0]; }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int i = i0 + kLiT * q;
            if (i < n) {
                const int ln = line_of(p[q].w);
                const int d = bump(lb_key(az_bin(p[q].x, p[q].y), ln));
                dst[d] = make_float4(p[q].x, p[q].y, p[q].z, __int_as_float((i << 7) | ln));
            }
        }
    }
}

// from_list = 0: grid (n_scans, 2), 16-bit counters, clouds of more than 65535 points deferred; 1: kLiBigGrid workgroups over the list
template <bool kHalf>
__global__ __launch_bounds__(kLiT) void k_line_index(BatchView b)
{
    extern __shared__ __align__(16) int s_cnt[];      // points per (line, bin); after the prefix: write cursor of the bucket
    __shared__ int s_wsum[kLiT / 64], s_emin[66], s_emax[66];
    if (kHalf) {
        const int s = b.scan0 + blockIdx.x;
        const bool surf = blockIdx.y == 1;
        if (b.feat_n[s * 4 + (surf ? 3 : 1)] > 65535) {
            if (threadIdx.x == 0) b.li_todo[1 + atomicAdd(&b.li_todo[0], 1)] = s * 2 + (surf ? 1 : 0);
            return;
        }
        line_index_cloud<true>(b, s, surf, b.feat_n[s * 4 + (surf ? 3 : 1)], s_cnt, s_wsum, s_emin, s_emax);
    } else {
        const int n_todo = b.li_todo[0];
        for (int k = blockIdx.x; k < n_todo; k += gridDim.x) {
            const int e = b.li_todo[1 + k];
            __syncthreads();                            // the previous cloud's scatter is done with the counters
            line_index_cloud<false>(b, e >> 1, (e & 1) != 0, b.feat_n[(e >> 1) * 4 + ((e & 1) ? 3 : 1)], s_cnt, s_wsum, s_emin, s_emax);
        }
    }
}

// k_compact and k_line_index<true> as ONE kernel (round 5, VERDICT r4 #2a): a workgroup compacts its scan's feature clouds and indexes the two "last"
// clouds right away, while they sit in this XCD's L2 -- the separate kernels wrote the clouds of all 4541 scans (1.2 GB) and read them back twice
// from HBM.  The 16-bit counters are used for one cloud after the other; a cloud of more than 65535 points goes to the full-width launch's list.
#ifndef LMONO_FUSE_COMPACT_INDEX
#define LMONO_FUSE_COMPACT_INDEX 1
#endif
__global__ __launch_bounds__(kLiT) void k_compact_index(BatchView b)
{
    extern __shared__ __align__(16) int s_cnt[];
    __shared__ int s_wsum[kLiT / 64], s_emin[66], s_emax[66];
    const int s = b.scan0 + blockIdx.x;
    int n_ls, n_lf;
    compact_scan<kLiT>(b, s, n_ls, n_lf);
    for (int cld = 0; cld < 2; cld++) {
        const int n = cld ? n_lf : n_ls;
        // this workgroup's own stores (the cloud) are read back below: they are out of the vector unit before anybody loads
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (n > 65535) { if (threadIdx.x == 0) b.li_todo[1 + atomicAdd(&b.li_todo[0], 1)] = s * 2 + cld; continue; }
        line_index_cloud<true>(b, s, cld != 0, n, s_cnt, s_wsum, s_emin, s_emax);
    }
}

// Slack of an azimuth arc in bins.  A point within r of the feature differs from it by at most asin(r / rho) in azimuth; both bins come
// from the same atan2f (error ~1e-6 rad = 6e-5 bins) and asin_upper bounds asin from above, so only float rounding needs cover.
// (1.5 bins until round 2: that alone made every arc 3 bins = 2.8 deg wider than the ball it serves.)
constexpr float kArcSlackBins = 0.05f;
// Radii of the scan-line walk's passes: a minimum found strictly inside the ball of a pass is final (every point outside the arc is
// farther).  0: the neighbouring lines right next to the nearest point (line spacing ~0.006-0.009 rho, voxel spacing 0.2 m);
// 1: 0.5 m + 5 % of the range; 2: the ring gap of far ground points (rho^2 dtheta / h); 3: DISTANCE_SQ_THRESHOLD.  Non-increasing
// entries are skipped by the callers.
// Radii of the walk's passes (a search strategy, not a result: any ascending ladder that ends at 5 m finds the same partners; scripts/ladder_sweep.sh
// re-measures variants with the index-exact tests in the loop).  Round 3: 0.2 + 0.012 rho / 0.5 + 0.05 rho -> 0.3 + 0.03 rho / 1 + 0.1 rho (-1..2 % of the pass).
#ifndef LMONO_WR_A0
#define LMONO_WR_A0 0.25f
#define LMONO_WR_B0 0.02f
#define LMONO_WR_A1 1.0f
#define LMONO_WR_B1 0.1f
#endif
__device__ __forceinline__ float walk_radius(int pass, float rho)
{
    return pass == 0 ? LMONO_WR_A0 + LMONO_WR_B0 * rho : (pass == 1 ? LMONO_WR_A1 + LMONO_WR_B1 * rho : (pass == 2 ? fminf(5.0f, 1.0f + 0.0045f * rho * rho) : 5.0f));
}

// asin(x) <= x (1 + 0.5708 x^2) on [0, 1] (equality at 0 and 1): conservative arc half-width without libm
__device__ __forceinline__ float asin_upper(float x) { return x * (1.0f + 0.5708f * x * x); }

// walk candidate as one u64 key: (float bits of d2) << 32 | rank in the reference's visiting order; starts at (25.0, 0),
// so only candidates with d2 < 25 ever replace it
typedef unsigned long long WalkBest;
__device__ __forceinline__ void walk_update(WalkBest &bst, float d, unsigned int seq)
{
    const WalkBest k = ((unsigned long long)__float_as_uint(d) << 32) | seq;
    bst = k < bst ? k : bst;
}

// ---- half-wave (32-lane) groups: two feature points per wave --------------------------------------------------
// The per-feature work is dominated by scalar-like instructions (hashing, probing, bookkeeping) and by chains of
// dependent loads, so each feature gets 32 lanes (the 27 cells of shell 1 fit) and independent loads are issued in
// batches before any of them is consumed.  Variants measured: profiles/r1/NOTES.md.
constexpr int kGroup = 32;

__device__ __forceinline__ unsigned int group_ballot(bool pred, int gbase)
{
    return (unsigned int)((__ballot(pred) >> gbase) & 0xffffffffull);
}
// minimum of a (hi, lo) key over the lanes of each 32-lane group, returned to every lane of the group: two 32-bit DPP
// reductions (row_shr 1/2/4/8, row_bcast:15 into rows 1 and 3 leave the group minima in lanes 31 and 63)
__device__ __forceinline__ unsigned int group_min_u32(unsigned int v, int gbase)
{
    v = min(v, dpp_mov_u32<0x111, 0xf>(v, ~0u));
    v = min(v, dpp_mov_u32<0x112, 0xf>(v, ~0u));
    v = min(v, dpp_mov_u32<0x114, 0xf>(v, ~0u));
    v = min(v, dpp_mov_u32<0x118, 0xf>(v, ~0u));
    v = min(v, dpp_mov_u32<0x142, 0xa>(v, ~0u));
    const unsigned int g0 = (unsigned int)__builtin_amdgcn_readlane((int)v, 31), g1 = (unsigned int)__builtin_amdgcn_readlane((int)v, 63);
    return gbase ? g1 : g0;
}
__device__ __forceinline__ unsigned long long group_min_u64(unsigned long long k, int gbase)
{
    const unsigned int hi = (unsigned int)(k >> 32), lo = (unsigned int)k;
    const unsigned int mh = group_min_u32(hi, gbase);
    const unsigned int ml = group_min_u32(hi == mh ? lo : ~0u, gbase);
    return ((unsigned long long)mh << 32) | ml;
}

// Nearest-point candidate as ONE u64 key: (float bits of d2) << 32 | index << 7 | line.  d2 >= 0, so the unsigned order
// of the key is the (d2, index) order of the reference's tie rule and the line rides along; ~0 = no candidate yet.
// The cell-sorted point copy carries index << 7 | line in .w (k_grid_build).
typedef unsigned long long NnBest;
constexpr NnBest kNnNone = ~0ull;
__device__ __forceinline__ void nn_update(NnBest &bst, const float4 &p, float qx, float qy, float qz)
{
    const float d = dist2f(p.x, p.y, p.z, qx, qy, qz);
    const NnBest k = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned int)__float_as_int(p.w);
    bst = k < bst ? k : bst;
}

// walk candidate from the (azimuth bin, line) index: .w = cloud index << 7 | line; only lines ra-2 .. ra+2 take part
__device__ __forceinline__ void walk_point(const float4 &cpt, int ra, int closest, int w_lo, int w_hi, bool edge,
                                           float qx, float qy, float qz, WalkBest &bs, WalkBest &bo)
{
    const int w = __float_as_int(cpt.w);
    const int j = w >> 7, v = w & 127;
    if (v < ra - 2 || v > ra + 2 || j == closest || j < w_lo || j >= w_hi) return;
    const bool fwd = j > closest;
    const unsigned int seq = fwd ? (unsigned int)(j - closest - 1) : kSeqBack + (unsigned int)(closest - 1 - j);
    const float d = dist2f(cpt.x, cpt.y, cpt.z, qx, qy, qz);
    const bool is_other = fwd ? (v > ra) : (v < ra);
    if (is_other) walk_update(bo, d, seq);
    else if (!edge) walk_update(bs, d, seq);
}

// sweep the cell runs held by the group's lanes (pack_run), four non-empty cells per round, ONE CELL PER 8-LANE SUB-GROUP:
// 1 m cells of a 0.2 m-voxelised cloud hold ~10 points (corner cells 1-3), so a 32-lane load per cell leaves most lanes
// idle while every wave instruction costs the same; with four cells side by side a round is one or two load + update
// steps instead of four.  Two steps are requested back to back.
__device__ __forceinline__ void nn_sweep(const float4 *gpts, int run, int gl, int gbase, float qx, float qy, float qz, NnBest &nb)
{
    unsigned int m = group_ballot(run >= (1 << 17), gbase);
    const int sub = gl >> 3, sl = gl & 7;
    while (m) {
        int src = -1;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int f = __ffs((int)m) - 1;      // -1 when no cell is left
            if (u == sub) src = f;
            m &= m - 1;
        }
        const int rv = __shfl(run, max(src, 0), kGroup);
        const int st = rv & 0x1ffff, cn = src >= 0 ? rv >> 17 : 0;
        int i = sl;
        while (__any(i < cn)) {
            float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0, v2 = v0, v3 = v0;
            if (i < cn) v0 = gpts[st + i];
            if (i + 8 < cn) v1 = gpts[st + i + 8];
            if (i + 16 < cn) v2 = gpts[st + i + 16];
            if (i + 24 < cn) v3 = gpts[st + i + 24];
            if (i < cn) nn_update(nb, v0, qx, qy, qz);
            if (i + 8 < cn) nn_update(nb, v1, qx, qy, qz);
            if (i + 16 < cn) nn_update(nb, v2, qx, qy, qz);
            if (i + 24 < cn) nn_update(nb, v3, qx, qy, qz);
            i += 32;
        }
    }
}

// the group sweeps one contiguous run of the (azimuth bin, line) index (all lines of an arc of bins), four loads per lane in flight
__device__ __forceinline__ void nn_sweep_arc(const float4 *lpts, int st, int cn, int gl, float qx, float qy, float qz, NnBest &nb)
{
    for (int i = gl; i < cn; i += 4 * kGroup) {
        float4 v4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v4[u] = lpts[st + min(i + u * kGroup, cn - 1)];
#pragma unroll
        for (int u = 0; u < 4; u++) if (i + u * kGroup < cn) nn_update(nb, v4[u], qx, qy, qz);
    }
}

// gl = lane inside the 32-lane group, gbase = first wave lane of the group (0 or 32)
template <bool kEdge>     // edge (corner cloud) and plane (surf cloud) features as two specialised copies: no per-lane pointer selects
__device__ __forceinline__ int4 correspond_g32(const BatchView &b, int k, int qi, const double *x, int gl, int gbase, int seed, int &closest_out)
{
    closest_out = -1;
    const int n_sharp = b.feat_n[k * 4 + 0];
    constexpr bool edge = kEdge;
    const float4 p = edge ? b.sharp[(size_t)k * kMaxSharp + qi] : b.flat[(size_t)k * kMaxFlat + (qi - n_sharp)];
    double rx, ry, rz;
    quat_rotate(x, (double)p.x, (double)p.y, (double)p.z, rx, ry, rz);
    const float qx = (float)(rx + x[4]), qy = (float)(ry + x[5]), qz = (float)(rz + x[6]);
    const int l = k - 1;
    const int cl = edge ? 0 : 1;
    const GridCell *cell = edge ? b.cg_cell + (size_t)l * kCornerTable : b.sg_cell + (size_t)l * kSurfTable;
    const float4 *gpts = edge ? b.cg_pts + (size_t)l * kMaxLessSharp : b.sg_pts + b.off[l];
    const float4 *lb_pts = edge ? b.lbc_pts + (size_t)l * kMaxLessSharp : b.lbs_pts + b.off[l];
    const int n_last = b.feat_n[l * 4 + (edge ? 1 : 3)];
    const unsigned int mask = (unsigned int)b.grid_mask[l * 2 + cl], pmask = min(mask, (unsigned int)(kGridPage - 1));
    int4 out = make_int4(-1, -1, -1, 0);
    if (n_last == 0) return out;

    // ---- exact NN, shell 1: lanes 0..26 probe one cell each
    const float fx = qx * kInvCell, fy = qy * kInvCell, fz = qz * kInvCell;
    const int cqx = (int)floorf(fx), cqy = (int)floorf(fy), cqz = (int)floorf(fz);
    int run = 0;          // pack_run(start, count) of this lane's cell, 0 = empty
    bool near = false;
    // Second outer iteration of a scan pair: the nearest point of the first one, seen from the updated pose, is a real
    // candidate; when it lies inside shell 1 only the cells whose box comes closer than it can hold the nearest point,
    // typically one or two of the 27 -- one sweep round settles the search exactly.
    NnBest nb = kNnNone;
    bool seeded = false;
    float sd = 0.f;
    if (seed >= 0 && seed < n_last) {
        const float4 pp = (edge ? b.less_sharp + (size_t)l * kMaxLessSharp : b.less_flat + b.off[l])[seed];
        const float d = dist2f(pp.x, pp.y, pp.z, qx, qy, qz);
        const int ln = (int)pp.w;
        if (d <= (kCell * 0.9999f) * (kCell * 0.9999f)) {
            seeded = true; sd = d;
            nb = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned int)((seed << 7) | (ln < 0 ? 0 : (ln > 65 ? 65 : ln)));
        }
    }
    if (gl < 27 && b.has_grid) {       // without grids (not built: the default search does not use them) the arc sweep below finds the nearest point
        const int dx = gl % 3 - 1, dy = (gl / 3) % 3 - 1, dz = gl / 9 - 1;
        // the 2x2x2 block of cells whose faces are all >= half a cell away from the query
        const int sx = (fx - (float)cqx) >= 0.5f ? 1 : -1, sy = (fy - (float)cqy) >= 0.5f ? 1 : -1, sz = (fz - (float)cqz) >= 0.5f ? 1 : -1;
        near = (dx == 0 || dx == sx) && (dy == 0 || dy == sy) && (dz == 0 || dz == sz);
        bool wanted = true;
        if (seeded) {
            const float lx = (float)(cqx + dx) * kCell, ly = (float)(cqy + dy) * kCell, lz = (float)(cqz + dz) * kCell;
            const float ex = fmaxf(fmaxf(lx - qx, qx - (lx + kCell)), 0.f), ey = fmaxf(fmaxf(ly - qy, qy - (ly + kCell)), 0.f), ez = fmaxf(fmaxf(lz - qz, qz - (lz + kCell)), 0.f);
            const float eb = fmaxf(sqrtf(ex * ex + ey * ey + ez * ez) - 1e-3f, 0.f);
            wanted = !(eb * eb > sd);
        }
        if (wanted) {
            const unsigned long long kk = cell_key(cqx + dx, cqy + dy, cqz + dz);
            unsigned int sl = hash_key(kk) & mask;
            while (true) {
                const GridCell e = cell[sl];
                if (e.key == kk) { run = pack_run(e.start, e.cnt); break; }
                if (e.key == kEmptyKey) break;
                sl = grid_next(sl, pmask);
            }
        }
    }
    unsigned long long best;
    if (seeded) {
        nn_sweep(gpts, run, gl, gbase, qx, qy, qz, nb);
        best = group_min_u64(nb, gbase);
    } else {
    // near block first: any point outside it is farther than half a cell
    nn_sweep(gpts, near ? run : 0, gl, gbase, qx, qy, qz, nb);
    best = group_min_u64(nb, gbase);
    {
        const float bound_h = 0.5f * kCell * 0.9999f;
        if (!(best != ~0ull && __uint_as_float((unsigned int)(best >> 32)) <= bound_h * bound_h)) {
            // remaining 19 cells, minus those whose box is farther than the best point found so far
            int run_far = near ? 0 : run;
            if (best != ~0ull && gl < 27) {
                const float lx = (float)(cqx + gl % 3 - 1) * kCell, ly = (float)(cqy + (gl / 3) % 3 - 1) * kCell, lz = (float)(cqz + gl / 9 - 1) * kCell;
                const float ex = fmaxf(fmaxf(lx - qx, qx - (lx + kCell)), 0.f), ey = fmaxf(fmaxf(ly - qy, qy - (ly + kCell)), 0.f), ez = fmaxf(fmaxf(lz - qz, qz - (lz + kCell)), 0.f);
                // 1e-3 m slack: the cell of a point is floor(coordinate), exact, so only the float box arithmetic needs margin
                const float eb = fmaxf(sqrtf(ex * ex + ey * ey + ez * ez) - 1e-3f, 0.f);
                if (eb * eb > __uint_as_float((unsigned int)(best >> 32))) run_far = 0;
            }
            nn_sweep(gpts, run_far, gl, gbase, qx, qy, qz, nb);
            best = group_min_u64(nb, gbase);
        }
    }
    }
    {
        const float bound = kCell * 0.9999f;
        // without grids nothing has been swept yet: the arc sweep below does the whole search (its radius is the seed's distance, or 5 m)
        const bool settled = b.has_grid && best != ~0ull && __uint_as_float((unsigned int)(best >> 32)) <= bound * bound;
        if (!settled) {
            // rare (~2 % of the features): the nearest point is not provably inside shell 1.  Instead of probing the
            // hash cells of shells 2..6 (up to ~150 rounds of dependent probes for a feature without any neighbour),
            // sweep the (line, azimuth-bin) index: every point closer than rr to the query lies within
            // +-asin(rr / rho) of its azimuth, so the candidates are one contiguous run per scan line (two when the
            // arc wraps) -- a few dozen independent coalesced loads.  rr = the best distance of shell 1, or 5 m.
            const int *table = b.lb_start + (size_t)(l * 2 + cl) * (kLineKeys + 1);
            const float rho = sqrtf(qx * qx + qy * qy);
            const float th = atan2f(qy, qx) + 3.14159265f;
            const float bd = best != ~0ull ? __uint_as_float((unsigned int)(best >> 32)) : 25.0f;
            const float rr = bd < 25.0f ? sqrtf(bd) * 1.0005f + 1e-3f : 5.0f;
            int b_lo = 0, nbins = kAzBins;
            if (rho > rr * 1.002f) {
                const float alpha = asin_upper(rr / rho) + kArcSlackBins * (6.28318531f / kAzBins);
                const int lo = (int)floorf((th - alpha) * (kAzBins / 6.28318531f));
                const int hi = (int)floorf((th + alpha) * (kAzBins / 6.28318531f));
                if (hi - lo + 1 < kAzBins) { b_lo = ((lo % kAzBins) + kAzBins) % kAzBins; nbins = hi - lo + 1; }
            }
            lb_for_runs(table, b_lo, nbins, 0, 65, [&](int s0, int cn) { nn_sweep_arc(lb_pts, s0, cn, gl, qx, qy, qz, nb); });
            best = group_min_u64(nb, gbase);
        }
    }
    if (best == ~0ull || !((double)__uint_as_float((unsigned int)(best >> 32)) < 25.0)) return out;
    const int closest = (int)((unsigned int)(best & 0xffffffffull) >> 7);
    const int ra = (int)(best & 127ull);
    closest_out = closest;

    // ---- scan-line walk over the (line, azimuth) index: lines ra-2 .. ra+2, index window (last_le[ra-3], first_ge[ra+3]).
    // Passes of growing radius (walk_radius): first only the arc that can hold the neighbouring lines' points right next to the
    // nearest point; minima found strictly inside a pass's ball are final because every point outside the arc is farther.
    const int *fge = b.line_first_ge + (size_t)(l * 2 + cl) * 66;
    const int *lle = b.line_last_le + (size_t)(l * 2 + cl) * 66;
    const int *table = b.lb_start + (size_t)(l * 2 + cl) * (kLineKeys + 1);
    const float rho = sqrtf(qx * qx + qy * qy);
    const float th = atan2f(qy, qx) + 3.14159265f;
    const int w_lo = ra - 3 >= 0 ? lle[ra - 3] + 1 : 0;
    const int w_hi = ra + 3 <= 65 ? fge[ra + 3] : n_last;
    const unsigned long long thr = pack_fu(25.0f, 0u);
    unsigned long long same = thr, other = thr;
    float rprev = 0.f;
    for (int pass = (rho > 5.01f ? 0 : 3); pass < 4; pass++) {
        const float rw = walk_radius(pass, rho);
        if (rw <= rprev) continue;                  // this pass would not widen the arc
        rprev = rw;
        int b_lo = 0, nbins = kAzBins;
        if (rho > 5.01f) {
            const float alpha = asin_upper(rw / rho) + kArcSlackBins * (6.28318531f / kAzBins);
            const int lo = (int)floorf((th - alpha) * (kAzBins / 6.28318531f));
            const int hi = (int)floorf((th + alpha) * (kAzBins / 6.28318531f));
            if (hi - lo + 1 < kAzBins) { b_lo = ((lo % kAzBins) + kAzBins) % kAzBins; nbins = hi - lo + 1; }
        }
        WalkBest bs = thr, bo = thr;
        lb_for_runs(table, b_lo, nbins, max(ra - 2, 0), min(ra + 2, 65), [&](int r0, int cnw) {
            for (int i = gl; i < cnw; i += 4 * kGroup) {
                float4 v4[4];
#pragma unroll
                for (int u = 0; u < 4; u++) v4[u] = lb_pts[r0 + min(i + u * kGroup, cnw - 1)];
#pragma unroll
                for (int u = 0; u < 4; u++) if (i + u * kGroup < cnw) walk_point(v4[u], ra, closest, w_lo, w_hi, edge, qx, qy, qz, bs, bo);
            }
        });
        same = group_min_u64(bs, gbase);
        other = group_min_u64(bo, gbase);
        if (rw >= 5.0f) break;
        const unsigned long long lim = pack_fu(rw * rw * 0.998f, 0u);   // squared distance strictly inside this pass's ball
        if (other < lim && (edge || same < lim)) break;
    }
    const int i_other = other < thr ? seq_to_index((unsigned int)(other & 0xffffffffull), closest) : -1;
    if (edge) {
        if (i_other >= 0) out = make_int4(closest, i_other, -1, 1);
        return out;
    }
    const int i_same = same < thr ? seq_to_index((unsigned int)(same & 0xffffffffull), closest) : -1;
    if (i_same >= 0 && i_other >= 0) out = make_int4(closest, i_same, i_other, 2);
    return out;
}

// one feature point served by a 32-lane group: search, correspondence indices, seed of the second outer iteration and the
// 64-B residual-block record of the solver
__device__ __forceinline__ void correspond_group(const BatchView &b, const OdomView &o, int c, int k, int qi, int outer, int gl, int gbase)
{
    const int n_sharp = b.feat_n[k * 4 + 0];
    const int l = k - 1;
    int4 *corr = (int4 *)o.corr + (size_t)c * kMaxQueries;
    int *seed_c = o.seed ? o.seed + (size_t)c * kMaxQueries : nullptr;
    int closest;
    const int seed = (outer == 1 && seed_c) ? seed_c[qi] : -1;
    const double *x = o.state + c * 8;
    const int4 r = qi < n_sharp ? correspond_g32<true>(b, k, qi, x, gl, gbase, seed, closest) : correspond_g32<false>(b, k, qi, x, gl, gbase, seed, closest);
    if (gl == 0) { corr[qi] = r; if (outer == 0 && seed_c) seed_c[qi] = closest; }
    // residual-block record for the solver: the feature point and its 2 (edge) or 3 (plane) partners, 64 B
    if (gl < 4) {
        const bool edge = qi < n_sharp;
        const float4 *cloud = edge ? b.less_sharp + (size_t)l * kMaxLessSharp : b.less_flat + b.off[l];
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gl == 0) {
            v = edge ? b.sharp[(size_t)k * kMaxSharp + qi] : b.flat[(size_t)k * kMaxFlat + (qi - n_sharp)];
            v.w = __int_as_float(r.w);
        } else if (r.w != 0) {
            const int idx = gl == 1 ? r.x : (gl == 2 ? r.y : r.z);
            if (idx >= 0) v = cloud[idx];
        }
        o.crec[((size_t)c * kMaxQueries + qi) * 4 + gl] = v;
    }
}

// the generic path (array-order walk) for one feature point, whole wave
__device__ __forceinline__ void correspond_wave(const BatchView &b, const OdomView &o, int c, int k, int qi, int lane)
{
    const int n_sharp = b.feat_n[k * 4 + 0];
    const int l = k - 1;
    int4 *corr = (int4 *)o.corr + (size_t)c * kMaxQueries;
    const double *x = o.state + c * 8;
    const int4 r = correspond_one(b, k, qi, x, lane);
    if (lane == 0) { corr[qi] = r; if (o.seed) o.seed[(size_t)c * kMaxQueries + qi] = -1; }
    if (lane < 4) {
        const bool edge = qi < n_sharp;
        const float4 *cloud = edge ? b.less_sharp + (size_t)l * kMaxLessSharp : b.less_flat + b.off[l];
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane == 0) { v = edge ? b.sharp[(size_t)k * kMaxSharp + qi] : b.flat[(size_t)k * kMaxFlat + (qi - n_sharp)]; v.w = __int_as_float(r.w); }
        else if (r.w != 0) { const int idx = lane == 1 ? r.x : (lane == 2 ? r.y : r.z); if (idx >= 0) v = cloud[idx]; }
        o.crec[((size_t)c * kMaxQueries + qi) * 4 + lane] = v;
    }
}

// step t of every chain: one 32-lane group per feature point of the chain's current scan (8 features per workgroup).
// XCD-aware mapping: workgroups are dealt round-robin over the 8 XCDs (b and b+8 share an XCD and its 4 MB L2), so
// the 1-D grid is decoded such that all blocks of a chain run on ONE XCD, one chain after the other: the ~2 MB of
// grid / index / point data of a chain's "last" scan then stay in that XCD's L2 for its 1836 features.
// (sizing the grid to the sensor's feature bound, 230 instead of 288 workgroups per chain for an HDL-64, measured 3 % slower)
constexpr int kCorrBlocks = kMaxQueries / 8;

#ifdef LMONO_DIAG_SEARCH
__global__ __launch_bounds__(256, 8) void k_correspond(BatchView b, OdomView o, int step, int outer)
{
    const int xcd = blockIdx.x & 7, u = blockIdx.x >> 3;
    const int c = o.chain0 + (u / kCorrBlocks) * 8 + xcd;
    const int qblock = u % kCorrBlocks;
    if (c >= o.chain1) return;
    const int lane = threadIdx.x & 63;
    const int gl = threadIdx.x & (kGroup - 1), gbase = lane & ~(kGroup - 1);
    int own;
    const int k = chain_scan(o, c, step, own);
    if (k < 0) return;
    const int n_sharp = b.feat_n[k * 4 + 0];
    const int nq = n_sharp + b.feat_n[k * 4 + 2];
    const double *x = o.state + c * 8;
    int4 *corr = (int4 *)o.corr + (size_t)c * kMaxQueries;
    const int l = k - 1;
    if (b.status[l] & (kStatusIrregularLines | kStatusDenseCell)) {
        // rare: array-order walk with the whole wave, the wave's two features one after the other
        for (int t = 0; t < 2; t++) {
            const int qi = qblock * 8 + (threadIdx.x >> 6) * 2 + t;
            if (qi >= nq) break;
            correspond_wave(b, o, c, k, qi, lane);
        }
        return;
    }
    const int qi = qblock * 8 + (threadIdx.x >> 5);
    if (qi >= nq) return;
    correspond_group(b, o, c, k, qi, outer, gl, gbase);
}
#endif // LMONO_DIAG_SEARCH

// The same search for the feature points of a device work list ([0] = count, then chain << 12 | feature index): the points the
// LDS tile search (corr_tile.hip) defers.  Phase 1 serves them with 32-lane groups; phase 2 (scan pairs flagged irregular / dense:
// rare) with the whole-wave array-order walk.  Fixed grid, items strided over it.
__global__ __launch_bounds__(256, 4) void k_correspond_list(BatchView b, OdomView o, int step, int outer, const unsigned int *wl, unsigned long long *stats)
{
    const unsigned int n = wl[0];
    if (blockIdx.x == 0 && threadIdx.x == 0 && stats && n) atomicAdd(&stats[0], (unsigned long long)n);       // chain groups run this kernel concurrently
    const int lane = threadIdx.x & 63;
    const int gl = threadIdx.x & (kGroup - 1), gbase = lane & ~(kGroup - 1);
    for (unsigned int it = blockIdx.x * 8 + (threadIdx.x >> 5); it < n; it += gridDim.x * 8) {
        const unsigned int e = wl[1 + it];
        const int c = (int)(e >> 12), qi = (int)(e & 4095u);
        int own;
        const int k = chain_scan(o, c, step, own);
        if (k < 0 || (b.status[k - 1] & (kStatusIrregularLines | kStatusDenseCell))) continue;
        correspond_group(b, o, c, k, qi, outer, gl, gbase);
    }
    for (unsigned int it = blockIdx.x * 4 + (threadIdx.x >> 6); it < n; it += gridDim.x * 4) {
        const unsigned int e = wl[1 + it];
        const int c = (int)(e >> 12), qi = (int)(e & 4095u);
        int own;
        const int k = chain_scan(o, c, step, own);
        if (k < 0 || !(b.status[k - 1] & (kStatusIrregularLines | kStatusDenseCell))) continue;
        correspond_wave(b, o, c, k, qi, lane);
    }
}

// ------------------------------------------------------------------------------------------------
// Residual blocks.  lp = q * cp + t (Eigen _transformVector polynomial); edge r = ((lp-a) x (lp-b)) / |a-b| (3 rows),
// plane r = (lp - j) . n (1 row).  d r / d lp is constant: [b-a]_x / |a-b| and n^T.
constexpr int kLmT = 256;   // threads per chain in k_lm_solve (512 measured slower, with and without register spills: reductions and the serial trust-region part dominate)
constexpr int kLmW = kLmT / 64;
struct LmAcc { double H[21]; double g[6]; double cost; };

// The LM solve below (accumulate_row .. k_lm_solve) allows contraction to FMA: its parity bar is a tolerance (1e-9 against the oracle's
// loop), and with -ffp-contract=off every a * b + c of the 6 x 6 normal equations is two dependent instructions on the one wave per
// SIMD that runs a chain's solve.  The de-skew transform of the SEARCH (quat_rotate in the correspondence kernels) stays uncontracted:
// its float result must equal the oracle's bit for bit.
__device__ __forceinline__ void quat_rotate_fma(const double *q, double vx, double vy, double vz, double &ox, double &oy, double &oz)
{
#pragma clang fp contract(fast)
    const double ux = q[0], uy = q[1], uz = q[2], w = q[3];
    const double uvx = 2.0 * (uy * vz - uz * vy);
    const double uvy = 2.0 * (uz * vx - ux * vz);
    const double uvz = 2.0 * (ux * vy - uy * vx);
    ox = vx + w * uvx + (uy * uvz - uz * uvy);
    oy = vy + w * uvy + (uz * uvx - ux * uvz);
    oz = vz + w * uvz + (ux * uvy - uy * uvx);
}

__device__ __forceinline__ void accumulate_row(LmAcc &a, const double *J, double r)
{
#pragma clang fp contract(fast)
    int t = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        a.g[i] += J[i] * r;
#pragma unroll
        for (int j = i; j < 6; j++) a.H[t++] += J[i] * J[j];
    }
}

__device__ __forceinline__ void huber(double s, double &rho0, double &rho1)
{
    const double a = 0.1, bb = 0.1 * 0.1;
    if (s > bb) {
        const double r = sqrt(s);
        rho0 = 2.0 * a * r - bb;
        rho1 = a / r;
        if (rho1 < DBL_MIN) rho1 = DBL_MIN;
    } else { rho0 = s; rho1 = 1.0; }
}

// A staged residual block (64 B in LDS, planes of 16 B): (cp, kind) as float4, then six doubles that do not depend on the pose and
// are computed once per launch instead of once per evaluation (up to nine per launch):
//   edge   a, e^ = (a - b) / |a - b|:          r = (lp - a) x e^   [= ((lp - a) x (lp - b)) / |a - b| since (lp - a) x (lp - a) = 0],  d r / d lp = -[e^]_x
//   plane  j, n = normalised((j - l) x (j - m)):  r = (lp - j) . n,  d r / d lp = n^T
__device__ __forceinline__ void stage_block(const float4 cp, const float4 A, const float4 B, const float4 Cc, double *P)
{
    const int kind = __float_as_int(cp.w);
    if (kind == 1) {
        const double ex = (double)A.x - (double)B.x, ey = (double)A.y - (double)B.y, ez = (double)A.z - (double)B.z;
        const double inv = 1.0 / sqrt(ex * ex + ey * ey + ez * ez);
        P[0] = (double)A.x; P[1] = (double)A.y; P[2] = (double)A.z;
        P[3] = ex * inv; P[4] = ey * inv; P[5] = ez * inv;
    } else {
        const double jx = (double)A.x, jy = (double)A.y, jz = (double)A.z;
        const double ux = jx - (double)B.x, uy = jy - (double)B.y, uz = jz - (double)B.z;
        const double wx = jx - (double)Cc.x, wy = jy - (double)Cc.y, wz = jz - (double)Cc.z;
        double nx = uy * wz - uz * wy, ny = uz * wx - ux * wz, nz = ux * wy - uy * wx;
        const double nn = sqrt(nx * nx + ny * ny + nz * nz);
        if (nn > 0.0) { nx /= nn; ny /= nn; nz /= nn; }
        P[0] = jx; P[1] = jy; P[2] = jz; P[3] = nx; P[4] = ny; P[5] = nz;
    }
}

// One Jacobian row of a residual block: d r / d lp = d (3-vector), scaled by sr.  d lp / d(local rotation) = -2 [R v]_x: the reference
// differentiates the polynomial lp = v + 2 w (u x v) + 2 u x (u x v) with respect to the four quaternion components and multiplies by
// the 4 x 3 plus-Jacobian of ceres::EigenQuaternionParameterization (x_new = (delta, 1) (x) x, a rotation by 2 |delta| in front of R);
// along those tangent directions the product is the derivative of the rotation itself -- the same numbers to rounding (7e-15 on entries
// of magnitude 20).  So the rotation part of the row is d^T (-2 [rv]_x) = 2 (rv x d), the translation part d.
__device__ __forceinline__ void lm_row(LmAcc &acc, double d0, double d1, double d2, double rvx, double rvy, double rvz, double res, double sr)
{
#pragma clang fp contract(fast)
    double J[6];
    const double s2 = 2.0 * sr;
    J[0] = (rvy * d2 - rvz * d1) * s2; J[1] = (rvz * d0 - rvx * d2) * s2; J[2] = (rvx * d1 - rvy * d0) * s2;
    J[3] = d0 * sr; J[4] = d1 * sr; J[5] = d2 * sr;
    accumulate_row(acc, J, res * sr);
}

// One residual block at the pose (Rm = rotation matrix of x's quaternion, row-major; x[4..6] = translation), as straight-line code per
// kind: the generic form (a rows loop of run-time length over a 3 x 3 derivative array) cost ~300 instructions per plane and ~460 per
// edge on the one wave per SIMD that runs a chain's solve.
template <bool kJac, bool kEdge>
__device__ __forceinline__ void eval_block(const float4 cp, const double2 P01, const double2 P23, const double2 P45, const double *Rm, const double *x, LmAcc &acc)
{
#pragma clang fp contract(fast)
    if (__float_as_int(cp.w) == 0) return;
    const double vx = (double)cp.x, vy = (double)cp.y, vz = (double)cp.z;
    const double rvx = Rm[0] * vx + Rm[1] * vy + Rm[2] * vz, rvy = Rm[3] * vx + Rm[4] * vy + Rm[5] * vz, rvz = Rm[6] * vx + Rm[7] * vy + Rm[8] * vz;
    const double dx = rvx + x[4] - P01.x, dy = rvy + x[5] - P01.y, dz = rvz + x[6] - P23.x;     // lp - a  /  lp - j
    const double qx = P23.y, qy = P45.x, qz = P45.y;                                            // e^      /  n
    double r0, r1 = 0.0, r2 = 0.0, sq;
    if (kEdge) { r0 = dy * qz - dz * qy; r1 = dz * qx - dx * qz; r2 = dx * qy - dy * qx; sq = r0 * r0 + r1 * r1 + r2 * r2; }
    else { r0 = dx * qx + dy * qy + dz * qz; sq = r0 * r0; }
    double rho0, rho1;
    huber(sq, rho0, rho1);
    acc.cost += 0.5 * rho0;
    if (!kJac) return;
    const double sr = rho1 == 1.0 ? 1.0 : sqrt(rho1);      // inliers: sqrt(1) = 1 exactly, without the 25-instruction fp64 square root
    if (kEdge) {
        // rows of d r / d lp = -[e^]_x
        lm_row(acc, 0.0, qz, -qy, rvx, rvy, rvz, r0, sr);
        lm_row(acc, -qz, 0.0, qx, rvx, rvy, rvz, r1, sr);
        lm_row(acc, qy, -qx, 0.0, rvx, rvy, rvz, r2, sr);
    } else
        lm_row(acc, qx, qy, qz, rvx, rvy, rvz, r0, sr);
}

// Sum over all residual blocks of a chain by the kLmT threads of its workgroup (wave butterfly, then the wave partials are added in
// fixed order); the sums -- H (21, upper triangle row by row), g (6), cost -- are left in LDS (s_sum) for every thread to read: the
// trust-region code below keeps no copy of them in registers.  srec = the chain's records in LDS, one plane of kMaxQueries float4
// per record field (conflict-free 16-B reads).
template <bool kJac>
__device__ __forceinline__ void evaluate_block(const float4 *srec, int n_edge, int nq, const double *x, double (*s_red)[28], double *s_sum)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    LmAcc acc;
    acc.cost = 0.0;
    if (kJac) {
#pragma unroll
        for (int i = 0; i < 21; i++) acc.H[i] = 0.0;
#pragma unroll
        for (int i = 0; i < 6; i++) acc.g[i] = 0.0;
    }
    // rotation matrix of the pose's quaternion (uniform; once per sweep instead of a quaternion rotation per block)
    double Rm[9];
    {
#pragma clang fp contract(fast)
        const double ux = x[0], uy = x[1], uz = x[2], w = x[3];
        Rm[0] = 1.0 - 2.0 * (uy * uy + uz * uz); Rm[1] = 2.0 * (ux * uy - w * uz);       Rm[2] = 2.0 * (ux * uz + w * uy);
        Rm[3] = 2.0 * (ux * uy + w * uz);       Rm[4] = 1.0 - 2.0 * (ux * ux + uz * uz); Rm[5] = 2.0 * (uy * uz - w * ux);
        Rm[6] = 2.0 * (ux * uz - w * uy);       Rm[7] = 2.0 * (uy * uz + w * ux);       Rm[8] = 1.0 - 2.0 * (ux * ux + uy * uy);
    }
    // a thread visits its blocks in ascending order as before (edge blocks are the first n_edge of a chain's list), but every loop body
    // is one kind: no divergence inside a wave except at unused blocks
    const double2 *sp = (const double2 *)srec;
    int qi = tid;
    for (; qi < n_edge; qi += kLmT)
        eval_block<kJac, true>(srec[qi], sp[kMaxQueries + qi], sp[2 * kMaxQueries + qi], sp[3 * kMaxQueries + qi], Rm, x, acc);
    for (; qi < nq; qi += kLmT)
        eval_block<kJac, false>(srec[qi], sp[kMaxQueries + qi], sp[2 * kMaxQueries + qi], sp[3 * kMaxQueries + qi], Rm, x, acc);
    acc.cost = wave_sum_d(acc.cost);
    if (kJac) {
#pragma unroll
        for (int i = 0; i < 21; i++) acc.H[i] = wave_sum_d(acc.H[i]);
#pragma unroll
        for (int i = 0; i < 6; i++) acc.g[i] = wave_sum_d(acc.g[i]);
    }
    __syncthreads();   // previous readers of s_red are done
    if (lane == 0) {
        s_red[wave][27] = acc.cost;
        if (kJac) {
            for (int i = 0; i < 21; i++) s_red[wave][i] = acc.H[i];
            for (int i = 0; i < 6; i++) s_red[wave][21 + i] = acc.g[i];
        }
    }
    __syncthreads();
    if (tid < 28 && (kJac || tid == 27)) {
        double t = 0.0;
        for (int w = 0; w < kLmW; w++) t += s_red[w][tid];
        s_sum[tid] = t;
    }
    __syncthreads();
}

// entry (i, j) of a symmetric 6 x 6 matrix stored as its upper triangle row by row (the order of accumulate_row)
__device__ __forceinline__ int sym6(int i, int j) { const int a = i < j ? i : j, b = i < j ? j : i; return a * 6 - a * (a - 1) / 2 + (b - a); }

__device__ __forceinline__ bool chol_solve6(const double *A, const double *bvec, double *xo)
{
    double L[36];
    for (int i = 0; i < 36; i++) L[i] = 0.0;
    for (int i = 0; i < 6; i++)
        for (int j = 0; j <= i; j++) {
            double s = A[i * 6 + j];
            for (int k = 0; k < j; k++) s -= L[i * 6 + k] * L[j * 6 + k];
            if (i == j) { if (!(s > 0.0)) return false; L[i * 6 + i] = sqrt(s); }
            else L[i * 6 + j] = s / L[j * 6 + j];
        }
    double y[6];
    for (int i = 0; i < 6; i++) { double s = bvec[i]; for (int k = 0; k < i; k++) s -= L[i * 6 + k] * y[k]; y[i] = s / L[i * 6 + i]; }
    for (int i = 5; i >= 0; i--) { double s = y[i]; for (int k = i + 1; k < 6; k++) s -= L[k * 6 + i] * xo[k]; xo[i] = s / L[i * 6 + i]; }
    return true;
}

// The same factorisation and substitutions on the LOWER triangle stored row by row (entry (i, j), j <= i, at i (i + 1) / 2 + j),
// in place: 21 doubles instead of two 6 x 6 arrays, same operations in the same order.
// Round 4: the 27 divisions by the factor's diagonal are multiplications by its six reciprocals (an fp64 division is ~35 instructions on the one wave
// per SIMD that runs a chain's solve, and every thread of the workgroup runs this redundantly four times per launch: ~24 k of a launch's ~126 k cycles).
// Same factorisation to rounding; the solve's parity bar is 1e-9 against the oracle's loop.
__device__ __forceinline__ bool chol_solve6_packed(double *L, const double *bvec, double *xo)
{
#pragma clang fp contract(fast)
    double inv[6];
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) {
            double s = L[i * (i + 1) / 2 + j];
#pragma unroll
            for (int k = 0; k < j; k++) s -= L[i * (i + 1) / 2 + k] * L[j * (j + 1) / 2 + k];
            if (i == j) { if (!(s > 0.0)) return false; const double d = sqrt(s); L[i * (i + 1) / 2 + i] = d; inv[i] = 1.0 / d; }
            else L[i * (i + 1) / 2 + j] = s * inv[j];
        }
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        double s = bvec[i];
#pragma unroll
        for (int k = 0; k < i; k++) s -= L[i * (i + 1) / 2 + k] * y[k];
        y[i] = s * inv[i];
    }
#pragma unroll
    for (int i = 5; i >= 0; i--) {
        double s = y[i];
#pragma unroll
        for (int k = i + 1; k < 6; k++) s -= L[k * (k + 1) / 2 + i] * xo[k];
        xo[i] = s * inv[i];
    }
    return true;
}

__device__ __forceinline__ void manifold_plus(const double *x, const double *d, double *o)
{
    const double nd = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (nd > 0.0) {
        const double sc = sin(nd) / nd;
        const double ax = sc * d[0], ay = sc * d[1], az = sc * d[2], aw = cos(nd);
        const double bx = x[0], by = x[1], bz = x[2], bw = x[3];
        o[3] = aw * bw - ax * bx - ay * by - az * bz;
        o[0] = aw * bx + ax * bw + ay * bz - az * by;
        o[1] = aw * by + ay * bw + az * bx - ax * bz;
        o[2] = aw * bz + az * bw + ax * by - ay * bx;
    } else { o[0] = x[0]; o[1] = x[1]; o[2] = x[2]; o[3] = x[3]; }
    o[4] = x[4] + d[3]; o[5] = x[5] + d[4]; o[6] = x[6] + d[5];
}

__device__ __forceinline__ double norm7(const double *x)
{
    double s = 0.0;
    for (int i = 0; i < 7; i++) s += x[i] * x[i];
    return sqrt(s);
}

__device__ __forceinline__ void unpack_sym(const double *Hu, double *H)
{
    int t = 0;
    for (int i = 0; i < 6; i++)
        for (int j = i; j < 6; j++) { H[i * 6 + j] = Hu[t]; H[j * 6 + i] = Hu[t]; t++; }
}

constexpr int kLmRecLds = 4 * kMaxQueries * 16;   // the chain's records in LDS
#ifndef LMONO_CF_T
#define LMONO_CF_T 128
#endif
constexpr int kThinBlocks = kMaxQueries / LMONO_CF_T;     // = kCfBlocks of corr_flat.hip: feature qi belongs to workgroup qi % kThinBlocks

// One kLmT-thread workgroup per chain.  Every thread runs the (uniform) trust-region control flow redundantly on the
// block-reduced sums; residual blocks come from the 64-B records written by k_correspond.
// k_lm_solve's evaluation: the chain's records staged in LDS
struct LmRecEval {
    const float4 *srec; int n_edge, nq; double (*s_red)[28]; double *s_sum;
    template <bool kJac> __device__ __forceinline__ void run(const double *x) const { evaluate_block<kJac>(srec, n_edge, nq, x, s_red, s_sum); }
};

// Ceres' trust-region loop (Levenberg-Marquardt, DENSE_QR restated on the 6 x 6 normal equations, <= 4 iterations) over
// an evaluation functor: ev.run<kJac>(pose) leaves the sums H (21), g (6), cost in s_sum (with the workgroup barriers that takes).  On
// entry s_sum holds the linearisation at x; on return x is the accepted pose, the same in every thread.  Shared by k_lm_solve
// (records in LDS, 256 threads) and k_odom_chain (records in registers, 1024 threads).
template <class Eval>
__device__ __forceinline__ void lm_trust_region(const Eval &ev, double *x, double *s_cur, double *s_sum, int n_used, int &iter)
{
    const int tid = threadIdx.x;
#ifndef LMONO_LM_MAX_ITER
#define LMONO_LM_MAX_ITER 4          // ceres max_num_iterations of laserOdometry (any other value is a TIMING experiment: results change)
#endif
    const int max_iter = LMONO_LM_MAX_ITER;
    const double function_tol = 1e-6, gradient_tol = 1e-10, parameter_tol = 1e-8;
    const double min_rel_decrease = 1e-3, min_diag = 1e-6, max_diag = 1e32, max_radius = 1e16, min_radius = 1e-32;
    double radius = 1e4, decrease_factor = 2.0;
    bool reuse_diagonal = false;
    int invalid_steps = 0;
    iter = 0;
    // The accepted linearisation (H, g, cost) lives in LDS (s_cur), a candidate's in s_sum: every thread runs the uniform trust-region
    // arithmetic on the same LDS numbers, and only the scaled, damped system of the current iteration is ever in registers -- with
    // H, Hs, a second matrix for the factor and two accumulator sets per thread the kernel needed ~350 registers and could not run
    // two waves per SIMD.
    if (tid < 28) s_cur[tid] = s_sum[tid];
    __syncthreads();
    double x_cost = s_cur[27];
    double scale[6], diag[6];
    double gmax = 0.0;
    for (int i = 0; i < 6; i++) gmax = fmax(gmax, fabs(s_cur[21 + i]));
    if (n_used > 0 && gmax > gradient_tol) {
        double x_norm = norm7(x);
        for (int i = 0; i < 6; i++) scale[i] = 1.0 / (1.0 + sqrt(s_cur[sym6(i, i)]));
        while (iter < max_iter) {
            iter++;
            double gs[6], A[21], stepv[6];
            // lower triangle of A = Hs = H scale_i scale_j, then + diag / radius on the diagonal
#pragma unroll
            for (int i = 0; i < 6; i++) {
                gs[i] = s_cur[21 + i] * scale[i];
#pragma unroll
                for (int j = 0; j <= i; j++) A[i * (i + 1) / 2 + j] = s_cur[sym6(i, j)] * scale[i] * scale[j];
            }
            if (!reuse_diagonal)
                for (int i = 0; i < 6; i++) { double d = A[i * (i + 1) / 2 + i]; d = d < min_diag ? min_diag : d; d = d > max_diag ? max_diag : d; diag[i] = d; }
            { const double ir = 1.0 / radius; for (int i = 0; i < 6; i++) A[i * (i + 1) / 2 + i] += diag[i] * ir; }
            bool ok = chol_solve6_packed(A, gs, stepv);
            for (int i = 0; i < 6; i++) if (!isfinite(stepv[i])) ok = false;
            double model_change = 0.0;
            if (ok) {
                for (int i = 0; i < 6; i++) stepv[i] = -stepv[i];
                double dg = 0.0, dHd = 0.0;
                for (int i = 0; i < 6; i++) { dg += stepv[i] * gs[i]; for (int j = 0; j < 6; j++) dHd += stepv[i] * (s_cur[sym6(i, j)] * scale[i] * scale[j]) * stepv[j]; }
                model_change = -(dg + 0.5 * dHd);
            }
            if (!ok || !(model_change > 0.0)) {
                if (++invalid_steps >= 5) break;
                radius *= 0.5; reuse_diagonal = true;
                continue;
            }
            invalid_steps = 0;
            double delta[6], cand[7];
            for (int i = 0; i < 6; i++) delta[i] = stepv[i] * scale[i];
            manifold_plus(x, delta, cand);
            // The candidate is evaluated WITH its Jacobian (it is accepted almost always, and then this is the linearisation of
            // the next iteration -- the same numbers a separate pass would give).  The last iteration only needs the cost, but a
            // second, cost-only copy of the sweep inside this loop costs the kernel its second wave per SIMD (81 spilled registers).
            const bool last = iter == max_iter;
            if (last) ev.template run<false>(cand);
            else ev.template run<true>(cand);
            const double cand_cost = s_sum[27];
            double sn = 0.0;
            for (int i = 0; i < 7; i++) sn += (x[i] - cand[i]) * (x[i] - cand[i]);
            sn = sqrt(sn);
            if (sn <= parameter_tol * (x_norm + parameter_tol)) break;
            if (fabs(x_cost - cand_cost) <= function_tol * x_cost) break;
            const double rel = (x_cost - cand_cost) / model_change;
            if (rel > min_rel_decrease) {
                for (int i = 0; i < 7; i++) x[i] = cand[i];
                if (last) break;                 // nothing after the last accepted step is used
                x_norm = norm7(x);
                x_cost = cand_cost;
                __syncthreads();                 // every thread has read the old linearisation
                if (tid < 28) s_cur[tid] = s_sum[tid];
                __syncthreads();
                const double tt = 2.0 * rel - 1.0;
                double den = 1.0 - tt * tt * tt;
                if (den < 1.0 / 3.0) den = 1.0 / 3.0;
                radius = radius / den;
                if (radius > max_radius) radius = max_radius;
                decrease_factor = 2.0; reuse_diagonal = false;
                gmax = 0.0;
                for (int i = 0; i < 6; i++) gmax = fmax(gmax, fabs(s_cur[21 + i]));
                if (gmax <= gradient_tol) break;
            } else {
                radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
            }
            if (radius <= min_radius) break;
        }
    }
}

__global__ __launch_bounds__(kLmT) void k_lm_solve(BatchView b, OdomView o, int step, int outer, unsigned int *wl_reset)
{
    const int c = o.clist ? o.clist[o.chain0 + blockIdx.x] : o.chain0 + (int)blockIdx.x;
    if (wl_reset && blockIdx.x == 0 && threadIdx.x == 0) wl_reset[0] = 0u;      // the work list of the next correspondence launch starts empty
    int s;
    const int k = chain_scan(o, c, step, s);
    if (k < 0) return;
    __shared__ double s_red[kLmW][28], s_sum[28], s_cur[28];
    __shared__ int s_used[kLmW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_edge = b.feat_n[k * 4 + 0], nq = n_edge + b.feat_n[k * 4 + 2];
    const float4 *crec = o.crec + (size_t)c * kMaxQueries * 4;
    // the chain's residual-block records (64 B each, <= 144 KB) are read once and stay in LDS for the up to nine
    // evaluations of this launch; eight 16-B loads per thread in flight
    extern __shared__ __align__(16) float4 s_rec[];
    int n_used = 0;
    {
        // one record per thread and round: its four 16-B quarters are requested together, the pose-independent part of the
        // residual block is computed once (stage_block) and the block goes to LDS as (cp, kind) + six doubles
        double2 *sp = (double2 *)s_rec;
        const bool thin = lead_in_thinned(o, k, s);
        for (int qi = tid; qi < nq; qi += kLmT) {
            float4 cp = crec[qi * 4];
            const float4 A = crec[qi * 4 + 1], B = crec[qi * 4 + 2], Cc = crec[qi * 4 + 3];
            if (thin && (qi % kThinBlocks) % kThinStride != 0) cp.w = 0.f;       // not searched in this launch: a stale record
            double P[6] = { 0.0, 0.0, 0.0, 0.0, 0.0, 0.0 };
            if (__float_as_int(cp.w) != 0) { stage_block(cp, A, B, Cc, P); n_used++; }
            s_rec[qi] = cp;
            sp[kMaxQueries + qi] = make_double2(P[0], P[1]);
            sp[2 * kMaxQueries + qi] = make_double2(P[2], P[3]);
            sp[3 * kMaxQueries + qi] = make_double2(P[4], P[5]);
        }
    }
    __syncthreads();
    double x[7];
    for (int i = 0; i < 7; i++) x[i] = o.state[c * 8 + i];

    LmRecEval ev{ s_rec, n_edge, nq, s_red, s_sum };
    int iter = 0;
    ev.run<true>(x);
    n_used = wave_sum_i(n_used);
    if (lane == 0) s_used[wave] = n_used;
    __syncthreads();
    n_used = 0;
    for (int w = 0; w < kLmW; w++) n_used += s_used[w];
    lm_trust_region(ev, x, s_cur, s_sum, n_used, iter);
    if (tid == 0) {
        for (int i = 0; i < 7; i++) o.state[c * 8 + i] = x[i];
        if (o.repair && outer == 1) {
            // repair chain: the pair's stored increment came from a warm start that failed the boundary check (or from a chain
            // re-started behind one).  Replace it; once the new one agrees with the stored one the rest of the chain stands.
            int e0, e1;
            chain_bounds(o.first, o.n_scans, o.n_chains, c, e0, e1);
            const double res = boundary_residual(x, o.incr + (size_t)k * 7);
            for (int i = 0; i < 7; i++) o.incr[(size_t)k * 7 + i] = x[i];
            o.rstat[c * 4 + 1] += 1;
            // stop after kRepairAgree CONSECUTIVE agreeing pairs: one agreement within tol does not protect against a pair that amplifies a
            // tol-sized difference (a correspondence set on a knife edge: seen once on the held-out sequence, 1e-6 -> 1.9e-4 rad, 11 mm of ATE)
            int agree = o.rstat[c * 4 + 2] >> 1;
            agree = res <= o.tol ? agree + 1 : 0;
            o.rstat[c * 4 + 2] = 1 | (agree << 1);
            if (agree >= kRepairAgree || k + 1 >= e1) { o.rstat[c * 4] = 1; atomicSub(&o.rcount[1], 1u); }
        } else if (o.incr && outer == 1 && k >= s)
            for (int i = 0; i < 7; i++) o.incr[(size_t)k * 7 + i] = x[i];
        if (o.ws && !o.repair && outer == 1 && k == s - 1)
            for (int i = 0; i < 7; i++) o.ws[c * 8 + i] = x[i];       // the warm start of the chain's first owned pair
        if (o.lm_info) { o.lm_info[c * 4 + outer] = iter; o.lm_info[c * 4 + 2 + outer] = n_used; }
    }
}

// Boundary validation of the chained schedule.  Chain c > 0 warm-starts its first owned pair (s_c - 1, s_c) from ws[c] (its own
// lead-in's estimate of the pair before), the strictly sequential schedule from incr[s_c - 1], which chain c - 1 owns.  Where the two
// agree within tol chain c's increments are the sequential schedule's (to the sensitivity of one pair to its warm start); where they
// do not, the chain is flagged: its slot is re-started from incr[s_c - 1] (state, ws) and listed in rcount[2..] for the repair
// launches.  One workgroup; the list is in chain order.  resid (may be null): [n_chains] residual of this check per boundary.
// ext: incr[first - 1] holds the previous rank's last increment, so chain 0's boundary is checked as well.
__global__ __launch_bounds__(256) void k_boundary_check(OdomView o, double *resid, int first_round, int ext)
{
    __shared__ int s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    for (int base = 0; base < o.n_chains; base += 256) {
        const int c = base + threadIdx.x;
        bool flag = false;
        if (c < o.n_chains) {
            int s, e;
            chain_bounds(o.first, o.n_scans, o.n_chains, c, s, e);
            double res = 0.0;
            if ((c > 0 || ext) && s > 0) res = boundary_residual(o.ws + c * 8, o.incr + (size_t)(s - 1) * 7);
            if (resid) resid[c] = res;
            flag = res > o.tol;
            if (first_round) { o.rstat[c * 4 + 1] = 0; o.rstat[c * 4 + 3] = 0; }
            o.rstat[c * 4] = flag ? 0 : 1;
            o.rstat[c * 4 + 2] = flag ? 1 : 0;
            if (flag) {
                o.rstat[c * 4 + 3] += 1;
                for (int i = 0; i < 7; i++) { const double v = o.incr[(size_t)(s - 1) * 7 + i]; o.state[c * 8 + i] = v; o.ws[c * 8 + i] = v; }
            }
        }
        // ordered compaction: wave ballots + a running base
        const unsigned long long m = __ballot(flag);
        __shared__ int s_w[4];
        if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = __popcll(m);
        __syncthreads();
        int pre = s_n;
        for (int w = 0; w < (int)(threadIdx.x >> 6); w++) pre += s_w[w];
        if (flag) o.rcount[2 + pre + __popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull))] = (unsigned int)c;
        __syncthreads();
        if (threadIdx.x == 0) s_n += s_w[0] + s_w[1] + s_w[2] + s_w[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) { o.rcount[0] = (unsigned int)s_n; o.rcount[1] = (unsigned int)s_n; }
}

__global__ void k_odom_init(OdomView o)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < o.n_chains) {
        double *st = o.state + i * 8;
        st[0] = st[1] = st[2] = 0.0; st[3] = 1.0; st[4] = st[5] = st[6] = 0.0; st[7] = 0.0;
        if (o.ws) { double *w = o.ws + i * 8; w[0] = w[1] = w[2] = 0.0; w[3] = 1.0; w[4] = w[5] = w[6] = 0.0; w[7] = 0.0; }
    }
    if (i < o.n_scans) {
        double *r = o.incr + (size_t)i * 7;
        r[0] = r[1] = r[2] = 0.0; r[3] = 1.0; r[4] = r[5] = r[6] = 0.0;
    }
}

// Lead-in seeding across chains (round 6, LMONO_OPT_LEAD_SEED).  Every chain that does not start at the sequence's first scan begins its
// lead-in from the identity (A-LOAM's initial para_q / para_t) and needs `lead` pairs to forget it.  After the FIRST step of the main pass every chain
// holds an estimate of the vehicle's increment at its own start -- 256 samples of a motion that changes slowly along the sequence.  A chain's sample can be
// an outlier (clutter: a first pair from the identity that settled on wrong correspondences); its neighbours', 1.8 s away, then describe its motion
// better than it does itself.  k_lead_seed_median replaces the state of every seeded chain by the component-wise median of the samples of chains
// c - 1, c, c + 1 (quaternion signs aligned to w >= 0, re-normalised): a constant-velocity prior that costs no step.  Only lead-in states are touched
// (a chain whose first step is an owned pair keeps its state: its increment is a result); the boundary validation stays the arbiter of every result.
// Two launches: medians into ws (unused until a chain's last lead-in pair), then ws -> state.
__device__ __forceinline__ double median3(double a, double b, double c) { return fmax(fmin(a, b), fmin(fmax(a, b), c)); }
__device__ __forceinline__ bool chain_seeded(const OdomView &o, int c)
{
    int s, e;
    chain_bounds(o.first, o.n_scans, o.n_chains, c, s, e);
    const int begin = max(s - o.lead, 0);
    return begin + 1 < s - 1;        // its first step is a lead-in pair and not its last one (whose result the boundary check reads from ws)
}
__global__ void k_lead_seed_median(OdomView o)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= o.n_chains || !chain_seeded(o, c)) return;
    const int c0 = min(max(c - 1, 0), max(o.n_chains - 3, 0));
    double v[3][7];
    for (int j = 0; j < 3; j++) {
        const double *st = o.state + (size_t)min(c0 + j, o.n_chains - 1) * 8;
        const double sg = st[3] < 0.0 ? -1.0 : 1.0;
        for (int i = 0; i < 4; i++) v[j][i] = sg * st[i];
        for (int i = 4; i < 7; i++) v[j][i] = st[i];
    }
    double m[7], nn = 0.0;
    for (int i = 0; i < 7; i++) m[i] = median3(v[0][i], v[1][i], v[2][i]);
    for (int i = 0; i < 4; i++) nn += m[i] * m[i];
    nn = sqrt(nn);
    const bool ok = nn > 0.5 && nn == nn;
    for (int i = 0; i < 4; i++) o.ws[c * 8 + i] = ok ? m[i] / nn : o.state[c * 8 + i];
    for (int i = 4; i < 7; i++) o.ws[c * 8 + i] = ok && m[i] == m[i] ? m[i] : o.state[c * 8 + i];
}
__global__ void k_lead_seed_apply(OdomView o)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= o.n_chains || !chain_seeded(o, c)) return;
    for (int i = 0; i < 7; i++) o.state[c * 8 + i] = o.ws[c * 8 + i];
}

// poses[k - first] = incr[first] (+) incr[first+1] (+) ... (+) incr[k]  (t_w += q_w * t ; q_w = q_w * q).
// incr[0] is the identity, so first = 0 gives poses relative to scan 0; for first > 0 the result is relative to scan
// first-1 (the previous rank's last scan).  One wave: every lane composes a contiguous segment, an inclusive wave scan
// composes the segment totals, then every lane re-bases its segment (SE(3) composition is associative; the result
// differs from the strictly sequential product only by rounding, ~1e-16).
__device__ __forceinline__ void se3_compose(const double *a, const double *b, double *o)
{
    double rx, ry, rz;
    quat_rotate(a, b[4], b[5], b[6], rx, ry, rz);
    const double ax = a[0], ay = a[1], az = a[2], aw = a[3];
    const double bx = b[0], by = b[1], bz = b[2], bw = b[3];
    const double t0 = a[4] + rx, t1 = a[5] + ry, t2 = a[6] + rz;
    o[3] = aw * bw - ax * bx - ay * by - az * bz;
    o[0] = aw * bx + ax * bw + ay * bz - az * by;
    o[1] = aw * by + ay * bw + az * bx - ax * bz;
    o[2] = aw * bz + az * bw + ax * by - ay * bx;
    o[4] = t0; o[5] = t1; o[6] = t2;
}

__global__ __launch_bounds__(64) void k_pose_prefix(const double *incr, double *poses, int first, int n)
{
    const int lane = threadIdx.x;
    incr += (size_t)first * 7;
    n -= first;
    const int seg = (n + 63) / 64;
    const int k0 = lane * seg, k1 = min(k0 + seg, n);
    double acc[7] = { 0, 0, 0, 1, 0, 0, 0 };
    for (int k = k0; k < k1; k++) {
        double o[7];
        se3_compose(acc, incr + (size_t)k * 7, o);
        for (int i = 0; i < 7; i++) acc[i] = o[i];
        for (int i = 0; i < 7; i++) poses[(size_t)k * 7 + i] = acc[i];
    }
    // inclusive scan of the segment totals over the lanes (Hillis-Steele with SE(3) composition, left operand = lower lanes)
    double tot[7];
    for (int i = 0; i < 7; i++) tot[i] = acc[i];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        double left[7], o[7];
        for (int i = 0; i < 7; i++) left[i] = __shfl_up(tot[i], d);
        if (lane >= d) { se3_compose(left, tot, o); for (int i = 0; i < 7; i++) tot[i] = o[i]; }
    }
    double base[7];
    for (int i = 0; i < 7; i++) base[i] = __shfl_up(tot[i], 1);
    if (lane > 0) {
        for (int k = k0; k < k1; k++) {
            double cur[7], o[7];
            for (int i = 0; i < 7; i++) cur[i] = poses[(size_t)k * 7 + i];
            se3_compose(base, cur, o);
            for (int i = 0; i < 7; i++) poses[(size_t)k * 7 + i] = o[i];
        }
    }
}

// poses[k] <- base (+) poses[k] with base = bases[0] (+) ... (+) bases[n_bases-1]  (multi-GPU: cumulative
// transforms of the lower ranks, gathered with one RCCL all-gather)
__global__ void k_pose_rebase(const double *bases, int n_bases, double *poses, int n)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    double qw[4] = { 0, 0, 0, 1 }, tw[3] = { 0, 0, 0 };
    for (int j = 0; j <= n_bases; j++) {
        const double *q = j < n_bases ? bases + (size_t)j * 7 : poses + (size_t)k * 7;
        const double *t = q + 4;
        double rx, ry, rz;
        quat_rotate(qw, t[0], t[1], t[2], rx, ry, rz);
        tw[0] += rx; tw[1] += ry; tw[2] += rz;
        const double ax = qw[0], ay = qw[1], az = qw[2], aw = qw[3];
        const double bx = q[0], by = q[1], bz = q[2], bw = q[3];
        qw[3] = aw * bw - ax * bx - ay * by - az * bz;
        qw[0] = aw * bx + ax * bw + ay * bz - az * by;
        qw[1] = aw * by + ay * bw + az * bx - ax * bz;
        qw[2] = aw * bz + az * bw + ax * by - ay * bx;
    }
    double *p = poses + (size_t)k * 7;
    p[0] = qw[0]; p[1] = qw[1]; p[2] = qw[2]; p[3] = qw[3]; p[4] = tw[0]; p[5] = tw[1]; p[6] = tw[2];
}

} // namespace lmono
