// lmono_amd/csrc/frontend.hip -- gfx950 kernels of the LiDAR front end (A-LOAM scanRegistration;
// source absent from the reference tree, behavioural spec SURVEY.md Appendix A.1).
//
// Five kernel stages per batch of scans over HBM-resident SoA/float4 pools:
//   k_ring_*     filter, ring id, azimuth, stable counting sort into ring-major order: four tile-parallel launches (ends, tag,
//                offsets, scatter); every wave owns a 1024-point segment in both passes, no barriers inside the passes
//   k_curvature  one workgroup per 1024-point tile: LDS tile with +-5 halo, curvature and neighbour-gap flags
//   k_select     one wave per (scan, ring): per-sector edge/planar selection as repeated 64-lane arg-max/arg-min (DPP
//                reductions) over register-resident (curvature, index) keys with neighbour suppression (no sort needed);
//                rings longer than the small LDS slice go through a work list to a second launch
//   k_voxel      one workgroup per (scan, ring): voxel-grid down-sampling of the less-flat points (run-length segments from
//                ballots, bucket sort of the segment keys in LDS; <9 slots> / <16 slots> instantiations + work list)
//   k_compact    one workgroup per scan: prefix sums over (ring, sector), compaction of the four clouds, line tables
// Arithmetic is float/double exactly as the CPU restatement evaluates it (compiled with -ffp-contract=off).
#include "batch.hpp"

namespace lmono {

__device__ __forceinline__ int ring_of(float angle, int n_lines, bool &discard)
{
    int id = 0;
    discard = false;
    if (n_lines == 16) {
        id = (int)((double)((angle + 15.0f) / 2.0f) + 0.5);
        if (id > n_lines - 1 || id < 0) discard = true;
    } else if (n_lines == 32) {
        id = (int)(((double)angle + 92.0 / 3.0) * 3.0 / 4.0);
        if (id > n_lines - 1 || id < 0) discard = true;
    } else {
        if ((double)angle >= -8.83)
            id = (int)((double)(2.0f - angle) * 3.0 + 0.5);
        else
            id = n_lines / 2 + (int)((-8.83 - (double)angle) * 2.0 + 0.5);
        if ((double)angle > 2.0 || (double)angle < -24.33 || id > 50 || id < 0) discard = true;
    }
    return id;
}

__device__ __forceinline__ bool point_valid(const float4 &p, float mr2)
{
    return isfinite(p.x) && isfinite(p.y) && isfinite(p.z) && !(p.x * p.x + p.y * p.y + p.z * p.z < mr2);
}

// ------------------------------------------------------------------------------------------------
// Ring sort = stable counting sort of a scan's valid points into ring-major order, as four tile-parallel launches (round 3: one
// workgroup per scan kept ~2 long workgroups per wave slot -- a third of the launch was its tail -- and made one scan alone take 1.1 ms):
//   k_ring_ends     per scan: first / last valid point -> start / end azimuth of the sweep
//   k_ring_tag      per 4096-point tile: ring id of every point, a [64]-ring histogram per 1024-point wave segment, the first point
//                   past the half sweep, the last valid point
//   k_ring_offsets  per scan: exclusive prefix of the segment histograms in segment order (= input order) per ring, ring_begin
//   k_ring_scatter  per tile: every wave walks its segment in input order with its own running counters (barrier-free multisplit)
// Segment g of scan s lives in row (off[s] >> 10) + s + g of seg_hist: rows of different scans never collide and need no table.
constexpr int kRtT = 256, kRtW = kRtT / 64;    // threads / waves of a tile workgroup
constexpr int kRtSeg = 1024;                   // points per wave segment (4 rounds of 256)
constexpr int kRtTile = kRtSeg * kRtW;
constexpr float kRingGuardDeg = 2e-3f;         // 20x the single-precision angle's error bound
constexpr float kHalfGuardRad = 4e-5f;         // 20x (the float subtractions near 3 pi / 2 round to 5e-7)

// points of scan s the ring sort looks at: the scan's slot, or the first n_limit of it (one-scan launches of the online stream: the rest of
// the slot is padding)
__device__ __forceinline__ int ring_scan_points(const BatchView &b, int s, int64_t off)
{
    const int n = (int)(b.off[s + 1] - off);
    return b.n_limit > 0 && b.n_limit < n ? b.n_limit : n;
}

__global__ __launch_bounds__(256) void k_ring_ends(BatchView b)
{
    const int s = b.scan0 + blockIdx.x;
    const int64_t off = b.off[s];
    const int n = ring_scan_points(b, s, off);
    const float4 *in = b.in + off;
    const int tid = threadIdx.x, lane = tid & 63;
    __shared__ int s_first;
    if (tid == 0) s_first = INT_MAX;
    __syncthreads();
    const float mr2 = b.min_range * b.min_range;
    // first valid point (it defines the sweep's start azimuth): forward in 256-point chunks until one holds a valid point -- almost always
    // the first.  The LAST valid point is found by k_ring_tag on its way through the scan (the low rings of a 64-line sensor end a scan with
    // thousands of points inside min_range; looking for it from here cost a full read of every scan, 1.9 ms of the round-3 front end).
    for (int c0 = 0; c0 < n; c0 += 256) {
        const int i = c0 + tid;
        int lf = (i < n && point_valid(in[i], mr2)) ? i : INT_MAX;
        lf = wave_min_i(lf);
        if (lane == 0 && lf != INT_MAX) atomicMin(&s_first, lf);
        __syncthreads();
        const int f = s_first;
        __syncthreads();                   // nobody is in the next chunk's atomicMin before everybody has read this one's result
        if (f != INT_MAX) break;
    }
    if (s_first == INT_MAX) {
        int *rb = b.ring_begin + s * 65;
        if (tid < 65) rb[tid] = 0;
        if (tid == 0) { b.n_cloud[s] = 0; b.scan_ends[s * 2] = -1; b.scan_ends[s * 2 + 1] = -1; b.scan_half[s] = INT_MAX; }
        return;
    }
    if (tid == 0) {
        const float4 p0 = in[s_first];
        b.scan_ori[s * 2] = (float)(-det_atan2((double)p0.y, (double)p0.x));
        b.scan_ends[s * 2] = s_first; b.scan_ends[s * 2 + 1] = -1;
        b.scan_half[s] = INT_MAX;
    }
}

__global__ __launch_bounds__(kRtT) void k_ring_tag(BatchView b)
{
    const int s = b.scan0 + blockIdx.y;
    const int64_t off = b.off[s];
    const int n = ring_scan_points(b, s, off);
    const int t0 = blockIdx.x * kRtTile;
    if (t0 >= n || b.scan_ends[s * 2] < 0) return;
    const float4 *in = b.in + off;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ int s_hist[kRtW][64];
    s_hist[wave][lane] = 0;
    __syncthreads();
    const float mr2 = b.min_range * b.min_range;
    const float startOri = b.scan_ori[s * 2];
    const int n_lines = b.n_lines;
    int lh = INT_MAX, ll = -1;
    const int c_lo = t0 + wave * kRtSeg, c_hi = min(c_lo + kRtSeg, n);
    // four points per thread and round, and the NEXT round's four requested before this round's are worked on (unconditional loads, index clamped
    // into the segment): a wave's segment is four rounds, each a load latency + ~450 instructions of angle work -- one behind the other they added up
    float4 nx[4];
#pragma unroll
    for (int q = 0; q < 4; q++) nx[q] = in[min(c_lo + 64 * q + lane, n - 1)];
    for (int r0 = c_lo; r0 < c_hi; r0 += 256) {
        float4 pq[4];
#pragma unroll
        for (int q = 0; q < 4; q++) pq[q] = nx[q];
#pragma unroll
        for (int q = 0; q < 4; q++) nx[q] = in[min(r0 + 256 + 64 * q + lane, n - 1)];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int i = r0 + 64 * q + lane;
            const float4 p = pq[q];
            int id = -1;
            if (i < c_hi && point_valid(p, mr2)) {
                ll = i;                        // i grows with q and r0: the lane's last valid point so far
                // Ring id and the half-sweep test decide between alternatives, so single-precision angles settle them whenever the
                // answer is the same over the whole error interval (ring_of is monotone in the angle; the half test has three
                // thresholds); the fp64 forms run only for the points that fall inside a guard band (and for x = y = 0).
                const float q2 = p.x * p.x + p.y * p.y;
                bool exact = !(q2 > 1e-30f);
                bool discard = false, past_half = false;
                if (!exact) {
                    const float af = atanf(p.z * __frsqrt_rn(q2)) * 57.29577951308232f;      // |error| < 1e-4 deg (2 ulp atanf, 1 ulp rsqrt)
                    bool d0, d1;
                    const int r0i = ring_of(af - kRingGuardDeg, n_lines, d0), r1i = ring_of(af + kRingGuardDeg, n_lines, d1);
                    if (d0 != d1 || (!d0 && r0i != r1i)) exact = true;
                    else { discard = d0; id = d0 ? -1 : r0i; }
                }
                if (!exact && !discard) {
                    const float dr = -atan2f(p.y, p.x) - startOri;                               // |error| < 2e-6 rad
                    float dw = dr;
                    if (dr < -(float)LM_PI_2) dw = dr + 2.0f * (float)LM_PI;
                    else if (dr > 1.5f * (float)LM_PI) dw = dr - 2.0f * (float)LM_PI;
                    if (fabsf(dr + (float)LM_PI_2) < kHalfGuardRad || fabsf(dr - 1.5f * (float)LM_PI) < kHalfGuardRad || fabsf(dw - (float)LM_PI) < kHalfGuardRad) exact = true;
                    else past_half = dw > (float)LM_PI;
                }
                if (exact) {
                    const float angle = (float)(det_atan((double)p.z / sqrt((double)(p.x * p.x + p.y * p.y))) * 180.0 / LM_PI);
                    const int r = ring_of(angle, n_lines, discard);
                    id = -1; past_half = false;
                    if (!discard) {
                        id = r;
                        float o1 = (float)(-det_atan2((double)p.y, (double)p.x));
                        if ((double)o1 < (double)startOri - LM_PI / 2.0) o1 = (float)((double)o1 + 2.0 * LM_PI);
                        else if ((double)o1 > (double)startOri + LM_PI * 3.0 / 2.0) o1 = (float)((double)o1 - 2.0 * LM_PI);
                        past_half = (double)(o1 - startOri) > LM_PI;
                    }
                }
                if (past_half) lh = min(lh, i);
            }
            if (i < c_hi) b.ring_tmp[off + i] = (int8_t)id;
            // histogram of the sub-tile by ballot matching (ring-major input: one or two rings per 64 points)
            unsigned long long rem = __ballot(id >= 0);
            while (rem) {
                const int src = __ffsll((long long)rem) - 1;
                const int k = __shfl(id, src);
                const unsigned long long m = __ballot(id == k);
                if (lane == src) s_hist[wave][k] += __popcll(m);
                rem &= ~m;
            }
        }
    }
    lh = wave_min_i(lh); ll = wave_max_i(ll);
    if (lane == 0 && lh != INT_MAX) atomicMin(&b.scan_half[s], lh);
    if (lane == 0 && ll >= 0) atomicMax(&b.scan_ends[s * 2 + 1], ll);
    __syncthreads();
    if (c_lo < n) b.seg_hist[((size_t)(off >> 10) + (size_t)s + (size_t)(c_lo >> 10)) * 64 + lane] = s_hist[wave][lane];
}

__global__ __launch_bounds__(64) void k_ring_offsets(BatchView b)
{
    const int s = b.scan0 + blockIdx.x;
    if (b.scan_ends[s * 2] < 0) return;
    const int64_t off = b.off[s];
    const int n = ring_scan_points(b, s, off);
    const int nseg = (n + kRtSeg - 1) / kRtSeg, r = threadIdx.x;
    if (r == 0) {                                  // end azimuth of the sweep from the last valid point k_ring_tag found
        const float4 p1 = b.in[off + b.scan_ends[s * 2 + 1]];
        const float startOri = b.scan_ori[s * 2];
        float endOri = (float)((double)(float)(-det_atan2((double)p1.y, (double)p1.x)) + 2.0 * LM_PI);
        if ((double)(endOri - startOri) > 3.0 * LM_PI) endOri = (float)((double)endOri - 2.0 * LM_PI);
        else if ((double)(endOri - startOri) < LM_PI) endOri = (float)((double)endOri + 2.0 * LM_PI);
        b.scan_ori[s * 2 + 1] = endOri;
    }
    int *h = b.seg_hist + ((size_t)(off >> 10) + (size_t)s) * 64 + r;
    int run = 0;
    int g = 0;
    for (; g + 8 <= nseg; g += 8) {              // eight rows in flight per round trip
        int t[8];
#pragma unroll
        for (int u = 0; u < 8; u++) t[u] = h[(size_t)(g + u) * 64];
#pragma unroll
        for (int u = 0; u < 8; u++) { h[(size_t)(g + u) * 64] = run; run += t[u]; }
    }
    for (; g < nseg; g++) { const int t = h[(size_t)g * 64]; h[(size_t)g * 64] = run; run += t; }
    // exclusive scan of the ring totals over the 64 lanes
    int incl = run;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (r >= o) incl += v; }
    int *rb = b.ring_begin + s * 65;
    rb[r] = incl - run;
    if (r == 63) { rb[64] = incl; b.n_cloud[s] = incl; }
}

__global__ __launch_bounds__(kRtT) void k_ring_scatter(BatchView b)
{
    const int s = b.scan0 + blockIdx.y;
    const int64_t off = b.off[s];
    const int n = ring_scan_points(b, s, off);
    const int t0 = blockIdx.x * kRtTile;
    if (t0 >= n || b.scan_ends[s * 2] < 0) return;
    const float4 *in = b.in + off;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ int s_pos[kRtW][64];            // next output position of (wave, ring)
    __shared__ double s_atan[17];              // kAtanTab in LDS: the azimuth of every kept point looks it up per lane
    if (tid < 17) s_atan[tid] = kAtanTab[tid];
    const int c_lo = t0 + wave * kRtSeg, c_hi = min(c_lo + kRtSeg, n);
    s_pos[wave][lane] = c_lo < n ? b.seg_hist[((size_t)(off >> 10) + (size_t)s + (size_t)(c_lo >> 10)) * 64 + lane] + b.ring_begin[s * 65 + lane] : 0;
    __syncthreads();
    const float startOri = b.scan_ori[s * 2], endOri = b.scan_ori[s * 2 + 1];
    const int half = b.scan_half[s];
    for (int r0 = c_lo; r0 < c_hi; r0 += 256) {
        int id[4], dstp[4];
        // unconditional loads (index clamped into the segment): behind their guards the four byte loads were waited for one by one
        signed char idb[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { const int i = r0 + 64 * q + lane; idb[q] = b.ring_tmp[off + min(i, c_hi - 1)]; }
#pragma unroll
        for (int q = 0; q < 4; q++) { const int i = r0 + 64 * q + lane; id[q] = (i < c_hi) ? (int)idb[q] : -1; }
        float4 pq[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int i = r0 + 64 * q + lane;
            pq[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (id[q] >= 0) pq[q] = in[i];
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            dstp[q] = 0;
            unsigned long long rem = __ballot(id[q] >= 0);
            while (rem) {
                const int src = __ffsll((long long)rem) - 1;
                const int k = __shfl(id[q], src);
                const unsigned long long m = __ballot(id[q] == k);
                int base = 0;
                if (lane == src) { base = s_pos[wave][k]; s_pos[wave][k] = base + __popcll(m); }
                base = __shfl(base, src);
                if (id[q] == k) dstp[q] = base + __popcll(m & ((1ull << lane) - 1ull));
                rem &= ~m;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int i = r0 + 64 * q + lane;
            if (id[q] >= 0) {
                const float4 p = pq[q];
                float ori = (float)(-det_atan2((double)p.y, (double)p.x, s_atan));      // the azimuth itself is a result (relTime): always the fp64 form
                if (i <= half) {
                    if ((double)ori < (double)startOri - LM_PI / 2.0) ori = (float)((double)ori + 2.0 * LM_PI);
                    else if ((double)ori > (double)startOri + LM_PI * 3.0 / 2.0) ori = (float)((double)ori - 2.0 * LM_PI);
                } else {
                    ori = (float)((double)ori + 2.0 * LM_PI);
                    if ((double)ori < (double)endOri - LM_PI * 3.0 / 2.0) ori = (float)((double)ori + 2.0 * LM_PI);
                    else if ((double)ori > (double)endOri + LM_PI / 2.0) ori = (float)((double)ori - 2.0 * LM_PI);
                }
                const float relTime = (ori - startOri) / (endOri - startOri);
                const float inten = (float)((double)id[q] + 0.1 * (double)relTime);
                b.cloud[off + dstp[q]] = make_float4(p.x, p.y, p.z, inten);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
constexpr int kCurvTile = 1024;

__global__ __launch_bounds__(256) void k_curvature(BatchView b)
{
    const int s = b.scan0 + blockIdx.y;
    const int n = b.n_cloud[s];
    const int t0 = blockIdx.x * kCurvTile;
    if (t0 >= n) return;
    const int64_t off = b.off[s];
    const float4 *c = b.cloud + off;
    __shared__ float sx[kCurvTile + 10], sy[kCurvTile + 10], sz[kCurvTile + 10];
    for (int k = threadIdx.x; k < kCurvTile + 10; k += 256) {
        const int i = t0 - 5 + k;
        float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i >= 0 && i < n) p = c[i];
        sx[k] = p.x; sy[k] = p.y; sz[k] = p.z;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < kCurvTile; k += 256) {
        const int i = t0 + k;
        if (i >= n) break;
        const int q = k + 5;
        float cv = 0.f;
        if (i >= 5 && i < n - 5) {
            const float dx = sx[q - 5] + sx[q - 4] + sx[q - 3] + sx[q - 2] + sx[q - 1] - 10 * sx[q] + sx[q + 1] + sx[q + 2] + sx[q + 3] + sx[q + 4] + sx[q + 5];
            const float dy = sy[q - 5] + sy[q - 4] + sy[q - 3] + sy[q - 2] + sy[q - 1] - 10 * sy[q] + sy[q + 1] + sy[q + 2] + sy[q + 3] + sy[q + 4] + sy[q + 5];
            const float dz = sz[q - 5] + sz[q - 4] + sz[q - 3] + sz[q - 2] + sz[q - 1] - 10 * sz[q] + sz[q + 1] + sz[q + 2] + sz[q + 3] + sz[q + 4] + sz[q + 5];
            cv = dx * dx + dy * dy + dz * dz;
        }
        b.curv[off + i] = cv;
        unsigned char g = 0;
        if (i + 1 < n) {
            const float gx = sx[q + 1] - sx[q], gy = sy[q + 1] - sy[q], gz = sz[q + 1] - sz[q];
            g = ((double)(gx * gx + gy * gy + gz * gz) > 0.05) ? 1 : 0;
        }
        b.gap[off + i] = g;
    }
}

// ------------------------------------------------------------------------------------------------
// k_select: one WAVE per (scan, ring); 4 rings per 256-thread workgroup, no workgroup barriers.
// The reference sorts every sector by curvature and walks the sorted order taking the next un-suppressed point
// (<= 20 largest with c > 0.1, then <= 4 smallest with c < 0.1).  Walking a sorted order and repeatedly taking the
// extremum of the not-yet-suppressed points visit the same points in the same order, so no sort is needed: each lane
// keeps its sector elements (curvature, suppressed flag) in registers and a pick is one 64-lane arg-max / arg-min
// over the packed key (curvature bits, index), which also reproduces the (curvature, index) tie order of the sort.
constexpr int kSelMaxPerLane = (kRingCap / 6 + 1 + 63) / 64;   // 11 elements per lane for the largest legal sector
constexpr int kSelWaveLds = 2 * kRingCap;                      // picked, gap bytes of one ring (unfused flavour; sel_slice_bytes below)

// key of a sector element: hi = curvature bits, lo = (index << 8 | suppression reach).  Ordering by (hi, lo) = ordering by
// (curvature, index), so the arg-max / arg-min over keys reproduces the sort's tie order; the winner's reach rides along.
// A pick is ONE 32-bit wave reduction: the curvature decides; the winner's low word is read from its lane when a single lane holds the
// extremum (the rule: equal curvatures in two lanes are rare), a second reduction over the low words settles the ties otherwise.
__device__ __forceinline__ unsigned int select_lo(int idx, unsigned int reach) { return ((unsigned int)idx << 8) | reach; }

// one sector of one ring, M register slots per lane (element m of a lane is ring-local index sp + lane + 64 m)
#ifndef LMONO_SEL_COMPACT
#define LMONO_SEL_COMPACT 1
#endif
constexpr int kSelScratch = 64 * 8;      // per wave: one (curvature bits, index | reach) pair per lane, the compacted candidates of a sector

// kFused (LMONO_FUSE_CURV_SELECT, round 5): the curvature is computed here, from the sector's points staged in LDS (xs / ys / zs: staged index 0 =
// ring point sp - 5), written to `curv` (the API returns it) instead of read from it, and `gap` is the SECTOR's bit mask (bit k = gap flag of ring
// point sp - 5 + k) instead of the ring's.  Same expressions in the same order as k_curvature.
template <int M, bool kFused>
__device__ __forceinline__ void select_sector(int lane, int j, int sp, int slen, int rbeg, float *curv, unsigned char *picked,
                                              signed char *label, const unsigned long long *gap, int *sel_sh, int *sel_sh_n, int *sel_fl, int *sel_fl_n,
                                              unsigned int *scratch, const float *sx = nullptr, const float *sy = nullptr, const float *sz = nullptr)
{
    unsigned int kh[M], kl[M];
    unsigned int dead = 0, big = 0, small = 0;     // bit m: suppressed / out of range; curvature > 0.1; < 0.1
    int n_big = 0;                                 // live points with c > 0.1 (wave-uniform), compacted into `scratch` as they are loaded
#pragma unroll
    for (int m = 0; m < M; m++) {
        const int e = lane + 64 * m;
        kh[m] = 0u; kl[m] = 0u;
        bool cand = false;
        if (e < slen) {
            const int i = sp + e;
            float c;
            if (kFused) {
                const int q = e + 5;
                const float dx = sx[q - 5] + sx[q - 4] + sx[q - 3] + sx[q - 2] + sx[q - 1] - 10 * sx[q] + sx[q + 1] + sx[q + 2] + sx[q + 3] + sx[q + 4] + sx[q + 5];
                const float dy = sy[q - 5] + sy[q - 4] + sy[q - 3] + sy[q - 2] + sy[q - 1] - 10 * sy[q] + sy[q + 1] + sy[q + 2] + sy[q + 3] + sy[q + 4] + sy[q + 5];
                const float dz = sz[q - 5] + sz[q - 4] + sz[q - 3] + sz[q - 2] + sz[q - 1] - 10 * sz[q] + sz[q + 1] + sz[q + 2] + sz[q + 3] + sz[q + 4] + sz[q + 5];
                c = dx * dx + dy * dy + dz * dz;
                curv[i] = c;
            } else c = curv[i];
            // suppression reach: the neighbour walk marks l = 1..5 forward while the gap before point i+l is short, and the same backward --
            // the zero runs above / below bit i of the ring's gap BIT mask (two independent 8-byte LDS reads; as bytes the two walks were up to
            // ten dependent single-byte reads per element, the latency that bound the kernel).  i >= 5 inside a sector.
            const int jw = kFused ? e : i - 5;
            const unsigned long long wa = gap[jw >> 6], wb = gap[(jw >> 6) + 1];
            const int sh = jw & 63;
            const unsigned int win = (unsigned int)((wa >> sh) | (sh ? wb << (64 - sh) : 0ull));      // bit k = gap flag of point i - 5 + k
            const int lf = min(5, __ffs((int)((win >> 5) | 32u)) - 1);
            const int lb = min(5, __clz((int)((win & 31u) << 27)));
            kh[m] = __float_as_uint(c); kl[m] = select_lo(i, (unsigned int)(lf | (lb << 4)));
            const bool pk = picked[i] != 0;
            if (pk) dead |= 1u << m;
            if ((double)c > 0.1) { big |= 1u << m; cand = !pk; }
            if ((double)c < 0.1) small |= 1u << m;
        } else dead |= 1u << m;
#if LMONO_SEL_COMPACT
        {
            const unsigned long long mk = __ballot(cand);
            const int pos = n_big + __popcll(mk & ((1ull << lane) - 1ull));
            if (cand && pos < 64) { scratch[2 * pos] = kh[m]; scratch[2 * pos + 1] = kl[m]; }
            n_big += __popcll(mk);
        }
#endif
    }
    // ---- largest curvature first: <= 2 sharp, <= 20 less sharp
    int largest = 0;
#if LMONO_SEL_COMPACT
    // Only the points with c > 0.1 that are not suppressed yet can be picked here -- a few dozen of a sector's ~300.  When they fit one per lane
    // they are compacted (ballot ranks, through 512 B of LDS): a pick then costs one comparison per lane + the wave reduction instead of a pass
    // over the lane's M slots.  Same live set, same (curvature, index) order: the same picks.  The slot bitmask `dead` is kept up to date for the
    // flat picks below.
    const bool compact = n_big <= 64;
    if (compact) {
        __builtin_amdgcn_wave_barrier();           // one wave: its LDS operations execute in order; the barrier keeps the compiler from reordering them
        unsigned int ch = 0u, cl = 0u;
        if (lane < n_big) { ch = scratch[2 * lane]; cl = scratch[2 * lane + 1]; }
        __builtin_amdgcn_wave_barrier();           // the next sector's writes stay behind these reads
        while (true) {
            const unsigned int mh = wave_max_u32_uniform(ch);
            if (mh == 0u) break;
            const unsigned long long who = __ballot(ch == mh);
            unsigned int lo;
            if ((who & (who - 1ull)) == 0ull) lo = (unsigned int)__builtin_amdgcn_readlane((int)cl, __ffsll((long long)who) - 1);
            else lo = wave_max_u32_uniform(ch == mh ? cl : 0u);
            const int pind = (int)(lo >> 8), lf = (int)(lo & 15u), lb = (int)((lo >> 4) & 15u);
            largest++;
            if (largest > 20) break;
            if (lane == 0) { sel_sh[j * 20 + largest - 1] = rbeg + pind; label[pind] = largest <= 2 ? 2 : 1; }
            if (lane <= lf + lb) picked[pind - lb + lane] = 1;
            // the compacted candidate of this lane is suppressed iff its index lies in [pind - lb, pind + lf]
            if ((unsigned int)((int)(cl >> 8) - (pind - lb)) <= (unsigned int)(lb + lf)) ch = 0u;
            {
                const int a = pind - lb - sp - lane;                 // slot m is hit iff a <= 64 m <= a + lb + lf
                const int m0 = (a + 63) >> 6;
                if (m0 >= 0 && m0 < M && 64 * m0 <= a + lb + lf) dead |= 1u << m0;
            }
        }
    }
    while (!compact) {
#else
    while (true) {
#endif
        const unsigned int live = big & ~dead;
        // the lane's own best: slots ascend in index, so ">=" keeps the larger index among equal curvatures (live curvatures are > 0.1: 0 = none)
        unsigned int bh = 0u, bl = 0u;
#pragma unroll
        for (int m = 0; m < M; m++) {
            const unsigned int h = kh[m] & (unsigned int)((int)(live << (31 - m)) >> 31);
            const bool t = h >= bh;
            bh = t ? h : bh; bl = t ? kl[m] : bl;
        }
        const unsigned int mh = wave_max_u32_uniform(bh);
        if (mh == 0u) break;
        const unsigned long long who = __ballot(bh == mh);
        unsigned int lo;
        if ((who & (who - 1ull)) == 0ull) lo = (unsigned int)__builtin_amdgcn_readlane((int)bl, __ffsll((long long)who) - 1);
        else lo = wave_max_u32_uniform(bh == mh ? bl : 0u);
        const int pind = (int)(lo >> 8), lf = (int)(lo & 15u), lb = (int)((lo >> 4) & 15u);
        largest++;
        if (largest > 20) break;
        if (lane == 0) { sel_sh[j * 20 + largest - 1] = rbeg + pind; label[pind] = largest <= 2 ? 2 : 1; }
        if (lane <= lf + lb) picked[pind - lb + lane] = 1;
        {
            // the suppressed range [pind - lb, pind + lf] is shorter than 64, so it meets at most one slot of this lane
            const int a = pind - lb - sp - lane;                 // slot m is hit iff a <= 64 m <= a + lb + lf
            const int m0 = (a + 63) >> 6;
            if (m0 >= 0 && m0 < M && 64 * m0 <= a + lb + lf) dead |= 1u << m0;
        }
    }
    if (lane == 0) sel_sh_n[j] = largest > 20 ? 20 : largest;
    // ---- smallest curvature first: <= 4 flat
    int smallest = 0;
    while (true) {
        const unsigned int live = small & ~dead;
        // "<" keeps the smaller index among equal curvatures; a dead slot reads as ~0 (a live curvature is < 0.1)
        unsigned int bh = ~0u, bl = ~0u;
#pragma unroll
        for (int m = 0; m < M; m++) {
            const unsigned int h = kh[m] | ~(unsigned int)((int)(live << (31 - m)) >> 31);
            const bool t = h < bh;
            bh = t ? h : bh; bl = t ? kl[m] : bl;
        }
        const unsigned int mh = wave_min_u32_uniform(bh);
        if (mh == ~0u) break;
        const unsigned long long who = __ballot(bh == mh);
        unsigned int lo;
        if ((who & (who - 1ull)) == 0ull) lo = (unsigned int)__builtin_amdgcn_readlane((int)bl, __ffsll((long long)who) - 1);
        else lo = wave_min_u32_uniform(bh == mh ? bl : ~0u);
        const int pind = (int)(lo >> 8), lf = (int)(lo & 15u), lb = (int)((lo >> 4) & 15u);
        if (lane == 0) { label[pind] = -1; sel_fl[j * 4 + smallest] = rbeg + pind; }
        smallest++;
        if (smallest >= 4) break;
        if (lane <= lf + lb) picked[pind - lb + lane] = 1;
        {
            // the suppressed range [pind - lb, pind + lf] is shorter than 64, so it meets at most one slot of this lane
            const int a = pind - lb - sp - lane;                 // slot m is hit iff a <= 64 m <= a + lb + lf
            const int m0 = (a + 63) >> 6;
            if (m0 >= 0 && m0 < M && 64 * m0 <= a + lb + lf) dead |= 1u << m0;
        }
    }
    if (lane == 0) sel_fl_n[j] = smallest;
}

// wave-uniform values as scalars (the compiler cannot prove uniformity of what is loaded through a per-wave index)
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ long long uni64(long long v)
{
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)v), hi = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)((unsigned long long)v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}
// (through address space 1: the integer round trip would otherwise leave a generic pointer and flat_ loads / stores behind)
template <typename T> __device__ __forceinline__ T *uni_ptr(T *p)
{
    typedef __attribute__((address_space(1))) T GT;
    return (T *)(GT *)uni64((long long)p);
}

#ifndef LMONO_FUSE_CURV_SELECT
#define LMONO_FUSE_CURV_SELECT 0
#endif
// LDS slice of one wave for rings of up to `cap` points.  Unfused: picked bytes + the ring's gap bit mask (cap bytes reserved).  Fused with the
// curvature: picked bytes + the staged x / y / z of one sector and its halo + the sector's gap words.
__host__ __device__ constexpr int sel_sector_cap(int cap) { return cap / 6 + 2; }
__host__ __device__ constexpr int sel_slice_bytes(int cap)
{
    return LMONO_FUSE_CURV_SELECT ? ((cap + 15) & ~15) + 3 * 4 * ((sel_sector_cap(cap) + 12 + 3) & ~3) + 8 * ((sel_sector_cap(cap) + 12) / 64 + 2) : 2 * cap;
}
// curvature of the cloud points first .. first + count - 1 of a scan straight from HBM (fused flavour: the points at a ring's ends, which no sector
// holds, and rings that take no part in the selection) -- k_curvature's expressions
__device__ __forceinline__ void curv_points_global(const float4 *c, float *curv, int n, int first, int count, int lane)
{
    for (int k = lane; k < count; k += 64) {
        const int i = first + k;
        float cv = 0.f;
        if (i >= 5 && i < n - 5) {
            float px[11], py[11], pz[11];
#pragma unroll
            for (int d = 0; d < 11; d++) { const float4 p = c[i - 5 + d]; px[d] = p.x; py[d] = p.y; pz[d] = p.z; }
            const float dx = px[0] + px[1] + px[2] + px[3] + px[4] - 10 * px[5] + px[6] + px[7] + px[8] + px[9] + px[10];
            const float dy = py[0] + py[1] + py[2] + py[3] + py[4] - 10 * py[5] + py[6] + py[7] + py[8] + py[9] + py[10];
            const float dz = pz[0] + pz[1] + pz[2] + pz[3] + pz[4] - 10 * pz[5] + pz[6] + pz[7] + pz[8] + pz[9] + pz[10];
            cv = dx * dx + dy * dy + dz * dz;
        }
        curv[i] = cv;
    }
}

// one ring by one wave; cap = ring points the wave's LDS slice (sel_slice_bytes(cap) at smem_w) can hold
__device__ __forceinline__ void select_ring(BatchView &b, int r, int s, int lane, unsigned char *smem_w, int cap, bool may_defer, unsigned int *scratch)
{
    constexpr bool kFused = LMONO_FUSE_CURV_SELECT != 0;
    r = uni(r); s = uni(s);
    const int64_t off = uni64(b.off[s]);
    const int *rb = b.ring_begin + s * 65;
    const int rbeg = uni(rb[r]), rend = uni(rb[r + 1]), len = rend - rbeg;
    const int S = rbeg + 5, E = rend - 6;
    int *sel_sh = uni_ptr(b.sel_sharp + (size_t)((s * 64 + r) * kSectors) * 20);
    int *sel_sh_n = uni_ptr(b.sel_sharp_n + (s * 64 + r) * kSectors);
    int *sel_fl = uni_ptr(b.sel_flat + (size_t)((s * 64 + r) * kSectors) * 4);
    int *sel_fl_n = uni_ptr(b.sel_flat_n + (s * 64 + r) * kSectors);
    const float4 *cl = uni_ptr(b.cloud + off);
    const int n_cloud = kFused ? uni(b.n_cloud[s]) : 0;
    if (E - S < 6 || len > kRingCap) {
        if (lane < kSectors) { sel_sh_n[lane] = 0; sel_fl_n[lane] = 0; }
        if (lane == 0 && len > kRingCap) atomicOr(&b.status[s], kStatusRingOverflow);
        for (int i = lane; i < len; i += 64) b.label[off + rbeg + i] = 0;
        if (kFused) curv_points_global(cl, uni_ptr(b.curv + off), n_cloud, rbeg, len, lane);
        return;
    }
    if (len > cap) {
        // longer than this launch's LDS slice: left to the second launch (full-size slices, small grid over the work list)
        if (may_defer && lane == 0) b.sel_todo[1 + atomicAdd(&b.sel_todo[0], 1)] = (s << 6) | r;
        return;
    }
    // labels go straight to HBM (zeroed here, ~144 single-byte stores per ring afterwards): the LDS slice only holds the
    // picked and gap bytes
    unsigned char *picked = smem_w;
    signed char *label = uni_ptr((signed char *)(b.label + off + rbeg));
    float *curv = uni_ptr(b.curv + off + rbeg);
    const int span = E - S;
    if (!kFused) {
        // gap flags of the ring as a bit mask, one 64-bit word per 64 points (+ one word of padding read by the last elements' windows)
        unsigned long long *gap = (unsigned long long *)(picked + cap);
        for (int i0 = 0; i0 < len + 64; i0 += 64) {
            const int i = i0 + lane;
            unsigned char g = 0;
            if (i < len) { picked[i] = 0; label[i] = 0; g = b.gap[off + rbeg + i]; }
            const unsigned long long mk = __ballot(g != 0);
            if (lane == 0) gap[i0 >> 6] = mk;
        }
        __builtin_amdgcn_wave_barrier();
        for (int j = 0; j < kSectors; j++) {
            const int sp = 5 + span * j / 6;
            const int ep = 5 + span * (j + 1) / 6 - 1;
            const int slen = ep - sp + 1;
            // HDL-64-sized sectors (<= 320 / <= 384 points) take the 5- / 6-slot instantiations, longer rings the full one
            if (slen <= 5 * 64) select_sector<5, false>(lane, j, sp, slen, rbeg, curv, picked, label, gap, sel_sh, sel_sh_n, sel_fl, sel_fl_n, scratch);
            else if (slen <= 6 * 64) select_sector<6, false>(lane, j, sp, slen, rbeg, curv, picked, label, gap, sel_sh, sel_sh_n, sel_fl, sel_fl_n, scratch);
            else select_sector<kSelMaxPerLane, false>(lane, j, sp, slen, rbeg, curv, picked, label, gap, sel_sh, sel_sh_n, sel_fl, sel_fl_n, scratch);
        }
        return;
    }
    // ---- fused with the curvature (VERDICT r4 #2b): no k_curvature launch; a sector's points and their halo are staged in LDS, the curvature of its
    // elements and the gap flags around them are computed from the stage
    const int scap = (sel_sector_cap(cap) + 12 + 3) & ~3;
    float *sx = (float *)(smem_w + ((cap + 15) & ~15)), *sy = sx + scap, *sz = sy + scap;
    unsigned long long *gap = (unsigned long long *)(sz + scap);
    for (int i = lane; i < len; i += 64) { picked[i] = 0; label[i] = 0; }
    // the ring's end points belong to no sector: their curvature comes straight from HBM
    curv_points_global(cl, uni_ptr(b.curv + off), n_cloud, rbeg, 5, lane);
    curv_points_global(cl, uni_ptr(b.curv + off), n_cloud, rend - 6, 6, lane);
    for (int j = 0; j < kSectors; j++) {
        const int sp = 5 + span * j / 6;
        const int ep = 5 + span * (j + 1) / 6 - 1;
        const int slen = ep - sp + 1;
        __builtin_amdgcn_wave_barrier();               // the previous sector is done with the stage
        // ring points sp - 5 .. ep + 6 (a point past the cloud reads as the origin: only its gap flag could use it, and that is forced to 0)
        for (int k = lane; k < slen + 12; k += 64) {
            const int ci = rbeg + sp - 5 + k;
            float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ci < n_cloud) p = cl[ci];
            sx[k] = p.x; sy[k] = p.y; sz[k] = p.z;
        }
        __builtin_amdgcn_wave_barrier();
        for (int k0 = 0; k0 < slen + 11 + 64; k0 += 64) {
            const int k = k0 + lane;
            bool g = false;
            if (k < slen + 11 && rbeg + sp - 5 + k + 1 < n_cloud) {
                const float gx = sx[k + 1] - sx[k], gy = sy[k + 1] - sy[k], gz = sz[k + 1] - sz[k];
                g = (double)(gx * gx + gy * gy + gz * gz) > 0.05;
            }
            const unsigned long long mk = __ballot(g);
            if (lane == 0) gap[k0 >> 6] = mk;
        }
        __builtin_amdgcn_wave_barrier();
        if (slen <= 5 * 64) select_sector<5, true>(lane, j, sp, slen, rbeg, curv, picked, label, gap, sel_sh, sel_sh_n, sel_fl, sel_fl_n, scratch, sx, sy, sz);
        else if (slen <= 6 * 64) select_sector<6, true>(lane, j, sp, slen, rbeg, curv, picked, label, gap, sel_sh, sel_sh_n, sel_fl, sel_fl_n, scratch, sx, sy, sz);
        else select_sector<kSelMaxPerLane, true>(lane, j, sp, slen, rbeg, curv, picked, label, gap, sel_sh, sel_sh_n, sel_fl, sel_fl_n, scratch, sx, sy, sz);
    }
}

// Two launches: the first gives every wave a slice for kSelSmallCap points (every HDL-64 ring; 18 KB per workgroup instead of
// 32 KB) over the whole (ring, scan) grid and defers longer rings to a work list; the second
// runs full-size slices as a small fixed grid over that list (normally empty).
constexpr int kSelSmallCap = 2304, kSelBigGrid = 128;

__global__ __launch_bounds__(256, 8) void k_select(BatchView b, int cap, int from_list)
{
    // the wave index as a scalar: ring, scan, the ring's bounds and every output pointer derived from it stay in SGPRs (as VGPRs they
    // pushed the 64-register budget of 8 waves / SIMD over and the pick loop reloaded a spilled pointer on every pick)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char *smem_w = smem + wave * sel_slice_bytes(cap);
    unsigned int *scratch = (unsigned int *)(smem + 4 * sel_slice_bytes(cap) + wave * kSelScratch);      // behind the four slices (a slice is a multiple of 8 bytes)
    if (!from_list) {
        select_ring(b, blockIdx.x * 4 + wave, b.scan0 + blockIdx.y, lane, smem_w, cap, true, scratch);
    } else {
        const int n_todo = b.sel_todo[0];
        for (int k = blockIdx.x * 4 + wave; k < n_todo; k += gridDim.x * 4) {
            const int e = b.sel_todo[1 + k];
            select_ring(b, e & 63, e >> 6, lane, smem_w, cap, false, scratch);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_voxel: one workgroup per (scan, ring): pcl::VoxelGrid(0.2 m, all fields averaged) over the ring's less-flat
// candidates (label <= 0 inside the sectors).  PCL sorts (cell, point) pairs and averages each cell's points in that
// order.  Consecutive ring points mostly share a voxel, so the points are first run-length compressed into segments
// (cell, first index); only the segments are sorted and a voxel's centroid is the float sum over its segments in
// (cell, first index) order = (cell, point index) order.
//
// Layout: point i = tid + 256 m of the ring sits in register slot m of thread tid, so wave w of slot m holds 64
// consecutive points and a ballot over the wave is a 64-bit piece of a per-ring bitmap (word i >> 6).  Segment heads,
// their count and their positions come from ballots and a 32-entry prefix over (slot, wave); "point i continues the run
// of point i - 1" is kept as a bitmap, which is all the accumulation needs to know about run lengths.
// Sort: rings of <= 1024 segments (the usual case) use a bucket sort -- bucket = cell index shifted down to <= 4096 values,
// histogram -> exclusive prefix -> cursors, keys scattered bucket by bucket in arrival order, then every key counts the
// smaller keys of its bucket (typically 1-3 members) and takes that place: the exact (cell, first index) order whatever the
// arrival order was.  Longer rings fall back to the register bitonic network.
//
// Two instantiations run back to back: <9 slots, 2048 buckets> takes the rings of <= 2304 points (every HDL-64 ring: ~80 VGPRs,
// 27 KB of LDS, 6 workgroups per CU) and appends a ring it cannot take (more points, or > 1024 segments: bitonic network) to a
// work list; <16 slots, 4096 buckets> runs as a small fixed grid over that list (normally empty).
constexpr int kVoxBucketSegsBig = 1024;   // bucket-sort capacity (segments) of the big instantiation; the small one takes its whole ring
constexpr int kVoxSmallSlots = 9, kVoxSmallBits = 10;
constexpr int kVoxBigSlots = kRingCap / 256, kVoxBigBits = 12;
constexpr int kVoxBigGrid = 512;    // workgroups of the second instantiation; each walks the work list with this stride
// LDS: key region (sorted keys | scattered copy | bucket counters; the big variant: kRingCap keys for the bitonic fallback),
// 256 scratch ints, the continuation bitmap and the last cells
template <int kSlots, int kBits, bool kSmall>
struct VoxCfg {
    static constexpr int kSegCap = kSmall ? 256 * kSlots : kVoxBucketSegsBig;
    // small: scattered keys as (cell u32 | first index u16) | sorted order as u16 positions into them | bucket counters, later
    // the voxel starts (u16)
    static constexpr int kHistBytes = (1 << kBits) * 4 > kSegCap * 2 ? (1 << kBits) * 4 : kSegCap * 2;
    static constexpr int kKeyBytes = kSmall ? kSegCap * 4 + kSegCap * 2 + kSegCap * 2 + kHistBytes : kRingCap * 8;
    static constexpr int kLds = kKeyBytes + 1024 + kSlots * 4 * 8 + kSlots * 4 * 4;
};
constexpr int kVoxLdsSmall = VoxCfg<kVoxSmallSlots, kVoxSmallBits, true>::kLds;
constexpr int kVoxLdsBig = VoxCfg<kVoxBigSlots, kVoxBigBits, false>::kLds;
static_assert(2 * kVoxBucketSegsBig * 8 + (1 << kVoxBigBits) * 4 <= kRingCap * 8, "bucket sort scratch must fit the key region");
#ifdef LMONO_VOX_PROF
#define VT(i) { if (blockIdx.x == 20 && blockIdx.y == 3 && threadIdx.x == 0) vt[i] = clock64(); }
#else
#define VT(i)
#endif

__device__ __forceinline__ float wave_min_f(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_max_f(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

template <int kVoxSlots, int kVoxBucketBits, bool kSmall>
__device__ __forceinline__ void voxel_ring(BatchView &b, int r, int s);

template <int kVoxSlots, int kVoxBucketBits, bool kSmall>
__global__ __launch_bounds__(256) void k_voxel(BatchView b)
{
    if (kSmall) voxel_ring<kVoxSlots, kVoxBucketBits, kSmall>(b, blockIdx.x, b.scan0 + blockIdx.y);
    else {
        const int n_todo = b.vox_todo[0];
        for (int k = blockIdx.x; k < n_todo; k += gridDim.x) {
            const int e = b.vox_todo[1 + k];
            voxel_ring<kVoxSlots, kVoxBucketBits, kSmall>(b, e & 63, e >> 6);
            __syncthreads();
        }
    }
}

template <int kVoxSlots, int kVoxBucketBits, bool kSmall>
__device__ __forceinline__ void voxel_ring(BatchView &b, int r, int s)
{
    constexpr int kVoxBuckets = 1 << kVoxBucketBits;
    constexpr int kCap = 256 * kVoxSlots;
    constexpr int kKeyBytes = VoxCfg<kVoxSlots, kVoxBucketBits, kSmall>::kKeyBytes;
    constexpr int kVoxBucketSegs = VoxCfg<kVoxSlots, kVoxBucketBits, kSmall>::kSegCap;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t off = b.off[s];
#ifdef LMONO_VOX_PROF
    long long vt[8];
#endif
    VT(0)
    const int *rb = b.ring_begin + s * 65;
    const int rbeg = rb[r], rend = rb[r + 1], len = rend - rbeg;
    const int S = rbeg + 5, E = rend - 6;
    if (kSmall && len > kCap && len <= kRingCap && E - S >= 6) {
        if (tid == 0) b.vox_todo[1 + atomicAdd(&b.vox_todo[0], 1)] = (s << 6) | r;    // the big instantiation's ring
        return;
    }
    if (E - S < 6 || len > kRingCap) {
        if (tid == 0) b.lf_n[s * 64 + r] = 0;
        return;
    }
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned long long *keys = (unsigned long long *)smem;                  // sorted segment keys (cell << 32 | first index)
    int *scr = (int *)(smem + kKeyBytes);                                   // 256 ints
    unsigned long long *contw = (unsigned long long *)(smem + kKeyBytes + 1024);   // [kVoxSlots * 4] run-continuation bitmap
    unsigned int *lastc = (unsigned int *)(contw + kVoxSlots * 4);          // [kVoxSlots * 4] cell of lane 63 of every (slot, wave)
    const signed char *label = (const signed char *)(b.label + off + rbeg);
    const float4 *cl = b.cloud + off + rbeg;
    const int c_lo = 5, c_hi = len - 7;   // sectors cover local [5, len-7]
    float mnx = FLT_MAX, mny = FLT_MAX, mnz = FLT_MAX, mxx = -FLT_MAX, mxy = -FLT_MAX, mxz = -FLT_MAX;
    int ncand = 0;
    // the ring's points stay in registers: all loads are issued back to back, and the cell pass needs no second trip
    float4 pr[kVoxSlots];
    unsigned int cand_mask = 0;
    {
        // every slot's label and point are requested before the first label is looked at (index clamped into the ring: len >= 12 here);
        // with the candidate test next to its load the compiler waited for one label byte after the other -- 9 round trips instead of 1
        signed char lbs[kVoxSlots];
#pragma unroll
        for (int m = 0; m < kVoxSlots; m++) {
            const int i = tid + 256 * m, ic = i < len ? i : len - 1;
            lbs[m] = label[ic]; pr[m] = cl[ic];
        }
#pragma unroll
        for (int m = 0; m < kVoxSlots; m++) {
            const int i = tid + 256 * m;
            if (i >= len) pr[m] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i >= c_lo && i <= c_hi && i < len && lbs[m] <= 0) cand_mask |= 1u << m;
        }
    }
#pragma unroll
    for (int m = 0; m < kVoxSlots; m++) {
        if ((cand_mask >> m) & 1u) {
            const float4 p = pr[m];
            mnx = fminf(mnx, p.x); mny = fminf(mny, p.y); mnz = fminf(mnz, p.z);
            mxx = fmaxf(mxx, p.x); mxy = fmaxf(mxy, p.y); mxz = fmaxf(mxz, p.z);
            ncand++;
        }
    }
    VT(1)
    float *fs = (float *)scr;   // [4 waves][6] + counts
    mnx = wave_min_f(mnx); mny = wave_min_f(mny); mnz = wave_min_f(mnz);
    mxx = wave_max_f(mxx); mxy = wave_max_f(mxy); mxz = wave_max_f(mxz);
    ncand = wave_sum_i(ncand);
    if (lane == 0) {
        fs[wave * 6 + 0] = mnx; fs[wave * 6 + 1] = mny; fs[wave * 6 + 2] = mnz;
        fs[wave * 6 + 3] = mxx; fs[wave * 6 + 4] = mxy; fs[wave * 6 + 5] = mxz;
        scr[32 + wave] = ncand;
    }
    __syncthreads();
    ncand = scr[32] + scr[33] + scr[34] + scr[35];
    if (ncand == 0) {
        if (tid == 0) b.lf_n[s * 64 + r] = 0;
        return;
    }
    mnx = fminf(fminf(fs[0], fs[6]), fminf(fs[12], fs[18]));
    mny = fminf(fminf(fs[1], fs[7]), fminf(fs[13], fs[19]));
    mnz = fminf(fminf(fs[2], fs[8]), fminf(fs[14], fs[20]));
    mxx = fmaxf(fmaxf(fs[3], fs[9]), fmaxf(fs[15], fs[21]));
    mxy = fmaxf(fmaxf(fs[4], fs[10]), fmaxf(fs[16], fs[22]));
    mxz = fmaxf(fmaxf(fs[5], fs[11]), fmaxf(fs[17], fs[23]));
    const float inv_leaf = 5.0f;
    const int minb0 = (int)floorf(mnx * inv_leaf), minb1 = (int)floorf(mny * inv_leaf), minb2 = (int)floorf(mnz * inv_leaf);
    const int div0 = (int)floorf(mxx * inv_leaf) - minb0 + 1, div1 = (int)floorf(mxy * inv_leaf) - minb1 + 1;
    const int mul1 = div0, mul2 = div0 * div1;
    // bucket of a cell for the bucket sort: its index shifted so that at most kVoxBuckets buckets cover the bounding box
    const unsigned long long ncell = (unsigned long long)div0 * (unsigned long long)div1 * (unsigned long long)((int)floorf(mxz * inv_leaf) - minb2 + 1);
    const int cell_bits = ncell > 1ull ? 64 - __clzll((long long)(ncell - 1ull)) : 0;
    const int bshift = cell_bits > kVoxBucketBits ? cell_bits - kVoxBucketBits : 0;
    // ---- cells (registers), cell of the last lane of every (slot, wave) for the neighbour test across waves
    unsigned int cellr[kVoxSlots];
#pragma unroll
    for (int m = 0; m < kVoxSlots; m++) {
        unsigned int cell = ~0u;
        if ((cand_mask >> m) & 1u) {
            const float4 p = pr[m];
            const int i0 = (int)(floorf(p.x * inv_leaf) - (float)minb0);
            const int i1 = (int)(floorf(p.y * inv_leaf) - (float)minb1);
            const int i2 = (int)(floorf(p.z * inv_leaf) - (float)minb2);
            cell = (unsigned int)(i0 + i1 * mul1 + i2 * mul2);
        }
        cellr[m] = cell;
        if (lane == 63) lastc[m * 4 + wave] = cell;
    }
    // small instantiation: the sorted order is a u16 index into the scattered keys (28 KB of LDS in all: 5 workgroups per CU;
    // the kernel is bound by resident workgroups -- 3 per CU at 46 KB measured 6.2 ms, 2 per CU 8.8 ms)
    unsigned long long *tmp = keys + kVoxBucketSegs;                                        // big only: scattered keys
    unsigned int *cellt = (unsigned int *)keys;                                             // small only: scattered (cell,
    unsigned short *idxt = (unsigned short *)((unsigned char *)keys + kVoxBucketSegs * 4);  //             first index)
    unsigned short *pos = (unsigned short *)((unsigned char *)keys + kVoxBucketSegs * 6);   // small only: sorted order
    int *hist = kSmall ? (int *)((unsigned char *)keys + kVoxBucketSegs * 8) : (int *)(keys + 2 * kVoxBucketSegs);
    auto skey = [&](int t) -> unsigned long long {
        if (kSmall) { const int q = pos[t]; return ((unsigned long long)cellt[q] << 32) | idxt[q]; }
        return keys[t];
    };
    __syncthreads();            // fs / scr[32..35] are read, lastc is written
    VT(2)
    // ---- segment heads: a candidate whose predecessor in the ring is not a candidate of the same cell
    unsigned int head_mask = 0;
#pragma unroll
    for (int m = 0; m < kVoxSlots; m++) {
        const int i = tid + 256 * m;
        unsigned int prev = (unsigned int)__shfl_up((int)cellr[m], 1);
        if (lane == 0) prev = wave > 0 ? lastc[m * 4 + wave - 1] : (m > 0 ? lastc[(m - 1) * 4 + 3] : ~0u);
        const bool cand = cellr[m] != ~0u;
        const bool cont = cand && i > 0 && prev == cellr[m];
        const bool head = cand && !cont;
        const unsigned long long hm = __ballot(head), cm = __ballot(cont);
        if (lane == 0) { contw[m * 4 + wave] = cm; scr[64 + m * 4 + wave] = __popcll(hm); }
        if (head) head_mask |= 1u << m;
    }
    __syncthreads();
    // position of a head in ring order = heads of the earlier (slot, wave) pairs + heads below it in its own ballot
    int nseg = 0;
    int hbase[kVoxSlots];
#pragma unroll
    for (int m = 0; m < kVoxSlots; m++) {
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const int c = scr[64 + m * 4 + w];
            if (w == wave) hbase[m] = nseg;
            nseg += c;
        }
    }
    int np2 = next_pow2(nseg);
    if (np2 < 2) np2 = 2;
    const bool bucket_path = nseg <= kVoxBucketSegs && ncell <= 0xffffffffull;
    if (kSmall && !bucket_path) {
        if (tid == 0) b.vox_todo[1 + atomicAdd(&b.vox_todo[0], 1)] = (s << 6) | r;    // needs the bitonic network: big instantiation
        return;
    }
    if (bucket_path) {
        for (int i = tid; i < kVoxBuckets; i += 256) hist[i] = 0;
        __syncthreads();
#pragma unroll
        for (int m = 0; m < kVoxSlots; m++)
            if ((head_mask >> m) & 1u) atomicAdd(&hist[cellr[m] >> bshift], 1);
        __syncthreads();
        VT(3)
        constexpr int kPer = kVoxBuckets / 256;
        int c[kPer], sum = 0;
#pragma unroll
        for (int q = 0; q < kPer; q++) { c[q] = hist[tid * kPer + q]; sum += c[q]; }
        const int incl_b = wave_scan_incl(sum);
        if (lane == 63) scr[44 + wave] = incl_b;
        __syncthreads();
        int run = incl_b - sum;
        for (int w = 0; w < wave; w++) run += scr[44 + w];
#pragma unroll
        for (int q = 0; q < kPer; q++) { hist[tid * kPer + q] = run; run += c[q]; }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < kVoxSlots; m++)
            if ((head_mask >> m) & 1u)
            {
                const int q = atomicAdd(&hist[cellr[m] >> bshift], 1);
                if (kSmall) { cellt[q] = cellr[m]; idxt[q] = (unsigned short)(tid + 256 * m); }
                else tmp[q] = ((unsigned long long)cellr[m] << 32) | (unsigned int)(tid + 256 * m);
            }
        __syncthreads();
        // a cursor is now the end of its bucket = the start of the next one
        for (int t = tid; t < nseg; t += 256) {
            const unsigned long long k = kSmall ? (((unsigned long long)cellt[t] << 32) | idxt[t]) : tmp[t];
            const int bk = (int)((unsigned int)(k >> 32) >> bshift);
            const int lo = bk > 0 ? hist[bk - 1] : 0, hi = hist[bk];
            int less = 0;
            for (int u = lo; u < hi; u++) less += (kSmall ? (((unsigned long long)cellt[u] << 32) | idxt[u]) : tmp[u]) < k ? 1 : 0;
            if (kSmall) pos[lo + less] = (unsigned short)t; else keys[lo + less] = k;
        }
        __syncthreads();
    } else if (!kSmall) {
#pragma unroll
        for (int m = 0; m < kVoxSlots; m++) {
            const unsigned long long hm = __ballot((head_mask >> m) & 1u);
            if ((head_mask >> m) & 1u)
                keys[hbase[m] + __popcll(hm & ((1ull << lane) - 1ull))] = ((unsigned long long)cellr[m] << 32) | (unsigned int)(tid + 256 * m);
        }
        for (int k = nseg + tid; k < np2; k += 256) keys[k] = ~0ull;
        __syncthreads();
        VT(3)
        bitonic_sort_u64(keys, np2);
    }
    VT(4)
    // ---- runs of equal cell over the sorted segments -> output voxels; vstart[o] = first sorted segment of voxel o
    int *vstart = bucket_path && !kSmall ? (int *)(keys + kVoxBucketSegs) : nullptr;   // big: the scattered copy is dead now
    unsigned short *vstart16 = (unsigned short *)hist;                                 // small: the bucket counters are dead after the ranking
    constexpr int kRounds = (kVoxBucketSegs + 255) / 256;
    int n_out = 0, obase[kRounds];
    {
        const int rounds = (nseg + 255) / 256;
        // voxel starts per round and wave -> prefix in sorted order (round, wave, lane)
        for (int j = 0; j < rounds; j++) {
            const int t = tid + 256 * j;
            const bool st = t < nseg && (t == 0 || (unsigned int)(skey(t) >> 32) != (unsigned int)(skey(t - 1) >> 32));
            const unsigned long long sm = __ballot(st);
            if (lane == 0) scr[128 + ((j * 4 + wave) & 127)] = __popcll(sm);
        }
    }
    __syncthreads();
    float4 *outp = b.lf_tmp + off + rbeg;
    VT(5)
    if (bucket_path) {
        const int rounds = (nseg + 255) / 256;
        int total = 0;
#pragma unroll
        for (int j = 0; j < kRounds; j++)
            for (int w = 0; w < 4; w++) { if (j < rounds) { if (w == wave) obase[j] = total; total += scr[128 + j * 4 + w]; } }
        n_out = total;
#pragma unroll
        for (int j = 0; j < kRounds; j++) {
            if (j >= rounds) break;
            const int t = tid + 256 * j;
            const bool st = t < nseg && (t == 0 || (unsigned int)(skey(t) >> 32) != (unsigned int)(skey(t - 1) >> 32));
            const unsigned long long sm = __ballot(st);
            if (st) { const int o = obase[j] + __popcll(sm & ((1ull << lane) - 1ull)); if (kSmall) vstart16[o] = (unsigned short)t; else vstart[o] = t; }
        }
        __syncthreads();
        // one voxel per thread and round: its segments in sorted order, every segment's points in index order
        for (int v = tid; v < n_out; v += 256) {
            const int t = kSmall ? (int)vstart16[v] : vstart[v];
            const unsigned int c = (unsigned int)(skey(t) >> 32);
            float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
            int cnt = 0;
            for (int u = t; u < nseg; u++) {
                const unsigned long long ku = skey(u);
                if ((unsigned int)(ku >> 32) != c) break;
                const int i0 = (int)(ku & 0xffffffffull);
                // run length from the continuation bitmap: points i0+1 .. while their bit is set
                int rl = 1;
                while (i0 + rl < len && ((contw[(i0 + rl) >> 6] >> ((i0 + rl) & 63)) & 1ull)) rl++;
                for (int b4 = 0; b4 < rl; b4 += 4) {
                    float4 q4[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) q4[k] = b4 + k < rl ? cl[i0 + b4 + k] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int k = 0; k < 4; k++) if (b4 + k < rl) { sx += q4[k].x; sy += q4[k].y; sz += q4[k].z; si += q4[k].w; }
                }
                cnt += rl;
            }
            const float fc = (float)cnt;
            outp[v] = make_float4(sx / fc, sy / fc, sz / fc, si / fc);
        }
    } else if (!kSmall) {
        // long rings: contiguous ranges of sorted segments per thread (serial prefix of the voxel starts)
        const int chunk2 = (np2 + 255) / 256;
        const int t_lo = tid * chunk2, t_hi = min(t_lo + chunk2, nseg);
        int nstart = 0;
        for (int t = t_lo; t < t_hi; t++) {
            const unsigned int c = (unsigned int)(keys[t] >> 32);
            if (t == 0 || c != (unsigned int)(keys[t - 1] >> 32)) nstart++;
        }
        const int incl = wave_scan_incl(nstart);
        __syncthreads();
        if (lane == 63) scr[48 + wave] = incl;
        __syncthreads();
        int base = incl - nstart;
        for (int w = 0; w < wave; w++) base += scr[48 + w];
        n_out = scr[48] + scr[49] + scr[50] + scr[51];
        int o = base;
        for (int t = t_lo; t < t_hi; t++) {
            const unsigned int c = (unsigned int)(keys[t] >> 32);
            if (t == 0 || c != (unsigned int)(keys[t - 1] >> 32)) {
                float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
                int cnt = 0;
                for (int u = t; u < nseg && (unsigned int)(keys[u] >> 32) == c; u++) {
                    const int i0 = (int)(keys[u] & 0xffffffffull);
                    int rl = 1;
                    while (i0 + rl < len && ((contw[(i0 + rl) >> 6] >> ((i0 + rl) & 63)) & 1ull)) rl++;
                    for (int b4 = 0; b4 < rl; b4 += 4) {
                        float4 q4[4];
#pragma unroll
                        for (int k = 0; k < 4; k++) q4[k] = b4 + k < rl ? cl[i0 + b4 + k] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                        for (int k = 0; k < 4; k++) if (b4 + k < rl) { sx += q4[k].x; sy += q4[k].y; sz += q4[k].z; si += q4[k].w; }
                    }
                    cnt += rl;
                }
                const float fc = (float)cnt;
                outp[o++] = make_float4(sx / fc, sy / fc, sz / fc, si / fc);
            }
        }
    }
    VT(6)
#ifdef LMONO_VOX_PROF
    if (blockIdx.x == 20 && blockIdx.y == 3 && tid == 0) printf("VOX len %d nseg %d nout %d | load %lld bbox+cell %lld heads %lld sort %lld runs %lld accum %lld\n", len, nseg, n_out, vt[1]-vt[0], vt[2]-vt[1], vt[3]-vt[2], vt[4]-vt[3], vt[5]-vt[4], vt[6]-vt[5]);
#endif
    if (tid == 0) b.lf_n[s * 64 + r] = n_out;
}

// ------------------------------------------------------------------------------------------------
// Line tables of a feature cloud (line = int(intensity), what laserOdometry reads as the scan line):
//   first_ge[t] = first index whose line is >= t, last_le[t] = last index whose line is <= t  (t = 0..65).
// The odometry walk around a nearest point of line ra visits exactly the index window (last_le[ra-3], first_ge[ra+3])
// provided no point precedes a point whose line is >= 3 lower ("regular"); irregular clouds are flagged and walked
// in array order instead.  Lines are not monotone in general: A-LOAM's relTime can be negative (line = ring - 1).
// ------------------------------------------------------------------------------------------------
#ifndef LMONO_COMPACT_T
#define LMONO_COMPACT_T 256
#endif
constexpr int kCompT = LMONO_COMPACT_T;      // threads per scan of the stand-alone kernel; 256 / 512 / 1024 measured the same 2.0 ms per pass (profiles/r4: bound by its traffic)
static_assert(kCompT >= 256 && kCompT % 64 == 0, "the four prefixes take one wave each");

// one scan's feature clouds compacted by kT threads (k_compact: kCompT; k_compact_index, odometry.hip: the line index's 1024); returns the sizes of the
// two "last" clouds (less sharp, less flat) to every thread
template <int kT>
__device__ __forceinline__ void compact_scan(const BatchView &b, int s, int &n_ls_out, int &n_lf_out)
{
    const int tid = threadIdx.x;
    const int64_t off = b.off[s];
    constexpr int NE = kMaxRings * kSectors;   // 384
    __shared__ int pre_sh[NE + 1], pre_ls[NE + 1], pre_fl[NE + 1], pre_lf[kMaxRings + 1];
    const int *nsh = b.sel_sharp_n + s * NE;
    const int *nfl = b.sel_flat_n + s * NE;
    // four exclusive prefixes, one per wave: (ring, sector) counts of sharp (<= 2), less sharp, flat; ring counts of less flat
    {
        const int lane = tid & 63, wave = tid >> 6;
        constexpr int kPer = NE / 64;   // 6 consecutive entries per lane
        int c[kPer], sum = 0;
        if (wave < 4) {
        // one source array per wave, six unconditional loads per lane in flight together
        const int *srcp = wave <= 1 ? nsh : (wave == 2 ? nfl : b.lf_n + s * 64);
#pragma unroll
        for (int q = 0; q < kPer; q++) {
            const int e = lane * kPer + q;
            const bool on = wave < 3 || e < kMaxRings;
            c[q] = srcp[on ? e : 0];
            if (!on) c[q] = 0;
        }
#pragma unroll
        for (int q = 0; q < kPer; q++) {
            if (wave == 0) c[q] = min(c[q], 2);
            sum += c[q];
        }
        const int incl = wave_scan_incl(sum);
        int run = incl - sum;
        int *dst = wave == 0 ? pre_sh : (wave == 1 ? pre_ls : (wave == 2 ? pre_fl : pre_lf));
        const int n_e = wave == 3 ? kMaxRings : NE;
#pragma unroll
        for (int q = 0; q < kPer; q++) {
            const int e = lane * kPer + q;
            if (e < n_e) dst[e] = run;
            run += c[q];
        }
        if (lane == 63) dst[n_e] = run;
        }
    }
    __syncthreads();
    const float4 *cl = b.cloud + off;
    float4 *sharp = b.sharp + (size_t)s * kMaxSharp;
    float4 *ls = b.less_sharp + (size_t)s * kMaxLessSharp;
    float4 *flat = b.flat + (size_t)s * kMaxFlat;
    float4 *lf = b.less_flat + off;
    const int *ssel = b.sel_sharp + (size_t)s * NE * 20;
    const int *fsel = b.sel_flat + (size_t)s * NE * 4;
    // line tables of the two "last" clouds are gathered while the points pass through (first / last index of every line)
    __shared__ int s_first[2][66], s_last[2][66], s_flag[2], s_rb[kMaxRings + 1];
    if (tid < 66) { s_first[0][tid] = INT_MAX; s_last[0][tid] = -1; s_first[1][tid] = INT_MAX; s_last[1][tid] = -1; }
    if (tid < 2) s_flag[tid] = 0;
    if (tid >= 128 && tid < 128 + kMaxRings + 1) s_rb[tid - 128] = b.ring_begin[s * 65 + tid - 128];
    __syncthreads();
    // four selection slots per thread and turn: the counts come from the LDS prefixes, the index loads and then the point gathers are issued
    // together (behind per-slot guards the compiler waits for every load in turn: three dependent round trips per slot, 30 slots per thread)
    for (int x0 = tid; x0 < NE * 20; x0 += 4 * kT) {
        int pos[4], idx[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int x = x0 + kT * q, e = x / 20, k = x % 20;
            pos[q] = (x < NE * 20 && k < pre_ls[e + 1] - pre_ls[e]) ? pre_ls[e] + k : -1;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) idx[q] = ssel[pos[q] >= 0 ? x0 + kT * q : 0];
        float4 p4[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { const float4 *src = pos[q] >= 0 ? cl + idx[q] : b.cloud; p4[q] = *src; }      // (an unused slot reads the batch's first point)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (pos[q] < 0) continue;
            const int x = x0 + kT * q, e = x / 20, k = x % 20;
            const float4 p = p4[q];
            ls[pos[q]] = p;
            if (k < 2) sharp[pre_sh[e] + k] = p;
            int v = (int)p.w;
            v = v < 0 ? 0 : (v > 65 ? 65 : v);
            atomicMin(&s_first[0][v], pos[q]);
            atomicMax(&s_last[0][v], pos[q]);
        }
    }
    for (int x0 = tid; x0 < NE * 4; x0 += 2 * kT) {
        int pos[2], idx[2];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int x = x0 + kT * q, e = x / 4, k = x % 4;
            pos[q] = (x < NE * 4 && k < pre_fl[e + 1] - pre_fl[e]) ? pre_fl[e] + k : -1;
        }
#pragma unroll
        for (int q = 0; q < 2; q++) idx[q] = fsel[pos[q] >= 0 ? x0 + kT * q : 0];
        float4 p2[2];
#pragma unroll
        for (int q = 0; q < 2; q++) { const float4 *src = pos[q] >= 0 ? cl + idx[q] : b.cloud; p2[q] = *src; }
#pragma unroll
        for (int q = 0; q < 2; q++) if (pos[q] >= 0) flat[pos[q]] = p2[q];
    }
    // less-flat cloud = the rings' voxel outputs back to back: every thread finds the ring of its output index by a binary
    // search over the ring prefix (all loads independent, four per thread in flight)
    const int n_lf = pre_lf[kMaxRings];
    for (int j0 = tid; j0 < n_lf; j0 += 4 * kT) {
        float4 v4[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int j = j0 + kT * q;
            v4[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < n_lf) {
                int lo = 0, hi = kMaxRings;
                while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (pre_lf[mid] <= j) lo = mid; else hi = mid; }
                v4[q] = b.lf_tmp[off + s_rb[lo] + (j - pre_lf[lo])];
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int j = j0 + kT * q;
            if (j < n_lf) {
                lf[j] = v4[q];
                int v = (int)v4[q].w;
                v = v < 0 ? 0 : (v > 65 ? 65 : v);
                atomicMin(&s_first[1][v], j);
                atomicMax(&s_last[1][v], j);
            }
        }
    }
    if (tid == 0) {
        b.feat_n[s * 4 + 0] = pre_sh[NE]; b.feat_n[s * 4 + 1] = pre_ls[NE];
        b.feat_n[s * 4 + 2] = pre_fl[NE]; b.feat_n[s * 4 + 3] = n_lf;
    }
    __syncthreads();
    // irregular iff some line a >= b + 3 starts before line b ends (see above)
    for (int x = tid; x < 2 * 66 * 66; x += kT) {
        const int cld = x / (66 * 66), y = x % (66 * 66), a = y / 66, bb = y % 66;
        if (a >= bb + 3 && s_first[cld][a] < s_last[cld][bb]) s_flag[cld] = 1;
    }
    __syncthreads();
    if (tid == 0 || tid == 64) {
        const int cld = tid >> 6;
        const int n = cld ? n_lf : pre_ls[NE];
        int *first_ge = b.line_first_ge + (size_t)(s * 2 + cld) * 66, *last_le = b.line_last_le + (size_t)(s * 2 + cld) * 66;
        int m = n;
        for (int t = 65; t >= 0; t--) { m = min(m, s_first[cld][t] == INT_MAX ? n : s_first[cld][t]); first_ge[t] = m; }
        int M = -1;
        for (int t = 0; t <= 65; t++) { M = max(M, s_last[cld][t]); last_le[t] = M; }
        if (s_flag[cld]) atomicOr(b.status + s, kStatusIrregularLines);
    }
    n_ls_out = pre_ls[NE]; n_lf_out = n_lf;
}

__global__ __launch_bounds__(kCompT) void k_compact(BatchView b)
{
    int n_ls, n_lf;
    compact_scan<kCompT>(b, b.scan0 + blockIdx.x, n_ls, n_lf);
}

} // namespace lmono
