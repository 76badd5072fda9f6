// lmono_amd/csrc/mapping.hip -- gfx950 kernels of the scan-to-map optimisation step of A-LOAM laserMapping (SURVEY.md
// Appendix A.4, row 8f-1; source absent from the reference tree, /root/reference/.gitmodules:1-3), batched over
// independent streams:
//   k_grid_clear / _insert / _starts / _fill   1 m hash grid over an arbitrary float4 cloud (the map clouds of the cube
//                  neighbourhood), many workgroups per (stream, cloud); 16-B cells, cell-sorted copy
//   k_map_correspond  32 lanes per down-sampled scan point: pointAssociateToMap, exact 5-NN among the 27 cells around it
//                  (every neighbour that can pass the "5th distance^2 < 1" test lies inside them)
//   k_map_factor   one thread per point: the line test (covariance of the 5 neighbours, Jacobi eigen-decomposition,
//                  largest > 3 x middle) or the plane fit (column-pivoted Householder least squares, all five within
//                  0.2) -> an 80-B double record
//   k_map_solve    one workgroup per stream: ceres::Solve restated (trust-region LM, <= 4 iterations, Huber 0.1) over
//                  the records (LidarEdgeFactor / LidarPlaneNormFactor, closed-form Jacobians)
// The map bookkeeping (cube array, voxel re-filtering) is not part of this file yet.
#include "batch.hpp"

namespace lmono {

struct CloudJob {
    const float4 *src;     // the cloud
    int n;                 // its size
    GridCell *cell;        // hash table (capacity tcap slots)
    int tcap;
    float4 *sorted;        // cell-sorted copy, .w = original index bits
    int *slot_of, *rank_of;
    int *mask_out;         // table size - 1 actually used (0: table unusable -> no correspondences)
    int *bump;             // the job's run allocator inside `sorted` (zeroed by k_grid_clear)
};

// Round 4: the grid of one cloud is built by MANY workgroups (blockIdx.y = job, blockIdx.x strides over the job's cells / points) in four launches --
// clear | insert | starts | fill -- instead of one 1024-thread workgroup per cloud (0.26 ms per frame for the ~100 k-point neighbourhood clouds, a
// chain of ~1 us global round trips on one CU).  The cells no longer sit in slot order inside `sorted`: a cell's run starts where an atomic bump of
// the job's counter put it.  Nothing reads the order of the runs (k_map_correspond ranks candidates by (distance, original index)).
__device__ __forceinline__ int cloud_grid_size(const CloudJob &J)
{
    int T = next_pow2(J.n + 1);
    if (T < 1024) T = 1024;
    return T > J.tcap ? 0 : T;
}

constexpr int kCgT = 256;      // threads per workgroup of the four kernels

__global__ __launch_bounds__(kCgT) void k_grid_clear(const CloudJob *jobs)
{
    const CloudJob J = jobs[blockIdx.y];
    const int T = cloud_grid_size(J);
    if (blockIdx.x == 0 && threadIdx.x == 0) { *J.mask_out = T ? T - 1 : 0; *J.bump = 0; }
    typedef __attribute__((address_space(1))) GridCell GCell;
    GridCell *cell = (GridCell *)(GCell *)J.cell;
    for (int i = blockIdx.x * kCgT + threadIdx.x; i < T; i += gridDim.x * kCgT) { GridCell e; e.key = kEmptyKey; e.start = 0; e.cnt = 0; cell[i] = e; }
}

__global__ __launch_bounds__(kCgT) void k_grid_insert(const CloudJob *jobs)
{
    const CloudJob J = jobs[blockIdx.y];
    const int T = cloud_grid_size(J), n = J.n;
    if (T == 0) return;
    const unsigned int mask = (unsigned int)(T - 1);
    typedef __attribute__((address_space(1))) const float4 GF4;
    typedef __attribute__((address_space(1))) GridCell GCell;
    typedef __attribute__((address_space(1))) int GI;
    GridCell *cell = (GridCell *)(GCell *)J.cell;
    const float4 *gsrc = (const float4 *)(GF4 *)J.src;
    GI *slot_of = (GI *)J.slot_of, *rank_of = (GI *)J.rank_of;
    const int stride = gridDim.x * kCgT, lane = threadIdx.x & 63;
    // The map clouds are voxel-filter outputs, cube by cube in ascending voxel order: neighbours in the cloud are often neighbours in space.  Lanes
    // whose points follow each other INTO THE SAME CELL form a run; the run's first lane claims the cell and takes the ranks for all of them (one
    // compare-and-swap chain and one add per run instead of per point: device-scope atomics are what bounds this kernel).
    for (int i0 = blockIdx.x * kCgT + (threadIdx.x & ~63); i0 < n; i0 += 2 * stride) {
        float4 p[2];
#pragma unroll
        for (int u = 0; u < 2; u++) p[u] = gsrc[min(i0 + stride * u + lane, n - 1)];
        unsigned long long key[2], hm[2];
        unsigned int sl[2];
        bool open[2], valid[2];
        int rl[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            key[u] = cell_key((int)floorf(p[u].x * kInvCell), (int)floorf(p[u].y * kInvCell), (int)floorf(p[u].z * kInvCell));
            sl[u] = hash_key(key[u]) & mask;
            valid[u] = i0 + stride * u + lane < n;
            const unsigned long long prev = __shfl_up(key[u], 1);
            const bool head = valid[u] && (lane == 0 || key[u] != prev);
            hm[u] = __ballot(head);
            const unsigned long long vm = __ballot(valid[u]);
            // run length of a head: up to the next head or the last valid lane
            const unsigned long long after = lane < 63 ? (hm[u] >> (lane + 1)) : 0ull;
            const int nxt = after ? lane + 1 + (int)__builtin_ctzll(after) : (int)__popcll(vm);
            rl[u] = head ? nxt - lane : 0;
            open[u] = head;
        }
        while (open[0] || open[1]) {
            unsigned long long old[2];
#pragma unroll
            for (int u = 0; u < 2; u++) old[u] = open[u] ? atomicCAS(&cell[sl[u]].key, kEmptyKey, key[u]) : key[u];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                if (!open[u]) continue;
                if (old[u] == kEmptyKey || old[u] == key[u]) open[u] = false;
                else sl[u] = (sl[u] + 1) & mask;
            }
        }
        int rk[2];
#pragma unroll
        for (int u = 0; u < 2; u++) rk[u] = rl[u] > 0 ? atomicAdd(&cell[sl[u]].cnt, rl[u]) : 0;
#pragma unroll
        for (int u = 0; u < 2; u++) {
            // every lane takes slot and first rank from its run's head: the highest head at or below it
            const unsigned long long below = hm[u] & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
            const int lead = below ? 63 - (int)__builtin_clzll(below) : 0;
            const int s_run = __shfl((int)sl[u], lead), r_run = __shfl(rk[u], lead);
            if (valid[u]) { slot_of[i0 + stride * u + lane] = s_run; rank_of[i0 + stride * u + lane] = r_run + (lane - lead); }
        }
    }
}

// every occupied cell takes its run of `sorted`: one bump of the job's counter per workgroup and turn (2048 cells)
__global__ __launch_bounds__(kCgT) void k_grid_starts(const CloudJob *jobs)
{
    const CloudJob J = jobs[blockIdx.y];
    const int T = cloud_grid_size(J);
    typedef __attribute__((address_space(1))) GridCell GCell;
    GridCell *cell = (GridCell *)(GCell *)J.cell;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ int s_w[kCgT / 64], s_base;
    for (int c0 = blockIdx.x * 8 * kCgT; c0 < T; c0 += gridDim.x * 8 * kCgT) {        // T is a multiple of 1024, >= 1024
        int cnt[8], local = 0;
#pragma unroll
        for (int u = 0; u < 8; u++) { const int ci = c0 + 8 * tid + u; cnt[u] = ci < T ? cell[ci].cnt : 0; local += cnt[u]; }
        const int incl = wave_scan_incl(local);
        if (lane == 63) s_w[wave] = incl;
        __syncthreads();
        int run = incl - local, tot = 0;
        for (int w = 0; w < kCgT / 64; w++) { run += w < wave ? s_w[w] : 0; tot += s_w[w]; }
        if (tid == 0) s_base = tot > 0 ? atomicAdd(J.bump, tot) : 0;
        __syncthreads();
        run += s_base;
#pragma unroll
        for (int u = 0; u < 8; u++) { const int ci = c0 + 8 * tid + u; if (ci < T && cnt[u] > 0) cell[ci].start = run; run += cnt[u]; }
        __syncthreads();
    }
}

__global__ __launch_bounds__(kCgT) void k_grid_fill(const CloudJob *jobs)
{
    const CloudJob J = jobs[blockIdx.y];
    const int T = cloud_grid_size(J), n = J.n;
    if (T == 0) return;
    typedef __attribute__((address_space(1))) const float4 GF4;
    typedef __attribute__((address_space(1))) float4 GF4W;
    typedef __attribute__((address_space(1))) const GridCell GCell;
    typedef __attribute__((address_space(1))) const int GI;
    const GridCell *cell = (const GridCell *)(GCell *)J.cell;
    const float4 *gsrc = (const float4 *)(GF4 *)J.src;
    float4 *gsorted = (float4 *)(GF4W *)J.sorted;
    GI *slot_of = (GI *)J.slot_of, *rank_of = (GI *)J.rank_of;
    for (int i = blockIdx.x * kCgT + threadIdx.x; i < n; i += gridDim.x * kCgT) {
        const float4 p = gsrc[i];
        const int st = cell[slot_of[i]].start;
        gsorted[st + rank_of[i]] = make_float4(p.x, p.y, p.z, __int_as_float(i));
    }
}

// the four launches of a table of cloud jobs; max_n = the largest cloud among them
static inline void launch_cloud_grids(hipStream_t st, const CloudJob *jobs_d, int n_jobs, int max_n, bool cleared = false)
{
    if (n_jobs <= 0) return;
    int T = 1024;
    while (T < max_n + 1) T <<= 1;
    const int per = 2048 / (n_jobs < 16 ? 1 : 4);           // elements per workgroup: few jobs -> more workgroups per job
    const unsigned bc = (unsigned)std::min(512, std::max(1, (T + per - 1) / per)), bp = (unsigned)std::min(512, std::max(1, (max_n + per - 1) / per));
    if (!cleared) hipLaunchKernelGGL(k_grid_clear, dim3(bc, (unsigned)n_jobs), dim3(kCgT), 0, st, jobs_d);      // (cleared: the tables were emptied by the caller's kernel)
    hipLaunchKernelGGL(k_grid_insert, dim3(bp, (unsigned)n_jobs), dim3(kCgT), 0, st, jobs_d);
    hipLaunchKernelGGL(k_grid_starts, dim3((unsigned)std::min(512, std::max(1, T / (8 * kCgT))), (unsigned)n_jobs), dim3(kCgT), 0, st, jobs_d);
    hipLaunchKernelGGL(k_grid_fill, dim3(bp, (unsigned)n_jobs), dim3(kCgT), 0, st, jobs_d);
}

// ---- small dense pieces, same arithmetic as oracle/lo_mapping.c --------------------------------------------------
#ifndef LMONO_MAP_FAST_ROT
#define LMONO_MAP_FAST_ROT 0      // 1: the 3 x 3 Jacobi rotations from v_rsq / v_rcp (a measurement: the oracle's arithmetic is the default)
#endif
// ascending eigenvalues / eigenvectors (columns) of a symmetric 3x3: cyclic Jacobi
__device__ void sym_eig3(const double *A, double *evals, double *evecs)
{
    double a[9], v[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    for (int k = 0; k < 9; k++) a[k] = A[k];
    for (int sweep = 0; sweep < 60; sweep++) {
        const double off = a[1] * a[1] + a[2] * a[2] + a[5] * a[5];
        const double dia = a[0] * a[0] + a[4] * a[4] + a[8] * a[8];
        if (off <= 1e-40 * (dia > 0 ? dia : 1.0) || off == 0.0) break;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                const double apq = a[p * 3 + q];
                if (apq == 0.0) continue;
#if LMONO_MAP_FAST_ROT
                double c, s;
                jacobi_cs(a[p * 3 + p], a[q * 3 + q], apq, c, s);
#else
                const double theta = (a[q * 3 + q] - a[p * 3 + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#endif
                for (int k = 0; k < 3; k++) {
                    const double akp = a[k * 3 + p], akq = a[k * 3 + q];
                    a[k * 3 + p] = c * akp - s * akq; a[k * 3 + q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; k++) {
                    const double apk = a[p * 3 + k], aqk = a[q * 3 + k];
                    a[p * 3 + k] = c * apk - s * aqk; a[q * 3 + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; k++) {
                    const double vkp = v[k * 3 + p], vkq = v[k * 3 + q];
                    v[k * 3 + p] = c * vkp - s * vkq; v[k * 3 + q] = s * vkp + c * vkq;
                }
            }
    }
    int ord[3] = { 0, 1, 2 };
    const double d[3] = { a[0], a[4], a[8] };
    for (int i = 0; i < 2; i++) for (int j = i + 1; j < 3; j++) if (d[ord[j]] < d[ord[i]]) { const int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
    for (int c = 0; c < 3; c++) { evals[c] = d[ord[c]]; for (int k = 0; k < 3; k++) evecs[k * 3 + c] = v[k * 3 + ord[c]]; }
}

// norm = argmin |A norm + 1| over the 5 rows (matA0.colPivHouseholderQr().solve(matB0)); true when all five points are
// within 0.2 of the plane
__device__ bool plane_fit5(const double *pts, double *norm, double &negative_OA_dot_norm)
{
    double A[15], b[5] = { -1, -1, -1, -1, -1 };
    for (int k = 0; k < 15; k++) A[k] = pts[k];
    int perm[3] = { 0, 1, 2 };
    for (int k = 0; k < 3; k++) {
        int piv = k; double best = -1.0;
        for (int c = k; c < 3; c++) { double s = 0; for (int r = k; r < 5; r++) s += A[r * 3 + c] * A[r * 3 + c]; if (s > best) { best = s; piv = c; } }
        if (piv != k) { for (int r = 0; r < 5; r++) { const double t = A[r * 3 + k]; A[r * 3 + k] = A[r * 3 + piv]; A[r * 3 + piv] = t; } const int t = perm[k]; perm[k] = perm[piv]; perm[piv] = t; }
        double nrm = 0; for (int r = k; r < 5; r++) nrm += A[r * 3 + k] * A[r * 3 + k];
        nrm = sqrt(nrm);
        if (nrm == 0.0) continue;
        const double alpha = A[k * 3 + k] > 0 ? -nrm : nrm;
        double v[5] = { 0, 0, 0, 0, 0 };
        for (int r = k; r < 5; r++) v[r] = A[r * 3 + k];
        v[k] -= alpha;
        double vv = 0; for (int r = k; r < 5; r++) vv += v[r] * v[r];
        if (vv == 0.0) continue;
        for (int c = k; c < 3; c++) {
            double dot = 0; for (int r = k; r < 5; r++) dot += v[r] * A[r * 3 + c];
            const double f = 2.0 * dot / vv;
            for (int r = k; r < 5; r++) A[r * 3 + c] -= f * v[r];
        }
        double dot = 0; for (int r = k; r < 5; r++) dot += v[r] * b[r];
        const double f = 2.0 * dot / vv;
        for (int r = k; r < 5; r++) b[r] -= f * v[r];
    }
    double y[3];
    for (int k = 2; k >= 0; k--) {
        double s = b[k];
        for (int c = k + 1; c < 3; c++) s -= A[k * 3 + c] * y[c];
        y[k] = A[k * 3 + k] != 0.0 ? s / A[k * 3 + k] : 0.0;
    }
    for (int k = 0; k < 3; k++) norm[perm[k]] = y[k];
    const double nn = sqrt(norm[0] * norm[0] + norm[1] * norm[1] + norm[2] * norm[2]);
    negative_OA_dot_norm = 1.0 / nn;
    for (int k = 0; k < 3; k++) norm[k] /= nn;
    for (int j = 0; j < 5; j++)
        if (fabs(norm[0] * pts[j * 3] + norm[1] * pts[j * 3 + 1] + norm[2] * pts[j * 3 + 2] + negative_OA_dot_norm) > 0.2) return false;
    return true;
}

// ---- correspondences ------------------------------------------------------------------------------------------------
struct MapRec { double cp[3]; double a[3]; double b[3]; int kind; int pad; };   // kind 0 none, 1 edge, 3 plane-norm

struct MapStream {
    // map clouds (grids built by the k_grid_* kernels) and down-sampled scan clouds of one stream
    const GridCell *cell[2]; const float4 *sorted[2]; const float4 *cloud[2]; const int *mask[2]; int n_map[2];
    const float4 *stack[2]; int n_stack[2];
    const int *n_stack_d;  // when set: the two sizes are read from the device (the voxel filter's counts, still in flight when the table is built); < 0 counts as 0
    MapRec *rec;           // [n_stack[0] + n_stack[1]]
    double *x;             // [8] q(xyzw), t
    int *stats;            // [8] n_edge[2], n_plane[2], lm_iters[2], pad
    int *nn_tmp;           // [n][5] neighbour indices of the current outer iteration (-1: the 5-NN test failed)
    int *nn_out;           // optional [n][5] neighbour indices of the accepted blocks of the last outer iteration (-1: none)
    // k_map_solve as a cluster of K workgroups: partial sums [kMsEvals][K][28] and arrival counters [2 outer][kMsEvals] (zeroed by the host per frame)
    double *part; unsigned int *bar;
};
constexpr int kMsEvals = 5;        // evaluations of one solve at most: the start + one per LM iteration (max 4)
constexpr int kMsMaxK = 8;

__device__ __forceinline__ void map_stack_sizes(const MapStream &S, int &n0, int &n1)
{
    n0 = S.n_stack[0]; n1 = S.n_stack[1];
    if (S.n_stack_d) { n0 = S.n_stack_d[0]; n1 = S.n_stack_d[1]; n0 = n0 < 0 ? 0 : n0; n1 = n1 < 0 ? 0 : n1; }
}

// sorted insertion of (d, idx) into a lane-local ascending top-5
__device__ __forceinline__ void top5_insert(float *td, int *ti, float d, int idx)
{
    if (!(d < td[4] || (d == td[4] && idx < ti[4]))) return;
    td[4] = d; ti[4] = idx;
#pragma unroll
    for (int k = 4; k > 0; k--) {
        const bool sw = td[k] < td[k - 1] || (td[k] == td[k - 1] && ti[k] < ti[k - 1]);
        const float fd = td[k]; const int fi = ti[k];
        td[k] = sw ? td[k - 1] : td[k]; ti[k] = sw ? ti[k - 1] : ti[k];
        td[k - 1] = sw ? fd : td[k - 1]; ti[k - 1] = sw ? fi : ti[k - 1];
    }
}

// Workgroup -> (block bx of nbx, stream) of the per-stream kernels.  nx = 0: a 2-D launch (x = block, y = stream).  nx > 0 (from 8 streams on): a 1-D launch
// of 8 * nx * ceil(ny / 8) workgroups decoded so that ALL workgroups of a stream run on ONE XCD (workgroup id mod 8), the streams of an XCD one after
// the other: a stream's map cloud and cell table (4-5 MB) then stay in that XCD's L2 instead of every L2 seeing every stream in flight.
__device__ __forceinline__ bool map_block_of(int nx, int ny, int &bx, int &nbx, int &stream)
{
    if (nx <= 0) { bx = (int)blockIdx.x; nbx = (int)gridDim.x; stream = (int)blockIdx.y; return true; }
    const int id = (int)blockIdx.x, j = id >> 3;
    stream = (id & 7) + 8 * (j / nx); bx = j % nx; nbx = nx;
    return stream < ny;
}
static inline dim3 map_stream_grid(int per_stream, int n_streams, int &nx)
{
    if (n_streams < 8) { nx = 0; return dim3((unsigned)per_stream, (unsigned)n_streams); }
    nx = per_stream;
    return dim3((unsigned)(8 * per_stream * ((n_streams + 7) / 8)));
}

__global__ __launch_bounds__(256) void k_map_correspond(const MapStream *streams, int outer, int nx, int ny)
{
    int bx, nbx, sidx;
    if (!map_block_of(nx, ny, bx, nbx, sidx)) return;
    const MapStream S = streams[sidx];
    __shared__ int s_pre[8][kGroup], s_cst[8][kGroup];
    int ns0, ns1;
    map_stack_sizes(S, ns0, ns1);
    const int nq = ns0 + ns1;
    // (grid-stride over the points: the launch may be sized before the voxel filter's counts are known to the host)
    for (int qi = bx * 8 + (threadIdx.x >> 5); qi < nq; qi += nbx * 8) {
    const int lane = threadIdx.x & 63, gl = threadIdx.x & 31, gbase = lane & ~31;
    const int which = qi < ns0 ? 0 : 1;
    const float4 p = S.stack[which][which ? qi - ns0 : qi];
    int nn[5] = { -1, -1, -1, -1, -1 };
    bool accepted = false;
    const unsigned int mask = (unsigned int)*S.mask[which];
    // the solve is skipped altogether unless the map holds > 10 corner and > 50 surf points (laserMapping.cpp)
    if (S.n_map[0] > 10 && S.n_map[1] > 50 && mask != 0) {
        // pointAssociateToMap: double transform, stored in a float point
        const double *x = S.x;
        double rx, ry, rz;
        quat_rotate(x, (double)p.x, (double)p.y, (double)p.z, rx, ry, rz);
        const float qx = (float)(rx + x[4]), qy = (float)(ry + x[5]), qz = (float)(rz + x[6]);
        const int cqx = (int)floorf(qx * kInvCell), cqy = (int)floorf(qy * kInvCell), cqz = (int)floorf(qz * kInvCell);
        int st = 0, cn = 0;
        if (gl < 27) {
            const unsigned long long kk = cell_key(cqx + gl % 3 - 1, cqy + (gl / 3) % 3 - 1, cqz + gl / 9 - 1);
            unsigned int sl = hash_key(kk) & mask;
            while (true) {
                const GridCell e = S.cell[which][sl];
                if (e.key == kk) { st = e.start; cn = e.cnt; break; }
                if (e.key == kEmptyKey) break;
                sl = (sl + 1) & mask;
            }
        }
        float td[5]; int ti[5];
#pragma unroll
        for (int k = 0; k < 5; k++) { td[k] = __uint_as_float(0x7f800000u); ti[k] = 0x7fffffff; }
        // The candidates of the 27 cells as ONE list (round 4): an inclusive prefix of the cells' counts over the group's lanes, then lane l takes
        // candidates l, l + 32, ... and finds each one's cell by bisection over the prefix (in LDS).  Cell by cell -- 32 lanes on a cell of 2 (surf,
        // 0.8 m voxels) to 15 (corner) points -- a query was 27 dependent load rounds with most lanes idle; the list is 2 to 10.  Same candidates,
        // and the top-5 order is by (distance, index): same neighbours.
        int incl = cn;
#pragma unroll
        for (int o = 1; o < kGroup; o <<= 1) { const int v = __shfl_up(incl, o, kGroup); if (gl >= o) incl += v; }
        const int total = __shfl(incl, kGroup - 1, kGroup);
        int *pre = s_pre[threadIdx.x >> 5], *cst = s_cst[threadIdx.x >> 5];
        pre[gl] = incl; cst[gl] = st - (incl - cn);          // candidate j of cell c sits at cst[c] + j
        __builtin_amdgcn_wave_barrier();                     // one wave: its LDS operations execute in order
        for (int j0 = gl; j0 < total; j0 += 2 * kGroup) {
            int at[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int j = min(j0 + u * kGroup, total - 1);
                int lo = 0;                                  // smallest cell whose inclusive prefix exceeds j
#pragma unroll
                for (int step = 16; step > 0; step >>= 1) lo += (lo + step <= 26 && pre[lo + step - 1] <= j) ? step : 0;
                at[u] = cst[lo] + j;
            }
            float4 cpt[2];
#pragma unroll
            for (int u = 0; u < 2; u++) cpt[u] = S.sorted[which][at[u]];
#pragma unroll
            for (int u = 0; u < 2; u++)
                if (j0 + u * kGroup < total) top5_insert(td, ti, dist2f(cpt[u].x, cpt[u].y, cpt[u].z, qx, qy, qz), __float_as_int(cpt[u].w));
        }
        __builtin_amdgcn_wave_barrier();                     // the next query's prefix is written behind these reads
        // merge the lanes' lists: five rounds of "smallest head wins, its lane pops"
        float d5 = 0.f;
        bool full = true;
#pragma unroll
        for (int r = 0; r < 5; r++) {
            const unsigned long long head = ti[0] == 0x7fffffff ? ~0ull : pack_fu(td[0], (unsigned int)ti[0]);
            const unsigned long long best = group_min_u64(head, gbase);
            if (best == ~0ull) { full = false; break; }
            nn[r] = (int)(unsigned int)(best & 0xffffffffull);
            d5 = __uint_as_float((unsigned int)(best >> 32));
            if (head == best) {
#pragma unroll
                for (int k = 0; k < 4; k++) { td[k] = td[k + 1]; ti[k] = ti[k + 1]; }
                td[4] = __uint_as_float(0x7f800000u); ti[4] = 0x7fffffff;
            }
        }
        accepted = full && (double)d5 < 1.0;
    }
    // the five neighbours (or -1) go to the per-point scratch; k_map_factor turns them into a residual block
    if (gl == 0) {
        int *o = S.nn_tmp + (size_t)qi * 5;
#pragma unroll
        for (int k = 0; k < 5; k++) o[k] = accepted ? nn[k] : -1;
    }
    }
}

// One thread per down-sampled scan point: the line test (covariance of the five neighbours, eigen-decomposition, largest
// > 3 x middle -> LidarEdgeFactor against centre +- 0.1 dir) or the plane fit (all five within 0.2 ->
// LidarPlaneNormFactor), written as an 80-B fp64 record.
__global__ __launch_bounds__(64) void k_map_factor(const MapStream *streams, int outer, int nx, int ny)
{
    int bx, nbx, sidx;
    if (!map_block_of(nx, ny, bx, nbx, sidx)) return;
    const MapStream S = streams[sidx];
    int ns0, ns1;
    map_stack_sizes(S, ns0, ns1);
    const int nq = ns0 + ns1;
    for (int qi = bx * 64 + threadIdx.x; qi < nq; qi += nbx * 64) {
    const int which = qi < ns0 ? 0 : 1;
    const int *nn = S.nn_tmp + (size_t)qi * 5;
    MapRec *rec = S.rec + qi;
    int kind = 0;
    if (nn[0] >= 0) {
        const float4 p = S.stack[which][which ? qi - ns0 : qi];
        double P[15];
        for (int j = 0; j < 5; j++) { const float4 c = S.cloud[which][nn[j]]; P[j * 3] = (double)c.x; P[j * 3 + 1] = (double)c.y; P[j * 3 + 2] = (double)c.z; }
        double ra[3] = { 0, 0, 0 }, rb[3] = { 0, 0, 0 };
        if (which == 0) {
            double c[3] = { 0, 0, 0 };
            for (int j = 0; j < 5; j++) { c[0] += P[j * 3]; c[1] += P[j * 3 + 1]; c[2] += P[j * 3 + 2]; }
            for (int k = 0; k < 3; k++) c[k] = c[k] / 5.0;
            double cov[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
            for (int j = 0; j < 5; j++) {
                const double z[3] = { P[j * 3] - c[0], P[j * 3 + 1] - c[1], P[j * 3 + 2] - c[2] };
                for (int a = 0; a < 3; a++) for (int bb = 0; bb < 3; bb++) cov[a * 3 + bb] += z[a] * z[bb];
            }
            double ev[3], evec[9];
            sym_eig3(cov, ev, evec);
            if (ev[2] > 3.0 * ev[1]) {
                kind = 1;
                for (int k = 0; k < 3; k++) { ra[k] = 0.1 * evec[k * 3 + 2] + c[k]; rb[k] = -0.1 * evec[k * 3 + 2] + c[k]; }
            }
        } else {
            double nrm[3], d;
            if (plane_fit5(P, nrm, d)) { kind = 3; ra[0] = d; for (int k = 0; k < 3; k++) rb[k] = nrm[k]; }
        }
        if (kind != 0) {
            rec->cp[0] = (double)p.x; rec->cp[1] = (double)p.y; rec->cp[2] = (double)p.z;
            for (int k = 0; k < 3; k++) { rec->a[k] = ra[k]; rec->b[k] = rb[k]; }
        }
    }
    rec->kind = kind;
    // block counts: one atomic per wave and kind
    const unsigned long long m1 = __ballot(kind == 1), m3 = __ballot(kind == 3);
    if (threadIdx.x == 0) {
        if (m1) atomicAdd(&S.stats[outer], __popcll(m1));
        if (m3) atomicAdd(&S.stats[2 + outer], __popcll(m3));
    }
    if (S.nn_out) for (int k = 0; k < 5; k++) S.nn_out[(size_t)qi * 5 + k] = kind != 0 ? nn[k] : -1;
    }
}

// ---- solve -----------------------------------------------------------------------------------------------------------
// Same structure as k_lm_solve (odometry.hip): straight-line edge / plane-norm blocks over a per-sweep rotation matrix, the rotation
// part of a Jacobian row as 2 (Rv x d), the accepted / candidate sums in LDS, packed in-place Cholesky.  No FMA contraction here: the
// tests compare this kernel's LM iteration counts with the oracle's, and a contracted trace can stop one iteration apart on a knife edge.
__device__ __forceinline__ void map_row(LmAcc &acc, double d0, double d1, double d2, double rvx, double rvy, double rvz, double res, double sr)
{
    double J[6];
    const double s2 = 2.0 * sr;
    J[0] = (rvy * d2 - rvz * d1) * s2; J[1] = (rvz * d0 - rvx * d2) * s2; J[2] = (rvx * d1 - rvy * d0) * s2;
    J[3] = d0 * sr; J[4] = d1 * sr; J[5] = d2 * sr;
    accumulate_row(acc, J, res * sr);
}

template <bool kJac, bool kEdge>
__device__ __forceinline__ void map_eval_block(const MapRec &R, const double *Rm, const double *x, LmAcc &acc)
{
    if (R.kind == 0) return;
    const double vx = R.cp[0], vy = R.cp[1], vz = R.cp[2];
    const double rvx = Rm[0] * vx + Rm[1] * vy + Rm[2] * vz, rvy = Rm[3] * vx + Rm[4] * vy + Rm[5] * vz, rvz = Rm[6] * vx + Rm[7] * vy + Rm[8] * vz;
    const double lx = rvx + x[4], ly = rvy + x[5], lz = rvz + x[6];
    double r0, r1 = 0.0, r2 = 0.0, sq, e0 = 0.0, e1 = 0.0, e2 = 0.0;
    if (kEdge) {
        const double ax = lx - R.a[0], ay = ly - R.a[1], az = lz - R.a[2];
        const double bx = lx - R.b[0], by = ly - R.b[1], bz = lz - R.b[2];
        const double ex = R.a[0] - R.b[0], ey = R.a[1] - R.b[1], ez = R.a[2] - R.b[2];
        const double den = sqrt(ex * ex + ey * ey + ez * ez);
        r0 = (ay * bz - az * by) / den; r1 = (az * bx - ax * bz) / den; r2 = (ax * by - ay * bx) / den;
        sq = r0 * r0 + r1 * r1 + r2 * r2;
        if (kJac) { const double inv = 1.0 / den; e0 = ex * inv; e1 = ey * inv; e2 = ez * inv; }
    } else {
        // LidarPlaneNormFactor: norm . point_w + negative_OA_dot_norm
        r0 = (R.b[0] * lx + R.b[1] * ly + R.b[2] * lz) + R.a[0];
        sq = r0 * r0;
    }
    double rho0, rho1;
    huber(sq, rho0, rho1);
    acc.cost += 0.5 * rho0;
    if (!kJac) return;
    const double sr = rho1 == 1.0 ? 1.0 : sqrt(rho1);      // inliers: sqrt(1) = 1 exactly, without the 25-instruction fp64 square root
    if (kEdge) {
        map_row(acc, 0.0, e2, -e1, rvx, rvy, rvz, r0, sr);
        map_row(acc, -e2, 0.0, e0, rvx, rvy, rvz, r1, sr);
        map_row(acc, e1, -e0, 0.0, rvx, rvy, rvz, r2, sr);
    } else
        map_row(acc, R.b[0], R.b[1], R.b[2], rvx, rvy, rvz, r0, sr);
}

// threads of k_map_solve: 1024 / 512 / 256 measured 1.65 / 1.50 / 1.55 ms per laserMapping frame (at 1024 the 128-register budget spills 152 VGPRs)
#ifndef LMONO_MS_T
#define LMONO_MS_T 512
#endif
constexpr int kMsT = LMONO_MS_T;
// (records per thread in flight 1 / 2 / 4, round 4: 1253 / 1174 / 1220 frames/s -- the evaluation is bound by its fp64 work on one CU, not by the loads)
// sums of all residual blocks of a stream into LDS (s_sum: H 21 | g 6 | cost); blocks [0, n_edge) are the corner points' (edges)
// Round 4: one stream's solve is a CLUSTER of K workgroups on one XCD (block b runs on XCD b mod 8).  Every workgroup runs the whole trust-region
// control flow on the same sums, so they all take the same branches; an evaluation is split between them (chunks of kMsT blocks in turn), each leaves
// its 28 partial sums in L2, ONE cluster barrier (an arrival counter per evaluation, agent-scope release / acquire), and everybody adds the K partials in
// rank order.  One workgroup alone was bound by the fp64 work of ~11 k residual blocks x <= 5 evaluations on a single CU (0.13 ms, twice per frame).
struct MsCluster { int K, rank, eval; double *part; unsigned int *bar; int *fail; };

// The barrier is wave 0's business alone: its lanes 0..27 hold the workgroup's partial sums, so only that wave releases (one L2 write-back, not
// eight), lane 0 arrives and polls with RELAXED loads (an acquire per poll is a cache invalidate per poll), one acquire behind the loop, and the rest of
// the workgroup waits at the workgroup barrier that follows in map_evaluate.
__device__ __forceinline__ void ms_cluster_barrier_wave0(MsCluster &cl)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");       // the partial sums are visible device-wide before the arrival
    if (threadIdx.x == 0) {
        unsigned int *ctr = cl.bar + cl.eval;
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // every workgroup of the cluster is resident by construction (the host keeps clusters x K within half the CUs); the spin is bounded all the
        // same -- a wave that never finishes can take the whole GPU down: after a few seconds the solve is marked failed and the barrier opens
        int spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned int)cl.K) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 26)) { *cl.fail = 1; break; }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");       // the other workgroups' partial sums are read with plain loads
}

template <bool kJac>
__device__ __forceinline__ void map_evaluate(const MapRec *rec, int n_edge, int nq, const double *x, double (*s_red)[28], double *s_sum, MsCluster &cl)
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
    double Rm[9];
    {
        const double ux = x[0], uy = x[1], uz = x[2], w = x[3];
        Rm[0] = 1.0 - 2.0 * (uy * uy + uz * uz); Rm[1] = 2.0 * (ux * uy - w * uz);       Rm[2] = 2.0 * (ux * uz + w * uy);
        Rm[3] = 2.0 * (ux * uy + w * uz);       Rm[4] = 1.0 - 2.0 * (ux * ux + uz * uz); Rm[5] = 2.0 * (uy * uz - w * ux);
        Rm[6] = 2.0 * (ux * uz - w * uy);       Rm[7] = 2.0 * (uy * uz + w * ux);       Rm[8] = 1.0 - 2.0 * (ux * ux + uy * uy);
    }
    // The blocks are dealt in chunks of kMsT to kMsMaxK = 8 VIRTUAL ranks in turn, whatever the cluster size is (round 6): virtual rank v
    // sums the chunks v, v + 8, .. -- one block per thread and chunk, the thread's blocks in ascending order, the wave by DPP, the waves in wave order --
    // and the evaluation's sums are the eight virtual ranks' added in rank order.  A workgroup of a K-cluster computes the virtual ranks rank, rank + K, ..
    // one after the other, so the BYTES of a stream's solve depend on the stream alone -- not on K, i.e. not on how many other streams share the call
    // (rounds 4-5 split the blocks by the REAL rank: a stream's rounding changed with the batch it travelled in, bounded at 1e-12).  K = 8 (one stream)
    // does exactly the work it did; K = 1 (64+ streams) pays seven more reductions per evaluation.
    __shared__ double s_vp[kMsMaxK][28];
    for (int v = cl.rank; v < kMsMaxK; v += cl.K) {
        acc.cost = 0.0;
        if (kJac) {
#pragma unroll
            for (int i = 0; i < 21; i++) acc.H[i] = 0.0;
#pragma unroll
            for (int i = 0; i < 6; i++) acc.g[i] = 0.0;
        }
        for (int q0 = v * kMsT; q0 < nq; q0 += kMsMaxK * kMsT) {
            const int qi = q0 + tid;
            if (qi >= nq) break;
            if (qi < n_edge) map_eval_block<kJac, true>(rec[qi], Rm, x, acc);
            else map_eval_block<kJac, false>(rec[qi], Rm, x, acc);
        }
        acc.cost = wave_sum_d_lane63(acc.cost);
        if (kJac) {
#pragma unroll
            for (int i = 0; i < 21; i++) acc.H[i] = wave_sum_d_lane63(acc.H[i]);
#pragma unroll
            for (int i = 0; i < 6; i++) acc.g[i] = wave_sum_d_lane63(acc.g[i]);
        }
        __syncthreads();
        if (lane == 63) {
            s_red[wave][27] = acc.cost;
            if (kJac) {
                for (int i = 0; i < 21; i++) s_red[wave][i] = acc.H[i];
                for (int i = 0; i < 6; i++) s_red[wave][21 + i] = acc.g[i];
            }
        }
        __syncthreads();
        if (wave == 0) {
            const bool mine = tid < 28 && (kJac || tid == 27);
            double t = 0.0;
            if (mine) for (int w = 0; w < kMsT / 64; w++) t += s_red[w][tid];
            if (mine) { if (cl.K > 1) cl.part[(cl.eval * kMsMaxK + v) * 28 + tid] = t; else s_vp[v][tid] = t; }
        }
    }
    if (wave == 0) {
        const bool mine = tid < 28 && (kJac || tid == 27);
        double t = 0.0;
        if (cl.K > 1) {
            ms_cluster_barrier_wave0(cl);
            if (mine) for (int v = 0; v < kMsMaxK; v++) t += cl.part[(cl.eval * kMsMaxK + v) * 28 + tid];
        } else {
            // (wave 0 wrote s_vp itself: its LDS operations execute in order)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (mine) for (int v = 0; v < kMsMaxK; v++) t += s_vp[v][tid];
        }
        if (mine) s_sum[tid] = t;
    }
    cl.eval++;
    __syncthreads();
}

// One kMsT-thread workgroup per stream (a frame has ~8 k residual blocks); the trust-region control flow runs redundantly
// in every thread on the block-reduced sums in LDS, exactly like k_lm_solve.
// grid: ceil(n_streams / 8) x 8 K blocks; block b = 8 K g + 8 rank + j works on stream 8 g + j: the K workgroups of a stream are 8 blocks apart
__global__ __launch_bounds__(kMsT) void k_map_solve(const MapStream *streams, int n_streams, int K, int outer)
{
    const int sidx = (int)(blockIdx.x / (8 * K)) * 8 + (int)(blockIdx.x % 8);
    if (sidx >= n_streams) return;
    const MapStream S = streams[sidx];
    MsCluster cl;
    cl.K = K; cl.rank = (int)(blockIdx.x % (8 * K)) / 8; cl.eval = 0; cl.part = S.part; cl.bar = S.bar + outer * 8; cl.fail = S.stats + 6;
    __shared__ double s_red[kMsT / 64][28], s_sum[28], s_cur[28];
    const int tid = threadIdx.x;
    int ns0, ns1;
    map_stack_sizes(S, ns0, ns1);
    const int n_edge = ns0, nq = ns0 + ns1;
    const int n_used = S.stats[outer] + S.stats[2 + outer];
    double x[7];
    for (int i = 0; i < 7; i++) x[i] = S.x[i];
    const int max_iter = 4;
    const double function_tol = 1e-6, gradient_tol = 1e-10, parameter_tol = 1e-8;
    const double min_rel_decrease = 1e-3, min_diag = 1e-6, max_diag = 1e32, max_radius = 1e16, min_radius = 1e-32;
    double radius = 1e4, decrease_factor = 2.0;
    bool reuse_diagonal = false;
    int invalid_steps = 0, iter = 0;
    // Round 5: the trust-region arithmetic between two evaluations (~1.5 k dependent fp64 instructions) runs on wave 0 alone -- every wave used to repeat it,
    // two waves per SIMD taking turns at the issue port -- and hands the next candidate to the workgroup through LDS: ctl 0 = evaluate it with its Jacobian
    // (the next linearisation when it is accepted), 1 = its cost only (the last iteration), 2 = finished.  Same arithmetic, same order of decisions.
    __shared__ double s_xc[8];
    __shared__ int s_ctl;
    if (n_used > 0) {
        map_evaluate<true>(S.rec, n_edge, nq, x, s_red, s_sum, cl);
        const bool w0 = tid < 64;
        bool first = true, last = false;
        double x_cost = 0.0, x_norm = 0.0, model_change = 0.0, scale[6] = { 0, 0, 0, 0, 0, 0 }, diag[6] = { 0, 0, 0, 0, 0, 0 }, cand[7] = { 0, 0, 0, 0, 0, 0, 0 };
        for (;;) {
            if (w0) {
                auto take_linearisation = [&]() {          // s_cur <- s_sum inside the wave (its LDS operations execute in order)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    if (tid < 28) s_cur[tid] = s_sum[tid];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                };
                bool proceed = true;
                if (first) {
                    take_linearisation();
                    x_cost = s_cur[27];
                    double gmax = 0.0;
                    for (int i = 0; i < 6; i++) gmax = fmax(gmax, fabs(s_cur[21 + i]));
                    if (gmax > gradient_tol) {
                        x_norm = norm7(x);
                        for (int i = 0; i < 6; i++) scale[i] = 1.0 / (1.0 + sqrt(s_cur[sym6(i, i)]));
                    } else proceed = false;
                } else {
                    const double cand_cost = s_sum[27];
                    double sn = 0.0;
                    for (int i = 0; i < 7; i++) sn += (x[i] - cand[i]) * (x[i] - cand[i]);
                    sn = sqrt(sn);
                    if (sn <= parameter_tol * (x_norm + parameter_tol)) proceed = false;
                    else if (fabs(x_cost - cand_cost) <= function_tol * x_cost) proceed = false;
                    else {
                        const double rel = (x_cost - cand_cost) / model_change;
                        if (rel > min_rel_decrease) {
                            for (int i = 0; i < 7; i++) x[i] = cand[i];
                            if (last) proceed = false;
                            else {
                                x_norm = norm7(x);
                                x_cost = cand_cost;
                                take_linearisation();
                                const double tt = 2.0 * rel - 1.0;
                                double den = 1.0 - tt * tt * tt;
                                if (den < 1.0 / 3.0) den = 1.0 / 3.0;
                                radius = radius / den;
                                if (radius > max_radius) radius = max_radius;
                                decrease_factor = 2.0; reuse_diagonal = false;
                                double gmax = 0.0;
                                for (int i = 0; i < 6; i++) gmax = fmax(gmax, fabs(s_cur[21 + i]));
                                if (gmax <= gradient_tol) proceed = false;
                            }
                        } else {
                            radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
                        }
                        if (proceed && radius <= min_radius) proceed = false;
                    }
                }
                int ctl = 2;
                while (proceed) {
                    if (iter >= max_iter) break;
                    iter++;
                    double gs[6], A[21], stepv[6];
#pragma unroll
                    for (int i = 0; i < 6; i++) {
                        gs[i] = s_cur[21 + i] * scale[i];
#pragma unroll
                        for (int j = 0; j <= i; j++) A[i * (i + 1) / 2 + j] = s_cur[sym6(i, j)] * scale[i] * scale[j];
                    }
                    if (!reuse_diagonal)
                        for (int i = 0; i < 6; i++) { double d = A[i * (i + 1) / 2 + i]; d = d < min_diag ? min_diag : d; d = d > max_diag ? max_diag : d; diag[i] = d; }
                    for (int i = 0; i < 6; i++) A[i * (i + 1) / 2 + i] += diag[i] / radius;
                    bool ok = chol_solve6_packed(A, gs, stepv);
                    for (int i = 0; i < 6; i++) if (!isfinite(stepv[i])) ok = false;
                    model_change = 0.0;
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
                    double delta[6];
                    for (int i = 0; i < 6; i++) delta[i] = stepv[i] * scale[i];
                    manifold_plus(x, delta, cand);
                    // candidate evaluated with its Jacobian (reused as the next linearisation); the last iteration needs the cost only
                    last = iter == max_iter;
                    ctl = last ? 1 : 0;
                    break;
                }
                if (tid == 0) { s_ctl = ctl; for (int i = 0; i < 7; i++) s_xc[i] = cand[i]; }
            }
            __syncthreads();
            const int ctl = s_ctl;
            if (ctl == 2) break;
            double xc[7];
            for (int i = 0; i < 7; i++) xc[i] = s_xc[i];
            if (ctl == 1) map_evaluate<false>(S.rec, n_edge, nq, xc, s_red, s_sum, cl);
            else map_evaluate<true>(S.rec, n_edge, nq, xc, s_red, s_sum, cl);
            first = false;
        }
    }
    __syncthreads();
    if (tid == 0 && cl.rank == 0) {
        for (int i = 0; i < 7; i++) S.x[i] = x[i];
        S.stats[4 + outer] = iter;
    }
}

// workgroups per stream: K = 8 (measured best for one stream, below), lowered until clusters x K stay within half the chip's 256 CUs -- every workgroup
// of a cluster must be resident while it spins; the spin is bounded (a cluster that is not resident sets stats[6] and the host returns LMONO_ENODEV).
// A stream's sums are formed per VIRTUAL rank (always eight, map_evaluate), so its bytes do not depend on K -- on how many streams share the call
// (tests/test_mapping_gpu.py::test_solve_cluster_sizes_agree: byte equality since round 6); the budget comes from the device's CU count (lmono_ctx::map_budget).
static inline int map_solve_cluster(int n_streams, int budget)
{
    static const int forced = [] { const char *e = getenv("LMONO_MAP_SOLVE_K"); return e ? atoi(e) : 0; }();        // measurement switch
    int K = forced > 0 ? forced : 8;          // one stream, K = 2 / 4 / 6 / 8: 2.24 / 2.45 / 2.52 / 2.56 k frames/s (round 5, the trust-region step on one wave; round 4's
                                              // kernel, where every wave repeated it, peaked at K = 4: 1.57 / 1.71 / 1.98 / 1.83 k for K = 1 / 2 / 4 / 8)
    if (K > kMsMaxK) K = kMsMaxK;
    const int groups = (n_streams + 7) / 8;
    while (K > 1 && groups * 8 * K > budget) K--;          // budget: workgroups that may spin on each other at once (lmono_ctx::map_budget, from the device's CU count)
    return K;
}
static inline void launch_map_solve(hipStream_t st, const MapStream *S_d, int n_streams, int outer, int budget)
{
    const int K = map_solve_cluster(n_streams, budget);
    hipLaunchKernelGGL(k_map_solve, dim3((unsigned)(((n_streams + 7) / 8) * 8 * K)), dim3(kMsT), 0, st, S_d, n_streams, K, outer);
}

} // namespace lmono

// ---- pcl::VoxelGrid on an arbitrary cloud (laserMapping's downSizeFilterCorner / downSizeFilterSurf) -------
// bounding box -> PCL's linear cell index -> stable LSD radix sort of (cell, point index) by cell -> one lane per run of equal cells sums its points
// in index order (PCL's float summation order) and writes the centroids in ascending cell order.  Bit-exact against oracle/lo_scanreg.c
// lo_voxel_filter.
//
// Round 4: a cloud is worked on by ceil(n / 2048) workgroups (a "tile" each) in a chain of short launches
//     k_vox_box | k_vox_keys (+ counts of pass 0) | [k_vox_count |] k_vox_pass  x passes | k_vox_heads | k_vox_centroids
// instead of one 1024-thread workgroup per cloud walking it end to end (0.32 ms per 25 k-point cloud in round 3, 0.23 ms with 8-bit digits: a chain of
// ~1 us global round trips on ONE compute unit, twice per laserMapping frame).  A pass reads the (tile, digit) counts of all tiles of its cloud
// (<= 32 x 512), turns them into its own first output position per digit, and ranks its 2048 elements by (wave, block of 64, lane) = index order:
// stable.  (A pass that added its scattered elements to the NEXT pass's counts with global atomics -- it knows the tile each of them lands in -- saved the
// counting launches but was slower: 17 us instead of 9 us per pass on one stream, bound by the atomics' rate over 64 streams; device-scope atomics
// are resolved beyond the XCDs' L2s.)
// The digit width adapts to the key: passes = ceil(bits / 9), width = ceil(bits / passes) (26-bit scan-cloud keys: three passes of 9 bits).
namespace lmono {

constexpr int kVoxCloudMax = 65536;
constexpr int kVxT = 256;                                // threads per workgroup
constexpr int kVxTile = 2048;                            // elements per workgroup: 8 per thread, 8 blocks of 64 per wave
constexpr int kVxBlocks = kVxTile / kVxT;                // blocks of 64 per wave
constexpr int kVxMaxTiles = kVoxCloudMax / kVxTile;      // 32
constexpr int kVxMaxDigit = 512;                         // 9-bit digits at most
constexpr int kVxHdr = 16;                               // ints of a job's header in its workspace
constexpr int kVxWsTile = 16 + kVxMaxDigit;              // ints of workspace per tile: box[8], heads[1] (+7 pad), one row of digit counts

struct VoxJob {
    const float4 *in; int n;
    float inv_leaf;
    float4 *out;           // capacity n
    int *n_out;            // number of centroids; -1: cloud rejected (too large, or more passes than were launched)
    unsigned int *key_a, *key_b;   // scratch [n] cell keys (ping-pong)
    int *idx_a, *idx_b;            // scratch [n] point indices (ping-pong)
    int *ws;               // workspace: kVxHdr + tiles * kVxWsTile ints (vox_ws_ints)
};

static inline size_t vox_ws_ints(int64_t n) { return (size_t)kVxHdr + (size_t)((n + kVxTile - 1) / kVxTile > 0 ? (n + kVxTile - 1) / kVxTile : 1) * kVxWsTile; }

struct VoxView {           // a job's workspace
    int *hdr;              // minb[3], mul1, mul2, passes, width
    float *box;            // [tiles][8]
    int *heads;            // [tiles][8] (first used)
    int *hist;             // [tiles][kVxMaxDigit]: the tiles' digit counts of the pass at hand
    int tiles;
};

__device__ __forceinline__ VoxView vox_view(const VoxJob &J)
{
    VoxView V;
    V.tiles = (J.n + kVxTile - 1) / kVxTile;
    V.hdr = J.ws; V.box = (float *)(J.ws + kVxHdr); V.heads = J.ws + kVxHdr + 8 * V.tiles; V.hist = J.ws + kVxHdr + 16 * V.tiles;
    return V;
}

// per-tile bounding boxes
__device__ __forceinline__ void vox_box_tile(const VoxJob *jobs, const int entry)
{
    const VoxJob J = jobs[entry >> 6];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = J.n, tile = entry & 63;
    if (n <= 0 || n > kVoxCloudMax) { if (tile == 0 && tid == 0) *J.n_out = n <= 0 ? 0 : -1; return; }
    const VoxView V = vox_view(J);
    if (tile >= V.tiles) return;
    typedef __attribute__((address_space(1))) const float4 GF4;
    const float4 *gin = (const float4 *)(GF4 *)J.in;
    __shared__ float s_box[4][6];
    float mn[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, mx[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    float4 p[kVxBlocks];
#pragma unroll
    for (int u = 0; u < kVxBlocks; u++) p[u] = gin[min(tile * kVxTile + kVxT * u + tid, n - 1)];          // a clamped slot repeats the last point: harmless for min / max
#pragma unroll
    for (int u = 0; u < kVxBlocks; u++) {
        mn[0] = fminf(mn[0], p[u].x); mn[1] = fminf(mn[1], p[u].y); mn[2] = fminf(mn[2], p[u].z);
        mx[0] = fmaxf(mx[0], p[u].x); mx[1] = fmaxf(mx[1], p[u].y); mx[2] = fmaxf(mx[2], p[u].z);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int k = 0; k < 3; k++) { mn[k] = fminf(mn[k], __shfl_xor(mn[k], o)); mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], o)); }
    if (lane == 0) for (int k = 0; k < 3; k++) { s_box[wave][k] = mn[k]; s_box[wave][3 + k] = mx[k]; }
    __syncthreads();
    if (tid < 6) {
        float v = s_box[0][tid];
        for (int w = 1; w < 4; w++) v = tid < 3 ? fminf(v, s_box[w][tid]) : fmaxf(v, s_box[w][tid]);
        V.box[tile * 8 + tid] = v;
    }
}

// cell keys in index order + the tile's digit counts of pass 0
__device__ __forceinline__ void vox_keys_tile(const VoxJob *jobs, const int entry)
{
    const VoxJob J = jobs[entry >> 6];
    const int tid = threadIdx.x, n = J.n, tile = entry & 63;
    if (n <= 0 || n > kVoxCloudMax) return;
    const VoxView V = vox_view(J);
    if (tile >= V.tiles) return;
    typedef __attribute__((address_space(1))) const float4 GF4;
    typedef __attribute__((address_space(1))) unsigned int GU;
    typedef __attribute__((address_space(1))) int GI;
    const float4 *gin = (const float4 *)(GF4 *)J.in;
    GU *ka = (GU *)J.key_a;
    GI *ia = (GI *)J.idx_a;
    __shared__ float s_box[6];
    __shared__ int s_h[kVxMaxDigit];
    float4 p[kVxBlocks];
#pragma unroll
    for (int u = 0; u < kVxBlocks; u++) p[u] = gin[min(tile * kVxTile + kVxT * u + tid, n - 1)];
    if (tid < 64) {                                   // the cloud's box from the tiles' boxes (<= 32 of them)
        float v[6];
#pragma unroll
        for (int k = 0; k < 6; k++) v[k] = V.box[min(tid, V.tiles - 1) * 8 + k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int k = 0; k < 6; k++) v[k] = k < 3 ? fminf(v[k], __shfl_xor(v[k], o)) : fmaxf(v[k], __shfl_xor(v[k], o));
        if (tid == 0) for (int k = 0; k < 6; k++) s_box[k] = v[k];
    }
    for (int d = tid; d < kVxMaxDigit; d += kVxT) s_h[d] = 0;
    __syncthreads();
    const float inv = J.inv_leaf;
    const int minb0 = (int)floorf(s_box[0] * inv), minb1 = (int)floorf(s_box[1] * inv), minb2 = (int)floorf(s_box[2] * inv);
    const int div0 = (int)floorf(s_box[3] * inv) - minb0 + 1, div1 = (int)floorf(s_box[4] * inv) - minb1 + 1, div2 = (int)floorf(s_box[5] * inv) - minb2 + 1;
    const int mul1 = div0, mul2 = div0 * div1;
    // the highest cell index decides the passes and the digit width
    const unsigned int max_cell = (unsigned int)((div0 - 1) + (div1 - 1) * mul1 + (div2 - 1) * mul2);
    int bits = 0;
    while (bits < 32 && (max_cell >> bits) != 0u) bits++;
    if (bits < 1) bits = 1;
    const int passes = (bits + 8) / 9, width = (bits + passes - 1) / passes;
    if (tile == 0 && tid == 0) { V.hdr[0] = passes; V.hdr[1] = width; }
    const unsigned int dmask = (1u << width) - 1u;
#pragma unroll
    for (int u = 0; u < kVxBlocks; u++) {
        const int i = tile * kVxTile + kVxT * u + tid;
        if (i >= n) break;
        const int i0c = (int)(floorf(p[u].x * inv) - (float)minb0);
        const int i1c = (int)(floorf(p[u].y * inv) - (float)minb1);
        const int i2c = (int)(floorf(p[u].z * inv) - (float)minb2);
        const unsigned int key = (unsigned int)(i0c + i1c * mul1 + i2c * mul2);
        ka[i] = key;
        ia[i] = i;
        atomicAdd(&s_h[key & dmask], 1);
    }
    __syncthreads();
    // this tile's digit counts of pass 0
    for (int d = tid; d < kVxMaxDigit; d += kVxT) V.hist[tile * kVxMaxDigit + d] = s_h[d];
}

// digit counts of one tile for pass `pass` (pass 0's come from k_vox_keys)
__device__ __forceinline__ void vox_count_tile(const VoxJob *jobs, const int entry, int pass)
{
    const VoxJob J = jobs[entry >> 6];
    const int tid = threadIdx.x, n = J.n, tile = entry & 63;
    if (n <= 0 || n > kVoxCloudMax) return;
    const VoxView V = vox_view(J);
    if (tile >= V.tiles) return;
    const int passes = V.hdr[0], width = V.hdr[1];
    if (pass >= passes) return;
    typedef __attribute__((address_space(1))) const unsigned int GU;
    GU *ka = (GU *)((pass & 1) ? J.key_b : J.key_a);
    __shared__ int s_h[kVxMaxDigit];
    unsigned int kk[kVxBlocks];
#pragma unroll
    for (int u = 0; u < kVxBlocks; u++) kk[u] = ka[min(tile * kVxTile + kVxT * u + tid, n - 1)];
    for (int d = tid; d < kVxMaxDigit; d += kVxT) s_h[d] = 0;
    __syncthreads();
    const int sh = pass * width;
    const unsigned int dmask = (1u << width) - 1u;
#pragma unroll
    for (int u = 0; u < kVxBlocks; u++) if (tile * kVxTile + kVxT * u + tid < n) atomicAdd(&s_h[(kk[u] >> sh) & dmask], 1);
    __syncthreads();
    int *row = V.hist + (size_t)tile * kVxMaxDigit;
    for (int d = tid; d < (1 << width); d += kVxT) row[d] = s_h[d];
}

// one stable pass over one tile
__device__ __forceinline__ void vox_pass_tile(const VoxJob *jobs, const int entry, int pass)
{
    const VoxJob J = jobs[entry >> 6];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = J.n, tile = entry & 63;
    if (n <= 0 || n > kVoxCloudMax) return;
    const VoxView V = vox_view(J);
    if (tile >= V.tiles) return;
    const int passes = V.hdr[0], width = V.hdr[1];
    if (pass >= passes) return;
    typedef __attribute__((address_space(1))) unsigned int GU;
    typedef __attribute__((address_space(1))) int GI;
    GU *ka = (GU *)((pass & 1) ? J.key_b : J.key_a), *kb = (GU *)((pass & 1) ? J.key_a : J.key_b);
    GI *ia = (GI *)((pass & 1) ? J.idx_b : J.idx_a), *ib = (GI *)((pass & 1) ? J.idx_a : J.idx_b);
    const int sh = pass * width, D = 1 << width;
    const unsigned int dmask = (unsigned int)D - 1u;
    const int *h_cur = V.hist;
    __shared__ int s_cnt[4][kVxMaxDigit];       // [wave][digit]: counts, then cursors
    __shared__ int s_wtot[4];
    // the wave's eight blocks of 64 consecutive elements
    const int w_lo = tile * kVxTile + wave * (kVxTile / 4), w_hi = min(w_lo + kVxTile / 4, n);
    unsigned int kk[kVxBlocks];
    int ii[kVxBlocks];
#pragma unroll
    for (int u = 0; u < kVxBlocks; u++) { const int ic = min(w_lo + 64 * u + lane, n - 1); kk[u] = ka[ic]; ii[u] = ia[ic]; }
    for (int d = tid; d < 4 * kVxMaxDigit; d += kVxT) (&s_cnt[0][0])[d] = 0;
    // first output position of every digit for this tile: (all tiles' counts of smaller digits) + (earlier tiles' counts of this digit)
    const int per = D > kVxT ? 2 : 1;                 // digits per thread: 2 tid, 2 tid + 1 for 9-bit digits
    int tot[2] = { 0, 0 }, before[2] = { 0, 0 };
    for (int q = 0; q < per; q++) {
        const int d = per * tid + q;
        if (d < D)
            for (int t0 = 0; t0 < V.tiles; t0 += 8) {                  // eight tiles' counts in flight (one at a time it was a ~1 us round trip per tile)
                int cth[8];
#pragma unroll
                for (int v = 0; v < 8; v++) cth[v] = h_cur[min(t0 + v, V.tiles - 1) * kVxMaxDigit + d];
#pragma unroll
                for (int v = 0; v < 8; v++) { const int t = t0 + v; tot[q] += t < V.tiles ? cth[v] : 0; before[q] += t < tile ? cth[v] : 0; }
            }
    }
    __syncthreads();
    const unsigned long long lt = (1ull << lane) - 1ull;
    auto peers_of = [&](bool ok, unsigned int dgt) -> unsigned long long {
        unsigned long long pm = __ballot(ok);
        for (int bq = 0; bq < width; bq++) {
            const unsigned long long m = __ballot(ok && ((dgt >> bq) & 1u));
            pm &= ((dgt >> bq) & 1u) ? m : ~m;
        }
        return ok ? pm : 0ull;
    };
    unsigned long long pm[kVxBlocks];
#pragma unroll
    for (int u = 0; u < kVxBlocks; u++) {
        const bool ok = w_lo + 64 * u + lane < w_hi;
        const unsigned int dgt = (kk[u] >> sh) & dmask;
        pm[u] = peers_of(ok, dgt);
        if (ok && (pm[u] & lt) == 0ull) s_cnt[wave][dgt] += __popcll(pm[u]);        // the digit's lowest lane of the block; one wave, one row, block order
    }
    {
        const int local = tot[0] + tot[1];
        const int incl = wave_scan_incl(local);
        if (lane == 63) s_wtot[wave] = incl;
        __syncthreads();
        int run = incl - local;
        for (int w = 0; w < wave; w++) run += s_wtot[w];
        for (int q = 0; q < per; q++) {
            const int d = per * tid + q;
            if (d < D) {
                int cur = run + before[q];
                for (int w = 0; w < 4; w++) { const int cw = s_cnt[w][d]; s_cnt[w][d] = cur; cur += cw; }
            }
            run += tot[q];
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kVxBlocks; u++) {
        const bool ok = w_lo + 64 * u + lane < w_hi;
        const unsigned int dgt = (kk[u] >> sh) & dmask;
        const bool lead = ok && (pm[u] & lt) == 0ull;
        int base = 0;
        if (lead) { base = s_cnt[wave][dgt]; s_cnt[wave][dgt] = base + __popcll(pm[u]); }
        base = __shfl(base, ok ? __ffsll((long long)pm[u]) - 1 : lane);
        const int dst = base + __popcll(pm[u] & lt);
        if (ok && (unsigned int)dst < (unsigned int)n) {       // (the counts add up to n: the bound only keeps a corrupted workspace from becoming a stray write)
            kb[dst] = kk[u]; ib[dst] = ii[u];
        }
    }
}

// run heads of a tile
__device__ __forceinline__ void vox_heads_tile(const VoxJob *jobs, const int entry, int max_passes)
{
    const VoxJob J = jobs[entry >> 6];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = J.n, tile = entry & 63;
    if (n <= 0 || n > kVoxCloudMax) return;
    const VoxView V = vox_view(J);
    if (tile >= V.tiles) return;
    const int passes = V.hdr[0];
    if (passes > max_passes) return;
    typedef __attribute__((address_space(1))) const unsigned int GU;
    GU *ka = (GU *)((passes & 1) ? J.key_b : J.key_a);
    __shared__ int s_h[4];
    int heads = 0;
    unsigned int k[kVxBlocks], kp[kVxBlocks];
#pragma unroll
    for (int u = 0; u < kVxBlocks; u++) { const int ic = min(tile * kVxTile + kVxT * u + tid, n - 1); k[u] = ka[ic]; kp[u] = ka[max(ic - 1, 0)]; }
#pragma unroll
    for (int u = 0; u < kVxBlocks; u++) { const int i = tile * kVxTile + kVxT * u + tid; heads += (i < n && (i == 0 || k[u] != kp[u])) ? 1 : 0; }
    heads = wave_sum_i(heads);
    if (lane == 0) s_h[wave] = heads;
    __syncthreads();
    if (tid == 0) V.heads[tile * 8] = s_h[0] + s_h[1] + s_h[2] + s_h[3];
}

// runs of equal cells -> centroids, in ascending cell order
__device__ __forceinline__ void vox_centroids_tile(const VoxJob *jobs, const int entry, int max_passes)
{
    const VoxJob J = jobs[entry >> 6];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = J.n, tile = entry & 63;
    if (n <= 0 || n > kVoxCloudMax) return;
    const VoxView V = vox_view(J);
    if (tile >= V.tiles) return;
    const int passes = V.hdr[0];
    if (passes > max_passes) { if (tile == 0 && tid == 0) *J.n_out = -1; return; }
    typedef __attribute__((address_space(1))) const float4 GF4;
    typedef __attribute__((address_space(1))) float4 GF4W;
    typedef __attribute__((address_space(1))) const unsigned int GU;
    typedef __attribute__((address_space(1))) const int GI;
    const float4 *gin = (const float4 *)(GF4 *)J.in;
    float4 *gout = (float4 *)(GF4W *)J.out;
    GU *ka = (GU *)((passes & 1) ? J.key_b : J.key_a);
    GI *ia = (GI *)((passes & 1) ? J.idx_b : J.idx_a);
    __shared__ int s_h[4];
    const int w_lo = tile * kVxTile + wave * (kVxTile / 4), w_hi = min(w_lo + kVxTile / 4, n);
    const unsigned long long lt = (1ull << lane) - 1ull;
    // every lane fetches ITS elements of the wave's eight blocks (key, key before, index: coalesced; point: gathered), all in flight together
    unsigned int c[kVxBlocks], kp[kVxBlocks];
    int jx[kVxBlocks];
#pragma unroll
    for (int u = 0; u < kVxBlocks; u++) { const int ic = min(w_lo + 64 * u + lane, n - 1); c[u] = ka[ic]; kp[u] = ka[max(ic - 1, 0)]; jx[u] = ia[ic]; }
    int before = 0;
    for (int t = lane; t < tile; t += 64) before += V.heads[t * 8];
    float4 pt[kVxBlocks];
#pragma unroll
    for (int u = 0; u < kVxBlocks; u++) pt[u] = gin[min((unsigned int)jx[u], (unsigned int)(n - 1))];
    before = wave_sum_i(before);
    int heads = 0;
#pragma unroll
    for (int u = 0; u < kVxBlocks; u++) { const int i = w_lo + 64 * u + lane; heads += __popcll(__ballot(i < w_hi && (i == 0 || c[u] != kp[u]))); }
    if (lane == 0) s_h[wave] = heads;
    __syncthreads();
    int o = before;
    for (int w = 0; w < wave; w++) o += s_h[w];
    if (tile == V.tiles - 1 && tid == kVxT - 1) *J.n_out = o + heads;
    // a head lane collects the rest of its run from the lanes behind it with shuffles, in run order (PCL's float summation order).  A run that leaves its
    // 64-element block goes on in the registers of the wave's next blocks (their leading lanes, read one by one in order); only a run that leaves the
    // wave's 512 elements goes on through memory.
    unsigned long long cmf[kVxBlocks];          // lanes that continue the run of the element before them (lane 0: the run of the previous block)
#pragma unroll
    for (int u = 0; u < kVxBlocks; u++) { const int i = w_lo + 64 * u + lane; cmf[u] = __ballot(i < w_hi && i > 0 && c[u] == kp[u]); }
    const unsigned int k_next = ka[min(w_hi, n - 1)];       // the key behind the wave's segment
    // the in-block part of every run, the eight blocks side by side (a run is summed in order, one shuffle round per element: alone, a block's
    // rounds are a chain of dependent LDS round trips as long as its longest run)
    float sx[kVxBlocks], sy[kVxBlocks], sz[kVxBlocks], si[kVxBlocks];
    int rlen[kVxBlocks], rl_all = 0;
#pragma unroll
    for (int u = 0; u < kVxBlocks; u++) {
        const bool head = w_lo + 64 * u + lane < w_hi && !((cmf[u] >> lane) & 1ull);
        // run length behind a head inside the block: consecutive continuation bits after its lane
        const unsigned long long after = lane < 63 ? (cmf[u] >> (lane + 1)) : 0ull;
        rlen[u] = head ? (int)__builtin_ctzll(~after | (1ull << 63)) : 0;
        rl_all = max(rl_all, rlen[u]);
        sx[u] = pt[u].x; sy[u] = pt[u].y; sz[u] = pt[u].z; si[u] = pt[u].w;
    }
    rl_all = (int)wave_max_i(rl_all);
    for (int t = 1; t <= rl_all; t++) {
#pragma unroll
        for (int u = 0; u < kVxBlocks; u++) {
            const float vx = __shfl_down(pt[u].x, t), vy = __shfl_down(pt[u].y, t), vz = __shfl_down(pt[u].z, t), vw = __shfl_down(pt[u].w, t);
            if (t <= rlen[u]) { sx[u] += vx; sy[u] += vy; sz[u] += vz; si[u] += vw; }
        }
    }
#pragma unroll
    for (int u = 0; u < kVxBlocks; u++) {
        const int r0 = w_lo + 64 * u;
        if (r0 >= w_hi) break;
        const int i = r0 + lane;
        const bool head = i < w_hi && !((cmf[u] >> lane) & 1ull);
        const unsigned long long hm = __ballot(head);
        const int rl = rlen[u];
        int cnt = 1 + rl;
        bool open = head && lane + rl == 63 && r0 + 64 < n;          // at most one lane: its run reaches the block's end
#pragma unroll
        for (int v = u + 1; v < kVxBlocks; v++) {
            if (__ballot(open) == 0ull) break;
            const int lead = cmf[v] == ~0ull ? 64 : (int)__builtin_ctzll(~cmf[v]);       // leading lanes of block v that continue the run
            for (int t = 0; t < lead; t++) {
                const float vx = __shfl(pt[v].x, t), vy = __shfl(pt[v].y, t), vz = __shfl(pt[v].z, t), vw = __shfl(pt[v].w, t);
                if (open) { sx[u] += vx; sy[u] += vy; sz[u] += vz; si[u] += vw; cnt++; }
            }
            if (lead < 64) open = false;
        }
        // still open behind the wave's last block: the rest comes from memory, four at a time (keys and indices first, then the points)
        if (open && w_lo + kVxTile / 4 < n && k_next == c[u]) {
            for (int u0 = w_lo + kVxTile / 4; u0 < n; u0 += 4) {
                unsigned int k4[4]; int j4[4];
#pragma unroll
                for (int v = 0; v < 4; v++) { const int uc = min(u0 + v, n - 1); k4[v] = ka[uc]; j4[v] = ia[uc]; }
                float4 p4[4];
#pragma unroll
                for (int v = 0; v < 4; v++) p4[v] = gin[min((unsigned int)j4[v], (unsigned int)(n - 1))];
                bool go = true;
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    if (!go || u0 + v >= n || k4[v] != c[u]) { go = false; continue; }
                    sx[u] += p4[v].x; sy[u] += p4[v].y; sz[u] += p4[v].z; si[u] += p4[v].w;
                    cnt++;
                }
                if (!go) break;
            }
        }
        if (head) {
            const float fc = (float)cnt;
            gout[o + __popcll(hm & lt)] = make_float4(sx[u] / fc, sy[u] / fc, sz[u] / fc, si[u] / fc);
        }
        o += __popcll(hm);
    }
}

// The kernels: one workgroup per entry of the tile table -- or, when the table was written by the device (n_tiles_d: its length, the laserMapping frame's map
// update), a fixed number of workgroups that stride over it.
__global__ __launch_bounds__(kVxT) void k_vox_box(const VoxJob *jobs, const int *tile_tab, const int *n_tiles_d)
{
    if (!n_tiles_d) { vox_box_tile(jobs, tile_tab[blockIdx.x]); return; }
    for (int t = blockIdx.x, nt = *n_tiles_d; t < nt; t += gridDim.x) { vox_box_tile(jobs, tile_tab[t]); __syncthreads(); }
}
__global__ __launch_bounds__(kVxT) void k_vox_keys(const VoxJob *jobs, const int *tile_tab, const int *n_tiles_d)
{
    if (!n_tiles_d) { vox_keys_tile(jobs, tile_tab[blockIdx.x]); return; }
    for (int t = blockIdx.x, nt = *n_tiles_d; t < nt; t += gridDim.x) { vox_keys_tile(jobs, tile_tab[t]); __syncthreads(); }
}
__global__ __launch_bounds__(kVxT) void k_vox_count(const VoxJob *jobs, const int *tile_tab, int pass, const int *n_tiles_d)
{
    if (!n_tiles_d) { vox_count_tile(jobs, tile_tab[blockIdx.x], pass); return; }
    for (int t = blockIdx.x, nt = *n_tiles_d; t < nt; t += gridDim.x) { vox_count_tile(jobs, tile_tab[t], pass); __syncthreads(); }
}
__global__ __launch_bounds__(kVxT) void k_vox_pass(const VoxJob *jobs, const int *tile_tab, int pass, const int *n_tiles_d)
{
    if (!n_tiles_d) { vox_pass_tile(jobs, tile_tab[blockIdx.x], pass); return; }
    for (int t = blockIdx.x, nt = *n_tiles_d; t < nt; t += gridDim.x) { vox_pass_tile(jobs, tile_tab[t], pass); __syncthreads(); }
}
__global__ __launch_bounds__(kVxT) void k_vox_heads(const VoxJob *jobs, const int *tile_tab, int max_passes, const int *n_tiles_d)
{
    if (!n_tiles_d) { vox_heads_tile(jobs, tile_tab[blockIdx.x], max_passes); return; }
    for (int t = blockIdx.x, nt = *n_tiles_d; t < nt; t += gridDim.x) { vox_heads_tile(jobs, tile_tab[t], max_passes); __syncthreads(); }
}
__global__ __launch_bounds__(kVxT) void k_vox_centroids(const VoxJob *jobs, const int *tile_tab, int max_passes, const int *n_tiles_d)
{
    if (!n_tiles_d) { vox_centroids_tile(jobs, tile_tab[blockIdx.x], max_passes); return; }
    for (int t = blockIdx.x, nt = *n_tiles_d; t < nt; t += gridDim.x) { vox_centroids_tile(jobs, tile_tab[t], max_passes); __syncthreads(); }
}

// One workgroup per entry of the tile table: (job << 6) | tile, every tile of every job (a job without points keeps one entry: its count must be
// written).  Built on the host next to the job table and uploaded with it.
static inline void vox_tile_table(const VoxJob *jobs, size_t n_jobs, std::vector<int> &tab)
{
    tab.clear();
    for (size_t j = 0; j < n_jobs; j++) {
        const int tiles = jobs[j].n > 0 && jobs[j].n <= kVoxCloudMax ? (jobs[j].n + kVxTile - 1) / kVxTile : 1;
        for (int t = 0; t < tiles; t++) tab.push_back((int)(j << 6) | t);
    }
}

// the launches of a table of voxel jobs; max_passes = the passes to launch (4 covers every 32-bit key; a job that needs more than were launched
// comes back rejected)
static inline void launch_voxel_jobs(hipStream_t st, const VoxJob *jobs_d, const int *tab_d, int n_tiles, int max_passes, const int *n_tiles_d = nullptr)
{
    if (n_tiles <= 0) return;
    const dim3 g((unsigned)n_tiles), b(kVxT);
    hipLaunchKernelGGL(k_vox_box, g, b, 0, st, jobs_d, tab_d, n_tiles_d);
    hipLaunchKernelGGL(k_vox_keys, g, b, 0, st, jobs_d, tab_d, n_tiles_d);
    for (int pass = 0; pass < max_passes; pass++) {
        if (pass > 0) hipLaunchKernelGGL(k_vox_count, g, b, 0, st, jobs_d, tab_d, pass, n_tiles_d);
        hipLaunchKernelGGL(k_vox_pass, g, b, 0, st, jobs_d, tab_d, pass, n_tiles_d);
    }
    hipLaunchKernelGGL(k_vox_heads, g, b, 0, st, jobs_d, tab_d, max_passes, n_tiles_d);
    hipLaunchKernelGGL(k_vox_centroids, g, b, 0, st, jobs_d, tab_d, max_passes, n_tiles_d);
}

} // namespace lmono

// ---- device-resident cube map: data movement kernels ------------------------------------------------------------------
namespace lmono {

struct CopyJob { const float4 *src; float4 *dst; int n; };

// one workgroup per job: dst[0..n) = src[0..n)
__global__ __launch_bounds__(256) void k_copy_jobs(const CopyJob *jobs)
{
    const CopyJob J = jobs[blockIdx.x];
    for (int i = threadIdx.x; i < J.n; i += 256) J.dst[i] = J.src[i];
}

// pointAssociateToMap of every down-sampled scan point with the refined pose (double transform stored in a float point,
// intensity kept) and the cube it falls into: cube = int((v + 25) / 50) + cen, one lower when v + 25 < 0.
// One job per (stream, cloud type); blockIdx.y = job.
// n_all set: the job's size is n_all[job] (read from the device: the voxel filter's counts) and its cube indices start at cube + sum of the sizes of
// the jobs before it (dense, in job order); otherwise n and cube are used as they are
struct AssignJob { const float4 *stack; int n; const int *n_all; int job; const double *x; int cen_w, cen_h, cen_d; float4 *out; int *cube; };

__global__ __launch_bounds__(256) void k_map_assign(const AssignJob *jobs)
{
    const AssignJob J = jobs[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    int n = J.n, base = 0;
    if (J.n_all) {
        n = J.n_all[J.job];
        if (J.n >= 0) for (int j = 0; j < J.job; j++) base += max(J.n_all[j], 0);       // (uniform: scalar loads)  n < 0: `cube` is the job's own array
    }
    if (i >= n) return;
    const float4 p = J.stack[i];
    const double *x = J.x;
    double rx, ry, rz;
    quat_rotate(x, (double)p.x, (double)p.y, (double)p.z, rx, ry, rz);
    const float4 s = make_float4((float)(rx + x[4]), (float)(ry + x[5]), (float)(rz + x[6]), p.w);
    J.out[i] = s;
    int ci = (int)(((double)s.x + 25.0) / 50.0) + J.cen_w, cj = (int)(((double)s.y + 25.0) / 50.0) + J.cen_h, ck = (int)(((double)s.z + 25.0) / 50.0) + J.cen_d;
    if ((double)s.x + 25.0 < 0) ci--;
    if ((double)s.y + 25.0 < 0) cj--;
    if ((double)s.z + 25.0 < 0) ck--;
    J.cube[base + i] = (ci >= 0 && ci < 21 && cj >= 0 && cj < 21 && ck >= 0 && ck < 11) ? ci + 21 * cj + 441 * ck : -1;
}

// dst[pos[i]] = src[i] for pos[i] >= 0; one job per (stream, cloud type); blockIdx.y = job
struct ScatterJob { const float4 *src; const int *pos; int n; float4 *dst; };

__global__ __launch_bounds__(256) void k_scatter_pos(const ScatterJob *jobs)
{
    const ScatterJob J = jobs[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < J.n && J.pos[i] >= 0) J.dst[J.pos[i]] = J.src[i];
}

} // namespace lmono

// ---- device-resident cube bookkeeping (round 5): the (offset, count) table of the 21 x 21 x 11 cubes, the neighbourhood gather and the plan of a frame's
// map update on the device.  With the table on the host (rounds 2-4) a frame waited twice for the device: once for the cube index of every stack point
// (the host planned the update from them), once for the sizes of the re-filtered cubes (the next frame's gather needs them).  Now the host enqueues the
// whole frame, waits ONCE for the refined pose, and the map update runs behind that wait, beside the caller and the next frame's scan filter.
namespace lmono {

constexpr int kMdW = 21, kMdH = 21, kMdD = 11, kMdCubes = kMdW * kMdH * kMdD;
constexpr int kMdValidMax = 75;              // 5 x 5 x 3 neighbourhood
constexpr int kMdTouched = 256;              // cubes one frame's update may touch, per cloud type (the neighbourhood + cubes outside it that receive points)
constexpr int kMdNeighMax = 1 << 20;         // = kMapNeighMax of the host side
constexpr int kMdArena = 6 << 20;            // = kMapArena
constexpr int kMdCatMax = kMdNeighMax + kVoxCloudMax;
constexpr int kMdGJobs = 2 * (kMdNeighMax / 4096 + kMdValidMax);          // chunks of at most 4096 points
constexpr int kMdCopy = 2 * (kMdNeighMax / 4096 + kMdTouched);
constexpr int kMdKeep = 2 * (kVoxCloudMax / 4096 + kMdTouched);
constexpr int kMdTiles = 2 * (kMdCatMax / kVxTile + kMdTouched);
constexpr int kMdErrTouched = 1, kMdErrCat = 2, kMdErrCube = 4, kMdErrWs = 8, kMdErrArena = 16, kMdErrNeigh = 32, kMdErrFilter = 64, kMdErrTables = 128;

struct MapDev {                              // one per mapper, lives as long as it does
    int2 tab[2][kMdCubes];                   // (offset into the live arena half, points) of every cube
    int2 tmp[2][kMdCubes];                   // the table before a shift
    int bump[2];                             // first free point of the live half
    int err;                                 // sticky error bits of the updates (kMdErr*)
    int n_gjobs;
    int last_sum[2];                         // points of the cubes the last update touched, per type   (bump .. last_sum: one read-back)
    CopyJob gjobs[kMdGJobs];                 // the neighbourhood gather
    CloudJob cj[2];                          // its grids (everything but n is constant)
};
struct MapFrame {                            // one per frame, uploaded by the host (two buffers, by frame parity)
    double x[8];                             // in: predicted pose, out: refined pose (the solve's S.x)
    int stats[8];
    unsigned int bar[16];
    int n_stack[2];                          // the scan filter's counts
    int n_map[2];                            // out: sizes of the neighbourhood clouds
    int snap[6];                             // out: the map's bump[2] | err | n_gjobs | last_sum[2] as k_map_plan_gather found them -- behind the previous frame's commit,
                                             // before this frame's update: the read-back of a frame takes THIS copy (x .. snap: one read-back), not the live words the
                                             // same frame's k_map_plan_update / k_map_commit are about to rewrite
    int cen[3], n_valid;
    int valid[kMdValidMax + 1];
};
struct MapUpd {                              // tables of one frame's update, written by k_map_plan_update
    int n_touched[2], sum_in[2];
    int n_copy, n_vox, n_tiles, n_keep, err, pad[3];
    int t_ind[2][kMdTouched], t_nin[2][kMdTouched], t_newoff[2][kMdTouched], t_job[2][kMdTouched];
    int nout[2 * kMdTouched];
    VoxJob vox[2 * kMdTouched];
    int tiles[kMdTiles];
    CopyJob copy[kMdCopy];
    CopyJob keep[kMdKeep];
};
struct MapDevCfg {                           // kernel argument
    MapDev *dev;
    MapFrame *frame;
    MapUpd *upd;
    MapStream *S;
    float4 *arena[2], *neigh[2], *cat[2], *newpts[2];
    unsigned int *vk[2];
    int *vi[2], *vws[2];
    int vws_cap;
    float inv_leaf[2];
    const int *cube_of[2];                   // cube index of every stack point, per cloud type (k_map_assign)
};

// the cube array moves by one cube along an axis (laserMapping's pointer rotation): tab[i] = tmp[i - dir]; the cubes that enter are empty
__global__ __launch_bounds__(256) void k_map_shift(MapDev *dev, int axis, int dir)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * kMdCubes) return;
    const int t = i / kMdCubes, c = i - t * kMdCubes;
    int ijk[3] = { c % kMdW, (c / kMdW) % kMdH, c / (kMdW * kMdH) };
    const int n[3] = { kMdW, kMdH, kMdD };
    ijk[axis] -= dir;
    int2 v = make_int2(0, 0);
    if (ijk[axis] >= 0 && ijk[axis] < n[axis]) v = dev->tmp[t][ijk[0] + kMdW * ijk[1] + kMdW * kMdH * ijk[2]];
    dev->tab[t][c] = v;
}

// the cubes of the neighbourhood, in validInd order, become the two map clouds: copy jobs, sizes for the grids and the optimisation
__global__ __launch_bounds__(192) void k_map_plan_gather(MapDevCfg cfg)
{
    __shared__ int2 s_seg[2][kMdValidMax + 1];
    MapDev *dev = cfg.dev;
    const MapFrame *F = cfg.frame;
    const int tid = threadIdx.x, nv = F->n_valid;
    if (tid < 2 * kMdValidMax) {
        const int t = tid / kMdValidMax, v = tid - t * kMdValidMax;
        s_seg[t][v] = v < nv ? dev->tab[t][F->valid[v]] : make_int2(0, 0);
    }
    __syncthreads();
    __shared__ int s_at[2][kMdValidMax + 1], s_job[2][kMdValidMax + 1], s_tot[2][2];
    if (tid < 128) {         // wave t sums type t: two cubes per lane (75 <= 128), one scan for the points and one for the copy jobs
        const int t = tid >> 6, lane = tid & 63;
        int n[2], jn[2];
#pragma unroll
        for (int u = 0; u < 2; u++) { const int v = 2 * lane + u; n[u] = v < nv ? max(s_seg[t][v].y, 0) : 0; jn[u] = (n[u] + 4095) / 4096; }
        const int in_n = wave_scan_incl(n[0] + n[1]), in_j = wave_scan_incl(jn[0] + jn[1]);
        int at = in_n - n[0] - n[1], jb = in_j - jn[0] - jn[1];
#pragma unroll
        for (int u = 0; u < 2; u++) { const int v = 2 * lane + u; if (v < nv) { s_at[t][v] = at; s_job[t][v] = jb; } at += n[u]; jb += jn[u]; }
        if (lane == 63) { s_tot[t][0] = in_n; s_tot[t][1] = in_j; }
    }
    __syncthreads();
    const bool over = s_tot[0][0] > kMdNeighMax || s_tot[1][0] > kMdNeighMax;       // refused: the frame runs on an empty neighbourhood and reports it
    if (tid >= 128 && tid < 134) {           // (this kernel runs behind the previous commit on the main stream: the words are that update's, whole)
        const int k = tid - 128;
        cfg.frame->snap[k] = k < 2 ? dev->bump[k] : k == 2 ? dev->err : k == 3 ? dev->n_gjobs : dev->last_sum[k - 4];
    }
    if (tid < 2) {
        const int at = over ? 0 : s_tot[tid][0];
        dev->cj[tid].n = at;
        cfg.S->n_map[tid] = at;
        cfg.frame->n_map[tid] = at;
    }
    if (tid == 0) {
        dev->n_gjobs = over ? 0 : s_tot[0][1] + s_tot[1][1];
        if (over) atomicOr(&dev->err, kMdErrNeigh);
    }
    if (over) return;
    if (tid < 2 * kMdValidMax) {
        const int t = tid / kMdValidMax, v = tid - t * kMdValidMax;
        if (v < nv) {
            const int2 sg = s_seg[t][v];
            for (int o = 0, j = s_job[t][v] + (t ? s_tot[0][1] : 0); o < sg.y; o += 4096, j++) {
                CopyJob J; J.src = cfg.arena[t] + sg.x + o; J.dst = cfg.neigh[t] + s_at[t][v] + o; J.n = min(4096, sg.y - o);
                dev->gjobs[j] = J;
            }
        }
    }
}

// jobs[0 .. *n_jobs): dst[0..n) = src[0..n), the workgroups stride over the table
__global__ __launch_bounds__(256) void k_copy_jobs_n(const CopyJob *jobs, const int *n_jobs)
{
    const int nj = *n_jobs;
    for (int j = blockIdx.x; j < nj; j += gridDim.x) {
        const CopyJob J = jobs[j];
        for (int i = threadIdx.x; i < J.n; i += 256) J.dst[i] = J.src[i];
    }
}

// k_map_plan_gather + k_copy_jobs_n + k_grid_clear in ONE launch (round 6: the single-stream frame is a chain of ~35 dependent launches; these three were
// ~17 us of kernels and two launch gaps).  EVERY workgroup forms the neighbourhood's plan for itself (150 table rows, two wave scans: a few microseconds,
// no cross-workgroup dependency); workgroup 0 publishes it (sizes for the grids and the optimisation, the read-back snapshot); then the workgroups stride
// over the copy chunks (<= 4096 points each, as k_map_plan_gather cut them) and over the cells of the two hash tables, which k_grid_insert expects empty.
constexpr int kMgcT = 256, kMgcGrid = 256;
__global__ __launch_bounds__(kMgcT) void k_map_gather_copy_clear(MapDevCfg cfg)
{
    __shared__ int2 s_seg[2][kMdValidMax + 1];
    __shared__ int s_at[2][kMdValidMax + 1], s_job[2][kMdValidMax + 1], s_tot[2][2];
    MapDev *dev = cfg.dev;
    const MapFrame *F = cfg.frame;
    const int tid = threadIdx.x, nv = F->n_valid;
    if (tid < 2 * kMdValidMax) {
        const int t = tid / kMdValidMax, v = tid - t * kMdValidMax;
        s_seg[t][v] = v < nv ? dev->tab[t][F->valid[v]] : make_int2(0, 0);
    }
    __syncthreads();
    if (tid < 128) {         // wave t sums type t: two cubes per lane, one scan for the points and one for the copy chunks
        const int t = tid >> 6, lane = tid & 63;
        int n[2], jn[2];
#pragma unroll
        for (int u = 0; u < 2; u++) { const int v = 2 * lane + u; n[u] = v < nv ? max(s_seg[t][v].y, 0) : 0; jn[u] = (n[u] + 4095) / 4096; }
        const int in_n = wave_scan_incl(n[0] + n[1]), in_j = wave_scan_incl(jn[0] + jn[1]);
        int at = in_n - n[0] - n[1], jb = in_j - jn[0] - jn[1];
#pragma unroll
        for (int u = 0; u < 2; u++) { const int v = 2 * lane + u; if (v < nv) { s_at[t][v] = at; s_job[t][v] = jb; } at += n[u]; jb += jn[u]; }
        if (lane == 63) { s_tot[t][0] = in_n; s_tot[t][1] = in_j; }
    }
    __syncthreads();
    const bool over = s_tot[0][0] > kMdNeighMax || s_tot[1][0] > kMdNeighMax;       // refused: the frame runs on an empty neighbourhood and reports it
    const int n_map[2] = { over ? 0 : s_tot[0][0], over ? 0 : s_tot[1][0] };
    if (blockIdx.x == 0) {
        if (tid >= 128 && tid < 134) {       // (behind the previous commit: the words are that update's, whole)
            const int k = tid - 128;
            cfg.frame->snap[k] = k < 2 ? dev->bump[k] : k == 2 ? dev->err : k == 3 ? dev->n_gjobs : dev->last_sum[k - 4];
        }
        if (tid < 2) { dev->cj[tid].n = n_map[tid]; cfg.S->n_map[tid] = n_map[tid]; cfg.frame->n_map[tid] = n_map[tid]; }
        if (tid == 0) {
            dev->n_gjobs = over ? 0 : s_tot[0][1] + s_tot[1][1];
            if (over) atomicOr(&dev->err, kMdErrNeigh);
        }
    }
    // the hash tables of the two clouds, emptied (k_grid_clear's part; the static fields of the jobs were written by the host when the mapper was made)
    for (int t = 0; t < 2; t++) {
        CloudJob J = dev->cj[t];
        J.n = n_map[t];
        const int T = cloud_grid_size(J);
        if (blockIdx.x == 0 && tid == 0) { *J.mask_out = T ? T - 1 : 0; *J.bump = 0; }
        typedef __attribute__((address_space(1))) GridCell GCell;
        GridCell *cell = (GridCell *)(GCell *)J.cell;
        for (int i = blockIdx.x * kMgcT + tid; i < T; i += gridDim.x * kMgcT) { GridCell e; e.key = kEmptyKey; e.start = 0; e.cnt = 0; cell[i] = e; }
    }
    if (over) return;
    // the copy chunks: chunk j of type t = the k-th 4096 points of cube v, found by bisection over the cubes' first chunks
    const int nj0 = s_tot[0][1], nj = nj0 + s_tot[1][1];
    for (int j = blockIdx.x; j < nj; j += gridDim.x) {
        const int t = j >= nj0 ? 1 : 0, jl = j - (t ? nj0 : 0);
        int lo = 0, hi = nv - 1;                     // last cube whose first chunk is <= jl (cubes without points share theirs with the next)
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_job[t][mid] <= jl) lo = mid; else hi = mid - 1; }
        // (an empty cube behind the one that holds chunk jl has the same first chunk: step back to the cube that has points)
        int v = lo;
        while (v > 0 && (max(s_seg[t][v].y, 0) + 4095) / 4096 <= jl - s_job[t][v]) v--;
        const int2 sg = s_seg[t][v];
        const int o = (jl - s_job[t][v]) * 4096, cnt = min(4096, sg.y - o);
        const float4 *src = cfg.arena[t] + sg.x + o;
        float4 *dst = cfg.neigh[t] + s_at[t][v] + o;
        for (int i = tid; i < cnt; i += kMgcT) dst[i] = src[i];
    }
}

// The plan of a frame's map update, both cloud types in one workgroup (threads 0..511: corner, 512..1023: surf).  Per type: how many stack points
// fall into every cube; the touched cubes in ascending cube order -- the neighbourhood's non-empty cubes and every other cube that receives points --;
// [old points | new points in stack order] of every touched cube laid out in `cat` (the new points are placed here; the old ones by copy jobs);
// fresh arena space behind the bump pointer; a voxel-filter job per touched cube of the neighbourhood, a plain copy for the others.
constexpr int kMuT = 1024, kMuH = 512, kMuW = kMuH / 64;
constexpr int kMuStage = 12288;        // stack points per cloud type whose cube indices are staged in LDS (a 64-ring scan leaves 4-7 k after the filter)
// runs of equal values among the wave's 64 lanes (v < 0: no value): true for the first lane of a run, len = the run's length.  One LDS add per RUN instead
// of per lane: an LDS atomic serialises the lanes that share an address, and 16 waves of them cost the kernel ~10 k cycles per pass.
__device__ __forceinline__ bool run_head(int v, int lane, int &len)
{
    const int prev = __shfl_up(v, 1);
    const bool head = v >= 0 && (lane == 0 || prev != v);
    const unsigned long long hm = __ballot(head), vm = __ballot(v >= 0);
    // the run ends before the next head or the next lane without a value
    const unsigned long long stop = (hm | ~vm) & (lane == 63 ? 0ull : (~0ull << (lane + 1)));
    len = head ? (stop ? (int)__builtin_ctzll(stop) : 64) - lane : 0;
    return head;
}
// value of lane `l` (the same l in every lane): v_readlane through a scalar register -- __shfl goes through the LDS crossbar (~130 cycles per hop of a
// dependent chain; the three loops below were chains of them: 70 k of the kernel's 100 k cycles)
__device__ __forceinline__ int lane_value(int v, int l) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(l)); }
__global__ __launch_bounds__(kMuT) void k_map_plan_update(MapDevCfg cfg)
{
    __shared__ int s_add[2][kMdCubes];
    __shared__ short s_slot[2][kMdCubes];
    __shared__ unsigned char s_isv[kMdCubes];
    __shared__ int s_ind[2][kMdTouched], s_nold[2][kMdTouched], s_nadd[2][kMdTouched], s_cat[2][kMdTouched], s_oldoff[2][kMdTouched];
    __shared__ int s_base[2][kMdTouched][5];        // first copy job, filter job, tile, workspace int, keep job of every touched cube (type-local)
    __shared__ int s_wh[2][kMuW][kMdTouched];
    __shared__ int s_wtot[2][kMuW], s_nt[2], s_tot[2][6], s_err;
    __shared__ unsigned short s_cube[2][kMuStage];      // the stack points' cube indices (0xffff: outside the array), read from memory once
    MapDev *dev = cfg.dev;
    MapUpd *U = cfg.upd;
    const MapFrame *F = cfg.frame;
#ifdef LMONO_MU_PROF
    unsigned long long mu_t[16];
    mu_t[0] = __builtin_readcyclecounter();
#endif
    const int tid = threadIdx.x, t = tid >> 9, lt = tid & (kMuH - 1), wv = lt >> 6, lane = tid & 63;
    const int ns0 = max(F->n_stack[0], 0), ns1 = max(F->n_stack[1], 0);
    const int n_st = t ? ns1 : ns0;
    const int *cube = cfg.cube_of[t];
    const unsigned long long ltm = (1ull << lane) - 1ull;
    // Everything this kernel reads was written by other kernels on other XCDs: every DEPENDENT round of global loads is a ~2 us trip to memory.  So:
    // the table rows of step 2 and the bump pointer are requested here, the stack points' cube indices are read once (step 1) and kept in LDS for
    // steps 5a / 5b, and the points themselves are requested one batch ahead of their use.
    const int c0 = lt * 10;
    int2 seg[10];
#pragma unroll
    for (int u = 0; u < 10; u++) seg[u] = c0 + u < kMdCubes ? dev->tab[t][c0 + u] : make_int2(0, 0);
    const int bump = dev->bump[t];
    const bool staged = n_st <= kMuStage;
    // a wave's piece of the stack: contiguous, so that the pieces' counts become stack-order positions in step 5
    const int piece = ((n_st + kMuW - 1) / kMuW + 63) & ~63;
    const int p_lo = min(wv * piece, n_st), p_hi = min(p_lo + piece, n_st);
    auto cube_at = [&](int i) -> int { if (staged) { const int v = (int)s_cube[t][i]; return v == 0xffff ? -1 : v; } return cube[i]; };
    for (int k = tid; k < 2 * kMdCubes; k += kMuT) { (&s_add[0][0])[k] = 0; (&s_slot[0][0])[k] = (short)-1; }
    for (int k = tid; k < kMdCubes; k += kMuT) s_isv[k] = 0;
    if (tid == 0) s_err = 0;
    __syncthreads();
#ifdef LMONO_MU_PROF
    if (tid == 0) mu_t[1] = __builtin_readcyclecounter();
#endif
    if (tid < F->n_valid) s_isv[F->valid[tid]] = 1;
    // 1. points per cube (lanes with the same cube add once), eight rounds of the wave's piece per trip to memory
    for (int i0 = p_lo; i0 < p_hi; i0 += 8 * 64) {
        int c8[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int i = i0 + 64 * u + lane; c8[u] = i < p_hi ? cube[i] : -1; }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int c = c8[u], i = i0 + 64 * u + lane;
            if (staged && i < p_hi) s_cube[t][i] = (unsigned short)(c < 0 ? 0xffff : c);
            // (64 consecutive stack points -- voxel order: rows along x through the whole cloud -- fall into 5-10 different cubes: a loop over the distinct
            // cubes of a round was ~250 cycles per cube)
            int len;
            if (run_head(c, lane, len)) atomicAdd(&s_add[t][c], len);
        }
    }
    __syncthreads();
#ifdef LMONO_MU_PROF
    if (tid == 0) mu_t[2] = __builtin_readcyclecounter();
#endif
    // 2. the touched cubes, ascending: ten cubes per thread, their flags counted and placed through a scan of the type's 512 threads
    {
        int cnt = 0;
        unsigned int fl = 0;
#pragma unroll
        for (int u = 0; u < 10; u++) {
            const int c = c0 + u;
            const int ad = c < kMdCubes ? s_add[t][c] : 0;
            const bool need = c < kMdCubes && (s_isv[c] ? seg[u].y + ad > 0 : ad > 0);
            if (need) { fl |= 1u << u; cnt++; }
        }
        const int incl = wave_scan_incl(cnt);
        if (lane == 63) s_wtot[t][wv] = incl;
        __syncthreads();
#ifdef LMONO_MU_PROF
    if (tid == 0) mu_t[3] = __builtin_readcyclecounter();
#endif
        int at = incl - cnt;
        for (int w = 0; w < wv; w++) at += s_wtot[t][w];
        if (lt == kMuH - 1) s_nt[t] = at + cnt;
#pragma unroll
        for (int u = 0; u < 10; u++)
            if (fl & (1u << u)) {
                if (at < kMdTouched) { s_ind[t][at] = c0 + u; s_nold[t][at] = seg[u].y; s_nadd[t][at] = s_add[t][c0 + u]; s_oldoff[t][at] = seg[u].x; s_slot[t][c0 + u] = (short)at; }
                at++;
            }
    }
    __syncthreads();
#ifdef LMONO_MU_PROF
    if (tid == 0) mu_t[4] = __builtin_readcyclecounter();
#endif
    if (lt == 0 && s_nt[t] > kMdTouched) atomicOr(&s_err, kMdErrTouched);
    const int nt = min(s_nt[t], kMdTouched);
    // 3. running sums over the touched cubes (wave 0 of the type, four cubes per lane): place in `cat`, copy jobs of the old points, filter jobs, tiles,
    // workspace, plain copies
    if (wv == 0) {
        int nin[4], q[4][6], loc[6] = { 0, 0, 0, 0, 0, 0 };
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int k = 4 * lane + u;
            const bool on = k < nt;
            const int nold = on ? s_nold[t][k] : 0;
            nin[u] = on ? nold + s_nadd[t][k] : 0;
            const bool filt = on && s_isv[s_ind[t][k]] != 0;
            if (filt && nin[u] > kVoxCloudMax) atomicOr(&s_err, kMdErrCube);
            q[u][0] = nin[u];
            q[u][1] = (nold + 4095) / 4096;
            q[u][2] = filt ? 1 : 0;
            q[u][3] = filt ? max(1, (nin[u] + kVxTile - 1) / kVxTile) : 0;
            q[u][4] = filt ? kVxHdr + max(1, (nin[u] + kVxTile - 1) / kVxTile) * kVxWsTile : 0;
            q[u][5] = (on && !filt) ? (nin[u] + 4095) / 4096 : 0;
#pragma unroll
            for (int z = 0; z < 6; z++) loc[z] += q[u][z];
        }
        int run[6];
#pragma unroll
        for (int z = 0; z < 6; z++) { const int incl = wave_scan_incl(loc[z]); run[z] = incl - loc[z]; if (lane == 63) s_tot[t][z] = incl; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int k = 4 * lane + u;
            if (k < nt) {
                s_base[t][k][0] = run[1]; s_base[t][k][1] = q[u][2] ? run[2] : -1; s_base[t][k][2] = run[3]; s_base[t][k][3] = run[4]; s_base[t][k][4] = run[5];
                s_cat[t][k] = run[0];
            }
#pragma unroll
            for (int z = 0; z < 6; z++) run[z] += q[u][z];
        }
    }
    __syncthreads();
#ifdef LMONO_MU_PROF
    if (tid == 0) mu_t[5] = __builtin_readcyclecounter();
#endif
    if (lt == 0) {
        int e = 0;
        if (s_tot[t][0] > kMdCatMax) e |= kMdErrCat;
        if (s_tot[t][4] > cfg.vws_cap) e |= kMdErrWs;
        if (bump + s_tot[t][0] > kMdArena) e |= kMdErrArena;
        if (e) atomicOr(&s_err, e);
    }
    if (tid == 0) {
        int e = 0;
        if (s_tot[0][1] + s_tot[1][1] > kMdCopy || s_tot[0][5] + s_tot[1][5] > kMdKeep || s_tot[0][3] + s_tot[1][3] > kMdTiles) e |= kMdErrTables;
        if (e) atomicOr(&s_err, e);
    }
    __syncthreads();
#ifdef LMONO_MU_PROF
    if (tid == 0) mu_t[6] = __builtin_readcyclecounter();
#endif
    const int err = s_err;
    if (err) {           // nothing of this update runs; the table keeps the map as it was
        if (tid == 0) { U->n_touched[0] = U->n_touched[1] = 0; U->sum_in[0] = U->sum_in[1] = 0; U->n_copy = U->n_vox = U->n_tiles = U->n_keep = 0; U->err = err; atomicOr(&dev->err, err); }
        return;
    }
    // 5. the new points behind their cube's old ones, in stack order: every wave counts a contiguous piece of the stack per touched cube, the pieces' counts
    // become first positions (piece by piece: stack order), every wave places its piece
    typedef __attribute__((address_space(1))) const float4 GF4;
    const float4 *src = (const float4 *)(GF4 *)cfg.newpts[t];
    for (int k = lane; k < kMdTouched; k += 64) s_wh[t][wv][k] = 0;          // (the wave's own row: its LDS operations execute in order)
    for (int i0 = p_lo; i0 < p_hi; i0 += 8 * 64) {
        int c8[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int i = i0 + 64 * u + lane; c8[u] = i < p_hi ? cube_at(i) : -1; }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int sl = c8[u] >= 0 ? (int)s_slot[t][c8[u]] : -1;
            int len;
            if (run_head(sl, lane, len)) atomicAdd(&s_wh[t][wv][sl], len);
        }
    }
    __syncthreads();
#ifdef LMONO_MU_PROF
    if (tid == 0) mu_t[7] = __builtin_readcyclecounter();
#endif
    for (int k = lt; k < nt; k += kMuH) {
        int run = s_cat[t][k] + s_nold[t][k];
        for (int w = 0; w < kMuW; w++) { const int c = s_wh[t][w][k]; s_wh[t][w][k] = run; run += c; }
    }
    __syncthreads();
#ifdef LMONO_MU_PROF
    if (tid == 0) mu_t[8] = __builtin_readcyclecounter();
#endif
    {
        // 5b. positions of eight rounds (the peers of a lane's cube from ballots over the bits of its index among the touched cubes, the cube's cursor moved by the peers' lowest
        // lane) while their points, requested together, are on their way; then the stores, together: a load / store pair per round made the compiler wait for every outstanding
        // memory operation -- the previous round's store included -- before each store (~4 k cycles per round)
        float4 *cat = cfg.cat[t];
        int nbits = 1;
        while ((1 << nbits) < nt) nbits++;          // (the pass is bound by its vector instructions: 16 waves on one compute unit)
        for (int i0 = p_lo; i0 < p_hi; i0 += 8 * 64) {
            float4 p8[8];
#pragma unroll
            for (int u = 0; u < 8; u++) p8[u] = src[max(min(i0 + 64 * u + lane, n_st - 1), 0)];
            int pos[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i = i0 + 64 * u + lane;
                const int c = i < p_hi ? cube_at(i) : -1;
                const int sl = c >= 0 ? (int)s_slot[t][c] : -1;
                const bool ok = sl >= 0;
                unsigned long long pm = __ballot(ok);
#pragma unroll
                for (int bq = 0; bq < 5; bq++) {
                    const unsigned long long mb = __ballot(ok && ((sl >> bq) & 1));
                    pm &= ((sl >> bq) & 1) ? mb : ~mb;
                }
                if (nbits > 5) {          // more than 32 touched cubes (uniform)
#pragma unroll
                    for (int bq = 5; bq < 8; bq++) {
                        const unsigned long long mb = __ballot(ok && ((sl >> bq) & 1));
                        pm &= ((sl >> bq) & 1) ? mb : ~mb;
                    }
                }
                pm = ok ? pm : 0ull;
                const int leader = ok ? (int)__ffsll((long long)pm) - 1 : lane;
                int cur = 0;
                if (ok && lane == leader) { cur = s_wh[t][wv][sl]; s_wh[t][wv][sl] = cur + (int)__popcll(pm); }
                cur = __shfl(cur, leader);
                pos[u] = ok ? cur + (int)__popcll(pm & ltm) : -1;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) if (pos[u] >= 0) cat[pos[u]] = p8[u];
        }
    }
#ifdef LMONO_MU_PROF
    if (tid == 0) mu_t[10] = __builtin_readcyclecounter();
#endif
    // 4. the tables (last: their stores' acknowledgements are waited for by nobody but the end of the kernel)
    {
        const int b_copy = t ? s_tot[0][1] : 0, b_vox = t ? s_tot[0][2] : 0, b_tile = t ? s_tot[0][3] : 0, b_keep = t ? s_tot[0][5] : 0;
        for (int k = lt; k < nt; k += kMuH) {
            const int ind = s_ind[t][k], nold = s_nold[t][k], nin = nold + s_nadd[t][k], coff = s_cat[t][k];
            const int old_off = s_oldoff[t][k];
            const int job = s_base[t][k][1] >= 0 ? b_vox + s_base[t][k][1] : -1;
            U->t_ind[t][k] = ind; U->t_nin[t][k] = nin; U->t_newoff[t][k] = bump + coff; U->t_job[t][k] = job;
            for (int o = 0, j = b_copy + s_base[t][k][0]; o < nold; o += 4096, j++) {
                CopyJob J; J.src = cfg.arena[t] + old_off + o; J.dst = cfg.cat[t] + coff + o; J.n = min(4096, nold - o);
                U->copy[j] = J;
            }
            float4 *dst = cfg.arena[t] + bump + coff;
            if (job >= 0) {
                VoxJob J;
                J.in = cfg.cat[t] + coff; J.n = nin; J.inv_leaf = cfg.inv_leaf[t]; J.out = dst; J.n_out = U->nout + job;
                J.key_a = cfg.vk[t] + 2 * coff; J.key_b = J.key_a + nin; J.idx_a = cfg.vi[t] + 2 * coff; J.idx_b = J.idx_a + nin;
                J.ws = cfg.vws[t] + s_base[t][k][3];
                U->vox[job] = J;
                const int tiles = max(1, (nin + kVxTile - 1) / kVxTile);
                for (int q = 0; q < tiles; q++) U->tiles[b_tile + s_base[t][k][2] + q] = (job << 6) | q;
            } else {
                for (int o = 0, j = b_keep + s_base[t][k][4]; o < nin; o += 4096, j++) {
                    CopyJob J; J.src = cfg.cat[t] + coff + o; J.dst = dst + o; J.n = min(4096, nin - o);
                    U->keep[j] = J;
                }
            }
        }
        if (lt == 0) { U->n_touched[t] = nt; U->sum_in[t] = s_tot[t][0]; }
        if (tid == 0) { U->n_copy = s_tot[0][1] + s_tot[1][1]; U->n_vox = s_tot[0][2] + s_tot[1][2]; U->n_tiles = s_tot[0][3] + s_tot[1][3]; U->n_keep = s_tot[0][5] + s_tot[1][5]; U->err = 0; }
    }
#ifdef LMONO_MU_PROF
    if (tid == 0) { mu_t[9] = __builtin_readcyclecounter(); printf("MUPROF n_st %d %d touched %d %d:", ns0, ns1, s_nt[0], s_nt[1]); for (int z = 1; z <= 9; z++) printf(" %llu", mu_t[z] - mu_t[z - 1]); printf(" | pass B %llu, tables %llu, rounds %d", mu_t[10] - mu_t[8], mu_t[9] - mu_t[10], (p_hi - p_lo + 63) / 64); printf("\n"); }
#endif
}

// the touched cubes take their new place and size (a filtered cube: the filter's count); the bump pointers move on
__global__ __launch_bounds__(kMuH) void k_map_commit(MapDevCfg cfg)
{
    MapDev *dev = cfg.dev;
    const MapUpd *U = cfg.upd;
    if (blockIdx.x > 0) {        // workgroups 1.. : the plain copies [old | new] -> arena of the touched cubes outside the neighbourhood (normally none)
        const int nj = U->n_keep;
        for (int j = blockIdx.x - 1; j < nj; j += gridDim.x - 1) {
            const CopyJob J = U->keep[j];
            for (int i = threadIdx.x; i < J.n; i += kMuH) J.dst[i] = J.src[i];
        }
        return;
    }
    __shared__ int s_err;
    if (threadIdx.x == 0) s_err = 0;
    __syncthreads();
    if (U->err == 0) {
        for (int t = 0; t < 2; t++)
            for (int k = threadIdx.x; k < U->n_touched[t]; k += kMuH) {
                const int job = U->t_job[t][k];
                const int n = job >= 0 ? U->nout[job] : U->t_nin[t][k];
                if (n < 0) atomicOr(&s_err, kMdErrFilter);           // the cube keeps its old points
                else dev->tab[t][U->t_ind[t][k]] = make_int2(U->t_newoff[t][k], n);
            }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (U->err == 0) { dev->bump[0] += U->sum_in[0]; dev->bump[1] += U->sum_in[1]; }
        if (s_err) atomicOr(&dev->err, s_err);
        dev->last_sum[0] = U->sum_in[0]; dev->last_sum[1] = U->sum_in[1];
    }
}

} // namespace lmono
