// lmono_amd/csrc/corr_thread.hip -- laserOdometry correspondence search, ONE THREAD PER FEATURE POINT.
//
// Same results as k_correspond (odometry.hip): the exact nearest point (float distance, lowest index on ties) of every de-skewed
// feature point in the previous scan's less-sharp / less-flat cloud, then the reference's scan-line walk.
//
// k_correspond gives 32 lanes to a feature point and pays a full wave instruction for every step of the (mostly scalar) control
// flow around a few dozen candidate points: ~440 vector instructions per feature (profiles/r2).  Here a feature point is ONE lane.
// That is possible because the candidates of a ball come as a handful of short contiguous runs of the (scan line, azimuth bin)-
// sorted copy of the cloud (k_line_index):
//   * a point p with |p - q| <= r lies within asin(r / rho_xy(q)) of q's azimuth and within asin(r / |q|) of q's elevation angle;
//   * lb_elev holds the elevation range of every line plus its monotone envelopes, so the lines a ball can meet are a short
//     interval [v1, v2] found by two binary searches in LDS, each line tested against its own range;
//   * on a line the arc is one run of the sorted copy (two when the arc wraps), read through the bucket table.
// A search with radius r is exact whenever its minimum is <= r; otherwise r grows to the distance found (or geometrically up to the
// 5 m of DISTANCE_SQ_THRESHOLD).  Lines are taken four at a time: their eight table entries are requested together, then one
// point of each run per step -- four independent gathers in flight per lane.
//
// Features whose ball would sweep thousands of points (no neighbour within metres at close range) go to a device work list served
// by k_correspond_list; the list is a few hundred entries per launch.
#include "batch.hpp"

namespace lmono {

constexpr int kCtT = 256;                     // threads per workgroup = feature points per workgroup
constexpr int kCtBlocks = kMaxQueries / kCtT; // workgroups per chain
static_assert(kMaxQueries % kCtT == 0, "feature capacity must be a multiple of the workgroup size");
constexpr float kCtR0 = 0.3f;                 // first search radius of an unseeded feature (m)
constexpr int kCtHeavy = 1536;                // (lines x bins) of one pass above which the feature is deferred to the list kernel

struct CtLds {
    float4 elev[2][66];                       // lb_elev of the two "last" clouds
    int fge[2][66], lle[2][66];
};

// work list of deferred feature points: [0] = count, then (chain << 12 | feature index)
__device__ __forceinline__ void ct_defer(unsigned int *wl, int c, int qi)
{
    const unsigned int slot = atomicAdd(wl, 1u);
    wl[1 + slot] = ((unsigned int)c << 12) | (unsigned int)qi;
}

// azimuth arc of radius r around the feature as one or two bin ranges of a table row: [a0, a1) and [0, w1) (w1 = 0: no wrap)
struct CtArc { int a0, a1, w1, nb; };
__device__ __forceinline__ bool ct_arc(float r, float rho, float th, CtArc &a)
{
    constexpr float kb = kAzBins / 6.28318531f;
    a.a0 = 0; a.a1 = kAzBins; a.w1 = 0; a.nb = kAzBins;
    if (!(rho > r * 1.002f)) return true;                 // the ball reaches the sensor axis: every azimuth
    const float alpha = asin_upper(r / rho) + kArcSlackBins / kb;
    const int lo = (int)floorf((th - alpha) * kb), hi = (int)floorf((th + alpha) * kb);
    const int n = hi - lo + 1;
    if (n >= kAzBins) return true;
    const int b0 = ((lo % kAzBins) + kAzBins) % kAzBins;
    a.nb = n; a.a0 = b0;
    if (b0 + n <= kAzBins) { a.a1 = b0 + n; a.w1 = 0; }
    else { a.a1 = kAzBins; a.w1 = b0 + n - kAzBins; }
    return true;
}

// first line whose envelope A (min of lo over lines <= v, non-increasing) is <= ehi; 66 when none
__device__ __forceinline__ int ct_first_line(const float4 *el, float ehi)
{
    int lo = 0, hi = 66;                                  // invariant: A[lo-1] > ehi, answer in [lo, hi]
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (el[mid].z <= ehi) hi = mid; else lo = mid + 1; }
    return lo;
}
// last line whose envelope B (max of hi over lines >= v, non-increasing) is >= elo; -1 when none
__device__ __forceinline__ int ct_last_line(const float4 *el, float elo)
{
    int lo = -1, hi = 65;                                 // answer in [lo, hi]
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (el[mid].w >= elo) lo = mid; else hi = mid - 1; }
    return lo;
}

template <bool kEdge>
__device__ __forceinline__ bool thread_search(const CtLds &L, const int *tg, const float4 *pts, float qx, float qy, float qz, int n_last, float seed_d,
                                              int4 &out, int &closest_out, float4 &A, float4 &B, float4 &C)
{
    constexpr int cl = kEdge ? 0 : 1;
    out = make_int4(-1, -1, -1, 0);
    closest_out = -1;
    A = B = C = make_float4(0.f, 0.f, 0.f, 0.f);
    const float rho2 = qx * qx + qy * qy, rho = sqrtf(rho2), R = sqrtf(rho2 + qz * qz);
    const float th = atan2f(qy, qx) + 3.14159265f;
    const float eq = elev_of(qx, qy, qz);
    const float4 *el = L.elev[cl];

    // ---- exact nearest point
    NnBest best = kNnNone;
    int apos = -1;
    float r = seed_d >= 0.f ? sqrtf(seed_d) * 1.0005f + 1e-3f : kCtR0;
    for (int pass = 0; pass < 12; pass++) {
        const float rr = fminf(r, 5.0f);                    // d2 < 25 means d < 5: a 5 m ball holds every admissible point
        CtArc a;
        ct_arc(rr, rho, th, a);
        const float beta = R > rr ? asin_upper(rr / R) + 5e-4f : 4.0f;
        const float elo = eq - beta, ehi = eq + beta;
        const int v1 = ct_first_line(el, ehi), v2 = ct_last_line(el, elo);
        if (v2 >= v1 && (v2 - v1 + 1) * a.nb > kCtHeavy) return false;
        for (int seg = 0; seg < 2; seg++) {
            if (seg == 1 && a.w1 == 0) break;
            const int c0 = seg ? 0 : a.a0, c1 = seg ? a.w1 : a.a1;
            for (int vb = v1; vb <= v2; vb += 4) {
                int s4[4], e4[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int v = vb + j;
                    s4[j] = 0; e4[j] = 0;
                    if (v <= v2) {
                        const float4 ev = el[v];
                        if (!(ev.y < elo || ev.x > ehi)) { s4[j] = tg[v * kAzBins + c0]; e4[j] = tg[v * kAzBins + c1]; }
                    }
                }
                int m = max(max(e4[0] - s4[0], e4[1] - s4[1]), max(e4[2] - s4[2], e4[3] - s4[3]));
                for (int i = 0; i < m; i++) {
                    float4 p4[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) { p4[j] = make_float4(0.f, 0.f, 0.f, 0.f); if (s4[j] + i < e4[j]) p4[j] = pts[s4[j] + i]; }
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (s4[j] + i < e4[j]) {
                            const float d = dist2f(p4[j].x, p4[j].y, p4[j].z, qx, qy, qz);
                            const NnBest key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned int)((__float_as_int(p4[j].w) << 7) | (vb + j));
                            if (key < best) { best = key; apos = s4[j] + i; }
                        }
                }
            }
        }
        if (best != kNnNone) {
            const float bd = __uint_as_float((unsigned int)(best >> 32));
            if (bd <= (rr * 0.9999f) * (rr * 0.9999f) || rr >= 5.0f) break;
            r = sqrtf(bd) * 1.0005f + 1e-3f;
        } else {
            if (rr >= 5.0f) break;
            r = rr * 2.5f;
        }
    }
    if (best == kNnNone || !((double)__uint_as_float((unsigned int)(best >> 32)) < 25.0)) return true;     // no correspondence
    const int closest = (int)((unsigned int)(best & 0xffffffffull) >> 7);
    const int ra = (int)(best & 127ull);
    closest_out = closest;

    // ---- scan-line walk on lines ra-2 .. ra+2 inside the index window the reference's loops can reach
    const int w_lo = ra - 3 >= 0 ? L.lle[cl][ra - 3] + 1 : 0;
    const int w_hi = ra + 3 <= 65 ? L.fge[cl][ra + 3] : n_last;
    const unsigned long long thr = pack_fu(25.0f, 0u);
    WalkBest same = thr, other = thr;
    int spos = -1, opos = -1;
    // radii: the neighbouring lines right next to the nearest point; the ring gap of far ground points (rho^2 dtheta / h); 5 m
    const float rad[4] = { walk_radius(0, rho), walk_radius(1, rho), walk_radius(2, rho), walk_radius(3, rho) };
#pragma unroll
    for (int pass = 0; pass < 4; pass++) {
        if (pass > 0 && rad[pass] <= rad[pass - 1]) continue;
        CtArc a;
        ct_arc(rad[pass], rho, th, a);
        if (5 * a.nb > kCtHeavy) return false;
        same = thr; other = thr; spos = -1; opos = -1;
        for (int seg = 0; seg < 2; seg++) {
            if (seg == 1 && a.w1 == 0) break;
            const int c0 = seg ? 0 : a.a0, c1 = seg ? a.w1 : a.a1;
            int s5[5], e5[5];
#pragma unroll
            for (int j = 0; j < 5; j++) {
                const int v = ra - 2 + j;
                s5[j] = 0; e5[j] = 0;
                if (v >= 0 && v <= 65 && !(kEdge && j == 2)) { s5[j] = tg[v * kAzBins + c0]; e5[j] = tg[v * kAzBins + c1]; }   // edges never use the nearest point's own line
            }
            int m = max(max(e5[0] - s5[0], e5[1] - s5[1]), max(max(e5[2] - s5[2], e5[3] - s5[3]), e5[4] - s5[4]));
            for (int i = 0; i < m; i++) {
                float4 p5[5];
#pragma unroll
                for (int j = 0; j < 5; j++) { p5[j] = make_float4(0.f, 0.f, 0.f, 0.f); if (s5[j] + i < e5[j]) p5[j] = pts[s5[j] + i]; }
#pragma unroll
                for (int j = 0; j < 5; j++) {
                    if (!(s5[j] + i < e5[j])) continue;
                    const int jj = __float_as_int(p5[j].w);
                    if (jj == closest || jj < w_lo || jj >= w_hi) continue;
                    const bool fwd = jj > closest;
                    const unsigned int seq = fwd ? (unsigned int)(jj - closest - 1) : kSeqBack + (unsigned int)(closest - 1 - jj);
                    const float d = dist2f(p5[j].x, p5[j].y, p5[j].z, qx, qy, qz);
                    const WalkBest key = ((unsigned long long)__float_as_uint(d) << 32) | seq;
                    const bool is_other = fwd ? (j > 2) : (j < 2);
                    if (is_other) { if (key < other) { other = key; opos = s5[j] + i; } }
                    else if (!kEdge) { if (key < same) { same = key; spos = s5[j] + i; } }
                }
            }
        }
        if (rad[pass] >= 5.0f) break;
        const unsigned long long lim = pack_fu(rad[pass] * rad[pass] * 0.998f, 0u);     // strictly inside the ball of this pass
        if (other < lim && (kEdge || same < lim)) break;
    }
    const int i_other = other < thr ? seq_to_index((unsigned int)(other & 0xffffffffull), closest) : -1;
    if (kEdge) {
        if (i_other >= 0) { out = make_int4(closest, i_other, -1, 1); A = pts[apos]; B = pts[opos]; }
        return true;
    }
    const int i_same = same < thr ? seq_to_index((unsigned int)(same & 0xffffffffull), closest) : -1;
    if (i_same >= 0 && i_other >= 0) { out = make_int4(closest, i_same, i_other, 2); A = pts[apos]; B = pts[spos]; C = pts[opos]; }
    return true;
}

// step `step`, outer iteration `outer` of every chain: one thread per feature point, kCtBlocks workgroups per chain.  The blocks
// of a chain are decoded onto ONE XCD (blocks b and b + 8 share an XCD): the chain's index and tables are fetched into one L2 only.
__global__ __launch_bounds__(kCtT) void k_corr_thread(BatchView b, OdomView o, int step, int outer, unsigned int *wl)
{
    __shared__ CtLds L;
    const int xcd = blockIdx.x & 7, u = blockIdx.x >> 3;
    const int c = o.chain0 + (u / kCtBlocks) * 8 + xcd;
    const int qb = u % kCtBlocks;
    if (c >= o.chain1) return;
    int own;
    const int k = chain_scan(o, c, step, own);
    if (k < 0) return;
    const int tid = threadIdx.x;
    const int l = k - 1;
    const int n_sharp = b.feat_n[k * 4 + 0];
    const int nq = n_sharp + b.feat_n[k * 4 + 2];
    if (qb * kCtT >= nq) return;
    const int qi = qb * kCtT + tid;
    if (b.status[l] & (kStatusIrregularLines | kStatusDenseCell)) {
        if (qi < nq) ct_defer(wl, c, qi);       // rare: the whole scan pair goes to the generic search
        return;
    }
    // the feature point and its seed are requested before the small tables are staged
    const bool edge = qi < n_sharp;
    float4 fp = make_float4(0.f, 0.f, 0.f, 0.f);
    int *seed_c = o.seed ? o.seed + (size_t)c * kMaxQueries : nullptr;
    int sidx = -1;
    if (qi < nq) {
        fp = edge ? b.sharp[(size_t)k * kMaxSharp + qi] : b.flat[(size_t)k * kMaxFlat + (qi - n_sharp)];
        if (outer == 1 && seed_c) sidx = seed_c[qi];
    }
    if (tid < 2 * 66) {
        const int cl = tid / 66, v = tid % 66;
        L.elev[cl][v] = b.lb_elev[(size_t)(l * 2 + cl) * 66 + v];
        L.fge[cl][v] = b.line_first_ge[(size_t)(l * 2 + cl) * 66 + v];
        L.lle[cl][v] = b.line_last_le[(size_t)(l * 2 + cl) * 66 + v];
    }
    const int n_last = b.feat_n[l * 4 + (edge ? 1 : 3)];
    const float4 *cloud = edge ? b.less_sharp + (size_t)l * kMaxLessSharp : b.less_flat + b.off[l];
    float4 sp = make_float4(0.f, 0.f, 0.f, 0.f);
    if (sidx >= 0 && sidx < n_last) sp = cloud[sidx]; else sidx = -1;
    __syncthreads();
    if (qi >= nq) return;
    // de-skew transform in fp64 as the reference's TransformToStart
    const double *x = o.state + c * 8;
    double rx, ry, rz;
    quat_rotate(x, (double)fp.x, (double)fp.y, (double)fp.z, rx, ry, rz);
    const float qx = (float)(rx + x[4]), qy = (float)(ry + x[5]), qz = (float)(rz + x[6]);
    float sd = -1.0f;
    if (sidx >= 0) { const float d = dist2f(sp.x, sp.y, sp.z, qx, qy, qz); if (d < 24.0f) sd = d; }
    const int *tg = b.lb_start + (size_t)(l * 2 + (edge ? 0 : 1)) * (kLineKeys + 1);
    const float4 *pts = edge ? b.lbc_pts + (size_t)l * kMaxLessSharp : b.lbs_pts + b.off[l];
    int4 r;
    int closest;
    float4 A, B, C;
    bool done = true;
    if (n_last == 0) { r = make_int4(-1, -1, -1, 0); closest = -1; A = B = C = make_float4(0.f, 0.f, 0.f, 0.f); }
    else done = edge ? thread_search<true>(L, tg, pts, qx, qy, qz, n_last, sd, r, closest, A, B, C)
                     : thread_search<false>(L, tg, pts, qx, qy, qz, n_last, sd, r, closest, A, B, C);
    if (!done) { ct_defer(wl, c, qi); return; }
    ((int4 *)o.corr + (size_t)c * kMaxQueries)[qi] = r;
    if (outer == 0 && seed_c) seed_c[qi] = closest;
    // residual-block record for the solver: the feature point and its 2 (edge) or 3 (plane) partners, 64 B
    fp.w = __int_as_float(r.w);
    float4 *rec = o.crec + ((size_t)c * kMaxQueries + qi) * 4;
    rec[0] = fp; rec[1] = A; rec[2] = B; rec[3] = C;
}

} // namespace lmono
