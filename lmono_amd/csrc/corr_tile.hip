// lmono_amd/csrc/corr_tile.hip -- laserOdometry correspondence search from LDS-staged point neighbourhoods.
//
// Same results as k_correspond (odometry.hip): the exact nearest point (float distance, lowest index on ties) of every de-skewed
// feature point in the previous scan's less-sharp / less-flat cloud, then the reference's scan-line walk.  What
// changes is where the candidates come from.  One workgroup owns ONE AZIMUTH SECTOR of one chain's current scan pair: it copies the
// (scan line, azimuth bin)-sorted "last" clouds of that sector plus a halo into LDS once (kTW of kAzBins bins, both clouds, with a
// local bucket table), collects the chain's feature points whose transformed position falls into the sector, and serves all of
// them from LDS, four lanes per feature point.
//
// Exactness.  A point p with |p - q| <= r lies within asin(r / rho_xy(q)) of q's azimuth and within asin(r / |q|) of q's
// elevation angle, so the candidates of a ball are: the lines whose elevation range [lb_elev] meets the elevation window, and on
// each of them the bucket run of the azimuth arc (asin bounded from above, +1.5 bins of slack as in k_correspond).  A search with
// radius r is exact whenever its minimum is <= r; otherwise the radius grows to the distance found (or geometrically, up to the
// 5 m of DISTANCE_SQ_THRESHOLD).  A feature whose arc leaves the loaded window (near-range points with wide arcs, far ground
// points whose scan-line partners are metres away) is NOT answered here: it goes to a device work list that k_correspond_list
// serves with the global-memory search.  The tile path therefore never approximates; it answers exactly or defers.
#include "batch.hpp"

namespace lmono {

constexpr int kTSect = 12;                    // azimuth sectors of a scan pair
constexpr int kTB = kAzBins / kTSect;         // bins per sector (32 = 30 deg)
constexpr int kTHalo = 12;                    // halo bins on either side (11.25 deg)
constexpr int kTW = kTB + 2 * kTHalo;         // loaded bins
static_assert(kAzBins % kTSect == 0, "sectors must tile the azimuth bins");
constexpr int kTCapC = 1280;                  // LDS capacity: corner points of a sector window
constexpr int kTCapS = 6144;                  //               surf points
constexpr int kTQCap = 448;                   // feature points of one sector served from LDS (the rest is deferred)
constexpr int kTT = 1024;                     // threads per workgroup
constexpr int kTG = 4;                        // lanes per feature point (a DPP quad)
constexpr float kTR0 = 0.3f;                  // first search radius of an unseeded feature (m)

struct TileLds {
    float4 pts[kTCapC + kTCapS];              // corner window, then surf window: x y z (original index bits)
    float4 qbuf[kTQCap];                      // transformed feature point, .w = feature index
    unsigned short tab[2][66][kTW + 2];       // local start of every (line, local bin) bucket; [kTW] = end of the line
    float2 elev[2][66];                       // elevation range of every line of the "last" clouds
    int lineN[2][66], lineG[2][66], lineF0[2][66], lineBase[2][67];
    int fge[2][66], lle[2][66];
    int nq, over[2];
};
constexpr int kTileLds = (int)sizeof(TileLds);
static_assert(sizeof(TileLds) <= 160 * 1024, "tile does not fit LDS");

// minimum over the four lanes of a quad, returned to all of them (quad_perm [1,0,3,2] and [2,3,0,1])
__device__ __forceinline__ unsigned long long quad_min_u64(unsigned long long v)
{
    unsigned long long o = lane_xor_u64<1>(v);
    v = o < v ? o : v;
    o = lane_xor_u64<2>(v);
    return o < v ? o : v;
}
__device__ __forceinline__ int quad_max_i(int v)
{
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));
    return max(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));
}

// work list of deferred feature points: [0] = count, then (chain << 12 | feature index)
__device__ __forceinline__ void defer_query(unsigned int *wl, int c, int qi)
{
    const unsigned int slot = atomicAdd(wl, 1u);
    wl[1 + slot] = ((unsigned int)c << 12) | (unsigned int)qi;
}

// one feature point, four lanes (g = lane inside the quad).  Returns false when the point must be deferred.
template <bool kEdge>
__device__ __forceinline__ bool tile_search(const TileLds &L, int t, int g, float qx, float qy, float qz, int n_last, bool seeded, float seed_d,
                                            int4 &out, int &closest_out, float4 &A, float4 &B, float4 &C)
{
    constexpr int cl = kEdge ? 0 : 1;
    const float4 *P = L.pts + (kEdge ? 0 : kTCapC);
    out = make_int4(-1, -1, -1, 0);
    closest_out = -1;
    A = B = C = make_float4(0.f, 0.f, 0.f, 0.f);
    const float rho2 = qx * qx + qy * qy, rho = sqrtf(rho2), R = sqrtf(rho2 + qz * qz);
    const float th = atan2f(qy, qx) + 3.14159265f;
    const float eq = elev_of(qx, qy, qz);
    constexpr float kb = kAzBins / 6.28318531f;
    const int wlo = t * kTB - kTHalo, whi = wlo + kTW;      // loaded bins, unwrapped: [wlo, whi)
    // the feature's own bin lies in [t kTB, (t+1) kTB), so arcs computed from th come out unwrapped the way the window is
    // (negative below bin 0 in sector 0, >= kAzBins past the last bin in the last sector)
    const float thu = th;
    auto arc = [&](float r, int &lb_lo, int &lb_hi) -> bool {
        if (!(rho > r * 1.002f)) return false;
        const float alpha = asin_upper(r / rho) + 1.5f / kb;
        const int lo = (int)floorf((thu - alpha) * kb), hi = (int)floorf((thu + alpha) * kb);
        if (lo < wlo || hi >= whi) return false;
        lb_lo = lo - wlo; lb_hi = hi - wlo;
        return true;
    };

    // ---- exact nearest point
    NnBest nb = kNnNone;
    int npos = -1;
    unsigned long long best = kNnNone;
    float r = seeded ? sqrtf(seed_d) * 1.0005f + 1e-3f : kTR0;
    for (int pass = 0; pass < 12; pass++) {
        const float rr = fminf(r, 5.0f);                    // d2 < 25 means d < 5: a 5 m ball holds every admissible point
        int b0, b1;
        if (!arc(rr, b0, b1)) return false;
        const float beta = asin_upper(fminf(rr / R, 1.0f)) + 5e-4f;
        const float elo = eq - beta, ehi = eq + beta;
        for (int v = g; v < 66; v += kTG) {
            const float2 ev = L.elev[cl][v];
            if (ev.y < elo || ev.x > ehi) continue;
            int i = L.tab[cl][v][b0];
            const int e = L.tab[cl][v][b1 + 1];
            for (; i < e; i++) {
                const float4 p = P[i];
                const float d = dist2f(p.x, p.y, p.z, qx, qy, qz);
                const NnBest key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned int)((__float_as_int(p.w) << 7) | v);
                if (key < nb) { nb = key; npos = i; }
            }
        }
        best = quad_min_u64(nb);
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
    const int apos = quad_max_i(nb == best ? npos : -1);

    // ---- scan-line walk on lines ra-2 .. ra+2 inside the index window the reference's loops can reach
    const int w_lo = ra - 3 >= 0 ? L.lle[cl][ra - 3] + 1 : 0;
    const int w_hi = ra + 3 <= 65 ? L.fge[cl][ra + 3] : n_last;
    const unsigned long long thr = pack_fu(25.0f, 0u);
    unsigned long long same = thr, other = thr;
    int spos = -1, opos = -1;
    // radii: the neighbouring lines right next to the nearest point; the ring gap of far ground points (rho^2 dtheta / h); 5 m
    const float rad[3] = { 0.5f + 0.05f * rho, fminf(5.0f, 1.0f + 0.0045f * rho2), 5.0f };
#pragma unroll
    for (int pass = 0; pass < 3; pass++) {
        if (pass > 0 && rad[pass] <= rad[pass - 1]) continue;
        int b0, b1;
        if (!arc(rad[pass], b0, b1)) return false;
        WalkBest bs = thr, bo = thr;
        int ps = -1, po = -1;
#pragma unroll
        for (int li = 0; li < 5; li++) {
            const int v = ra - 2 + li;
            if (v < 0 || v > 65 || (kEdge && li == 2)) continue;        // edges never use the nearest point's own line
            const int e = L.tab[cl][v][b1 + 1];
            for (int i = L.tab[cl][v][b0] + g; i < e; i += kTG) {
                const float4 p = P[i];
                const int j = __float_as_int(p.w);
                if (j == closest || j < w_lo || j >= w_hi) continue;
                const bool fwd = j > closest;
                const unsigned int seq = fwd ? (unsigned int)(j - closest - 1) : kSeqBack + (unsigned int)(closest - 1 - j);
                const float d = dist2f(p.x, p.y, p.z, qx, qy, qz);
                const WalkBest key = ((unsigned long long)__float_as_uint(d) << 32) | seq;
                const bool is_other = fwd ? (v > ra) : (v < ra);
                if (is_other) { if (key < bo) { bo = key; po = i; } }
                else if (!kEdge) { if (key < bs) { bs = key; ps = i; } }
            }
        }
        same = quad_min_u64(bs);
        other = quad_min_u64(bo);
        spos = quad_max_i(bs == same ? ps : -1);
        opos = quad_max_i(bo == other ? po : -1);
        if (rad[pass] >= 5.0f) break;
        const unsigned long long lim = pack_fu(rad[pass] * rad[pass] * 0.998f, 0u);     // strictly inside the ball of this pass
        if (other < lim && (kEdge || same < lim)) break;
    }
    const int i_other = other < thr ? seq_to_index((unsigned int)(other & 0xffffffffull), closest) : -1;
    if (kEdge) {
        if (i_other >= 0) { out = make_int4(closest, i_other, -1, 1); A = P[apos]; B = P[opos]; }
        return true;
    }
    const int i_same = same < thr ? seq_to_index((unsigned int)(same & 0xffffffffull), closest) : -1;
    if (i_same >= 0 && i_other >= 0) { out = make_int4(closest, i_same, i_other, 2); A = P[apos]; B = P[spos]; C = P[opos]; }
    return true;
}

// step `step`, outer iteration `outer` of every chain: one workgroup per (chain, azimuth sector).  The sector blocks of a chain
// are decoded onto ONE XCD (blocks b and b + 8 share an XCD) so that the chain's clouds are fetched into one L2 only.
__global__ __launch_bounds__(kTT) void k_corr_tile(BatchView b, OdomView o, int step, int outer, unsigned int *wl)
{
    extern __shared__ __align__(16) unsigned char t_raw[];
    TileLds &L = *reinterpret_cast<TileLds *>(t_raw);
    const int xcd = blockIdx.x & 7, u = blockIdx.x >> 3;
    const int c = (u / kTSect) * 8 + xcd;
    const int t = u % kTSect;
    if (c >= o.n_chains) return;
    int own;
    const int k = chain_scan(o, c, step, own);
    if (k < 0) return;
    const int tid = threadIdx.x;
    const int l = k - 1;
    const int n_sharp = b.feat_n[k * 4 + 0];
    const int nq = n_sharp + b.feat_n[k * 4 + 2];
    if (b.status[l] & (kStatusIrregularLines | kStatusDenseCell)) {
        // rare: the whole scan pair goes to the generic search
        if (t == 0) {
            __shared__ unsigned int s_base;
            if (tid == 0) s_base = atomicAdd(wl, (unsigned int)nq);
            __syncthreads();
            for (int qi = tid; qi < nq; qi += kTT) wl[1 + s_base + qi] = ((unsigned int)c << 12) | (unsigned int)qi;
        }
        return;
    }
    const int n_last_c = b.feat_n[l * 4 + 1], n_last_s = b.feat_n[l * 4 + 3];
    const int g0 = t * kTB - kTHalo;             // first loaded bin, unwrapped (negative in sector 0)

    // ---- per-line geometry of the window: N_v points on the line, F0_v = points before bin g0 (periodic extension)
    if (tid < 2 * 66) {
        const int cl = tid / 66, v = tid % 66;
        const int *tg = b.lb_start + (size_t)(l * 2 + cl) * (kLineKeys + 1) + v * kAzBins;
        const int G = tg[0], N = tg[kAzBins] - G;
        const int gb = g0 < 0 ? g0 + kAzBins : g0;
        const int F0 = tg[gb] - G - (g0 < 0 ? N : 0);
        int ue = g0 + kTW, adj = 0;
        if (ue > kAzBins) { ue -= kAzBins; adj = N; }
        const int len = (ue < 0 ? tg[ue + kAzBins] - G - N : tg[ue] - G + adj) - F0;
        L.lineG[cl][v] = G; L.lineN[cl][v] = N; L.lineF0[cl][v] = F0;
        L.lineBase[cl][v] = len;                   // lengths first, prefix below
        L.elev[cl][v] = b.lb_elev[(size_t)(l * 2 + cl) * 66 + v];
        L.fge[cl][v] = b.line_first_ge[(size_t)(l * 2 + cl) * 66 + v];
        L.lle[cl][v] = b.line_last_le[(size_t)(l * 2 + cl) * 66 + v];
    }
    if (tid == 0) { L.nq = 0; L.over[0] = L.over[1] = 0; }
    __syncthreads();
    if (tid < 2) {
        int run = 0;
        for (int v = 0; v < 66; v++) { const int len = L.lineBase[tid][v]; L.lineBase[tid][v] = run; run += len; }
        L.lineBase[tid][66] = run;
        if (run > (tid == 0 ? kTCapC : kTCapS)) L.over[tid] = 1;
    }
    __syncthreads();
    const bool over = L.over[0] || L.over[1];
    if (!over) {
        // ---- local bucket table
        for (int x = tid; x < 2 * 66 * (kTW + 1); x += kTT) {
            const int cl = x / (66 * (kTW + 1)), y = x % (66 * (kTW + 1)), v = y / (kTW + 1), lb = y % (kTW + 1);
            const int *tg = b.lb_start + (size_t)(l * 2 + cl) * (kLineKeys + 1) + v * kAzBins;
            int uu = g0 + lb, adj = 0;
            const int N = L.lineN[cl][v];
            if (uu < 0) { uu += kAzBins; adj = -N; } else if (uu > kAzBins) { uu -= kAzBins; adj = N; }
            L.tab[cl][v][lb] = (unsigned short)(L.lineBase[cl][v] + (tg[uu] - L.lineG[cl][v] + adj) - L.lineF0[cl][v]);
        }
        // ---- the windows' points: flat copy, every element finds its line by a binary search over the line bases
#pragma unroll
        for (int cl = 0; cl < 2; cl++) {
            const int total = L.lineBase[cl][66];
            const float4 *src = cl ? b.lbs_pts + b.off[l] : b.lbc_pts + (size_t)l * kMaxLessSharp;
            float4 *dst = L.pts + (cl ? kTCapC : 0);
            for (int i0 = tid; i0 < total; i0 += 4 * kTT) {
                float4 v4[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int i = i0 + kTT * q;
                    v4[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (i < total) {
                        int lo = 0, hi = 66;
                        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (L.lineBase[cl][mid] <= i) lo = mid; else hi = mid; }
                        const int N = L.lineN[cl][lo];
                        int pos = L.lineF0[cl][lo] + (i - L.lineBase[cl][lo]);     // position on the line, periodic
                        pos = pos < 0 ? pos + N : (pos >= N ? pos - N : pos);
                        v4[q] = src[L.lineG[cl][lo] + pos];
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; q++) { const int i = i0 + kTT * q; if (i < total) dst[i] = v4[q]; }
            }
        }
    }
    // ---- the chain's feature points of this sector (de-skew transform in fp64, as the reference's TransformToStart)
    const double *x = o.state + c * 8;
    for (int qi = tid; qi < nq; qi += kTT) {
        const float4 p = qi < n_sharp ? b.sharp[(size_t)k * kMaxSharp + qi] : b.flat[(size_t)k * kMaxFlat + (qi - n_sharp)];
        double rx, ry, rz;
        quat_rotate(x, (double)p.x, (double)p.y, (double)p.z, rx, ry, rz);
        const float qx = (float)(rx + x[4]), qy = (float)(ry + x[5]), qz = (float)(rz + x[6]);
        if (az_bin(qx, qy) / kTB != t) continue;
        bool keep = !over;
        if (keep) {
            const int slot = atomicAdd(&L.nq, 1);
            if (slot < kTQCap) L.qbuf[slot] = make_float4(qx, qy, qz, __int_as_float(qi));
            else keep = false;
        }
        if (!keep) defer_query(wl, c, qi);
    }
    __syncthreads();
    if (over) return;
    const int nqs = min(L.nq, kTQCap);
    int4 *corr = (int4 *)o.corr + (size_t)c * kMaxQueries;
    int *seed_c = o.seed ? o.seed + (size_t)c * kMaxQueries : nullptr;
    const int g = tid & (kTG - 1);
    for (int base = 0; base < nqs; base += kTT / kTG) {
        const int slot = base + tid / kTG;
        if (slot >= nqs) break;                       // whole quads leave together
        const float4 q = L.qbuf[slot];
        const int qi = __float_as_int(q.w);
        const bool edge = qi < n_sharp;
        // second outer iteration: the first one's nearest point, seen from the updated pose, bounds the search radius
        bool seeded = false;
        float sd = 0.f;
        if (outer == 1 && seed_c) {
            const int sidx = seed_c[qi];
            if (sidx >= 0 && sidx < (edge ? n_last_c : n_last_s)) {
                const float4 pp = (edge ? b.less_sharp + (size_t)l * kMaxLessSharp : b.less_flat + b.off[l])[sidx];
                sd = dist2f(pp.x, pp.y, pp.z, q.x, q.y, q.z);
                seeded = sd < 24.0f;
            }
        }
        int4 r;
        int closest;
        float4 A, B, C;
        const bool done = edge ? tile_search<true>(L, t, g, q.x, q.y, q.z, n_last_c, seeded, sd, r, closest, A, B, C)
                               : tile_search<false>(L, t, g, q.x, q.y, q.z, n_last_s, seeded, sd, r, closest, A, B, C);
        if (g == 0) {
            if (!done) defer_query(wl, c, qi);
            else {
                corr[qi] = r;
                if (outer == 0 && seed_c) seed_c[qi] = closest;
                // residual-block record for the solver: the feature point and its 2 (edge) or 3 (plane) partners, 64 B
                float4 cp = edge ? b.sharp[(size_t)k * kMaxSharp + qi] : b.flat[(size_t)k * kMaxFlat + (qi - n_sharp)];
                cp.w = __int_as_float(r.w);
                float4 *rec = o.crec + ((size_t)c * kMaxQueries + qi) * 4;
                rec[0] = cp; rec[1] = A; rec[2] = B; rec[3] = C;
            }
        }
    }
}

} // namespace lmono
