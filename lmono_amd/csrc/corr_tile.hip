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
constexpr int kTG = 16;                       // lanes per feature point (a DPP row)
constexpr float kTR0 = 0.3f;                  // first search radius of an unseeded feature (m)

struct TileLds {
    float4 pts[kTCapC + kTCapS];              // corner window, then surf window: x y z (original index bits)
    float4 qbuf[kTQCap];                      // transformed feature point, .w = feature index
    float qseed[kTQCap];                      // squared distance of the seed (nearest point of the first outer iteration), < 0: none
    unsigned short tab[2][66][kTW + 2];       // local start of every (line, local bin) bucket; [kTW] = end of the line
    float2 elev[2][66];                       // elevation range of every line of the "last" clouds
    int lineN[2][66], lineG[2][66], lineF0[2][66], lineBase[2][67];
    int fge[2][66], lle[2][66];
    int nq, over[2];
};
constexpr int kTileLds = (int)sizeof(TileLds);
static_assert(sizeof(TileLds) <= 160 * 1024, "tile does not fit LDS");

// minimum over the 16 lanes of a DPP row, returned to all of them (rotations by 8, 4, 2, 1 inside the row)
__device__ __forceinline__ unsigned long long row_min_u64(unsigned long long v)
{
    unsigned long long o = dpp_perm_u64<0x128>(v);      // row_ror:8
    v = o < v ? o : v;
    o = dpp_perm_u64<0x124>(v);                         // row_ror:4
    v = o < v ? o : v;
    o = dpp_perm_u64<0x122>(v);                         // row_ror:2
    v = o < v ? o : v;
    o = dpp_perm_u64<0x121>(v);                         // row_ror:1
    return o < v ? o : v;
}
__device__ __forceinline__ int row_max_i(int v)
{
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x128, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x124, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x122, 0xf, 0xf, false));
    return max(v, __builtin_amdgcn_update_dpp(v, v, 0x121, 0xf, 0xf, false));
}

// work list of deferred feature points: [0] = count, then (chain << 12 | feature index)
__device__ __forceinline__ void defer_query(unsigned int *wl, int c, int qi)
{
    const unsigned int slot = atomicAdd(wl, 1u);
    wl[1 + slot] = ((unsigned int)c << 12) | (unsigned int)qi;
}

// -DLMONO_TILE_PROF (diagnostic build, scripts/prof_tile.py): cycles per phase of the workgroups and the reasons of the deferrals
// are summed into the context's statistics words; in the product build nothing of this executes
#ifdef LMONO_TILE_PROF
#define TP_STAMP(i) if (threadIdx.x == 0) { tp_t[i] = __builtin_amdgcn_s_memtime(); }
#define TP_REASON(code) (tp_reason = (code))
#else
#define TP_STAMP(i)
#define TP_REASON(code)
#endif

// One feature point, the 16 lanes of a DPP row (g = lane inside the row, row = row inside the wave).  Returns false when the
// point must be deferred.  All control flow that depends on the feature is row-uniform, so the row's DPP exchanges are safe.
//
// Nearest point: one lane per scan line of the elevation window.  The lines a ball of radius r can meet are found by testing
// every line's elevation range (66 lines = five rounds of 16 lanes, the hits collected with wave ballots); lane i then sweeps
// the azimuth arc of the i-th line from the first hit on -- one contiguous run of the LDS point array.
template <bool kEdge>
__device__ __forceinline__ bool tile_search(const TileLds &L, int t, int g, int row, float qx, float qy, float qz, int n_last, float seed_d,
                                            int4 &out, int &closest_out, float4 &A, float4 &B, float4 &C, int &tp_reason)
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
    auto arc = [&](float r, int &lb_lo, int &lb_hi) -> bool {
        if (!(rho > r * 1.002f)) return false;
        const float alpha = asin_upper(r / rho) + 1.5f / kb;
        const int lo = (int)floorf((th - alpha) * kb), hi = (int)floorf((th + alpha) * kb);
        if (lo < wlo || hi >= whi) return false;
        lb_lo = lo - wlo; lb_hi = hi - wlo;
        return true;
    };

    // ---- exact nearest point
    NnBest nb = kNnNone;
    int npos = -1;
    unsigned long long best = kNnNone;
    float r = seed_d >= 0.f ? sqrtf(seed_d) * 1.0005f + 1e-3f : kTR0;
    for (int pass = 0; pass < 12; pass++) {
        const float rr = fminf(r, 5.0f);                    // d2 < 25 means d < 5: a 5 m ball holds every admissible point
        int b0, b1;
        if (!arc(rr, b0, b1)) { TP_REASON(pass == 0 ? (seed_d >= 0.f ? 1 : 2) : 3); return false; }
        const float beta = asin_upper(fminf(rr / R, 1.0f)) + 5e-4f;
        const float elo = eq - beta, ehi = eq + beta;
        // lines whose elevation range meets [elo, ehi]: bit v of (mlo, mhi)
        unsigned long long mlo = 0ull;
        unsigned int mhi = 0u;
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const int v = 16 * j + g;
            bool hit = false;
            if (v < 66) { const float2 ev = L.elev[cl][v]; hit = !(ev.y < elo || ev.x > ehi); }
            const unsigned int m16 = (unsigned int)(__ballot(hit) >> (16 * row)) & 0xffffu;
            if (j < 4) mlo |= (unsigned long long)m16 << (16 * j); else mhi = m16;
        }
        if (mlo | mhi) {
            const int v_first = mlo ? __ffsll((long long)mlo) - 1 : 64 + __ffs((int)mhi) - 1;
            const int v_last = mhi ? 64 + (31 - __clz((int)mhi)) : 63 - __clzll((long long)mlo);
            for (int v0 = v_first; v0 <= v_last; v0 += kTG) {
                const int v = v0 + g;
                const bool on = v <= v_last && (v < 64 ? (mlo >> v) & 1ull : (mhi >> (v - 64)) & 1u);
                if (on) {
                    int i = L.tab[cl][v][b0];
                    const int e = L.tab[cl][v][b1 + 1];
                    for (; i < e; i += 2) {
                        const float4 p0 = P[i];
                        const float4 p1 = P[min(i + 1, e - 1)];
                        const float d0 = dist2f(p0.x, p0.y, p0.z, qx, qy, qz), d1 = dist2f(p1.x, p1.y, p1.z, qx, qy, qz);
                        const NnBest k0 = ((unsigned long long)__float_as_uint(d0) << 32) | (unsigned int)((__float_as_int(p0.w) << 7) | v);
                        const NnBest k1 = ((unsigned long long)__float_as_uint(d1) << 32) | (unsigned int)((__float_as_int(p1.w) << 7) | v);
                        if (k0 < nb) { nb = k0; npos = i; }
                        if (k1 < nb) { nb = k1; npos = min(i + 1, e - 1); }
                    }
                }
            }
        }
        best = row_min_u64(nb);
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
    const int apos = row_max_i(nb == best ? npos : -1);

    // ---- scan-line walk on lines ra-2 .. ra+2 inside the index window the reference's loops can reach: three lanes per line
    // (lane 15 idle), every lane takes every third point of its line's arc
    const int w_lo = ra - 3 >= 0 ? L.lle[cl][ra - 3] + 1 : 0;
    const int w_hi = ra + 3 <= 65 ? L.fge[cl][ra + 3] : n_last;
    const unsigned long long thr = pack_fu(25.0f, 0u);
    unsigned long long same = thr, other = thr;
    int spos = -1, opos = -1;
    const int wli = g / 3, wsub = g - 3 * wli;               // line ra - 2 + wli
    const int wv = ra - 2 + wli;
    const bool wline = wli < 5 && wv >= 0 && wv <= 65 && !(kEdge && wli == 2);   // edges never use the nearest point's own line
    // radii: the neighbouring lines right next to the nearest point; the ring gap of far ground points (rho^2 dtheta / h); 5 m
    const float rad[3] = { 0.5f + 0.05f * rho, fminf(5.0f, 1.0f + 0.0045f * rho2), 5.0f };
#pragma unroll
    for (int pass = 0; pass < 3; pass++) {
        if (pass > 0 && rad[pass] <= rad[pass - 1]) continue;
        int b0, b1;
        if (!arc(rad[pass], b0, b1)) { TP_REASON(4 + pass); return false; }
        WalkBest bs = thr, bo = thr;
        int ps = -1, po = -1;
        if (wline) {
            const int e = L.tab[cl][wv][b1 + 1];
            for (int i = L.tab[cl][wv][b0] + wsub; i < e; i += 3) {
                const float4 p = P[i];
                const int j = __float_as_int(p.w);
                if (j == closest || j < w_lo || j >= w_hi) continue;
                const bool fwd = j > closest;
                const unsigned int seq = fwd ? (unsigned int)(j - closest - 1) : kSeqBack + (unsigned int)(closest - 1 - j);
                const float d = dist2f(p.x, p.y, p.z, qx, qy, qz);
                const WalkBest key = ((unsigned long long)__float_as_uint(d) << 32) | seq;
                const bool is_other = fwd ? (wv > ra) : (wv < ra);
                if (is_other) { if (key < bo) { bo = key; po = i; } }
                else if (!kEdge) { if (key < bs) { bs = key; ps = i; } }
            }
        }
        other = row_min_u64(bo);
        opos = row_max_i(bo == other ? po : -1);
        if (!kEdge) {
            same = row_min_u64(bs);
            spos = row_max_i(bs == same ? ps : -1);
        }
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
__global__ __launch_bounds__(kTT) void k_corr_tile(BatchView b, OdomView o, int step, int outer, unsigned int *wl, unsigned long long *stats)
{
#ifdef LMONO_TILE_PROF
    unsigned long long tp_t[8] = { 0 };
#endif
    TP_STAMP(0)
    extern __shared__ __align__(16) unsigned char t_raw[];
    TileLds &L = *reinterpret_cast<TileLds *>(t_raw);
    const int xcd = blockIdx.x & 7, u = blockIdx.x >> 3;
    const int c = (u / kTSect) * 8 + xcd;
    const int t = u % kTSect;
    if (c >= o.n_chains) return;
    int own;
    const int k = chain_scan(o, c, step, own);
    if (k < 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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

    // ---- the chain's feature points are requested first (their latency runs under the window set-up below): de-skew transform in
    // fp64 as the reference's TransformToStart, azimuth sector; second outer iteration: the first one's nearest point as a seed
    constexpr int kQPer = (kMaxQueries + kTT - 1) / kTT;      // 3 feature points per thread at most
    const double *x = o.state + c * 8;
    int *seed_c = o.seed ? o.seed + (size_t)c * kMaxQueries : nullptr;
    float4 fq[kQPer];
    int fsd[kQPer];
#pragma unroll
    for (int j = 0; j < kQPer; j++) {
        const int qi = tid + kTT * j;
        fq[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        fsd[j] = -1;
        if (qi < nq) {
            fq[j] = qi < n_sharp ? b.sharp[(size_t)k * kMaxSharp + qi] : b.flat[(size_t)k * kMaxFlat + (qi - n_sharp)];
            if (outer == 1 && seed_c) fsd[j] = seed_c[qi];
        }
    }

    // ---- per-line geometry of the window: N_v points on the line, F0_v = points before bin g0 (periodic extension)
    if (tid < 2 * 66) {
        const int cl = tid / 66, v = tid % 66;
        const int *tg = b.lb_start + (size_t)(l * 2 + cl) * (kLineKeys + 1) + v * kAzBins;
        const int gb = g0 < 0 ? g0 + kAzBins : g0;
        int ue = g0 + kTW;
        const bool wrap_hi = ue > kAzBins;
        if (wrap_hi) ue -= kAzBins;
        const int G = tg[0], tK = tg[kAzBins], tA = tg[gb], tE = tg[ue];       // four independent loads
        const float4 ev4 = b.lb_elev[(size_t)(l * 2 + cl) * 66 + v];
        const float2 ev = make_float2(ev4.x, ev4.y);
        const int fge = b.line_first_ge[(size_t)(l * 2 + cl) * 66 + v], lle = b.line_last_le[(size_t)(l * 2 + cl) * 66 + v];
        const int N = tK - G;
        const int F0 = tA - G - (g0 < 0 ? N : 0);
        const int len = (tE - G + (wrap_hi ? N : 0)) - F0;
        L.lineG[cl][v] = G; L.lineN[cl][v] = N; L.lineF0[cl][v] = F0;
        L.lineBase[cl][v] = len;                   // lengths first, prefix below
        L.elev[cl][v] = ev; L.fge[cl][v] = fge; L.lle[cl][v] = lle;
    }
    // raw bucket starts of the window, requested before the line bases are known (they only depend on the sector)
    constexpr int kTabN = 2 * 66 * (kTW + 1);
    constexpr int kTabPer = (kTabN + kTT - 1) / kTT;
    int traw[kTabPer];
#pragma unroll
    for (int j = 0; j < kTabPer; j++) {
        const int xx = tid + kTT * j;
        traw[j] = 0;
        if (xx < kTabN) {
            const int cl = xx / (66 * (kTW + 1)), y = xx % (66 * (kTW + 1)), v = y / (kTW + 1), lb = y % (kTW + 1);
            int uu = g0 + lb;
            uu = uu < 0 ? uu + kAzBins : (uu > kAzBins ? uu - kAzBins : uu);
            traw[j] = b.lb_start[(size_t)(l * 2 + cl) * (kLineKeys + 1) + v * kAzBins + uu];
        }
    }
    if (tid == 0) { L.nq = 0; L.over[0] = L.over[1] = 0; }
    __syncthreads();
    TP_STAMP(1)
    // exclusive prefix of the line lengths: wave 0 corner cloud, wave 1 surf cloud (lines 0..63 by a wave scan, 64 and 65 behind)
    if (wave < 2) {
        const int len = L.lineBase[wave][lane];
        const int incl = wave_scan_incl(len);
        const int tot64 = __shfl(incl, 63);
        const int l64 = L.lineBase[wave][64], l65 = L.lineBase[wave][65];
        L.lineBase[wave][lane] = incl - len;
        if (lane == 0) {
            L.lineBase[wave][64] = tot64; L.lineBase[wave][65] = tot64 + l64; L.lineBase[wave][66] = tot64 + l64 + l65;
            if (tot64 + l64 + l65 > (wave == 0 ? kTCapC : kTCapS)) L.over[wave] = 1;
        }
    }
    __syncthreads();
    TP_STAMP(2)
    const bool over = L.over[0] || L.over[1];
    if (!over) {
        // ---- local bucket table from the raw starts
#pragma unroll
        for (int j = 0; j < kTabPer; j++) {
            const int xx = tid + kTT * j;
            if (xx < kTabN) {
                const int cl = xx / (66 * (kTW + 1)), y = xx % (66 * (kTW + 1)), v = y / (kTW + 1), lb = y % (kTW + 1);
                const int uu = g0 + lb, N = L.lineN[cl][v];
                const int adj = uu < 0 ? -N : (uu > kAzBins ? N : 0);
                L.tab[cl][v][lb] = (unsigned short)(L.lineBase[cl][v] + (traw[j] - L.lineG[cl][v] + adj) - L.lineF0[cl][v]);
            }
        }
        TP_STAMP(3)
        // ---- the windows' points: flat copy, every element finds its line by a binary search over the line bases
#pragma unroll
        for (int cl = 0; cl < 2; cl++) {
            const int total = L.lineBase[cl][66];
            const float4 *src = cl ? b.lbs_pts + b.off[l] : b.lbc_pts + (size_t)l * kMaxLessSharp;
            float4 *dst = L.pts + (cl ? kTCapC : 0);
            constexpr int kU = 6;
            for (int i0 = tid; i0 < total; i0 += kU * kTT) {
                float4 v4[kU];
#pragma unroll
                for (int q = 0; q < kU; q++) {
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
                for (int q = 0; q < kU; q++) { const int i = i0 + kTT * q; if (i < total) dst[i] = v4[q]; }
            }
        }
    }
    TP_STAMP(4)
    // ---- feature points of this sector: edges fill the queue from the front, planes from the back (rows stay homogeneous)
    {
        float sdv[kQPer];
        bool mine[kQPer];
        float4 tq[kQPer];
#pragma unroll
        for (int j = 0; j < kQPer; j++) {
            const int qi = tid + kTT * j;
            double rx, ry, rz;
            quat_rotate(x, (double)fq[j].x, (double)fq[j].y, (double)fq[j].z, rx, ry, rz);
            tq[j] = make_float4((float)(rx + x[4]), (float)(ry + x[5]), (float)(rz + x[6]), __int_as_float(qi));
            mine[j] = qi < nq && az_bin(tq[j].x, tq[j].y) / kTB == t;
            sdv[j] = -1.0f;
        }
        // seed points of this sector's features (second outer iteration), requested together
        float4 sp[kQPer];
#pragma unroll
        for (int j = 0; j < kQPer; j++) {
            const int qi = tid + kTT * j;
            const bool edge = qi < n_sharp;
            sp[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (mine[j] && fsd[j] >= 0 && fsd[j] < (edge ? n_last_c : n_last_s))
                sp[j] = (edge ? b.less_sharp + (size_t)l * kMaxLessSharp : b.less_flat + b.off[l])[fsd[j]];
            else fsd[j] = -1;
        }
#pragma unroll
        for (int j = 0; j < kQPer; j++) {
            const int qi = tid + kTT * j;
            if (!mine[j]) continue;
            if (fsd[j] >= 0) {
                const float sd = dist2f(sp[j].x, sp[j].y, sp[j].z, tq[j].x, tq[j].y, tq[j].z);
                if (sd < 24.0f) sdv[j] = sd;
            }
            bool keep = !over;
            if (keep) {
                const int cnt = atomicAdd(&L.nq, qi < n_sharp ? 1 : 0x10000);    // low half: edges, high half: planes
                const int ne = cnt & 0xffff, np = cnt >> 16;
                if (ne + np < kTQCap) {
                    const int slot = qi < n_sharp ? ne : kTQCap - 1 - np;
                    L.qbuf[slot] = tq[j];
                    L.qseed[slot] = sdv[j];
                } else keep = false;
            }
            if (!keep) defer_query(wl, c, qi);
        }
    }
    __syncthreads();
    TP_STAMP(5)
    if (over) return;
    const int n_e = min(L.nq & 0xffff, kTQCap), n_p = min(L.nq >> 16, kTQCap - n_e);
    int4 *corr = (int4 *)o.corr + (size_t)c * kMaxQueries;
    const int g = tid & (kTG - 1), row = (tid >> 4) & 3;
    constexpr int kRows = kTT / kTG;
    // edges occupy slots [0, n_e), planes [kTQCap - n_p, kTQCap); rows take the edges first, then the planes
    const int ne_pad = (n_e + 3) & ~3;           // planes start on a fresh wave: the two specialised copies never share one
    for (int base = 0; base < ne_pad + n_p; base += kRows) {
        const int idx = base + tid / kTG;
        const bool edge = idx < ne_pad;
        const int slot = edge ? idx : kTQCap - 1 - (idx - ne_pad);
        if (edge ? idx >= n_e : idx - ne_pad >= n_p) continue;          // whole rows skip together
        const float4 q = L.qbuf[slot];
        const int qi = __float_as_int(q.w);
        const float sd = L.qseed[slot];
        int4 r;
        int closest;
        float4 A, B, C;
        int tp_reason = 0;
        const bool done = edge ? tile_search<true>(L, t, g, row, q.x, q.y, q.z, n_last_c, sd, r, closest, A, B, C, tp_reason)
                               : tile_search<false>(L, t, g, row, q.x, q.y, q.z, n_last_s, sd, r, closest, A, B, C, tp_reason);
        if (g == 0) {
#ifdef LMONO_TILE_PROF
            if (!done) atomicAdd(&stats[8 + tp_reason], 1ull);
#endif
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
#ifdef LMONO_TILE_PROF
    __syncthreads();
    TP_STAMP(6)
    if (tid == 0) {
        atomicAdd(&stats[1], 1ull);
        for (int i = 0; i < 6; i++) atomicAdd(&stats[2 + i], tp_t[i + 1] - tp_t[i]);
        atomicAdd(&stats[15], (unsigned long long)(n_e + n_p));
    }
#endif
}

} // namespace lmono
