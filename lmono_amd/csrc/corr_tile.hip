// lmono_amd/csrc/corr_tile.hip -- laserOdometry correspondence search from LDS-staged point neighbourhoods.
//
// Same results as k_correspond (odometry.hip): the exact nearest point (float distance, lowest index on ties) of every de-skewed
// feature point in the previous scan's less-sharp / less-flat cloud, then the reference's scan-line walk.  What
// changes is where the candidates come from.  One workgroup owns ONE AZIMUTH SECTOR of one chain's current scan pair: it copies the
// (scan line, azimuth bin)-sorted "last" clouds of that sector plus a halo into LDS once (kTW of kAzBins bins, both clouds, with a
// local bucket table), collects the chain's feature points whose transformed position falls into the sector, and serves all of
// them from LDS in one round, four lanes (a DPP quad) per feature point.
//
// Exactness.  A point p with |p - q| <= r lies within asin(r / rho_xy(q)) of q's azimuth and within asin(r / |q|) of q's
// elevation angle, so the candidates of a ball are: the lines whose elevation range [lb_elev] meets the elevation window (found
// through a small per-tile table of the ranges' monotone envelopes, then tested line by line), and on each of them the bucket run
// of the azimuth arc (asin bounded from above, +1.5 bins of slack as in k_correspond).  A search with radius r is exact whenever its
// minimum is <= r; otherwise the radius grows to the distance found (or geometrically, up to the 5 m of DISTANCE_SQ_THRESHOLD).
// A feature whose arc leaves the loaded window (near-range points with wide arcs) is NOT answered here: it goes to the device work
// list of k_correspond_list.  The tile path therefore never approximates; it answers exactly or defers (~1 % of the features).
#include "batch.hpp"

namespace lmono {

constexpr int kTSect = 16;                    // azimuth sectors of a scan pair
constexpr int kTB = kAzBins / kTSect;         // bins per sector (24 = 22.5 deg)
constexpr int kTHalo = 12;                    // halo bins on either side (11.25 deg)
constexpr int kTW = kTB + 2 * kTHalo;         // loaded bins
static_assert(kAzBins % kTSect == 0, "sectors must tile the azimuth bins");
constexpr int kTCapC = 1280;                  // LDS capacity: corner points of a sector window
constexpr int kTCapS = 6144;                  //               surf points
constexpr int kTQCap = 256;                   // feature points of one sector served from LDS (the rest is deferred)
constexpr int kTT = 1024;                     // threads per workgroup
constexpr int kTG = 4;                        // lanes per feature point (a DPP quad)
constexpr float kTR0 = 0.3f;                  // first search radius of an unseeded feature (m)
constexpr int kTLinesPerWave = (2 * 66 + kTT / 64 - 1) / (kTT / 64);   // (cloud, line) pairs copied by one wave
constexpr int kTElevBins = 176;               // elevation -> line window table: 0.25 deg steps over [-32, +12) deg
constexpr float kTElevLo = -0.5585054f, kTElevStep = 0.004363323f;     // -32 deg, 0.25 deg in rad

struct TileLds {
    float4 pts[kTCapC + kTCapS];              // corner window, then surf window: x y z (original index bits)
    float4 q[kTQCap];                         // transformed feature point, .w = feature index
    float4 qg[kTQCap];                        // rho_xy, |q|, azimuth + pi, elevation angle
    float qseed[kTQCap];                      // squared distance of the seed (nearest point of the first outer iteration), < 0: none
    unsigned short tab[2][66][kTW + 2];       // local start of every (line, local bin) bucket; [kTW] = end of the line
    float2 elev[2][66];                       // elevation range of every line of the "last" clouds
    unsigned char vfirst[2][kTElevBins];      // first line that can reach down to an elevation bin (66: none)
    signed char vlast[2][kTElevBins];         // last line that can reach up to it (-1: none)
    float4 env[2][66];                        // lb_elev (lo, hi, A, B) staged for the table build
    int lineN[2][66], lineG[2][66], lineF0[2][66], lineLen[2][66], lineBase[2][67];
    int fge[2][66], lle[2][66];
    int nq, nq_ok, over[2];
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

// -DLMONO_TILE_PROF (diagnostic build, scripts/prof_tile.py): cycles per phase of the workgroups and the reasons of the deferrals
// are summed into the context's statistics words; in the product build nothing of this executes
#ifdef LMONO_TILE_PROF
#define TP_STAMP(i) if (threadIdx.x == 0) { tp_t[i] = __builtin_amdgcn_s_memtime(); }
#define TP_REASON(code) (tp_reason = (code))
#define TP_COUNT(var, n) ((var) += (n))
#else
#define TP_STAMP(i)
#define TP_REASON(code)
#define TP_COUNT(var, n)
#endif

__device__ __forceinline__ int elev_bin(float e)
{
    const int bi = (int)floorf((e - kTElevLo) / kTElevStep);
    return bi < 0 ? 0 : (bi >= kTElevBins ? kTElevBins - 1 : bi);
}

// One feature point, the four lanes of a DPP quad (g = lane inside the quad).  Returns false when the point must be deferred.
// All control flow that depends on the feature is quad-uniform, so the quad's DPP exchanges are safe.
template <bool kEdge>
__device__ __forceinline__ bool tile_search(const TileLds &L, int t, int g, float qx, float qy, float qz, float rho, float R, float th, float eq,
                                            int n_last, float seed_d, int4 &out, int &closest_out, float4 &A, float4 &B, float4 &C, int &tp_reason, int *tp_cnt)
{
    constexpr int cl = kEdge ? 0 : 1;
    const float4 *P = L.pts + (kEdge ? 0 : kTCapC);
    out = make_int4(-1, -1, -1, 0);
    closest_out = -1;
    A = B = C = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr float kb = kAzBins / 6.28318531f;
    const int wlo = t * kTB - kTHalo, whi = wlo + kTW;      // loaded bins, unwrapped: [wlo, whi)
    // the feature's own bin lies in [t kTB, (t+1) kTB), so arcs computed from th come out unwrapped the way the window is
    // (negative below bin 0 in sector 0, >= kAzBins past the last bin in the last sector)
    auto arc = [&](float r, int &lb_lo, int &lb_hi) -> bool {
        if (!(rho > r * 1.002f)) return false;
        const float alpha = asin_upper(r / rho) + kArcSlackBins / kb;
        const int lo = (int)floorf((th - alpha) * kb), hi = (int)floorf((th + alpha) * kb);
        if (lo < wlo || hi >= whi) return false;
        lb_lo = lo - wlo; lb_hi = hi - wlo;
        return true;
    };

    // ---- exact nearest point: the quad's lanes take the lines of the elevation window round-robin
    NnBest nb = kNnNone;
    int npos = -1;
    unsigned long long best = kNnNone;
    float r = seed_d >= 0.f ? sqrtf(seed_d) * 1.0005f + 1e-3f : kTR0;
    for (int pass = 0; pass < 12; pass++) {
        const float rr = fminf(r, 5.0f);                    // d2 < 25 means d < 5: a 5 m ball holds every admissible point
        int b0, b1;
        if (!arc(rr, b0, b1)) { TP_REASON(pass == 0 ? (seed_d >= 0.f ? 1 : 2) : 3); return false; }
        TP_COUNT(tp_cnt[0], 1);
        const float beta = asin_upper(fminf(rr / R, 1.0f)) + 5e-4f;
        const float elo = eq - beta, ehi = eq + beta;
        const int v1 = L.vfirst[cl][elev_bin(ehi)], v2 = L.vlast[cl][elev_bin(elo)];
        for (int v = v1 + g; v <= v2; v += kTG) {
            const float2 ev = L.elev[cl][v];
            if (ev.y < elo || ev.x > ehi) continue;
            int i = L.tab[cl][v][b0];
            const int e = L.tab[cl][v][b1 + 1];
            TP_COUNT(tp_cnt[1], e - i); TP_COUNT(tp_cnt[4], 1);
            for (; i < e; i += 2) {
                const int i1 = min(i + 1, e - 1);
                const float4 p0 = P[i];
                const float4 p1 = P[i1];
                const float d0 = dist2f(p0.x, p0.y, p0.z, qx, qy, qz), d1 = dist2f(p1.x, p1.y, p1.z, qx, qy, qz);
                const NnBest k0 = ((unsigned long long)__float_as_uint(d0) << 32) | (unsigned int)((__float_as_int(p0.w) << 7) | v);
                const NnBest k1 = ((unsigned long long)__float_as_uint(d1) << 32) | (unsigned int)((__float_as_int(p1.w) << 7) | v);
                if (k0 < nb) { nb = k0; npos = i; }
                if (k1 < nb) { nb = k1; npos = i1; }
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

    // ---- scan-line walk on lines ra-2 .. ra+2 inside the index window the reference's loops can reach: every lane takes every
    // fourth point of each line's arc
    const int w_lo = ra - 3 >= 0 ? L.lle[cl][ra - 3] + 1 : 0;
    const int w_hi = ra + 3 <= 65 ? L.fge[cl][ra + 3] : n_last;
    const unsigned long long thr = pack_fu(25.0f, 0u);
    unsigned long long same = thr, other = thr;
    int spos = -1, opos = -1;
    const float rad[4] = { walk_radius(0, rho), walk_radius(1, rho), walk_radius(2, rho), walk_radius(3, rho) };
#pragma unroll
    for (int pass = 0; pass < 4; pass++) {
        if (pass > 0 && rad[pass] <= rad[pass - 1]) continue;
        int b0, b1;
        if (!arc(rad[pass], b0, b1)) { TP_REASON(pass < 2 ? 4 : 3 + pass); return false; }
        TP_COUNT(tp_cnt[2], 1);
        WalkBest bs = thr, bo = thr;
        int ps = -1, po = -1;
#pragma unroll
        for (int li = 0; li < 5; li++) {
            const int v = ra - 2 + li;
            if (v < 0 || v > 65 || (kEdge && li == 2)) continue;        // edges never use the nearest point's own line
            const int e = L.tab[cl][v][b1 + 1];
            if (g == 0) TP_COUNT(tp_cnt[3], e - L.tab[cl][v][b0]);
            for (int i = L.tab[cl][v][b0] + g; i < e; i += kTG) {
                const float4 p = P[i];
                const int j = __float_as_int(p.w);
                if (j == closest || j < w_lo || j >= w_hi) continue;
                const bool fwd = j > closest;
                const unsigned int seq = fwd ? (unsigned int)(j - closest - 1) : kSeqBack + (unsigned int)(closest - 1 - j);
                const float d = dist2f(p.x, p.y, p.z, qx, qy, qz);
                const WalkBest key = ((unsigned long long)__float_as_uint(d) << 32) | seq;
                const bool is_other = fwd ? (li > 2) : (li < 2);
                if (is_other) { if (key < bo) { bo = key; po = i; } }
                else if (!kEdge) { if (key < bs) { bs = key; ps = i; } }
            }
        }
        other = quad_min_u64(bo);
        opos = quad_max_i(bo == other ? po : -1);
        if (!kEdge) {
            same = quad_min_u64(bs);
            spos = quad_max_i(bs == same ? ps : -1);
        }
        if (rad[pass] >= 5.0f) break;
        const unsigned long long lim = pack_fu(rad[pass] * rad[pass] * 0.998f, 0u);     // strictly inside the ball of this pass
        if (other < lim && (kEdge || same < lim)) break;
    }
#ifdef LMONO_TILE_PROF
    {   // distance the walk needed, as a fraction of r1 = 0.5 + 0.05 rho: histogram bucket in tp_cnt[5] (0..9, 9 = no partner)
        const unsigned long long need = (kEdge || other > same) ? other : same;
        const float dn = need < thr ? sqrtf(__uint_as_float((unsigned int)(need >> 32))) / (0.5f + 0.05f * rho) : 100.f;
        tp_cnt[5] = dn >= 100.f ? 9 : min(8, (int)(dn * 4.0f));
    }
#endif
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
    const int c = o.chain0 + (u / kTSect) * 8 + xcd;
    const int t = u % kTSect;
    if (c >= o.chain1) return;
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

    // ---- everything that only depends on (chain, sector) is requested first: the chain's feature points and seeds, the raw bucket
    // starts of the window, and -- per wave -- the geometry of the (cloud, line) pairs the wave will copy
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
    // lane j of a wave owns the wave's j-th (cloud, line) pair: pair = wave + 16 j
    const int pair = wave + (kTT / 64) * lane;
    const bool has_pair = lane < kTLinesPerWave && pair < 2 * 66;
    if (has_pair) {
        const int cl = pair / 66, v = pair % 66;
        const int *tg = b.lb_start + (size_t)(l * 2 + cl) * (kLineKeys + 1) + v * kAzBins;
        const int gb = g0 < 0 ? g0 + kAzBins : g0;
        int ue = g0 + kTW;
        const bool wrap_hi = ue > kAzBins;
        if (wrap_hi) ue -= kAzBins;
        const int G = tg[0], tK = tg[kAzBins], tA = tg[gb], tE = tg[ue];       // four independent loads
        const float4 ev = b.lb_elev[(size_t)(l * 2 + cl) * 66 + v];
        const int fge = b.line_first_ge[(size_t)(l * 2 + cl) * 66 + v], lle = b.line_last_le[(size_t)(l * 2 + cl) * 66 + v];
        const int N = tK - G;
        const int F0 = tA - G - (g0 < 0 ? N : 0);
        L.lineG[cl][v] = G; L.lineN[cl][v] = N; L.lineF0[cl][v] = F0; L.lineLen[cl][v] = (tE - G + (wrap_hi ? N : 0)) - F0;
        L.elev[cl][v] = make_float2(ev.x, ev.y); L.env[cl][v] = ev; L.fge[cl][v] = fge; L.lle[cl][v] = lle;
    }
    if (tid == 0) { L.nq = 0; L.nq_ok = 0; L.over[0] = L.over[1] = 0; }
    __syncthreads();
    TP_STAMP(1)
    // exclusive prefix of the line lengths: wave 0 corner cloud, wave 1 surf cloud (lines 0..63 by a wave scan, 64 and 65 behind)
    if (wave < 2) {
        const int len = L.lineLen[wave][lane];
        const int incl = wave_scan_incl(len);
        const int tot64 = __shfl(incl, 63);
        const int l64 = L.lineLen[wave][64], l65 = L.lineLen[wave][65];
        L.lineBase[wave][lane] = incl - len;
        if (lane == 0) {
            L.lineBase[wave][64] = tot64; L.lineBase[wave][65] = tot64 + l64; L.lineBase[wave][66] = tot64 + l64 + l65;
            if (tot64 + l64 + l65 > (wave == 0 ? kTCapC : kTCapS)) L.over[wave] = 1;
        }
    } else if (wave < 8) {
        // elevation -> line window tables from the envelopes of the ranges (A_v = min of lo over lines <= v, B_v = max of hi over
        // lines >= v, both non-increasing): first line whose A reaches down to the bin's upper edge, last line whose B reaches up to
        // its lower edge -- conservative by one bin, every line is tested against its own range in the search
        for (int xx = tid - 128; xx < 2 * kTElevBins; xx += 384) {
            const int cl = xx / kTElevBins, bi = xx % kTElevBins;
            const float e_lo = kTElevLo + kTElevStep * (float)bi, e_hi = e_lo + kTElevStep;
            const bool last_bin = bi == kTElevBins - 1, first_bin = bi == 0;
            int lo = 0, hi = 66;                          // first v with A_v <= e_hi (the top bin stands for everything above it)
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (last_bin || L.env[cl][mid].z <= e_hi) hi = mid; else lo = mid + 1; }
            L.vfirst[cl][bi] = (unsigned char)(lo > 65 ? 66 : lo);
            int lo2 = -1, hi2 = 65;                       // last v with B_v >= e_lo (the bottom bin stands for everything below it)
            while (lo2 < hi2) { const int mid = (lo2 + hi2 + 1) >> 1; if (first_bin || L.env[cl][mid].w >= e_lo) lo2 = mid; else hi2 = mid - 1; }
            L.vlast[cl][bi] = (signed char)lo2;
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
        // ---- the windows' points: every wave copies its (cloud, line) pairs, one 64-point piece of each in flight together
        {
            int pG = 0, pN = 0, pF0 = 0, pLen = 0, pBase = 0;
            if (has_pair) { const int cl = pair / 66, v = pair % 66; pG = L.lineG[cl][v]; pN = L.lineN[cl][v]; pF0 = L.lineF0[cl][v]; pLen = L.lineLen[cl][v]; pBase = L.lineBase[cl][v]; }
            int max_len = pLen;
#pragma unroll
            for (int o2 = 1; o2 < 16; o2 <<= 1) max_len = max(max_len, __shfl_xor(max_len, o2));
            max_len = __shfl(max_len, 0);                 // lanes 0 .. 15 hold the wave's pairs
            for (int i0 = 0; i0 < max_len; i0 += 64) {
                float4 cp[kTLinesPerWave];
#pragma unroll
                for (int j = 0; j < kTLinesPerWave; j++) {
                    const int pj = wave + (kTT / 64) * j;
                    const int G = __shfl(pG, j), N = __shfl(pN, j), F0 = __shfl(pF0, j), len = __shfl(pLen, j);
                    const float4 *src = pj >= 66 ? b.lbs_pts + b.off[l] : b.lbc_pts + (size_t)l * kMaxLessSharp;
                    const int i = i0 + lane;
                    cp[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (pj < 2 * 66 && i < len) {
                        int pos = F0 + i;                 // position on the line, periodic
                        pos = pos < 0 ? pos + N : (pos >= N ? pos - N : pos);
                        cp[j] = src[G + pos];
                    }
                }
#pragma unroll
                for (int j = 0; j < kTLinesPerWave; j++) {
                    const int pj = wave + (kTT / 64) * j;
                    const int len = __shfl(pLen, j), base = __shfl(pBase, j);
                    const int i = i0 + lane;
                    if (pj < 2 * 66 && i < len) L.pts[(pj >= 66 ? kTCapC : 0) + base + i] = cp[j];
                }
            }
        }
    }
    TP_STAMP(4)
    // ---- feature points of this sector: edges fill the queue from the front, planes from the back (waves stay homogeneous)
    {
        float sdv[kQPer];
        bool mine[kQPer];
        float4 tq[kQPer], tg4[kQPer];
#pragma unroll
        for (int j = 0; j < kQPer; j++) {
            const int qi = tid + kTT * j;
            double rx, ry, rz;
            quat_rotate(x, (double)fq[j].x, (double)fq[j].y, (double)fq[j].z, rx, ry, rz);
            tq[j] = make_float4((float)(rx + x[4]), (float)(ry + x[5]), (float)(rz + x[6]), __int_as_float(qi));
            mine[j] = qi < nq && az_bin(tq[j].x, tq[j].y) / kTB == t;
            sdv[j] = -1.0f;
            tg4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
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
            const float rho2 = tq[j].x * tq[j].x + tq[j].y * tq[j].y, rho = sqrtf(rho2);
            tg4[j] = make_float4(rho, sqrtf(rho2 + tq[j].z * tq[j].z), atan2f(tq[j].y, tq[j].x) + 3.14159265f, elev_of(tq[j].x, tq[j].y, tq[j].z));
            if (fsd[j] >= 0) {
                const float sd = dist2f(sp[j].x, sp[j].y, sp[j].z, tq[j].x, tq[j].y, tq[j].z);
                if (sd < 24.0f) sdv[j] = sd;
            }
            bool keep = !over;
            if (keep) {
                const int cnt = atomicAdd(&L.nq, qi < n_sharp ? 1 : 0x10000);    // low half: edges, high half: planes
                const int ne = cnt & 0xffff, np = cnt >> 16;
                if (ne + np < kTQCap) {
                    // once the queue is full every later attempt fails, so the successes of either kind are a prefix of its attempts
                    const int slot = qi < n_sharp ? ne : kTQCap - 1 - np;
                    L.q[slot] = tq[j]; L.qg[slot] = tg4[j]; L.qseed[slot] = sdv[j];
                    atomicAdd(&L.nq_ok, qi < n_sharp ? 1 : 0x10000);
                } else keep = false;
            }
            if (!keep) defer_query(wl, c, qi);
        }
    }
    __syncthreads();
    TP_STAMP(5)
    if (over) return;
    const int n_e = L.nq_ok & 0xffff, n_p = L.nq_ok >> 16;     // features queued (edges from the front, planes from the back)
    int4 *corr = (int4 *)o.corr + (size_t)c * kMaxQueries;
    const int g = tid & (kTG - 1);
#ifdef LMONO_TILE_PROF
    int tp_acc[12] = { 0 };
    int tp_hist[10] = { 0 };
#endif
    constexpr int kQuads = kTT / kTG, kQuadsPerWave = 64 / kTG;
    // edges occupy slots [0, n_e), planes [kTQCap - n_p, kTQCap); quads take the edges first, then the planes
    const int ne_pad = (n_e + kQuadsPerWave - 1) / kQuadsPerWave * kQuadsPerWave;   // planes start on a fresh wave: the two specialised copies never share one
    for (int base = 0; base < ne_pad + n_p; base += kQuads) {
        const int idx = base + tid / kTG;
        const bool edge = idx < ne_pad;
        const int slot = edge ? idx : kTQCap - 1 - (idx - ne_pad);
        if (edge ? idx >= n_e : idx - ne_pad >= n_p) continue;          // whole quads skip together
        const float4 q = L.q[slot], qg = L.qg[slot];
        const int qi = __float_as_int(q.w);
        const float sd = L.qseed[slot];
        int4 r;
        int closest;
        float4 A, B, C;
        int tp_reason = 0;
        int tp_cnt[6] = { 0, 0, 0, 0, 0, -1 };  // NN passes, NN candidates, walk passes, walk candidates, NN lines, walk-distance bucket (diagnostic build)
        const bool done = edge ? tile_search<true>(L, t, g, q.x, q.y, q.z, qg.x, qg.y, qg.z, qg.w, n_last_c, sd, r, closest, A, B, C, tp_reason, tp_cnt)
                               : tile_search<false>(L, t, g, q.x, q.y, q.z, qg.x, qg.y, qg.z, qg.w, n_last_s, sd, r, closest, A, B, C, tp_reason, tp_cnt);
#ifdef LMONO_TILE_PROF
        { const int eo = edge ? 0 : 6; tp_acc[eo + 1] += tp_cnt[1]; tp_acc[eo + 3] += tp_cnt[3]; tp_acc[eo + 4] += tp_cnt[4];
          if (g == 0) { tp_acc[eo + 0] += tp_cnt[0]; tp_acc[eo + 2] += tp_cnt[2]; tp_acc[eo + 5] += 1; if (tp_cnt[5] >= 0) tp_hist[tp_cnt[5]] += 1; } }
#endif
        if (g == 0) {
#ifdef LMONO_TILE_PROF
            if (!done) atomicAdd(&stats[8 + tp_reason], 1ull);
#endif
            if (!done) defer_query(wl, c, qi);
            else {
                corr[qi] = r;
                if (outer == 0 && seed_c) seed_c[qi] = closest;
                // residual-block record for the solver: the feature point and its 2 (edge) or 3 (plane) partners, 64 B
                float4 cp0 = edge ? b.sharp[(size_t)k * kMaxSharp + qi] : b.flat[(size_t)k * kMaxFlat + (qi - n_sharp)];
                cp0.w = __int_as_float(r.w);
                float4 *rec = o.crec + ((size_t)c * kMaxQueries + qi) * 4;
                rec[0] = cp0; rec[1] = A; rec[2] = B; rec[3] = C;
            }
        }
    }
#ifdef LMONO_TILE_PROF
    for (int i = 0; i < 12; i++) { const int v = wave_sum_i(tp_acc[i]); if (lane == 0 && v) atomicAdd(&stats[16 + i], (unsigned long long)v); }
    for (int i = 0; i < 10; i++) { const int v = wave_sum_i(tp_hist[i]); if (lane == 0 && v) atomicAdd(&stats[28 + i], (unsigned long long)v); }
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
