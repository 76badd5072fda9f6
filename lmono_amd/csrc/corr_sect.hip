// lmono_amd/csrc/corr_sect.hip -- the flattened candidate sweeps of corr_flat.hip over an LDS-STAGED azimuth sector.
//
// k_corr_flat's time goes into its candidate gathers: a round's ~2000 candidates sit in ~1000 runs of about two points, one cache line
// each, fetched by 16-B gathers that miss L1 (profiles/r2/NOTES.md: 66 % of a workgroup's cycles, 41 % of a launch stalled on lines whose
// fill is in flight).  Here a workgroup serves the feature points of ONE azimuth sector of a chain's scan pair (kCsSect sectors of
// kCsSW bins; k_lm_solve leaves the features of the next search sorted by sector, feat_sectors below): it first copies the sector's
// window -- all 66 scan lines x (sector + kCsHalo bins on both sides) of the two "last" clouds' (line, azimuth bin)-sorted copies and the
// matching slice of their start tables -- into LDS with coalesced loads, then runs the SAME rounds (run requests -> resolve -> flattened
// sweep -> decide; nearest point, then the scan-line walk with its radius ladder and the seeded second outer iteration) with every
// table lookup and every candidate read served from LDS.  A feature whose ball leaves the window (arc wider than the halo) is handed
// to the per-chain deferred list, which a second launch of k_corr_flat (list mode, global memory) serves: same arithmetic, so the
// correspondences are identical whichever kernel produced them.
#include "batch.hpp"

namespace lmono {

constexpr int kCsT = 256;                          // threads per workgroup
constexpr int kCsF = 128;                          // feature points per batch (lanes 0 .. 127 own one each)
constexpr int kCsPer = 4;                          // run requests resolved per lane and round
constexpr int kCsPool = kCsT * kCsPer;
constexpr int kCsSect = 32;                        // azimuth sectors
static_assert(kCsSect == kCsSectFwd, "k_lm_solve's scratch is sized for the sector count");
constexpr int kCsSW = kAzBins / kCsSect;           // 12 bins = 11.25 deg
constexpr int kCsHalo = 8;                         // 7.5 deg on both sides
constexpr int kCsWB = kCsSW + 2 * kCsHalo;         // window: 28 bins
constexpr int kCsCapS = 2304, kCsCapC = 576;       // staged points: less flat / less sharp (7.3 % of the clouds on average)
static_assert(kAzBins % kCsSect == 0, "sectors must tile the azimuth bins");

struct CsLds {
    float4 pts_s[kCsCapS];
    float4 pts_c[kCsCapC];
    unsigned short tg[2][66][kCsWB + 1];           // first staged point of every (cloud, line, window bin); entry kCsWB = end of the row
    float4 elev[2][66];
    int fge[2][66], lle[2][66];
    float4 q[kCsF];
    unsigned long long best[kCsF], same[kCsF], other[kCsF];
    int closest[kCsF], wlo[kCsF], whi[kCsF];
    CfRun pool[kCsPool];
    int row_len[2 * 66 + 4];
    int n_pool, n_cand, wsum[kCsT / 64], ovf;
};
static_assert(sizeof(CsLds) <= 80 * 1024, "two workgroups per CU");

// The features of the NEXT search of chain c (scan pair (k - 1, k) at pose x) sorted by azimuth sector of their de-skewed position:
// fs_off[c][0 .. kCsSect] and fs_list[c][..] (feature indices).  Run by the chain's solver workgroup at the end of a solve (and by
// k_feat_sectors before the first search): the pose is the one the search will use, the de-skew the search's own arithmetic.
__device__ __forceinline__ void feat_sectors(const BatchView &b, const OdomView &o, int c, int k, const double *x, bool thin, int *s_cnt /* [kCsSect + 1] LDS */)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    const int n_sharp = b.feat_n[k * 4 + 0], nq = n_sharp + b.feat_n[k * 4 + 2];
    for (int i = tid; i <= kCsSect; i += nt) s_cnt[i] = 0;
    __syncthreads();
    constexpr float kb = kAzBins / 6.28318531f;
    auto sector_of = [&](int qi) {
        const float4 fp = qi < n_sharp ? b.sharp[(size_t)k * kMaxSharp + qi] : b.flat[(size_t)k * kMaxFlat + (qi - n_sharp)];
        double rx, ry, rz;
        quat_rotate(x, (double)fp.x, (double)fp.y, (double)fp.z, rx, ry, rz);
        const float qx = (float)(rx + x[4]), qy = (float)(ry + x[5]);
        const float th = atan2f(qy, qx) + 3.14159265f;
        int fb = (int)floorf(th * kb);
        fb = fb < 0 ? 0 : (fb >= kAzBins ? kAzBins - 1 : fb);
        return fb / kCsSW;
    };
    for (int qi = tid; qi < nq; qi += nt) {
        if (thin && (qi % kThinBlocks) % kThinStride != 0) continue;
        atomicAdd(&s_cnt[sector_of(qi)], 1);
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int s = 0; s <= kCsSect; s++) { const int n = s_cnt[s]; s_cnt[s] = run; o.fs_off[c * (kCsSect + 1) + s] = run; run += n; }
    }
    __syncthreads();
    unsigned short *list = o.fs_list + (size_t)c * kMaxQueries;
    for (int qi = tid; qi < nq; qi += nt) {
        if (thin && (qi % kThinBlocks) % kThinStride != 0) continue;
        const int pos = atomicAdd(&s_cnt[sector_of(qi)], 1);
        if (pos >= 0 && pos < kMaxQueries) list[pos] = (unsigned short)qi;
    }
}

__global__ __launch_bounds__(256) void k_feat_sectors(BatchView b, OdomView o, int step)
{
    __shared__ int s_cnt[kCsSect + 1];
    const int c = o.clist ? o.clist[o.chain0 + blockIdx.x] : o.chain0 + (int)blockIdx.x;
    int own;
    const int k = chain_scan(o, c, step, own);
    if (threadIdx.x == 0 && o.dl_cnt) o.dl_cnt[c] = 0;
    if (k < 0) return;
    double x[7];
    for (int i = 0; i < 7; i++) x[i] = o.state[c * 8 + i];
    feat_sectors(b, o, c, k, x, lead_in_thinned(o, k, own), s_cnt);
}

// stages 1b and 2 of a round over the staged window (cf. cf_sweep): table entries and candidates come from LDS
template <bool kWalk>
__device__ __forceinline__ void cs_sweep(CsLds &L)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_pool = min(L.n_pool, kCsPool);
    const unsigned short *tgl = &L.tg[0][0][0];
    unsigned int st[kCsPer], en[kCsPer];
#pragma unroll
    for (int j = 0; j < kCsPer; j++) {
        const int i = tid * kCsPer + j;
        st[j] = 0; en[j] = 0;
        if (i < n_pool) { const CfRun rq = L.pool[i]; st[j] = tgl[rq.start]; en[j] = tgl[rq.pre]; }
    }
    int sum = 0;
#pragma unroll
    for (int j = 0; j < kCsPer; j++) sum += (int)(en[j] - st[j]);
    const int incl = wave_scan_incl(sum);
    if (lane == 63) L.wsum[wave] = incl;
    __syncthreads();
    int run = incl - sum;
    for (int w = 0; w < wave; w++) run += L.wsum[w];
    if (tid == kCsT - 1) L.n_cand = run + sum;
#pragma unroll
    for (int j = 0; j < kCsPer; j++) {
        const int i = tid * kCsPer + j;
        if (i < n_pool) { L.pool[i].start = st[j]; L.pool[i].pre = (unsigned int)run; L.pool[i].len = (unsigned short)(en[j] - st[j]); }
        run += (int)(en[j] - st[j]);
    }
    __syncthreads();
    const int T = L.n_cand;
    if (T <= 0 || n_pool <= 0) return;
    const int ch = (T + kCsT - 1) / kCsT;
    int j0 = tid * ch;
    const int j1 = min(j0 + ch, T);
    if (j0 >= j1) return;
    int seg;
    {
        int lo = 0, hi = n_pool - 1;
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if ((int)L.pool[mid].pre <= j0) lo = mid; else hi = mid - 1; }
        seg = lo;
    }
    CfRun cur = L.pool[seg];
    int off = j0 - (int)cur.pre;
    int owner = -1;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    unsigned long long m0 = ~0ull, m1 = ~0ull;
    int closest = 0, w_lo = 0, w_hi = 0;
    auto flush = [&]() {
        if (owner < 0) return;
        if (!kWalk) { if (m0 != ~0ull) atomicMin(&L.best[owner], m0); }
        else { if (m0 != ~0ull) atomicMin(&L.same[owner], m0); if (m1 != ~0ull) atomicMin(&L.other[owner], m1); }
    };
    for (; j0 < j1; j0++) {
        while (off >= (int)cur.len) { off -= (int)cur.len; seg++; cur = L.pool[seg]; }
        const float4 p = (cur.tag & 0x80) ? L.pts_s[cur.start + off] : L.pts_c[cur.start + off];
        off++;
        const int ow = cur.owner, tag = cur.tag & 0x7f;
        if (ow != owner) {
            flush();
            owner = ow; m0 = ~0ull; m1 = ~0ull;
            const float4 qq = L.q[ow];
            qx = qq.x; qy = qq.y; qz = qq.z;
            if (kWalk) { closest = L.closest[ow]; w_lo = L.wlo[ow]; w_hi = L.whi[ow]; }
        }
        const float d = dist2f(p.x, p.y, p.z, qx, qy, qz);
        if (!kWalk) {
            const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned int)((__float_as_int(p.w) << 7) | tag);
            m0 = key < m0 ? key : m0;
        } else {
            const int jj = __float_as_int(p.w);
            if (jj == closest || jj < w_lo || jj >= w_hi) continue;
            const bool fwd = jj > closest;
            const unsigned int seq = fwd ? (unsigned int)(jj - closest - 1) : kSeqBack + (unsigned int)(closest - 1 - jj);
            const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | seq;
            const bool is_other = fwd ? (tag > 2) : (tag < 2);
            if (is_other) m1 = key < m1 ? key : m1; else m0 = key < m0 ? key : m0;
        }
    }
    flush();
}

// local run request of the arc (cf_arc, global bins) on line v of cloud cl, or false when the arc leaves the staged window
__device__ __forceinline__ bool cs_local_arc(const CfArc &a, int wb0, int &ls, int &le)
{
    const int n = (a.a1 - a.a0) + a.w1;
    if (n >= kAzBins) return false;
    ls = a.a0 - wb0;
    ls = ls < 0 ? ls + kAzBins : (ls >= kAzBins ? ls - kAzBins : ls);
    le = ls + n;
    return le <= kCsWB;
}

__global__ __launch_bounds__(kCsT, 2) void k_corr_sect(BatchView b, OdomView o, int step, int outer, int defer_every, unsigned long long *dbg)
{
    extern __shared__ __align__(16) unsigned char cs_smem[];
    CsLds &L = *reinterpret_cast<CsLds *>(cs_smem);
    const int xcd = blockIdx.x & 7, u = blockIdx.x >> 3;
    const int ci = o.chain0 + (u / kCsSect) * 8 + xcd;
    const int sct = u % kCsSect;
    if (ci >= o.chain1) return;
    const int c = o.clist ? o.clist[ci] : ci;
    int own;
    const int k = chain_scan(o, c, step, own);
    if (k < 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l = k - 1;
    const int f_begin = o.fs_off[c * (kCsSect + 1) + sct], n_f = o.fs_off[c * (kCsSect + 1) + sct + 1] - f_begin;
    if (n_f <= 0) return;
    const unsigned short *flist = o.fs_list + (size_t)c * kMaxQueries + f_begin;
    unsigned short *dl = o.dl + (size_t)c * kMaxQueries;
    auto defer = [&](int qi) { const int slot = atomicAdd(&o.dl_cnt[c], 1); if (slot >= 0 && slot < kMaxQueries) dl[slot] = (unsigned short)qi; else if (dbg) atomicAdd(&dbg[2], 1ull); };
    const int n_sharp = b.feat_n[k * 4 + 0];
    if (b.status[l] & (kStatusIrregularLines | kStatusDenseCell)) {
        for (int i = tid; i < n_f; i += kCsT) defer(flist[i]);        // rare: the whole scan pair goes to the generic searches
        return;
    }
    // ---- stage the window: bins [wb0, wb0 + kCsWB) of all 66 lines of both "last" clouds
    const int wb0 = (sct * kCsSW - kCsHalo + kAzBins) % kAzBins;
    const bool wraps = wb0 + kCsWB > kAzBins;
    const int *tgG[2] = { b.lb_start + (size_t)(l * 2 + 0) * (kLineKeys + 1), b.lb_start + (size_t)(l * 2 + 1) * (kLineKeys + 1) };
    const float4 *ptsG[2] = { b.lbc_pts + (size_t)l * kMaxLessSharp, b.lbs_pts + b.off[l] };
    for (int xx = tid; xx < 2 * 66; xx += kCsT) {
        const int tc = xx / 66, v = xx % 66;
        L.elev[tc][v] = b.lb_elev[(size_t)(l * 2 + tc) * 66 + v];
        L.fge[tc][v] = b.line_first_ge[(size_t)(l * 2 + tc) * 66 + v];
        L.lle[tc][v] = b.line_last_le[(size_t)(l * 2 + tc) * 66 + v];
    }
    // (1) the window's slice of the two start tables, raw, into LDS: every address is pure index arithmetic, so a thread's ~16 loads are
    // in flight together (one round trip).  Per row: entries of the kCsWB + 1 window bins (wrapped past bin 384), then the row's end
    // (bin 384) and start (bin 0) for windows that wrap.  The raw slice lives where the per-feature slots and the pool will be.
    constexpr int kRaw = kCsWB + 3;
    int *raw = reinterpret_cast<int *>(L.q);
    static_assert(sizeof(int) * 2 * 66 * kRaw <= sizeof(L.q) + sizeof(L.best) + sizeof(L.same) + sizeof(L.other) + sizeof(L.closest) + sizeof(L.wlo) + sizeof(L.whi) + sizeof(L.pool),
                  "the raw table slice must fit the area it borrows");
    for (int e = tid; e < 2 * 66 * kRaw; e += kCsT) {
        const int r = e / kRaw, lb = e % kRaw, tc = r / 66, v = r % 66;
        const int gb = lb <= kCsWB ? wb0 + lb : (lb == kCsWB + 1 ? kAzBins : 0);
        raw[e] = tgG[tc][v * kAzBins + (gb <= kAzBins ? gb : gb - kAzBins)];
    }
    if (tid == 0) L.ovf = 0;
    __syncthreads();
    // (2) points of every row inside the window, and their exclusive prefix per cloud
    for (int xx = tid; xx < 2 * 66; xx += kCsT) {
        const int *rw = raw + xx * kRaw;
        L.row_len[xx] = wraps ? (rw[kCsWB + 1] - rw[0]) + (rw[kCsWB] - rw[kCsWB + 2]) : rw[kCsWB] - rw[0];
    }
    __syncthreads();
    if (wave == 0) {
        for (int tc = 0; tc < 2; tc++) {
            const int a = L.row_len[tc * 66 + lane], a2 = lane + 64 < 66 ? L.row_len[tc * 66 + 64 + lane] : 0;
            const int inc = wave_scan_incl(a);
            const int tot = __shfl(inc, 63);
            const int inc2 = wave_scan_incl(a2);
            __builtin_amdgcn_wave_barrier();
            L.row_len[tc * 66 + lane] = inc - a;
            if (lane + 64 < 66) L.row_len[tc * 66 + 64 + lane] = tot + inc2 - a2;
            const int total = tot + __shfl(inc2, 63);
            if (lane == 0) { L.row_len[132 + tc] = total; if (total > (tc ? kCsCapS : kCsCapC)) L.ovf = 1; }
        }
    }
    __syncthreads();
    if (L.ovf) {
        for (int i = tid; i < n_f; i += kCsT) defer(flist[i]);        // a window denser than the staging buffers (not seen on S1)
        return;
    }
    // (3) local offset of every (cloud, line, window bin)
    for (int e = tid; e < 2 * 66 * (kCsWB + 1); e += kCsT) {
        const int r = e / (kCsWB + 1), lb = e % (kCsWB + 1), tc = r / 66, v = r % 66;
        const int *rw = raw + r * kRaw;
        const int offp = wb0 + lb <= kAzBins ? rw[lb] - rw[0] : (rw[kCsWB + 1] - rw[0]) + (rw[lb] - rw[kCsWB + 2]);
        L.tg[tc][v][lb] = (unsigned short)(L.row_len[r] + offp);
    }
    // (4) the points, flattened over both clouds: staged point i of a cloud belongs to the row whose prefix holds it (binary search in
    // LDS), its source address follows from the raw slice -- a thread's ~10 loads are independent of each other
    {
        const int n_c = L.row_len[132], n_s = L.row_len[133];
        for (int i0 = tid; i0 < n_c + n_s; i0 += 4 * kCsT) {
            float4 v4[4];
            int dsti[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = i0 + u * kCsT;
                dsti[u] = -1;
                if (i < n_c + n_s) {
                    const int tc = i >= n_c ? 1 : 0, li = i - (tc ? n_c : 0);
                    const int *pre = L.row_len + tc * 66;
                    int lo = 0, hi = 65;
                    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (pre[mid] <= li) lo = mid; else hi = mid - 1; }
                    const int *rw = raw + (tc * 66 + lo) * kRaw;
                    const int within = li - pre[lo];
                    const int nA = (wraps ? rw[kCsWB + 1] : rw[kCsWB]) - rw[0];
                    const int src = within < nA ? rw[0] + within : rw[kCsWB + 2] + (within - nA);
                    v4[u] = ptsG[tc][src];
                    dsti[u] = tc ? kCsCapC + li : li;             // pts_c and pts_s are contiguous: one index space
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) if (dsti[u] >= 0) (dsti[u] >= kCsCapC ? L.pts_s[dsti[u] - kCsCapC] : L.pts_c[dsti[u]]) = v4[u];
        }
    }
    __syncthreads();

    int *seed_c = o.seed ? o.seed + (size_t)c * kMaxQueries : nullptr;
    const double *x = o.state + c * 8;
    for (int f0 = 0; f0 < n_f; f0 += kCsF) {
        const int nq = n_sharp + b.feat_n[k * 4 + 2];
        int qi = (tid < kCsF && f0 + tid < n_f) ? (int)flist[f0 + tid] : -1;
        if (qi >= nq) { qi = -1; if (dbg) atomicAdd(&dbg[3], 1ull); }      // (a list entry can never exceed the pair's feature count)
        const bool have = qi >= 0;
        const bool edge = have && qi < n_sharp;
        const int cl = edge ? 0 : 1;
        float4 fp = make_float4(0.f, 0.f, 0.f, 0.f);
        int sidx = -1;
        if (have) {
            fp = edge ? b.sharp[(size_t)k * kMaxSharp + qi] : b.flat[(size_t)k * kMaxFlat + (qi - n_sharp)];
            if (outer == 1 && seed_c) sidx = seed_c[qi];
        }
        int4 prev = make_int4(-1, -1, -1, 0);
        if (have && outer == 1 && seed_c && sidx >= 0) prev = ((const int4 *)o.corr + (size_t)c * kMaxQueries)[qi];
        const int n_last = b.feat_n[l * 4 + (edge ? 1 : 3)];
        const float4 *cloud = edge ? b.less_sharp + (size_t)l * kMaxLessSharp : b.less_flat + b.off[l];
        float4 sp = make_float4(0.f, 0.f, 0.f, 0.f);
        if (sidx >= 0 && sidx < n_last) sp = cloud[sidx]; else sidx = -1;
        double rx, ry, rz;
        quat_rotate(x, (double)fp.x, (double)fp.y, (double)fp.z, rx, ry, rz);
        const float qx = (float)(rx + x[4]), qy = (float)(ry + x[5]), qz = (float)(rz + x[6]);
        float sd = -1.0f;
        if (sidx >= 0) { const float d = dist2f(sp.x, sp.y, sp.z, qx, qy, qz); if (d < 24.0f) sd = d; }
        __syncthreads();                           // the previous batch has read its per-feature slots
        if (tid < kCsF) { L.q[tid] = make_float4(qx, qy, qz, 0.f); L.best[tid] = ~0ull; }
        const float rho2 = qx * qx + qy * qy, rho = sqrtf(rho2), R = sqrtf(rho2 + qz * qz);
        const float th = atan2f(qy, qx) + 3.14159265f;
        const float eq = elev_of(qx, qy, qz);
        const unsigned char tag_cl = edge ? 0 : 0x80;
        const int ent0 = cl * 66 * (kCsWB + 1);
        bool alive = have && n_last > 0;
        bool deferred = false;
        if (defer_every > 0 && have && qi % defer_every == 0) { alive = false; deferred = true; }
        float r = sd >= 0.f ? sqrtf(sd) * 1.0005f + 1e-3f : (edge ? kCfR0Edge : kCfR0Plane);
        __syncthreads();
        // ================= nearest point =================
        for (int round = 0; round < 64; round++) {
            if (tid == 0) L.n_pool = 0;
            __syncthreads();
            const float rr = fminf(r, 5.0f);
            bool posted = false;
            if (alive) {
                const float4 *el = L.elev[cl];
                CfArc a;
                cf_arc(rr, rho, th, a);
                int ls, le;
                if (!cs_local_arc(a, wb0, ls, le)) { alive = false; deferred = true; }
                else {
                    const float beta = R > rr ? asin_upper(rr / R) + 5e-4f : 4.0f;
                    const float elo = eq - beta, ehi = eq + beta;
                    const int v1 = cf_first_line(el, ehi), v2 = cf_last_line(el, elo);
                    int nreq = 0;
                    for (int v = v1; v <= v2; v++) { const float4 ev = el[v]; nreq += !(ev.y < elo || ev.x > ehi) ? 1 : 0; }
                    if (nreq > kCsPool) { alive = false; deferred = true; }
                    else {
                        int slot = nreq > 0 ? atomicAdd(&L.n_pool, nreq) : 0;
                        if (slot + nreq <= kCsPool) {
                            posted = true;
                            for (int v = v1; v <= v2; v++) {
                                const float4 ev = el[v];
                                if (ev.y < elo || ev.x > ehi) continue;
                                CfRun rq;
                                rq.start = (unsigned int)(ent0 + v * (kCsWB + 1) + ls); rq.pre = (unsigned int)(ent0 + v * (kCsWB + 1) + le);
                                rq.len = 0; rq.owner = (unsigned char)tid; rq.tag = (unsigned char)(tag_cl | v);
                                L.pool[slot++] = rq;
                            }
                        } else {
                            for (; slot < kCsPool; slot++) { CfRun rq; rq.start = 0; rq.pre = 0; rq.len = 0; rq.owner = (unsigned char)tid; rq.tag = 0; L.pool[slot] = rq; }
                        }
                    }
                }
            }
            __syncthreads();
            cs_sweep<false>(L);
            __syncthreads();
            if (alive && posted) {
                const unsigned long long best = L.best[tid];
                if (best != ~0ull) {
                    const float bd = __uint_as_float((unsigned int)(best >> 32));
                    if (bd <= (rr * 0.9999f) * (rr * 0.9999f) || rr >= 5.0f) alive = false;
                    else r = sqrtf(bd) * 1.0005f + 1e-3f;
                } else {
                    if (rr >= 5.0f) alive = false;
                    else r = rr * 2.5f;
                }
            }
            if (!__syncthreads_or(alive ? 1 : 0)) break;
        }
        if (alive) { alive = false; deferred = true; }

        // ================= scan-line walk =================
        const unsigned long long nn = tid < kCsF ? L.best[tid] : ~0ull;
        const unsigned long long thr = pack_fu(25.0f, 0u);
        bool walking = have && n_last > 0 && !deferred && nn != ~0ull && (double)__uint_as_float((unsigned int)(nn >> 32)) < 25.0;
        const int closest = (int)((unsigned int)(nn & 0xffffffffull) >> 7);
        const int ra = (int)(nn & 127ull);
        if (walking) {
            L.closest[tid] = closest;
            L.wlo[tid] = ra - 3 >= 0 ? L.lle[cl][ra - 3] + 1 : 0;
            L.whi[tid] = ra + 3 <= 65 ? L.fge[cl][ra + 3] : n_last;
        }
        const float rad[4] = { walk_radius(0, rho), walk_radius(1, rho), walk_radius(2, rho), walk_radius(3, rho) };
        int wpass = 0;
        unsigned long long same = thr, other = thr;
        float r_seed = -1.0f;
        if (walking && prev.w != 0 && prev.x == closest) {
            const int i_o = edge ? prev.y : prev.z;
            const float4 po = cloud[i_o];
            float d = dist2f(po.x, po.y, po.z, qx, qy, qz);
            if (!edge) { const float4 ps = cloud[prev.y]; d = fmaxf(d, dist2f(ps.x, ps.y, ps.z, qx, qy, qz)); }
            if (d < 24.0f) r_seed = sqrtf(d) * 1.002f + 1e-3f;
        }
        for (int round = 0; round < 64; round++) {
            if (tid == 0) L.n_pool = 0;
            __syncthreads();
            bool posted = false;
            if (walking) {
                while (wpass > 0 && wpass < 4 && rad[wpass] <= rad[wpass - 1]) wpass++;
                if (wpass >= 4) walking = false;
            }
            const bool seeded = r_seed > 0.0f;
            const float r_now = seeded ? r_seed : rad[wpass < 4 ? wpass : 3];
            if (walking) {
                CfArc a;
                cf_arc(r_now, rho, th, a);
                int ls, le;
                if (!cs_local_arc(a, wb0, ls, le)) { walking = false; deferred = true; }
                else {
                    int nreq = 0;
#pragma unroll
                    for (int j = 0; j < 5; j++) { const int v = ra - 2 + j; nreq += (v >= 0 && v <= 65 && !(edge && j == 2)) ? 1 : 0; }
                    int slot = nreq > 0 ? atomicAdd(&L.n_pool, nreq) : 0;
                    if (slot + nreq <= kCsPool) {
                        posted = true;
                        L.same[tid] = thr; L.other[tid] = thr;
#pragma unroll
                        for (int j = 0; j < 5; j++) {
                            const int v = ra - 2 + j;
                            if (!(v >= 0 && v <= 65 && !(edge && j == 2))) continue;
                            CfRun rq;
                            rq.start = (unsigned int)(ent0 + v * (kCsWB + 1) + ls); rq.pre = (unsigned int)(ent0 + v * (kCsWB + 1) + le);
                            rq.len = 0; rq.owner = (unsigned char)tid; rq.tag = (unsigned char)(tag_cl | j);
                            L.pool[slot++] = rq;
                        }
                    } else
                        for (; slot < kCsPool; slot++) { CfRun rq; rq.start = 0; rq.pre = 0; rq.len = 0; rq.owner = (unsigned char)tid; rq.tag = 0; L.pool[slot] = rq; }
                }
            }
            __syncthreads();
            cs_sweep<true>(L);
            __syncthreads();
            if (walking && posted) {
                same = L.same[tid]; other = L.other[tid];
                if (!seeded && r_now >= 5.0f) walking = false;
                else {
                    const unsigned long long lim = pack_fu(r_now * r_now * 0.998f, 0u);
                    if (other < lim && (edge || same < lim)) walking = false;
                    else if (seeded) r_seed = -1.0f;
                    else wpass++;
                }
            }
            if (!__syncthreads_or(walking ? 1 : 0)) break;
        }
        if (have) {
            if (deferred || walking) defer(qi);
            else {
                int4 rres = make_int4(-1, -1, -1, 0);
                int closest_out = -1;
                if (n_last > 0 && nn != ~0ull && (double)__uint_as_float((unsigned int)(nn >> 32)) < 25.0) {
                    closest_out = closest;
                    const int i_other = other < thr ? seq_to_index((unsigned int)(other & 0xffffffffull), closest) : -1;
                    if (edge) { if (i_other >= 0) rres = make_int4(closest, i_other, -1, 1); }
                    else {
                        const int i_same = same < thr ? seq_to_index((unsigned int)(same & 0xffffffffull), closest) : -1;
                        if (i_same >= 0 && i_other >= 0) rres = make_int4(closest, i_same, i_other, 2);
                    }
                }
                ((int4 *)o.corr + (size_t)c * kMaxQueries)[qi] = rres;
                if (outer == 0 && seed_c) seed_c[qi] = closest_out;
                float4 A = make_float4(0.f, 0.f, 0.f, 0.f), B = A, C = A;
                if (rres.w != 0) { A = cloud[rres.x]; B = cloud[rres.y]; if (rres.z >= 0) C = cloud[rres.z]; }
                fp.w = __int_as_float(rres.w);
                float4 *rec = o.crec + ((size_t)c * kMaxQueries + qi) * 4;
                rec[0] = fp; rec[1] = A; rec[2] = B; rec[3] = C;
            }
        }
    }
}

} // namespace lmono
