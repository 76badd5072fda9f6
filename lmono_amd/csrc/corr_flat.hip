// lmono_amd/csrc/corr_flat.hip -- laserOdometry correspondence search as FLATTENED candidate sweeps.
//
// Same results as k_correspond (odometry.hip): the exact nearest point (float distance, lowest index on ties) of every de-skewed
// feature point in the previous scan's less-sharp / less-flat cloud, then the reference's scan-line walk.
//
// Measured (profiles/r2/NOTES.md): giving a feature point 32 lanes (k_correspond), 16 lanes over an LDS tile (k_corr_tile) or 4
// lanes all cost ~440 vector instructions per feature -- the lanes idle while a few dozen candidates are visited in short, ragged
// runs -- and one lane per feature with its own loops (k_corr_thread) is bound by chains of dependent loads and by the slowest
// lane.  Here the ragged work of 256 feature points is flattened through LDS, so that every lane always has a candidate:
//
//   round   1a  each feature's lane (the "owner") turns its search ball into run REQUESTS: one per scan line of the elevation
//               window, (row, first bin, last bin) of the (line, azimuth bin)-sorted copy of the cloud (k_line_index)
//           1b  the workgroup resolves all requests together (every lane 8 requests: 16 independent table loads in flight) into
//               runs (start, length) and takes the exclusive prefix of the lengths
//           2   the candidates of ALL runs are one index space [0, T): every lane takes a contiguous chunk of T / 256 candidates
//               (8 independent 16-B gathers in flight), keeps the running minimum of the feature its chunk is in and posts it with
//               one LDS atomic min per feature it touches.  A feature with 2000 candidates is spread over 40 lanes; one with 20
//               shares a lane with its neighbours
//           3   the owners read their minimum: settled (d <= r), or the radius grows and the feature joins the next round
//   then the same machinery runs the scan-line walk (lines ra-2 .. ra+2, growing arcs, two minima per feature).
//
// Exactness: a point p with |p - q| <= r lies within asin(r / rho_xy(q)) of q's azimuth and within asin(r / |q|) of q's elevation
// angle; lb_elev holds every line's elevation range and its monotone envelopes, so the lines a ball can meet are an interval found
// by two binary searches, each line tested against its own range.  A search of radius r is exact when its minimum is <= r.
// Features whose requests do not fit the round's pool are served in the next round; a feature whose single ball needs more runs
// than the whole pool goes to the device work list of k_correspond_list.
#include "batch.hpp"

namespace lmono {

#ifndef LMONO_CF_T
#define LMONO_CF_T 128
#endif
constexpr int kCfT = LMONO_CF_T;               // threads per workgroup = feature points per workgroup
constexpr int kCfBlocks = kMaxQueries / kCfT; // workgroups per chain
static_assert(kMaxQueries % kCfT == 0, "feature capacity must be a multiple of the workgroup size");
static_assert(kCfBlocks == kThinBlocks, "k_lm_solve skips the records of the workgroups a thinned lead-in pair does not run");
#ifndef LMONO_CF_PER
#define LMONO_CF_PER 10         // after the gather fix of round 3: 6 / 7 / 8 / 10 / 12 / 14 runs per lane -> 42.36 / 42.23 / 41.89 / 41.57 / 42.07 / 44.62 ms per pass
#endif
#ifndef LMONO_CF_U
#define LMONO_CF_U 4
#endif
#ifndef LMONO_CF_WAVES
#define LMONO_CF_WAVES 1
#endif
constexpr int kCfPer = LMONO_CF_PER;           // runs per lane and round
constexpr int kCfPool = kCfT * kCfPer;        // run descriptors per round
// first search radius of an unseeded feature (m): the less-sharp cloud is sparse (<= 20 points per ring and sector), the 0.2 m-voxelised
// less-flat cloud dense.  Any value is exact; measured per bench step: (0.3, 0.3) 49.0 ms, (0.5, 0.3) 48.8, (0.5, 0.2) 48.1, (0.5, 0.15) 47.8,
// (0.5, 0.1) 48.4, (0.8, 0.3) 49.0; a term proportional to the range did not help.
#ifndef LMONO_R0_PLANE
#define LMONO_R0_PLANE 0.15f
#define LMONO_R0_EDGE 0.5f
#endif
constexpr float kCfR0Edge = LMONO_R0_EDGE, kCfR0Plane = LMONO_R0_PLANE;
constexpr int kCfU = LMONO_CF_U;               // gathers in flight per lane

struct CfRun {
    unsigned int start;                       // request: table entry of the first bin; resolved: first point of the run
    unsigned int pre;                         // request: table entry behind the last bin; resolved: candidates before this run
    unsigned short len;
    unsigned char owner;                      // feature (lane) the run belongs to (kCfT <= 256)
    unsigned char tag;                        // bit 7: surf cloud; low bits: scan line (nearest point) / line offset 0..4 (walk)
};
static_assert(sizeof(CfRun) == 12, "run descriptor layout");

struct CfLds {
    float4 elev[2][66];                       // lb_elev of the two "last" clouds
    int fge[2][66], lle[2][66];
    float4 q[kCfT];                           // de-skewed feature points
    unsigned long long best[kCfT];            // nearest point: (d2 bits) << 32 | index << 7 | line
    unsigned long long same[kCfT], other[kCfT];
    int closest[kCfT], wlo[kCfT], whi[kCfT], ra[kCfT];
    CfRun pool[kCfPool];
    int n_pool, n_cand, wsum[kCfT / 64];
};

__device__ __forceinline__ void cf_defer(unsigned int *wl, int c, int qi)
{
    const unsigned int slot = atomicAdd(wl, 1u);
    wl[1 + slot] = ((unsigned int)c << 12) | (unsigned int)qi;
}

// azimuth arc of radius r around the feature as one or two bin ranges of a table row: [a0, a1) and [0, w1) (w1 = 0: no wrap)
struct CfArc { int a0, a1, w1; };
__device__ __forceinline__ void cf_arc(float r, float rho, float th, CfArc &a)
{
    constexpr float kb = kAzBins / 6.28318531f;
    a.a0 = 0; a.a1 = kAzBins; a.w1 = 0;
    if (!(rho > r * 1.002f)) return;                      // the ball reaches the sensor axis: every azimuth
    const float alpha = asin_upper(r / rho) + kArcSlackBins / kb;
    const int lo = (int)floorf((th - alpha) * kb), hi = (int)floorf((th + alpha) * kb);
    const int n = hi - lo + 1;
    if (n >= kAzBins) return;
    const int b0 = ((lo % kAzBins) + kAzBins) % kAzBins;
    a.a0 = b0;
    if (b0 + n <= kAzBins) a.a1 = b0 + n;
    else { a.a1 = kAzBins; a.w1 = b0 + n - kAzBins; }
}

// first line whose envelope A (min of lo over lines <= v, non-increasing) is <= ehi; 66 when none
__device__ __forceinline__ int cf_first_line(const float4 *el, float ehi)
{
    int lo = 0, hi = 66;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (el[mid].z <= ehi) hi = mid; else lo = mid + 1; }
    return lo;
}
// last line whose envelope B (max of hi over lines >= v, non-increasing) is >= elo; -1 when none
__device__ __forceinline__ int cf_last_line(const float4 *el, float elo)
{
    int lo = -1, hi = 65;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (el[mid].w >= elo) lo = mid; else hi = mid - 1; }
    return lo;
}

// stages 1b and 2 of a round, executed by the whole workgroup.  kWalk = false: nearest point (minimum into L.best);
// kWalk = true: scan-line walk (minima into L.same / L.other).
#ifdef LMONO_TILE_PROF
#define CF_STAMP(v) { if (threadIdx.x == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); (v) += t_ - cf_t; cf_t = t_; } }
#else
#define CF_STAMP(v)
#endif

template <bool kWalk>
__device__ __forceinline__ void cf_sweep(CfLds &L, const int *tg_c, const int *tg_s, const float4 *pts_c, const float4 *pts_s, unsigned long long &cf_t, unsigned long long *cf_acc)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_pool = min(L.n_pool, kCfPool);
    // ---- 1b: resolve this lane's kCfPer consecutive requests (all table loads in flight together), prefix of the lengths
    unsigned int st[kCfPer], en[kCfPer];
#pragma unroll
    for (int j = 0; j < kCfPer; j++) {
        const int i = tid * kCfPer + j;
        st[j] = 0; en[j] = 0;
        if (i < n_pool) {
            const CfRun rq = L.pool[i];
            const int *tg = (rq.tag & 0x80) ? tg_s : tg_c;
            st[j] = (unsigned int)tg[rq.start]; en[j] = (unsigned int)tg[rq.pre];
        }
    }
    int sum = 0;
#pragma unroll
    for (int j = 0; j < kCfPer; j++) sum += (int)(en[j] - st[j]);
    const int incl = wave_scan_incl(sum);
    if (lane == 63) L.wsum[wave] = incl;
    __syncthreads();
    int run = incl - sum;
    for (int w = 0; w < wave; w++) run += L.wsum[w];
    if (tid == kCfT - 1) L.n_cand = run + sum;
#pragma unroll
    for (int j = 0; j < kCfPer; j++) {
        const int i = tid * kCfPer + j;
        if (i < n_pool) {
            L.pool[i].start = st[j];
            L.pool[i].pre = (unsigned int)run;
            L.pool[i].len = (unsigned short)min(en[j] - st[j], 65535u);
        }
        run += (int)(en[j] - st[j]);
    }
    __syncthreads();
    CF_STAMP(cf_acc[1])
    // ---- 2: this lane's contiguous chunk of the candidate index space
    const int T = L.n_cand;
    if (T <= 0 || n_pool <= 0) return;
    const int ch = (T + kCfT - 1) / kCfT;
    int j0 = tid * ch;
    const int j1 = min(j0 + ch, T);
    if (j0 >= j1) return;
    // run that holds candidate j0: last run with pre <= j0 (runs of length 0 share their pre with the next run)
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
    unsigned long long m0 = ~0ull, m1 = ~0ull;      // running minima of the current owner (nearest / same, other)
    int closest = 0, w_lo = 0, w_hi = 0;
    auto flush = [&]() {
        if (owner < 0) return;
        if (!kWalk) { if (m0 != ~0ull) atomicMin(&L.best[owner], m0); }
        else { if (m0 != ~0ull) atomicMin(&L.same[owner], m0); if (m1 != ~0ull) atomicMin(&L.other[owner], m1); }
    };
    while (j0 < j1) {
        // up to kCfU candidates: addresses first (LDS only), then the gathers together, then the arithmetic
        unsigned int addr[kCfU];
        unsigned short meta[kCfU];            // owner << 8 | tag
#pragma unroll
        for (int u = 0; u < kCfU; u++) {
            addr[u] = 0xffffffffu; meta[u] = 0;
            if (j0 + u < j1) {
                while (off >= (int)cur.len) { off -= (int)cur.len; seg++; cur = L.pool[seg]; }
                addr[u] = cur.start + (unsigned int)off;
                meta[u] = (unsigned short)(((unsigned int)cur.owner << 8) | cur.tag);
                off++;
            }
        }
        // UNCONDITIONAL gathers (an unused slot reads entry 0 of the corner copy, which always exists): behind a branch the compiler sank the first
        // use of every point into its load's block and waited for each gather in turn (s_waitcnt vmcnt(0) after every global_load_dwordx4 --
        // rounds 2 and 3 ran with ONE gather in flight per lane whatever kCfU said)
        float4 p[kCfU];
#pragma unroll
        for (int u = 0; u < kCfU; u++) {
            const bool live = addr[u] != 0xffffffffu;
            const float4 *src = (live && (meta[u] & 0x80)) ? pts_s : pts_c;
            p[u] = src[live ? addr[u] : 0u];
        }
        __builtin_amdgcn_sched_barrier(0);          // the arithmetic below stays below the loads
#pragma unroll
        for (int u = 0; u < kCfU; u++) {
            if (addr[u] == 0xffffffffu) continue;
            const int ow = meta[u] >> 8, tag = meta[u] & 0x7f;
            if (ow != owner) {
                flush();
                owner = ow; m0 = ~0ull; m1 = ~0ull;
                const float4 qq = L.q[ow];
                qx = qq.x; qy = qq.y; qz = qq.z;
                if (kWalk) { closest = L.closest[ow]; w_lo = L.wlo[ow]; w_hi = L.whi[ow]; }
            }
            const float d = dist2f(p[u].x, p[u].y, p[u].z, qx, qy, qz);
            if (!kWalk) {
                const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned int)((__float_as_int(p[u].w) << 7) | tag);
                m0 = key < m0 ? key : m0;
            } else {
                const int jj = __float_as_int(p[u].w);
                if (jj == closest || jj < w_lo || jj >= w_hi) continue;
                const bool fwd = jj > closest;
                const unsigned int seq = fwd ? (unsigned int)(jj - closest - 1) : kSeqBack + (unsigned int)(closest - 1 - jj);
                const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | seq;
                const bool is_other = fwd ? (tag > 2) : (tag < 2);          // tag = line offset 0..4, 2 = the nearest point's own line
                if (is_other) m1 = key < m1 ? key : m1; else m0 = key < m0 ? key : m0;
            }
        }
        j0 += kCfU;
    }
    flush();
}

// step `step`, outer iteration `outer` of every chain: kCfBlocks workgroups of kCfT feature points per chain (measured: 256 threads
// 30.0 ms of correspondence search per bench step, 128 threads 26.6, 64 threads 31.4; 4 or 8 gathers in flight make no difference), decoded onto ONE XCD
// per chain (blocks b and b + 8 share an XCD): the chain's index and tables are fetched into one L2 only.
// list_mode != 0: the workgroups serve the chain's deferred list (o.dl / o.dl_cnt, left by the sector-staged search of corr_sect.hip)
// instead of dealing all feature points: workgroup qb takes entries qb, qb + kCfBlocks, ...
__global__ __launch_bounds__(kCfT, LMONO_CF_WAVES) void k_corr_flat(BatchView b, OdomView o, int step, int outer, unsigned int *wl, int defer_every, unsigned long long *stats, int list_mode)
{
    __shared__ CfLds L;
    const int xcd = blockIdx.x & 7, u = blockIdx.x >> 3;
    const int ci = o.chain0 + (u / kCfBlocks) * 8 + xcd;
    const int qb = u % kCfBlocks;
    if (ci >= o.chain1) return;
    const int c = o.clist ? o.clist[ci] : ci;
    int own;
    const int k = chain_scan(o, c, step, own);
    if (k < 0) return;
    if (!list_mode && lead_in_thinned(o, k, own) && qb % kThinStride != 0) return;      // early lead-in pair: every kThinStride-th share of the features
    const int tid = threadIdx.x;
    const int l = k - 1;
    const int n_sharp = b.feat_n[k * 4 + 0];
    const int nq = n_sharp + b.feat_n[k * 4 + 2];
    int n_list = 0;
    if (list_mode) { n_list = o.dl_cnt[c]; if (qb == 0 && tid == 0 && stats && n_list) atomicAdd(&stats[1], (unsigned long long)n_list); if (qb >= n_list) return; }
    else if (qb >= nq) return;
    // the chain's features are dealt round-robin over its kCfBlocks workgroups: every workgroup gets the same share of edge and plane
    // features (blocks of consecutive features gave workgroups of very different weight, and an almost empty last one)
    // (list mode: the deferred features are the expensive ones -- wide balls -- so they are dealt round-robin over the workgroups as well)
    const int qi = list_mode ? (tid * kCfBlocks + qb < n_list ? (int)o.dl[(size_t)c * kMaxQueries + tid * kCfBlocks + qb] : nq) : tid * kCfBlocks + qb;
    if (b.status[l] & (kStatusIrregularLines | kStatusDenseCell)) {
        if (qi < nq) cf_defer(wl, c, qi);       // rare: the whole scan pair goes to the generic search
        return;
    }
    // the feature point and its seed are requested before the small tables are staged
    const bool edge = qi < n_sharp;
    const int cl = edge ? 0 : 1;
    float4 fp = make_float4(0.f, 0.f, 0.f, 0.f);
    int *seed_c = o.seed ? o.seed + (size_t)c * kMaxQueries : nullptr;
    int sidx = -1;
    if (qi < nq) {
        fp = edge ? b.sharp[(size_t)k * kMaxSharp + qi] : b.flat[(size_t)k * kMaxFlat + (qi - n_sharp)];
        if (outer == 1 && seed_c) sidx = seed_c[qi];
    }
    // second outer iteration: the partners the walk found in the first one bound its radius (below)
    int4 prev = make_int4(-1, -1, -1, 0);
    if (qi < nq && outer == 1 && seed_c && sidx >= 0) prev = ((const int4 *)o.corr + (size_t)c * kMaxQueries)[qi];
    for (int xx = tid; xx < 2 * 66; xx += kCfT) {
        const int tc = xx / 66, v = xx % 66;
        L.elev[tc][v] = b.lb_elev[(size_t)(l * 2 + tc) * 66 + v];
        L.fge[tc][v] = b.line_first_ge[(size_t)(l * 2 + tc) * 66 + v];
        L.lle[tc][v] = b.line_last_le[(size_t)(l * 2 + tc) * 66 + v];
    }
    const int n_last = b.feat_n[l * 4 + (edge ? 1 : 3)];
    const float4 *cloud = edge ? b.less_sharp + (size_t)l * kMaxLessSharp : b.less_flat + b.off[l];
    float4 sp = make_float4(0.f, 0.f, 0.f, 0.f);
    if (sidx >= 0 && sidx < n_last) sp = cloud[sidx]; else sidx = -1;
    // de-skew transform in fp64 as the reference's TransformToStart
    const double *x = o.state + c * 8;
    double rx, ry, rz;
    quat_rotate(x, (double)fp.x, (double)fp.y, (double)fp.z, rx, ry, rz);
    const float qx = (float)(rx + x[4]), qy = (float)(ry + x[5]), qz = (float)(rz + x[6]);
    float sd = -1.0f;
    if (sidx >= 0) { const float d = dist2f(sp.x, sp.y, sp.z, qx, qy, qz); if (d < 24.0f) sd = d; }
    L.q[tid] = make_float4(qx, qy, qz, 0.f);
    L.best[tid] = ~0ull;
    const int *tg_c = b.lb_start + (size_t)(l * 2 + 0) * (kLineKeys + 1), *tg_s = b.lb_start + (size_t)(l * 2 + 1) * (kLineKeys + 1);
    const float4 *pts_c = b.lbc_pts + (size_t)l * kMaxLessSharp, *pts_s = b.lbs_pts + b.off[l];
    const float rho2 = qx * qx + qy * qy, rho = sqrtf(rho2), R = sqrtf(rho2 + qz * qz);
    const float th = atan2f(qy, qx) + 3.14159265f;
    const float eq = elev_of(qx, qy, qz);
    const unsigned char tag_cl = edge ? 0 : 0x80;
    unsigned long long cf_t = 0, cf_acc[10] = { 0 };      // diagnostic build: cycles per stage (1a, 1b, 2, 3, setup, epilogue), rounds, candidates
#ifdef LMONO_TILE_PROF
    if (tid == 0) cf_t = __builtin_amdgcn_s_memtime();
#endif
    bool alive = qi < nq && n_last > 0;       // still looking for its nearest point
    bool deferred = false;
    if (!list_mode && defer_every > 0 && qi < nq && qi % defer_every == 0) { alive = false; deferred = true; }      // test hook: exercise the fall-back kernel
    float r = sd >= 0.f ? sqrtf(sd) * 1.0005f + 1e-3f : (edge ? kCfR0Edge : kCfR0Plane);
    __syncthreads();

    CF_STAMP(cf_acc[4])
    // ================= nearest point =================
    for (int round = 0; round < 64; round++) {
        if (tid == 0) L.n_pool = 0;
        __syncthreads();
        // ---- 1a: run requests of the features still searching
        const float rr = fminf(r, 5.0f);                    // d2 < 25 means d < 5: a 5 m ball holds every admissible point
        bool posted = false;
        if (alive) {
            const float4 *el = L.elev[cl];
            CfArc a;
            cf_arc(rr, rho, th, a);
            const float beta = R > rr ? asin_upper(rr / R) + 5e-4f : 4.0f;
            const float elo = eq - beta, ehi = eq + beta;
            const int v1 = cf_first_line(el, ehi), v2 = cf_last_line(el, elo);
            int nl = 0;
            for (int v = v1; v <= v2; v++) { const float4 ev = el[v]; nl += !(ev.y < elo || ev.x > ehi) ? 1 : 0; }
            const int nreq = nl * (a.w1 ? 2 : 1);
            if (nreq > kCfPool) { alive = false; deferred = true; }       // a single ball larger than the pool: list kernel
            else {
                int slot = nreq > 0 ? atomicAdd(&L.n_pool, nreq) : 0;
                if (slot + nreq <= kCfPool) {
                    posted = true;
                    for (int v = v1; v <= v2; v++) {
                        const float4 ev = el[v];
                        if (ev.y < elo || ev.x > ehi) continue;
                        CfRun rq;
                        rq.start = (unsigned int)(v * kAzBins + a.a0); rq.pre = (unsigned int)(v * kAzBins + a.a1);
                        rq.len = 0; rq.owner = (unsigned char)tid; rq.tag = (unsigned char)(tag_cl | v);
                        L.pool[slot++] = rq;
                        if (a.w1) { rq.start = (unsigned int)(v * kAzBins); rq.pre = (unsigned int)(v * kAzBins + a.w1); L.pool[slot++] = rq; }
                    }
                } else {
                    // the pool of this round is full: the feature posts again in the next round; the part of its reservation that
                    // lies inside the pool becomes empty runs
                    for (; slot < kCfPool; slot++) { CfRun rq; rq.start = 0; rq.pre = 0; rq.len = 0; rq.owner = (unsigned char)tid; rq.tag = 0; L.pool[slot] = rq; }
                }
            }
        }
        __syncthreads();
        CF_STAMP(cf_acc[0])
        cf_sweep<false>(L, tg_c, tg_s, pts_c, pts_s, cf_t, cf_acc);
        __syncthreads();
        CF_STAMP(cf_acc[2])
#ifdef LMONO_TILE_PROF
        if (tid == 0) { cf_acc[6] += 1; cf_acc[8] += (unsigned long long)L.n_cand; }
#endif
        // ---- 3: owners decide
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
        CF_STAMP(cf_acc[3])
        if (!__syncthreads_or(alive ? 1 : 0)) break;
    }
    if (alive) { alive = false; deferred = true; }            // round budget exhausted (never observed): list kernel

    // ================= scan-line walk =================
    const unsigned long long nn = L.best[tid];
    const unsigned long long thr = pack_fu(25.0f, 0u);
    bool walking = qi < nq && n_last > 0 && !deferred && nn != ~0ull && (double)__uint_as_float((unsigned int)(nn >> 32)) < 25.0;
    const int closest = (int)((unsigned int)(nn & 0xffffffffull) >> 7);
    const int ra = (int)(nn & 127ull);
    if (walking) {
        L.closest[tid] = closest;
        L.wlo[tid] = ra - 3 >= 0 ? L.lle[cl][ra - 3] + 1 : 0;
        L.whi[tid] = ra + 3 <= 65 ? L.fge[cl][ra + 3] : n_last;
    }
    // radii: the neighbouring lines right next to the nearest point; the ring gap of far ground points (rho^2 dtheta / h); 5 m
    const float rad[4] = { walk_radius(0, rho), walk_radius(1, rho), walk_radius(2, rho), walk_radius(3, rho) };
    int wpass = 0;
    unsigned long long same = thr, other = thr;
    // Seeded walk: when the nearest point is the one of the first outer iteration, the partners found then are still admissible
    // candidates, so the new minima are no farther than they are: ONE pass with the ball that just holds them is exact (their own
    // keys are strictly inside it).  Otherwise, and if that pass does not settle, the radius ladder below runs as usual.
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
            int nl = 0;
#pragma unroll
            for (int j = 0; j < 5; j++) { const int v = ra - 2 + j; nl += (v >= 0 && v <= 65 && !(edge && j == 2)) ? 1 : 0; }
            const int nreq = nl * (a.w1 ? 2 : 1);
            int slot = nreq > 0 ? atomicAdd(&L.n_pool, nreq) : 0;
            if (slot + nreq <= kCfPool) {
                posted = true;
                L.same[tid] = thr; L.other[tid] = thr;
#pragma unroll
                for (int j = 0; j < 5; j++) {
                    const int v = ra - 2 + j;
                    if (!(v >= 0 && v <= 65 && !(edge && j == 2))) continue;      // edges never use the nearest point's own line
                    CfRun rq;
                    rq.start = (unsigned int)(v * kAzBins + a.a0); rq.pre = (unsigned int)(v * kAzBins + a.a1);
                    rq.len = 0; rq.owner = (unsigned char)tid; rq.tag = (unsigned char)(tag_cl | j);
                    L.pool[slot++] = rq;
                    if (a.w1) { rq.start = (unsigned int)(v * kAzBins); rq.pre = (unsigned int)(v * kAzBins + a.w1); L.pool[slot++] = rq; }
                }
            } else
                for (; slot < kCfPool; slot++) { CfRun rq; rq.start = 0; rq.pre = 0; rq.len = 0; rq.owner = (unsigned char)tid; rq.tag = 0; L.pool[slot] = rq; }
        }
        __syncthreads();
        CF_STAMP(cf_acc[0])
        cf_sweep<true>(L, tg_c, tg_s, pts_c, pts_s, cf_t, cf_acc);
        __syncthreads();
        CF_STAMP(cf_acc[2])
#ifdef LMONO_TILE_PROF
        if (tid == 0) { cf_acc[7] += 1; cf_acc[9] += (unsigned long long)L.n_cand; }
#endif
        if (walking && posted) {
            same = L.same[tid]; other = L.other[tid];
            if (!seeded && r_now >= 5.0f) walking = false;
            else {
                const unsigned long long lim = pack_fu(r_now * r_now * 0.998f, 0u);     // strictly inside the ball of this pass
                if (other < lim && (edge || same < lim)) walking = false;
                else if (seeded) r_seed = -1.0f;          // (not expected) back to the ladder
                else wpass++;
            }
        }
        CF_STAMP(cf_acc[3])
        if (!__syncthreads_or(walking ? 1 : 0)) break;
    }
#ifdef LMONO_TILE_PROF
    if (tid == 0 && stats) { atomicAdd(&stats[1], 1ull); for (int i = 0; i < 10; i++) atomicAdd(&stats[2 + i], cf_acc[i]); }
#endif
    if (qi >= nq) return;
    if (deferred || walking) { cf_defer(wl, c, qi); return; }
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
    // residual-block record for the solver: the feature point and its 2 (edge) or 3 (plane) partners, 64 B
    float4 A = make_float4(0.f, 0.f, 0.f, 0.f), B = A, C = A;
    if (rres.w != 0) { A = cloud[rres.x]; B = cloud[rres.y]; if (rres.z >= 0) C = cloud[rres.z]; }
    fp.w = __int_as_float(rres.w);
    float4 *rec = o.crec + ((size_t)c * kMaxQueries + qi) * 4;
    rec[0] = fp; rec[1] = A; rec[2] = B; rec[3] = C;
}

} // namespace lmono
