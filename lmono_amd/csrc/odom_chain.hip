// lmono_amd/csrc/odom_chain.hip -- laserOdometry of a whole chain in ONE workgroup: the default schedule of lmono_odom_batch[_d].
//
// A chain is a strictly sequential program (per scan pair: 2 x [correspondence search -> <= 4 LM iterations]); chains never need each
// other.  The launch-per-phase schedule (k_corr_flat / k_correspond_list / k_lm_solve, lmono_hip.hip: odom_launch_steps) made every
// chain wait for the slowest one 2 x 3 times per scan pair and ran the solves on a quarter of the CUs.  Here one 1024-thread workgroup
// (16 waves: one CU) owns a chain from its first lead-in pair to its last owned pair -- one launch for all steps, the only
// synchronisation is the workgroup's own barrier:
//
//   search   the pair's feature points are dealt round-robin into wave tasks of <= 64 points; a wave runs k_corr_flat's flattened
//            candidate sweeps on its task entirely by itself (its own LDS slice: pool of run requests, per-feature minima; no workgroup
//            barrier inside the search).  Same arithmetic, same radii, same exactness argument as corr_flat.hip -- identical indices.
//            Features the flat search cannot serve (a ball needing more runs than the pool, scan pairs flagged irregular) go to a
//            per-chain list served by the 32-lane-group / whole-wave searches of odometry.hip right after the tasks.
//   solve    the 1024 threads hold <= 3 residual blocks each in registers (pose-independent part staged once per solve), sweeps reduce
//            through a wave butterfly and 16 LDS partials; the trust-region loop is k_lm_solve's (lm_trust_region, odometry.hip).
//
// Repair launches (o.repair, boundary validation) run the same kernel over the flagged chains: a repair chain stops by itself at the
// first re-computed increment that agrees with the stored one, so the host needs no lock-step chunks.
#include "batch.hpp"

namespace lmono {

constexpr int kOcT = 1024;                       // threads per chain
constexpr int kOcW = kOcT / 64;                  // waves per chain
#ifndef LMONO_OC_PER
#define LMONO_OC_PER 6
#endif
constexpr int kOcPer = LMONO_OC_PER;             // run requests per lane and round (pool of a wave = 64 * kOcPer)
constexpr int kOcPool = 64 * kOcPer;
constexpr int kOcBlocks = (kMaxQueries + kOcT - 1) / kOcT;      // residual blocks per thread in the solve

struct OcWave {
    float4 q[64];                                // de-skewed feature points of the task
    unsigned long long best[64], same[64], other[64];
    int closest[64], wlo[64], whi[64];
    CfRun pool[kOcPool];
    int n_pool, n_cand, pad[2];
};

// lm_trust_region (odometry.hip) as a state machine whose state lives in LDS: wave 0 runs the uniform trust-region arithmetic between
// the sweeps, every wave reads what to evaluate next -- nothing of it stays in registers across a sweep.  Same operations in the same
// order as lm_trust_region, so the iterates are the same numbers.
struct OcTr {
    double x[7], cand[7], scale[6], diag[6];
    double radius, decrease_factor, x_cost, x_norm, model_change;
    int reuse_diagonal, invalid_steps, iter, done, last, pad;
};

// everything the search and the solve need to know about the scan pair being worked on: filled once per pair, read from LDS by the
// out-of-line phase functions (passing the kernel's argument structs by reference would put them -- ~60 pointers -- into scratch)
struct OcPair {
    const float4 *sharp, *flat;                  // the pair's feature points (scan k)
    const float4 *cloud_c, *cloud_s;             // "last" clouds (scan k - 1): less sharp, less flat
    const float4 *pts_c, *pts_s;                 // ... sorted by (line, azimuth bin)
    const int *tg_c, *tg_s;                      // ... their (line, bin) start tables
    int4 *corr; int *seed; float4 *crec;         // the chain's correspondence / seed / residual-block arrays
    unsigned int *defer_list;
    int n_last_c, n_last_s, n_sharp, nq, thin, defer_every, outer, pad;
};

struct OcLds {
    OcPair pair;
    float4 elev[2][66];                          // lb_elev of the two "last" clouds of the pair being worked on
    int fge[2][66], lle[2][66];
    OcWave w[kOcW];
    double red[kOcW][28], sum[28], cur[28];
    double x[8];                                 // the chain's state q(xyzw), t
    OcTr tr;
    int used[kOcW];
    int n_defer, stop;
};
static_assert(sizeof(OcLds) <= 160 * 1024, "OcLds exceeds the 160 KB LDS of a gfx950 CU");

// a pointer read from LDS is a generic one to the compiler; these all point to HBM
template <class T> __device__ __forceinline__ T *oc_g(T *p) { return (T *)__builtin_assume_aligned((void *)p, sizeof(T) >= 16 ? 16 : sizeof(T)); }

__device__ __forceinline__ void oc_wsync()
{
    // the lanes of a wave execute in lock-step and a wave's LDS operations complete in order: only the compiler must not move LDS
    // accesses across this point
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// stages 1b and 2 of a round (corr_flat.hip: cf_sweep) for ONE wave
template <bool kWalk>
__device__ __forceinline__ void oc_sweep(OcWave &L, const int *tg_c, const int *tg_s, const float4 *pts_c, const float4 *pts_s, int lane)
{
    const int n_pool = min(L.n_pool, kOcPool);
    unsigned int st[kOcPer], en[kOcPer];
#pragma unroll
    for (int j = 0; j < kOcPer; j++) {
        const int i = lane * kOcPer + j;
        st[j] = 0; en[j] = 0;
        if (i < n_pool) {
            const CfRun rq = L.pool[i];
            const int *tg = (rq.tag & 0x80) ? tg_s : tg_c;
            st[j] = (unsigned int)tg[rq.start]; en[j] = (unsigned int)tg[rq.pre];
        }
    }
    int sum = 0;
#pragma unroll
    for (int j = 0; j < kOcPer; j++) sum += (int)(en[j] - st[j]);
    const int incl = wave_scan_incl(sum);
    int run = incl - sum;
    const int T = __shfl(incl, 63);
#pragma unroll
    for (int j = 0; j < kOcPer; j++) {
        const int i = lane * kOcPer + j;
        if (i < n_pool) {
            L.pool[i].start = st[j];
            L.pool[i].pre = (unsigned int)run;
            L.pool[i].len = (unsigned short)min(en[j] - st[j], 65535u);
        }
        run += (int)(en[j] - st[j]);
    }
    oc_wsync();
    if (T <= 0 || n_pool <= 0) return;
    const int ch = (T + 63) / 64;
    int j0 = lane * ch;
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
    while (j0 < j1) {
        unsigned int addr[kCfU];
        unsigned short meta[kCfU];
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
        // unconditional gathers, as in cf_sweep (behind a branch the compiler waits for every gather in turn)
        float4 p[kCfU];
#pragma unroll
        for (int u = 0; u < kCfU; u++) {
            const bool live = addr[u] != 0xffffffffu;
            const float4 *src = (live && (meta[u] & 0x80)) ? pts_s : pts_c;
            p[u] = src[live ? addr[u] : 0u];
        }
        __builtin_amdgcn_sched_barrier(0);
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
                const bool is_other = fwd ? (tag > 2) : (tag < 2);
                if (is_other) m1 = key < m1 ? key : m1; else m0 = key < m0 ? key : m0;
            }
        }
        j0 += kCfU;
    }
    flush();
}

// One wave task of the correspondence search: feature qi (-1: idle lane) of scan pair (k - 1, k), outer iteration `outer`, at the
// chain's pose x (LDS).  Restates k_corr_flat's body for a 64-lane group; see corr_flat.hip for the method and its exactness.
__device__ __noinline__ void oc_search_task(OcLds &S, OcWave &L, int qi, int lane)
{
    const OcPair &P = S.pair;
    const int nq = P.nq, n_sharp = P.n_sharp, outer = P.outer, defer_every = P.defer_every;
    unsigned int *defer_list = P.defer_list;
    const bool have = qi >= 0 && qi < nq;
    const bool edge = have && qi < n_sharp;
    const int cl = edge ? 0 : 1;
    float4 fp = make_float4(0.f, 0.f, 0.f, 0.f);
    int *seed_c = (int *)oc_g(P.seed);
    int sidx = -1;
    if (have) {
        fp = edge ? oc_g(P.sharp)[qi] : oc_g(P.flat)[qi - n_sharp];
        if (outer == 1) sidx = seed_c[qi];
    }
    int4 prev = make_int4(-1, -1, -1, 0);
    if (have && outer == 1 && sidx >= 0) prev = oc_g(P.corr)[qi];
    const int n_last = edge ? P.n_last_c : P.n_last_s;
    const float4 *cloud = oc_g(edge ? P.cloud_c : P.cloud_s);
    float4 sp = make_float4(0.f, 0.f, 0.f, 0.f);
    if (sidx >= 0 && sidx < n_last) sp = cloud[sidx]; else sidx = -1;
    const double *x = S.x;
    double rx, ry, rz;
    quat_rotate(x, (double)fp.x, (double)fp.y, (double)fp.z, rx, ry, rz);
    const float qx = (float)(rx + x[4]), qy = (float)(ry + x[5]), qz = (float)(rz + x[6]);
    float sd = -1.0f;
    if (sidx >= 0) { const float d = dist2f(sp.x, sp.y, sp.z, qx, qy, qz); if (d < 24.0f) sd = d; }
    oc_wsync();                                   // the previous task of this wave has read its slice
    L.q[lane] = make_float4(qx, qy, qz, 0.f);
    L.best[lane] = ~0ull;
    const int *tg_c = oc_g(P.tg_c), *tg_s = oc_g(P.tg_s);
    const float4 *pts_c = oc_g(P.pts_c), *pts_s = oc_g(P.pts_s);
    const float rho2 = qx * qx + qy * qy, rho = sqrtf(rho2), R = sqrtf(rho2 + qz * qz);
    const float th = atan2f(qy, qx) + 3.14159265f;
    const float eq = elev_of(qx, qy, qz);
    const unsigned char tag_cl = edge ? 0 : 0x80;
    bool alive = have && n_last > 0;
    bool deferred = false;
    if (defer_every > 0 && have && qi % defer_every == 0) { alive = false; deferred = true; }      // test hook: exercise the fall-back searches
    float r = sd >= 0.f ? sqrtf(sd) * 1.0005f + 1e-3f : (edge ? kCfR0Edge : kCfR0Plane);
    oc_wsync();

    // ================= nearest point =================
    for (int round = 0; round < 64; round++) {
        if (lane == 0) L.n_pool = 0;
        oc_wsync();
        const float rr = fminf(r, 5.0f);
        bool posted = false;
        if (alive) {
            const float4 *el = S.elev[cl];
            CfArc a;
            cf_arc(rr, rho, th, a);
            const float beta = R > rr ? asin_upper(rr / R) + 5e-4f : 4.0f;
            const float elo = eq - beta, ehi = eq + beta;
            const int v1 = cf_first_line(el, ehi), v2 = cf_last_line(el, elo);
            int nl = 0;
            for (int v = v1; v <= v2; v++) { const float4 ev = el[v]; nl += !(ev.y < elo || ev.x > ehi) ? 1 : 0; }
            const int nreq = nl * (a.w1 ? 2 : 1);
            if (nreq > kOcPool) { alive = false; deferred = true; }
            else {
                int slot = nreq > 0 ? atomicAdd(&L.n_pool, nreq) : 0;
                if (slot + nreq <= kOcPool) {
                    posted = true;
                    for (int v = v1; v <= v2; v++) {
                        const float4 ev = el[v];
                        if (ev.y < elo || ev.x > ehi) continue;
                        CfRun rq;
                        rq.start = (unsigned int)(v * kAzBins + a.a0); rq.pre = (unsigned int)(v * kAzBins + a.a1);
                        rq.len = 0; rq.owner = (unsigned char)lane; rq.tag = (unsigned char)(tag_cl | v);
                        L.pool[slot++] = rq;
                        if (a.w1) { rq.start = (unsigned int)(v * kAzBins); rq.pre = (unsigned int)(v * kAzBins + a.w1); L.pool[slot++] = rq; }
                    }
                } else {
                    for (; slot < kOcPool; slot++) { CfRun rq; rq.start = 0; rq.pre = 0; rq.len = 0; rq.owner = (unsigned char)lane; rq.tag = 0; L.pool[slot] = rq; }
                }
            }
        }
        oc_wsync();
        oc_sweep<false>(L, tg_c, tg_s, pts_c, pts_s, lane);
        oc_wsync();
        if (alive && posted) {
            const unsigned long long best = L.best[lane];
            if (best != ~0ull) {
                const float bd = __uint_as_float((unsigned int)(best >> 32));
                if (bd <= (rr * 0.9999f) * (rr * 0.9999f) || rr >= 5.0f) alive = false;
                else r = sqrtf(bd) * 1.0005f + 1e-3f;
            } else {
                if (rr >= 5.0f) alive = false;
                else r = rr * 2.5f;
            }
        }
        if (!__ballot(alive)) break;
    }
    if (alive) { alive = false; deferred = true; }

    // ================= scan-line walk =================
    const unsigned long long nn = L.best[lane];
    const unsigned long long thr = pack_fu(25.0f, 0u);
    bool walking = have && n_last > 0 && !deferred && nn != ~0ull && (double)__uint_as_float((unsigned int)(nn >> 32)) < 25.0;
    const int closest = (int)((unsigned int)(nn & 0xffffffffull) >> 7);
    const int ra = (int)(nn & 127ull);
    if (walking) {
        L.closest[lane] = closest;
        L.wlo[lane] = ra - 3 >= 0 ? S.lle[cl][ra - 3] + 1 : 0;
        L.whi[lane] = ra + 3 <= 65 ? S.fge[cl][ra + 3] : n_last;
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
        if (lane == 0) L.n_pool = 0;
        oc_wsync();
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
            if (slot + nreq <= kOcPool) {
                posted = true;
                L.same[lane] = thr; L.other[lane] = thr;
#pragma unroll
                for (int j = 0; j < 5; j++) {
                    const int v = ra - 2 + j;
                    if (!(v >= 0 && v <= 65 && !(edge && j == 2))) continue;
                    CfRun rq;
                    rq.start = (unsigned int)(v * kAzBins + a.a0); rq.pre = (unsigned int)(v * kAzBins + a.a1);
                    rq.len = 0; rq.owner = (unsigned char)lane; rq.tag = (unsigned char)(tag_cl | j);
                    L.pool[slot++] = rq;
                    if (a.w1) { rq.start = (unsigned int)(v * kAzBins); rq.pre = (unsigned int)(v * kAzBins + a.w1); L.pool[slot++] = rq; }
                }
            } else
                for (; slot < kOcPool; slot++) { CfRun rq; rq.start = 0; rq.pre = 0; rq.len = 0; rq.owner = (unsigned char)lane; rq.tag = 0; L.pool[slot] = rq; }
        }
        oc_wsync();
        oc_sweep<true>(L, tg_c, tg_s, pts_c, pts_s, lane);
        oc_wsync();
        if (walking && posted) {
            same = L.same[lane]; other = L.other[lane];
            if (!seeded && r_now >= 5.0f) walking = false;
            else {
                const unsigned long long lim = pack_fu(r_now * r_now * 0.998f, 0u);
                if (other < lim && (edge || same < lim)) walking = false;
                else if (seeded) r_seed = -1.0f;
                else wpass++;
            }
        }
        if (!__ballot(walking)) break;
    }
    if (!have) return;
    if (deferred || walking) { const int slot = atomicAdd(&S.n_defer, 1); defer_list[slot] = (unsigned int)qi; return; }
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
    ((int4 *)oc_g(P.corr))[qi] = rres;
    if (outer == 0) seed_c[qi] = closest_out;
    float4 A = make_float4(0.f, 0.f, 0.f, 0.f), B = A, C = A;
    if (rres.w != 0) { A = cloud[rres.x]; B = cloud[rres.y]; if (rres.z >= 0) C = cloud[rres.z]; }
    fp.w = __int_as_float(rres.w);
    float4 *rec = (float4 *)oc_g(P.crec) + (size_t)qi * 4;
    rec[0] = fp; rec[1] = A; rec[2] = B; rec[3] = C;
}

// ---- solve ------------------------------------------------------------------------------------------------------------------------
// One sweep over the chain's residual-block records at pose xe (LDS): every thread re-reads its <= kOcBlocks records (64 B each, L1 / L2
// hits after the first sweep) and stages their pose-independent part again -- holding them in registers across the trust-region steps
// cost the kernel 600 spilled registers at the 128 a 16-wave workgroup may use.  Sums H (21), g (6), cost -> S.sum.
template <bool kJac>
__device__ __noinline__ void oc_evaluate(OcLds &S, const double *xe, int *n_used_out)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float4 *crec = oc_g(S.pair.crec);
    const int nq = S.pair.nq;
    const bool thin = S.pair.thin != 0;
    LmAcc acc;
    acc.cost = 0.0;
    if (kJac) {
#pragma unroll
        for (int i = 0; i < 21; i++) acc.H[i] = 0.0;
#pragma unroll
        for (int i = 0; i < 6; i++) acc.g[i] = 0.0;
    }
    double x[7], Rm[9];
    for (int i = 0; i < 7; i++) x[i] = xe[i];
    {
#pragma clang fp contract(fast)
        const double ux = x[0], uy = x[1], uz = x[2], w = x[3];
        Rm[0] = 1.0 - 2.0 * (uy * uy + uz * uz); Rm[1] = 2.0 * (ux * uy - w * uz);       Rm[2] = 2.0 * (ux * uz + w * uy);
        Rm[3] = 2.0 * (ux * uy + w * uz);       Rm[4] = 1.0 - 2.0 * (ux * ux + uz * uz); Rm[5] = 2.0 * (uy * uz - w * ux);
        Rm[6] = 2.0 * (ux * uz - w * uy);       Rm[7] = 2.0 * (uy * uz + w * ux);       Rm[8] = 1.0 - 2.0 * (ux * ux + uy * uy);
    }
    int n_used = 0;
    for (int u = 0; u < kOcBlocks; u++) {
        const int qi = tid + kOcT * u;
        if (qi >= nq || (thin && (qi % kThinBlocks) % kThinStride != 0)) continue;
        const float4 cp = crec[qi * 4];
        const int kind = __float_as_int(cp.w);
        if (kind == 0) continue;
        const float4 A = crec[qi * 4 + 1], Bq = crec[qi * 4 + 2], Cq = crec[qi * 4 + 3];
        double P[6];
        stage_block(cp, A, Bq, Cq, P);
        n_used++;
        if (kind == 1) eval_block<kJac, true>(cp, make_double2(P[0], P[1]), make_double2(P[2], P[3]), make_double2(P[4], P[5]), Rm, x, acc);
        else eval_block<kJac, false>(cp, make_double2(P[0], P[1]), make_double2(P[2], P[3]), make_double2(P[4], P[5]), Rm, x, acc);
    }
    acc.cost = wave_sum_d(acc.cost);
    if (kJac) {
#pragma unroll
        for (int i = 0; i < 21; i++) acc.H[i] = wave_sum_d(acc.H[i]);
#pragma unroll
        for (int i = 0; i < 6; i++) acc.g[i] = wave_sum_d(acc.g[i]);
    }
    if (n_used_out) n_used = wave_sum_i(n_used);
    __syncthreads();   // previous readers of S.red / S.sum are done
    if (lane == 0) {
        S.red[wave][27] = acc.cost;
        if (kJac) {
            for (int i = 0; i < 21; i++) S.red[wave][i] = acc.H[i];
            for (int i = 0; i < 6; i++) S.red[wave][21 + i] = acc.g[i];
        }
        if (n_used_out) S.used[wave] = n_used;
    }
    __syncthreads();
    if (tid < 28 && (kJac || tid == 27)) {
        double t = 0.0;
        for (int w = 0; w < kOcW; w++) t += S.red[w][tid];
        S.sum[tid] = t;
    }
    if (n_used_out) { int t = 0; for (int w = 0; w < kOcW; w++) t += S.used[w]; *n_used_out = t; }
    __syncthreads();
}

// next candidate from the accepted linearisation S.cur (loops over invalid steps); sets T.done when the loop ends
__device__ __noinline__ void oc_tr_next(OcTr &T, const double *s_cur, int lane)
{
    const int max_iter = 4;
    const double min_diag = 1e-6, max_diag = 1e32;
    double radius = T.radius;
    int reuse_diagonal = T.reuse_diagonal, invalid_steps = T.invalid_steps, iter = T.iter;
    double scale[6], diag[6], x[7];
    for (int i = 0; i < 6; i++) { scale[i] = T.scale[i]; diag[i] = T.diag[i]; }
    for (int i = 0; i < 7; i++) x[i] = T.x[i];
    int done = 0, last = 0;
    double model_change = 0.0, cand[7] = { 0, 0, 0, 0, 0, 0, 0 };
    while (true) {
        if (iter >= max_iter) { done = 1; break; }
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
            if (++invalid_steps >= 5) { done = 1; break; }
            radius *= 0.5; reuse_diagonal = 1;
            continue;
        }
        invalid_steps = 0;
        double delta[6];
        for (int i = 0; i < 6; i++) delta[i] = stepv[i] * scale[i];
        manifold_plus(x, delta, cand);
        last = iter == max_iter;
        break;
    }
    if (lane == 0) {
        T.radius = radius; T.reuse_diagonal = reuse_diagonal; T.invalid_steps = invalid_steps; T.iter = iter; T.done = done; T.last = last;
        T.model_change = model_change;
        for (int i = 0; i < 6; i++) T.diag[i] = diag[i];
        for (int i = 0; i < 7; i++) T.cand[i] = cand[i];
    }
}

// after the first sweep at T.x (sums in S.sum): accept it as the linearisation, first candidate
__device__ __forceinline__ void oc_tr_begin(OcTr &T, double *s_cur, const double *s_sum, int n_used, int lane)
{
    const double gradient_tol = 1e-10;
    if (lane < 28) s_cur[lane] = s_sum[lane];
    oc_wsync();
    double gmax = 0.0;
    for (int i = 0; i < 6; i++) gmax = fmax(gmax, fabs(s_cur[21 + i]));
    if (lane == 0) {
        T.radius = 1e4; T.decrease_factor = 2.0; T.reuse_diagonal = 0; T.invalid_steps = 0; T.iter = 0; T.done = 0; T.last = 0;
        T.x_cost = s_cur[27];
        T.x_norm = norm7(T.x);
        for (int i = 0; i < 6; i++) T.scale[i] = 1.0 / (1.0 + sqrt(s_cur[sym6(i, i)]));
    }
    oc_wsync();
    if (!(n_used > 0 && gmax > gradient_tol)) { if (lane == 0) T.done = 1; return; }
    oc_tr_next(T, s_cur, lane);
}

// after the sweep at T.cand (sums in S.sum): accept / reject, radius update, next candidate
__device__ __forceinline__ void oc_tr_update(OcTr &T, double *s_cur, const double *s_sum, int lane)
{
    const double function_tol = 1e-6, gradient_tol = 1e-10, parameter_tol = 1e-8, min_rel_decrease = 1e-3, max_radius = 1e16, min_radius = 1e-32;
    const double cand_cost = s_sum[27], x_cost = T.x_cost;
    double sn = 0.0;
    for (int i = 0; i < 7; i++) sn += (T.x[i] - T.cand[i]) * (T.x[i] - T.cand[i]);
    sn = sqrt(sn);
    int done = 0;
    double radius = T.radius;
    if (sn <= parameter_tol * (T.x_norm + parameter_tol)) done = 1;
    else if (fabs(x_cost - cand_cost) <= function_tol * x_cost) done = 1;
    else {
        const double rel = (x_cost - cand_cost) / T.model_change;
        if (rel > min_rel_decrease) {
            const int last = T.last;
            double xn[7];
            for (int i = 0; i < 7; i++) xn[i] = T.cand[i];
            oc_wsync();
            if (lane == 0) for (int i = 0; i < 7; i++) T.x[i] = xn[i];
            if (last) done = 1;
            else {
                if (lane < 28) s_cur[lane] = s_sum[lane];
                const double tt = 2.0 * rel - 1.0;
                double den = 1.0 - tt * tt * tt;
                if (den < 1.0 / 3.0) den = 1.0 / 3.0;
                radius = radius / den;
                if (radius > max_radius) radius = max_radius;
                if (lane == 0) { T.x_norm = norm7(xn); T.x_cost = cand_cost; T.radius = radius; T.decrease_factor = 2.0; T.reuse_diagonal = 0; }
                oc_wsync();
                double gmax = 0.0;
                for (int i = 0; i < 6; i++) gmax = fmax(gmax, fabs(s_cur[21 + i]));
                if (gmax <= gradient_tol) done = 1;
            }
        } else {
            const double df = T.decrease_factor;
            radius = radius / df;
            if (lane == 0) { T.radius = radius; T.decrease_factor = df * 2.0; T.reuse_diagonal = 1; }
            oc_wsync();
        }
        if (!done && radius <= min_radius) done = 1;
    }
    if (done) { if (lane == 0) T.done = 1; return; }
    oc_wsync();
    oc_tr_next(T, s_cur, lane);
}

// the rare paths, kept out of line: their register needs must not shape the allocation of the hot loops
__device__ __noinline__ void oc_fallback_groups(const BatchView &b, const OdomView &o, int c, int k, int outer, const unsigned int *defer_list, int nd)
{
    const int tid = threadIdx.x, lane = tid & 63;
    for (int it = tid >> 5; it < nd; it += kOcT / 32)
        correspond_group(b, o, c, k, (int)defer_list[it], outer, lane & (kGroup - 1), lane & ~(kGroup - 1));
}
__device__ __noinline__ void oc_fallback_waves(const BatchView &b, const OdomView &o, int c, int k, int nq, bool thin)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int qi = wave; qi < nq; qi += kOcW) {
        if (thin && (qi % kThinBlocks) % kThinStride != 0) continue;
        correspond_wave(b, o, c, k, qi, lane);
    }
}

__global__ __launch_bounds__(kOcT) void k_odom_chain(BatchView b, OdomView o, int max_steps, unsigned int *wl, int defer_every, unsigned long long *stats)
{
    extern __shared__ __align__(16) unsigned char oc_smem[];
    OcLds &S = *reinterpret_cast<OcLds *>(oc_smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = o.clist ? o.clist[o.chain0 + blockIdx.x] : o.chain0 + (int)blockIdx.x;
    if (tid < 8) S.x[tid] = o.state[c * 8 + tid];
    if (tid == 0) S.stop = 0;
    unsigned int *defer_list = wl + 1 + (size_t)c * kMaxQueries;       // this chain's slice of the batch's work-list buffer
    int s_own, e_own;
    chain_bounds(o.first, o.n_scans, o.n_chains, c, s_own, e_own);
    unsigned long long n_deferred = 0;
#ifdef LMONO_OC_PROF
    unsigned long long t_search = 0, t_solve = 0, t_mark = 0, n_pairs = 0;
#define OC_MARK(acc) { if (tid == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); (acc) += t_ - t_mark; t_mark = t_; } }
#else
#define OC_MARK(acc)
#endif
    __syncthreads();
    for (int step = 0; step < max_steps; step++) {
        int own;
        const int k = o.repair ? (s_own + step < e_own && !S.stop ? s_own + step : -1) : chain_scan(o, c, step, own);
        if (k < 0) break;                         // uniform: S.stop is only written before a barrier
        const int l = k - 1;
        const int n_sharp = b.feat_n[k * 4 + 0];
        const int nq = n_sharp + b.feat_n[k * 4 + 2];
        const bool thin = lead_in_thinned(o, k, s_own);
        const bool irregular = (b.status[l] & (kStatusIrregularLines | kStatusDenseCell)) != 0;
        for (int xx = tid; xx < 2 * 66; xx += kOcT) {
            const int tc = xx / 66, v = xx % 66;
            S.elev[tc][v] = b.lb_elev[(size_t)(l * 2 + tc) * 66 + v];
            S.fge[tc][v] = b.line_first_ge[(size_t)(l * 2 + tc) * 66 + v];
            S.lle[tc][v] = b.line_last_le[(size_t)(l * 2 + tc) * 66 + v];
        }
        if (tid == 0) {
            OcPair &P = S.pair;
            P.sharp = b.sharp + (size_t)k * kMaxSharp; P.flat = b.flat + (size_t)k * kMaxFlat;
            P.cloud_c = b.less_sharp + (size_t)l * kMaxLessSharp; P.cloud_s = b.less_flat + b.off[l];
            P.pts_c = b.lbc_pts + (size_t)l * kMaxLessSharp; P.pts_s = b.lbs_pts + b.off[l];
            P.tg_c = b.lb_start + (size_t)(l * 2 + 0) * (kLineKeys + 1); P.tg_s = b.lb_start + (size_t)(l * 2 + 1) * (kLineKeys + 1);
            P.corr = (int4 *)o.corr + (size_t)c * kMaxQueries; P.seed = o.seed + (size_t)c * kMaxQueries; P.crec = o.crec + (size_t)c * kMaxQueries * 4;
            P.defer_list = defer_list;
            P.n_last_c = b.feat_n[l * 4 + 1]; P.n_last_s = b.feat_n[l * 4 + 3]; P.n_sharp = n_sharp; P.nq = nq; P.thin = thin ? 1 : 0; P.defer_every = defer_every;
        }
        // feature list of the pair: all nq, or (thinned lead-in pair) those of every kThinStride-th of the kThinBlocks shares the launch-
        // per-phase schedule deals them into: idx -> qi = kThinBlocks * (idx / n_keep) + kThinStride * (idx % n_keep)
        constexpr int n_keep = (kThinBlocks + kThinStride - 1) / kThinStride;
        const int n_list = thin ? ((nq + kThinBlocks - 1) / kThinBlocks) * n_keep : nq;
        const int n_tasks = kOcW * ((n_list + kOcT - 1) / kOcT);                       // every wave the same number of tasks
        for (int outer = 0; outer < 2; outer++) {
            if (tid == 0) { S.n_defer = 0; S.pair.outer = outer; }
            __syncthreads();                      // tables, pair record, state and the counter are in place
#ifdef LMONO_OC_PROF
            if (tid == 0) t_mark = __builtin_amdgcn_s_memtime();
#endif
            if (!irregular) {
                for (int task = wave; task < n_tasks; task += kOcW) {
                    const int idx = lane * n_tasks + task;
                    int qi = -1;
                    if (idx < n_list) qi = thin ? kThinBlocks * (idx / n_keep) + kThinStride * (idx % n_keep) : idx;
                    oc_search_task(S, S.w[wave], qi, lane);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __syncthreads();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                // features the flat search handed on (rare): 32 lanes each, the round-1 search
                const int nd = S.n_defer;
                if (tid == 0) n_deferred += (unsigned long long)nd;
                if (nd > 0) oc_fallback_groups(b, o, c, k, outer, defer_list, nd);
            } else {
                // scan pairs whose line ids are too disordered for the windowed searches: the whole-wave array-order walk, every feature
                oc_fallback_waves(b, o, c, k, nq, thin);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // the records other waves wrote are read from L2

            OC_MARK(t_search)
            // ================= solve =================
            OcTr &T = S.tr;
            if (tid < 7) T.x[tid] = S.x[tid];
            __syncthreads();
            int n_used = 0;
#ifndef OC_NO_SOLVE
            oc_evaluate<true>(S, T.x, &n_used);
            if (wave == 0) oc_tr_begin(T, S.cur, S.sum, n_used, lane);
            __syncthreads();
            while (!T.done) {
                if (T.last) oc_evaluate<false>(S, T.cand, nullptr);
                else oc_evaluate<true>(S, T.cand, nullptr);
                if (wave == 0) oc_tr_update(T, S.cur, S.sum, lane);
                __syncthreads();
            }
#endif
            double x[7];
            for (int i = 0; i < 7; i++) x[i] = T.x[i];
            const int iter = T.iter;
            __syncthreads();                      // every thread has read the sums and holds the same x
            if (tid == 0) {
                for (int i = 0; i < 7; i++) { S.x[i] = x[i]; o.state[c * 8 + i] = x[i]; }
                if (o.lm_info) { o.lm_info[c * 4 + outer] = iter; o.lm_info[c * 4 + 2 + outer] = n_used; }
                if (outer == 1) {
                    if (o.repair) {
                        const double res = boundary_residual(x, o.incr + (size_t)k * 7);
                        for (int i = 0; i < 7; i++) o.incr[(size_t)k * 7 + i] = x[i];
                        o.rstat[c * 4 + 1] += 1;
                        int agree = o.rstat[c * 4 + 2] >> 1;
                        agree = res <= o.tol ? agree + 1 : 0;
                        o.rstat[c * 4 + 2] = 1 | (agree << 1);
                        if (agree >= kRepairAgree || k + 1 >= e_own) { o.rstat[c * 4] = 1; S.stop = 1; }
                    } else {
                        if (o.incr && k >= s_own) for (int i = 0; i < 7; i++) o.incr[(size_t)k * 7 + i] = x[i];
                        if (o.ws && k == s_own - 1) for (int i = 0; i < 7; i++) o.ws[c * 8 + i] = x[i];
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // the fall-back searches read the state from o.state
            OC_MARK(t_solve)
#ifdef LMONO_OC_PROF
            if (tid == 0) n_pairs++;
#endif
        }
    }
    if (tid == 0 && stats && n_deferred) atomicAdd(&stats[0], n_deferred);
#ifdef LMONO_OC_PROF
    if (tid == 0 && stats) { atomicAdd(&stats[1], n_pairs); atomicAdd(&stats[2], t_search); atomicAdd(&stats[3], t_solve); }
#endif
}

} // namespace lmono
