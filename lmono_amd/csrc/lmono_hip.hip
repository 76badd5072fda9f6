// lmono_amd/csrc/lmono_hip.hip -- C ABI (include/lmono_hip.h) over the gfx950 kernels.  Single translation unit:
// the kernel files are included so that one `hipcc -shared` produces liblmono_hip.so.
#include "frontend.hip"
#include "odometry.hip"
#include "corr_flat.hip"
#include "mapping.hip"
#include "ba.hip"
#include "ba_solve.hip"
#include "feat.hip"
#include "marg.hip"

#include <string>
#include <vector>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <atomic>
#include <functional>
#include <memory>
#include <chrono>
#include <algorithm>
#include <cstdio>
#include <cstring>

constexpr int kListGrid = 256;       // workgroups of k_correspond_list (fixed: the work list's length is only known on the device; almost always empty)

using namespace lmono;

struct EvSet { hipEvent_t e[10]; bool reg = false, odom = false; std::vector<hipEvent_t> kev; int n_kev = 0; };  // kev: (begin, mid, end) per odometry launch pair

#include "host_workers.hpp"

// per-thread tables of lmono_mapper_process_batch's update plan (indexed by cube: 21 x 21 x 11)
struct MapPlanScratch {
    struct Run { int ind, at, len; };
    std::vector<char> is_valid; std::vector<int> add, fill, cand; std::vector<int64_t> cat_off; std::vector<Run> runs;
    MapPlanScratch() : is_valid((size_t)lmono::kMdCubes, 0), add((size_t)lmono::kMdCubes, 0), fill((size_t)lmono::kMdCubes, 0), cat_off((size_t)lmono::kMdCubes, -1) {}
};

struct lmono_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    std::vector<EvSet> sets;   // one event set per scanreg/odometry call since the last lmono_timing_reset
    int n_sets = 0;
    hipEvent_t *ev = nullptr;  // events of the current call
    int opt[LMONO_OPT_COUNT] = { 3, 0, 4, -1, 1000, 0, 0 };   // (LMONO_OPT_LEAD_SEED: 0 until measured)   // LMONO_OPT_CORR_TILE: 3 = flattened sweeps (default, needs no hash grid), 0 = 32-lane groups on the hash grid (diagnostic build)
    hipStream_t gstream[8] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };   // streams of the odometry's chain groups (LMONO_OPT_ODOM_STREAMS > 1)
    hipEvent_t gev[9] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
    unsigned long long *stats_d = nullptr;   // [0] feature points deferred by the tile search since the last lmono_timing_reset
    hipStream_t copy_stream = nullptr;       // H2D staging of lmono_batch_stage_h (runs beside the compute stream)
    hipStream_t own_stream = nullptr;        // lmono_use_own_stream: a stream of the library's that this context runs on
    bool odom_prio = false, many_queues = false;     // measurement switches read from the environment at creation (LMONO_ODOM_STREAM_PRIORITY, GPU_MAX_HW_QUEUES >= 8)
    // scratch arena of the small host-array entry points (triangulate, outlier scores, marginalise ...): chunks are allocated once and
    // reused by every later call -- no hipMalloc / hipFree (both synchronise the device) in a steady-state frame loop
    struct Chunk { char *base; size_t cap; };
    std::vector<Chunk> arena;
    size_t arena_chunk = 0, arena_off = 0;
    int n_cu = 0;                            // compute units of the device (hipDeviceProp_t::multiProcessorCount)
    int map_budget = 0;
    int cluster_budget = 0;                  // workgroups a cluster kernel (k_ba_solve<kCl>, k_map_solve) may keep resident while they poll each other: half the CUs
    char *stage = nullptr;                   // pinned staging of DevBuf's small uploads (stage_all: every buffer ever allocated, freed with the context)
    size_t stage_cap = 0;
    std::vector<void *> stage_all;
    std::vector<MapPlanScratch> plan_scratch;
    std::unique_ptr<HostWorkers> workers;    // created by the first batched call that has per-stream host work for them (LMONO_LIB_THREADS, default min(8, cores))

    hipEvent_t *next_set()
    {
        constexpr int kMaxSets = 1024;
        if (n_sets == (int)sets.size()) {
            if (n_sets >= kMaxSets) { sets[n_sets - 1].reg = sets[n_sets - 1].odom = false; return sets[n_sets - 1].e; }
            EvSet s;
            for (auto &e : s.e) if (hipEventCreate(&e) != hipSuccess) return nullptr;
            sets.push_back(s);
        }
        sets[n_sets].reg = sets[n_sets].odom = false;
        sets[n_sets].n_kev = 0;
        return sets[n_sets++].e;
    }
};

struct lmono_scan_batch {
    lmono_ctx *ctx = nullptr;
    int n_cap = 0;
    int64_t pts_cap = 0;
    int n_scans = 0;
    int64_t total = 0;
    int max_pts = 0;
    bool registered = false;
    bool grid_built = false;       // k_grid_build has run for this registration
    std::vector<int64_t> off_h;
    bool validation_pending = false;   // lmono_odom_shard_main_d ran: lmono_odom_shard_validate is the batch's first (whole) validation
    std::vector<void *> allocs;
    BatchView v{};
    int64_t *off_d = nullptr;
    float *in_owned = nullptr;     // staging buffer of lmono_scanreg_batch_h (pts_cap points), allocated on first use
    hipEvent_t staged_ev = nullptr, in_free_ev = nullptr;   // lmono_batch_stage_h: copy finished / the front end has read the staging buffer
    int64_t staged_points = -1;
    std::vector<int> feat_h;       // host copy of feat_n [n_scans][4], fetched on first use after a registration
    // odometry workspace
    int chains_cap = 0;
    double *state = nullptr, *incr = nullptr, *poses = nullptr, *xq = nullptr;
    int *corr = nullptr, *lm_info = nullptr, *corr_pair = nullptr, *seed = nullptr;
    float4 *crec = nullptr, *crec_pair = nullptr;
    unsigned int *wl = nullptr;            // work list of feature points the LDS tile search defers: [0] = count
    size_t wl_cap = 0;
    // boundary validation of the chained schedule
    double *ws = nullptr, *resid_d = nullptr;
    int *rstat = nullptr;
    unsigned int *rcount = nullptr;
    hipEvent_t rep_ev[2] = { nullptr, nullptr };
    lmono_boundary_report brep{};
    std::vector<double> resid_h;
    std::vector<int> rerun_h;
    int last_chains = 0, last_lead = 0, last_first = 0;
};

#define HIP_TRY(ctx, expr)                                                                   \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                  \
            return e_ == hipErrorOutOfMemory ? LMONO_ENOMEM : LMONO_ENODEV;                  \
        }                                                                                    \
    } while (0)

#ifdef LMONO_DIAG_SEARCH
extern "C" const char *lmono_version(void) { return "lmono-hip 0.4 (gfx950, diagnostic build: + hash-grid search)"; }
#else
extern "C" const char *lmono_version(void) { return "lmono-hip 0.4 (gfx950)"; }
#endif

extern "C" lmono_ctx *lmono_create(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return nullptr;
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    lmono_ctx *c = new lmono_ctx();
    c->device = device;
    if (const char *e = getenv("LMONO_ODOM_STREAM_PRIORITY")) c->odom_prio = atoi(e) != 0;
    if (const char *e = getenv("GPU_MAX_HW_QUEUES")) c->many_queues = atoi(e) >= 8;
    {
        // residency budget of the cluster kernels: every workgroup of a cluster spins on its partners, so all of them must be resident at once.  k_ba_solve
        // takes a whole CU's LDS (one workgroup per CU); half the CUs leaves room for whatever else the card runs (LMONO_CLUSTER_BUDGET overrides: CU masks,
        // CPX partitions report their own multiProcessorCount and need no override)
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) != hipSuccess) { delete c; return nullptr; }
        c->n_cu = prop.multiProcessorCount;
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_ba_solve<true, false>, kBaT, 0) != hipSuccess || per_cu < 1) per_cu = 1;
        c->cluster_budget = std::max(8, c->n_cu * per_cu / 2);
        c->map_budget = std::max(8, c->n_cu / 2);         // k_map_solve: one workgroup per CU assumed, as measured (more fit, none are counted on)
        if (const char *e = getenv("LMONO_CLUSTER_BUDGET")) { const int v = atoi(e); if (v >= 1) c->cluster_budget = c->map_budget = v; }
    }
    if (hipMalloc((void **)&c->stats_d, 320) != hipSuccess || hipMemset(c->stats_d, 0, 320) != hipSuccess) { delete c; return nullptr; }
    // the selection kernel needs ~62 KB of dynamic LDS
    if (hipFuncSetAttribute((const void *)k_select, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * sel_slice_bytes(kRingCap) + 4 * kSelScratch) != hipSuccess) { delete c; return nullptr; }
    if (hipFuncSetAttribute((const void *)k_voxel<kVoxSmallSlots, kVoxSmallBits, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kVoxLdsSmall) != hipSuccess) { delete c; return nullptr; }
    if (hipFuncSetAttribute((const void *)k_voxel<kVoxBigSlots, kVoxBigBits, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kVoxLdsBig) != hipSuccess) { delete c; return nullptr; }
    if (hipFuncSetAttribute((const void *)k_lm_solve, hipFuncAttributeMaxDynamicSharedMemorySize, kLmRecLds) != hipSuccess) { delete c; return nullptr; }
#ifdef LMONO_DIAG_SEARCH
    if (hipFuncSetAttribute((const void *)k_grid_build, hipFuncAttributeMaxDynamicSharedMemorySize, kGridLds) != hipSuccess) { delete c; return nullptr; }
#endif
    if (hipFuncSetAttribute((const void *)k_line_index<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLiLdsHalf) != hipSuccess) { delete c; return nullptr; }
    if (hipFuncSetAttribute((const void *)k_compact_index, hipFuncAttributeMaxDynamicSharedMemorySize, kLiLdsHalf) != hipSuccess) { delete c; return nullptr; }
    if (hipFuncSetAttribute((const void *)k_line_index<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLiLdsFull) != hipSuccess) { delete c; return nullptr; }
    // the BA-side kernels with large dynamic LDS: per device, so per context (a second context on another GPU needs them too)
    if (hipFuncSetAttribute((const void *)k_marginalize, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(MargLds)) != hipSuccess) { delete c; return nullptr; }
    if (hipFuncSetAttribute((const void *)k_marg_second_new, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Marg2Lds)) != hipSuccess) { delete c; return nullptr; }
#ifdef LMONO_TILE_PROF
    {
        int nb = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)k_corr_flat, kCfT, 0);
        std::fprintf(stderr, "[lmono diag] k_corr_flat: %d workgroups of %d threads per CU by the occupancy query, static LDS %zu B\n", nb, kCfT, sizeof(CfLds));
    }
#endif
    return c;
}

extern "C" void lmono_destroy(lmono_ctx *c)
{
    if (!c) return;
#ifdef LMONO_BOUNDS
    {
        // the checked build reports when a context goes (a C++ caller -- estimator_seq -- has no other way to ask)
        unsigned long long o[4] = { 0, 0, 0, 0 };
        (void)hipDeviceSynchronize();
        if (hipMemcpyFromSymbol(o, HIP_SYMBOL(g_ba_oob), sizeof(o)) == hipSuccess)
            fprintf(stderr, "[lmono bounds] k_ba_solve: %llu access(es) outside the batch's allocation (first: ba_solve.hip:%llu, byte offset %lld, block %llu)\n", o[0], o[1], (long long)o[2], o[3]);
    }
#endif
    for (auto &s : c->sets) { for (auto &e : s.e) (void)hipEventDestroy(e); for (auto &e : s.kev) (void)hipEventDestroy(e); }
    if (c->stats_d) (void)hipFree(c->stats_d);
    for (auto &s : c->gstream) if (s) (void)hipStreamDestroy(s);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    for (void *h : c->stage_all) (void)hipHostFree(h);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    for (auto &ch : c->arena) (void)hipFree(ch.base);
    for (auto &e : c->gev) if (e) (void)hipEventDestroy(e);
    delete c;
}

extern "C" const char *lmono_last_error(const lmono_ctx *c) { return c ? c->err.c_str() : "null context"; }

extern "C" int lmono_set_stream(lmono_ctx *c, void *s)
{
    if (!c) return LMONO_EINVAL;
    c->stream = (hipStream_t)s;
    return LMONO_OK;
}

extern "C" int lmono_use_own_stream(lmono_ctx *c)
{
    if (!c) return LMONO_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->own_stream) HIP_TRY(c, hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    return LMONO_OK;
}

extern "C" int lmono_set_option(lmono_ctx *c, int key, int value)
{
    if (!c || key < 0 || key >= LMONO_OPT_COUNT) return LMONO_EINVAL;
#ifdef LMONO_DIAG_SEARCH
    const bool grid_search = true;
#else
    const bool grid_search = false;  // the round-1 hash-grid search (0) is compiled into the diagnostic build only
#endif
    const bool ok = key == LMONO_OPT_CORR_TILE ? (value == 3 || (value == 0 && grid_search))
                  : key == LMONO_OPT_DEFER_EVERY ? value >= 0
                  : key == LMONO_OPT_ODOM_STREAMS ? (value >= 1 && value <= 8)
                  : key == LMONO_OPT_BOUNDARY_TOL ? value >= 0
                  : key == LMONO_OPT_BA_CLUSTER ? (value == 0 || value == 1 || value == 2 || value == 4 || value == 8)
                  : key == LMONO_OPT_LEAD_SEED ? (value >= 0 && value <= 1)
                  : value >= -1;                                     // LMONO_OPT_LEAD_FULL
    if (!ok) { c->err = "lmono_set_option: value out of range for this option"; return LMONO_EINVAL; }
    c->opt[key] = value;
    return LMONO_OK;
}

extern "C" int lmono_get_option(lmono_ctx *c, int key, int *value)
{
    if (!c || !value || key < 0 || key >= LMONO_OPT_COUNT) return LMONO_EINVAL;
    *value = c->opt[key];
    return LMONO_OK;
}

extern "C" int lmono_synchronize(lmono_ctx *c)
{
    if (!c) return LMONO_EINVAL;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return LMONO_OK;
}

template <typename T>
static bool dalloc(lmono_scan_batch *b, T *&p, size_t count)
{
    void *q = nullptr;
    if (hipMalloc(&q, (count > 0 ? count : 1) * sizeof(T)) != hipSuccess) return false;
    b->allocs.push_back(q);
    p = (T *)q;
    return true;
}

extern "C" void lmono_batch_destroy(lmono_scan_batch *b)
{
    if (!b) return;
    for (void *p : b->allocs) (void)hipFree(p);
    for (auto &e : b->rep_ev) if (e) (void)hipEventDestroy(e);
    if (b->staged_ev) (void)hipEventDestroy(b->staged_ev);
    if (b->in_free_ev) (void)hipEventDestroy(b->in_free_ev);
    delete b;
}

extern "C" lmono_scan_batch *lmono_batch_create(lmono_ctx *c, int n_cap, int64_t pts_cap)
{
    if (!c || n_cap <= 0 || pts_cap <= 0) return nullptr;
    if (hipSetDevice(c->device) != hipSuccess) return nullptr;
    lmono_scan_batch *b = new lmono_scan_batch();
    b->ctx = c; b->n_cap = n_cap; b->pts_cap = pts_cap;
    BatchView &v = b->v;
    const size_t T = (size_t)pts_cap, N = (size_t)n_cap;
    bool ok = true;
    ok = ok && dalloc(b, b->off_d, N + 1);
    ok = ok && dalloc(b, v.cloud, T) && dalloc(b, v.curv, T) && dalloc(b, v.label, T) && dalloc(b, v.gap, T);
    ok = ok && dalloc(b, v.ring_tmp, T);
    ok = ok && dalloc(b, v.seg_hist, ((T >> 10) + N + 1) * 64) && dalloc(b, v.scan_ends, N * 2) && dalloc(b, v.scan_half, N) && dalloc(b, v.scan_ori, N * 2);
    ok = ok && dalloc(b, v.ring_begin, N * 65) && dalloc(b, v.n_cloud, N) && dalloc(b, v.status, N);
    ok = ok && dalloc(b, v.sel_sharp, N * 64 * 6 * 20) && dalloc(b, v.sel_sharp_n, N * 64 * 6);
    ok = ok && dalloc(b, v.sel_flat, N * 64 * 6 * 4) && dalloc(b, v.sel_flat_n, N * 64 * 6);
    ok = ok && dalloc(b, v.lf_tmp, T) && dalloc(b, v.lf_n, N * 64) && dalloc(b, v.vox_todo, N * 64 + 1) && dalloc(b, v.sel_todo, N * 64 + 1) && dalloc(b, v.li_todo, N * 2 + 1);
    ok = ok && dalloc(b, v.sharp, N * kMaxSharp) && dalloc(b, v.less_sharp, N * kMaxLessSharp);
    ok = ok && dalloc(b, v.flat, N * kMaxFlat) && dalloc(b, v.less_flat, T);
    ok = ok && dalloc(b, v.feat_n, N * 4) && dalloc(b, v.line_first_ge, N * 2 * 66) && dalloc(b, v.line_last_le, N * 2 * 66);
    ok = ok && dalloc(b, v.cg_cell, N * kCornerTable) && dalloc(b, v.sg_cell, N * kSurfTable);
    ok = ok && dalloc(b, v.cg_pts, N * kMaxLessSharp) && dalloc(b, v.sg_pts, T) && dalloc(b, v.grid_mask, N * 2);
    ok = ok && dalloc(b, v.lbc_pts, N * kMaxLessSharp + kLbPad) && dalloc(b, v.lbs_pts, T + kLbPad) && dalloc(b, v.lb_start, N * 2 * (kLineKeys + 1)) && dalloc(b, v.lb_elev, N * 2 * 66);
    ok = ok && dalloc(b, b->incr, N * 7) && dalloc(b, b->poses, N * 7) && dalloc(b, b->xq, 8);
    ok = ok && dalloc(b, b->corr_pair, (size_t)kMaxQueries * 4) && dalloc(b, b->crec_pair, (size_t)kMaxQueries * 4);
    if (!ok) {
        c->err = "lmono_batch_create: hipMalloc failed";
        lmono_batch_destroy(b);
        return nullptr;
    }
    v.off = b->off_d;
    return b;
}

// rings a sensor can produce (scanRegistration keeps rings 0..50 of a 64-line sensor): grids of the per-ring kernels
static int rings_used(int n_lines) { return n_lines == 64 ? 51 : n_lines; }

static int check_launch(lmono_ctx *c, const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { c->err = std::string(what) + ": " + hipGetErrorString(e); return LMONO_ENODEV; }
    return LMONO_OK;
}

// The front end (scanRegistration) over scans scan0 .. scan0 + n_scans - 1 of the batch: the per-scan kernels' grids cover n_scans scans, the
// batch view tells them where they start.  A whole-batch registration is (0, n); the online stream registers one slot at a time.
static int scanreg_launch(lmono_ctx *c, lmono_scan_batch *b, int scan0, int n_scans, int64_t max_pts, int n_limit = 0)
{
    BatchView v = b->v;
    v.scan0 = scan0; v.n_limit = n_limit;
    hipStream_t st = c->stream;
    c->ev = c->next_set();
    if (!c->ev) { c->err = "hipEventCreate failed"; return LMONO_ENODEV; }
    c->sets[c->n_sets - 1].reg = true;
    HIP_TRY(c, hipMemsetAsync(v.status + scan0, 0, sizeof(int) * n_scans, st));
    HIP_TRY(c, hipMemsetAsync(v.vox_todo, 0, sizeof(int), st));
    HIP_TRY(c, hipMemsetAsync(v.sel_todo, 0, sizeof(int), st));
    HIP_TRY(c, hipMemsetAsync(v.li_todo, 0, sizeof(int), st));
    HIP_TRY(c, hipEventRecord(c->ev[0], st));
    const int rt_tiles = (int)((max_pts + kRtTile - 1) / kRtTile);
    hipLaunchKernelGGL(k_ring_ends, dim3(n_scans), dim3(256), 0, st, v);
    if (rt_tiles > 0) hipLaunchKernelGGL(k_ring_tag, dim3(rt_tiles, n_scans), dim3(kRtT), 0, st, v);
    hipLaunchKernelGGL(k_ring_offsets, dim3(n_scans), dim3(64), 0, st, v);
    if (rt_tiles > 0) hipLaunchKernelGGL(k_ring_scatter, dim3(rt_tiles, n_scans), dim3(kRtT), 0, st, v);
    HIP_TRY(c, hipEventRecord(c->ev[1], st));
    const int tiles = (int)((max_pts + kCurvTile - 1) / kCurvTile);
#if !LMONO_FUSE_CURV_SELECT
    if (tiles > 0) hipLaunchKernelGGL(k_curvature, dim3(tiles, n_scans), dim3(256), 0, st, v);
#else
    (void)tiles;                                    // the curvature is computed inside k_select
#endif
    HIP_TRY(c, hipEventRecord(c->ev[2], st));
    // the kernels below run one workgroup per ring of the sensor; the counters of the rings it cannot produce stay zero
    const int n_rings = rings_used(b->v.n_lines);
    HIP_TRY(c, hipMemsetAsync(v.sel_sharp_n + (size_t)scan0 * 64 * 6, 0, sizeof(int) * (size_t)n_scans * 64 * 6, st));
    HIP_TRY(c, hipMemsetAsync(v.sel_flat_n + (size_t)scan0 * 64 * 6, 0, sizeof(int) * (size_t)n_scans * 64 * 6, st));
    HIP_TRY(c, hipMemsetAsync(v.lf_n + (size_t)scan0 * 64, 0, sizeof(int) * (size_t)n_scans * 64, st));
    hipLaunchKernelGGL(k_select, dim3((n_rings + 3) / 4, n_scans), dim3(256), 4 * sel_slice_bytes(kSelSmallCap) + 4 * kSelScratch, st, v, kSelSmallCap, 0);
    hipLaunchKernelGGL(k_select, dim3(kSelBigGrid), dim3(256), 4 * sel_slice_bytes(kRingCap) + 4 * kSelScratch, st, v, (int)kRingCap, 1);
    HIP_TRY(c, hipEventRecord(c->ev[3], st));
    hipLaunchKernelGGL((k_voxel<kVoxSmallSlots, kVoxSmallBits, true>), dim3(n_rings, n_scans), dim3(256), kVoxLdsSmall, st, v);
    hipLaunchKernelGGL((k_voxel<kVoxBigSlots, kVoxBigBits, false>), dim3(kVoxBigGrid), dim3(256), kVoxLdsBig, st, v);
    HIP_TRY(c, hipEventRecord(c->ev[4], st));
#if LMONO_FUSE_COMPACT_INDEX
    hipLaunchKernelGGL(k_compact_index, dim3(n_scans), dim3(kLiT), kLiLdsHalf, st, v);       // compaction + (line, bin) index of the two "last" clouds
#else
    hipLaunchKernelGGL(k_compact, dim3(n_scans), dim3(kCompT), 0, st, v);
#endif
    HIP_TRY(c, hipEventRecord(c->ev[5], st));
    // the hash grids serve the 32-lane-group search (LMONO_OPT_CORR_TILE 0) and the deferred lists of modes 1 and 2; the default
    // (flattened sweeps) works on the line index alone, so the grids are built on demand (ensure_grid)
#ifdef LMONO_DIAG_SEARCH
    if (c->opt[LMONO_OPT_CORR_TILE] != 3) {
        hipLaunchKernelGGL(k_grid_build, dim3(n_scans, 1 + kGridPar), dim3(1024), kGridLds, st, v);
        b->grid_built = true; v.has_grid = 1; b->v.has_grid = 1;
    }
#endif
    HIP_TRY(c, hipEventRecord(c->ev[6], st));
#if !LMONO_FUSE_COMPACT_INDEX
    hipLaunchKernelGGL(k_line_index<true>, dim3(n_scans, 2), dim3(kLiT), kLiLdsHalf, st, v);
#endif
    hipLaunchKernelGGL(k_line_index<false>, dim3(kLiBigGrid), dim3(kLiT), kLiLdsFull, st, v);
    HIP_TRY(c, hipEventRecord(c->ev[7], st));
    return check_launch(c, "scanreg kernels");
}

extern "C" int lmono_scanreg_batch(lmono_ctx *c, lmono_scan_batch *b, const float *xyzi_d, const int64_t *offsets_h,
                                   int n_scans, int n_lines, float min_range);

extern "C" int lmono_scanreg_batch_h(lmono_ctx *c, lmono_scan_batch *b, const float *xyzi_h, const int64_t *offsets_h,
                                     int n_scans, int n_lines, float min_range)
{
    if (!c || !b || !xyzi_h || !offsets_h || n_scans <= 0) return LMONO_EINVAL;
    if (offsets_h[0] != 0) { c->err = "offsets must start at 0"; return LMONO_EINVAL; }
    const int64_t total = offsets_h[n_scans];
    if (total < 0 || total > b->pts_cap) { c->err = "batch: too many points"; return LMONO_ECAPACITY; }
    HIP_TRY(c, hipSetDevice(c->device));
    if (!b->in_owned) {
        void *q = nullptr;
        HIP_TRY(c, hipMalloc(&q, (size_t)(b->pts_cap > 0 ? b->pts_cap : 1) * 16));
        b->allocs.push_back(q);
        b->in_owned = (float *)q;
    }
    if (total > 0) {
        // staged: the caller's (pageable) buffer is free again when this returns
        HIP_TRY(c, hipMemcpyAsync(b->in_owned, xyzi_h, (size_t)total * 16, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return lmono_scanreg_batch(c, b, b->in_owned, offsets_h, n_scans, n_lines, min_range);
}

extern "C" int lmono_scanreg_batch(lmono_ctx *c, lmono_scan_batch *b, const float *xyzi_d, const int64_t *offsets_h,
                                   int n_scans, int n_lines, float min_range)
{
    if (!c || !b || !xyzi_d || !offsets_h || n_scans <= 0) return LMONO_EINVAL;
    if (n_lines != 16 && n_lines != 32 && n_lines != 64) { c->err = "n_lines must be 16, 32 or 64"; return LMONO_EINVAL; }
    if (n_scans > b->n_cap) { c->err = "batch: too many scans"; return LMONO_ECAPACITY; }
    int64_t max_pts = 0;
    for (int s = 0; s < n_scans; s++) {
        const int64_t m = offsets_h[s + 1] - offsets_h[s];
        if (m < 0 || m > INT_MAX / 2) { c->err = "bad offsets"; return LMONO_EINVAL; }
        max_pts = m > max_pts ? m : max_pts;
    }
    const int64_t total = offsets_h[n_scans] - offsets_h[0];
    if (offsets_h[0] != 0) { c->err = "offsets must start at 0"; return LMONO_EINVAL; }
    if (total > b->pts_cap) { c->err = "batch: too many points"; return LMONO_ECAPACITY; }
    HIP_TRY(c, hipSetDevice(c->device));
    b->off_h.assign(offsets_h, offsets_h + n_scans + 1);
    b->n_scans = n_scans; b->total = total; b->max_pts = (int)max_pts; b->registered = false; b->grid_built = false;
    b->feat_h.clear();
    BatchView &v = b->v;
    v.in = (const float4 *)xyzi_d; v.n_scans = n_scans; v.scan0 = 0; v.n_lines = n_lines; v.min_range = min_range; v.has_grid = 0;
    HIP_TRY(c, hipMemcpyAsync(b->off_d, b->off_h.data(), sizeof(int64_t) * (n_scans + 1), hipMemcpyHostToDevice, c->stream));
    int rc = scanreg_launch(c, b, 0, n_scans, max_pts);
    if (rc) return rc;
    b->registered = true;
    return LMONO_OK;
}

// ---- streamed input: H2D of the next working set beside the compute of the current one ---------------------------------------------
extern "C" void *lmono_host_alloc(lmono_ctx *c, size_t bytes)
{
    if (!c || bytes == 0 || hipSetDevice(c->device) != hipSuccess) return nullptr;
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { c->err = "lmono_host_alloc: hipHostMalloc failed"; return nullptr; }
    return p;
}
extern "C" void lmono_host_free(lmono_ctx *c, void *p) { if (c && p) (void)hipHostFree(p); }

extern "C" int lmono_batch_stage_h(lmono_ctx *c, lmono_scan_batch *b, const float *xyzi_h, int64_t total_points)
{
    if (!c || !b || !xyzi_h || total_points < 0) return LMONO_EINVAL;
    if (total_points > b->pts_cap) { c->err = "batch: too many points"; return LMONO_ECAPACITY; }
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->copy_stream) HIP_TRY(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    if (!b->in_owned) {
        void *q = nullptr;
        HIP_TRY(c, hipMalloc(&q, (size_t)(b->pts_cap > 0 ? b->pts_cap : 1) * 16));
        b->allocs.push_back(q);
        b->in_owned = (float *)q;
    }
    if (!b->staged_ev) HIP_TRY(c, hipEventCreateWithFlags(&b->staged_ev, hipEventDisableTiming));
    if (!b->in_free_ev) HIP_TRY(c, hipEventCreateWithFlags(&b->in_free_ev, hipEventDisableTiming));
    // the front end of this batch's previous registration may still read the staging buffer
    else HIP_TRY(c, hipStreamWaitEvent(c->copy_stream, b->in_free_ev, 0));
    if (total_points > 0) HIP_TRY(c, hipMemcpyAsync(b->in_owned, xyzi_h, (size_t)total_points * 16, hipMemcpyHostToDevice, c->copy_stream));
    HIP_TRY(c, hipEventRecord(b->staged_ev, c->copy_stream));
    b->staged_points = total_points;
    return LMONO_OK;
}

extern "C" int lmono_scanreg_batch_staged(lmono_ctx *c, lmono_scan_batch *b, const int64_t *offsets_h, int n_scans, int n_lines, float min_range)
{
    if (!c || !b || !offsets_h || n_scans <= 0) return LMONO_EINVAL;
    if (b->staged_points < 0 || !b->in_owned) { c->err = "lmono_scanreg_batch_staged: nothing staged (lmono_batch_stage_h first)"; return LMONO_EINVAL; }
    if (offsets_h[n_scans] != b->staged_points) { c->err = "lmono_scanreg_batch_staged: offsets do not match the staged points"; return LMONO_EINVAL; }
    HIP_TRY(c, hipStreamWaitEvent(c->stream, b->staged_ev, 0));          // the device waits for the copy, the host does not
    const int rc = lmono_scanreg_batch(c, b, b->in_owned, offsets_h, n_scans, n_lines, min_range);
    HIP_TRY(c, hipEventRecord(b->in_free_ev, c->stream));
    b->staged_points = -1;
    return rc;
}

extern "C" int lmono_timing_reset(lmono_ctx *c)
{
    if (!c) return LMONO_EINVAL;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemset(c->stats_d, 0, 320));
    c->n_sets = 0;
    return LMONO_OK;
}

extern "C" int lmono_timing_read(lmono_ctx *c, double *ms, int cap, int *n_scanreg_calls, int *n_odom_calls)
{
    if (!c || !ms) return LMONO_EINVAL;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    double sum[13] = { 0 };
    int nr = 0, no = 0;
    float t;
    for (int i = 0; i < c->n_sets; i++) {
        const EvSet &s = c->sets[i];
        if (s.reg) {
            nr++;
            if (hipEventElapsedTime(&t, s.e[0], s.e[7]) == hipSuccess) sum[0] += t;
            for (int k = 0; k < 7; k++)
                if (hipEventElapsedTime(&t, s.e[k], s.e[k + 1]) == hipSuccess) sum[2 + k] += t;
        }
        if (s.odom) {
            no++;
            if (hipEventElapsedTime(&t, s.e[8], s.e[9]) == hipSuccess) sum[1] += t;
            for (int k = 0; k + 2 < s.n_kev; k += 3) {
                if (k + 2 >= (int)s.kev.size()) break;
                if (hipEventElapsedTime(&t, s.kev[k], s.kev[k + 1]) == hipSuccess) sum[9] += t;
                if (hipEventElapsedTime(&t, s.kev[k + 1], s.kev[k + 2]) == hipSuccess) sum[10] += t;
                sum[11] += 1.0;
            }
        }
    }
    {
        unsigned long long st[40] = { 0 };
        HIP_TRY(c, hipMemcpy(st, c->stats_d, 320, hipMemcpyDeviceToHost));
        sum[12] = (double)st[0];
        for (int i = 1; i < 40 && 12 + i < cap; i++) ms[12 + i] = (double)st[i];     // diagnostic words (LMONO_TILE_PROF builds)
    }
    for (int i = 0; i < cap && i < 13; i++) ms[i] = sum[i];
    if (n_scanreg_calls) *n_scanreg_calls = nr;
    if (n_odom_calls) *n_odom_calls = no;
    return LMONO_OK;
}

extern "C" int lmono_batch_counts(lmono_ctx *c, lmono_scan_batch *b, int32_t *counts_h)
{
    if (!c || !b || !counts_h || !b->registered) return LMONO_EINVAL;
    const int n = b->n_scans;
    std::vector<int> nc(n), fn(n * 4), stt(n);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(nc.data(), b->v.n_cloud, sizeof(int) * n, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(fn.data(), b->v.feat_n, sizeof(int) * n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(stt.data(), b->v.status, sizeof(int) * n, hipMemcpyDeviceToHost));
    for (int s = 0; s < n; s++) {
        counts_h[6 * s] = nc[s];
        for (int k = 0; k < 4; k++) counts_h[6 * s + 1 + k] = fn[4 * s + k];
        counts_h[6 * s + 5] = stt[s];
    }
    return LMONO_OK;
}

extern "C" int lmono_batch_get_cloud(lmono_ctx *c, lmono_scan_batch *b, int scan, int which, float *out_h, int cap)
{
    if (!c || !b || !out_h || !b->registered || scan < 0 || scan >= b->n_scans || which < 0 || which > 4) return LMONO_EINVAL;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    int n = 0;
    const float4 *src = nullptr;
    int fn[4];
    HIP_TRY(c, hipMemcpy(fn, b->v.feat_n + scan * 4, sizeof(fn), hipMemcpyDeviceToHost));
    switch (which) {
    case 0: HIP_TRY(c, hipMemcpy(&n, b->v.n_cloud + scan, sizeof(int), hipMemcpyDeviceToHost)); src = b->v.cloud + b->off_h[scan]; break;
    case 1: n = fn[0]; src = b->v.sharp + (size_t)scan * kMaxSharp; break;
    case 2: n = fn[1]; src = b->v.less_sharp + (size_t)scan * kMaxLessSharp; break;
    case 3: n = fn[2]; src = b->v.flat + (size_t)scan * kMaxFlat; break;
    default: n = fn[3]; src = b->v.less_flat + b->off_h[scan]; break;
    }
    if (n > cap) { c->err = "get_cloud: output capacity too small"; return LMONO_ECAPACITY; }
    if (n > 0) HIP_TRY(c, hipMemcpy(out_h, src, sizeof(float4) * n, hipMemcpyDeviceToHost));
    return n;
}

extern "C" int lmono_batch_get_curvature(lmono_ctx *c, lmono_scan_batch *b, int scan, float *curv_h, int32_t *label_h, int cap)
{
    if (!c || !b || !b->registered || scan < 0 || scan >= b->n_scans) return LMONO_EINVAL;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    int n = 0;
    HIP_TRY(c, hipMemcpy(&n, b->v.n_cloud + scan, sizeof(int), hipMemcpyDeviceToHost));
    if (n > cap) { c->err = "get_curvature: output capacity too small"; return LMONO_ECAPACITY; }
    if (curv_h && n > 0) HIP_TRY(c, hipMemcpy(curv_h, b->v.curv + b->off_h[scan], sizeof(float) * n, hipMemcpyDeviceToHost));
    if (label_h && n > 0) {
        std::vector<int8_t> tmp(n);
        HIP_TRY(c, hipMemcpy(tmp.data(), b->v.label + b->off_h[scan], n, hipMemcpyDeviceToHost));
        for (int i = 0; i < n; i++) label_h[i] = tmp[i];
    }
    return n;
}

static int ensure_odom_ws(lmono_ctx *c, lmono_scan_batch *b, int n_chains)
{
    if (n_chains <= b->chains_cap) return LMONO_OK;
    // (re)allocate; old buffers stay in allocs and are freed with the batch
    bool ok = dalloc(b, b->state, (size_t)n_chains * 8) && dalloc(b, b->corr, (size_t)n_chains * kMaxQueries * 4) &&
              dalloc(b, b->lm_info, (size_t)n_chains * 4) && dalloc(b, b->crec, (size_t)n_chains * kMaxQueries * 4) &&
              dalloc(b, b->seed, (size_t)n_chains * kMaxQueries) && dalloc(b, b->wl, 8 * ((size_t)n_chains * kMaxQueries + 1)) &&
              dalloc(b, b->ws, (size_t)n_chains * 8) && dalloc(b, b->resid_d, (size_t)n_chains) && dalloc(b, b->rstat, (size_t)n_chains * 4) &&
              dalloc(b, b->rcount, (size_t)n_chains + 2);
    if (!ok) { c->err = "odometry workspace: hipMalloc failed"; return LMONO_ENOMEM; }
    b->chains_cap = n_chains;
    return LMONO_OK;
}

// hash grids of a registered batch, for the searches that use them
static int ensure_grid(lmono_ctx *c, lmono_scan_batch *b)
{
    if (b->grid_built) return LMONO_OK;
#ifdef LMONO_DIAG_SEARCH
    hipLaunchKernelGGL(k_grid_build, dim3(b->n_scans, 1 + kGridPar), dim3(1024), kGridLds, c->stream, b->v);
    int rc = check_launch(c, "k_grid_build");
    if (rc) return rc;
    b->grid_built = true; b->v.has_grid = 1;
    return LMONO_OK;
#else
    c->err = "the hash-grid searches exist in the diagnostic build only (-DLMONO_DIAG_SEARCH)";
    return LMONO_EINVAL;
#endif
}

// Chain-group streams of a context: forks the library's group streams off the context stream, joins them again on every exit path.
struct GroupFork {
    lmono_ctx *c; int G, g_own; bool forked = false;
    GroupFork(lmono_ctx *c_, int G_, int g_own_) : c(c_), G(G_), g_own(g_own_) {}
    int fork()
    {
        if (G <= 1) return LMONO_OK;
        for (int g = g_own; g < G; g++)
            if (!c->gstream[g]) {
                // LMONO_ODOM_STREAM_PRIORITY=1 (measurement switch): the chain groups' streams at the highest priority, so that their short dependent
                // launches are dispatched ahead of another context's wide grids (a front end running beside the odometry)
                int lo = 0, hi = 0;
                if (c->odom_prio && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess) HIP_TRY(c, hipStreamCreateWithPriority(&c->gstream[g], hipStreamNonBlocking, hi));
                else HIP_TRY(c, hipStreamCreateWithFlags(&c->gstream[g], hipStreamNonBlocking));
            }
        for (int g = 0; g <= G && g < 9; g++) if (!c->gev[g]) HIP_TRY(c, hipEventCreateWithFlags(&c->gev[g], hipEventDisableTiming));
        HIP_TRY(c, hipEventRecord(c->gev[0], c->stream));
        forked = true;                           // from here on the group streams may carry work: join() must run
        for (int g = g_own; g < G; g++) HIP_TRY(c, hipStreamWaitEvent(c->gstream[g], c->gev[0], 0));
        return LMONO_OK;
    }
    // every forked stream is waited on by the context stream; if recording / waiting itself fails, the stream is synchronised instead,
    // so that nothing enqueued later (or the destruction of the batch) can overtake the group's kernels
    int join()
    {
        if (!forked) return LMONO_OK;
        forked = false;
        int rc = LMONO_OK;
        for (int g = g_own; g < G; g++) {
            if (hipEventRecord(c->gev[g + 1], c->gstream[g]) != hipSuccess || hipStreamWaitEvent(c->stream, c->gev[g + 1], 0) != hipSuccess) {
                (void)hipStreamSynchronize(c->gstream[g]);
                c->err = "odometry: joining a chain-group stream failed"; rc = LMONO_ENODEV;
            }
        }
        return rc;
    }
    ~GroupFork() { (void)join(); }
};

// number of chain groups a launch sequence over n_ch chains uses on this context
static int odom_groups(const lmono_ctx *c, int n_ch)
{
    constexpr int kMinChainsPerGroup = 32;
    int G = c->opt[LMONO_OPT_ODOM_STREAMS];
    G = G < 1 ? 1 : (G > 8 ? 8 : G);
    if (c->stream != nullptr && G > 3 && !c->many_queues) G = 3;      // (GPU_MAX_HW_QUEUES >= 8 in the environment: one hardware queue per stream anyway)
    while (G > 1 && n_ch / G < kMinChainsPerGroup) G--;      // a group below 32 chains cannot fill its share of the CUs
    if (c->opt[LMONO_OPT_CORR_TILE] != 3) G = 1;            // only the default search is grouped
    return G;
}

// Steps [step_a, step_b) x 2 outer iterations of the chains [0, n_ch) of view o (o.clist set: of the listed chains), in G chain groups.
// Chain groups: with LMONO_OPT_ODOM_STREAMS = G > 1 the chains are cut into G groups, each advancing on its own stream, so that
// one group's solve (one workgroup per chain: a quarter of the CUs' wave slots at most) and the ragged tail of its search kernel
// run beside the other groups' searches.  Group 0 carries the per-kernel events.  The runtime maps streams onto 4 hardware queues
// (GPU_MAX_HW_QUEUES): the null stream on one of its own, created streams round-robin on the other three.  So the default context
// (null stream) runs group 0 on the null stream + 3 group streams = 4 distinct queues; a context on a caller-created stream runs
// at most 3 groups, all on the library's own streams (4 created streams would put two groups on one queue: 64-70 instead of 51 ms
// per step measured), and the caller's stream only forks and joins.
static int odom_launch_steps(lmono_ctx *c, lmono_scan_batch *b, const OdomView &o, int n_ch, int step_a, int step_b, int G, EvSet *es, int *ne)
{
    const int tile = c->opt[LMONO_OPT_CORR_TILE];
    hipStream_t st = c->stream;
    const int g_own = st == nullptr ? 1 : 0;                  // first group that runs on a stream of the library
    const size_t wl_stride = (size_t)b->chains_cap * kMaxQueries + 1;
    auto kev = [&](int i) -> hipEvent_t {
        if (!es) return nullptr;
        while ((int)es->kev.size() <= i) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return nullptr; es->kev.push_back(e); }
        return es->kev[i];
    };
    GroupFork fork(c, G, g_own);
    int rc = fork.fork();
    if (rc) return rc;
    for (int step = step_a; step < step_b; step++) {
        for (int outer = 0; outer < 2; outer++) {
            for (int g = 0; g < G; g++) {
                hipStream_t sg = (G == 1 || g < g_own) ? st : c->gstream[g];
                OdomView og = o;
                og.chain0 = (int)((long long)g * n_ch / G); og.chain1 = (int)((long long)(g + 1) * n_ch / G);
                const int ng = og.chain1 - og.chain0;
                if (ng <= 0) continue;
                unsigned int *wlg = b->wl + g * wl_stride;
                hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
                if (g == 0 && es) { e0 = kev(*ne); e1 = kev(*ne + 1); e2 = kev(*ne + 2); }
                if (e0 && e1 && e2) (void)hipEventRecord(e0, sg);
                if (tile == 3) {
                    hipLaunchKernelGGL(k_corr_flat, dim3(8 * ((ng + 7) / 8) * kCfBlocks), dim3(kCfT), 0, sg, b->v, og, step, outer, wlg, c->opt[LMONO_OPT_DEFER_EVERY], c->stats_d);
                    hipLaunchKernelGGL(k_correspond_list, dim3(kListGrid), dim3(256), 0, sg, b->v, og, step, outer, (const unsigned int *)wlg, c->stats_d);
                }
#ifdef LMONO_DIAG_SEARCH
                else
                    hipLaunchKernelGGL(k_correspond, dim3(8 * ((ng + 7) / 8) * kCorrBlocks), dim3(256), 0, sg, b->v, og, step, outer);
#endif
                if (e0 && e1 && e2) (void)hipEventRecord(e1, sg);
                hipLaunchKernelGGL(k_lm_solve, dim3(ng), dim3(kLmT), kLmRecLds, sg, b->v, og, step, outer, tile ? wlg : (unsigned int *)nullptr);
                if (e0 && e1 && e2) { (void)hipEventRecord(e2, sg); *ne += 3; }
            }
        }
    }
    return fork.join();
}

// Boundary validation + repair rounds of the chained schedule ("self-validating chains").  ext: incr[first - 1]
// was supplied by the caller (previous rank's last increment).  Synchronises the context stream (the flagged count decides what is
// launched).  Results in b->brep / b->resid_h / b->rerun_h.
static int odom_validate(lmono_ctx *c, lmono_scan_batch *b, OdomView o, bool ext, bool first_call, EvSet *es = nullptr, int *ne = nullptr)
{
    hipStream_t st = c->stream;
    const int n_chains = o.n_chains;
    lmono_boundary_report &R = b->brep;
    if (first_call) {
        R = lmono_boundary_report{};
        R.n_chains = n_chains; R.tol = o.tol;
        b->resid_h.assign(n_chains, 0.0); b->rerun_h.assign(n_chains, 0);
    }
    if (!b->rep_ev[0]) for (auto &e : b->rep_ev) HIP_TRY(c, hipEventCreate(&e));
    HIP_TRY(c, hipEventRecord(b->rep_ev[0], st));
    int max_len = 0;
    for (int ch = 0; ch < n_chains; ch++) { int s, e; chain_bounds(o.first, o.n_scans, n_chains, ch, s, e); max_len = e - s > max_len ? e - s : max_len; }
    const int tile = c->opt[LMONO_OPT_CORR_TILE];
    std::vector<int> rs((size_t)n_chains * 4);
    const int max_rounds = n_chains + 1;
    int still_flagged = 0;               // boundaries the LAST check of the loop flagged (non-zero only when the round cap ends the loop)
    for (int round = 0; round <= max_rounds; round++) {
        const bool very_first = first_call && round == 0;
        hipLaunchKernelGGL(k_boundary_check, dim3(1), dim3(256), 0, st, o, very_first ? b->resid_d : (double *)nullptr, very_first ? 1 : 0, ext ? 1 : 0);
        unsigned int cnt[2] = { 0, 0 };
        HIP_TRY(c, hipMemcpyAsync(cnt, b->rcount, sizeof(cnt), hipMemcpyDeviceToHost, st));
        if (very_first) HIP_TRY(c, hipMemcpyAsync(b->resid_h.data(), b->resid_d, sizeof(double) * n_chains, hipMemcpyDeviceToHost, st));     // both copies on the
        HIP_TRY(c, hipStreamSynchronize(st));                                                                                                 // context stream, one wait
        if (very_first) {
            for (int ch = 0; ch < n_chains; ch++) R.max_resid = b->resid_h[ch] > R.max_resid ? b->resid_h[ch] : R.max_resid;
            // boundary_residual() answers 1e300 for a NaN increment: no repair can make such a boundary agree -- report it instead of re-running
            // its chain in every round
            if (R.max_resid >= 1e299) { c->err = "odometry: a chain boundary holds a NaN increment (a scan pair without a usable solution)"; return LMONO_ESCAN; }
        }
        const int nf = (int)cnt[0];
        still_flagged = nf;
        if (nf == 0 || round == max_rounds) break;      // the check behind the last allowed round only counts what is left
        R.flagged += nf; R.rounds += 1;
        OdomView orp = o;
        orp.repair = 1; orp.clist = (const int *)(b->rcount + 2); orp.lead_full = -1;
        const int G = odom_groups(c, nf);
        if (tile) for (int g = 0; g < G; g++) HIP_TRY(c, hipMemsetAsync(b->wl + g * ((size_t)b->chains_cap * kMaxQueries + 1), 0, sizeof(unsigned int), st));
        // a repair chain usually agrees with the stored increments after a few pairs: launch in chunks, ask the device how many still run
        int done = 0, chunk = 3;          // 3, 6, 8, 8 ...: most chains agree after 2-4 pairs, the slowest after ~9 (2, 4, 8 launched 14 steps for those 9)
        while (done < max_len) {
            const int upto = done + chunk < max_len ? done + chunk : max_len;
            orp.step0 = 0;
            int rc = odom_launch_steps(c, b, orp, nf, done, upto, G, es, ne);      // group 0's repair launches are timed like the main pass's
            if (rc) return rc;
            done = upto;
            HIP_TRY(c, hipMemcpyAsync(cnt, b->rcount, sizeof(cnt), hipMemcpyDeviceToHost, st));
            HIP_TRY(c, hipStreamSynchronize(st));
            if (cnt[1] == 0) break;
            chunk = chunk < 8 ? chunk * 2 : 8;
        }
    }
    HIP_TRY(c, hipEventRecord(b->rep_ev[1], st));
    HIP_TRY(c, hipMemcpyAsync(rs.data(), b->rstat, sizeof(int) * 4 * n_chains, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, b->rep_ev[0], b->rep_ev[1]) == hipSuccess) R.repair_ms += ms;
    R.pairs_rerun = 0; R.chains_rerun = 0;
    R.unresolved = still_flagged;        // boundaries above the tolerance after the last round (0 unless the round cap ended the loop)
    for (int ch = 0; ch < n_chains; ch++) {
        b->rerun_h[ch] = rs[ch * 4 + 1];
        R.pairs_rerun += rs[ch * 4 + 1]; R.chains_rerun += rs[ch * 4 + 3] > 0 ? 1 : 0;
    }
    return check_launch(c, "boundary validation");
}

static OdomView odom_view(lmono_ctx *c, lmono_scan_batch *b, int n_chains, int lead, int first)
{
    OdomView o{};
    o.n_scans = b->n_scans; o.n_chains = n_chains; o.lead = lead; o.first = first; o.fixed_k = -1; o.chain0 = 0; o.chain1 = n_chains;
    o.lead_full = c->opt[LMONO_OPT_CORR_TILE] == 3 ? c->opt[LMONO_OPT_LEAD_FULL] : -1;     // only the default search thins lead-in pairs
    o.state = b->state; o.corr = b->corr; o.incr = b->incr; o.lm_info = b->lm_info; o.crec = b->crec; o.seed = b->seed;
    o.ws = b->ws; o.repair = 0; o.step0 = 0; o.clist = nullptr; o.rstat = b->rstat; o.rcount = b->rcount;
    o.tol = 1e-9 * (double)c->opt[LMONO_OPT_BOUNDARY_TOL];
    return o;
}

static int odom_run(lmono_ctx *c, lmono_scan_batch *b, int n_chains, int lead, int first, double *incr_d, double *poses_d, bool want_poses, bool validate = true)
{
    if (!c || !b || !b->registered || lead < 0 || first < 0 || first >= b->n_scans) return LMONO_EINVAL;
    const int n = b->n_scans;
    if (n_chains < 1) n_chains = 1;
    if (n_chains > n - first) n_chains = n - first;
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = ensure_odom_ws(c, b, n_chains);
    if (rc) return rc;
    OdomView o = odom_view(c, b, n_chains, lead, first);
    b->last_chains = n_chains; b->last_lead = lead; b->last_first = first;
    int max_steps = 0;
    for (int ch = 0; ch < n_chains; ch++) {
        int s, e;
        chain_bounds(first, n, n_chains, ch, s, e);
        const int begin = s - lead > 0 ? s - lead : 0;
        const int steps = e - begin - 1;
        max_steps = steps > max_steps ? steps : max_steps;
    }
    hipStream_t st = c->stream;
    if (c->n_sets == 0 || c->sets[c->n_sets - 1].odom) c->ev = c->next_set();
    if (!c->ev) { c->err = "hipEventCreate failed"; return LMONO_ENODEV; }
    c->sets[c->n_sets - 1].odom = true;
    HIP_TRY(c, hipEventRecord(c->ev[8], st));
    const int ninit = n > n_chains ? n : n_chains;
    hipLaunchKernelGGL(k_odom_init, dim3((ninit + 255) / 256), dim3(256), 0, st, o);
    const int tile = c->opt[LMONO_OPT_CORR_TILE];
    if (tile != 3) { rc = ensure_grid(c, b); if (rc) return rc; }
    const int G = odom_groups(c, n_chains);
    if (tile) for (int g = 0; g < G; g++) HIP_TRY(c, hipMemsetAsync(b->wl + g * ((size_t)b->chains_cap * kMaxQueries + 1), 0, sizeof(unsigned int), st));
    EvSet &es = c->sets[c->n_sets - 1];
    int ne = 0;
    if (c->opt[LMONO_OPT_LEAD_SEED] > 0 && n_chains > 2 && lead >= 3 && max_steps > 1 && o.ws) {
        // the first step of every chain, then the lead-in states are re-seeded from the neighbouring chains' first results (k_lead_seed_median)
        rc = odom_launch_steps(c, b, o, n_chains, 0, 1, G, &es, &ne);
        if (rc) return rc;
        hipLaunchKernelGGL(k_lead_seed_median, dim3((n_chains + 255) / 256), dim3(256), 0, st, o);
        hipLaunchKernelGGL(k_lead_seed_apply, dim3((n_chains + 255) / 256), dim3(256), 0, st, o);
        rc = odom_launch_steps(c, b, o, n_chains, 1, max_steps, G, &es, &ne);
    } else
        rc = odom_launch_steps(c, b, o, n_chains, 0, max_steps, G, &es, &ne);
    if (rc) return rc;
    es.n_kev = ne;
    // the chained schedule validates itself: every chain's warm start against its predecessor's last increment, repair where they differ
    b->brep = lmono_boundary_report{};
    b->brep.n_chains = n_chains;
    b->resid_h.assign((size_t)n_chains, 0.0); b->rerun_h.assign((size_t)n_chains, 0);      // a run without validation reports zeros, not the previous layout's values
    // (the repair launches carry no per-kernel events: the correspondence / solve sums of lmono_timing_read are the main pass's; the
    // repair's device time is lmono_boundary_report.repair_ms)
    b->validation_pending = !validate;
    if (validate && o.tol > 0.0 && n_chains > 1) { rc = odom_validate(c, b, o, false, true); if (rc) return rc; }
    if (want_poses) hipLaunchKernelGGL(k_pose_prefix, dim3(1), dim3(64), 0, st, (const double *)b->incr, b->poses, first, n);
    HIP_TRY(c, hipEventRecord(c->ev[9], st));
    rc = check_launch(c, "odometry kernels");
    if (rc) return rc;
    if (incr_d) HIP_TRY(c, hipMemcpyAsync(incr_d, b->incr, sizeof(double) * 7 * n, hipMemcpyDeviceToDevice, st));
    if (poses_d) HIP_TRY(c, hipMemcpyAsync(poses_d, b->poses, sizeof(double) * 7 * (n - first), hipMemcpyDeviceToDevice, st));
    return LMONO_OK;
}

extern "C" int lmono_odom_batch_d(lmono_ctx *c, lmono_scan_batch *b, int n_chains, int lead, double *incr_d, double *poses_d)
{
    return odom_run(c, b, n_chains, lead, 0, incr_d, poses_d, poses_d != nullptr);
}

extern "C" int lmono_odom_shard_d(lmono_ctx *c, lmono_scan_batch *b, int n_chains, int lead, int first_owned, double *incr_d)
{
    return odom_run(c, b, n_chains, lead, first_owned, incr_d, nullptr, false);
}

extern "C" int lmono_odom_shard_main_d(lmono_ctx *c, lmono_scan_batch *b, int n_chains, int lead, int first_owned, double *incr_d)
{
    return odom_run(c, b, n_chains, lead, first_owned, incr_d, nullptr, false, false);
}

extern "C" int lmono_odom_shard_validate(lmono_ctx *c, lmono_scan_batch *b, const double *prev_incr_h, double *incr_d, int *changed_last)
{
    if (!c || !b || !b->registered || b->last_chains < 1 || (prev_incr_h && b->last_first < 1)) return LMONO_EINVAL;
    if (!prev_incr_h && !b->validation_pending) return LMONO_EINVAL;     // without an external boundary there is only the deferred validation to run
    HIP_TRY(c, hipSetDevice(c->device));
    OdomView o = odom_view(c, b, b->last_chains, b->last_lead, b->last_first);
    const int n = b->n_scans;
    double before[7], after[7];
    // every copy below is ordered on the context stream (a blocking hipMemcpy would wait for whatever another context has queued on the null stream)
    HIP_TRY(c, hipMemcpyAsync(before, b->incr + (size_t)(n - 1) * 7, sizeof(before), hipMemcpyDeviceToHost, c->stream));
    if (prev_incr_h) HIP_TRY(c, hipMemcpyAsync(b->incr + (size_t)(o.first - 1) * 7, prev_incr_h, sizeof(double) * 7, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    // after lmono_odom_shard_main_d this is the batch's ONE validation: the rank boundary (chain 0) goes through the same repair rounds as the
    // chain boundaries inside the rank instead of a second tail of sequential steps behind them
    const bool first_call = b->validation_pending;
    b->validation_pending = false;
    if (first_call) { b->brep = lmono_boundary_report{}; b->brep.n_chains = b->last_chains; b->resid_h.assign((size_t)b->last_chains, 0.0); b->rerun_h.assign((size_t)b->last_chains, 0); }
    if (o.tol > 0.0 && (prev_incr_h || b->last_chains > 1)) { int rc = odom_validate(c, b, o, prev_incr_h != nullptr, first_call); if (rc) return rc; }
    HIP_TRY(c, hipMemcpyAsync(after, b->incr + (size_t)(n - 1) * 7, sizeof(after), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (changed_last) *changed_last = std::memcmp(before, after, sizeof(before)) != 0 ? 1 : 0;
    if (incr_d) HIP_TRY(c, hipMemcpyAsync(incr_d, b->incr, sizeof(double) * 7 * n, hipMemcpyDeviceToDevice, c->stream));
    return LMONO_OK;
}

extern "C" int lmono_odom_boundary_report(lmono_ctx *c, lmono_scan_batch *b, lmono_boundary_report *rep, double *resid_h, int32_t *rerun_h, int cap)
{
    if (!c || !b || !rep) return LMONO_EINVAL;
    *rep = b->brep;
    const int n = b->brep.n_chains;
    if ((resid_h || rerun_h) && cap < n) { c->err = "boundary_report: output capacity too small"; return LMONO_ECAPACITY; }
    for (int i = 0; i < n && i < (int)b->resid_h.size(); i++) { if (resid_h) resid_h[i] = b->resid_h[i]; if (rerun_h) rerun_h[i] = b->rerun_h[i]; }
    return LMONO_OK;
}

extern "C" int lmono_odom_batch(lmono_ctx *c, lmono_scan_batch *b, int n_chains, int lead, double *incr_h, double *poses_h)
{
    int rc = odom_run(c, b, n_chains, lead, 0, nullptr, nullptr, poses_h != nullptr);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const int n = b->n_scans;
    if (incr_h) HIP_TRY(c, hipMemcpy(incr_h, b->incr, sizeof(double) * 7 * n, hipMemcpyDeviceToHost));
    if (poses_h) HIP_TRY(c, hipMemcpy(poses_h, b->poses, sizeof(double) * 7 * n, hipMemcpyDeviceToHost));
    return LMONO_OK;
}

// ---- online laserOdometry: one scan per call (A-LOAM's node callbacks) -------------------------------------------------------------
// A stream owns a batch of history + 1 fixed-size slots.  Scan t is registered into the slot behind scan t - 1's (the raw points are
// copied into the slot, the rest of the slot is NaN: scanRegistration drops NaN points first), so "last" = the previous slot keeps its
// feature clouds and (line, azimuth bin) index from the previous call; one chain, pinned to the new slot, runs the scan pair from the
// stream's para_q / para_t.  When the slots run out, the last slot's "last" data move to slot 0 and the cycle restarts at slot 1.
struct lmono_odom_stream {
    lmono_ctx *ctx = nullptr;
    lmono_scan_batch *batch = nullptr;
    float *in_d = nullptr;
    int cap_pts = 0, n_slots = 0, slot = 0;
    long long frame = 0;
    double para[8] = { 0, 0, 0, 1, 0, 0, 0, 0 };       // q_last_curr (x y z w), t_last_curr
    double q_w[4] = { 0, 0, 0, 1 }, t_w[3] = { 0, 0, 0 };
};

extern "C" void lmono_odom_stream_destroy(lmono_odom_stream *s)
{
    if (!s) return;
    if (s->batch) lmono_batch_destroy(s->batch);
    if (s->in_d) (void)hipFree(s->in_d);
    delete s;
}

extern "C" lmono_odom_stream *lmono_odom_stream_create(lmono_ctx *c, int max_points, int n_lines, float min_range, int history)
{
    if (!c || max_points <= 0 || history < 1 || (n_lines != 16 && n_lines != 32 && n_lines != 64)) return nullptr;
    if (c->opt[LMONO_OPT_CORR_TILE] != 3) { c->err = "lmono_odom_stream: needs the default correspondence search (LMONO_OPT_CORR_TILE 3)"; return nullptr; }
    if (hipSetDevice(c->device) != hipSuccess) return nullptr;
    lmono_odom_stream *s = new lmono_odom_stream();
    s->ctx = c; s->cap_pts = max_points; s->n_slots = history + 1;
    s->batch = lmono_batch_create(c, s->n_slots, (int64_t)s->n_slots * max_points);
    if (!s->batch || hipMalloc((void **)&s->in_d, (size_t)s->n_slots * max_points * 16) != hipSuccess) { c->err = "lmono_odom_stream_create: allocation failed"; lmono_odom_stream_destroy(s); return nullptr; }
    lmono_scan_batch *b = s->batch;
    b->off_h.resize(s->n_slots + 1);
    for (int i = 0; i <= s->n_slots; i++) b->off_h[i] = (int64_t)i * max_points;
    b->n_scans = s->n_slots; b->total = (int64_t)s->n_slots * max_points; b->max_pts = max_points;
    BatchView &v = b->v;
    v.in = (const float4 *)s->in_d; v.n_scans = s->n_slots; v.scan0 = 0; v.n_lines = n_lines; v.min_range = min_range; v.has_grid = 0;
    bool ok = hipMemcpy(b->off_d, b->off_h.data(), sizeof(int64_t) * (s->n_slots + 1), hipMemcpyHostToDevice) == hipSuccess;
    // empty slots: no points, no features
    ok = ok && hipMemset(s->in_d, 0xff, (size_t)s->n_slots * max_points * 16) == hipSuccess;
    ok = ok && hipMemset(v.n_cloud, 0, sizeof(int) * s->n_slots) == hipSuccess && hipMemset(v.feat_n, 0, sizeof(int) * 4 * s->n_slots) == hipSuccess;
    ok = ok && hipMemset(v.status, 0, sizeof(int) * s->n_slots) == hipSuccess;
    ok = ok && ensure_odom_ws(c, b, 1) == LMONO_OK;
    if (!ok) { c->err = "lmono_odom_stream_create: initialisation failed"; lmono_odom_stream_destroy(s); return nullptr; }
    b->registered = true;
    return s;
}

// everything of slot `from` that a scan pair reads of its "last" scan, copied to slot `to`
static int stream_copy_last(lmono_ctx *c, lmono_scan_batch *b, int from, int to)
{
    hipStream_t st = c->stream;
    BatchView &v = b->v;
    const int64_t of = b->off_h[from], ot = b->off_h[to];
    const size_t P = (size_t)(b->off_h[1] - b->off_h[0]);
#define CP(arr, stride, off_from, off_to) HIP_TRY(c, hipMemcpyAsync((arr) + (off_to), (arr) + (off_from), sizeof(*(arr)) * (stride), hipMemcpyDeviceToDevice, st))
    CP(v.feat_n, 4, (size_t)from * 4, (size_t)to * 4);
    CP(v.n_cloud, 1, (size_t)from, (size_t)to);
    CP(v.status, 1, (size_t)from, (size_t)to);
    CP(v.less_sharp, kMaxLessSharp, (size_t)from * kMaxLessSharp, (size_t)to * kMaxLessSharp);
    CP(v.less_flat, P, (size_t)of, (size_t)ot);
    CP(v.lbc_pts, kMaxLessSharp, (size_t)from * kMaxLessSharp, (size_t)to * kMaxLessSharp);
    CP(v.lbs_pts, P, (size_t)of, (size_t)ot);
    CP(v.lb_start, 2 * (kLineKeys + 1), (size_t)from * 2 * (kLineKeys + 1), (size_t)to * 2 * (kLineKeys + 1));
    CP(v.lb_elev, 2 * 66, (size_t)from * 2 * 66, (size_t)to * 2 * 66);
    CP(v.line_first_ge, 2 * 66, (size_t)from * 2 * 66, (size_t)to * 2 * 66);
    CP(v.line_last_le, 2 * 66, (size_t)from * 2 * 66, (size_t)to * 2 * 66);
#undef CP
    return LMONO_OK;
}

extern "C" int lmono_odom_step(lmono_ctx *c, lmono_odom_stream *s, const float *xyzi, int n_points, int on_device, int use_warm_start,
                               double *q_last_curr, double *t_last_curr, double *q_w_curr, double *t_w_curr, int32_t *info)
{
    if (!c || !s || s->ctx != c || !xyzi || n_points < 0) return LMONO_EINVAL;
    if (n_points > s->cap_pts) { c->err = "lmono_odom_step: more points than the stream's slots hold"; return LMONO_ECAPACITY; }
    if (use_warm_start && (!q_last_curr || !t_last_curr)) return LMONO_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    lmono_scan_batch *b = s->batch;
    hipStream_t st = c->stream;
    int next = s->frame == 0 ? 1 : s->slot + 1;
    if (next >= s->n_slots) {
        int rc = stream_copy_last(c, b, s->slot, 0);
        if (rc) return rc;
        next = 1;
    }
    float *dst = s->in_d + (size_t)next * s->cap_pts * 4;
    if (n_points > 0) HIP_TRY(c, hipMemcpyAsync(dst, xyzi, (size_t)n_points * 16, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
    if (n_points < s->cap_pts) HIP_TRY(c, hipMemsetAsync(dst + (size_t)n_points * 4, 0xff, (size_t)(s->cap_pts - n_points) * 16, st));
    int rc = scanreg_launch(c, b, next, 1, n_points, n_points > 0 ? n_points : 1);
    if (rc) return rc;
    b->feat_h.clear();
    int iters[4] = { 0, 0, 0, 0 };
    if (s->frame > 0) {
        if (use_warm_start) { for (int i = 0; i < 4; i++) s->para[i] = q_last_curr[i]; for (int i = 0; i < 3; i++) s->para[4 + i] = t_last_curr[i]; }
        HIP_TRY(c, hipMemcpyAsync(b->state, s->para, sizeof(double) * 8, hipMemcpyHostToDevice, st));
        OdomView o = odom_view(c, b, 1, 0, 0);
        o.fixed_k = next; o.ws = nullptr; o.lead_full = -1;
        HIP_TRY(c, hipMemsetAsync(b->wl, 0, sizeof(unsigned int), st));
        rc = odom_launch_steps(c, b, o, 1, 0, 1, 1, nullptr, nullptr);
        if (rc) return rc;
        HIP_TRY(c, hipMemcpyAsync(s->para, b->state, sizeof(double) * 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipMemcpyAsync(iters, b->lm_info, sizeof(iters), hipMemcpyDeviceToHost, st));
    }
    int fn[6] = { 0, 0, 0, 0, 0, 0 };
    HIP_TRY(c, hipMemcpyAsync(fn, b->v.n_cloud + next, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(fn + 1, b->v.feat_n + next * 4, sizeof(int) * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(fn + 5, b->v.status + next, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    if (s->frame > 0) {
        // laserOdometry's accumulation: t_w_curr += q_w_curr * t_last_curr; q_w_curr = q_w_curr * q_last_curr
        const double *q = s->para, *t = s->para + 4;
        const double ux = s->q_w[0], uy = s->q_w[1], uz = s->q_w[2], w = s->q_w[3];
        const double uvx = 2.0 * (uy * t[2] - uz * t[1]), uvy = 2.0 * (uz * t[0] - ux * t[2]), uvz = 2.0 * (ux * t[1] - uy * t[0]);
        s->t_w[0] += t[0] + w * uvx + (uy * uvz - uz * uvy);
        s->t_w[1] += t[1] + w * uvy + (uz * uvx - ux * uvz);
        s->t_w[2] += t[2] + w * uvz + (ux * uvy - uy * uvx);
        const double bx = q[0], by = q[1], bz = q[2], bw = q[3];
        const double nw = w * bw - ux * bx - uy * by - uz * bz, nx = w * bx + ux * bw + uy * bz - uz * by;
        const double ny = w * by + uy * bw + uz * bx - ux * bz, nz = w * bz + uz * bw + ux * by - uy * bx;
        s->q_w[0] = nx; s->q_w[1] = ny; s->q_w[2] = nz; s->q_w[3] = nw;
    }
    if (q_last_curr) for (int i = 0; i < 4; i++) q_last_curr[i] = s->para[i];
    if (t_last_curr) for (int i = 0; i < 3; i++) t_last_curr[i] = s->para[4 + i];
    if (q_w_curr) for (int i = 0; i < 4; i++) q_w_curr[i] = s->q_w[i];
    if (t_w_curr) for (int i = 0; i < 3; i++) t_w_curr[i] = s->t_w[i];
    if (info) { for (int i = 0; i < 6; i++) info[i] = fn[i]; info[6] = (iters[0] << 8) | iters[1]; info[7] = iters[3]; }
    s->slot = next; s->frame++;
    if (fn[5] & kStatusRingOverflow) { c->err = "lmono_odom_step: a ring holds more than LMONO_RING_CAP points"; return LMONO_ESCAN; }
    return LMONO_OK;
}

extern "C" int lmono_odom_stream_scan(lmono_odom_stream *s, lmono_scan_batch **batch, int *scan)
{
    if (!s || s->frame == 0) return LMONO_EINVAL;
    if (batch) *batch = s->batch;
    if (scan) *scan = s->slot;
    return LMONO_OK;
}

extern "C" int lmono_odom_correspond(lmono_ctx *c, lmono_scan_batch *b, int scan, const double q[4], const double t[3],
                                     int32_t *corr_h, int cap)
{
    if (!c || !b || !b->registered || scan < 1 || scan >= b->n_scans || !q || !t || !corr_h) return LMONO_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    double x[8] = { q[0], q[1], q[2], q[3], t[0], t[1], t[2], 0.0 };
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(b->xq, x, sizeof(x), hipMemcpyHostToDevice));
    int fn[4];
    HIP_TRY(c, hipMemcpy(fn, b->v.feat_n + scan * 4, sizeof(fn), hipMemcpyDeviceToHost));
    const int nq = fn[0] + fn[2];
    if (nq > cap) { c->err = "odom_correspond: output capacity too small"; return LMONO_ECAPACITY; }
    OdomView o{};
    o.n_scans = b->n_scans; o.n_chains = 1; o.lead = 0; o.fixed_k = scan; o.chain0 = 0; o.chain1 = 1; o.lead_full = -1;
    o.state = b->xq; o.corr = b->corr_pair; o.incr = nullptr; o.lm_info = nullptr; o.crec = b->crec_pair; o.seed = nullptr;
    int rc;
    if (c->opt[LMONO_OPT_CORR_TILE] != 3) { rc = ensure_grid(c, b); if (rc) return rc; }
    if (c->opt[LMONO_OPT_CORR_TILE]) {
        rc = ensure_odom_ws(c, b, 1);
        if (rc) return rc;
        HIP_TRY(c, hipMemsetAsync(b->wl, 0, sizeof(unsigned int), c->stream));
        hipLaunchKernelGGL(k_corr_flat, dim3(8 * kCfBlocks), dim3(kCfT), 0, c->stream, b->v, o, 0, 0, b->wl, c->opt[LMONO_OPT_DEFER_EVERY], c->stats_d);
        hipLaunchKernelGGL(k_correspond_list, dim3(kListGrid), dim3(256), 0, c->stream, b->v, o, 0, 0, (const unsigned int *)b->wl, c->stats_d);
    }
#ifdef LMONO_DIAG_SEARCH
    else
        hipLaunchKernelGGL(k_correspond, dim3(8 * kCorrBlocks), dim3(256), 0, c->stream, b->v, o, 0, 0);
#endif
    rc = check_launch(c, "k_correspond");
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (nq > 0) HIP_TRY(c, hipMemcpy(corr_h, b->corr_pair, sizeof(int) * 4 * nq, hipMemcpyDeviceToHost));
    return nq;
}

extern "C" int lmono_pose_prefix_d(lmono_ctx *c, const double *incr_d, int first, int n, double *poses_d)
{
    if (!c || !incr_d || !poses_d || first < 0 || n <= first) return LMONO_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(k_pose_prefix, dim3(1), dim3(64), 0, c->stream, incr_d, poses_d, first, n);
    return check_launch(c, "k_pose_prefix");
}

extern "C" int lmono_pose_rebase_d(lmono_ctx *c, const double *bases_d, int n_bases, double *poses_d, int n)
{
    if (!c || !poses_d || n <= 0 || n_bases < 0 || (n_bases > 0 && !bases_d)) return LMONO_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(k_pose_rebase, dim3((n + 255) / 256), dim3(256), 0, c->stream, bases_d, n_bases, poses_d, n);
    return check_launch(c, "k_pose_rebase");
}

// ---- BA factors -------------------------------------------------------------------------------------------------
extern "C" int lmono_factor_eval_blocks_d(lmono_ctx *c, int kind, int count, const double *params_d, const double *consts_d,
                                          const double *info_d, double *r_d, double *J_d, const unsigned char *block_mask_d)
{
    if (!c || kind < 0 || kind > 3 || count < 0 || !params_d || !consts_d || !info_d || !r_d) return LMONO_EINVAL;
    if (count == 0) return LMONO_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(k_factor_eval, dim3((count + 63) / 64), dim3(64), 0, c->stream, kind, count, params_d, consts_d, info_d, r_d, J_d, block_mask_d);
    return check_launch(c, "k_factor_eval");
}

extern "C" int lmono_factor_eval_d(lmono_ctx *c, int kind, int count, const double *params_d, const double *consts_d,
                                   const double *info_d, double *r_d, double *J_d)
{
    return lmono_factor_eval_blocks_d(c, kind, count, params_d, consts_d, info_d, r_d, J_d, nullptr);
}

extern "C" int lmono_factor_eval_blocks(lmono_ctx *c, int kind, int count, const double *params_h, const double *consts_h,
                                        const double *info_h, double *r_h, double *J_h, const unsigned char *block_mask_h)
{
    if (!c || kind < 0 || kind > 3 || count < 0 || !params_h || !consts_h || !info_h || !r_h) return LMONO_EINVAL;
    if (count == 0) return LMONO_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const FactorDims d = factor_dims(kind);
    double *p = nullptr, *cn = nullptr, *inf = nullptr, *r = nullptr, *J = nullptr;
    unsigned char *mk = nullptr;
    int rc = LMONO_OK;
    auto cleanup = [&]() { (void)hipFree(p); (void)hipFree(cn); (void)hipFree(inf); (void)hipFree(r); (void)hipFree(J); (void)hipFree(mk); };
#define TRYF(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { c->err = std::string(#expr) + ": " + hipGetErrorString(e_); cleanup(); return LMONO_ENODEV; } } while (0)
    TRYF(hipMalloc((void **)&p, sizeof(double) * d.np * count));
    TRYF(hipMalloc((void **)&cn, sizeof(double) * d.nc * count));
    TRYF(hipMalloc((void **)&inf, sizeof(double) * d.ni));
    TRYF(hipMalloc((void **)&r, sizeof(double) * d.nr * count));
    if (J_h) TRYF(hipMalloc((void **)&J, sizeof(double) * d.nj * count));
    if (J_h && block_mask_h) {
        TRYF(hipMalloc((void **)&mk, (size_t)count));
        TRYF(hipMemcpy(mk, block_mask_h, (size_t)count, hipMemcpyHostToDevice));
        // blocks the caller did not ask for keep the caller's bytes: start from the caller's J
        TRYF(hipMemcpy(J, J_h, sizeof(double) * d.nj * count, hipMemcpyHostToDevice));
    }
    TRYF(hipMemcpy(p, params_h, sizeof(double) * d.np * count, hipMemcpyHostToDevice));
    TRYF(hipMemcpy(cn, consts_h, sizeof(double) * d.nc * count, hipMemcpyHostToDevice));
    TRYF(hipMemcpy(inf, info_h, sizeof(double) * d.ni, hipMemcpyHostToDevice));
    rc = lmono_factor_eval_blocks_d(c, kind, count, p, cn, inf, r, J, mk);
    if (rc == LMONO_OK) {
        TRYF(hipStreamSynchronize(c->stream));
        TRYF(hipMemcpy(r_h, r, sizeof(double) * d.nr * count, hipMemcpyDeviceToHost));
        if (J_h) TRYF(hipMemcpy(J_h, J, sizeof(double) * d.nj * count, hipMemcpyDeviceToHost));
    }
#undef TRYF
    cleanup();
    return rc;
}

extern "C" int lmono_factor_eval(lmono_ctx *c, int kind, int count, const double *params_h, const double *consts_h,
                                 const double *info_h, double *r_h, double *J_h)
{
    return lmono_factor_eval_blocks(c, kind, count, params_h, consts_h, info_h, r_h, J_h, nullptr);
}

// ---- BA window solve ---------------------------------------------------------------------------------------------
struct lmono_ba_batch {
    lmono_ctx *ctx = nullptr;
    // ONE device allocation holds every array of the problem: [uploaded arrays | scratch that starts zeroed], each 256-B aligned, laid out
    // anew by every ba_fill; ONE pinned host buffer stages the uploaded part.  A frame loop (lmono_ba_batch_update per frame) therefore costs
    // one H2D copy and one memset per frame instead of 21 pageable copies and 6 memsets, and allocates nothing in steady state.
    char *blob = nullptr; size_t blob_cap = 0;
    char *stage = nullptr; size_t stage_cap = 0;
    BaBatch v{};
    int n_windows = 0, total_feat = 0, total_obs = 0;
    int cluster = 1;            // workgroups per window of the last fill (the scratch is sized for it)
    bool big = false;           // a window of the last fill holds more than kBaLdsFeat features: the kBig kernels (per-feature vectors in an L2 scratch)
    bool flags_clean = false;   // the cluster's flag words and the failure flag are zero (a fill zeroes them; a solve dirties them)
    double *poses0 = nullptr, *ex0 = nullptr, *invd0 = nullptr;   // initial state for lmono_ba_batch_reset
    // the state (poses | ex | inverse depths: neighbours in the blob) as it was before the last CLUSTER solve, and that solve's iteration cap: a cluster whose
    // workgroups were not all resident gives up (bounded polls) and lmono_ba_batch_read runs the solve again with one workgroup per window from here
    // -- same bytes by construction
    char *pre = nullptr; size_t pre_bytes = 0;
    int retries = 0;            // cluster solves that had to be run again (diagnosis; LMONO_BA_TEST_FAIL exercises the path)
};

// the arrays of one ba_fill: laid out first (add), then staged / placed in one go (commit)
struct BaPack {
    struct Item { void **dst; const void *src; size_t bytes, off; };
    std::vector<Item> items;
    size_t up = 0, zero = 0;
    template <typename T> void add(T *&dst, const T *src, size_t count)
    {
        const size_t bytes = (count > 0 ? count : 1) * sizeof(T), al = (bytes + 255) & ~(size_t)255;
        items.push_back({ (void **)&dst, (const void *)src, src ? count * sizeof(T) : 0, src ? up : zero });
        (src ? up : zero) += al;
    }
    int commit(lmono_ctx *c, lmono_ba_batch *b)
    {
        const size_t total = up + zero;
        if (b->blob_cap < total) {
            if (b->blob) HIP_TRY(c, hipFree(b->blob));
            b->blob = nullptr; b->blob_cap = 0;
            const size_t cap = total + total / 2;
            HIP_TRY(c, hipMalloc((void **)&b->blob, cap));
            b->blob_cap = cap;
        }
        if (b->stage_cap < up) {
            if (b->stage) HIP_TRY(c, hipHostFree(b->stage));
            b->stage = nullptr; b->stage_cap = 0;
            const size_t cap = up + up / 2;
            HIP_TRY(c, hipHostMalloc((void **)&b->stage, cap, hipHostMallocDefault));
            b->stage_cap = cap;
        }
        for (const Item &it : items) {
            if (it.src) { memcpy(b->stage + it.off, it.src, it.bytes); *it.dst = b->blob + it.off; }
            else *it.dst = b->blob + up + it.off;
        }
        if (up) HIP_TRY(c, hipMemcpyAsync(b->blob, b->stage, up, hipMemcpyHostToDevice, c->stream));
        if (zero) HIP_TRY(c, hipMemsetAsync(b->blob + up, 0, zero, c->stream));
        b->flags_clean = true;
        return LMONO_OK;
    }
};

extern "C" void lmono_ba_batch_destroy(lmono_ba_batch *b)
{
    if (!b) return;
    if (b->retries > 0 && getenv("LMONO_BA_REPORT_RETRIES")) fprintf(stderr, "[lmono] lmono_ba_batch: %d cluster solve(s) gave up and were run again with one workgroup per window\n", b->retries);
    if (b->ctx) (void)hipStreamSynchronize(b->ctx->stream);
    if (b->blob) (void)hipFree(b->blob);
    if (b->stage) (void)hipHostFree(b->stage);
    delete b;
}

// validate the descriptor, build the pair-ordered tables and (re)load every device array of the batch
static int ba_fill(lmono_ctx *c, lmono_ba_batch *b, const lmono_ba_desc *d)
{
    if (!d || d->n_windows <= 0 || !d->feat_off || !d->obs_off || !d->flags || !d->poses || !d->ex) { c->err = "lmono_ba_batch: bad descriptor"; return LMONO_EINVAL; }
    if (!d->laser_info || !d->mono_info || !d->prior_w || !d->laser_consts || !d->prior_T || (d->feat_off[d->n_windows] > 0 && !d->inv_depth)) { c->err = "lmono_ba_batch_create: a descriptor array is NULL"; return LMONO_EINVAL; }
    if (d->obs_off[d->n_windows] > 0 && (!d->obs_feat || !d->obs_i || !d->obs_j || !d->obs_pts)) { c->err = "lmono_ba_batch_create: observation arrays are NULL"; return LMONO_EINVAL; }
    HIP_TRY(c, hipSetDevice(c->device));
    const int W = d->n_windows;
    const int TF = d->feat_off[W], TO = d->obs_off[W];
    for (int w = 0; w < W; w++) {
        if (d->feat_off[w + 1] - d->feat_off[w] > kBaMaxFeat) { c->err = "lmono_ba_batch_create: more than LMONO_BA_MAX_FEATURES (" + std::to_string(kBaMaxFeat) + ") features in a window"; return LMONO_ECAPACITY; }
        if (d->flags[4 * w] < 2 || d->flags[4 * w] > kBaMaxPoses) { c->err = "lmono_ba_batch_create: n_poses must be 2..11"; return LMONO_EINVAL; }
    }
    // first observation of every feature: observations must be grouped by (window, feature) in ascending order
    std::vector<int> fo((size_t)TF + 1, 0);
    {
        int o = 0;
        for (int w = 0; w < W; w++) {
            const int f0 = d->feat_off[w], f1 = d->feat_off[w + 1], oe = d->obs_off[w + 1];
            o = d->obs_off[w];
            for (int f = f0; f < f1; f++) {
                fo[f] = o;
                while (o < oe && d->obs_feat[o] == f - f0) {
                    const int np = d->flags[4 * w];
                    if (d->obs_i[o] < 0 || d->obs_i[o] >= np || d->obs_j[o] < 0 || d->obs_j[o] >= np || d->obs_i[o] == d->obs_j[o]) { c->err = "lmono_ba_batch_create: bad observation frame"; return LMONO_EINVAL; }
                    o++;
                }
            }
            if (o != oe) { c->err = "lmono_ba_batch_create: observations are not grouped by feature"; return LMONO_EINVAL; }
        }
        fo[TF] = TO;
    }
    // frame pairs of every window (descending observation count: the waves take them from a work counter) and the
    // pair-ordered observation list
    std::vector<int> pair_off((size_t)W + 1, 0), pair_ij, pair_slot, pobs_off((size_t)W + 1, 0), slot_info, anchor((size_t)TF, -1);
    std::vector<double> slot_pts;
    std::vector<unsigned short> seg_tab;            // segments (<= 16 slots of one pair): pair (window-local) | index inside the pair << 7
    std::vector<int> seg_off((size_t)W + 1, 0), pair_seg, n_multi((size_t)W, 0);
    std::vector<int> slot_obs;                      // the observation (host order) behind every slot: its scratch record is indexed by observation,
                                                    // so a feature's records are contiguous for the per-feature sums of k_ba_solve
    slot_info.reserve((size_t)TO); slot_obs.reserve((size_t)TO); slot_pts.reserve((size_t)TO * 4); seg_tab.reserve((size_t)TO / 8 + (size_t)W * 16);
    pair_ij.reserve((size_t)W * 64); pair_slot.reserve((size_t)W * 65); pair_seg.reserve((size_t)W * 65);
    for (int w = 0; w < W; w++) {
        pair_off[w] = (int)pair_ij.size(); pobs_off[w] = (int)slot_info.size(); seg_off[w] = (int)seg_tab.size();
        const int f0 = d->feat_off[w], f1 = d->feat_off[w + 1];
        for (int f = f0; f < f1; f++) if (fo[f + 1] > fo[f]) anchor[f] = d->obs_i[fo[f]];
        if (d->flags[4 * w + 3]) {   // use_mono == 0: the projection factors are not part of the problem
            // counting sort of the window's observations by frame pair (ascending observation index inside a pair), pairs by (observer j, anchor i) ascending
            // (round 6: the order in which a one-workgroup solve can add a pair's tile into H_pp as soon as it is formed -- see ba_linearise_lds; rounds 4-5
            // sorted by descending size for the waves' work counter, which the 16-slot segments made pointless) -- no per-window allocations: a lock-step
            // batch of Estimators fills hundreds of windows per frame
            constexpr int kKeys = kBaMaxPoses * kBaMaxPoses;
            int cnt[kKeys], base[kKeys], order[kKeys], local_of[kKeys], n_keys = 0;
            for (int key = 0; key < kKeys; key++) cnt[key] = 0;
            const int o0w = d->obs_off[w], o1w = d->obs_off[w + 1];
            for (int o = o0w; o < o1w; o++) cnt[d->obs_i[o] * kBaMaxPoses + d->obs_j[o]]++;
            for (int j = 0; j < kBaMaxPoses; j++) for (int i = 0; i < kBaMaxPoses; i++) { const int key = i * kBaMaxPoses + j; if (cnt[key] > 0) order[n_keys++] = key; }
            const size_t slot0 = slot_info.size();
            int run = 0;
            for (int local = 0; local < n_keys; local++) {
                const int key = order[local];
                local_of[key] = local; base[key] = run;
                pair_ij.push_back((key / kBaMaxPoses) | ((key % kBaMaxPoses) << 8));
                pair_slot.push_back(run);
                pair_seg.push_back((int)seg_tab.size() - seg_off[w]);
                const int nseg = (cnt[key] + kBaSeg - 1) / kBaSeg;
                for (int sidx = 0; sidx < nseg; sidx++) seg_tab.push_back((unsigned short)(local | (sidx << 7)));
                if (nseg > 1) n_multi[w]++;
                run += cnt[key];
            }
            slot_obs.resize(slot0 + (size_t)run); slot_info.resize(slot0 + (size_t)run); slot_pts.resize((slot0 + (size_t)run) * 4);
            for (int o = o0w; o < o1w; o++) {
                const int key = d->obs_i[o] * kBaMaxPoses + d->obs_j[o];
                const size_t sl = slot0 + (size_t)base[key]++;
                slot_obs[sl] = o;
                slot_info[sl] = d->obs_feat[o] | (local_of[key] << 16);
                memcpy(&slot_pts[sl * 4], &d->obs_pts[(size_t)o * 4], 4 * sizeof(double));
            }
        }
        pair_slot.push_back((int)slot_info.size() - pobs_off[w]);   // n_pairs + 1 entries per window
        pair_seg.push_back((int)seg_tab.size() - seg_off[w]);
    }
    pair_off[W] = (int)pair_ij.size(); pobs_off[W] = (int)slot_info.size(); seg_off[W] = (int)seg_tab.size();
    b->ctx = c; b->n_windows = W; b->total_feat = TF; b->total_obs = TO;
    BaBatch &v = b->v;
    v.n_windows = W; v.max_iter = 30;
    double info[42];
    memcpy(info, d->laser_info, 36 * sizeof(double)); memcpy(info + 36, d->mono_info, 4 * sizeof(double)); memcpy(info + 40, d->prior_w, 2 * sizeof(double));
    int *feat_off = nullptr, *obs_off = nullptr, *flags = nullptr, *anch = nullptr, *poff = nullptr, *pij = nullptr, *psoff = nullptr, *sinfo_d = nullptr, *pslot_d = nullptr;
    int *fobs_d = nullptr, *oslot_d = nullptr, *segoff_d = nullptr, *pseg_d = nullptr, *nmulti_d = nullptr;
    unsigned short *segtab_d = nullptr;
    const unsigned short uzero = 0;
    double *spts_d = nullptr, *laser = nullptr, *prior = nullptr, *infod = nullptr;
    const int izero = 0; const double dzero = 0.0;       // a present (non-NULL) source for arrays that may be empty
    BaPack pk;
    pk.add(feat_off, d->feat_off, (size_t)W + 1); pk.add(obs_off, d->obs_off, (size_t)W + 1);
    std::vector<double> zsum((size_t)W * 6, 0.0);
    pk.add(flags, d->flags, (size_t)W * 4);
    pk.add(v.fail, &izero, (size_t)1); pk.add(v.summary, (const double *)zsum.data(), (size_t)W * 6);        // [failure flag | summaries | poses | ex | inverse depths]: the results, one read-back
    pk.add(v.poses, d->poses, (size_t)W * kBaMaxPoses * 7);
    pk.add(v.ex, d->ex, (size_t)W * 7); pk.add(v.inv_depth, TF ? d->inv_depth : &dzero, (size_t)TF);
    pk.add(anch, TF ? anchor.data() : &izero, (size_t)TF);
    pk.add(poff, pair_off.data(), (size_t)W + 1); pk.add(pij, pair_ij.empty() ? &izero : pair_ij.data(), pair_ij.size());
    pk.add(psoff, pobs_off.data(), (size_t)W + 1); pk.add(sinfo_d, slot_info.empty() ? &izero : slot_info.data(), slot_info.size());
    pk.add(spts_d, slot_pts.empty() ? &dzero : slot_pts.data(), slot_pts.size()); pk.add(pslot_d, pair_slot.data(), pair_slot.size());
    pk.add(laser, d->laser_consts, (size_t)W * 10 * 24); pk.add(prior, d->prior_T, (size_t)W * 16);
    pk.add(infod, (const double *)info, (size_t)42);
    pk.add(b->poses0, d->poses, (size_t)W * kBaMaxPoses * 7); pk.add(b->ex0, d->ex, (size_t)W * 7);
    pk.add(b->invd0, TF ? d->inv_depth : &dzero, (size_t)TF);
    pk.add(fobs_d, (const int *)fo.data(), (size_t)TF + 1); pk.add(oslot_d, slot_obs.empty() ? &izero : slot_obs.data(), slot_obs.size());
    pk.add(segoff_d, (const int *)seg_off.data(), (size_t)W + 1); pk.add(pseg_d, (const int *)pair_seg.data(), pair_seg.size());
    pk.add(nmulti_d, (const int *)n_multi.data(), (size_t)W); pk.add(segtab_d, seg_tab.empty() ? &uzero : seg_tab.data(), seg_tab.size());
    pk.add(v.obsc, (const double *)nullptr, (size_t)TO * kBaObsRec);
    b->big = false;
    for (int w = 0; w < W; w++) if (d->feat_off[w + 1] - d->feat_off[w] > kBaLdsFeat) b->big = true;
    v.feat_cap = b->big ? kBaMaxFeat : kBaLdsFeat;
    pk.add(v.hpd, (const double *)nullptr, (size_t)W * v.feat_cap * kBaPS);
    pk.add(v.bigv, (const double *)nullptr, b->big ? (size_t)W * 8 * kBaMaxFeat : (size_t)1);
    // workgroups per window: several when the batch leaves most of the chip idle (every workgroup of a window must be resident while it polls: at most
    // half the CUs).  LMONO_BA_CLUSTER = 1 / 2 / 4 forces it (measurement switch); the results do not depend on it, bit for bit.
    {
        static const int env = [] { const char *e = getenv("LMONO_BA_CLUSTER"); return e ? atoi(e) : 0; }();        // measurement switch
        const int forced = c->opt[LMONO_OPT_BA_CLUSTER] > 0 ? c->opt[LMONO_OPT_BA_CLUSTER] : env;
        int K = forced > 0 ? forced : kBaMaxK;
        // (a window of few segments gains nothing from the last doubling and pays its hand-offs: the Estimator's own windows, ~45 segments, run 0.5 % faster
        // at 4 than at 8, the 110-segment bench window 5 % slower; the bytes are the same either way)
        if (forced <= 0 && (int)seg_tab.size() < 64 * W) K = 4;
        if (K > kBaMaxK) K = kBaMaxK;
        if (K == 3) K = 2; else if (K > 4 && K < 8) K = 4;
        while (K > 1 && ((W + 7) / 8) * 8 * K > c->cluster_budget) K >>= 1;      // (256 CUs: 128 workgroups -- 8 for up to 16 windows ... 1 above 64)
        b->cluster = K;
    }
    pk.add(v.pairdat, (const double *)nullptr, (size_t)b->cluster * pair_ij.size() * kBaPairRec);
    pk.add(v.mbox, (const double *)nullptr, (size_t)W * kBaMbox);
    pk.add(v.bar, (const unsigned int *)nullptr, (size_t)W * kBaBar);
    pk.add(v.hred, (const double *)nullptr, b->cluster > 1 ? (size_t)W * kBaHred : (size_t)1);
    pk.add(v.fdg, (const double *)nullptr, b->cluster > 1 ? (size_t)W * 2 * v.feat_cap : (size_t)1);
    v.n_pairs_total = (int)pair_ij.size();
    pk.add(v.pairH, (const double *)nullptr, (seg_tab.size() + pair_ij.size() + (size_t)W) * kBaPairTile);
    pk.add(v.gprog, (const int *)nullptr, (size_t)W * kBaGprog);
    pk.add(v.cpart, (const double *)nullptr, seg_tab.size());
    pk.add(v.cand, (const double *)nullptr, (size_t)W * v.feat_cap);
    {
        auto al = [](size_t bytes) { return ((bytes ? bytes : 8) + 255) & ~(size_t)255; };
        b->pre_bytes = al(sizeof(double) * (size_t)W * kBaMaxPoses * 7) + al(sizeof(double) * (size_t)W * 7) + al(sizeof(double) * (size_t)TF);
        pk.add(b->pre, (const char *)nullptr, b->pre_bytes);
    }
    // everything is staged in the batch's pinned buffer: the vectors above may go, and nothing waits here
    { const int rc = pk.commit(c, b); if (rc) { c->err = "lmono_ba_batch_create: device allocation / upload failed"; return LMONO_ENOMEM; } }
    v.feat_off = feat_off; v.obs_off = obs_off; v.flags = flags; v.feat_anchor = anch;
    v.pair_off = poff; v.pair_ij = pij; v.pobs_off = psoff; v.slot_info = sinfo_d; v.slot_pts = spts_d; v.pair_slot = pslot_d;
    v.laser_consts = laser; v.prior_T = prior; v.info = infod;
    v.feat_obs_off = fobs_d; v.slot_obs = oslot_d;
    v.seg_off = segoff_d; v.seg_tab = segtab_d; v.pair_seg = pseg_d; v.n_multi = nmulti_d;
    v.lds_ok = 1;
    for (int o = 0; o < TO && v.lds_ok; o++) if (d->obs_i[o] >= d->obs_j[o]) v.lds_ok = 0;
    v.blob_lo = b->blob; v.blob_hi = b->blob + pk.up + pk.zero;      // (the arrays of THIS fill: what lies behind them in a larger, re-used allocation is out of bounds too)
    return LMONO_OK;
}

extern "C" lmono_ba_batch *lmono_ba_batch_create(lmono_ctx *c, const lmono_ba_desc *d)
{
    if (!c) return nullptr;
    lmono_ba_batch *b = new lmono_ba_batch();
    b->ctx = c;
    if (ba_fill(c, b, d) != LMONO_OK) { lmono_ba_batch_destroy(b); return nullptr; }
    return b;
}

// Load another set of windows into an existing batch (the Estimator's next frame): device arrays are reused where they are large
// enough, so a steady-state frame loop allocates nothing.  On error the batch holds no valid problem until the next update.
extern "C" int lmono_ba_batch_update(lmono_ctx *c, lmono_ba_batch *b, const lmono_ba_desc *d)
{
    if (!c || !b || b->ctx != c) return LMONO_EINVAL;
    HIP_TRY(c, hipStreamSynchronize(c->stream));      // a solve of the previous problem may still read the arrays
    const int rc = ba_fill(c, b, d);
    if (rc != LMONO_OK) { b->n_windows = 0; b->total_feat = 0; b->total_obs = 0; }     // no problem: solve / reset / read refuse
    return rc;
}

static int ba_launch_single(lmono_ctx *c, lmono_ba_batch *b)
{
    if (b->big) hipLaunchKernelGGL((k_ba_solve<false, true>), dim3(b->n_windows), dim3(kBaT), 0, c->stream, b->v, 1, 0);
    else hipLaunchKernelGGL((k_ba_solve<false, false>), dim3(b->n_windows), dim3(kBaT), 0, c->stream, b->v, 1, 0);          // its LDS is static (g_ba_lds)
    return check_launch(c, "k_ba_solve");
}

extern "C" int lmono_ba_solve(lmono_ctx *c, lmono_ba_batch *b, int max_iterations)
{
    if (!c || !b || max_iterations < 0) return LMONO_EINVAL;
    if (b->n_windows <= 0) { c->err = "lmono_ba_solve: the batch holds no problem (failed update)"; return LMONO_EINVAL; }
    HIP_TRY(c, hipSetDevice(c->device));
    b->v.max_iter = max_iterations;
    if (b->cluster > 1) {
        // the flag words start at zero in every launch (the first solve after a fill finds them zeroed with the rest of the scratch)
        if (!b->flags_clean) {
            HIP_TRY(c, hipMemsetAsync(b->v.bar, 0, sizeof(unsigned int) * (size_t)b->n_windows * kBaBar, c->stream));
            HIP_TRY(c, hipMemsetAsync(b->v.fail, 0, sizeof(int), c->stream));
        }
        b->flags_clean = false;
        // the state this solve starts from, for the one-workgroup re-run of a cluster that was not resident (lmono_ba_batch_read)
        if ((const char *)(b->v.inv_depth + b->total_feat) - (const char *)b->v.poses <= (ptrdiff_t)b->pre_bytes && (const char *)b->v.ex > (const char *)b->v.poses)
            HIP_TRY(c, hipMemcpyAsync(b->pre, b->v.poses, (size_t)((const char *)(b->v.inv_depth + b->total_feat) - (const char *)b->v.poses), hipMemcpyDeviceToDevice, c->stream));
        static const int test_fail = [] { const char *e = getenv("LMONO_BA_TEST_FAIL"); return e ? atoi(e) : 0; }();   // test hook: the cluster gives up at its first poll
        if (test_fail) HIP_TRY(c, hipMemsetAsync(b->v.fail, 1, 1, c->stream));
        // bit 0 LMONO_BA_SPREAD (test hook: a window's workgroups on different XCDs), bit 1 LMONO_BA_SHARE_SUMS (measurement switch: the cluster shares the
        // leader's ordered sums -- byte-identical, measured slower, off)
        static const int spread = [] { const char *e = getenv("LMONO_BA_SPREAD"); const char *h = getenv("LMONO_BA_SHARE_SUMS"); return ((e && atoi(e)) ? 1 : 0) | ((h && atoi(h)) ? 2 : 0); }();
        const dim3 grid(((b->n_windows + 7) / 8) * 8 * b->cluster);
        if (b->big) hipLaunchKernelGGL((k_ba_solve<true, true>), grid, dim3(kBaT), 0, c->stream, b->v, b->cluster, spread);
        else hipLaunchKernelGGL((k_ba_solve<true, false>), grid, dim3(kBaT), 0, c->stream, b->v, b->cluster, spread);
        return check_launch(c, "k_ba_solve");
    }
    return ba_launch_single(c, b);
}

// Diagnostic: the bounds-checked build's record (-DLMONO_BOUNDS, lmono_amd/csrc/ba_solve.hip ba_chk): out[0] accesses of k_ba_solve outside the batch's
// allocation since the library was loaded, out[1] source line of the first, out[2] its byte offset from the allocation's start, out[3] its block.
// The product build has no checks and answers LMONO_EINVAL.
extern "C" int lmono_debug_bounds(lmono_ctx *c, unsigned long long *out4)
{
    if (!c || !out4) return LMONO_EINVAL;
#ifdef LMONO_BOUNDS
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_ba_oob), 4 * sizeof(unsigned long long)));
    return LMONO_OK;
#else
    c->err = "lmono_debug_bounds: this build carries no bounds checks (build with -DLMONO_BOUNDS)";
    return LMONO_EINVAL;
#endif
}

extern "C" int lmono_ba_batch_reset(lmono_ctx *c, lmono_ba_batch *b)
{
    if (!c || !b) return LMONO_EINVAL;
    if (b->n_windows <= 0) { c->err = "lmono_ba_batch_reset: the batch holds no problem (failed update)"; return LMONO_EINVAL; }
    HIP_TRY(c, hipMemcpyAsync(b->v.poses, b->poses0, sizeof(double) * (size_t)b->n_windows * kBaMaxPoses * 7, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(b->v.ex, b->ex0, sizeof(double) * (size_t)b->n_windows * 7, hipMemcpyDeviceToDevice, c->stream));
    if (b->total_feat > 0) HIP_TRY(c, hipMemcpyAsync(b->v.inv_depth, b->invd0, sizeof(double) * (size_t)b->total_feat, hipMemcpyDeviceToDevice, c->stream));
    return LMONO_OK;
}

static int ba_read(lmono_ctx *c, lmono_ba_batch *b, double *poses_h, double *ex_h, double *inv_depth_h, double *summary_h, bool may_retry);
extern "C" int lmono_ba_batch_read(lmono_ctx *c, lmono_ba_batch *b, double *poses_h, double *ex_h, double *inv_depth_h, double *summary_h)
{
    return ba_read(c, b, poses_h, ex_h, inv_depth_h, summary_h, true);
}
static int ba_read(lmono_ctx *c, lmono_ba_batch *b, double *poses_h, double *ex_h, double *inv_depth_h, double *summary_h, bool may_retry)
{
    if (!c || !b) return LMONO_EINVAL;
    if (b->n_windows <= 0) { c->err = "lmono_ba_batch_read: the batch holds no problem (failed update)"; return LMONO_EINVAL; }
    // (failure flag | summaries | poses | ex | inverse depths) are neighbours in the batch's allocation: a small batch -- the Estimator's one window per
    // frame -- comes back as ONE copy into the batch's pinned staging buffer (free between an upload and the next) instead of five copies into pageable
    // memory, each of which the runtime stages and waits for on its own.
    const size_t w = (size_t)b->n_windows;
    const char *lo = (const char *)b->v.fail;
    const size_t bytes = (size_t)((const char *)(b->v.inv_depth + b->total_feat) - lo);
    int failed = 0;
    if ((const char *)b->v.summary > lo && (const char *)b->v.poses > (const char *)b->v.summary && (const char *)b->v.ex > (const char *)b->v.poses &&
        (const char *)b->v.inv_depth > (const char *)b->v.ex && bytes + 256 <= b->stage_cap && bytes <= ((size_t)256 << 10)) {
        char *sa = b->stage;
        HIP_TRY(c, hipMemcpyAsync(sa, lo, bytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));      // stream-ordered behind the solve; nothing goes through the null stream
        if (poses_h) memcpy(poses_h, sa + ((const char *)b->v.poses - lo), sizeof(double) * w * kBaMaxPoses * 7);
        if (ex_h) memcpy(ex_h, sa + ((const char *)b->v.ex - lo), sizeof(double) * w * 7);
        if (inv_depth_h && b->total_feat > 0) memcpy(inv_depth_h, sa + ((const char *)b->v.inv_depth - lo), sizeof(double) * (size_t)b->total_feat);
        if (summary_h) memcpy(summary_h, sa + ((const char *)b->v.summary - lo), sizeof(double) * w * 6);
        if (b->cluster > 1) memcpy(&failed, sa, sizeof(int));
    } else {
        if (poses_h) HIP_TRY(c, hipMemcpyAsync(poses_h, b->v.poses, sizeof(double) * w * kBaMaxPoses * 7, hipMemcpyDeviceToHost, c->stream));
        if (ex_h) HIP_TRY(c, hipMemcpyAsync(ex_h, b->v.ex, sizeof(double) * w * 7, hipMemcpyDeviceToHost, c->stream));
        if (inv_depth_h && b->total_feat > 0) HIP_TRY(c, hipMemcpyAsync(inv_depth_h, b->v.inv_depth, sizeof(double) * (size_t)b->total_feat, hipMemcpyDeviceToHost, c->stream));
        if (summary_h) HIP_TRY(c, hipMemcpyAsync(summary_h, b->v.summary, sizeof(double) * w * 6, hipMemcpyDeviceToHost, c->stream));
        if (b->cluster > 1) HIP_TRY(c, hipMemcpyAsync(&failed, b->v.fail, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    if (failed && may_retry && b->pre) {
        // A workgroup of some window's cluster did not arrive within the poll bound (a CU mask, a partition, another process's resident workgroups: the
        // residency budget is a guess about a card this context does not own).  Every window of the launch may have stopped early, so the whole solve runs
        // again from the state it started from with ONE workgroup per window, which needs nobody resident but itself -- the same bytes (the sums are formed
        // per segment in segment order whatever K is).
        b->retries++;
        const size_t range = (size_t)((const char *)(b->v.inv_depth + b->total_feat) - (const char *)b->v.poses);
        HIP_TRY(c, hipMemcpyAsync(b->v.poses, b->pre, range, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(c, hipMemsetAsync(b->v.fail, 0, sizeof(int), c->stream));
        const int rc = ba_launch_single(c, b);
        if (rc) return rc;
        return ba_read(c, b, poses_h, ex_h, inv_depth_h, summary_h, false);
    }
    if (failed) { c->err = "k_ba_solve: a workgroup of a window's cluster did not arrive (not all resident?): the solve is void"; return LMONO_ENODEV; }
    return LMONO_OK;
}

// ---- per-feature kernels (triangulation, depth refinement, outlier scores, depth shift) ---------------------------
namespace {
// Scratch of one ABI call, carved from the context's arena (released when the scope ends; the chunks stay).  Uploads are asynchronous on
// the context stream: nothing here touches the null stream or synchronises the device, so two contexts on two host threads overlap.
// Round 4: small uploads are STAGED -- copied into the context's pinned staging buffer at the same spacing as their device allocations and sent by
// ready() as one copy per run of adjacent allocations (the per-feature calls of a frame made ~25 separate pageable uploads, each a staged, host-blocking
// copy of its own).  Every user calls ready() behind its last up() and before its first launch.  A call's staged bytes are consumed before the call
// returns (every call ends waiting for its results), so the next call may overwrite them.
struct DevBuf {
    lmono_ctx *c;
    size_t chunk0, off0;
    bool used = false;
    static constexpr size_t kStageMax = (size_t)256 << 10;      // larger uploads (clouds) go directly
    char *run_dst = nullptr;         // device address of the pending run's first byte
    size_t run_at = 0, run_bytes = 0, stage_used = 0;
    explicit DevBuf(lmono_ctx *c_) : c(c_), chunk0(c_ ? c_->arena_chunk : 0), off0(c_ ? c_->arena_off : 0) {}
    bool send_run()
    {
        if (run_bytes == 0) return true;
        const bool sent = hipMemcpyAsync(run_dst, c->stage + run_at, run_bytes, hipMemcpyHostToDevice, c->stream) == hipSuccess;
        run_dst = nullptr; run_bytes = 0;
        return sent;
    }
    // every staged upload is on its way (call once, behind the last up() and before the first launch)
    void ready(bool &ok) { if (c && !send_run()) ok = false; }
    // Results come back the same way (round 5): small read-backs are queued into the pinned staging buffer -- device-adjacent ones as ONE copy -- and
    // handed to the caller's (pageable) arrays by fetch(), which waits for the stream once.  A copy into pageable memory is staged by the runtime on
    // its own and waited for one by one: ~20 us each in the Estimator's frame loop, four of them per frame.
    struct Pending { char *dst; size_t at, bytes; };
    std::vector<Pending> downs;
    const char *drun_src = nullptr; size_t drun_at = 0, drun_bytes = 0;
    bool flush_down()
    {
        if (drun_bytes == 0) return true;
        const bool sent = hipMemcpyAsync(c->stage + drun_at, drun_src, drun_bytes, hipMemcpyDeviceToHost, c->stream) == hipSuccess;
        drun_src = nullptr; drun_bytes = 0;
        return sent;
    }
    bool down(void *dst, const void *src, size_t bytes)
    {
        const size_t al = (bytes + 255) & ~(size_t)255;
        if (!c->stage || bytes > kStageMax || stage_used + al > c->stage_cap)
            return flush_down() && hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream) == hipSuccess;
        if (drun_bytes > 0 && (const char *)src != drun_src + drun_bytes && !flush_down()) return false;
        if (drun_bytes == 0) { drun_src = (const char *)src; drun_at = stage_used; }
        downs.push_back({ (char *)dst, stage_used, bytes });
        stage_used += al; drun_bytes += al;
        return true;
    }
    // every queued read-back is in the caller's arrays (waits for the stream)
    bool fetch()
    {
        if (!flush_down() || hipStreamSynchronize(c->stream) != hipSuccess) return false;
        for (const Pending &p : downs) memcpy(p.dst, c->stage + p.at, p.bytes);
        downs.clear();
        return true;
    }
    // the scratch goes back to the arena only once nothing queued on the stream can still touch it (a no-op wait on the normal path,
    // where the call has already waited for its results; it matters on the early error returns)
    ~DevBuf() { if (c) { if (used) (void)hipStreamSynchronize(c->stream); c->arena_chunk = chunk0; c->arena_off = off0; } }
    template <typename T> T *up(const T *src, size_t n, bool &ok)
    {
        if (!ok || !c) { ok = false; return nullptr; }
        used = true;
        const size_t bytes = (((n > 0 ? n : 1) * sizeof(T)) + 255) & ~(size_t)255;
        while (c->arena_chunk < c->arena.size() && c->arena_off + bytes > c->arena[c->arena_chunk].cap) { c->arena_chunk++; c->arena_off = 0; }
        if (c->arena_chunk == c->arena.size()) {
            size_t cap = c->arena.empty() ? (size_t)1 << 20 : 2 * c->arena.back().cap;
            while (cap < bytes) cap <<= 1;
            void *q = nullptr;
            if (hipMalloc(&q, cap) != hipSuccess) { ok = false; return nullptr; }
            c->arena.push_back({ (char *)q, cap });
            c->arena_off = 0;
        }
        char *q = c->arena[c->arena_chunk].base + c->arena_off;
        c->arena_off += bytes;
        if (!src || n == 0) { if (!send_run()) ok = false; return (T *)q; }        // scratch: the run of adjacent uploads ends here
        if (bytes > kStageMax) {
            if (!send_run() || hipMemcpyAsync(q, src, n * sizeof(T), hipMemcpyHostToDevice, c->stream) != hipSuccess) ok = false;
            return (T *)q;
        }
        if (stage_used + bytes > c->stage_cap) {
            // grow: the old buffer may still be read by a copy in flight, so it is kept until the context is destroyed (a handful of doublings at most)
            if (!send_run()) ok = false;
            size_t cap = c->stage_cap ? 2 * c->stage_cap : (size_t)4 << 20;
            while (cap < bytes) cap <<= 1;
            void *h = nullptr;
            if (hipHostMalloc(&h, cap, hipHostMallocDefault) != hipSuccess) { ok = false; return nullptr; }
            c->stage_all.push_back(h); c->stage = (char *)h; c->stage_cap = cap; stage_used = 0;
        }
        if (run_bytes > 0 && q != run_dst + run_bytes) { if (!send_run()) ok = false; }       // not adjacent on the device (a new arena chunk)
        if (run_bytes == 0) { run_dst = q; run_at = stage_used; }
        memcpy(c->stage + stage_used, src, n * sizeof(T));
        stage_used += bytes; run_bytes += bytes;
        return (T *)q;
    }
};
// result back to a host array: on the context stream (a blocking hipMemcpy would go through the null stream and wait for other contexts)
template <typename T> hipError_t dl_async(lmono_ctx *c, T *dst_h, const T *src_d, size_t n) { return hipMemcpyAsync(dst_h, src_d, n * sizeof(T), hipMemcpyDeviceToHost, c->stream); }
}

static int feat_setup(lmono_ctx *c, DevBuf &db, FeatBatch &B, int n_windows, const int *feat_off, const double *Rs, const double *Ps, const double *tlc,
                      const int *start_frame, const int *obs_off, const double *pts, const double *depth)
{
    if (!c || n_windows <= 0 || !feat_off || !Rs || !Ps || !tlc || !start_frame || !obs_off || !pts || !depth) return LMONO_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    const int F = feat_off[n_windows];
    for (int w = 0; w < n_windows; w++) if (feat_off[w + 1] - feat_off[w] > LMONO_BA_MAX_FEATURES) { c->err = "more than LMONO_BA_MAX_FEATURES (" + std::to_string(LMONO_BA_MAX_FEATURES) + ") tracks in a window"; return LMONO_ECAPACITY; }
    const int TO = F > 0 ? obs_off[F] : 0;
    bool ok = true;
    B.n_windows = n_windows;
    B.feat_off = db.up(feat_off, (size_t)n_windows + 1, ok);
    B.Rs = db.up(Rs, (size_t)n_windows * 99, ok); B.Ps = db.up(Ps, (size_t)n_windows * 33, ok); B.tlc = db.up(tlc, (size_t)n_windows * 16, ok);
    B.start_frame = db.up(start_frame, (size_t)F, ok); B.obs_off = db.up(obs_off, (size_t)F + 1, ok); B.pts = db.up(pts, (size_t)TO * 2, ok);
    B.depth = db.up(depth, (size_t)F, ok);
    B.solve_flag = db.up((const int *)nullptr, (size_t)F, ok); B.score = db.up((const double *)nullptr, (size_t)F, ok);
    B.x = nullptr; B.cand = nullptr;
    db.ready(ok);
    if (!ok) { c->err = "per-feature kernels: device allocation / upload failed"; return LMONO_ENOMEM; }
    return LMONO_OK;
}

extern "C" int lmono_triangulate(lmono_ctx *c, int n_windows, const int *feat_off_h, const double *Rs_h, const double *Ps_h, const double *tlc_h,
                                 const int *start_frame_h, const int *obs_off_h, const double *pts_h, double *depth_h, int *solve_flag_h,
                                 int track_cnt, int window_size, double factor_weight, int refine_max_iter)
{
    DevBuf db(c); FeatBatch B{};
    int rc = feat_setup(c, db, B, n_windows, feat_off_h, Rs_h, Ps_h, tlc_h, start_frame_h, obs_off_h, pts_h, depth_h);
    if (rc) return rc;
    B.track_cnt = track_cnt; B.window_size = window_size; B.weight = factor_weight; B.max_iter = refine_max_iter;
    const int F = feat_off_h[n_windows];
    if (F == 0) return LMONO_OK;
    hipLaunchKernelGGL(k_triangulate_init, dim3((F + 127) / 128), dim3(128), 0, c->stream, B);
    if (refine_max_iter >= 0) {
        // one observation per thread when every window's (track, observation) pairs fit the kernel's LDS (the Estimator's windows do): same bits, a quarter of the time
        bool items = true;
        for (int w = 0; w < n_windows && items; w++) {
            const int nf = feat_off_h[w + 1] - feat_off_h[w];
            const int no = nf > 0 ? obs_off_h[feat_off_h[w + 1]] - obs_off_h[feat_off_h[w]] : 0;
            if (nf > kDrT || no > kDrItems) items = false;
        }
        if (items) hipLaunchKernelGGL(k_depth_refine_items, dim3(n_windows), dim3(kDrT), 0, c->stream, B);
        else hipLaunchKernelGGL(k_depth_refine, dim3(n_windows), dim3(256), 0, c->stream, B);
    }
    rc = check_launch(c, "k_triangulate_init/k_depth_refine");
    if (rc) return rc;
    bool ok = db.down(depth_h, B.depth, sizeof(double) * F);
    if (solve_flag_h && refine_max_iter >= 0) ok = ok && db.down(solve_flag_h, B.solve_flag, sizeof(int) * F);
    if (!ok || !db.fetch()) { c->err = "lmono_triangulate: read-back failed"; return LMONO_ENODEV; }      // the results are in the caller's arrays
    return LMONO_OK;
}

extern "C" int lmono_outlier_scores(lmono_ctx *c, int n_windows, const int *feat_off_h, const double *Rs_h, const double *Ps_h, const double *tlc_h,
                                    const int *start_frame_h, const int *obs_off_h, const double *pts_h, const double *depth_h,
                                    int track_cnt, double factor_weight, double *score_h)
{
    if (!score_h) return LMONO_EINVAL;
    DevBuf db(c); FeatBatch B{};
    int rc = feat_setup(c, db, B, n_windows, feat_off_h, Rs_h, Ps_h, tlc_h, start_frame_h, obs_off_h, pts_h, depth_h);
    if (rc) return rc;
    B.track_cnt = track_cnt; B.window_size = 0; B.weight = factor_weight; B.max_iter = 0;
    const int F = feat_off_h[n_windows];
    if (F == 0) return LMONO_OK;
    hipLaunchKernelGGL(k_outlier_scores, dim3((F + 127) / 128), dim3(128), 0, c->stream, B);
    rc = check_launch(c, "k_outlier_scores");
    if (rc) return rc;
    if (!db.down(score_h, B.score, sizeof(double) * F) || !db.fetch()) { c->err = "lmono_outlier_scores: read-back failed"; return LMONO_ENODEV; }
    return LMONO_OK;
}

extern "C" int lmono_shift_depth(lmono_ctx *c, const double *back_R0, const double *back_P0, const double *R1, const double *P1, const double *tlc,
                                 int n, const double *pt_i_h, const double *depth_h, double *depth_out_h)
{
    if (!c || !back_R0 || !back_P0 || !R1 || !P1 || !tlc || n < 0 || !pt_i_h || !depth_h || !depth_out_h) return LMONO_EINVAL;
    if (n == 0) return LMONO_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    double poses[40];
    memcpy(poses, back_R0, 72); memcpy(poses + 9, back_P0, 24); memcpy(poses + 12, R1, 72); memcpy(poses + 21, P1, 24); memcpy(poses + 24, tlc, 128);
    DevBuf db(c); bool ok = true;
    double *pd = db.up(poses, 40, ok), *pt = db.up(pt_i_h, (size_t)n * 2, ok), *d = db.up(depth_h, (size_t)n, ok), *o = db.up((const double *)nullptr, (size_t)n, ok);
    db.ready(ok);
    if (!ok) { c->err = "lmono_shift_depth: device allocation / upload failed"; return LMONO_ENOMEM; }
    hipLaunchKernelGGL(k_shift_depth, dim3((n + 127) / 128), dim3(128), 0, c->stream, (const double *)pd, n, (const double *)pt, (const double *)d, o, (const int *)nullptr);
    int rc = check_launch(c, "k_shift_depth");
    if (rc) return rc;
    if (!db.down(depth_out_h, o, sizeof(double) * n) || !db.fetch()) { c->err = "lmono_shift_depth: read-back failed"; return LMONO_ENODEV; }
    return LMONO_OK;
}

extern "C" int lmono_shift_depth_batch(lmono_ctx *c, int n_windows, const double *frames_h, const int *track_off_h,
                                       const double *pt_i_h, const double *depth_h, double *depth_out_h)
{
    if (!c || n_windows <= 0 || !frames_h || !track_off_h) return LMONO_EINVAL;
    const int n = track_off_h[n_windows];
    if (n < 0 || track_off_h[0] != 0) return LMONO_EINVAL;
    if (n == 0) return LMONO_OK;
    if (!pt_i_h || !depth_h || !depth_out_h) return LMONO_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    std::vector<int> win((size_t)n);
    for (int w = 0; w < n_windows; w++) {
        if (track_off_h[w + 1] < track_off_h[w]) { c->err = "lmono_shift_depth_batch: track offsets must ascend"; return LMONO_EINVAL; }
        for (int f = track_off_h[w]; f < track_off_h[w + 1]; f++) win[(size_t)f] = w;
    }
    DevBuf db(c); bool ok = true;
    double *pd = db.up(frames_h, (size_t)n_windows * 40, ok), *pt = db.up(pt_i_h, (size_t)n * 2, ok), *d = db.up(depth_h, (size_t)n, ok);
    int *wd = db.up(win.data(), (size_t)n, ok);
    double *o = db.up((const double *)nullptr, (size_t)n, ok);
    db.ready(ok);
    if (!ok) { c->err = "lmono_shift_depth_batch: device allocation / upload failed"; return LMONO_ENOMEM; }
    hipLaunchKernelGGL(k_shift_depth, dim3((n + 127) / 128), dim3(128), 0, c->stream, (const double *)pd, n, (const double *)pt, (const double *)d, o, (const int *)wd);
    int rc = check_launch(c, "k_shift_depth");
    if (rc) return rc;
    if (!db.down(depth_out_h, o, sizeof(double) * n) || !db.fetch()) { c->err = "lmono_shift_depth_batch: read-back failed"; return LMONO_ENODEV; }
    return LMONO_OK;
}

// ---- marginalisation prior ----------------------------------------------------------------------------------------
extern "C" int lmono_marginalize(lmono_ctx *c, int n_windows, const int *feat_off_h, const int *obs_off_h, const double *poses_h, const double *ex_h,
                                 const double *inv_depth_h, const int *obs_feat_h, const int *obs_j_h, const double *obs_pts_h,
                                 const double *laser01_h, const double *laser_info_h, const double *mono_info_h,
                                 double *lin_J_h, double *lin_r_h, int *status_h)
{
    if (!c || n_windows <= 0 || !feat_off_h || !obs_off_h || !poses_h || !ex_h || !laser01_h || !laser_info_h || !mono_info_h || !lin_J_h || !lin_r_h) return LMONO_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    const int TF = feat_off_h[n_windows], TO = obs_off_h[n_windows];
    std::vector<int> fo((size_t)TF + 1, 0);
    for (int w = 0; w < n_windows; w++) {
        if (feat_off_h[w + 1] - feat_off_h[w] > kMargMaxF0) { c->err = "lmono_marginalize: more than 160 tracks anchored at frame 0"; return LMONO_ECAPACITY; }
        int o = obs_off_h[w];
        for (int f = feat_off_h[w]; f < feat_off_h[w + 1]; f++) {
            fo[f] = o;
            while (o < obs_off_h[w + 1] && obs_feat_h[o] == f - feat_off_h[w]) {
                if (obs_j_h[o] < 1 || obs_j_h[o] > 10) { c->err = "lmono_marginalize: observation frame must be 1..10"; return LMONO_EINVAL; }
                o++;
            }
        }
        if (o != obs_off_h[w + 1]) { c->err = "lmono_marginalize: observations are not grouped by track"; return LMONO_EINVAL; }
    }
    fo[TF] = TO;
    double info[40];
    memcpy(info, laser_info_h, 36 * sizeof(double)); memcpy(info + 36, mono_info_h, 4 * sizeof(double));
    DevBuf db(c); bool ok = true;
    MargBatch B{};
    B.n_windows = n_windows;
    B.feat_off = db.up(feat_off_h, (size_t)n_windows + 1, ok); B.obs_off = db.up(obs_off_h, (size_t)n_windows + 1, ok);
    B.poses = db.up(poses_h, (size_t)n_windows * 77, ok); B.ex = db.up(ex_h, (size_t)n_windows * 7, ok);
    B.inv_depth = db.up(inv_depth_h, (size_t)TF, ok); B.feat_obs_off = db.up(fo.data(), (size_t)TF + 1, ok);
    B.obs_j = db.up(obs_j_h, (size_t)TO, ok); B.obs_pts = db.up(obs_pts_h, (size_t)TO * 4, ok);
    B.laser01 = db.up(laser01_h, (size_t)n_windows * 24, ok); B.info = db.up(info, (size_t)40, ok);
    B.lin_J = db.up((const double *)nullptr, (size_t)n_windows * kMargN * kMargN, ok); B.lin_r = db.up((const double *)nullptr, (size_t)n_windows * kMargN, ok);
    B.status = db.up((const int *)nullptr, (size_t)n_windows, ok);
    db.ready(ok);
    if (!ok) { c->err = "lmono_marginalize: device allocation / upload failed"; return LMONO_ENOMEM; }
    hipLaunchKernelGGL(k_marginalize, dim3(n_windows), dim3(kMgT), sizeof(MargLds), c->stream, B);
    int rc = check_launch(c, "k_marginalize");
    if (rc) return rc;
    // (the three outputs are neighbours in the scratch: one copy through the pinned staging buffer)
    bool got = db.down(lin_J_h, B.lin_J, sizeof(double) * (size_t)n_windows * kMargN * kMargN) && db.down(lin_r_h, B.lin_r, sizeof(double) * (size_t)n_windows * kMargN);
    if (got && status_h) got = db.down(status_h, B.status, sizeof(int) * (size_t)n_windows);
    if (!got || !db.fetch()) { c->err = "lmono_marginalize: read-back failed"; return LMONO_ENODEV; }      // the results are in the caller's arrays
    return LMONO_OK;
}

extern "C" int lmono_marg_evaluate(lmono_ctx *c, int n_windows, const double *lin_J_h, const double *lin_r_h, const double *x0_h, const double *x_h, double *residual_h)
{
    if (!c || n_windows <= 0 || !lin_J_h || !lin_r_h || !x0_h || !x_h || !residual_h) return LMONO_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    DevBuf db(c); bool ok = true;
    const double *J = db.up(lin_J_h, (size_t)n_windows * kMargN * kMargN, ok), *r = db.up(lin_r_h, (size_t)n_windows * kMargN, ok);
    const double *x0 = db.up(x0_h, (size_t)n_windows * 77, ok), *x = db.up(x_h, (size_t)n_windows * 77, ok);
    double *res = db.up((const double *)nullptr, (size_t)n_windows * kMargN, ok);
    db.ready(ok);
    if (!ok) { c->err = "lmono_marg_evaluate: device allocation / upload failed"; return LMONO_ENOMEM; }
    hipLaunchKernelGGL(k_marg_evaluate, dim3(n_windows), dim3(128), 0, c->stream, n_windows, J, r, x0, x, res);
    int rc = check_launch(c, "k_marg_evaluate");
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(residual_h, res, sizeof(double) * (size_t)n_windows * kMargN, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));      // the results are in the caller's arrays
    return LMONO_OK;
}


// MARGIN_SECOND_NEW (Estimator.cc:1406-1470): the previous prior loses one of its blocks.
extern "C" int lmono_marg_second_new(lmono_ctx *c, int n_windows, int n_blocks, int drop_block, const double *lin_J_h, const double *lin_r_h,
                                     const double *x0_h, const double *x_h, double *lin_J_out_h, double *lin_r_out_h, int *status_h)
{
    if (!c || n_windows <= 0 || n_blocks < 2 || n_blocks > 11 || drop_block < 0 || drop_block >= n_blocks || !lin_J_h || !lin_r_h || !x0_h || !x_h ||
        !lin_J_out_h || !lin_r_out_h) return LMONO_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n0 = 6 * (size_t)n_blocks, n = n0 - 6, W = (size_t)n_windows;
    DevBuf db(c); bool ok = true;
    Marg2Batch B{};
    B.n_windows = n_windows; B.nb = n_blocks; B.drop = drop_block;
    B.lin_J = db.up(lin_J_h, W * n0 * n0, ok); B.lin_r = db.up(lin_r_h, W * n0, ok);
    B.x0 = db.up(x0_h, W * n_blocks * 7, ok); B.x = db.up(x_h, W * n_blocks * 7, ok);
    B.out_J = db.up((const double *)nullptr, W * n * n, ok); B.out_r = db.up((const double *)nullptr, W * n, ok);
    B.status = db.up((const int *)nullptr, W, ok);
    db.ready(ok);
    if (!ok) { c->err = "lmono_marg_second_new: device allocation / upload failed"; return LMONO_ENOMEM; }
    hipLaunchKernelGGL(k_marg_second_new, dim3(n_windows), dim3(kMgT), sizeof(Marg2Lds), c->stream, B);
    int rc = check_launch(c, "k_marg_second_new");
    if (rc) return rc;
    bool got = db.down(lin_J_out_h, B.out_J, sizeof(double) * W * n * n) && db.down(lin_r_out_h, B.out_r, sizeof(double) * W * n);
    if (got && status_h) got = db.down(status_h, B.status, sizeof(int) * W);
    if (!got || !db.fetch()) { c->err = "lmono_marg_second_new: read-back failed"; return LMONO_ENODEV; }      // the results are in the caller's arrays
    return LMONO_OK;
}

// ---- scan-to-map optimisation step of laserMapping, batched over independent streams ------------------------
extern "C" int lmono_map_refine(lmono_ctx *c, int n_streams,
                                const float *corner_map_h, const int64_t *corner_map_off, const float *surf_map_h, const int64_t *surf_map_off,
                                const float *corner_stack_h, const int64_t *corner_stack_off, const float *surf_stack_h, const int64_t *surf_stack_off,
                                double *pose_qt, int32_t *stats_h, int32_t *nn_out_h)
{
    if (!c || n_streams <= 0 || !corner_map_off || !surf_map_off || !corner_stack_off || !surf_stack_off || !pose_qt) return LMONO_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    const int64_t tot[4] = { corner_map_off[n_streams], surf_map_off[n_streams], corner_stack_off[n_streams], surf_stack_off[n_streams] };
    const float *src_h[4] = { corner_map_h, surf_map_h, corner_stack_h, surf_stack_h };
    for (int k = 0; k < 4; k++) if (tot[k] < 0 || (tot[k] > 0 && !src_h[k])) return LMONO_EINVAL;
    DevBuf db(c);
    bool ok = true;
    float4 *cloud_d[4];
    for (int k = 0; k < 4; k++) cloud_d[k] = (float4 *)db.up(src_h[k], (size_t)tot[k] * 4, ok);
    // hash tables, cell-sorted copies and scratch of the 2 n_streams map clouds
    std::vector<int64_t> toff((size_t)2 * n_streams + 1, 0);
    for (int s = 0; s < n_streams; s++)
        for (int w = 0; w < 2; w++) {
            const int64_t n = (w ? surf_map_off : corner_map_off)[s + 1] - (w ? surf_map_off : corner_map_off)[s];
            if (n < 0 || n > (1 << 24) - 2) { c->err = "lmono_map_refine: bad map cloud size"; return LMONO_EINVAL; }
            int T = 1024;
            while (T < n + 1) T <<= 1;
            toff[(size_t)2 * s + w + 1] = toff[(size_t)2 * s + w] + T;
        }
    GridCell *cells = (GridCell *)db.up((const char *)nullptr, (size_t)toff.back() * sizeof(GridCell), ok);
    float4 *sorted[2] = { (float4 *)db.up((const float *)nullptr, (size_t)tot[0] * 4, ok), (float4 *)db.up((const float *)nullptr, (size_t)tot[1] * 4, ok) };
    int *slot[2] = { db.up((const int *)nullptr, (size_t)tot[0], ok), db.up((const int *)nullptr, (size_t)tot[1], ok) };
    int *rank[2] = { db.up((const int *)nullptr, (size_t)tot[0], ok), db.up((const int *)nullptr, (size_t)tot[1], ok) };
    int *masks = db.up((const int *)nullptr, (size_t)4 * n_streams, ok);     // [2 n] masks, [2 n] run allocators
    const int64_t nq_total = tot[2] + tot[3];
    MapRec *rec = (MapRec *)db.up((const char *)nullptr, (size_t)(nq_total > 0 ? nq_total : 1) * sizeof(MapRec), ok);
    int *nn_d = nn_out_h ? db.up((const int *)nullptr, (size_t)(nq_total > 0 ? nq_total : 1) * 5, ok) : nullptr;
    int *nn_tmp_d = db.up((const int *)nullptr, (size_t)(nq_total > 0 ? nq_total : 1) * 5, ok);
    std::vector<double> xh((size_t)n_streams * 8, 0.0);
    for (int s = 0; s < n_streams; s++) for (int k = 0; k < 7; k++) xh[(size_t)s * 8 + k] = pose_qt[(size_t)s * 7 + k];
    double *x_d = db.up(xh.data(), xh.size(), ok);
    std::vector<int> zero((size_t)n_streams * 16, 0);
    int *stats_d = db.up(zero.data(), (size_t)n_streams * 8, ok);
    unsigned int *bar_d = (unsigned int *)db.up(zero.data(), (size_t)n_streams * 16, ok);          // cluster barriers of the two solves, zeroed
    double *part_d = db.up((const double *)nullptr, (size_t)n_streams * kMsEvals * kMsMaxK * 28, ok);
    db.ready(ok);
    if (!ok) { c->err = "lmono_map_refine: device allocation / upload failed"; return LMONO_ENOMEM; }
    std::vector<CloudJob> jobs((size_t)2 * n_streams);
    std::vector<MapStream> st((size_t)n_streams);
    int max_nq = 0, max_nmap = 0;
    int64_t rec_at = 0;
    for (int s = 0; s < n_streams; s++) {
        MapStream &S = st[(size_t)s];
        for (int w = 0; w < 2; w++) {
            const int64_t *moff = w ? surf_map_off : corner_map_off, *soff = w ? surf_stack_off : corner_stack_off;
            CloudJob &J = jobs[(size_t)2 * s + w];
            J.src = cloud_d[w] + moff[s]; J.n = (int)(moff[s + 1] - moff[s]);
            max_nmap = std::max(max_nmap, J.n);
            J.cell = cells + toff[(size_t)2 * s + w]; J.tcap = (int)(toff[(size_t)2 * s + w + 1] - toff[(size_t)2 * s + w]);
            J.sorted = sorted[w] + moff[s]; J.slot_of = slot[w] + moff[s]; J.rank_of = rank[w] + moff[s];
            J.mask_out = masks + 2 * s + w; J.bump = masks + 2 * n_streams + 2 * s + w;
            S.cell[w] = J.cell; S.sorted[w] = J.sorted; S.cloud[w] = J.src; S.mask[w] = J.mask_out; S.n_map[w] = J.n;
            S.stack[w] = cloud_d[2 + w] + soff[s]; S.n_stack[w] = (int)(soff[s + 1] - soff[s]);
        }
        S.rec = rec + rec_at; S.x = x_d + (size_t)s * 8; S.stats = stats_d + (size_t)s * 8;
        S.part = part_d + (size_t)s * kMsEvals * kMsMaxK * 28; S.bar = bar_d + (size_t)s * 16;
        S.nn_out = nn_d ? nn_d + rec_at * 5 : nullptr;
        S.nn_tmp = nn_tmp_d + rec_at * 5;
        const int nq = S.n_stack[0] + S.n_stack[1];
        rec_at += nq;
        max_nq = nq > max_nq ? nq : max_nq;
    }
    CloudJob *jobs_d = db.up(jobs.data(), jobs.size(), ok);
    MapStream *st_d = db.up(st.data(), st.size(), ok);
    db.ready(ok);
    if (!ok) { c->err = "lmono_map_refine: device allocation / upload failed"; return LMONO_ENOMEM; }
    hipStream_t stream = c->stream;
    struct Events {             // destroyed on every return path
        hipEvent_t e[3] = { nullptr, nullptr, nullptr };
        ~Events() { for (hipEvent_t x : e) if (x) (void)hipEventDestroy(x); }
    } evs;
    for (hipEvent_t &x : evs.e) if (hipEventCreate(&x) != hipSuccess) { c->err = "hipEventCreate failed"; return LMONO_ENODEV; }
    const hipEvent_t ev0 = evs.e[0], ev1 = evs.e[1], ev2 = evs.e[2];
    (void)hipEventRecord(ev0, stream);
    launch_cloud_grids(stream, (const CloudJob *)jobs_d, 2 * n_streams, max_nmap);
    (void)hipEventRecord(ev1, stream);
    for (int outer = 0; outer < 2; outer++) {
        if (max_nq > 0) {
            hipLaunchKernelGGL(k_map_correspond, dim3((max_nq + 7) / 8, n_streams), dim3(256), 0, stream, (const MapStream *)st_d, outer, 0, n_streams);
            hipLaunchKernelGGL(k_map_factor, dim3((max_nq + 63) / 64, n_streams), dim3(64), 0, stream, (const MapStream *)st_d, outer, 0, n_streams);
        }
        launch_map_solve(stream, (const MapStream *)st_d, n_streams, outer, c->map_budget);
    }
    (void)hipEventRecord(ev2, stream);
    int rc = check_launch(c, "map refine kernels");
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(stream));
    float ms_grid = 0.f, ms_opt = 0.f;
    (void)hipEventElapsedTime(&ms_grid, ev0, ev1); (void)hipEventElapsedTime(&ms_opt, ev1, ev2);
    HIP_TRY(c, hipMemcpy(xh.data(), x_d, sizeof(double) * xh.size(), hipMemcpyDeviceToHost));
    for (int s = 0; s < n_streams; s++) for (int k = 0; k < 7; k++) pose_qt[(size_t)s * 7 + k] = xh[(size_t)s * 8 + k];
    {
        std::vector<int> sh((size_t)n_streams * 8);
        HIP_TRY(c, hipMemcpy(sh.data(), stats_d, sizeof(int) * sh.size(), hipMemcpyDeviceToHost));
        for (int s = 0; s < n_streams; s++) if (sh[(size_t)s * 8 + 6]) { c->err = "lmono_map_refine: a solve's cluster barrier timed out"; return LMONO_ENODEV; }
    }
    if (stats_h) {
        HIP_TRY(c, hipMemcpy(stats_h, stats_d, sizeof(int) * (size_t)n_streams * 8, hipMemcpyDeviceToHost));
        // device time of the whole batch in microseconds: grid build, then the two correspond + solve rounds
        for (int s = 0; s < n_streams; s++) { stats_h[(size_t)s * 8 + 6] = (int)(ms_grid * 1e3f); stats_h[(size_t)s * 8 + 7] = (int)(ms_opt * 1e3f); }
    }
    if (nn_out_h && nq_total > 0) HIP_TRY(c, hipMemcpy(nn_out_h, nn_d, sizeof(int) * (size_t)nq_total * 5, hipMemcpyDeviceToHost));
    return LMONO_OK;
}

// ---- pcl::VoxelGrid on arbitrary clouds (laserMapping's scan and cube filters), a batch of clouds per call -----------------
extern "C" int lmono_voxel_filter(lmono_ctx *c, int n_clouds, const float *xyzi_h, const int64_t *off, const float *leaf_h,
                                  float *out_h, int64_t *out_off)
{
    if (!c || n_clouds <= 0 || !off || !leaf_h || !out_h || !out_off) return LMONO_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    const int64_t total = off[n_clouds];
    if (off[0] != 0 || total < 0 || (total > 0 && !xyzi_h)) return LMONO_EINVAL;
    for (int k = 0; k < n_clouds; k++) {
        const int64_t n = off[k + 1] - off[k];
        if (n < 0 || n > kVoxCloudMax) { c->err = "lmono_voxel_filter: a cloud holds more than 65536 points"; return LMONO_ECAPACITY; }
        if (!(leaf_h[k] > 0.f)) { c->err = "lmono_voxel_filter: leaf size must be positive"; return LMONO_EINVAL; }
    }
    DevBuf db(c);
    bool ok = true;
    float4 *in_d = (float4 *)db.up(xyzi_h, (size_t)total * 4, ok);
    float4 *out_d = (float4 *)db.up((const float *)nullptr, (size_t)total * 4, ok);
    unsigned int *ka = db.up((const unsigned int *)nullptr, (size_t)total, ok), *kb = db.up((const unsigned int *)nullptr, (size_t)total, ok);
    int *ia = db.up((const int *)nullptr, (size_t)total, ok), *ib = db.up((const int *)nullptr, (size_t)total, ok);
    int *nout_d = db.up((const int *)nullptr, (size_t)n_clouds, ok);
    size_t ws_total = 0;
    for (int k = 0; k < n_clouds; k++) ws_total += vox_ws_ints(off[k + 1] - off[k]);
    int *ws_d = db.up((const int *)nullptr, ws_total, ok);
    std::vector<VoxJob> jobs((size_t)n_clouds);
    size_t ws_at = 0;
    for (int k = 0; k < n_clouds; k++) {
        VoxJob &J = jobs[(size_t)k];
        J.ws = ws_d + ws_at; ws_at += vox_ws_ints(off[k + 1] - off[k]);
        J.in = in_d + off[k]; J.n = (int)(off[k + 1] - off[k]); J.inv_leaf = 1.0f / leaf_h[k];
        J.out = out_d + off[k]; J.n_out = nout_d + k;
        J.key_a = ka + off[k]; J.key_b = kb + off[k]; J.idx_a = ia + off[k]; J.idx_b = ib + off[k];
    }
    VoxJob *jobs_d = db.up(jobs.data(), jobs.size(), ok);
    std::vector<int> tab;
    vox_tile_table(jobs.data(), jobs.size(), tab);
    int *tab_d = db.up(tab.data(), tab.size(), ok);
    db.ready(ok);
    if (!ok) { c->err = "lmono_voxel_filter: device allocation / upload failed"; return LMONO_ENOMEM; }
    launch_voxel_jobs(c->stream, (const VoxJob *)jobs_d, tab_d, (int)tab.size(), 4);
    int rc = check_launch(c, "voxel filter kernels");
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    std::vector<int> nout((size_t)n_clouds);
    HIP_TRY(c, hipMemcpy(nout.data(), nout_d, sizeof(int) * (size_t)n_clouds, hipMemcpyDeviceToHost));
    out_off[0] = 0;
    for (int k = 0; k < n_clouds; k++) {
        if (nout[(size_t)k] < 0) { c->err = "lmono_voxel_filter: kernel rejected a cloud"; return LMONO_ECAPACITY; }
        out_off[k + 1] = out_off[k] + nout[(size_t)k];
        if (nout[(size_t)k] > 0)
            HIP_TRY(c, hipMemcpy(out_h + 4 * out_off[k], out_d + off[k], sizeof(float) * 4 * (size_t)nout[(size_t)k], hipMemcpyDeviceToHost));
    }
    return LMONO_OK;
}

// ---- laserMapping with a device-resident cube map -------------------------------------------------------------
// The 21 x 21 x 11 cube array of laserMapping.cpp lives in two HBM arenas (corner, surf); the host keeps, per cube, only
// (offset, count).  Per frame the scan's feature clouds are taken from the scan batch in HBM, every numeric step runs in
// the kernels above, and what crosses PCIe is the pose (56 B), the cube index of every down-sampled scan point and the
// sizes of the re-filtered cubes.
namespace {
constexpr int kMapW = 21, kMapH = 21, kMapD = 11, kMapCubes = kMapW * kMapH * kMapD;
constexpr int64_t kMapArena = 6 << 20;        // points per arena half
constexpr int kMapNeighMax = 1 << 20;         // map points of the cube neighbourhood handed to one optimisation
constexpr int kMapStackMax = kVoxCloudMax;
struct Seg { int64_t off = 0; int n = 0; };
}
struct lmono_mapper {
    lmono_ctx *ctx = nullptr;
    float leaf[2] = { 0.4f, 0.8f };
    int cen[3] = { 10, 10, 5 };
    double q_wmap_wodom[4] = { 0, 0, 0, 1 }, t_wmap_wodom[3] = { 0, 0, 0 };
    std::vector<Seg> cube[2];
    std::vector<void *> allocs;
    std::vector<void *> pinned;                 // hipHostMalloc'ed mail boxes
    int *pin_i = nullptr; double *pin_x = nullptr; int *pin_cube = nullptr; int *pin_nout = nullptr; char *pin_blob = nullptr;
    size_t pin_i_cap = 0, pin_x_cap = 0, pin_cube_cap = 0, pin_nout_cap = 0, pin_blob_cap = 0;
    float4 *arena[2][2] = { { nullptr, nullptr }, { nullptr, nullptr } };   // [type][half]
    int half[2] = { 0, 0 };
    int64_t bump[2] = { 0, 0 };
    // per-frame workspace
    float4 *stack[2] = { nullptr, nullptr }, *newpts[2] = { nullptr, nullptr }, *neigh[2] = { nullptr, nullptr }, *sorted[2] = { nullptr, nullptr }, *cat[2] = { nullptr, nullptr };
    unsigned int *vk[2] = { nullptr, nullptr };
    hipStream_t side = nullptr;                 // second stream of a frame: the neighbourhood gather + grids run beside the scan clouds' voxel filter
    hipEvent_t ev_side = nullptr, ev_sizes = nullptr;
    double *solve_part = nullptr;               // k_map_solve's cluster: partial sums [kMsEvals][kMsMaxK][28]
    int *vws[2] = { nullptr, nullptr };         // voxel filter workspace (vox_ws_ints per job), vws_cap ints each
    size_t vws_cap = 0;
    int *vi[2] = { nullptr, nullptr }, *slot[2] = { nullptr, nullptr }, *rank[2] = { nullptr, nullptr }, *cube_of[2] = { nullptr, nullptr }, *pos[2] = { nullptr, nullptr };
    GridCell *cells[2] = { nullptr, nullptr };
    int tcap = 0;
    int *masks = nullptr, *nout = nullptr, *stats = nullptr;
    int *nout_big = nullptr;        // output sizes of the cube filter jobs of a batched call (owned by the first mapper)
    size_t nout_cap = 0;
    // contiguous mail boxes of a batched call (owned by the first mapper, grown on demand): poses, counters, cube indices /
    // placements of all streams travel in one copy per phase
    int *cubebuf = nullptr;
    size_t cubebuf_cap = 0;
    double *x = nullptr;
    MapRec *rec = nullptr;
    int *nn_tmp = nullptr;          // [2 kMapStackMax][5]
    void *jobs = nullptr;           // device scratch for job arrays
    size_t jobs_bytes = 0;
    MapStream *stream_d = nullptr;
    // device-resident bookkeeping (lmono_mapper_process; the batched entry keeps the tables on the host and converts on entry)
    bool dev_mode = false;
    MapDev *dev = nullptr;
    MapUpd *upd = nullptr;
    char *fblob[2] = { nullptr, nullptr };      // a frame's control block + job tables on the device, by frame parity
    char *fstage[2] = { nullptr, nullptr };     // their pinned staging buffers
    int parity = 0;
    unsigned int *vk_s[2] = { nullptr, nullptr };   // the scan filter's own workspace (it runs beside the previous frame's map update)
    int *vi_s[2] = { nullptr, nullptr }, *vws_s[2] = { nullptr, nullptr };
    int *cube_of_d = nullptr;
    hipEvent_t ev_commit = nullptr, ev_assign = nullptr, ev_pose = nullptr, ev_solve = nullptr;
    char *pin_back = nullptr;                   // pinned: the frame's read-back
    int bump_seen[2] = { 0, 0 }, nmap_seen[2] = { 0, 0 };
};

template <typename T> static bool mp_alloc(lmono_mapper *m, T *&p, size_t n)
{
    void *q = nullptr;
    if (hipMalloc(&q, (n > 0 ? n : 1) * sizeof(T)) != hipSuccess) return false;
    m->allocs.push_back(q);
    p = (T *)q;
    return true;
}

extern "C" void lmono_mapper_destroy(lmono_mapper *m)
{
    if (!m) return;
    if (m->ctx) (void)hipStreamSynchronize(m->ctx->stream);
    if (m->side) { (void)hipStreamSynchronize(m->side); (void)hipStreamDestroy(m->side); }
    if (m->ev_side) (void)hipEventDestroy(m->ev_side);
    if (m->ev_sizes) (void)hipEventDestroy(m->ev_sizes);
    for (hipEvent_t e : { m->ev_commit, m->ev_assign, m->ev_pose, m->ev_solve }) if (e) (void)hipEventDestroy(e);
    for (void *q : m->allocs) (void)hipFree(q);
    for (void *q : m->pinned) (void)hipHostFree(q);
    delete m;
}

extern "C" lmono_mapper *lmono_mapper_create(lmono_ctx *c, float line_res, float plane_res)
{
    if (!c || !(line_res > 0.f) || !(plane_res > 0.f)) return nullptr;
    if (hipSetDevice(c->device) != hipSuccess) return nullptr;
    lmono_mapper *m = new lmono_mapper();
    m->ctx = c; m->leaf[0] = line_res; m->leaf[1] = plane_res;
    m->cube[0].assign((size_t)kMapCubes, Seg()); m->cube[1].assign((size_t)kMapCubes, Seg());
    m->tcap = 1;
    m->vws_cap = (size_t)128 * kVxHdr + (size_t)((kMapNeighMax + kMapStackMax) / kVxTile + 128) * kVxWsTile;     // <= 75 cube jobs per type and frame
    while (m->tcap < kMapNeighMax + 1) m->tcap <<= 1;
    bool ok = true;
    for (int t = 0; t < 2 && ok; t++) {
        ok = ok && mp_alloc(m, m->arena[t][0], (size_t)kMapArena) && mp_alloc(m, m->arena[t][1], (size_t)kMapArena) &&
             mp_alloc(m, m->stack[t], (size_t)kMapStackMax) && mp_alloc(m, m->newpts[t], (size_t)kMapStackMax) &&
             mp_alloc(m, m->neigh[t], (size_t)kMapNeighMax) && mp_alloc(m, m->sorted[t], (size_t)kMapNeighMax) &&
             mp_alloc(m, m->cat[t], (size_t)kMapNeighMax + kMapStackMax) &&
             mp_alloc(m, m->vk[t], (size_t)2 * (kMapNeighMax + kMapStackMax)) && mp_alloc(m, m->vi[t], (size_t)2 * (kMapNeighMax + kMapStackMax)) &&
             mp_alloc(m, m->slot[t], (size_t)kMapNeighMax) && mp_alloc(m, m->rank[t], (size_t)kMapNeighMax) &&
             mp_alloc(m, m->cube_of[t], (size_t)kMapStackMax) && mp_alloc(m, m->pos[t], (size_t)kMapStackMax) &&
             mp_alloc(m, m->cells[t], (size_t)m->tcap) && mp_alloc(m, m->vws[t], m->vws_cap);
    }
    m->jobs_bytes = 1 << 20;
    m->nout_cap = 1024;
    ok = ok && mp_alloc(m, m->masks, 4) && mp_alloc(m, m->nout, 2 * 256) && mp_alloc(m, m->nout_big, m->nout_cap) && mp_alloc(m, m->stats, 8) && mp_alloc(m, m->x, 8) &&
         mp_alloc(m, m->rec, (size_t)2 * kMapStackMax) && mp_alloc(m, m->nn_tmp, (size_t)10 * kMapStackMax) && mp_alloc(m, (char *&)m->jobs, m->jobs_bytes) && mp_alloc(m, m->stream_d, 1) &&
         mp_alloc(m, m->solve_part, (size_t)kMsEvals * kMsMaxK * 28);
    ok = ok && hipStreamCreateWithFlags(&m->side, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&m->ev_side, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&m->ev_sizes, hipEventDisableTiming) == hipSuccess;
    constexpr size_t kFrameBlob = 4096;
    ok = ok && mp_alloc(m, m->dev, 1) && mp_alloc(m, m->upd, 1) && mp_alloc(m, m->fblob[0], kFrameBlob) && mp_alloc(m, m->fblob[1], kFrameBlob) && mp_alloc(m, m->cube_of_d, (size_t)2 * kMapStackMax);
    for (int t = 0; t < 2 && ok; t++)
        ok = ok && mp_alloc(m, m->vk_s[t], (size_t)2 * kMapStackMax) && mp_alloc(m, m->vi_s[t], (size_t)2 * kMapStackMax) && mp_alloc(m, m->vws_s[t], vox_ws_ints(kMapStackMax));
    for (hipEvent_t *e : { &m->ev_commit, &m->ev_assign, &m->ev_pose, &m->ev_solve }) ok = ok && hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess;
    if (ok) {
        void *q = nullptr;
        ok = hipHostMalloc(&q, 2 * kFrameBlob + 256, hipHostMallocDefault) == hipSuccess;
        if (ok) { m->pinned.push_back(q); m->fstage[0] = (char *)q; m->fstage[1] = (char *)q + kFrameBlob; m->pin_back = (char *)q + 2 * kFrameBlob; }
    }
    if (!ok) { c->err = "lmono_mapper_create: device allocation failed"; lmono_mapper_destroy(m); return nullptr; }
    return m;
}

extern "C" int lmono_mapper_reset(lmono_ctx *c, lmono_mapper *m)
{
    if (!c || !m) return LMONO_EINVAL;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipStreamSynchronize(m->side));
    m->dev_mode = false;          // the (empty) table goes back to the device with the next lmono_mapper_process
    m->bump_seen[0] = m->bump_seen[1] = 0; m->nmap_seen[0] = m->nmap_seen[1] = 0;
    for (int t = 0; t < 2; t++) { m->cube[(size_t)t].assign((size_t)kMapCubes, Seg()); m->half[t] = 0; m->bump[t] = 0; }
    m->cen[0] = 10; m->cen[1] = 10; m->cen[2] = 5;
    m->q_wmap_wodom[0] = m->q_wmap_wodom[1] = m->q_wmap_wodom[2] = 0.0; m->q_wmap_wodom[3] = 1.0;
    m->t_wmap_wodom[0] = m->t_wmap_wodom[1] = m->t_wmap_wodom[2] = 0.0;
    return LMONO_OK;
}

// compaction: copy every live segment of one type into the other arena half
static int mapper_compact(lmono_mapper *m, int t)
{
    lmono_ctx *c = m->ctx;
    HIP_TRY(c, hipStreamSynchronize(c->stream));      // rare path: the frame's tables in the job scratch are done with before it is overwritten
    std::vector<CopyJob> jobs;
    const int nh = m->half[t] ^ 1;
    int64_t at = 0;
    for (Seg &s : m->cube[(size_t)t]) {
        if (s.n == 0) continue;
        jobs.push_back({ m->arena[t][m->half[t]] + s.off, m->arena[t][nh] + at, s.n });
        s.off = at; at += s.n;
    }
    if (jobs.size() * sizeof(CopyJob) > m->jobs_bytes) { c->err = "lmono_mapper: job scratch too small"; return LMONO_ECAPACITY; }
    if (!jobs.empty()) {
        HIP_TRY(c, hipMemcpyAsync(m->jobs, jobs.data(), jobs.size() * sizeof(CopyJob), hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_copy_jobs, dim3((unsigned)jobs.size()), dim3(256), 0, c->stream, (const CopyJob *)m->jobs);
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    m->half[t] = nh; m->bump[t] = at;
    return LMONO_OK;
}

namespace {
template <typename T> int mp_grow(lmono_ctx *c, lmono_mapper *m, T *&p, size_t &cap, size_t need)
{
    if (need <= cap) return LMONO_OK;
    size_t nc = cap ? cap : 1024;
    while (nc < need) nc <<= 1;
    T *q = nullptr;
    if (!mp_alloc(m, q, nc)) { c->err = "lmono_mapper: allocation failed"; return LMONO_ENOMEM; }
    p = q; cap = nc;
    return LMONO_OK;
}
// pinned host mail boxes of the frame's read-backs (a copy into pageable memory holds the calling thread until it is done; into pinned memory it is
// asynchronous -- the host goes on enqueueing): grown on demand, freed with the mapper
template <typename T> int pin_grow(lmono_ctx *c, lmono_mapper *m, T *&p, size_t &cap, size_t need)
{
    if (need <= cap) return LMONO_OK;
    size_t nc = cap ? cap : 1024;
    while (nc < need) nc <<= 1;
    void *q = nullptr;
    if (hipHostMalloc(&q, nc * sizeof(T), hipHostMallocDefault) != hipSuccess) { c->err = "lmono_mapper: pinned allocation failed"; return LMONO_ENOMEM; }
    m->pinned.push_back(q);
    p = (T *)q; cap = nc;
    return LMONO_OK;
}
// job tables of one phase for every stream go through one pinned-free staging path: a scratch device buffer owned by the
// first mapper of the call, grown on demand
// Every table of a call gets its OWN piece of the scratch (bump allocation, 256-B aligned): no phase waits for the previous phase's
// kernels to be done with "the" job table before it uploads its own (round 4: four stream synchronisations per frame less).  A scratch
// that runs out is replaced by a larger one; the old one stays allocated (kernels in flight still read it) until the mapper is destroyed.
struct JobScratch {
    lmono_mapper *owner;
    size_t used = 0;
    void *last = nullptr;        // device address of the table uploaded last
    // space for a table whose entries point into the table itself: place() it (its device address is `last`), fill it, send() it
    int place(lmono_ctx *c, size_t bytes)
    {
        used = (used + 255) & ~(size_t)255;
        if (used + bytes > owner->jobs_bytes) {
            void *q = nullptr;
            size_t nb = owner->jobs_bytes;
            while (nb < bytes || nb < 2 * (used + bytes)) nb <<= 1;
            if (hipMalloc(&q, nb) != hipSuccess) { c->err = "lmono_mapper: job scratch allocation failed"; return LMONO_ENOMEM; }
            owner->allocs.push_back(q);
            owner->jobs = q; owner->jobs_bytes = nb; used = 0;
        }
        last = (char *)owner->jobs + used;
        used += bytes;
        return LMONO_OK;
    }
    int send(lmono_ctx *c, const void *src, size_t bytes, hipStream_t st)
    {
        if (bytes > 0 && hipMemcpyAsync(last, src, bytes, hipMemcpyHostToDevice, st) != hipSuccess) { c->err = "lmono_mapper: job upload failed"; return LMONO_ENODEV; }
        return LMONO_OK;
    }
    int upload(lmono_ctx *c, const void *src, size_t bytes, hipStream_t st)
    {
        const int rc = place(c, bytes);
        return rc ? rc : send(c, src, bytes, st);
    }
};
// a copy of n points as jobs of at most kCopyChunk points: one workgroup per job, and a 16 k-point cube in one workgroup was a 12-us kernel
constexpr int kCopyChunk = 4096;
static inline void push_copy(std::vector<CopyJob> &jobs, const float4 *src, float4 *dst, int n)
{
    for (int at = 0; at < n; at += kCopyChunk) jobs.push_back({ src + at, dst + at, std::min(kCopyChunk, n - at) });
}
// a voxel job table and its tile table in one upload: [VoxJob x n | int x tiles]; `blob` is the staging buffer (alive until the stream is waited for)
int upload_vox_jobs(lmono_ctx *c, JobScratch &js, const std::vector<VoxJob> &jobs, std::vector<char> &blob, hipStream_t st)
{
    std::vector<int> tab;
    vox_tile_table(jobs.data(), jobs.size(), tab);
    blob.resize(jobs.size() * sizeof(VoxJob) + tab.size() * sizeof(int));
    memcpy(blob.data(), jobs.data(), jobs.size() * sizeof(VoxJob));
    memcpy(blob.data() + jobs.size() * sizeof(VoxJob), tab.data(), tab.size() * sizeof(int));
    return js.upload(c, blob.data(), blob.size(), st);
}
void mp_qrot(const double *q, const double *v, double *o)
{
    const double ux = q[0], uy = q[1], uz = q[2], w = q[3];
    const double uvx = 2.0 * (uy * v[2] - uz * v[1]), uvy = 2.0 * (uz * v[0] - ux * v[2]), uvz = 2.0 * (ux * v[1] - uy * v[0]);
    o[0] = v[0] + w * uvx + (uy * uvz - uz * uvy); o[1] = v[1] + w * uvy + (uz * uvx - ux * uvz); o[2] = v[2] + w * uvz + (ux * uvy - uy * uvx);
}
void mp_qmul(const double *a, const double *bq, double *o)
{
    o[3] = a[3] * bq[3] - a[0] * bq[0] - a[1] * bq[1] - a[2] * bq[2];
    o[0] = a[3] * bq[0] + a[0] * bq[3] + a[1] * bq[2] - a[2] * bq[1];
    o[1] = a[3] * bq[1] + a[1] * bq[3] + a[2] * bq[0] - a[0] * bq[2];
    o[2] = a[3] * bq[2] + a[2] * bq[3] + a[0] * bq[1] - a[1] * bq[0];
}
int mp_cube_of(double v, int cen) { int q = (int)((v + 25.0) / 50.0) + cen; if (v + 25.0 < 0) q--; return q; }
void mp_shift(lmono_mapper *m, int axis, int dir)
{
    const int n[3] = { kMapW, kMapH, kMapD }, stride[3] = { 1, kMapW, kMapW * kMapH };
    const int a1 = (axis + 1) % 3, a2 = (axis + 2) % 3;
    for (int t = 0; t < 2; t++)
        for (int u = 0; u < n[a1]; u++)
            for (int v = 0; v < n[a2]; v++) {
                const int base = u * stride[a1] + v * stride[a2];
                std::vector<Seg> &arr = m->cube[(size_t)t];
                if (dir > 0) { for (int i = n[axis] - 1; i >= 1; i--) arr[(size_t)(base + i * stride[axis])] = arr[(size_t)(base + (i - 1) * stride[axis])]; arr[(size_t)base] = Seg(); }
                else { for (int i = 0; i < n[axis] - 1; i++) arr[(size_t)(base + i * stride[axis])] = arr[(size_t)(base + (i + 1) * stride[axis])]; arr[(size_t)(base + (n[axis] - 1) * stride[axis])] = Seg(); }
            }
}
// Host state of a mapper that lmono_mapper_process_batch changes; put back when the call fails after changing it.  The device
// side needs no undo: new points only ever land in free arena space (behind `bump`, or in the idle half during compaction).
struct MapperUndo {
    lmono_mapper *m;
    std::vector<Seg> cube[2];
    int cen[3], half[2];
    int64_t bump[2];
    double q[4], t[3];
    explicit MapperUndo(lmono_mapper *mp) : m(mp)
    {
        for (int k = 0; k < 2; k++) { cube[k] = m->cube[(size_t)k]; half[k] = m->half[k]; bump[k] = m->bump[k]; }
        for (int k = 0; k < 3; k++) { cen[k] = m->cen[k]; t[k] = m->t_wmap_wodom[k]; }
        for (int k = 0; k < 4; k++) q[k] = m->q_wmap_wodom[k];
    }
    void restore()
    {
        for (int k = 0; k < 2; k++) { m->cube[(size_t)k].swap(cube[k]); m->half[k] = half[k]; m->bump[k] = bump[k]; }
        for (int k = 0; k < 3; k++) { m->cen[k] = cen[k]; m->t_wmap_wodom[k] = t[k]; }
        for (int k = 0; k < 4; k++) m->q_wmap_wodom[k] = q[k];
    }
};
struct MapperUndoAll {
    std::vector<MapperUndo> u;
    bool committed = false;
    ~MapperUndoAll() { if (!committed) for (MapperUndo &x : u) x.restore(); }
};
struct FrameState {            // one stream's frame
    double x[8];
    std::vector<int> valid;
    int n_last[2], n_stack[2], n_map[2];
    bool solve;
    const int *cube_h[2] = { nullptr, nullptr };       // cube index of every stack point: into the pinned read-back
};
}

// ---- the cube table on the device (lmono_mapper_process) or on the host (lmono_mapper_process_batch, compaction): converted when the other side needs it
static int mapper_tables_to_host(lmono_ctx *c, lmono_mapper *m)
{
    if (!m->dev_mode) return LMONO_OK;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipStreamSynchronize(m->side));
    std::vector<int2> tab((size_t)2 * kMapCubes);
    int tail[6];
    HIP_TRY(c, hipMemcpy(tab.data(), &m->dev->tab[0][0], sizeof(int2) * tab.size(), hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(tail, &m->dev->bump[0], sizeof(tail), hipMemcpyDeviceToHost));
    for (int t = 0; t < 2; t++) {
        for (int k = 0; k < kMapCubes; k++) { Seg &sg = m->cube[(size_t)t][(size_t)k]; sg.off = tab[(size_t)t * kMapCubes + k].x; sg.n = tab[(size_t)t * kMapCubes + k].y; }
        m->bump[t] = tail[t];
    }
    m->dev_mode = false;
    if (tail[2]) { c->err = "lmono_mapper: a map update on the device was refused (capacity)"; return LMONO_ECAPACITY; }
    return LMONO_OK;
}
static int mapper_tables_to_device(lmono_ctx *c, lmono_mapper *m)
{
    if (m->dev_mode) return LMONO_OK;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    std::vector<int2> tab((size_t)2 * kMapCubes);
    for (int t = 0; t < 2; t++)
        for (int k = 0; k < kMapCubes; k++) { const Seg &sg = m->cube[(size_t)t][(size_t)k]; tab[(size_t)t * kMapCubes + k] = make_int2((int)sg.off, sg.n); }
    HIP_TRY(c, hipMemcpy(&m->dev->tab[0][0], tab.data(), sizeof(int2) * tab.size(), hipMemcpyHostToDevice));
    int tail[6] = { (int)m->bump[0], (int)m->bump[1], 0, 0, 0, 0 };
    HIP_TRY(c, hipMemcpy(&m->dev->bump[0], tail, sizeof(tail), hipMemcpyHostToDevice));
    CloudJob cj[2];
    for (int t = 0; t < 2; t++) {
        cj[t].src = m->neigh[t]; cj[t].n = 0; cj[t].cell = m->cells[t]; cj[t].tcap = m->tcap; cj[t].sorted = m->sorted[t];
        cj[t].slot_of = m->slot[t]; cj[t].rank_of = m->rank[t]; cj[t].mask_out = m->masks + t; cj[t].bump = m->masks + 2 + t;
    }
    HIP_TRY(c, hipMemcpy(&m->dev->cj[0], cj, sizeof(cj), hipMemcpyHostToDevice));
    m->bump_seen[0] = (int)m->bump[0]; m->bump_seen[1] = (int)m->bump[1];
    m->dev_mode = true;
    return LMONO_OK;
}

// One frame of one mapper with the bookkeeping on the device.  Streams: `side` takes the frame's upload, the scan clouds' voxel filter (beside the
// previous frame's map update, which is still running on the main stream), then -- behind that update's commit -- the shift of the cube array, the
// neighbourhood gather and its grids; the main stream takes the optimisation, hands the pose to the host, and goes on with the cube assignment, the
// update's plan, its copies and filters and the commit while the caller already prepares the next frame.
static int mapper_process_dev(lmono_ctx *c, lmono_mapper *m, lmono_scan_batch *b, int scan, const double *q_wodom, const double *t_wodom,
                              double *q_w_curr, double *t_w_curr, int32_t *stats_h)
{
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t st = c->stream, side = m->side;
    if (b->feat_h.empty()) {
        b->feat_h.assign((size_t)b->n_scans * 4, 0);
        HIP_TRY(c, hipStreamSynchronize(st));
        HIP_TRY(c, hipMemcpy(b->feat_h.data(), b->v.feat_n, sizeof(int) * 4 * (size_t)b->n_scans, hipMemcpyDeviceToHost));
    }
    const int n_last[2] = { b->feat_h[(size_t)scan * 4 + 1], b->feat_h[(size_t)scan * 4 + 3] };
    if (n_last[0] > kMapStackMax || n_last[1] > kMapStackMax) { c->err = "lmono_mapper: scan cloud too large"; return LMONO_ECAPACITY; }
    int rc;
    // arena space: what the host saw last is two updates old; compact (on the host's copy of the table: rare) long before the bump pointer can reach the end
    static const int64_t compact_at = [] { const char *e = getenv("LMONO_MAP_COMPACT_AT"); return e ? (int64_t)atoll(e) : (int64_t)0; }();      // test hook: compact as soon as a
                                                                                                                                              // bump pointer has passed this many points
    for (int t = 0; t < 2; t++)
        if ((int64_t)m->bump_seen[t] + 2 * ((int64_t)m->nmap_seen[t] + 3 * kMapStackMax) + kMapStackMax > kMapArena || (compact_at > 0 && m->bump_seen[t] > compact_at)) {
            if ((rc = mapper_tables_to_host(c, m))) return rc;
            if ((rc = mapper_compact(m, t))) return rc;
        }
    if ((rc = mapper_tables_to_device(c, m))) return rc;
    // ---- host: transformAssociateToMap, centre cube, shifts (on copies: the mapper changes when the frame has succeeded)
    double x[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, tmp[3];
    mp_qmul(m->q_wmap_wodom, q_wodom, x);
    mp_qrot(m->q_wmap_wodom, t_wodom, tmp);
    for (int k = 0; k < 3; k++) x[4 + k] = tmp[k] + m->t_wmap_wodom[k];
    int cen[3] = { m->cen[0], m->cen[1], m->cen[2] };
    int cc[3] = { mp_cube_of(x[4], cen[0]), mp_cube_of(x[5], cen[1]), mp_cube_of(x[6], cen[2]) };
    const int dims[3] = { kMapW, kMapH, kMapD };
    std::vector<std::pair<int, int>> shifts;
    for (int a = 0; a < 3; a++) {
        while (cc[a] < 3) { shifts.push_back({ a, +1 }); cc[a]++; cen[a]++; }
        while (cc[a] >= dims[a] - 3) { shifts.push_back({ a, -1 }); cc[a]--; cen[a]--; }
    }
    // ---- the frame's block: [MapFrame | VoxJob x 2 | tile table | MapStream | AssignJob x 2]
    const int par = m->parity;
    char *blob = m->fblob[par], *stage = m->fstage[par];
    auto al = [](size_t v) { return (v + 15) & ~(size_t)15; };
    VoxJob vj[2];
    vj[0].n = n_last[0]; vj[1].n = n_last[1];
    std::vector<int> tab;
    vox_tile_table(vj, 2, tab);
    const size_t o_vox = al(sizeof(MapFrame)), o_tab = al(o_vox + 2 * sizeof(VoxJob)), o_S = al(o_tab + tab.size() * sizeof(int)), o_aj = al(o_S + sizeof(MapStream)),
                 bytes = o_aj + 2 * sizeof(AssignJob);
    if (bytes > 4096) { c->err = "lmono_mapper: frame block too large"; return LMONO_ECAPACITY; }
    MapFrame *F_d = (MapFrame *)blob;
    memset(stage, 0, bytes);
    MapFrame *F = (MapFrame *)stage;
    for (int k = 0; k < 8; k++) F->x[k] = x[k];
    F->cen[0] = cen[0]; F->cen[1] = cen[1]; F->cen[2] = cen[2];
    F->n_valid = 0;
    for (int i = cc[0] - 2; i <= cc[0] + 2; i++)
        for (int j = cc[1] - 2; j <= cc[1] + 2; j++)
            for (int k = cc[2] - 1; k <= cc[2] + 1; k++)
                if (i >= 0 && i < kMapW && j >= 0 && j < kMapH && k >= 0 && k < kMapD) F->valid[F->n_valid++] = i + kMapW * j + kMapW * kMapH * k;
    for (int t = 0; t < 2; t++) {
        VoxJob &J = vj[t];
        J.in = t ? b->v.less_flat + b->off_h[(size_t)scan] : b->v.less_sharp + (size_t)scan * kMaxLessSharp;
        J.inv_leaf = 1.0f / m->leaf[t]; J.out = m->stack[t]; J.n_out = &F_d->n_stack[t];
        J.key_a = m->vk_s[t]; J.key_b = m->vk_s[t] + kMapStackMax; J.idx_a = m->vi_s[t]; J.idx_b = m->vi_s[t] + kMapStackMax; J.ws = m->vws_s[t];
    }
    memcpy(stage + o_vox, vj, sizeof(vj));
    if (!tab.empty()) memcpy(stage + o_tab, tab.data(), tab.size() * sizeof(int));
    MapStream S;
    memset(&S, 0, sizeof(S));
    for (int t = 0; t < 2; t++) {
        S.cell[t] = m->cells[t]; S.sorted[t] = m->sorted[t]; S.cloud[t] = m->neigh[t]; S.mask[t] = m->masks + t; S.n_map[t] = 0;
        S.stack[t] = m->stack[t]; S.n_stack[t] = 0;
    }
    S.n_stack_d = &F_d->n_stack[0]; S.rec = m->rec; S.x = &F_d->x[0]; S.stats = &F_d->stats[0]; S.nn_out = nullptr; S.nn_tmp = m->nn_tmp;
    S.part = m->solve_part; S.bar = &F_d->bar[0];
    memcpy(stage + o_S, &S, sizeof(S));
    AssignJob aj[2];
    for (int t = 0; t < 2; t++) aj[t] = { m->stack[t], -1, &F_d->n_stack[0], t, &F_d->x[0], cen[0], cen[1], cen[2], m->newpts[t], m->cube_of_d + (size_t)t * kMapStackMax };      // (n = -1: the job's own array of cube indices)
    memcpy(stage + o_aj, aj, sizeof(aj));
    MapDevCfg cfg;
    cfg.dev = m->dev; cfg.frame = F_d; cfg.upd = m->upd; cfg.S = (MapStream *)(blob + o_S);
    for (int t = 0; t < 2; t++) {
        cfg.arena[t] = m->arena[t][m->half[t]]; cfg.neigh[t] = m->neigh[t]; cfg.cat[t] = m->cat[t]; cfg.newpts[t] = m->newpts[t];
        cfg.vk[t] = m->vk[t]; cfg.vi[t] = m->vi[t]; cfg.vws[t] = m->vws[t]; cfg.inv_leaf[t] = 1.0f / m->leaf[t];
    }
    cfg.vws_cap = (int)m->vws_cap; cfg.cube_of[0] = m->cube_of_d; cfg.cube_of[1] = m->cube_of_d + kMapStackMax;
    // ---- side stream: upload, scan filter (behind the previous frame's cube assignment: it reads the stack), then behind the previous commit: shifts, gather, grids
    HIP_TRY(c, hipMemcpyAsync(blob, stage, bytes, hipMemcpyHostToDevice, side));
    HIP_TRY(c, hipStreamWaitEvent(side, m->ev_assign, 0));
    launch_voxel_jobs(side, (const VoxJob *)(blob + o_vox), (const int *)(blob + o_tab), (int)tab.size(), 4);
    HIP_TRY(c, hipStreamWaitEvent(side, m->ev_commit, 0));
    for (const std::pair<int, int> &sh : shifts) {
        HIP_TRY(c, hipMemcpyAsync(&m->dev->tmp[0][0], &m->dev->tab[0][0], sizeof(int2) * 2 * kMapCubes, hipMemcpyDeviceToDevice, side));
        hipLaunchKernelGGL(k_map_shift, dim3((2 * kMapCubes + 255) / 256), dim3(256), 0, side, m->dev, sh.first, sh.second);
    }
    // the neighbourhood's plan, its copies and the emptied hash tables in one launch (LMONO_MAP_MERGED=0: the three launches of rounds 4-5, for measurements)
    static const bool merged = [] { const char *e = getenv("LMONO_MAP_MERGED"); return !(e && atoi(e) == 0); }();
    if (merged)
        hipLaunchKernelGGL(k_map_gather_copy_clear, dim3(kMgcGrid), dim3(kMgcT), 0, side, cfg);
    else {
        hipLaunchKernelGGL(k_map_plan_gather, dim3(1), dim3(192), 0, side, cfg);
        hipLaunchKernelGGL(k_copy_jobs_n, dim3(128), dim3(256), 0, side, (const CopyJob *)m->dev->gjobs, (const int *)&m->dev->n_gjobs);
    }
    {
        const int est = std::max(65536, std::max(m->nmap_seen[0], m->nmap_seen[1]) * 5 / 4 + 32768);
        launch_cloud_grids(side, (const CloudJob *)m->dev->cj, 2, est, merged);
    }
    HIP_TRY(c, hipEventRecord(m->ev_side, side));
    // ---- main stream: the optimisation
    HIP_TRY(c, hipStreamWaitEvent(st, m->ev_side, 0));
    {
        const MapStream *S_d = (const MapStream *)(blob + o_S);
        const int max_nq = n_last[0] + n_last[1];
        const int per_stream = std::max(1, std::max((max_nq + 31) / 32, std::min((max_nq + 7) / 8, 2048)));
        for (int outer = 0; outer < 2; outer++) {
            if (max_nq > 0) {
                hipLaunchKernelGGL(k_map_correspond, dim3((unsigned)per_stream, 1u), dim3(256), 0, st, S_d, outer, 0, 1);
                hipLaunchKernelGGL(k_map_factor, dim3((unsigned)std::max(1, (per_stream + 7) / 8), 1u), dim3(64), 0, st, S_d, outer, 0, 1);
            }
            launch_map_solve(st, S_d, 1, outer, c->map_budget);
        }
    }
    HIP_TRY(c, hipEventRecord(m->ev_solve, st));
    // the read-back leaves on the side stream (the main stream goes straight on): [x | stats | bar | n_stack | n_map | snap] of the frame -- snap = the map's
    // [bump | err | . | last_sum] as k_map_plan_gather copied them at the head of this frame (the live words are rewritten by this frame's own
    // update on the main stream while the side stream reads; the snapshot is the previous update's, whole, as the header promises for stats[7])
    constexpr size_t kBackA = sizeof(double) * 8 + sizeof(int) * 8 + sizeof(unsigned int) * 16 + sizeof(int) * 4;
    static_assert(offsetof(MapFrame, snap) == kBackA, "the frame's read-back is one copy: x .. snap");
    HIP_TRY(c, hipStreamWaitEvent(side, m->ev_solve, 0));
    HIP_TRY(c, hipMemcpyAsync(m->pin_back, blob, kBackA + sizeof(int) * 6, hipMemcpyDeviceToHost, side));
    HIP_TRY(c, hipEventRecord(m->ev_pose, side));
    // ---- main stream: the scan joins the map
    {
        const int max_n = std::max(n_last[0], n_last[1]);
        if (max_n > 0) hipLaunchKernelGGL(k_map_assign, dim3((max_n + 255) / 256, 2), dim3(256), 0, st, (const AssignJob *)(blob + o_aj));
        HIP_TRY(c, hipEventRecord(m->ev_assign, st));
        // (k_map_assign's work inside the plan kernel -- one launch and one gap fewer in front of it -- was built and measured: 2.50-2.61 k frames/s against
        // 2.76-2.80 k; one compute unit transforming 11 k points costs more than the launch it saves)
        hipLaunchKernelGGL(k_map_plan_update, dim3(1), dim3(kMuT), 0, st, cfg);
        hipLaunchKernelGGL(k_copy_jobs_n, dim3(128), dim3(256), 0, st, (const CopyJob *)m->upd->copy, (const int *)&m->upd->n_copy);
        int cube_passes = 1;
        for (int t = 0; t < 2; t++) {
            const double per_axis = 50.0 / (double)m->leaf[t] + 3.0;
            int bits = 1;
            while (bits < 32 && std::ldexp(1.0, bits) < per_axis * per_axis * per_axis) bits++;
            cube_passes = std::max(cube_passes, std::min(4, (bits + 8) / 9));
        }
        launch_voxel_jobs(st, (const VoxJob *)m->upd->vox, (const int *)m->upd->tiles, 160, cube_passes, (const int *)&m->upd->n_tiles);
        hipLaunchKernelGGL(k_map_commit, dim3(1 + 16), dim3(kMuH), 0, st, cfg);          // (workgroup 0 commits, the others do the plain copies)
        HIP_TRY(c, hipEventRecord(m->ev_commit, st));
    }
    m->parity ^= 1;
    m->cen[0] = cen[0]; m->cen[1] = cen[1]; m->cen[2] = cen[2];          // the device's table has moved: so has the mapper's centre, whatever the frame's fate
    // ---- the one wait of the frame
    HIP_TRY(c, hipEventSynchronize(m->ev_pose));
    rc = check_launch(c, "mapper kernels");
    if (rc) return rc;
    const double *xb = (const double *)m->pin_back;
    const int *sb = (const int *)(m->pin_back + sizeof(double) * 8);
    const int *nb = (const int *)(m->pin_back + sizeof(double) * 8 + sizeof(int) * 8 + sizeof(unsigned int) * 16);
    const int *mb = (const int *)(m->pin_back + kBackA);
    if (sb[6]) { c->err = "lmono_mapper: a solve's cluster barrier timed out"; return LMONO_ENODEV; }
    if (nb[0] < 0 || nb[1] < 0) { c->err = "lmono_mapper: voxel filter rejected a scan cloud"; return LMONO_ECAPACITY; }
    if (mb[2]) { c->err = "lmono_mapper: the map update of an earlier frame was refused on the device (capacity)"; return LMONO_ECAPACITY; }
    for (int k = 0; k < 4; k++) q_w_curr[k] = xb[k];
    for (int k = 0; k < 3; k++) t_w_curr[k] = xb[4 + k];
    if (stats_h) {
        for (int k = 0; k < 6; k++) stats_h[k] = sb[k];
        stats_h[6] = nb[2] + nb[3];           // map points of the neighbourhood
        stats_h[7] = mb[4] + mb[5];           // points of the cubes the PREVIOUS frame's update rebuilt (this frame's update runs behind the return)
    }
    m->bump_seen[0] = mb[0]; m->bump_seen[1] = mb[1];
    m->nmap_seen[0] = nb[2]; m->nmap_seen[1] = nb[3];
    // transformUpdate; the shifts are the mapper's now
    {
        const double n2 = q_wodom[0] * q_wodom[0] + q_wodom[1] * q_wodom[1] + q_wodom[2] * q_wodom[2] + q_wodom[3] * q_wodom[3];
        const double qi[4] = { -q_wodom[0] / n2, -q_wodom[1] / n2, -q_wodom[2] / n2, q_wodom[3] / n2 };
        mp_qmul(xb, qi, m->q_wmap_wodom);
        mp_qrot(m->q_wmap_wodom, t_wodom, tmp);
        for (int k = 0; k < 3; k++) m->t_wmap_wodom[k] = xb[4 + k] - tmp[k];
    }
    return LMONO_OK;
}

// n mappers (independent streams), each advanced by one frame: every phase is one launch for all streams
extern "C" int lmono_mapper_process_batch(lmono_ctx *c, int n, lmono_mapper *const *ms, lmono_scan_batch *const *bs, const int *scans,
                                          const double *q_wodom, const double *t_wodom, double *q_w_curr, double *t_w_curr, int32_t *stats_h)
{
    if (!c || n <= 0 || !ms || !bs || !scans || !q_wodom || !t_wodom || !q_w_curr || !t_w_curr) return LMONO_EINVAL;
    for (int s = 0; s < n; s++) {
        if (!ms[s] || !bs[s] || !bs[s]->registered || scans[s] < 0 || scans[s] >= bs[s]->n_scans) return LMONO_EINVAL;
        for (int u = 0; u < s; u++) if (ms[u] == ms[s]) { c->err = "lmono_mapper_process_batch: a mapper appears twice"; return LMONO_EINVAL; }
    }
    HIP_TRY(c, hipSetDevice(c->device));
    for (int s = 0; s < n; s++) { const int rcv = mapper_tables_to_host(c, ms[s]); if (rcv) return rcv; }
    hipStream_t st = c->stream;
    JobScratch js{ ms[0] };
    HIP_TRY(c, hipStreamSynchronize(st));        // the previous frame's kernels are done with the job scratch (normally a no-op: a frame ends with a read-back)
    std::vector<FrameState> F((size_t)n);
    int rc;
    MapperUndoAll undo;                      // every error return below leaves the mappers as they were on entry
    undo.u.reserve((size_t)n);
    for (int s = 0; s < n; s++) undo.u.emplace_back(ms[s]);
    // staging vectors of asynchronous copies live until the function returns (every path syncs the stream before that)
    std::vector<ScatterJob> sj;
    std::vector<char> vox_blob, side_blob;
    const bool prof = getenv("LMONO_MAP_PROF") != nullptr;
    auto tnow = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tp[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    tp[0] = tnow();
    // ---- phase 1 (host): transformAssociateToMap, centre cube, shifts, neighbourhood; scan cloud sizes
    std::vector<int> fn((size_t)n * 4);
    for (int s = 0; s < n; s++) {
        lmono_scan_batch *b = bs[s];
        if (b->feat_h.empty()) {
            b->feat_h.assign((size_t)b->n_scans * 4, 0);
            HIP_TRY(c, hipStreamSynchronize(st));
            HIP_TRY(c, hipMemcpy(b->feat_h.data(), b->v.feat_n, sizeof(int) * 4 * (size_t)b->n_scans, hipMemcpyDeviceToHost));
        }
        for (int k = 0; k < 4; k++) fn[(size_t)s * 4 + k] = b->feat_h[(size_t)scans[s] * 4 + k];
    }
    for (int s = 0; s < n; s++) {
        lmono_mapper *m = ms[s];
        FrameState &f = F[(size_t)s];
        double tmp[3];
        for (int k = 0; k < 8; k++) f.x[k] = 0.0;
        mp_qmul(m->q_wmap_wodom, q_wodom + 4 * s, f.x);
        mp_qrot(m->q_wmap_wodom, t_wodom + 3 * s, tmp);
        for (int k = 0; k < 3; k++) f.x[4 + k] = tmp[k] + m->t_wmap_wodom[k];
        int ci = mp_cube_of(f.x[4], m->cen[0]), cj = mp_cube_of(f.x[5], m->cen[1]), ck = mp_cube_of(f.x[6], m->cen[2]);
        while (ci < 3) { mp_shift(m, 0, +1); ci++; m->cen[0]++; }
        while (ci >= kMapW - 3) { mp_shift(m, 0, -1); ci--; m->cen[0]--; }
        while (cj < 3) { mp_shift(m, 1, +1); cj++; m->cen[1]++; }
        while (cj >= kMapH - 3) { mp_shift(m, 1, -1); cj--; m->cen[1]--; }
        while (ck < 3) { mp_shift(m, 2, +1); ck++; m->cen[2]++; }
        while (ck >= kMapD - 3) { mp_shift(m, 2, -1); ck--; m->cen[2]--; }
        for (int i = ci - 2; i <= ci + 2; i++)
            for (int j = cj - 2; j <= cj + 2; j++)
                for (int k = ck - 1; k <= ck + 1; k++)
                    if (i >= 0 && i < kMapW && j >= 0 && j < kMapH && k >= 0 && k < kMapD) f.valid.push_back(i + kMapW * j + kMapW * kMapH * k);
        f.n_last[0] = fn[(size_t)s * 4 + 1]; f.n_last[1] = fn[(size_t)s * 4 + 3];
        if (f.n_last[0] > kMapStackMax || f.n_last[1] > kMapStackMax) { c->err = "lmono_mapper: scan cloud too large"; return LMONO_ECAPACITY; }
    }
    tp[1] = tnow();
    // Round 4: the frame's first half is enqueued in one go -- the host does not wait for the voxel filter's counts before it launches the optimisation
    // (the kernels read them from the device, their launches are sized by the clouds' sizes before the filter and stride), it learns them from an event
    // while the optimisation runs, enqueues the cube assignment behind the solve, and waits ONCE for poses, statistics and cube indices.  Everything the
    // first half needs travels in ONE upload (the frame blob: poses, zeroed statistics and barrier counters, the filter's counts, all job tables).
    // ---- phase 3 first, on the side stream (both are chains of short launches that leave most of the chip idle, and neither needs the other): map clouds
    // of the neighbourhoods, concatenated in validInd order, and their grids for the streams whose map is large enough
    hipStream_t side = ms[0]->side;
    struct SideGuard {            // no return path leaves work on the side stream behind
        hipStream_t s;
        ~SideGuard() { (void)hipStreamSynchronize(s); }
    } side_guard{ side };
    std::vector<int> act;
    {
        std::vector<CopyJob> jobs;
        for (int s = 0; s < n; s++) {
            lmono_mapper *m = ms[s];
            FrameState &f = F[(size_t)s];
            for (int t = 0; t < 2; t++) {
                f.n_map[t] = 0;
                for (int ind : f.valid) {
                    const Seg &sg = m->cube[(size_t)t][(size_t)ind];
                    if (sg.n == 0) continue;
                    if (f.n_map[t] + sg.n > kMapNeighMax) { c->err = "lmono_mapper: neighbourhood holds more than 1 Mi points"; return LMONO_ECAPACITY; }
                    push_copy(jobs, m->arena[t][m->half[t]] + sg.off, m->neigh[t] + f.n_map[t], sg.n);
                    f.n_map[t] += sg.n;
                }
            }
            f.solve = f.n_map[0] > 10 && f.n_map[1] > 50;
            if (f.solve) act.push_back(s);
        }
        // one upload: [CopyJob x jobs | CloudJob x 2 act]
        std::vector<CloudJob> cj((size_t)2 * act.size());
        int max_nmap = 0;
        for (size_t a = 0; a < act.size(); a++) {
            lmono_mapper *m = ms[act[a]];
            FrameState &f = F[(size_t)act[a]];
            for (int t = 0; t < 2; t++) {
                CloudJob &J = cj[2 * a + (size_t)t];
                J.src = m->neigh[t]; J.n = f.n_map[t]; J.cell = m->cells[t]; J.tcap = m->tcap; J.sorted = m->sorted[t];
                J.slot_of = m->slot[t]; J.rank_of = m->rank[t]; J.mask_out = m->masks + t; J.bump = m->masks + 2 + t;
                max_nmap = std::max(max_nmap, f.n_map[t]);
            }
        }
        if (!jobs.empty() || !cj.empty()) {
            const size_t cj_at = (jobs.size() * sizeof(CopyJob) + 15) & ~(size_t)15;
            std::vector<char> &blob = side_blob;          // function scope: alive until every stream has been waited for
            blob.resize(cj_at + cj.size() * sizeof(CloudJob));
            if (!jobs.empty()) memcpy(blob.data(), jobs.data(), jobs.size() * sizeof(CopyJob));
            if (!cj.empty()) memcpy(blob.data() + cj_at, cj.data(), cj.size() * sizeof(CloudJob));
            if ((rc = js.upload(c, blob.data(), blob.size(), side))) return rc;
            if (!jobs.empty()) hipLaunchKernelGGL(k_copy_jobs, dim3((unsigned)jobs.size()), dim3(256), 0, side, (const CopyJob *)js.last);
            if (!cj.empty()) launch_cloud_grids(side, (const CloudJob *)((const char *)js.last + cj_at), (int)cj.size(), max_nmap);
        }
        HIP_TRY(c, hipEventRecord(ms[0]->ev_side, side));
    }
    // ---- the frame blob: [x 8n doubles | stats 8n | barriers 16n | filter counts 2n | VoxJob 2n | tile table | MapStream act | AssignJob 2n]
    if ((rc = pin_grow(c, ms[0], ms[0]->pin_i, ms[0]->pin_i_cap, (size_t)2 * n)) || (rc = pin_grow(c, ms[0], ms[0]->pin_x, ms[0]->pin_x_cap, (size_t)12 * n))) return rc;
    int *ns_h = ms[0]->pin_i;                                       // pinned: [n][2] filter counts
    double *xh = ms[0]->pin_x;                                      // pinned: [n][8] poses, then [n][8] statistics (ints)
    int *stats = (int *)(ms[0]->pin_x + (size_t)8 * n);
    for (size_t k = 0; k < (size_t)8 * n; k++) stats[k] = 0;
    const double *x_d = nullptr;
    const AssignJob *aj_d = nullptr;
    int n_last_total = 0;
    {
        std::vector<VoxJob> vj((size_t)2 * n);
        std::vector<int> tab;
        std::vector<MapStream> S(act.size());
        std::vector<AssignJob> aj((size_t)2 * n);
        for (int s = 0; s < n; s++) n_last_total += F[(size_t)s].n_last[0] + F[(size_t)s].n_last[1];
        if ((rc = mp_grow(c, ms[0], ms[0]->cubebuf, ms[0]->cubebuf_cap, (size_t)n_last_total + 1))) return rc;
        // sizes first: the tile table needs the jobs' sizes only
        for (int s = 0; s < n; s++) for (int t = 0; t < 2; t++) vj[(size_t)2 * s + t].n = F[(size_t)s].n_last[t];
        vox_tile_table(vj.data(), vj.size(), tab);
        auto al = [](size_t v) { return (v + 15) & ~(size_t)15; };
        const size_t o_x = 0, o_stats = o_x + sizeof(double) * 8 * (size_t)n, o_bar = o_stats + sizeof(int) * 8 * (size_t)n, o_ns = o_bar + sizeof(int) * 16 * (size_t)n,
                     o_vox = al(o_ns + sizeof(int) * 2 * (size_t)n), o_tab = al(o_vox + vj.size() * sizeof(VoxJob)), o_S = al(o_tab + tab.size() * sizeof(int)),
                     o_aj = al(o_S + S.size() * sizeof(MapStream)), bytes = o_aj + aj.size() * sizeof(AssignJob);
        if ((rc = js.place(c, bytes))) return rc;
        char *base = (char *)js.last;
        double *xb = (double *)(base + o_x);
        int *statbuf = (int *)(base + o_stats), *nsb = (int *)(base + o_ns);
        unsigned int *barbuf = (unsigned int *)(base + o_bar);
        x_d = xb; aj_d = (const AssignJob *)(base + o_aj);
        for (int s = 0; s < n; s++)
            for (int t = 0; t < 2; t++) {
                lmono_mapper *m = ms[s];
                VoxJob &J = vj[(size_t)2 * s + t];
                J.in = t ? bs[s]->v.less_flat + bs[s]->off_h[(size_t)scans[s]] : bs[s]->v.less_sharp + (size_t)scans[s] * kMaxLessSharp;
                J.inv_leaf = 1.0f / m->leaf[t]; J.out = m->stack[t]; J.n_out = nsb + 2 * s + t;
                J.key_a = m->vk[t]; J.key_b = m->vk[t] + kMapStackMax; J.idx_a = m->vi[t]; J.idx_b = m->vi[t] + kMapStackMax;
                J.ws = m->vws[t];
                aj[(size_t)2 * s + t] = { m->stack[t], 0, nsb, 2 * s + t, xb + 8 * s, m->cen[0], m->cen[1], m->cen[2], m->newpts[t], ms[0]->cubebuf };
            }
        int max_nq = 0;        // upper bound: the clouds before the filter
        for (size_t a = 0; a < act.size(); a++) {
            lmono_mapper *m = ms[act[a]];
            FrameState &f = F[(size_t)act[a]];
            for (int t = 0; t < 2; t++) {
                S[a].cell[t] = m->cells[t]; S[a].sorted[t] = m->sorted[t]; S[a].cloud[t] = m->neigh[t]; S[a].mask[t] = m->masks + t; S[a].n_map[t] = f.n_map[t];
                S[a].stack[t] = m->stack[t]; S[a].n_stack[t] = 0;
            }
            S[a].n_stack_d = nsb + 2 * act[a];
            S[a].rec = m->rec; S[a].x = xb + 8 * act[a]; S[a].stats = statbuf + 8 * act[a]; S[a].nn_out = nullptr; S[a].nn_tmp = m->nn_tmp;
            S[a].part = m->solve_part; S[a].bar = barbuf + 16 * act[a];
            max_nq = std::max(max_nq, f.n_last[0] + f.n_last[1]);
        }
        vox_blob.assign(bytes, 0);                                   // statistics, barrier counters and filter counts start at zero
        for (int s = 0; s < n; s++) for (int k = 0; k < 8; k++) ((double *)(vox_blob.data() + o_x))[(size_t)8 * s + k] = F[(size_t)s].x[k];
        memcpy(vox_blob.data() + o_vox, vj.data(), vj.size() * sizeof(VoxJob));
        if (!tab.empty()) memcpy(vox_blob.data() + o_tab, tab.data(), tab.size() * sizeof(int));
        if (!S.empty()) memcpy(vox_blob.data() + o_S, S.data(), S.size() * sizeof(MapStream));
        memcpy(vox_blob.data() + o_aj, aj.data(), aj.size() * sizeof(AssignJob));
        if ((rc = js.send(c, vox_blob.data(), bytes, st))) return rc;
        // ---- phase 2: VoxelGrid of the scan clouds (read in place from the scan batches), on the main stream
        launch_voxel_jobs(st, (const VoxJob *)(base + o_vox), (const int *)(base + o_tab), (int)tab.size(), 4);
        HIP_TRY(c, hipMemcpyAsync(ns_h, nsb, sizeof(int) * 2 * (size_t)n, hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipEventRecord(ms[0]->ev_sizes, st));
        tp[2] = tnow();
        tp[3] = tp[2];
        // ---- phase 4: optimisation (2 x [correspond + solve]) behind both streams, enqueued at once
        HIP_TRY(c, hipStreamWaitEvent(st, ms[0]->ev_side, 0));
        if (!act.empty()) {
            const MapStream *S_d = (const MapStream *)(base + o_S);
            // the filter keeps a fraction of a scan cloud: a quarter of the bound's blocks (at least 256 per stream while few streams run) stride over the rest
            const int per_stream = std::max(1, std::max((max_nq + 31) / 32, std::min((max_nq + 7) / 8, (int)(2048 / act.size()))));
            for (int outer = 0; outer < 2; outer++) {
                if (max_nq > 0) {
                    // (from 8 streams on: every stream's workgroups on one XCD -- map_block_of; LMONO_MAP_XCD=0: the 2-D launch, for measurements)
                    static const bool xcd_off = [] { const char *e = getenv("LMONO_MAP_XCD"); return e && atoi(e) == 0; }();
                    int nx_c = 0, nx_f = 0;
                    const int na = (int)act.size(), per_f = std::max(1, (per_stream + 7) / 8);
                    const dim3 g_c = xcd_off ? dim3((unsigned)per_stream, (unsigned)na) : map_stream_grid(per_stream, na, nx_c);
                    const dim3 g_f = xcd_off ? dim3((unsigned)per_f, (unsigned)na) : map_stream_grid(per_f, na, nx_f);
                    hipLaunchKernelGGL(k_map_correspond, g_c, dim3(256), 0, st, S_d, outer, nx_c, na);
                    hipLaunchKernelGGL(k_map_factor, g_f, dim3(64), 0, st, S_d, outer, nx_f, na);
                }
                launch_map_solve(st, S_d, (int)act.size(), outer, c->map_budget);
            }
        }
    }
    // the filter's counts (the optimisation is running): exact sizes for the assignment and the copies back
    HIP_TRY(c, hipEventSynchronize(ms[0]->ev_sizes));
    for (int s = 0; s < n; s++) { F[(size_t)s].n_stack[0] = ns_h[(size_t)2 * s]; F[(size_t)s].n_stack[1] = ns_h[(size_t)2 * s + 1]; }
    for (int s = 0; s < n; s++) if (F[(size_t)s].n_stack[0] < 0 || F[(size_t)s].n_stack[1] < 0) { c->err = "lmono_mapper: voxel filter rejected a scan cloud"; return LMONO_ECAPACITY; }
    // ---- phase 5: pointAssociateToMap + cube index of every stack point with the refined pose, behind the solve (its table went up with the blob: sizes
    // and the dense layout of the cube indices come from the device-side counts); then the one wait
    {
        size_t total = 0;
        int max_n = 0;
        std::vector<size_t> at((size_t)2 * n);
        for (int s = 0; s < n; s++) for (int t = 0; t < 2; t++) { at[(size_t)2 * s + t] = total; total += (size_t)F[(size_t)s].n_stack[t]; max_n = std::max(max_n, F[(size_t)s].n_stack[t]); }
        if ((rc = pin_grow(c, ms[0], ms[0]->pin_cube, ms[0]->pin_cube_cap, total + 1))) return rc;
        int *cube_all = ms[0]->pin_cube;
        if (max_n > 0) {
            hipLaunchKernelGGL(k_map_assign, dim3((max_n + 255) / 256, 2 * n), dim3(256), 0, st, aj_d);
            HIP_TRY(c, hipMemcpyAsync(cube_all, ms[0]->cubebuf, sizeof(int) * total, hipMemcpyDeviceToHost, st));
        }
        if (!act.empty()) HIP_TRY(c, hipMemcpyAsync(xh, x_d, sizeof(double) * 8 * (size_t)n + sizeof(int) * 8 * (size_t)n, hipMemcpyDeviceToHost, st));      // poses + statistics: adjacent in the blob
        HIP_TRY(c, hipStreamSynchronize(st));
        tp[4] = tnow();
        for (int s : act) if (stats[(size_t)s * 8 + 6]) { c->err = "lmono_mapper: a solve's cluster barrier timed out"; return LMONO_ENODEV; }
        for (int s : act) for (int k = 0; k < 8; k++) F[(size_t)s].x[k] = xh[(size_t)8 * s + k];
        for (int s = 0; s < n; s++)
            for (int t = 0; t < 2; t++) {
                const int ns = F[(size_t)s].n_stack[t];
                (void)ns;
                F[(size_t)s].cube_h[t] = cube_all + at[(size_t)2 * s + t];
            }
    }
    // results, transformUpdate (host)
    for (int s = 0; s < n; s++) {
        lmono_mapper *m = ms[s];
        FrameState &f = F[(size_t)s];
        for (int k = 0; k < 4; k++) q_w_curr[4 * s + k] = f.x[k];
        for (int k = 0; k < 3; k++) t_w_curr[3 * s + k] = f.x[4 + k];
        if (stats_h) {
            for (int k = 0; k < 6; k++) stats_h[(size_t)s * 8 + k] = stats[(size_t)s * 8 + k];
            stats_h[(size_t)s * 8 + 6] = f.n_map[0] + f.n_map[1];       // map points of the neighbourhood
            stats_h[(size_t)s * 8 + 7] = 0;                             // points of the cubes the update rebuilds: added below
        }
        const double *qo = q_wodom + 4 * s, *to = t_wodom + 3 * s;
        const double n2 = qo[0] * qo[0] + qo[1] * qo[1] + qo[2] * qo[2] + qo[3] * qo[3];
        const double qi[4] = { -qo[0] / n2, -qo[1] / n2, -qo[2] / n2, qo[3] / n2 };
        double tmp[3];
        mp_qmul(f.x, qi, m->q_wmap_wodom);
        mp_qrot(m->q_wmap_wodom, to, tmp);
        for (int k = 0; k < 3; k++) m->t_wmap_wodom[k] = f.x[4 + k] - tmp[k];
    }
    tp[5] = tnow();
    // ---- phase 6: the scans join the cubes.  Host: per touched cube [old points | new points in stack order]; device: build
    // them, re-filter the cubes of the neighbourhoods into fresh arena space, the other touched cubes keep [old | new]
    struct Touched { int s, t, ind, n_in; int64_t cat_off; bool filter; };
    std::vector<Touched> touched;
    std::vector<CopyJob> copy;
    // (the per-cube tables are allocated once per call and only the entries a stream touched are reset: 64 streams x 2 types x 4851 cubes of
    // fresh vectors were a millisecond of host time per batched frame)
    size_t pos_total = 0;
    int max_ns = 0;
    std::vector<size_t> pos_at((size_t)2 * n);
    for (int s = 0; s < n; s++) for (int t = 0; t < 2; t++) { pos_at[(size_t)2 * s + t] = pos_total; pos_total += (size_t)F[(size_t)s].n_stack[t]; max_ns = std::max(max_ns, F[(size_t)s].n_stack[t]); }
    // the update's tables travel in ONE upload from a pinned staging buffer, the placements first (they are written straight into it):
    // [pos | CopyJob x copy | ScatterJob x 2 n | VoxJob x vox | tile table | CopyJob x keep]; the tables behind the placements are bounded per stream
    auto al16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t o_copy = al16((pos_total + 1) * sizeof(int)), tables_cap = (size_t)n * (96 << 10);
    if ((rc = pin_grow(c, ms[0], ms[0]->pin_blob, ms[0]->pin_blob_cap, o_copy + tables_cap))) return rc;
    int *pos_all = (int *)ms[0]->pin_blob;
    sj.resize((size_t)2 * n);
    {
        // The streams are independent: every stream's plan is made by one worker thread into the stream's own lists (touched cubes, copy jobs) and its own
        // range of the placements; the lists are joined in stream order afterwards, so the tables are the ones a single loop over the streams writes.
        // (64 streams on the caller's thread: 0.75 ms of a 3.5-ms batched frame with the GPU idle.)
        typedef MapPlanScratch::Run Run;
        typedef MapPlanScratch PlanScratch;
        struct PlanOut { std::vector<Touched> touched; std::vector<CopyJob> copy; const char *err = nullptr; };
        if (!c->workers && n >= 8) {
            int nt = 0;
            if (const char *e = getenv("LMONO_LIB_THREADS")) nt = atoi(e);
            if (nt <= 0) nt = (int)std::min(8u, std::max(1u, std::thread::hardware_concurrency()));        // (4 / 8 / 16 threads: plan 0.36 / 0.26 / 0.25 ms for 64 streams)
            c->workers.reset(new HostWorkers(nt));
        }
        const int T = c->workers && n >= 8 ? c->workers->threads() : 1;
        if ((int)c->plan_scratch.size() < T) c->plan_scratch.resize((size_t)T);      // (every stream leaves its thread's tables as it found them: they live from call to call)
        std::vector<PlanScratch> &scratch = c->plan_scratch;
        std::vector<PlanOut> outs((size_t)n);
        auto plan_stream = [&](int s, int tid) {
            PlanScratch &ps = scratch[(size_t)tid];
            std::vector<char> &is_valid = ps.is_valid;
            std::vector<int> &add = ps.add, &fill = ps.fill, &cand = ps.cand;
            std::vector<int64_t> &cat_off = ps.cat_off;
            std::vector<Run> &runs = ps.runs;
            PlanOut &out = outs[(size_t)s];
            lmono_mapper *m = ms[s];
            FrameState &f = F[(size_t)s];
            for (int ind : f.valid) is_valid[(size_t)ind] = 1;
            for (int t = 0; t < 2 && !out.err; t++) {
                // candidates in ascending cube order: the neighbourhood and every other cube a new point falls into
                // (the stack is in voxel order: neighbours in it mostly share a cube -- the per-point work is done per RUN of equal cube indices)
                cand.assign(f.valid.begin(), f.valid.end());
                runs.clear();
                int *ph = pos_all + pos_at[(size_t)2 * s + t];
                {
                    const int *ch = f.cube_h[t];
                    const int ns = f.n_stack[t];
                    for (int i = 0; i < ns;) {
                        const int ind = ch[i];
                        int e = i + 1;
                        while (e < ns && ch[e] == ind) e++;
                        if (ind >= 0) {
                            runs.push_back({ ind, i, e - i });
                            if (add[(size_t)ind] == 0 && !is_valid[(size_t)ind]) cand.push_back(ind);
                            add[(size_t)ind] += e - i;
                        } else
                            for (int k = i; k < e; k++) ph[k] = -1;        // outside the cube array: not placed
                        i = e;
                    }
                }
                std::sort(cand.begin(), cand.end());
                int64_t at = 0;
                for (int ind : cand) {
                    const Seg &sg = m->cube[(size_t)t][(size_t)ind];
                    const bool v = is_valid[(size_t)ind] != 0;
                    if (!((v && sg.n + add[(size_t)ind] > 0) || (!v && add[(size_t)ind] > 0))) continue;
                    cat_off[(size_t)ind] = at;
                    if (sg.n > 0) push_copy(out.copy, m->arena[t][m->half[t]] + sg.off, m->cat[t] + at, sg.n);
                    const int n_in = sg.n + add[(size_t)ind];
                    if (v && n_in > kVoxCloudMax && !out.err) out.err = "lmono_mapper: a cube holds more than 65536 points";
                    out.touched.push_back({ s, t, ind, n_in, at, v });
                    at += n_in;
                }
                if (at > (int64_t)kMapNeighMax + kMapStackMax && !out.err) out.err = "lmono_mapper: frame touches more points than the workspace holds";
                if (!out.err)
                    for (const Run &r : runs) {
                        const int p0 = (int)(cat_off[(size_t)r.ind] + m->cube[(size_t)t][(size_t)r.ind].n + fill[(size_t)r.ind]);
                        fill[(size_t)r.ind] += r.len;
                        for (int k = 0; k < r.len; k++) ph[r.at + k] = p0 + k;
                    }
                for (int ind : cand) { add[(size_t)ind] = 0; fill[(size_t)ind] = 0; cat_off[(size_t)ind] = -1; }
                sj[(size_t)2 * s + t] = { m->newpts[t], nullptr, f.n_stack[t], m->cat[t] };       // .pos: set when the blob is placed
            }
            for (int ind : f.valid) is_valid[(size_t)ind] = 0;
        };
        bool planned = true;
        if (T > 1) planned = c->workers->run(n, plan_stream);
        else { try { for (int s = 0; s < n; s++) plan_stream(s, 0); } catch (...) { planned = false; } }
        if (!planned) { c->plan_scratch.clear(); c->err = "lmono_mapper: out of host memory while planning the map update"; return LMONO_ENOMEM; }      // (the tables may be half-written)
        for (int s = 0; s < n; s++) if (outs[(size_t)s].err) { c->err = outs[(size_t)s].err; return LMONO_ECAPACITY; }
        size_t nt_total = 0, nc_total = 0;
        for (const PlanOut &o : outs) { nt_total += o.touched.size(); nc_total += o.copy.size(); }
        touched.reserve(nt_total); copy.reserve(nc_total);
        for (const PlanOut &o : outs) { touched.insert(touched.end(), o.touched.begin(), o.touched.end()); copy.insert(copy.end(), o.copy.begin(), o.copy.end()); }
    }
    if (stats_h) for (const Touched &T : touched) stats_h[(size_t)T.s * 8 + 7] += T.n_in;
    const double tpa = tnow();
    // arena space (an output is never larger than its input); compaction reads only the tables, `cat` is already built
    {
        std::vector<int64_t> need((size_t)2 * n, 0);
        for (const Touched &T : touched) need[(size_t)2 * T.s + T.t] += T.n_in;
        for (int s = 0; s < n; s++)
            for (int t = 0; t < 2; t++)
                if (ms[s]->bump[t] + need[(size_t)2 * s + t] > kMapArena) {
                    if ((rc = mapper_compact(ms[s], t))) return rc;
                    if (ms[s]->bump[t] + need[(size_t)2 * s + t] > kMapArena) { c->err = "lmono_mapper: map arena exhausted"; return LMONO_ECAPACITY; }
                }
    }
    std::vector<VoxJob> vox;
    std::vector<CopyJob> keep;
    std::vector<size_t> vox_t;
    std::vector<int64_t> new_off(touched.size());
    std::vector<size_t> ws_at((size_t)2 * n, 0);
    int cube_passes = 1;
    for (size_t k = 0; k < touched.size(); k++) {
        const Touched &T = touched[k];
        lmono_mapper *m = ms[T.s];
        float4 *dst = m->arena[T.t][m->half[T.t]] + m->bump[T.t];
        new_off[k] = m->bump[T.t];
        if (T.filter) {
            VoxJob J;
            J.in = m->cat[T.t] + T.cat_off; J.n = T.n_in; J.inv_leaf = 1.0f / m->leaf[T.t]; J.out = dst; J.n_out = nullptr;
            J.key_a = m->vk[T.t] + 2 * T.cat_off; J.key_b = J.key_a + T.n_in; J.idx_a = m->vi[T.t] + 2 * T.cat_off; J.idx_b = J.idx_a + T.n_in;
            size_t &wa = ws_at[(size_t)2 * T.s + T.t];
            if (wa + vox_ws_ints(T.n_in) > m->vws_cap) { c->err = "lmono_mapper: voxel workspace exhausted"; return LMONO_ECAPACITY; }
            J.ws = m->vws[T.t] + wa; wa += vox_ws_ints(T.n_in);
            // a cube is a 50 m box: at most 50 / leaf + 3 cells per axis, whatever it holds -> the passes its keys can need
            {
                const double per_axis = 50.0 / (double)m->leaf[T.t] + 3.0;
                int bits = 1;
                while (bits < 32 && std::ldexp(1.0, bits) < per_axis * per_axis * per_axis) bits++;
                cube_passes = std::max(cube_passes, std::min(4, (bits + 8) / 9));
            }
            vox.push_back(J); vox_t.push_back(k);
        } else {
            push_copy(keep, m->cat[T.t] + T.cat_off, dst, T.n_in);
        }
        m->bump[T.t] += T.n_in;
    }
    // output sizes of the filter jobs land in one array owned by the first mapper (grown on demand)
    if ((rc = pin_grow(c, ms[0], ms[0]->pin_nout, ms[0]->pin_nout_cap, vox.size() + 1))) return rc;
    int *nout_h = ms[0]->pin_nout;
    if (!vox.empty()) {
        if (vox.size() > ms[0]->nout_cap) {
            int *q = nullptr;
            size_t cap = ms[0]->nout_cap;
            while (cap < vox.size()) cap <<= 1;
            if (!mp_alloc(ms[0], q, cap)) { c->err = "lmono_mapper: allocation failed"; return LMONO_ENOMEM; }
            ms[0]->nout_big = q; ms[0]->nout_cap = cap;
        }
        for (size_t k = 0; k < vox.size(); k++) vox[k].n_out = ms[0]->nout_big + k;
    }
    // ONE upload for the whole update (layout above)
    {
        std::vector<int> tab;
        vox_tile_table(vox.data(), vox.size(), tab);
        auto al = al16;
        const size_t o_pos = 0, o_sj = al(o_copy + copy.size() * sizeof(CopyJob)), o_vox = al(o_sj + sj.size() * sizeof(ScatterJob)),
                     o_tab = al(o_vox + vox.size() * sizeof(VoxJob)), o_keep = al(o_tab + tab.size() * sizeof(int)), bytes = o_keep + keep.size() * sizeof(CopyJob);
        if (bytes > o_copy + tables_cap) { c->err = "lmono_mapper: the update's job tables exceed their staging bound"; return LMONO_ECAPACITY; }
        if ((rc = js.place(c, bytes))) return rc;
        const char *base = (const char *)js.last;
        for (int s = 0; s < n; s++) for (int t = 0; t < 2; t++) sj[(size_t)2 * s + t].pos = (const int *)(base + o_pos) + pos_at[(size_t)2 * s + t];
        char *stage = ms[0]->pin_blob;
        if (!copy.empty()) memcpy(stage + o_copy, copy.data(), copy.size() * sizeof(CopyJob));
        memcpy(stage + o_sj, sj.data(), sj.size() * sizeof(ScatterJob));
        if (!vox.empty()) memcpy(stage + o_vox, vox.data(), vox.size() * sizeof(VoxJob));
        if (!tab.empty()) memcpy(stage + o_tab, tab.data(), tab.size() * sizeof(int));
        if (!keep.empty()) memcpy(stage + o_keep, keep.data(), keep.size() * sizeof(CopyJob));
        if ((rc = js.send(c, stage, bytes, st))) return rc;
        if (!copy.empty()) hipLaunchKernelGGL(k_copy_jobs, dim3((unsigned)copy.size()), dim3(256), 0, st, (const CopyJob *)(base + o_copy));
        if (max_ns > 0) hipLaunchKernelGGL(k_scatter_pos, dim3((max_ns + 255) / 256, 2 * n), dim3(256), 0, st, (const ScatterJob *)(base + o_sj));
        if (!vox.empty()) {
            launch_voxel_jobs(st, (const VoxJob *)(base + o_vox), (const int *)(base + o_tab), (int)tab.size(), cube_passes);
            HIP_TRY(c, hipMemcpyAsync(nout_h, ms[0]->nout_big, sizeof(int) * vox.size(), hipMemcpyDeviceToHost, st));
        }
        if (!keep.empty()) hipLaunchKernelGGL(k_copy_jobs, dim3((unsigned)keep.size()), dim3(256), 0, st, (const CopyJob *)(base + o_keep));
    }
    const double tpb = tnow();
    HIP_TRY(c, hipStreamSynchronize(st));       // ONE wait for the frame's map update: filter sizes are back, `cat` and the job tables are free
    for (size_t k = 0; k < touched.size(); k++) {
        const Touched &T = touched[k];
        Seg &sg = ms[T.s]->cube[(size_t)T.t][(size_t)T.ind];
        sg.off = new_off[k];
        sg.n = T.n_in;
    }
    for (size_t v = 0; v < vox.size(); v++) {
        if (nout_h[v] < 0) { c->err = "lmono_mapper: voxel filter rejected a cube"; return LMONO_ECAPACITY; }
        const Touched &T = touched[vox_t[v]];
        ms[T.s]->cube[(size_t)T.t][(size_t)T.ind].n = nout_h[v];
    }
    tp[6] = tnow();
    rc = check_launch(c, "mapper kernels");
    if (rc) return rc;
    undo.committed = true;
    if (prof) fprintf(stderr, "MAPPROF n=%d ms: host1 %.2f voxel %.2f gather %.2f optimise %.2f assign %.2f update %.2f (plan %.2f tables+launch %.2f wait %.2f)\n", n, tp[1] - tp[0], tp[2] - tp[1], tp[3] - tp[2], tp[4] - tp[3], tp[5] - tp[4], tp[6] - tp[5], tpa - tp[5], tpb - tpa, tp[6] - tpb);
    if (prof) {
        int mx = 0, n_small = 0; long sum = 0;
        for (const VoxJob &J : vox) { mx = std::max(mx, J.n); sum += J.n; n_small += J.n <= 2048; }
        fprintf(stderr, "MAPSIZES scan %d %d -> %d %d  map %d %d  cube jobs %zu (<= 2048: %d) max %d sum %ld kept %zu\n", F[0].n_last[0], F[0].n_last[1], F[0].n_stack[0], F[0].n_stack[1],
                F[0].n_map[0], F[0].n_map[1], vox.size(), n_small, mx, sum, keep.size());
    }
    return LMONO_OK;
}

extern "C" int lmono_mapper_process(lmono_ctx *c, lmono_mapper *m, lmono_scan_batch *b, int scan, const double q_wodom[4], const double t_wodom[3],
                                    double q_w_curr[4], double t_w_curr[3], int32_t *stats_h)
{
    if (!c || !m || !b || !q_wodom || !t_wodom || !q_w_curr || !t_w_curr) return LMONO_EINVAL;
    if (!b->registered || scan < 0 || scan >= b->n_scans) return LMONO_EINVAL;
    static const bool host_tables = getenv("LMONO_MAP_HOST_TABLES") != nullptr;       // measurement switch: the round-4 frame (cube table on the host, two waits)
    if (host_tables) return lmono_mapper_process_batch(c, 1, &m, &b, &scan, q_wodom, t_wodom, q_w_curr, t_w_curr, stats_h);
    return mapper_process_dev(c, m, b, scan, q_wodom, t_wodom, q_w_curr, t_w_curr, stats_h);
}

extern "C" int lmono_mapper_cube(lmono_ctx *c, lmono_mapper *m, int which, int i, int j, int k, float *out_h, int cap)
{
    if (!c || !m || which < 0 || which > 1 || i < 0 || i >= kMapW || j < 0 || j >= kMapH || k < 0 || k >= kMapD) return LMONO_EINVAL;
    Seg s = m->cube[(size_t)which][(size_t)(i + kMapW * j + kMapW * kMapH * k)];
    if (m->dev_mode) {          // the table is on the device: behind the last frame's map update
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        int2 e;
        HIP_TRY(c, hipMemcpy(&e, &m->dev->tab[which][i + kMapW * j + kMapW * kMapH * k], sizeof(e), hipMemcpyDeviceToHost));
        s.off = e.x; s.n = e.y;
    }
    if (!out_h) return s.n;
    if (s.n > cap) return LMONO_ECAPACITY;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (s.n > 0) HIP_TRY(c, hipMemcpy(out_h, m->arena[which][m->half[which]] + s.off, sizeof(float) * 4 * (size_t)s.n, hipMemcpyDeviceToHost));
    return s.n;
}

#include "colour_abi.hip"
#include "posegraph_abi.hip"
