// posegraph_abi.hip -- C ABI of the loop-closure pose graph (included by lmono_hip.hip after lmono_ctx is defined).
// Graph construction (edges, incidence lists, reverse Cuthill-McKee order) is host-side set-up done once per graph; the
// numeric rounds run in k_pg_linearise / k_pg_step (posegraph.hip).
#pragma once
#include "posegraph.hip"

#include <cmath>
#include <numeric>

struct lmono_pose_graph {
    lmono_ctx *ctx = nullptr;
    PgView v{};
    int n = 0, n_edges = 0, w = 0;
    int64_t reduce_count = 0;
    std::vector<void *> allocs;
    std::vector<double> pitch_h, roll_h, x0_h;
};

template <typename T> static bool pg_upload(lmono_pose_graph *g, const T *&dst, const std::vector<T> &src)
{
    void *q = nullptr;
    if (hipMalloc(&q, std::max<size_t>(src.size(), 1) * sizeof(T)) != hipSuccess) return false;
    g->allocs.push_back(q);
    if (!src.empty() && hipMemcpy(q, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return false;
    dst = (const T *)q;
    return true;
}
template <typename T> static bool pg_alloc(lmono_pose_graph *g, T *&dst, size_t count)
{
    void *q = nullptr;
    if (hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) return false;
    g->allocs.push_back(q);
    if (hipMemset(q, 0, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) return false;
    dst = (T *)q;
    return true;
}

extern "C" void lmono_pose_graph_destroy(lmono_pose_graph *g)
{
    if (!g) return;
    for (void *p : g->allocs) (void)hipFree(p);
    delete g;
}

// mathutils::R2ypr (include/utils/math_utils.h:187-202) of the rotation of q (x y z w), degrees
static void pg_q2ypr(const double *q, double *ypr)
{
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double R00 = 1 - 2 * (y * y + z * z), R01 = 2 * (x * y - z * w), R02 = 2 * (x * z + y * w);
    const double R10 = 2 * (x * y + z * w), R11 = 1 - 2 * (x * x + z * z), R12 = 2 * (y * z - x * w);
    const double R20 = 2 * (x * z - y * w);
    const double yy = std::atan2(R10, R00);
    const double pp = std::atan2(-R20, R00 * std::cos(yy) + R10 * std::sin(yy));
    const double rr = std::atan2(R02 * std::sin(yy) - R12 * std::cos(yy), -R01 * std::sin(yy) + R11 * std::cos(yy));
    ypr[0] = yy / kPgPi * 180.0; ypr[1] = pp / kPgPi * 180.0; ypr[2] = rr / kPgPi * 180.0;
}

// YawPitchRollToRotationMatrix (Loop_Detector.h:129-147) followed by Eigen::Quaterniond(Matrix3d)
static void pg_ypr2q(const double *ypr, double *q)
{
    const double y = ypr[0] / 180.0 * kPgPi, p = ypr[1] / 180.0 * kPgPi, r = ypr[2] / 180.0 * kPgPi;
    const double cy = std::cos(y), sy = std::sin(y), cp = std::cos(p), sp = std::sin(p), cr = std::cos(r), sr = std::sin(r);
    const double R[9] = { cy * cp, -sy * cr + cy * sp * sr, sy * sr + cy * sp * cr, sy * cp, cy * cr + sy * sp * sr, -cy * sr + sy * sp * cr, -sp, cp * sr, cp * cr };
    const double tr = R[0] + R[4] + R[8];
    if (tr > 0) {
        double t = std::sqrt(tr + 1.0);
        q[3] = 0.5 * t; t = 0.5 / t;
        q[0] = (R[7] - R[5]) * t; q[1] = (R[2] - R[6]) * t; q[2] = (R[3] - R[1]) * t;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[4 * i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        double t = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
        q[i] = 0.5 * t; t = 0.5 / t;
        q[3] = (R[3 * k + j] - R[3 * j + k]) * t; q[j] = (R[3 * j + i] + R[3 * i + j]) * t; q[k] = (R[3 * k + i] + R[3 * i + k]) * t;
    }
}

extern "C" lmono_pose_graph *lmono_pose_graph_create(lmono_ctx *c, int n, const double *poses_tq_h, int n_loops, const int32_t *loops_h, const double *loop_info_h)
{
    if (!c) return nullptr;
    if (n < 2 || !poses_tq_h || n_loops < 0 || (n_loops > 0 && (!loops_h || !loop_info_h))) { c->err = "lmono_pose_graph_create: bad arguments"; return nullptr; }
    for (int k = 0; k < n_loops; k++)
        if (loops_h[2 * k] < 0 || loops_h[2 * k] >= n || loops_h[2 * k + 1] < 0 || loops_h[2 * k + 1] >= n || loops_h[2 * k] == loops_h[2 * k + 1]) { c->err = "lmono_pose_graph_create: loop index out of range"; return nullptr; }
    if (hipSetDevice(c->device) != hipSuccess) { c->err = "hipSetDevice failed"; return nullptr; }
    lmono_pose_graph *g = new lmono_pose_graph();
    g->ctx = c; g->n = n;
    std::vector<double> x((size_t)n * 4);
    g->pitch_h.resize((size_t)n); g->roll_h.resize((size_t)n);
    for (int i = 0; i < n; i++) {
        double ypr[3];
        pg_q2ypr(poses_tq_h + 7 * (size_t)i + 3, ypr);
        x[4 * (size_t)i] = ypr[0]; g->pitch_h[(size_t)i] = ypr[1]; g->roll_h[(size_t)i] = ypr[2];
        for (int k = 0; k < 3; k++) x[4 * (size_t)i + 1 + k] = poses_tq_h[7 * (size_t)i + k];
    }
    // edges: every keyframe to its (up to) four predecessors with the odometry's relative pose, then the loops
    std::vector<int> ea, eb, el;
    std::vector<double> em;
    for (int i = 1; i < n; i++)
        for (int j = 1; j <= 4 && i - j >= 0; j++) {
            const int a = i - j;
            const double *q = poses_tq_h + 7 * (size_t)a + 3, *ta = poses_tq_h + 7 * (size_t)a, *ti = poses_tq_h + 7 * (size_t)i;
            const double d[3] = { ti[0] - ta[0], ti[1] - ta[1], ti[2] - ta[2] };
            const double qx = -q[0], qy = -q[1], qz = -q[2], qw = q[3];      // q_a^-1 (t_i - t_a)
            const double uvx = 2.0 * (qy * d[2] - qz * d[1]), uvy = 2.0 * (qz * d[0] - qx * d[2]), uvz = 2.0 * (qx * d[1] - qy * d[0]);
            ea.push_back(a); eb.push_back(i); el.push_back(0);
            em.push_back(d[0] + qw * uvx + (qy * uvz - qz * uvy));
            em.push_back(d[1] + qw * uvy + (qz * uvx - qx * uvz));
            em.push_back(d[2] + qw * uvz + (qx * uvy - qy * uvx));
            em.push_back(x[4 * (size_t)i] - x[4 * (size_t)a]);
        }
    for (int k = 0; k < n_loops; k++) {
        ea.push_back(loops_h[2 * k]); eb.push_back(loops_h[2 * k + 1]); el.push_back(1);
        for (int q = 0; q < 3; q++) em.push_back(loop_info_h[8 * (size_t)k + q]);
        em.push_back(loop_info_h[8 * (size_t)k + 7]);
    }
    const int ne = (int)ea.size();
    g->n_edges = ne;
    // incidence lists
    std::vector<int> inc_start((size_t)n + 1, 0), inc_edge((size_t)ne * 2);
    for (int e = 0; e < ne; e++) { inc_start[(size_t)ea[(size_t)e] + 1]++; inc_start[(size_t)eb[(size_t)e] + 1]++; }
    for (int i = 0; i < n; i++) inc_start[(size_t)i + 1] += inc_start[(size_t)i];
    {
        std::vector<int> fill(inc_start.begin(), inc_start.end() - 1);
        for (int e = 0; e < ne; e++) { inc_edge[(size_t)fill[(size_t)ea[(size_t)e]]++] = e; inc_edge[(size_t)fill[(size_t)eb[(size_t)e]]++] = e; }
    }
    // reverse Cuthill-McKee: breadth-first levels, neighbours by increasing degree, reversed
    std::vector<int> order; order.reserve((size_t)n);
    std::vector<char> seen((size_t)n, 0);
    auto degree = [&](int v) { return inc_start[(size_t)v + 1] - inc_start[(size_t)v]; };
    for (int root = 0; root < n; root++) {
        if (seen[(size_t)root]) continue;
        size_t head = order.size();
        order.push_back(root); seen[(size_t)root] = 1;
        while (head < order.size()) {
            const int u = order[head++];
            const size_t first = order.size();
            for (int k = inc_start[(size_t)u]; k < inc_start[(size_t)u + 1]; k++) {
                const int e = inc_edge[(size_t)k], o = ea[(size_t)e] == u ? eb[(size_t)e] : ea[(size_t)e];
                if (!seen[(size_t)o]) { seen[(size_t)o] = 1; order.push_back(o); }
            }
            std::sort(order.begin() + (long)first, order.end(), [&](int p, int q) { const int dp = degree(p), dq = degree(q); return dp != dq ? dp < dq : p < q; });
        }
    }
    std::vector<int> pos((size_t)n), node_at((size_t)n);
    for (int i = 0; i < n; i++) { pos[(size_t)order[(size_t)(n - 1 - i)]] = i; node_at[(size_t)i] = order[(size_t)(n - 1 - i)]; }
    int w = 1;
    for (int e = 0; e < ne; e++) w = std::max(w, std::abs(pos[(size_t)ea[(size_t)e]] - pos[(size_t)eb[(size_t)e]]));
    if (w > kPgMaxW) { c->err = "lmono_pose_graph_create: graph bandwidth " + std::to_string(w) + " blocks exceeds " + std::to_string(kPgMaxW); delete g; return nullptr; }
    g->w = w;
    // the band solver's panel lives in dynamic LDS (83 KB at w = 67, <= 143 KB at the widest band); per device, so set per graph
    if (hipFuncSetAttribute((const void *)k_pg_step, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(std::max(pg_panel_doubles(120), pg_panel_doubles(kPgMaxW)) * sizeof(double))) != hipSuccess) {      // the largest any graph can ask for
        c->err = "lmono_pose_graph_create: cannot reserve the band solver's LDS panel"; delete g; return nullptr;
    }
    const size_t hsz = (size_t)n * (size_t)(w + 1) * 16;
    g->reduce_count = (int64_t)(hsz + 5 * (size_t)n);
    PgView &v = g->v;
    v.n = n; v.w = w; v.n_edges = ne;
    bool ok = pg_upload(g, v.ea, ea) && pg_upload(g, v.eb, eb) && pg_upload(g, v.eloop, el) && pg_upload(g, v.emeas, em) &&
              pg_upload(g, v.inc_start, inc_start) && pg_upload(g, v.inc_edge, inc_edge) && pg_upload(g, v.pos, pos) && pg_upload(g, v.node_at, node_at) &&
              pg_upload(g, v.pitch, g->pitch_h) && pg_upload(g, v.roll, g->roll_h) &&
              pg_alloc(g, v.x, (size_t)n * 4) && pg_alloc(g, v.cand, (size_t)n * 4) && pg_alloc(g, v.lin, (size_t)g->reduce_count) &&
              pg_alloc(g, v.cur, (size_t)g->reduce_count) && pg_alloc(g, v.Aw, hsz) && pg_alloc(g, v.scale, (size_t)n * 4) && pg_alloc(g, v.diag, (size_t)n * 4) &&
              pg_alloc(g, v.gs, (size_t)n * 4) && pg_alloc(g, v.sol, (size_t)n * 4) && pg_alloc(g, v.st, 1);
    ok = ok && hipMemcpy(v.x, x.data(), x.size() * sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
    g->x0_h = x;
    if (!ok) { c->err = "lmono_pose_graph_create: device allocation failed"; lmono_pose_graph_destroy(g); return nullptr; }
    return g;
}

extern "C" int lmono_pose_graph_reset(lmono_ctx *c, lmono_pose_graph *g)
{
    if (!c || !g || g->ctx != c) return LMONO_EINVAL;
    HIP_TRY(c, hipMemcpyAsync(g->v.x, g->x0_h.data(), g->x0_h.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(g->v.st, 0, sizeof(PgState), c->stream));
    return LMONO_OK;
}

extern "C" int lmono_pose_graph_info(lmono_pose_graph *g, int64_t *reduce_count, int *bandwidth, int *n_edges)
{
    if (!g) return LMONO_EINVAL;
    if (reduce_count) *reduce_count = g->reduce_count;
    if (bandwidth) *bandwidth = g->w;
    if (n_edges) *n_edges = g->n_edges;
    return LMONO_OK;
}

extern "C" void *lmono_pose_graph_reduce_buffer(lmono_pose_graph *g) { return g ? (void *)g->v.lin : nullptr; }

extern "C" int lmono_pose_graph_set_reduce_buffer(lmono_pose_graph *g, void *buffer_d)
{
    if (!g || !buffer_d) return LMONO_EINVAL;
    g->v.lin = (double *)buffer_d;
    return LMONO_OK;
}

extern "C" int lmono_pose_graph_linearise(lmono_ctx *c, lmono_pose_graph *g, int rank, int world)
{
    if (!c || !g || g->ctx != c || world < 1 || rank < 0 || rank >= world) return LMONO_EINVAL;
    const int lo = (int)((int64_t)g->n * rank / world), hi = (int)((int64_t)g->n * (rank + 1) / world);
    HIP_TRY(c, hipMemsetAsync(g->v.lin, 0, sizeof(double) * (size_t)g->reduce_count, c->stream));
    k_pg_linearise<<<(unsigned)((g->n + 255) / 256), 256, 0, c->stream>>>(g->v, lo, hi);
    return check_launch(c, "k_pg_linearise");
}

extern "C" int lmono_pose_graph_step(lmono_ctx *c, lmono_pose_graph *g, int max_iter, int *done)
{
    if (!c || !g || g->ctx != c || max_iter < 0) return LMONO_EINVAL;
    k_pg_step<<<1, kPgT, pg_panel_doubles(g->w) * sizeof(double), c->stream>>>(g->v, max_iter);
    if (int rc = check_launch(c, "k_pg_step")) return rc;
    if (done) {
        PgState s;
        HIP_TRY(c, hipMemcpyAsync(&s, g->v.st, sizeof s, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        *done = s.done;
    }
    return LMONO_OK;
}

extern "C" int lmono_pose_graph_optimize(lmono_ctx *c, lmono_pose_graph *g, int max_iter)
{
    if (!c || !g || g->ctx != c || max_iter < 0) return LMONO_EINVAL;
    int done = 0;
    for (int round = 0; round <= max_iter && !done; round++) {
        if (int rc = lmono_pose_graph_linearise(c, g, 0, 1)) return rc;
        if (int rc = lmono_pose_graph_step(c, g, max_iter, &done)) return rc;
    }
    return LMONO_OK;
}

extern "C" int lmono_pose_graph_result(lmono_ctx *c, lmono_pose_graph *g, double *poses_tq_h, double *stats)
{
    if (!c || !g || g->ctx != c) return LMONO_EINVAL;
    std::vector<double> x((size_t)g->n * 4);
    PgState s;
    HIP_TRY(c, hipMemcpyAsync(x.data(), g->v.x, x.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(&s, g->v.st, sizeof s, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (poses_tq_h)
        for (int i = 0; i < g->n; i++) {
            const double ypr[3] = { x[4 * (size_t)i], g->pitch_h[(size_t)i], g->roll_h[(size_t)i] };
            for (int k = 0; k < 3; k++) poses_tq_h[7 * (size_t)i + k] = x[4 * (size_t)i + 1 + k];
            pg_ypr2q(ypr, poses_tq_h + 7 * (size_t)i + 3);
        }
    if (stats) { stats[0] = s.iter; stats[1] = s.cost0; stats[2] = s.x_cost; stats[3] = g->w; stats[4] = s.accepted; stats[5] = s.rejected; }
    return LMONO_OK;
}
