// colour_abi.hip -- C ABI of the colour projection (included by lmono_hip.hip after lmono_ctx is defined)
#pragma once
#include "colour.hip"

struct lmono_map_builder {
    lmono_ctx *ctx = nullptr;
    ColourCam cam{};
    int max_pts = 0;
    int64_t map_cap = 0, map_n = 0;      // rgb_map (world frame), device resident
    int last_n = 0;                      // cloud size of the last frame
    int64_t last_off = 0;                // where the last frame's world cloud starts in rgb_map
    std::vector<void *> allocs;
    float4 *cloud = nullptr;             // staging of the host-buffer entry
    unsigned char *bgr = nullptr;
    unsigned int *key = nullptr;
    unsigned char *depth = nullptr;
    int *row_cnt = nullptr;
    PtRgb *cam_out = nullptr, *map = nullptr;
    BilateralTab *tab = nullptr;
    // per-call job table + results of a batch led by this builder
    ColourJob *jobs = nullptr;
    int *results = nullptr;
    int jobs_cap = 0;
};

template <typename T> static bool mb_alloc(lmono_map_builder *m, T *&p, size_t n)
{
    void *q = nullptr;
    if (hipMalloc(&q, (n ? n : 1) * sizeof(T)) != hipSuccess) return false;
    m->allocs.push_back(q);
    p = (T *)q;
    return true;
}

extern "C" void lmono_map_builder_destroy(lmono_map_builder *m)
{
    if (!m) return;
    for (void *p : m->allocs) (void)hipFree(p);
    delete m;
}

extern "C" lmono_map_builder *lmono_map_builder_create(lmono_ctx *c, const lmono_camera *cam, int max_cloud_points, int64_t map_capacity_points)
{
    if (!c) return nullptr;
    if (!cam || cam->width < 8 || cam->height < 8 || cam->width > 8192 || cam->height > 8192 || max_cloud_points <= 0 || max_cloud_points >= (1 << 24) ||
        map_capacity_points < (int64_t)cam->width * cam->height || cam->kernel_size < 1 || cam->kernel_size > kMaxMorphK || cam->kernel_size % 2 == 0 ||
        !(cam->fx != 0.0) || !(cam->fy != 0.0)) {
        c->err = "lmono_map_builder_create: bad camera / capacities (odd kernel_size <= 11, image 8..8192, map capacity >= one image)";
        return nullptr;
    }
    if (hipSetDevice(c->device) != hipSuccess) { c->err = "hipSetDevice failed"; return nullptr; }
    lmono_map_builder *m = new lmono_map_builder();
    m->ctx = c;
    ColourCam &k = m->cam;
    k.w = cam->width; k.h = cam->height;
    k.fx = cam->fx; k.fy = cam->fy; k.cx = cam->cx; k.cy = cam->cy; k.k1 = cam->k1; k.k2 = cam->k2; k.p1 = cam->p1; k.p2 = cam->p2;
    k.ik11 = 1.0 / k.fx; k.ik13 = -k.cx / k.fx; k.ik22 = 1.0 / k.fy; k.ik23 = -k.cy / k.fy;
    k.distort = !(k.k1 == 0.0 && k.k2 == 0.0 && k.p1 == 0.0 && k.p2 == 0.0);
    k.ksize = cam->kernel_size; k.blur = cam->blur_type == 0 ? 0 : 1;
    {   // cv::getStructuringElement: FULL -> MORPH_RECT, CROSS -> MORPH_CROSS, anything else -> MORPH_ELLIPSE (Map_Builder.cc:343-355)
        const int K = k.ksize, r = K / 2;
        const double inv_r2 = r ? 1.0 / ((double)r * r) : 0.0;
        for (int i = 0; i < K; i++) {
            int j1 = 0, j2 = 0;
            if (cam->kernel_type == 0 || (cam->kernel_type == 1 && i == r)) { j1 = 0; j2 = K; }
            else if (cam->kernel_type == 1) { j1 = r; j2 = r + 1; }
            else {
                const int dy = i - r;
                const int dx = (int)std::lrint(r * std::sqrt((r * r - dy * dy) * inv_r2));
                j1 = std::max(r - dx, 0); j2 = std::min(r + dx + 1, K);
            }
            for (int j = 0; j < K; j++) k.mask[i * K + j] = (j >= j1 && j < j2) ? 1 : 0;
        }
        k.mask_rect = 1;
        for (int i = 0; i < K * K; i++) if (!k.mask[i]) k.mask_rect = 0;
    }
    m->max_pts = max_cloud_points;
    m->map_cap = map_capacity_points;
    const size_t np = (size_t)k.w * k.h;
    bool ok = mb_alloc(m, m->cloud, (size_t)max_cloud_points) && mb_alloc(m, m->bgr, np * 3) && mb_alloc(m, m->key, np) && mb_alloc(m, m->depth, np) &&
              mb_alloc(m, m->row_cnt, (size_t)k.h) && mb_alloc(m, m->cam_out, np) && mb_alloc(m, m->map, (size_t)map_capacity_points) && mb_alloc(m, m->tab, 1);
    if (ok) {
        // cv::bilateralFilter(src, dst, 5, 1.5, 2.0) weight tables (Map_Builder.cc:398): d = 5 -> radius 2, taps with r <= radius in row-major order
        BilateralTab t{};
        const double sigma_color = 1.5, sigma_space = 2.0;
        const double cc = -0.5 / (sigma_color * sigma_color), sc = -0.5 / (sigma_space * sigma_space);
        for (int i = 0; i < 256; i++) t.color[i] = (float)std::exp(i * i * cc);
        int n = 0;
        for (int i = -2; i <= 2; i++)
            for (int j = -2; j <= 2; j++) {
                const double r = std::sqrt((double)i * i + (double)j * j);
                if (r > 2) continue;
                t.space[n] = (float)std::exp(r * r * sc); t.di[n] = i; t.dj[n] = j; n++;
            }
        ok = hipMemcpy(m->tab, &t, sizeof t, hipMemcpyHostToDevice) == hipSuccess && hipMemset(m->key, 0, np * sizeof(unsigned int)) == hipSuccess &&
             hipMemset(m->depth, 0, np) == hipSuccess;
    }
    if (!ok) { c->err = "lmono_map_builder_create: device allocation failed"; lmono_map_builder_destroy(m); return nullptr; }
    return m;
}

static void mb_rotation(const double *q, double *R)     // Eigen::Quaterniond::toRotationMatrix (Map_Builder.cc:318)
{
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1.0 - (txx + tyy);
}

static int colour_launch_fill(lmono_ctx *c, int K, dim3 grid, const ColourJob *jobs, const BilateralTab *tab)
{
    switch (K) {
    case 1: k_depth_fill<1><<<grid, kColT, 0, c->stream>>>(jobs, tab); break;
    case 3: k_depth_fill<3><<<grid, kColT, 0, c->stream>>>(jobs, tab); break;
    case 5: k_depth_fill<5><<<grid, kColT, 0, c->stream>>>(jobs, tab); break;
    case 7: k_depth_fill<7><<<grid, kColT, 0, c->stream>>>(jobs, tab); break;
    case 9: k_depth_fill<9><<<grid, kColT, 0, c->stream>>>(jobs, tab); break;
    case 11: k_depth_fill<11><<<grid, kColT, 0, c->stream>>>(jobs, tab); break;
    default: c->err = "k_depth_fill: unsupported kernel size"; return LMONO_EINVAL;
    }
    return check_launch(c, "k_depth_fill");
}

extern "C" int lmono_associate_to_map_batch(lmono_ctx *c, int n_streams, lmono_map_builder *const *mbs, const float *const *xyzi_d, const int *n_points,
                                            const double *transforms, const uint8_t *const *bgr_d, const double *q_wc, const double *t_wc, int *n_out)
{
    if (!c || n_streams <= 0 || !mbs || !xyzi_d || !n_points || !transforms || !bgr_d || !q_wc || !t_wc) return LMONO_EINVAL;
    for (int s = 0; s < n_streams; s++) {
        if (!mbs[s] || mbs[s]->ctx != c || !bgr_d[s] || n_points[s] < 0 || n_points[s] >= (1 << 24) || (n_points[s] > 0 && !xyzi_d[s])) { c->err = "lmono_associate_to_map_batch: bad stream arguments"; return LMONO_EINVAL; }
        for (int u = 0; u < s; u++) if (mbs[u] == mbs[s]) { c->err = "lmono_associate_to_map_batch: map builders must be distinct"; return LMONO_EINVAL; }
        if (mbs[s]->map_n + (int64_t)mbs[s]->cam.w * mbs[s]->cam.h > mbs[s]->map_cap) { c->err = "lmono_associate_to_map_batch: rgb_map is full (clear it or create it larger)"; return LMONO_ECAPACITY; }
    }
    lmono_map_builder *lead = mbs[0];
    if (lead->jobs_cap < n_streams) {
        int cap = std::max(lead->jobs_cap, 1);
        while (cap < n_streams) cap <<= 1;
        ColourJob *jb = nullptr; int *rs = nullptr;
        if (!mb_alloc(lead, jb, (size_t)cap) || !mb_alloc(lead, rs, (size_t)cap * 2)) { c->err = "lmono_associate_to_map_batch: job table allocation failed"; return LMONO_ENOMEM; }
        lead->jobs = jb; lead->results = rs; lead->jobs_cap = cap;
    }
    std::vector<ColourJob> jobs((size_t)n_streams);
    int max_n = 0, max_tiles = 0, max_h = 0;
    std::vector<int> ks;
    for (int s = 0; s < n_streams; s++) {
        lmono_map_builder *m = mbs[s];
        ColourJob &j = jobs[(size_t)s];
        j.cam = m->cam;
        j.cloud = (const float4 *)xyzi_d[s]; j.n = n_points[s];
        j.bgr = bgr_d[s]; j.key = m->key; j.depth = m->depth; j.row_cnt = m->row_cnt;
        j.cam_out = m->cam_out; j.world_out = m->map + m->map_n; j.world_cap = m->map_cap - m->map_n;
        j.n_out = lead->results + 2 * s;
        const double *M = transforms + 16 * (size_t)s;
        for (int k = 0; k < 12; k++) j.M[k] = M[k];
        mb_rotation(q_wc + 4 * (size_t)s, j.R);
        for (int k = 0; k < 3; k++) j.T[k] = t_wc[3 * (size_t)s + k];
        max_n = std::max(max_n, j.n);
        max_tiles = std::max(max_tiles, ((m->cam.w + kFillTW - 1) / kFillTW) * ((m->cam.h + kFillTH - 1) / kFillTH));
        max_h = std::max(max_h, m->cam.h);
        if (std::find(ks.begin(), ks.end(), m->cam.ksize) == ks.end()) ks.push_back(m->cam.ksize);
    }
    HIP_TRY(c, hipMemcpyAsync(lead->jobs, jobs.data(), sizeof(ColourJob) * (size_t)n_streams, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(lead->results, 0, sizeof(int) * 2 * (size_t)n_streams, c->stream));
    if (max_n > 0) {
        k_colour_splat<<<dim3((unsigned)((max_n + kColT - 1) / kColT), (unsigned)n_streams), kColT, 0, c->stream>>>(lead->jobs);
        if (int rc = check_launch(c, "k_colour_splat")) return rc;
    }
    for (int K : ks)
        if (int rc = colour_launch_fill(c, K, dim3((unsigned)max_tiles, (unsigned)n_streams), lead->jobs, lead->tab)) return rc;
    k_colour_count<<<dim3((unsigned)max_h, (unsigned)n_streams), kColT, 0, c->stream>>>(lead->jobs);
    if (int rc = check_launch(c, "k_colour_count")) return rc;
    k_colour_lift<<<dim3((unsigned)max_h, (unsigned)n_streams), kColT, 0, c->stream>>>(lead->jobs);
    if (int rc = check_launch(c, "k_colour_lift")) return rc;
    std::vector<int> res((size_t)n_streams * 2);
    HIP_TRY(c, hipMemcpyAsync(res.data(), lead->results, sizeof(int) * res.size(), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    int rc = LMONO_OK;
    for (int s = 0; s < n_streams; s++) {
        lmono_map_builder *m = mbs[s];
        m->last_n = res[(size_t)2 * s]; m->last_off = m->map_n;
        if (res[(size_t)2 * s + 1]) { c->err = "lmono_associate_to_map_batch: rgb_map overflow"; rc = LMONO_ECAPACITY; }
        else m->map_n += m->last_n;                            // processMapping: *rgb_map += rgb_cloud (Map_Builder.cc:52-60)
        if (n_out) n_out[s] = m->last_n;
    }
    return rc;
}

extern "C" int lmono_associate_to_map(lmono_ctx *c, lmono_map_builder *m, const float *xyzi_h, int n_points, const double transform[16],
                                      const uint8_t *bgr_h, const double q_wc[4], const double t_wc[3], int *n_out)
{
    if (!c || !m || m->ctx != c || !bgr_h || !transform || !q_wc || !t_wc || n_points < 0 || (n_points > 0 && !xyzi_h)) return LMONO_EINVAL;
    if (n_points > m->max_pts) { c->err = "lmono_associate_to_map: cloud larger than max_cloud_points"; return LMONO_ECAPACITY; }
    if (n_points > 0) HIP_TRY(c, hipMemcpyAsync(m->cloud, xyzi_h, sizeof(float4) * (size_t)n_points, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(m->bgr, bgr_h, (size_t)m->cam.w * m->cam.h * 3, hipMemcpyHostToDevice, c->stream));
    const float *xd = (const float *)m->cloud;
    const uint8_t *bd = m->bgr;
    return lmono_associate_to_map_batch(c, 1, &m, &xd, &n_points, transform, &bd, q_wc, t_wc, n_out);
}

extern "C" int lmono_map_builder_depth(lmono_ctx *c, lmono_map_builder *m, uint8_t *depth_h)
{
    if (!c || !m || m->ctx != c || !depth_h) return LMONO_EINVAL;
    HIP_TRY(c, hipMemcpyAsync(depth_h, m->depth, (size_t)m->cam.w * m->cam.h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return LMONO_OK;
}

extern "C" int lmono_map_builder_cloud(lmono_ctx *c, lmono_map_builder *m, int which, lmono_point_rgb *out_h, int cap)
{
    if (!c || !m || m->ctx != c || which < 0 || which > 1) return LMONO_EINVAL;
    if (out_h) {
        if (cap < m->last_n) { c->err = "lmono_map_builder_cloud: output capacity too small"; return LMONO_ECAPACITY; }
        if (which == 1 && m->last_off + m->last_n > m->map_n) { c->err = "lmono_map_builder_cloud: the world cloud of the last frame is no longer in rgb_map"; return LMONO_EINVAL; }
        const PtRgb *src = which == 0 ? m->cam_out : m->map + m->last_off;
        if (m->last_n > 0) HIP_TRY(c, hipMemcpyAsync(out_h, src, sizeof(PtRgb) * (size_t)m->last_n, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return m->last_n;
}

extern "C" int64_t lmono_map_builder_map(lmono_ctx *c, lmono_map_builder *m, lmono_point_rgb *out_h, int64_t cap)
{
    if (!c || !m || m->ctx != c) return LMONO_EINVAL;
    if (out_h) {
        if (cap < m->map_n) { c->err = "lmono_map_builder_map: output capacity too small"; return LMONO_ECAPACITY; }
        if (m->map_n > 0 && hipMemcpyAsync(out_h, m->map, sizeof(PtRgb) * (size_t)m->map_n, hipMemcpyDeviceToHost, c->stream) != hipSuccess) { c->err = "lmono_map_builder_map: copy failed"; return LMONO_ENODEV; }
        if (hipStreamSynchronize(c->stream) != hipSuccess) { c->err = "lmono_map_builder_map: synchronize failed"; return LMONO_ENODEV; }
    }
    return m->map_n;
}

extern "C" int lmono_map_builder_clear(lmono_ctx *c, lmono_map_builder *m)
{
    if (!c || !m || m->ctx != c) return LMONO_EINVAL;
    m->map_n = 0;                                               // rgb_map->clear() (Map_Builder.cc:81)
    return LMONO_OK;
}
