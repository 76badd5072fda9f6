// colour.hip -- colour projection of lmono's map builder on gfx950.
//
// One frame = MapBuilder::associateToMap (mono_lidar_mapping/src/map_builder/Map_Builder.cc:213-334) with the cloud
// transform of the node fused in (map_build_node.cc:216-225):
//   k_colour_splat   transform + PinholeCamera::spaceToPlane + 8-bit depth splat 100 - z (:225-247).  The reference
//                    loop is sequential ("the last point of a pixel wins"): a 32-bit atomicMax on (index + 1) << 8 | value
//                    reproduces that order-independently.
//   k_depth_fill<K>  MapBuilder::depthFill (:336-403) fused into one LDS-tiled kernel: dilate(K, KERNEL_TYPE) ->
//                    close(K rect) -> dilate(7 rect) + zero fill -> median 5 -> bilateral 5 (1.5, 2.0) | Gaussian 5x5.
//                    A 64 x 32 output tile carries a halo of 7 + 3 (K/2) pixels; every stage applies OpenCV's border
//                    rule of that stage at the image border (ignore / replicate / reflect-101).
//   k_colour_count   per image row: pixels that survive the tests of the back-projection loop (:275-310); also re-zeroes
//                    the splat keys for the next frame.
//   k_colour_lift    PinholeCamera::liftProjective back-projection with the frame's colour in row-major pixel order
//                    (stable compaction: row base = sum of the previous row counts, in-row ballot prefix), camera-frame cloud
//                    (pub_rgb_points_) and world-frame cloud (:315-322) appended to the device-resident rgb_map.
// All of it is HBM / LDS bound byte work; nothing here is reshaped for the matrix cores.
#pragma once
#include "common.hpp"
#include "median_net.hpp"

namespace lmono {

constexpr int kColT = 256;
constexpr int kFillTW = 64, kFillTH = 32;
constexpr int kMaxMorphK = 11;

struct PtRgb { float x, y, z; unsigned int bgra; };

struct ColourCam {
    int w, h;
    double fx, fy, cx, cy, k1, k2, p1, p2;
    double ik11, ik13, ik22, ik23;     // PinholeCamera.cc:292-295
    int distort;                       // PinholeCamera.cc:278-289
    int ksize, blur;
    int mask_rect;                     // every element of mask set (KERNEL_TYPE FULL): the separable path applies
    unsigned char mask[kMaxMorphK * kMaxMorphK];
};

struct ColourJob {
    ColourCam cam;
    const float4 *cloud;
    int n;
    const unsigned char *bgr;
    unsigned int *key;
    unsigned char *depth;
    int *row_cnt;
    PtRgb *cam_out, *world_out;
    long long world_cap;
    int *n_out;                        // [0] points of the frame, [1] 1 when the world cloud did not fit
    double M[12], R[9], T[3];
};

struct BilateralTab { float color[256]; float space[13]; int di[13], dj[13]; };

// PinholeCamera::distortion, PinholeCamera.cc:646-662
__device__ __forceinline__ void col_distortion(const ColourCam &c, double ux, double uy, double &dx, double &dy)
{
    const double mx2 = ux * ux, my2 = uy * uy, mxy = ux * uy, rho2 = mx2 + my2;
    const double rad = c.k1 * rho2 + c.k2 * rho2 * rho2;
    dx = ux * rad + 2.0 * c.p1 * mxy + c.p2 * (rho2 + 2.0 * mx2);
    dy = uy * rad + 2.0 * c.p2 * mxy + c.p1 * (rho2 + 2.0 * my2);
}

__global__ __launch_bounds__(kColT) void k_colour_splat(const ColourJob *jobs)
{
    const ColourJob &j = jobs[blockIdx.y];
    const int i = blockIdx.x * kColT + threadIdx.x;
    if (i >= j.n) return;
    const float4 p = j.cloud[i];
    const double x = p.x, y = p.y, z = p.z;
    const float px = (float)(j.M[0] * x + j.M[1] * y + j.M[2] * z + j.M[3]);
    const float py = (float)(j.M[4] * x + j.M[5] * y + j.M[6] * z + j.M[7]);
    const float pz = (float)(j.M[8] * x + j.M[9] * y + j.M[10] * z + j.M[11]);
    if (pz < 0) return;
    double ux = (double)px / (double)pz, uy = (double)py / (double)pz;      // spaceToPlane, PinholeCamera.cc:520-545
    if (j.cam.distort) {
        double dx, dy;
        col_distortion(j.cam, ux, uy, dx, dy);
        ux = ux + dx; uy = uy + dy;
    }
    const float u = (float)(j.cam.fx * ux + j.cam.cx), v = (float)(j.cam.fy * uy + j.cam.cy);
    if (u > 0 && u < (float)j.cam.w && v > 0 && v < (float)j.cam.h) {
        const unsigned int val = (unsigned int)(int)(100.0 - (double)pz) & 0xffu;
        atomicMax(&j.key[(int)v * j.cam.w + (int)u], ((unsigned int)(i + 1) << 8) | val);
    }
}

template <int W, int H, int MX, int MY, typename F>
__device__ __forceinline__ void fill_region(F f)
{
    constexpr int RW = W - 2 * MX, RH = H - 2 * MY;
    for (int idx = threadIdx.x; idx < RW * RH; idx += kColT) {
        const int ry = idx / RW;
        f(idx - ry * RW + MX, ry + MY);
    }
}

// separable rectangular max / min of half-width A: src (valid from margin M) -> tmp (rows M, cols M + A) -> dst (margin M + A);
// positions outside the image receive `outside` (the neutral value of the stage that reads dst next)
template <int P, int RH, int M, int A, bool kMin>
__device__ __forceinline__ void fill_rect(const unsigned char *src, unsigned char *tmp, unsigned char *dst, int ox, int oy, int w, int h, int outside)
{
    fill_region<P, RH, M + A, M>([&](int rx, int ry) {
        int v = kMin ? 255 : 0;
#pragma unroll
        for (int d = -A; d <= A; d++) { const int s = src[ry * P + rx + d]; v = kMin ? min(v, s) : max(v, s); }
        tmp[ry * P + rx] = (unsigned char)v;
    });
    __syncthreads();
    fill_region<P, RH, M + A, M + A>([&](int rx, int ry) {
        int v = kMin ? 255 : 0;
#pragma unroll
        for (int d = -A; d <= A; d++) { const int s = tmp[(ry + d) * P + rx]; v = kMin ? min(v, s) : max(v, s); }
        const int gx = ox + rx, gy = oy + ry;
        if (gx < 0 || gx >= w || gy < 0 || gy >= h) v = outside;
        dst[ry * P + rx] = (unsigned char)v;
    });
    __syncthreads();
}

template <int K>
__global__ __launch_bounds__(kColT) void k_depth_fill(const ColourJob *jobs, const BilateralTab *tab)
{
    constexpr int A = K / 2, R0 = 7 + 3 * A, P = kFillTW + 2 * R0, RH = kFillTH + 2 * R0;
    __shared__ unsigned char sA[P * RH], sB[P * RH], sC[P * RH];
    __shared__ float s_cw[256];
    const ColourJob &j = jobs[blockIdx.y];
    if (j.cam.ksize != K) return;
    const int w = j.cam.w, h = j.cam.h;
    const int tiles_x = (w + kFillTW - 1) / kFillTW, tiles_y = (h + kFillTH - 1) / kFillTH;
    if ((int)blockIdx.x >= tiles_x * tiles_y) return;
    const int ty = blockIdx.x / tiles_x, x0 = (blockIdx.x - ty * tiles_x) * kFillTW, y0 = ty * kFillTH;
    const int ox = x0 - R0, oy = y0 - R0;
    s_cw[threadIdx.x] = tab->color[threadIdx.x];
    // S0: the splat image (the value byte of the winning key)
    fill_region<P, RH, 0, 0>([&](int rx, int ry) {
        const int gx = ox + rx, gy = oy + ry;
        int v = 0;
        if (gx >= 0 && gx < w && gy >= 0 && gy < h) v = (int)(j.key[(size_t)gy * w + gx] & 0xffu);
        sA[ry * P + rx] = (unsigned char)v;
    });
    __syncthreads();
    // S1 = dilate(S0, KERNEL_TYPE element), :359-360.  FULL is separable; the general element walks its mask.
    if (j.cam.mask_rect) {
        fill_rect<P, RH, 0, A, false>(sA, sC, sB, ox, oy, w, h, 0);
    } else {
        fill_region<P, RH, A, A>([&](int rx, int ry) {
            int v = 0;
            for (int i = 0; i < K; i++)
#pragma unroll
                for (int jj = 0; jj < K; jj++)
                    if (j.cam.mask[i * K + jj]) v = max(v, (int)sA[(ry + i - A) * P + rx + jj - A]);
            const int gx = ox + rx, gy = oy + ry;
            if (gx < 0 || gx >= w || gy < 0 || gy >= h) v = 0;
            sB[ry * P + rx] = (unsigned char)v;
        });
        __syncthreads();
    }
    fill_rect<P, RH, A, A, false>(sB, sC, sA, ox, oy, w, h, 255);        // :364 close = dilate ...
    fill_rect<P, RH, 2 * A, A, true>(sA, sC, sB, ox, oy, w, h, 0);       //      ... then erode: sB = hole_fill
    fill_rect<P, RH, 3 * A, 3, false>(sB, sC, sA, ox, oy, w, h, 0);      // :365 dilate 7 x 7
    fill_region<P, RH, 3 * A + 3, 3 * A + 3>([&](int rx, int ry) {       // :367-376
        const int s = sB[ry * P + rx];
        if (s != 0) sA[ry * P + rx] = (unsigned char)s;
    });
    __syncthreads();
    // median 5 x 5, BORDER_REPLICATE (:393), by the min / max / med3 network of median_net.hpp
    fill_region<P, RH, 3 * A + 5, 3 * A + 5>([&](int rx, int ry) {
        const int gx = ox + rx, gy = oy + ry;
        if (gx < 0 || gx >= w || gy < 0 || gy >= h) return;
        int val[25];
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const int yy = min(max(gy + i - 2, 0), h - 1) - oy;
#pragma unroll
            for (int jj = 0; jj < 5; jj++) val[i * 5 + jj] = sA[yy * P + min(max(gx + jj - 2, 0), w - 1) - ox];
        }
        const int lo = mednet::median25(val);
        sC[ry * P + rx] = (unsigned char)lo;
    });
    __syncthreads();
    // blur (:395-401), BORDER_REFLECT_101, written straight to the depth map
    fill_region<P, RH, R0, R0>([&](int rx, int ry) {
        const int gx = ox + rx, gy = oy + ry;
        if (gx >= w || gy >= h) return;
        int out;
        if (j.cam.blur == 0) {
            // taps of the radius-2 disc in row-major order (the order cv::bilateralFilter accumulates in)
            constexpr int kDi[13] = { -2, -1, -1, -1, 0, 0, 0, 0, 0, 1, 1, 1, 2 }, kDj[13] = { 0, -1, 0, 1, -2, -1, 0, 1, 2, -1, 0, 1, 0 };
            int rowo[5], colo[5];
#pragma unroll
            for (int d = 0; d < 5; d++) {
                int yy = gy + d - 2, xx = gx + d - 2;
                yy = yy < 0 ? -yy : yy; yy = yy >= h ? 2 * h - 2 - yy : yy;
                xx = xx < 0 ? -xx : xx; xx = xx >= w ? 2 * w - 2 - xx : xx;
                rowo[d] = (yy - oy) * P; colo[d] = xx - ox;
            }
            int val[13];
            const int val0 = sC[ry * P + rx];
            bool flat = true;
#pragma unroll
            for (int k = 0; k < 13; k++) { val[k] = sC[rowo[kDi[k] + 2] + colo[kDj[k] + 2]]; flat = flat && val[k] == val0; }
            out = val0;                  // a constant window: sum / wsum = val0 up to float rounding, far inside cvRound's half
            if (!flat) {
                float sum = 0.f, wsum = 0.f;
#pragma unroll
                for (int k = 0; k < 13; k++) {
                    const float wt = tab->space[k] * s_cw[abs(val[k] - val0)];
                    sum += (float)val[k] * wt;
                    wsum += wt;
                }
                out = (int)rintf(sum / wsum);
            }
        } else {
            int s = 0;
#pragma unroll
            for (int i = -2; i <= 2; i++) {
                int yy = gy + i;
                yy = yy < 0 ? -yy : yy; yy = yy >= h ? 2 * h - 2 - yy : yy;
                const int wi = i == 0 ? 6 : (i == -1 || i == 1 ? 4 : 1);
#pragma unroll
                for (int jj = -2; jj <= 2; jj++) {
                    int xx = gx + jj;
                    xx = xx < 0 ? -xx : xx; xx = xx >= w ? 2 * w - 2 - xx : xx;
                    const int wj = jj == 0 ? 6 : (jj == -1 || jj == 1 ? 4 : 1);
                    s += wi * wj * (int)sC[(yy - oy) * P + xx - ox];
                }
            }
            out = (s + 128) >> 8;
        }
        j.depth[(size_t)gy * w + gx] = (unsigned char)out;
    });
}

// the tests of the back-projection loop, Map_Builder.cc:275-310; liftProjective PinholeCamera.cc:450-510
__device__ __forceinline__ bool col_backproject(const ColourCam &c, int i, int row, int dbyte, float &x, float &y, float &z)
{
    const int dv = 100 - dbyte;
    if (dv <= 0 || dv >= 70) return false;
    const double mx_d = c.ik11 * (double)i + c.ik13, my_d = c.ik22 * (double)row + c.ik23;
    double mx_u = mx_d, my_u = my_d;
    if (c.distort) {
        double dx, dy;
        col_distortion(c, mx_d, my_d, dx, dy);
        mx_u = mx_d - dx; my_u = my_d - dy;
        for (int it = 1; it < 8; it++) {
            col_distortion(c, mx_u, my_u, dx, dy);
            mx_u = mx_d - dx; my_u = my_d - dy;
        }
    }
    x = (float)((double)dv * mx_u);      // depth_value * b.x() / b.z() with b.z() = 1
    y = (float)((double)dv * my_u);
    z = (float)dv;
    if (fabsf(x) > 20.f && (double)y > 1.8) return false;
    return true;
}

__global__ __launch_bounds__(kColT) void k_colour_count(const ColourJob *jobs)
{
    __shared__ int s_w[kColT / kWave];
    const ColourJob &j = jobs[blockIdx.y];
    const int row = blockIdx.x, w = j.cam.w;
    if (row >= j.cam.h) return;
    int cnt = 0;
    for (int i = threadIdx.x; i < w; i += kColT) {
        float x, y, z;
        cnt += col_backproject(j.cam, i, row, j.depth[(size_t)row * w + i], x, y, z) ? 1 : 0;
        j.key[(size_t)row * w + i] = 0u;
    }
    cnt = wave_sum_i(cnt);
    if (lane_id() == 0) s_w[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) j.row_cnt[row] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

__global__ __launch_bounds__(kColT) void k_colour_lift(const ColourJob *jobs)
{
    __shared__ int s_w[kColT / kWave];
    __shared__ int s_base;
    const ColourJob &j = jobs[blockIdx.y];
    const int row = blockIdx.x, w = j.cam.w, h = j.cam.h;
    if (row >= h) return;
    int part = 0;
    for (int r = threadIdx.x; r < row; r += kColT) part += j.row_cnt[r];
    part = wave_sum_i(part);
    if (lane_id() == 0) s_w[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) s_base = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    __syncthreads();
    long long base = s_base;
    const int wave = threadIdx.x >> 6, lane = lane_id();
    bool overflow = false;
    for (int i0 = 0; i0 < w; i0 += kColT) {
        const int i = i0 + threadIdx.x;
        float x = 0.f, y = 0.f, z = 0.f;
        bool ok = false;
        if (i < w) ok = col_backproject(j.cam, i, row, j.depth[(size_t)row * w + i], x, y, z);
        const unsigned long long bal = __ballot(ok);
        __syncthreads();
        if (lane == 0) s_w[wave] = __popcll(bal);
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int q = 0; q < kColT / kWave; q++) { const int c = s_w[q]; total += c; before += q < wave ? c : 0; }
        if (ok) {
            const long long o = base + before + __popcll(bal & ((1ull << lane) - 1ull));
            const unsigned char *px = j.bgr + 3 * ((size_t)row * w + i);
            PtRgb p;
            p.x = x; p.y = y; p.z = z;
            p.bgra = (unsigned int)px[0] | (unsigned int)px[1] << 8 | (unsigned int)px[2] << 16 | 0xff000000u;
            j.cam_out[o] = p;
            if (o < j.world_cap) {
                const double dx = x, dy = y, dz = z;       // Map_Builder.cc:315-322, pcl::transformPointCloud(Matrix4d)
                PtRgb q;
                q.x = (float)(j.R[0] * dx + j.R[1] * dy + j.R[2] * dz + j.T[0]);
                q.y = (float)(j.R[3] * dx + j.R[4] * dy + j.R[5] * dz + j.T[1]);
                q.z = (float)(j.R[6] * dx + j.R[7] * dy + j.R[8] * dz + j.T[2]);
                q.bgra = p.bgra;
                j.world_out[o] = q;
            } else overflow = true;
        }
        base += total;
    }
    if (overflow) j.n_out[1] = 1;
    if (row == h - 1 && threadIdx.x == 0) j.n_out[0] = (int)base;
}

}  // namespace lmono
