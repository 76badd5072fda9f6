/*
 * oracle/lo_colour.c -- TEST INFRASTRUCTURE (see lo_oracle.h).  CPU restatement of the colour projection of
 * lmono's map builder: MapBuilder::associateToMap (mono_lidar_mapping/src/map_builder/Map_Builder.cc:213-334),
 * MapBuilder::depthFill (:336-403), MapBuilder::Point3DTo2D (:405-416), the cloud transform of the node
 * (mono_lidar_mapping/src/map_build_node.cc:216-225) and camodocal's PinholeCamera
 * (camera_models/src/camera_models/PinholeCamera.cc:450-545, :646-662).
 *
 * PARITY UNPINNED for the image filters: cv::dilate / morphologyEx / medianBlur / bilateralFilter / GaussianBlur and
 * cv::getStructuringElement come from OpenCV, a system dependency of the reference (package.xml) that is neither
 * vendored nor installed here.  They are restated from OpenCV's documented definitions (8-bit, single channel):
 *   dilate / erode   max / min over the structuring element, anchor at the centre, pixels outside the image ignored
 *   medianBlur 5     median of the 5 x 5 window, BORDER_REPLICATE
 *   bilateralFilter  d = 5: taps with sqrt(i^2 + j^2) <= 2 in row-major order, float weights
 *                    exp(-r^2 / 2 sigma_s^2) * exp(-dI^2 / 2 sigma_c^2), float accumulation, cvRound, BORDER_REFLECT_101
 *   GaussianBlur 5x5 sigma 0 -> the fixed 1-4-6-4-1 / 16 kernel, exact fixed point, round half up, BORDER_REFLECT_101
 * The display-only products of associateToMap (HSV circles :244, JET heat map :255) are not restated.
 */
#include "lo_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static int has_distortion(const lo_cam *c) { return !(c->k1 == 0.0 && c->k2 == 0.0 && c->p1 == 0.0 && c->p2 == 0.0); } /* PinholeCamera.cc:278-289 */

/* PinholeCamera::distortion, PinholeCamera.cc:646-662 */
static void distortion(const lo_cam *c, double ux, double uy, double *dx, double *dy)
{
    const double mx2 = ux * ux, my2 = uy * uy, mxy = ux * uy, rho2 = mx2 + my2;
    const double rad = c->k1 * rho2 + c->k2 * rho2 * rho2;
    *dx = ux * rad + 2.0 * c->p1 * mxy + c->p2 * (rho2 + 2.0 * mx2);
    *dy = uy * rad + 2.0 * c->p2 * mxy + c->p1 * (rho2 + 2.0 * my2);
}

/* PinholeCamera::spaceToPlane, PinholeCamera.cc:520-545 */
int lo_space_to_plane(const lo_cam *c, const double P[3], double p[2])
{
    double ux = P[0] / P[2], uy = P[1] / P[2];
    if (has_distortion(c)) {
        double dx, dy;
        distortion(c, ux, uy, &dx, &dy);
        ux = ux + dx; uy = uy + dy;
    }
    p[0] = c->fx * ux + c->cx;
    p[1] = c->fy * uy + c->cy;
    return 0;
}

/* PinholeCamera::liftProjective, PinholeCamera.cc:450-510 (the recursive distortion model, n = 8) */
void lo_lift_projective(const lo_cam *c, double u, double v, double ray[3])
{
    const double inv_k11 = 1.0 / c->fx, inv_k13 = -c->cx / c->fx, inv_k22 = 1.0 / c->fy, inv_k23 = -c->cy / c->fy; /* :292-295 */
    const double mx_d = inv_k11 * u + inv_k13, my_d = inv_k22 * v + inv_k23;
    double mx_u = mx_d, my_u = my_d;
    if (has_distortion(c)) {
        double dx, dy;
        distortion(c, mx_d, my_d, &dx, &dy);
        mx_u = mx_d - dx; my_u = my_d - dy;
        for (int i = 1; i < 8; i++) {
            distortion(c, mx_u, my_u, &dx, &dy);
            mx_u = mx_d - dx; my_u = my_d - dy;
        }
    }
    ray[0] = mx_u; ray[1] = my_u; ray[2] = 1.0;
}

/* cv::getStructuringElement (MORPH_RECT / MORPH_CROSS / MORPH_ELLIPSE), square k x k, anchor at the centre */
void lo_structuring_element(int type, int k, uint8_t *mask)
{
    const int r = k / 2, c = k / 2;
    const double inv_r2 = r ? 1.0 / ((double)r * r) : 0.0;
    for (int i = 0; i < k; i++) {
        int j1 = 0, j2 = 0;
        if (type == 0 || (type == 1 && i == r)) { j1 = 0; j2 = k; }
        else if (type == 1) { j1 = c; j2 = c + 1; }
        else {
            const int dy = i - r;
            if (abs(dy) <= r) {
                const int dx = (int)lrint(c * sqrt((r * r - dy * dy) * inv_r2));
                j1 = c - dx > 0 ? c - dx : 0;
                j2 = c + dx + 1 < k ? c + dx + 1 : k;
            }
        }
        for (int j = 0; j < k; j++) mask[i * k + j] = (j >= j1 && j < j2) ? 1 : 0;
    }
}

void lo_morph(const uint8_t *src, uint8_t *dst, int w, int h, const uint8_t *mask, int k, int op)
{
    const int a = k / 2;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int v = op ? 255 : 0;
            for (int i = 0; i < k; i++) {
                const int yy = y + i - a;
                if (yy < 0 || yy >= h) continue;
                for (int j = 0; j < k; j++) {
                    const int xx = x + j - a;
                    if (!mask[i * k + j] || xx < 0 || xx >= w) continue;
                    const int s = src[yy * w + xx];
                    if (op ? s < v : s > v) v = s;
                }
            }
            dst[y * w + x] = (uint8_t)v;
        }
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static int reflect101(int v, int n) { if (v < 0) v = -v; if (v >= n) v = 2 * n - 2 - v; return v; }
static int cmp_u8(const void *a, const void *b) { return (int)*(const uint8_t *)a - (int)*(const uint8_t *)b; }

void lo_median5(const uint8_t *src, uint8_t *dst, int w, int h)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            uint8_t v[25];
            int n = 0;
            for (int i = -2; i <= 2; i++)
                for (int j = -2; j <= 2; j++) v[n++] = src[clampi(y + i, 0, h - 1) * w + clampi(x + j, 0, w - 1)];
            qsort(v, 25, 1, cmp_u8);
            dst[y * w + x] = v[12];
        }
}

void lo_bilateral5(const uint8_t *src, uint8_t *dst, int w, int h, double sigma_color, double sigma_space)
{
    const int radius = 2;
    float color_weight[256], space_weight[25];
    int ofs_i[25], ofs_j[25], maxk = 0;
    if (sigma_color <= 0) sigma_color = 1;
    if (sigma_space <= 0) sigma_space = 1;
    const double gauss_color_coeff = -0.5 / (sigma_color * sigma_color), gauss_space_coeff = -0.5 / (sigma_space * sigma_space);
    for (int i = 0; i < 256; i++) color_weight[i] = (float)exp(i * i * gauss_color_coeff);
    for (int i = -radius; i <= radius; i++)
        for (int j = -radius; j <= radius; j++) {
            const double r = sqrt((double)i * i + (double)j * j);
            if (r > radius) continue;
            space_weight[maxk] = (float)exp(r * r * gauss_space_coeff);
            ofs_i[maxk] = i; ofs_j[maxk++] = j;
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float sum = 0, wsum = 0;
            const int val0 = src[y * w + x];
            for (int k = 0; k < maxk; k++) {
                const int val = src[reflect101(y + ofs_i[k], h) * w + reflect101(x + ofs_j[k], w)];
                const float wt = space_weight[k] * color_weight[abs(val - val0)];
                sum += val * wt;
                wsum += wt;
            }
            dst[y * w + x] = (uint8_t)lrintf(sum / wsum);
        }
}

void lo_gauss5(const uint8_t *src, uint8_t *dst, int w, int h)
{
    static const int kw[5] = { 1, 4, 6, 4, 1 };
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int i = -2; i <= 2; i++)
                for (int j = -2; j <= 2; j++) s += kw[i + 2] * kw[j + 2] * src[reflect101(y + i, h) * w + reflect101(x + j, w)];
            dst[y * w + x] = (uint8_t)((s + 128) >> 8);
        }
}

/* MapBuilder::depthFill, Map_Builder.cc:336-403 */
void lo_depth_fill(const lo_cam *c, uint8_t *depth)
{
    const int w = c->width, h = c->height, k = c->kernel_size;
    const size_t n = (size_t)w * h;
    uint8_t *a = malloc(n), *b = malloc(n), *mask = malloc((size_t)k * k), *rect = malloc((size_t)k * k), rect7[49];
    lo_structuring_element(c->kernel_type, k, mask);          /* :343-355 */
    lo_structuring_element(0, k, rect);
    memset(rect7, 1, sizeof rect7);
    lo_morph(depth, a, w, h, mask, k, 0);                     /* :359-360 dilate */
    lo_morph(a, b, w, h, rect, k, 0);                         /* :364 MORPH_CLOSE = dilate ... */
    lo_morph(b, a, w, h, rect, k, 1);                         /*       ... then erode: a = hole_fill */
    lo_morph(a, b, w, h, rect7, 7, 0);                        /* :365 */
    for (size_t i = 0; i < n; i++) if (a[i] == 0) a[i] = b[i];  /* :367-376 (uchar < 0.1 <=> == 0) */
    lo_median5(a, b, w, h);                                   /* :393 */
    if (c->blur_type == 0) lo_bilateral5(b, depth, w, h, 1.5, 2.0);   /* :396-398 */
    else lo_gauss5(b, depth, w, h);                           /* :399-401 */
    free(a); free(b); free(mask); free(rect);
}

/* map_build_node.cc:216-225 (pcl::transformPointCloud with a Matrix4d: double arithmetic, float store) followed by the
 * projection loop of associateToMap, Map_Builder.cc:225-247 */
void lo_depth_splat(const lo_cam *c, const float *xyzi, int n, const double M[16], uint8_t *depth)
{
    const int w = c->width, h = c->height;
    for (int i = 0; i < n; i++) {
        const double x = xyzi[4 * i], y = xyzi[4 * i + 1], z = xyzi[4 * i + 2];
        const float px = (float)(M[0] * x + M[1] * y + M[2] * z + M[3]);
        const float py = (float)(M[4] * x + M[5] * y + M[6] * z + M[7]);
        const float pz = (float)(M[8] * x + M[9] * y + M[10] * z + M[11]);
        if (pz < 0) continue;                                 /* :227 */
        const double P[3] = { px, py, pz };
        double p[2];
        lo_space_to_plane(c, P, p);
        const float u = (float)p[0], v = (float)p[1];         /* cv::Point2f, :415 */
        if (u > 0 && u < (float)w && v > 0 && v < (float)h) { /* :234 */
            const double d = pz;
            /* :238 `at<uchar>(xy.y, xy.x) = 100 - depth`: float -> int index truncation; double -> uchar through int
             * (what x86-64 compilers emit; formally undefined above 100 m, where it wraps modulo 256) */
            depth[(int)v * w + (int)u] = (uint8_t)(int32_t)(100.0 - d);
        }
    }
}

/* Map_Builder.cc:275-312 */
int lo_backproject(const lo_cam *c, const uint8_t *depth, const uint8_t *bgr, lo_pt_rgb *out)
{
    const int w = c->width, h = c->height;
    int n = 0;
    for (int j = 0; j < h; j++)
        for (int i = 0; i < w; i++) {
            const int dv = 100 - depth[j * w + i];
            if (dv <= 0 || dv >= 70) continue;
            double b[3];
            lo_lift_projective(c, (double)i, (double)j, b);
            lo_pt_rgb p;
            p.x = (float)(dv * b[0] / b[2]);
            p.y = (float)(dv * b[1] / b[2]);
            p.z = (float)dv;
            const uint8_t *px = bgr + 3 * ((size_t)j * w + i);
            p.bgra = (uint32_t)px[0] | (uint32_t)px[1] << 8 | (uint32_t)px[2] << 16 | 0xff000000u;
            if (fabsf(p.x) > 20 && (double)p.y > 1.8) continue;    /* :305 */
            out[n++] = p;
        }
    return n;
}

/* Map_Builder.cc:315-322: Eigen::Quaterniond::toRotationMatrix + pcl::transformPointCloud(Matrix4d) */
void lo_transform_rgb(const lo_pt_rgb *in, int n, const double q[4], const double t[3], lo_pt_rgb *out)
{
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    const double R[9] = { 1.0 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1.0 - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1.0 - (txx + tyy) };
    for (int i = 0; i < n; i++) {
        const double px = in[i].x, py = in[i].y, pz = in[i].z;
        out[i].x = (float)(R[0] * px + R[1] * py + R[2] * pz + t[0]);
        out[i].y = (float)(R[3] * px + R[4] * py + R[5] * pz + t[1]);
        out[i].z = (float)(R[6] * px + R[7] * py + R[8] * pz + t[2]);
        out[i].bgra = in[i].bgra;
    }
}

int lo_associate_to_map(const lo_cam *c, const float *xyzi, int n, const double M[16], const uint8_t *bgr,
                        const double q[4], const double t[3], uint8_t *depth_out, lo_pt_rgb *cam_out, lo_pt_rgb *world_out)
{
    const size_t np = (size_t)c->width * c->height;
    uint8_t *depth = calloc(np, 1);
    lo_depth_splat(c, xyzi, n, M, depth);
    lo_depth_fill(c, depth);
    const int m = lo_backproject(c, depth, bgr, cam_out);
    lo_transform_rgb(cam_out, m, q, t, world_out);
    if (depth_out) memcpy(depth_out, depth, np);
    free(depth);
    return m;
}
