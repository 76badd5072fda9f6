/*
 * oracle/lo_scanreg.c -- TEST INFRASTRUCTURE (CPU oracle).  PARITY UNPINNED.
 *
 * Restates A-LOAM scanRegistration.cpp::laserCloudHandler (source absent from
 * /root/reference; spec = SURVEY.md Appendix A.1, reference call sites
 * /root/reference/README.md:54-60, mono_lidar_mapping/config/kitti_config_00.yaml:5-9).
 *
 * Numeric model (x86-64, no FMA contraction): PointXYZI fields are float; every
 * float expression of the upstream code is evaluated in float in source order;
 * libm atan / atan2 results are modelled as "double result rounded to float"
 * using lo_det_math.h; std::sort ties are broken by ascending point index.
 * pcl::VoxelGrid (leaf 0.2, downsample_all_data=true) is restated in
 * voxel_downsample() below.
 */
#include "lo_oracle.h"
#include "lo_det_math.h"
#include <stdlib.h>
#include <string.h>

static int ring_id(float angle, int n_scans, int *discard)
{
    int id = 0;
    *discard = 0;
    if (n_scans == 16) {
        id = (int)((double)((angle + 15.0f) / 2.0f) + 0.5);
        if (id > n_scans - 1 || id < 0) *discard = 1;
    } else if (n_scans == 32) {
        id = (int)(((double)angle + 92.0 / 3.0) * 3.0 / 4.0);
        if (id > n_scans - 1 || id < 0) *discard = 1;
    } else { /* 64 */
        if ((double)angle >= -8.83)
            id = (int)((double)(2.0f - angle) * 3.0 + 0.5);
        else
            id = n_scans / 2 + (int)((-8.83 - (double)angle) * 2.0 + 0.5);
        if ((double)angle > 2.0 || (double)angle < -24.33 || id > 50 || id < 0) *discard = 1;
    }
    return id;
}

typedef struct { float c; int idx; } sort_key;
static int cmp_key(const void *a, const void *b)
{
    const sort_key *x = (const sort_key *)a, *y = (const sort_key *)b;
    if (x->c < y->c) return -1;
    if (x->c > y->c) return 1;
    return (x->idx > y->idx) - (x->idx < y->idx);
}

static float gap2(const lo_pt *a, const lo_pt *b)
{
    float dx = a->x - b->x, dy = a->y - b->y, dz = a->z - b->z;
    return dx * dx + dy * dy + dz * dz;
}

static void mark_neighbours(const lo_pt *cloud, int *picked, int ind)
{
    for (int l = 1; l <= 5; l++) {
        if ((double)gap2(&cloud[ind + l], &cloud[ind + l - 1]) > 0.05) break;
        picked[ind + l] = 1;
    }
    for (int l = -1; l >= -5; l--) {
        if ((double)gap2(&cloud[ind + l], &cloud[ind + l + 1]) > 0.05) break;
        picked[ind + l] = 1;
    }
}

/* pcl::VoxelGrid<PointXYZI>::applyFilter, leaf = 0.2 (inverse leaf = 5.0f), all fields averaged.
 * Returns number of output points appended at out. */
typedef struct { unsigned int cell; int idx; } vox_key;
static int cmp_vox(const void *a, const void *b)
{
    const vox_key *x = (const vox_key *)a, *y = (const vox_key *)b;
    if (x->cell != y->cell) return x->cell < y->cell ? -1 : 1;
    return (x->idx > y->idx) - (x->idx < y->idx);
}
/* pcl::VoxelGrid with a cubic leaf (inverse_leaf_size_ = 1.0f / leaf as PCL computes it), all fields averaged, output in
 * ascending cell index, the points of a cell summed in index order */
int lo_voxel_filter(const lo_pt *in, int n, float inv_leaf, lo_pt *out)
{
    if (n == 0) return 0;
    float mn[3] = { in[0].x, in[0].y, in[0].z }, mx[3] = { in[0].x, in[0].y, in[0].z };
    for (int i = 1; i < n; i++) {
        const float p[3] = { in[i].x, in[i].y, in[i].z };
        for (int k = 0; k < 3; k++) {
            if (p[k] < mn[k]) mn[k] = p[k];
            if (p[k] > mx[k]) mx[k] = p[k];
        }
    }
    int min_b[3], max_b[3], div_b[3];
    for (int k = 0; k < 3; k++) {
        min_b[k] = (int)floorf(mn[k] * inv_leaf);
        max_b[k] = (int)floorf(mx[k] * inv_leaf);
        div_b[k] = max_b[k] - min_b[k] + 1;
    }
    const int mul1 = div_b[0], mul2 = div_b[0] * div_b[1];
    vox_key *keys = (vox_key *)malloc(sizeof(vox_key) * (size_t)n);
    for (int i = 0; i < n; i++) {
        int i0 = (int)(floorf(in[i].x * inv_leaf) - (float)min_b[0]);
        int i1 = (int)(floorf(in[i].y * inv_leaf) - (float)min_b[1]);
        int i2 = (int)(floorf(in[i].z * inv_leaf) - (float)min_b[2]);
        keys[i].cell = (unsigned int)(i0 + i1 * mul1 + i2 * mul2);
        keys[i].idx = i;
    }
    qsort(keys, (size_t)n, sizeof(vox_key), cmp_vox);
    int n_out = 0;
    for (int s = 0; s < n;) {
        int e = s + 1;
        while (e < n && keys[e].cell == keys[s].cell) e++;
        float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
        for (int k = s; k < e; k++) {
            const lo_pt *p = &in[keys[k].idx];
            sx += p->x; sy += p->y; sz += p->z; si += p->i;
        }
        float cnt = (float)(e - s);
        out[n_out].x = sx / cnt; out[n_out].y = sy / cnt; out[n_out].z = sz / cnt; out[n_out].i = si / cnt;
        n_out++;
        s = e;
    }
    free(keys);
    return n_out;
}
static int voxel_downsample(const lo_pt *in, int n, lo_pt *out) { return lo_voxel_filter(in, n, 5.0f /* 1.0f / 0.2f rounds to 5.0f */, out); }

int lo_scanreg(const float *xyzi, int n, int n_scans, float min_range,
               lo_pt *cloud, float *curvature, int32_t *label,
               lo_pt *sharp, lo_pt *less_sharp, lo_pt *flat, lo_pt *less_flat,
               lo_scanreg_info *info)
{
    memset(info, 0, sizeof(*info));
    if (n_scans != 16 && n_scans != 32 && n_scans != 64) return -1;
    /* removeNaNFromPointCloud + removeClosedPointCloud */
    float *in = (float *)malloc(sizeof(float) * 3 * (size_t)(n > 0 ? n : 1));
    int m = 0;
    for (int i = 0; i < n; i++) {
        float x = xyzi[4 * i], y = xyzi[4 * i + 1], z = xyzi[4 * i + 2];
        if (!(isfinite(x) && isfinite(y) && isfinite(z))) continue;
        if (x * x + y * y + z * z < min_range * min_range) continue;
        in[3 * m] = x; in[3 * m + 1] = y; in[3 * m + 2] = z;
        m++;
    }
    if (m == 0) { free(in); return 0; }

    float startOri = (float)(-lo_atan2((double)in[1], (double)in[0]));
    float endOri = (float)((double)(float)(-lo_atan2((double)in[3 * (m - 1) + 1], (double)in[3 * (m - 1)])) + 2.0 * LO_PI);
    if ((double)(endOri - startOri) > 3.0 * LO_PI) endOri = (float)((double)endOri - 2.0 * LO_PI);
    else if ((double)(endOri - startOri) < LO_PI) endOri = (float)((double)endOri + 2.0 * LO_PI);

    /* pass 1: ring id + intensity per kept point, in input order */
    int *ring = (int *)malloc(sizeof(int) * (size_t)m);
    float *inten = (float *)malloc(sizeof(float) * (size_t)m);
    int counts[LO_MAX_RINGS] = { 0 };
    int half_passed = 0;
    for (int i = 0; i < m; i++) {
        float x = in[3 * i], y = in[3 * i + 1], z = in[3 * i + 2];
        float angle = (float)(lo_atan((double)z / sqrt((double)(x * x + y * y))) * 180.0 / LO_PI);
        int discard;
        int id = ring_id(angle, n_scans, &discard);
        if (discard) { ring[i] = -1; continue; }
        float ori = (float)(-lo_atan2((double)y, (double)x));
        if (!half_passed) {
            if ((double)ori < (double)startOri - LO_PI / 2.0) ori = (float)((double)ori + 2.0 * LO_PI);
            else if ((double)ori > (double)startOri + LO_PI * 3.0 / 2.0) ori = (float)((double)ori - 2.0 * LO_PI);
            if ((double)(ori - startOri) > LO_PI) half_passed = 1;
        } else {
            ori = (float)((double)ori + 2.0 * LO_PI);
            if ((double)ori < (double)endOri - LO_PI * 3.0 / 2.0) ori = (float)((double)ori + 2.0 * LO_PI);
            else if ((double)ori > (double)endOri + LO_PI / 2.0) ori = (float)((double)ori - 2.0 * LO_PI);
        }
        float relTime = (ori - startOri) / (endOri - startOri);
        ring[i] = id;
        inten[i] = (float)((double)id + 0.1 * (double)relTime);
        counts[id]++;
    }
    /* concatenate ring buckets in ring order (stable) */
    int pos[LO_MAX_RINGS];
    int total = 0;
    for (int r = 0; r < LO_MAX_RINGS; r++) {
        info->ring_begin[r] = total;
        pos[r] = total;
        if (r < n_scans) {
            info->scan_start[r] = total + 5;
            total += counts[r];
            info->scan_end[r] = total - 6;
        }
    }
    info->ring_begin[LO_MAX_RINGS] = total;
    for (int r = n_scans; r < LO_MAX_RINGS; r++) { info->scan_start[r] = total + 5; info->scan_end[r] = total - 6; }
    info->n_cloud = total;
    for (int i = 0; i < m; i++) {
        if (ring[i] < 0) continue;
        lo_pt *p = &cloud[pos[ring[i]]++];
        p->x = in[3 * i]; p->y = in[3 * i + 1]; p->z = in[3 * i + 2]; p->i = inten[i];
    }
    free(ring); free(inten); free(in);

    /* curvature */
    int *picked = (int *)calloc((size_t)(total > 0 ? total : 1), sizeof(int));
    for (int i = 0; i < total; i++) { curvature[i] = 0.f; label[i] = 0; }
    for (int i = 5; i < total - 5; i++) {
        const lo_pt *c = cloud;
        float dx = c[i - 5].x + c[i - 4].x + c[i - 3].x + c[i - 2].x + c[i - 1].x - 10 * c[i].x + c[i + 1].x + c[i + 2].x + c[i + 3].x + c[i + 4].x + c[i + 5].x;
        float dy = c[i - 5].y + c[i - 4].y + c[i - 3].y + c[i - 2].y + c[i - 1].y - 10 * c[i].y + c[i + 1].y + c[i + 2].y + c[i + 3].y + c[i + 4].y + c[i + 5].y;
        float dz = c[i - 5].z + c[i - 4].z + c[i - 3].z + c[i - 2].z + c[i - 1].z - 10 * c[i].z + c[i + 1].z + c[i + 2].z + c[i + 3].z + c[i + 4].z + c[i + 5].z;
        curvature[i] = dx * dx + dy * dy + dz * dz;
    }

    sort_key *keys = (sort_key *)malloc(sizeof(sort_key) * (size_t)(total > 0 ? total : 1));
    lo_pt *cand = (lo_pt *)malloc(sizeof(lo_pt) * (size_t)(total > 0 ? total : 1));
    int ns = 0, nls = 0, nf = 0, nlf = 0;
    for (int r = 0; r < n_scans; r++) {
        int S = info->scan_start[r], E = info->scan_end[r];
        if (E - S < 6) continue;
        int n_cand = 0;
        for (int j = 0; j < 6; j++) {
            int sp = S + (E - S) * j / 6;
            int ep = S + (E - S) * (j + 1) / 6 - 1;
            int len = ep - sp + 1;
            for (int k = 0; k < len; k++) { keys[k].c = curvature[sp + k]; keys[k].idx = sp + k; }
            qsort(keys, (size_t)len, sizeof(sort_key), cmp_key);

            int largest = 0;
            for (int k = len - 1; k >= 0; k--) {
                int ind = keys[k].idx;
                if (picked[ind] == 0 && (double)curvature[ind] > 0.1) {
                    largest++;
                    if (largest <= 2) {
                        label[ind] = 2;
                        sharp[ns++] = cloud[ind];
                        less_sharp[nls++] = cloud[ind];
                    } else if (largest <= 20) {
                        label[ind] = 1;
                        less_sharp[nls++] = cloud[ind];
                    } else break;
                    picked[ind] = 1;
                    mark_neighbours(cloud, picked, ind);
                }
            }
            int smallest = 0;
            for (int k = 0; k < len; k++) {
                int ind = keys[k].idx;
                if (picked[ind] == 0 && (double)curvature[ind] < 0.1) {
                    label[ind] = -1;
                    flat[nf++] = cloud[ind];
                    smallest++;
                    if (smallest >= 4) break;
                    picked[ind] = 1;
                    mark_neighbours(cloud, picked, ind);
                }
            }
            for (int k = sp; k <= ep; k++)
                if (label[k] <= 0) cand[n_cand++] = cloud[k];
        }
        nlf += voxel_downsample(cand, n_cand, less_flat + nlf);
    }
    free(keys); free(cand); free(picked);
    info->n_sharp = ns; info->n_less_sharp = nls; info->n_flat = nf; info->n_less_flat = nlf;
    return 0;
}
