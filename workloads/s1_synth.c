/*
 * workloads/s1_synth.c -- INPUT PLUMBING (not the hot path, not the oracle).  Synthetic HDL-64 scan generator "S1"
 * (SURVEY.md 8d): ground plane + axis-aligned boxes + vertical cylinders, ring-major point order
 * like KITTI velodyne .bin files (float32 x y z reflectance), one clockwise sweep per ring starting
 * at the rear of the vehicle, Gaussian range noise, random drop-outs, no motion distortion
 * (KITTI scans are already de-skewed; A-LOAM runs them with DISTORTION 0).
 * This is input plumbing shared by tests and bench.py; it is not part of the hot path.
 */
#include "s1_synth.h"
#include <math.h>
#include <stdlib.h>

static inline uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
static inline double u01(uint64_t h) { return ((double)(h >> 11) + 0.5) * (1.0 / 9007199254740992.0); }

int lo_synth_scan(const lo_world *w, const double pose[4], uint64_t scan_id, float *out)
{
    const double px = pose[0], py = pose[1], pz = pose[2], yaw = pose[3];
    const double cyaw = cos(yaw), syaw = sin(yaw);
    int n = 0;
    /* azimuth-binned culling: objects whose bounding circle can be hit by rays of each world-azimuth bin */
    enum { NB = 720, MAXOBJ = 512 };
    const int nobj = w->n_boxes + w->n_cyls;
    unsigned short *bins = NULL; short *bcount = NULL;
    const int use_bins = nobj <= MAXOBJ;
    if (use_bins) {
        bins = (unsigned short *)malloc(sizeof(unsigned short) * (size_t)NB * MAXOBJ);
        bcount = (short *)calloc(NB, sizeof(short));
        for (int o = 0; o < nobj; o++) {
            double cx, cy, R;
            if (o < w->n_boxes) {
                const double *B = w->boxes + 6 * o;
                cx = 0.5 * (B[0] + B[3]) - px; cy = 0.5 * (B[1] + B[4]) - py;
                R = sqrt(0.25 * (B[3] - B[0]) * (B[3] - B[0]) + 0.25 * (B[4] - B[1]) * (B[4] - B[1]));
            } else {
                const double *Cc = w->cyls + 4 * (o - w->n_boxes);
                cx = Cc[0] - px; cy = Cc[1] - py; R = Cc[2];
            }
            double d = sqrt(cx * cx + cy * cy);
            int lo = 0, hi = NB - 1;
            if (d > R * 1.0001) {
                double th = atan2(cy, cx), al = asin(R / d) + 2.0 * M_PI / NB;
                lo = (int)floor((th - al) / (2.0 * M_PI) * NB); hi = (int)floor((th + al) / (2.0 * M_PI) * NB);
            }
            for (int bb = lo; bb <= hi; bb++) {
                int bi = ((bb % NB) + NB) % NB;
                bins[bi * MAXOBJ + bcount[bi]++] = (unsigned short)o;
            }
        }
    }
    /* moving cylinders at this scan's time (clutter; none in the tidy worlds) */
    enum { MAXMOV = 16 };
    double mov[MAXMOV][4];
    const int n_mov = w->n_moving < MAXMOV ? w->n_moving : MAXMOV;
    for (int m = 0; m < n_mov; m++) {
        const double *M = w->moving + 6 * m;
        const double t = (double)scan_id * 0.1;
        double x = M[0] + M[2] * t, y = M[1] + M[3] * t;
        x = fmod(fmod(x + 95.0, 190.0) + 190.0, 190.0) - 95.0; y = fmod(fmod(y + 95.0, 190.0) + 190.0, 190.0) - 95.0;
        mov[m][0] = x; mov[m][1] = y; mov[m][2] = M[4]; mov[m][3] = M[5];
    }
    double *caz = (double *)malloc(sizeof(double) * 2 * (size_t)w->n_az);
    int *kbin = (int *)malloc(sizeof(int) * (size_t)w->n_az);
    for (int k = 0; k < w->n_az; k++) {
        double az = M_PI - ((double)k + 0.5) * (2.0 * M_PI / (double)w->n_az);
        caz[2 * k] = cos(az); caz[2 * k + 1] = sin(az);
        int bi = (int)floor((az + yaw) / (2.0 * M_PI) * NB);
        kbin[k] = ((bi % NB) + NB) % NB;
    }
    for (int r = 0; r < w->n_rings; r++) {
        const double ce = cos(w->elev_rad[r]), se = sin(w->elev_rad[r]);
        /* a dropped azimuth sector of this ring (clutter) */
        int drop0 = -1, drop1 = -1;
        if (w->sector_drop > 0.0) {
            const uint64_t hs = splitmix64(w->seed ^ 0xD509ull ^ splitmix64(scan_id * 0x9E3779B1ull + (uint64_t)r));
            if (u01(hs) < w->sector_drop) {
                const uint64_t hs2 = splitmix64(hs), hs3 = splitmix64(hs2);
                drop0 = (int)(u01(hs2) * w->n_az);
                drop1 = drop0 + (int)((0.05 + 0.10 * u01(hs3)) * w->n_az);
            }
        }
        for (int k = 0; k < w->n_az; k++) {
            if (drop0 >= 0 && ((k >= drop0 && k < drop1) || (k + w->n_az >= drop0 && k + w->n_az < drop1))) continue;
            uint64_t h0 = splitmix64(w->seed ^ splitmix64(scan_id * 0x100000001B3ull + (uint64_t)r * 65536ull + (uint64_t)k));
            uint64_t h1 = splitmix64(h0), h2 = splitmix64(h1);
            if (u01(h0) < w->dropout) continue;
            /* clockwise sweep starting just past the rear (-x) direction */
            double dsx = ce * caz[2 * k], dsy = ce * caz[2 * k + 1], dsz = se;
            double dx = cyaw * dsx - syaw * dsy, dy = syaw * dsx + cyaw * dsy, dz = dsz;
            double best = w->max_range;
            if (dz < 0.0) {
                double t = (w->ground_z - pz) / dz;
                if (t > 0.0 && t < best) best = t;
            }
            const double d2n = dx * dx + dy * dy;
            int bi = 0, nlist = nobj;
            if (use_bins) {
                bi = kbin[k];
                nlist = bcount[bi];
            }
            for (int li = 0; li < nlist; li++) {
                const int b = use_bins ? bins[bi * MAXOBJ + li] : li;
                if (b >= w->n_boxes) continue;
                const double *B = w->boxes + 6 * b;
                /* 2-D bounding-circle cull */
                double cx = 0.5 * (B[0] + B[3]) - px, cy = 0.5 * (B[1] + B[4]) - py;
                double hx = 0.5 * (B[3] - B[0]), hy = 0.5 * (B[4] - B[1]);
                double cr = cx * dy - cy * dx;
                if (cr * cr > (hx * hx + hy * hy) * d2n) continue;
                double t0 = 0.0, t1 = best;
                const double o[3] = { px, py, pz }, d[3] = { dx, dy, dz };
                int hit = 1;
                for (int a = 0; a < 3 && hit; a++) {
                    if (fabs(d[a]) < 1e-12) { if (o[a] < B[a] || o[a] > B[a + 3]) hit = 0; continue; }
                    double ta = (B[a] - o[a]) / d[a], tb = (B[a + 3] - o[a]) / d[a];
                    if (ta > tb) { double tt = ta; ta = tb; tb = tt; }
                    if (ta > t0) t0 = ta;
                    if (tb < t1) t1 = tb;
                    if (t0 > t1) hit = 0;
                }
                if (hit && t0 > 0.0 && t0 < best) best = t0;
            }
            for (int li = 0; li < nlist; li++) {
                const int oc = use_bins ? bins[bi * MAXOBJ + li] : li;
                if (oc < w->n_boxes) continue;
                const int c = oc - w->n_boxes;
                const double *C = w->cyls + 4 * c;
                double cx = C[0] - px, cy = C[1] - py, R = C[2];
                double cr = cx * dy - cy * dx;
                if (cr * cr > R * R * d2n) continue;
                double bq = cx * dx + cy * dy;
                double disc = bq * bq - d2n * (cx * cx + cy * cy - R * R);
                if (disc < 0.0) continue;
                double t = (bq - sqrt(disc)) / d2n;
                if (t <= 0.0 || t >= best) continue;
                double hz = pz + t * dz;
                if (hz > C[3] || hz < w->ground_z) continue;
                best = t;
            }
            for (int m = 0; m < n_mov; m++) {
                double cx = mov[m][0] - px, cy = mov[m][1] - py, R = mov[m][2];
                double cr = cx * dy - cy * dx;
                if (cr * cr > R * R * d2n) continue;
                double bq = cx * dx + cy * dy;
                double disc = bq * bq - d2n * (cx * cx + cy * cy - R * R);
                if (disc < 0.0) continue;
                double t = (bq - sqrt(disc)) / d2n;
                if (t <= 0.0 || t >= best) continue;
                double hz = pz + t * dz;
                if (hz > mov[m][3] || hz < w->ground_z) continue;
                best = t;
            }
            if (!(best < w->max_range)) continue;
            /* a stray return: a random range in front of the surface (clutter) */
            if (w->stray_frac > 0.0) {
                const uint64_t h3 = splitmix64(h2 ^ 0x5712A7ull), h4 = splitmix64(h3);
                if (u01(h3) < w->stray_frac && best > 2.0) best = 2.0 + (best - 2.0) * u01(h4);
            }
            /* Box-Muller range noise */
            double g = sqrt(-2.0 * log(u01(h1))) * cos(2.0 * M_PI * u01(h2));
            double rng = best + w->range_sigma * g;
            if (rng < 0.5) continue;
            out[4 * n + 0] = (float)(rng * dsx);
            out[4 * n + 1] = (float)(rng * dsy);
            out[4 * n + 2] = (float)(rng * dsz);
            out[4 * n + 3] = (float)u01(splitmix64(h2));
            n++;
        }
    }
    free(bins); free(bcount); free(caz); free(kbin);
    return n;
}

/* Batch version: scan s is written at out + s * slot_floats; counts[s] = points written. */
void lo_synth_scans(const lo_world *w, const double *poses, uint64_t scan_id0, int n_scans,
                    float *out, int64_t slot_floats, int32_t *counts)
{
#pragma omp parallel for schedule(dynamic, 1)
    for (int s = 0; s < n_scans; s++)
        counts[s] = lo_synth_scan(w, poses + 4 * s, scan_id0 + (uint64_t)s, out + (int64_t)s * slot_floats);
}
