/*
 * oracle/lo_kdtree.c -- TEST INFRASTRUCTURE (CPU oracle).  PARITY UNPINNED.
 *
 * Exact 1-nearest-neighbour search standing in for pcl::KdTreeFLANN<PointXYZI>::
 * nearestKSearch(p, 1, ...) as used by A-LOAM laserOdometry (source absent; SURVEY.md
 * Appendix A.2).  FLANN's KDTreeSingleIndexAdaptor with eps = 0 is an exact search, so any
 * exact method returns the same neighbour; the distance is FLANN L2_Simple<float>:
 * float ((dx*dx + dy*dy) + dz*dz).  Ties are broken towards the lowest index.
 */
#include "lo_oracle.h"
#include <stdlib.h>
#include <float.h>

#define LEAF 12

typedef struct {
    int left, right;   /* children (-1 for leaf) */
    int lo, hi;        /* index range into perm for leaves */
    int dim;
    float split_lo, split_hi; /* max of left side / min of right side along dim */
} kd_node;

struct lo_kdtree {
    const lo_pt *pts;
    int n;
    int *perm;
    kd_node *nodes;
    int n_nodes;
};

static inline float coord(const lo_pt *p, int d) { return d == 0 ? p->x : (d == 1 ? p->y : p->z); }

static void select_nth(const lo_pt *pts, int *perm, int lo, int hi, int nth, int d)
{
    /* quickselect on perm[lo..hi) by (coord, index) */
    while (hi - lo > 1) {
        int pi = perm[lo + (hi - lo) / 2];
        float pv = coord(&pts[pi], d);
        int i = lo, j = hi - 1;
        while (i <= j) {
            while (coord(&pts[perm[i]], d) < pv || (coord(&pts[perm[i]], d) == pv && perm[i] < pi)) i++;
            while (coord(&pts[perm[j]], d) > pv || (coord(&pts[perm[j]], d) == pv && perm[j] > pi)) j--;
            if (i <= j) { int t = perm[i]; perm[i] = perm[j]; perm[j] = t; i++; j--; }
        }
        if (nth <= j) hi = j + 1;
        else if (nth >= i) lo = i;
        else return;
    }
}

static int build(lo_kdtree *t, int lo, int hi)
{
    int id = t->n_nodes++;
    kd_node *nd = &t->nodes[id];
    nd->lo = lo; nd->hi = hi; nd->left = nd->right = -1; nd->dim = 0;
    if (hi - lo <= LEAF) return id;
    float mn[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, mx[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (int i = lo; i < hi; i++)
        for (int d = 0; d < 3; d++) {
            float v = coord(&t->pts[t->perm[i]], d);
            if (v < mn[d]) mn[d] = v;
            if (v > mx[d]) mx[d] = v;
        }
    int d = 0;
    if (mx[1] - mn[1] > mx[d] - mn[d]) d = 1;
    if (mx[2] - mn[2] > mx[d] - mn[d]) d = 2;
    int mid = lo + (hi - lo) / 2;
    select_nth(t->pts, t->perm, lo, hi, mid, d);
    float slo = -FLT_MAX, shi = FLT_MAX;
    for (int i = lo; i < mid; i++) { float v = coord(&t->pts[t->perm[i]], d); if (v > slo) slo = v; }
    for (int i = mid; i < hi; i++) { float v = coord(&t->pts[t->perm[i]], d); if (v < shi) shi = v; }
    int l = build(t, lo, mid);
    int r = build(t, mid, hi);
    nd = &t->nodes[id];
    nd->dim = d; nd->split_lo = slo; nd->split_hi = shi; nd->left = l; nd->right = r;
    return id;
}

lo_kdtree *lo_kdtree_build(const lo_pt *pts, int n)
{
    lo_kdtree *t = (lo_kdtree *)calloc(1, sizeof(*t));
    t->pts = pts; t->n = n;
    t->perm = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; i++) t->perm[i] = i;
    t->nodes = (kd_node *)malloc(sizeof(kd_node) * (size_t)(2 * (n / 1 + 1) + 2));
    t->n_nodes = 0;
    if (n > 0) build(t, 0, n);
    return t;
}

void lo_kdtree_free(lo_kdtree *t)
{
    if (!t) return;
    free(t->perm); free(t->nodes); free(t);
}

static inline float dist2(const lo_pt *p, float qx, float qy, float qz)
{
    float dx = p->x - qx, dy = p->y - qy, dz = p->z - qz;
    return dx * dx + dy * dy + dz * dz;
}

static void search(const lo_kdtree *t, int id, float qx, float qy, float qz, float *best, int *best_i)
{
    const kd_node *nd = &t->nodes[id];
    if (nd->left < 0) {
        for (int i = nd->lo; i < nd->hi; i++) {
            int pi = t->perm[i];
            float d = dist2(&t->pts[pi], qx, qy, qz);
            if (d < *best || (d == *best && pi < *best_i)) { *best = d; *best_i = pi; }
        }
        return;
    }
    float qv = nd->dim == 0 ? qx : (nd->dim == 1 ? qy : qz);
    /* lower bounds on the distance to each side (conservative: <= so that ties are visited) */
    float dl = qv - nd->split_lo; /* >0 when q is right of the whole left side  */
    float dr = nd->split_hi - qv; /* >0 when q is left of the whole right side  */
    int first = dl < dr ? nd->left : nd->right;
    int second = first == nd->left ? nd->right : nd->left;
    search(t, first, qx, qy, qz, best, best_i);
    float gap = second == nd->left ? dl : dr;
    if (gap <= 0.f || gap * gap <= *best) search(t, second, qx, qy, qz, best, best_i);
}

int lo_kdtree_nn(const lo_kdtree *t, float qx, float qy, float qz, float *d2)
{
    if (t->n == 0) { *d2 = FLT_MAX; return -1; }
    float best = FLT_MAX; int bi = -1;
    search(t, 0, qx, qy, qz, &best, &bi);
    *d2 = best;
    return bi;
}

/* ---- k nearest (pcl::KdTreeFLANN::nearestKSearch(k) as used by laserMapping with k = 5): exact, ascending
 * (distance, index); a bounded insertion list stands in for FLANN's result set ---- */
typedef struct { int k, n; int *idx; float *d2; } knn_set;
static inline float knn_worst(const knn_set *s) { return s->n < s->k ? FLT_MAX : s->d2[s->n - 1]; }
static void knn_push(knn_set *s, float d, int pi)
{
    if (s->n == s->k) {
        const float wd = s->d2[s->n - 1]; const int wi = s->idx[s->n - 1];
        if (!(d < wd || (d == wd && pi < wi))) return;
        s->n--;
    }
    int j = s->n++;
    while (j > 0 && (s->d2[j - 1] > d || (s->d2[j - 1] == d && s->idx[j - 1] > pi))) { s->d2[j] = s->d2[j - 1]; s->idx[j] = s->idx[j - 1]; j--; }
    s->d2[j] = d; s->idx[j] = pi;
}
static void search_k(const lo_kdtree *t, int id, float qx, float qy, float qz, knn_set *s)
{
    const kd_node *nd = &t->nodes[id];
    if (nd->left < 0) {
        for (int i = nd->lo; i < nd->hi; i++) { const int pi = t->perm[i]; knn_push(s, dist2(&t->pts[pi], qx, qy, qz), pi); }
        return;
    }
    const float qv = nd->dim == 0 ? qx : (nd->dim == 1 ? qy : qz);
    const float dl = qv - nd->split_lo, dr = nd->split_hi - qv;
    const int first = dl < dr ? nd->left : nd->right;
    const int second = first == nd->left ? nd->right : nd->left;
    search_k(t, first, qx, qy, qz, s);
    const float gap = second == nd->left ? dl : dr;
    if (gap <= 0.f || gap * gap <= knn_worst(s)) search_k(t, second, qx, qy, qz, s);
}
int lo_kdtree_knn(const lo_kdtree *t, float qx, float qy, float qz, int k, int *idx, float *d2)
{
    knn_set s = { k, 0, idx, d2 };
    if (t->n > 0 && k > 0) search_k(t, 0, qx, qy, qz, &s);
    return s.n;
}

int lo_brute_nn(const lo_pt *pts, int n, float qx, float qy, float qz, float *d2)
{
    float best = FLT_MAX; int bi = -1;
    for (int i = 0; i < n; i++) {
        float d = dist2(&pts[i], qx, qy, qz);
        if (d < best) { best = d; bi = i; }
    }
    *d2 = best;
    return bi;
}
