// lmono_amd/csrc/batch.hpp -- device-resident working set of a batch of scans (HBM layout).
#pragma once
#include "common.hpp"

namespace lmono {

// hash grid over a "last" feature cloud (cell edge 1 m)
constexpr int kCornerTable = 16384;   // > kMaxLessSharp
constexpr int kSurfTable = 131072;    // surf cloud capacity = kSurfTable - 1 points
constexpr unsigned long long kEmptyKey = ~0ull;

// one hash-grid slot: 16 B so that a probe is a single dwordx4 load
struct __attribute__((aligned(16))) GridCell {
    unsigned long long key;   // packed cell coordinates, kEmptyKey when unused
    int start;                // first point of the cell in the cell-sorted copy
    int cnt;                  // points in the cell
};

struct BatchView {
    // ---- inputs
    const float4 *in;        // [total] raw x y z reflectance
    const int64_t *off;      // [n_scans + 1] point offsets (device copy)
    int n_scans;
    int scan0;               // first scan of this launch (the per-scan kernels' grids cover scans scan0 .. scan0 + grid - 1; 0 for a whole batch)
    int n_lines;
    int n_limit;             // > 0: the ring sort reads only the first n_limit points of a scan's slot (one-scan launches of the online stream)
    int has_grid;            // the hash grids (cg_* / sg_*) of this registration have been built (k_grid_build runs on demand)
    float min_range;
    // ---- ring-sorted cloud (same offsets as the input; n_cloud[s] valid points)
    float4 *cloud;           // [total] x y z (ring + 0.1 relTime)
    float *curv;             // [total]
    int8_t *label;           // [total]
    uint8_t *gap;            // [total] gap[i] = |p[i+1]-p[i]|^2 > 0.05
    int8_t *ring_tmp;        // [total] ring id per input point (-1 = discarded)
    int *seg_hist;           // [(total >> 10) + n_scans + 1][64] ring histogram of every 1024-point segment, then its exclusive prefix within the scan
    int *scan_ends;          // [n_scans][2] first / last valid input point (-1: none)
    int *scan_half;          // [n_scans] first input point past the half sweep (INT_MAX: none)
    float *scan_ori;         // [n_scans][2] start / end azimuth of the sweep
    int *ring_begin;         // [n_scans][65]
    int *n_cloud;            // [n_scans]
    int *status;             // [n_scans]
    // ---- per (scan, ring, sector) selections, indices local to the scan's cloud
    int *sel_sharp;          // [n_scans][64][6][20]
    int *sel_sharp_n;        // [n_scans][64][6]
    int *sel_flat;           // [n_scans][64][6][4]
    int *sel_flat_n;         // [n_scans][64][6]
    float4 *lf_tmp;          // [total] voxel-filtered less-flat points in ring slots
    int *lf_n;               // [n_scans][64]
    int *sel_todo;           // [1 + n_scans * 64] work list of k_select's second launch (rings longer than the small LDS slice)
    int *li_todo;            // [1 + n_scans * 2] work list of k_line_index's full-width launch: count, then scan * 2 + cloud
    int *vox_todo;           // [1 + n_scans * 64] work list of k_voxel's second instantiation: count, then (scan << 6 | ring)
    // ---- final feature clouds
    float4 *sharp;           // [n_scans][kMaxSharp]
    float4 *less_sharp;      // [n_scans][kMaxLessSharp]
    float4 *flat;            // [n_scans][kMaxFlat]
    float4 *less_flat;       // [total] (scan offsets as input)
    int *feat_n;             // [n_scans][4] sharp, less_sharp, flat, less_flat
    int *line_first_ge;      // [n_scans][2][66] (less_sharp, less_flat): first index with int(intensity) >= t
    int *line_last_le;       // [n_scans][2][66] last index with int(intensity) <= t
    // ---- hash grids of less_sharp / less_flat (used as the "last" clouds of the next scan)
    GridCell *cg_cell;       // [n_scans][kCornerTable]
    GridCell *sg_cell;       // [n_scans][kSurfTable]
    float4 *cg_pts;          // [n_scans][kMaxLessSharp]  cell-sorted copy, .w = original index bits
    float4 *sg_pts;          // [total]
    float4 *lbc_pts;         // [n_scans][kMaxLessSharp] less_sharp sorted by (line, azimuth bin), .w = original index bits
    float4 *lbs_pts;         // [total] same for less_flat
    int *lb_start;           // [n_scans][2][66*384+1] start of every (line, bin) bucket
    float4 *lb_elev;         // [n_scans][2][66] elevation angles of every line's points: (min, max, min over lines <= v, max over lines >= v)
    int *grid_mask;          // [n_scans][2] table size - 1 actually used (corner, surf): power of two > n
};

} // namespace lmono
