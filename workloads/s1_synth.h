/* workloads/s1_synth.h -- synthetic HDL-64 scan generator "S1" (SURVEY.md 8d): input plumbing for tests and bench.py */
#ifndef S1_SYNTH_H
#define S1_SYNTH_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int n_boxes;   const double *boxes;     /* [n_boxes][6]  xmin ymin zmin xmax ymax zmax */
    int n_cyls;    const double *cyls;      /* [n_cyls][4]   cx cy radius height(top z)     */
    double ground_z;
    int n_rings;   const double *elev_rad;  /* [n_rings] elevation angles                  */
    int n_az;                               /* azimuth steps per ring                       */
    double range_sigma, dropout, max_range;
    uint64_t seed;
    /* clutter (all zero = the tidy S1 world; appended so that the old worlds generate the same scans bit for bit):
     * stray_frac: fraction of the returns that come back at a RANDOM range between 2 m and the true one (dust, rain, multipath);
     * moving[n_moving][6]: cylinders cx0 cy0 vx vy radius top_z that move with constant velocity (t = scan_id * 0.1 s, wrapping in +-95 m);
     * sector_drop: probability that a (scan, ring) loses one azimuth sector of 5 .. 15 % of the ring (occlusion by the vehicle, a failing laser) */
    double stray_frac;
    int n_moving;  const double *moving;
    double sector_drop;
} lo_world;

/* pose: sensor position (x,y,z) and yaw.  Writes up to n_rings*n_az points (ring-major,
 * azimuth ascending in firing order) as xyzi float32; returns the number written. */
int lo_synth_scan(const lo_world *w, const double pose_xyzyaw[4], uint64_t scan_id, float *xyzi_out);
/* n scans at once (OpenMP over scans); out holds n slots of slot_floats floats, counts[n] points written per scan */
void lo_synth_scans(const lo_world *w, const double *poses_xyzyaw, uint64_t scan_id0, int n, float *out, int64_t slot_floats, int32_t *counts);

#ifdef __cplusplus
}
#endif
#endif
