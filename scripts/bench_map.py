"""Throughput probe of the scan-to-map optimisation step (lmono_map_refine): n_streams independent streams, every one
with the map and scan clouds of frame `frame` of the synthetic S1 sequence.  Prints device times from the library's
hipEvents and the CPU oracle's time for the same step (test infrastructure, used here as the baseline only)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import lmono_amd
from oracle import oracle as O
import ctypes as C

n_streams = int(sys.argv[1]) if len(sys.argv) > 1 else 64
frame = int(sys.argv[2]) if len(sys.argv) > 2 else 10
w = O.S1World(); traj = w.trajectory(frame + 1); x, off = w.scans(traj)
ref = O.run_sequence(x, off)


def clouds(k):
    n = int(off[k + 1] - off[k]); bufs = [np.zeros((n, 4), np.float32) for _ in range(5)]
    curv = np.zeros(n, np.float32); label = np.zeros(n, np.int32); info = O.ScanregInfo(); P = lambda a: a.ctypes.data_as(C.c_void_p)
    O.lib().lo_scanreg(P(x[off[k]:off[k + 1]]), n, 64, C.c_float(5.0), P(bufs[0]), P(curv), P(label), P(bufs[1]), P(bufs[2]), P(bufs[3]), P(bufs[4]), C.byref(info))
    return bufs[2][:info.n_less_sharp].copy(), bufs[4][:info.n_less_flat].copy()


m = O.Map()
for k in range(frame):
    ls, lf = clouds(k); m.process(ls, lf, ref["poses"][k, :4], ref["poses"][k, 4:])
ls, lf = clouds(frame)
cmap, smap = m.all_points(0), m.all_points(1)
cs, ss = O.voxel_filter(ls, 0.4), O.voxel_filter(lf, 0.8)
x0 = ref["poses"][frame].copy()
t0 = time.time(); xr, st, _ = O.map_refine(cmap, smap, cs, ss, x0); cpu_s = time.time() - t0
ctx = lmono_amd.Context(0)
for _ in range(2):
    poses, stats, _ = ctx.map_refine([cmap] * n_streams, [smap] * n_streams, [cs] * n_streams, [ss] * n_streams, np.tile(x0, (n_streams, 1)))
assert np.abs(poses - xr).max() < 1e-9
print("map %d + %d points, scan %d + %d points, %d streams" % (len(cmap), len(smap), len(cs), len(ss), n_streams))
print("device: grid build %.3f ms, optimisation %.3f ms  ->  %.0f frames/s (optimisation step, device time)" % (stats[0, 6] / 1e3, stats[0, 7] / 1e3, n_streams / ((stats[0, 6] + stats[0, 7]) / 1e6)))
print("CPU oracle (1 thread): %.1f ms per frame -> %.1f frames/s" % (cpu_s * 1e3, 1 / cpu_s))
