"""ctypes bindings of the CPU oracle (oracle/liblmono_oracle.so).

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  PARITY UNPINNED (see oracle/lo_oracle.h).  The synthetic S1 world / trajectory generator is input
plumbing and lives in workloads/s1.py (re-exported here for the tests).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liblmono_oracle.so")
MAX_RINGS = 64


def build(force=False):
    """Compile the oracle with gcc (no-op when the .so is newer than its sources)."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(s) for s in srcs)):
        return _LIB_PATH
    subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])
    return _LIB_PATH


class ScanregInfo(C.Structure):
    _fields_ = [("n_cloud", C.c_int),
                ("ring_begin", C.c_int * (MAX_RINGS + 1)),
                ("scan_start", C.c_int * MAX_RINGS),
                ("scan_end", C.c_int * MAX_RINGS),
                ("n_sharp", C.c_int), ("n_less_sharp", C.c_int), ("n_flat", C.c_int), ("n_less_flat", C.c_int)]


class OdomStats(C.Structure):
    _fields_ = [("n_corner_corr", C.c_int * 2), ("n_plane_corr", C.c_int * 2), ("lm_iters", C.c_int * 2),
                ("initial_cost", C.c_double * 2), ("final_cost", C.c_double * 2)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.lo_kdtree_build.restype = C.c_void_p
        _lib.lo_kdtree_nn.restype = C.c_int
        _lib.lo_brute_nn.restype = C.c_int
    return _lib


def _fp(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def scanreg(xyzi, n_lines=64, min_range=5.0):
    """Returns dict(cloud, curvature, label, sharp, less_sharp, flat, less_flat, info)."""
    xyzi = np.ascontiguousarray(xyzi, dtype=np.float32).reshape(-1, 4)
    n = xyzi.shape[0]
    cap = max(n, 1)
    cloud = np.zeros((cap, 4), np.float32)
    curv = np.zeros(cap, np.float32)
    label = np.zeros(cap, np.int32)
    outs = [np.zeros((cap, 4), np.float32) for _ in range(4)]
    info = ScanregInfo()
    rc = lib().lo_scanreg(_fp(xyzi, C.c_float), C.c_int(n), C.c_int(n_lines), C.c_float(min_range),
                          _fp(cloud, C.c_float), _fp(curv, C.c_float), _fp(label, C.c_int32),
                          *[_fp(o, C.c_float) for o in outs], C.byref(info))
    if rc != 0:
        raise RuntimeError("lo_scanreg failed: %d" % rc)
    nc = info.n_cloud
    return dict(cloud=cloud[:nc], curvature=curv[:nc], label=label[:nc],
                sharp=outs[0][:info.n_sharp], less_sharp=outs[1][:info.n_less_sharp],
                flat=outs[2][:info.n_flat], less_flat=outs[3][:info.n_less_flat],
                ring_begin=np.array(info.ring_begin[:], np.int32), info=info)


def odom_step(sharp, flat, corner_last, surf_last, q, t, use_kdtree=True, want_corr=False):
    arrs = [np.ascontiguousarray(a, np.float32).reshape(-1, 4) for a in (sharp, flat, corner_last, surf_last)]
    q = np.array(q, np.float64).copy()
    t = np.array(t, np.float64).copy()
    st = OdomStats()
    corr = np.zeros((2, arrs[0].shape[0] + arrs[1].shape[0], 4), np.int32) if want_corr else None
    lib().lo_odom_step(_fp(arrs[0], C.c_float), C.c_int(arrs[0].shape[0]), _fp(arrs[1], C.c_float), C.c_int(arrs[1].shape[0]),
                       _fp(arrs[2], C.c_float), C.c_int(arrs[2].shape[0]), _fp(arrs[3], C.c_float), C.c_int(arrs[3].shape[0]),
                       _fp(q, C.c_double), _fp(t, C.c_double), C.c_int(1 if use_kdtree else 0), C.byref(st),
                       _fp(corr, C.c_int32) if want_corr else None)
    return q, t, st, corr


def nn(points, queries, use_kdtree=True):
    pts = np.ascontiguousarray(points, np.float32).reshape(-1, 4)
    qs = np.ascontiguousarray(queries, np.float32).reshape(-1, 3)
    idx = np.zeros(len(qs), np.int32)
    d2 = np.zeros(len(qs), np.float32)
    L = lib()
    tree = C.c_void_p(L.lo_kdtree_build(_fp(pts, C.c_float), C.c_int(len(pts)))) if use_kdtree else None
    dd = C.c_float()
    for i, qq in enumerate(qs):
        if use_kdtree:
            idx[i] = L.lo_kdtree_nn(tree, C.c_float(qq[0]), C.c_float(qq[1]), C.c_float(qq[2]), C.byref(dd))
        else:
            idx[i] = L.lo_brute_nn(_fp(pts, C.c_float), C.c_int(len(pts)), C.c_float(qq[0]), C.c_float(qq[1]), C.c_float(qq[2]), C.byref(dd))
        d2[i] = dd.value
    if use_kdtree:
        L.lo_kdtree_free(tree)
    return idx, d2


def run_sequence(xyzi, offsets, n_lines=64, min_range=5.0, n_chains=1, lead=0, use_kdtree=True, threads=1):
    """Full CPU path over a sequence.  Returns dict(incr[n,7], poses[n,7], feat_counts[n,4], stage_ms[2])."""
    xyzi = np.ascontiguousarray(xyzi, np.float32).reshape(-1, 4)
    offsets = np.ascontiguousarray(offsets, np.int64)
    n = len(offsets) - 1
    incr = np.zeros((n, 7)); poses = np.zeros((n, 7))
    counts = np.zeros((n, 4), np.int32)
    ms = np.zeros(2)
    rc = lib().lo_run_sequence(_fp(xyzi, C.c_float), _fp(offsets, C.c_int64), C.c_int(n), C.c_int(n_lines), C.c_float(min_range),
                               C.c_int(n_chains), C.c_int(lead), C.c_int(1 if use_kdtree else 0), C.c_int(threads),
                               _fp(incr, C.c_double), _fp(poses, C.c_double), _fp(counts, C.c_int32), _fp(ms, C.c_double))
    if rc != 0:
        raise RuntimeError("lo_run_sequence failed: %d" % rc)
    return dict(incr=incr, poses=poses, feat_counts=counts, stage_ms=ms)


# --------------------------------------------------------------------------------------------
# laserMapping
# --------------------------------------------------------------------------------------------
class MapStats(C.Structure):
    _fields_ = [("n_corner_stack", C.c_int), ("n_surf_stack", C.c_int), ("n_corner_map", C.c_int), ("n_surf_map", C.c_int),
                ("n_edge", C.c_int * 2), ("n_plane", C.c_int * 2), ("lm_iters", C.c_int * 2), ("final_cost", C.c_double * 2)]


class Map:
    """A-LOAM laserMapping state (21 x 21 x 11 cubes of 50 m); process() = one frame."""

    def __init__(self, line_res=0.4, plane_res=0.8):
        L = lib()
        L.lo_map_create.restype = C.c_void_p
        L.lo_map_create.argtypes = [C.c_float, C.c_float]
        L.lo_map_free.argtypes = [C.c_void_p]
        L.lo_map_process.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.lo_map_cube.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.lo_map_centre.argtypes = [C.c_void_p, C.c_void_p]
        self.h = L.lo_map_create(line_res, plane_res)

    def __del__(self):
        if getattr(self, "h", None):
            lib().lo_map_free(self.h)
            self.h = None

    def process(self, corner_last, surf_last, q_wodom, t_wodom):
        """corner_last / surf_last: [n,4] float32 (less-sharp / less-flat clouds).  Returns (q_w_curr, t_w_curr, MapStats)."""
        c = np.ascontiguousarray(corner_last, np.float32).reshape(-1, 4); s = np.ascontiguousarray(surf_last, np.float32).reshape(-1, 4)
        q = np.ascontiguousarray(q_wodom, np.float64); t = np.ascontiguousarray(t_wodom, np.float64)
        qo = np.zeros(4); to = np.zeros(3); st = MapStats()
        lib().lo_map_process(self.h, c.ctypes.data, len(c), s.ctypes.data, len(s), q.ctypes.data, t.ctypes.data,
                             qo.ctypes.data, to.ctypes.data, C.byref(st))
        return qo, to, st

    def cube(self, which, i, j, k):
        ptr = C.c_void_p()
        n = lib().lo_map_cube(self.h, which, i, j, k, C.byref(ptr))
        if n == 0:
            return np.zeros((0, 4), np.float32)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), (n, 4)).copy()

    def centre(self):
        cen = np.zeros(3, np.int32)
        lib().lo_map_centre(self.h, cen.ctypes.data)
        return cen

    def all_points(self, which):
        out = [self.cube(which, i, j, k) for k in range(11) for j in range(21) for i in range(21)]
        return np.concatenate(out) if out else np.zeros((0, 4), np.float32)


def run_mapping(xyzi, offsets, poses_odom, n_lines=64, min_range=5.0, line_res=0.4, plane_res=0.8, threads=1):
    """scanRegistration + laserMapping over a sequence with the given laserOdometry poses [n,7].  Returns dict(poses[n,7],
    stats[n], stage_ms[2])."""
    xyzi = np.ascontiguousarray(xyzi, np.float32).reshape(-1, 4)
    offsets = np.ascontiguousarray(offsets, np.int64)
    po = np.ascontiguousarray(poses_odom, np.float64)
    n = len(offsets) - 1
    out = np.zeros((n, 7)); ms = np.zeros(2)
    stats = (MapStats * n)()
    rc = lib().lo_run_mapping(_fp(xyzi, C.c_float), _fp(offsets, C.c_int64), C.c_int(n), C.c_int(n_lines), C.c_float(min_range),
                              C.c_float(line_res), C.c_float(plane_res), C.c_int(threads), _fp(po, C.c_double), _fp(out, C.c_double),
                              stats, _fp(ms, C.c_double))
    if rc != 0:
        raise RuntimeError("lo_run_mapping failed: %d" % rc)
    return dict(poses=out, stats=list(stats), stage_ms=ms)


class Corr(C.Structure):
    _fields_ = [("kind", C.c_int), ("cp", C.c_float * 3), ("a", C.c_double * 3), ("b", C.c_double * 3)]


def voxel_filter(pts, leaf):
    """pcl::VoxelGrid restated: [n,4] float32 -> centroids in ascending cell order."""
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 4)
    out = np.zeros_like(pts)
    L = lib()
    L.lo_voxel_filter.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_void_p]
    n = L.lo_voxel_filter(pts.ctypes.data, len(pts), C.c_float(np.float32(1.0) / np.float32(leaf)), out.ctypes.data)
    return out[:n].copy()


def map_refine(cmap, smap, cstack, sstack, x):
    """Optimisation part of one laserMapping frame on explicit clouds.  Returns (x [7], MapStats, list of Corr of the last
    outer iteration)."""
    cm = np.ascontiguousarray(cmap, np.float32).reshape(-1, 4); sm = np.ascontiguousarray(smap, np.float32).reshape(-1, 4)
    cs = np.ascontiguousarray(cstack, np.float32).reshape(-1, 4); ss = np.ascontiguousarray(sstack, np.float32).reshape(-1, 4)
    xo = np.ascontiguousarray(x, np.float64).copy()
    st = MapStats(); n_out = C.c_int(0)
    corr = (Corr * (len(cs) + len(ss) + 1))()
    L = lib()
    L.lo_map_refine.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.lo_map_refine(cm.ctypes.data, len(cm), sm.ctypes.data, len(sm), cs.ctypes.data, len(cs), ss.ctypes.data, len(ss),
                    xo.ctypes.data, C.byref(st), corr, C.byref(n_out))
    return xo, st, list(corr[:n_out.value])


def knn(pts, q, k):
    """k nearest neighbours of q in pts [n,4] through the kd-tree: (indices, squared distances)."""
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 4)
    L = lib()
    L.lo_kdtree_build.restype = C.c_void_p
    L.lo_kdtree_build.argtypes = [C.c_void_p, C.c_int]
    L.lo_kdtree_free.argtypes = [C.c_void_p]
    L.lo_kdtree_knn.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p]
    t = L.lo_kdtree_build(pts.ctypes.data, len(pts))
    idx = np.zeros(k, np.int32); d2 = np.zeros(k, np.float32)
    n = L.lo_kdtree_knn(t, float(q[0]), float(q[1]), float(q[2]), k, idx.ctypes.data, d2.ctypes.data)
    L.lo_kdtree_free(t)
    return idx[:n], d2[:n]


def sym_eig3(A):
    A = np.ascontiguousarray(A, np.float64).reshape(9)
    ev = np.zeros(3); vec = np.zeros(9)
    lib().lo_sym_eig3(A.ctypes.data_as(C.c_void_p), ev.ctypes.data_as(C.c_void_p), vec.ctypes.data_as(C.c_void_p))
    return ev, vec.reshape(3, 3)


def plane_fit5(pts):
    P = np.ascontiguousarray(pts, np.float64).reshape(15)
    n = np.zeros(3); d = C.c_double()
    ok = lib().lo_plane_fit5(P.ctypes.data_as(C.c_void_p), n.ctypes.data_as(C.c_void_p), C.byref(d))
    return bool(ok), n, d.value


# Synthetic workload S1 lives in workloads/s1.py (input plumbing, not part of the oracle); re-exported for the tests.
from workloads.s1 import S1World, hdl64_elevations_rad  # noqa: E402,F401


def gt_relative(poses):
    """Ground-truth sensor-frame poses relative to scan 0: returns [n,7] (q xyzw, t)."""
    out = np.zeros((len(poses), 7))
    x0, y0, _, yaw0 = poses[0]
    c, s = np.cos(-yaw0), np.sin(-yaw0)
    for k, (x, y, z, yaw) in enumerate(poses):
        dx, dy = x - x0, y - y0
        out[k, 4] = c * dx - s * dy
        out[k, 5] = s * dx + c * dy
        out[k, 6] = z - poses[0][2]
        half = 0.5 * (yaw - yaw0)
        out[k, 2] = np.sin(half); out[k, 3] = np.cos(half)
    return out


def ate(poses_a, poses_b):
    """RMS translation difference between two [n,7] pose arrays expressed in the same frame (no alignment)."""
    d = poses_a[:, 4:7] - poses_b[:, 4:7]
    return float(np.sqrt((d ** 2).sum(1).mean()))


# --------------------------------------------------------------------------------------------
# BA factors (oracle/lo_ba.c); packed layouts as in include/lmono_hip.h
# --------------------------------------------------------------------------------------------
FACTOR_DIMS = {  # kind: (n_params, n_consts, n_info, n_res, n_jac)
    0: (14, 24, 36, 6, 84),    # LASER
    1: (22, 4, 4, 2, 44),      # MONO
    2: (7, 16, 2, 6, 42),      # PRIOR
    3: (1, 44, 1, 2, 2),       # REPROJ
}


def factor_eval(kind, params, consts, info, want_jac=True):
    npar, ncon, ninf, nres, njac = FACTOR_DIMS[kind]
    params = np.ascontiguousarray(params, np.float64).reshape(-1, npar)
    consts = np.ascontiguousarray(consts, np.float64).reshape(-1, ncon)
    info = np.ascontiguousarray(info, np.float64).reshape(ninf)
    n = len(params)
    r = np.zeros((n, nres)); J = np.zeros((n, njac)) if want_jac else None
    rc = lib().lo_factor_eval(C.c_int(kind), C.c_int(n), _fp(params, C.c_double), _fp(consts, C.c_double), _fp(info, C.c_double),
                              _fp(r, C.c_double), _fp(J, C.c_double) if want_jac else None)
    if rc != 0:
        raise RuntimeError("lo_factor_eval failed")
    return r, J


# --------------------------------------------------------------------------------------------
# BA window solve (oracle/lo_ba_solve.c)
# --------------------------------------------------------------------------------------------
class BaProblemC(C.Structure):
    _fields_ = [("n_poses", C.c_int), ("n_feat", C.c_int), ("n_obs", C.c_int),
                ("use_prior", C.c_int), ("ex_constant", C.c_int), ("use_mono", C.c_int), ("max_iter", C.c_int),
                ("poses", C.POINTER(C.c_double)), ("ex", C.POINTER(C.c_double)), ("inv_depth", C.POINTER(C.c_double)),
                ("obs_feat", C.POINTER(C.c_int32)), ("obs_i", C.POINTER(C.c_int32)), ("obs_j", C.POINTER(C.c_int32)),
                ("obs_pts", C.POINTER(C.c_double)), ("laser_consts", C.POINTER(C.c_double)),
                ("laser_info", C.POINTER(C.c_double)), ("mono_info", C.POINTER(C.c_double)),
                ("prior_T", C.POINTER(C.c_double)), ("prior_w", C.POINTER(C.c_double))]


class BaSummary(C.Structure):
    _fields_ = [("initial_cost", C.c_double), ("final_cost", C.c_double), ("iterations", C.c_int), ("termination", C.c_int),
                ("n_successful", C.c_int), ("n_unsuccessful", C.c_int)]


def ba_solve(w, max_iter=30):
    """w: dict from tests/ba_cases.make_window().  Returns (poses, ex, inv_depth, summary) without touching w."""
    poses = np.ascontiguousarray(w["poses"], np.float64).copy()
    ex = np.ascontiguousarray(w["ex"], np.float64).copy()
    invd = np.ascontiguousarray(w["inv_depth"], np.float64).copy()
    keep = [np.ascontiguousarray(w[k], np.int32) for k in ("obs_feat", "obs_i", "obs_j")]
    pts = np.ascontiguousarray(w["obs_pts"], np.float64)
    lc = np.ascontiguousarray(w["laser_consts"], np.float64)
    li = np.ascontiguousarray(w["laser_info"], np.float64); mi = np.ascontiguousarray(w["mono_info"], np.float64)
    pT = np.ascontiguousarray(w["prior_T"], np.float64); pw = np.ascontiguousarray(w["prior_w"], np.float64)
    p = BaProblemC(len(poses), len(invd), len(keep[0]), int(w["use_prior"]), int(w["ex_constant"]), int(w["use_mono"]), max_iter,
                   _fp(poses, C.c_double), _fp(ex, C.c_double), _fp(invd, C.c_double),
                   _fp(keep[0], C.c_int32), _fp(keep[1], C.c_int32), _fp(keep[2], C.c_int32),
                   _fp(pts, C.c_double), _fp(lc, C.c_double), _fp(li, C.c_double), _fp(mi, C.c_double), _fp(pT, C.c_double), _fp(pw, C.c_double))
    sm = BaSummary()
    lib().lo_ba_solve(C.byref(p), C.byref(sm))
    return poses, ex, invd, sm


def ba_reanchor(poses, R0_before, P0_before):
    poses = np.ascontiguousarray(poses, np.float64)
    n = len(poses)
    R = np.zeros((n, 9)); P = np.zeros((n, 3))
    R0 = np.ascontiguousarray(R0_before, np.float64); P0 = np.ascontiguousarray(P0_before, np.float64)
    lib().lo_ba_reanchor(_fp(poses, C.c_double), C.c_int(n), _fp(R0, C.c_double), _fp(P0, C.c_double), _fp(R, C.c_double), _fp(P, C.c_double))
    return R.reshape(n, 3, 3), P


# --------------------------------------------------------------------------------------------
# per-feature numerics (oracle/lo_ba_feat.c)
# --------------------------------------------------------------------------------------------
def _feat_args(Rs, Ps, tlc, start, off, pts):
    return (np.ascontiguousarray(Rs, np.float64).reshape(-1, 9), np.ascontiguousarray(Ps, np.float64).reshape(-1, 3),
            np.ascontiguousarray(tlc, np.float64).reshape(16), np.ascontiguousarray(start, np.int32),
            np.ascontiguousarray(off, np.int32), np.ascontiguousarray(pts, np.float64).reshape(-1, 2))


def triangulate(Rs, Ps, tlc, start, off, pts, depth, track_cnt=3, window_size=10, weight=1500.0, refine_iters=50):
    """FeatureManager::triangulate: returns (depth_after_init, depth_after_refine, solve_flag)."""
    Rs, Ps, tlc, start, off, pts = _feat_args(Rs, Ps, tlc, start, off, pts)
    n = len(start)
    d = np.ascontiguousarray(depth, np.float64).copy()
    lib().lo_triangulate_init(_fp(Rs, C.c_double), _fp(Ps, C.c_double), _fp(tlc, C.c_double), C.c_int(n), _fp(start, C.c_int32),
                              _fp(off, C.c_int32), _fp(pts, C.c_double), _fp(d, C.c_double), C.c_int(track_cnt))
    d0 = d.copy()
    flag = np.zeros(n, np.int32)
    if refine_iters >= 0:
        lib().lo_depth_refine(_fp(Rs, C.c_double), _fp(Ps, C.c_double), _fp(tlc, C.c_double), C.c_int(n), _fp(start, C.c_int32),
                              _fp(off, C.c_int32), _fp(pts, C.c_double), _fp(d, C.c_double), _fp(flag, C.c_int32), C.c_int(track_cnt),
                              C.c_int(window_size), C.c_double(weight), C.c_int(refine_iters))
    return d0, d, flag


def outlier_scores(Rs, Ps, tlc, start, off, pts, depth, track_cnt=3, weight=1500.0):
    Rs, Ps, tlc, start, off, pts = _feat_args(Rs, Ps, tlc, start, off, pts)
    n = len(start)
    d = np.ascontiguousarray(depth, np.float64)
    sc = np.zeros(n)
    lib().lo_outlier_scores(_fp(Rs, C.c_double), _fp(Ps, C.c_double), _fp(tlc, C.c_double), C.c_int(n), _fp(start, C.c_int32),
                            _fp(off, C.c_int32), _fp(pts, C.c_double), _fp(d, C.c_double), C.c_int(track_cnt), C.c_double(weight), _fp(sc, C.c_double))
    return sc


def shift_depth(back_R0, back_P0, R1, P1, tlc, pt_i, depth):
    a = [np.ascontiguousarray(v, np.float64).ravel() for v in (back_R0, back_P0, R1, P1, tlc)]
    pt = np.ascontiguousarray(pt_i, np.float64).reshape(-1, 2); d = np.ascontiguousarray(depth, np.float64)
    out = np.zeros(len(d))
    lib().lo_shift_depth(*[_fp(v, C.c_double) for v in a], C.c_int(len(d)), _fp(pt, C.c_double), _fp(d, C.c_double), _fp(out, C.c_double))
    return out


def marginalize(w, pt_j_from="left"):
    """MARGIN_OLD prior from a window dict (tests/ba_cases.make_window).  Returns (lin_J [66,66], lin_r [66], m, x0 [11,7])."""
    sel = np.nonzero(np.asarray(w["obs_i"]) == 0)[0]
    feats = sorted(set(int(f) for f in np.asarray(w["obs_feat"])[sel]))
    remap = {f: k for k, f in enumerate(feats)}
    obs_feat = np.array([remap[int(f)] for f in np.asarray(w["obs_feat"])[sel]], np.int32)
    obs_j = np.ascontiguousarray(np.asarray(w["obs_j"])[sel], np.int32)
    pts = np.ascontiguousarray(np.asarray(w["obs_pts"])[sel], np.float64)
    invd = np.ascontiguousarray(np.asarray(w["inv_depth"])[feats], np.float64)
    poses = np.ascontiguousarray(w["poses"], np.float64); ex = np.ascontiguousarray(w["ex"], np.float64)
    lc = np.ascontiguousarray(w["laser_consts"][0], np.float64)
    li = np.ascontiguousarray(w["laser_info"], np.float64); mi = np.ascontiguousarray(w["mono_info"], np.float64)
    J = np.zeros((66, 66)); r = np.zeros(66); m = C.c_int(0)
    lib().lo_marginalize(_fp(poses, C.c_double), _fp(ex, C.c_double), C.c_int(len(feats)), _fp(invd, C.c_double), C.c_int(len(sel)),
                         _fp(obs_feat, C.c_int32), _fp(obs_j, C.c_int32), _fp(pts, C.c_double), _fp(lc, C.c_double), _fp(li, C.c_double),
                         _fp(mi, C.c_double), _fp(J, C.c_double), _fp(r, C.c_double), C.byref(m))
    x0 = np.concatenate([ex[None], poses[1:]], 0)
    return J, r, m.value, x0, dict(feats=feats, obs_feat=obs_feat, obs_j=obs_j, pts=pts, invd=invd)


def marg_evaluate(lin_J, lin_r, x0, x, want_jac=True):
    lin_J = np.ascontiguousarray(lin_J, np.float64); lin_r = np.ascontiguousarray(lin_r, np.float64)
    x0 = np.ascontiguousarray(x0, np.float64); x = np.ascontiguousarray(x, np.float64)
    res = np.zeros(66); jac = np.zeros((11, 66, 7)) if want_jac else None
    lib().lo_marg_evaluate(_fp(lin_J, C.c_double), _fp(lin_r, C.c_double), _fp(x0, C.c_double), _fp(x, C.c_double), _fp(res, C.c_double),
                           _fp(jac, C.c_double) if want_jac else None)
    return res, jac


def marg_second_new(lin_J, lin_r, x0, x, drop_block):
    """MARGIN_SECOND_NEW: previous prior (lin_J [n0,n0], lin_r [n0], x0 [nb,7]) evaluated at x [nb,7], block drop_block eliminated.
    -> (J [n,n], r [n]), n = n0 - 6, kept blocks in their old order."""
    lin_J = np.ascontiguousarray(lin_J, np.float64); lin_r = np.ascontiguousarray(lin_r, np.float64)
    x0 = np.ascontiguousarray(x0, np.float64); x = np.ascontiguousarray(x, np.float64)
    nb = len(x); n = 6 * nb - 6
    J = np.zeros((n, n)); r = np.zeros(n)
    rc = lib().lo_marg_second_new(C.c_int(nb), C.c_int(drop_block), _fp(lin_J, C.c_double), _fp(lin_r, C.c_double), _fp(x0, C.c_double), _fp(x, C.c_double),
                                  _fp(J, C.c_double), _fp(r, C.c_double))
    if rc != 0:
        raise ValueError("lo_marg_second_new failed")
    return J, r


def prior_dx(x0, x):
    """dx of Marginalization::Evaluate (MarginalizationFactor.cc:323-343) for any number of 7-wide blocks: numpy, for the tests."""
    x0 = np.asarray(x0, np.float64).reshape(-1, 7); x = np.asarray(x, np.float64).reshape(-1, 7)
    out = np.zeros((len(x), 6))
    out[:, :3] = x[:, :3] - x0[:, :3]
    for k in range(len(x)):
        a0 = x0[k, 3:]; a = x[k, 3:]
        n2 = (a0 ** 2).sum()
        i = np.array([-a0[0], -a0[1], -a0[2], a0[3]]) / n2
        rw = i[3] * a[3] - i[0] * a[0] - i[1] * a[1] - i[2] * a[2]
        rv = np.array([i[3] * a[0] + i[0] * a[3] + i[1] * a[2] - i[2] * a[1],
                       i[3] * a[1] + i[1] * a[3] + i[2] * a[0] - i[0] * a[2],
                       i[3] * a[2] + i[2] * a[3] + i[0] * a[1] - i[1] * a[0]])
        out[k, 3:] = 2.0 * (rv if rw >= 0 else -rv)
    return out.ravel()


# ---- colour projection (MapBuilder::associateToMap / depthFill; lo_colour.c) ----
class Cam(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("k1", C.c_double), ("k2", C.c_double), ("p1", C.c_double), ("p2", C.c_double),
                ("kernel_size", C.c_int), ("kernel_type", C.c_int), ("blur_type", C.c_int)]


PT_RGB = np.dtype([("x", np.float32), ("y", np.float32), ("z", np.float32), ("bgra", np.uint32)])


def kitti00_cam(width=1241, height=376, kernel_size=5, kernel_type=0, blur_type=0, dist=(0.0, 0.0, 0.0, 0.0)):
    """config/kitti00_cam.yaml + config/kitti_map_config_00.yaml (FULL kernel, bilateral blur, kernel_size 5)."""
    return Cam(width, height, 718.856, 718.856, 607.1928, 185.2157, dist[0], dist[1], dist[2], dist[3], kernel_size, kernel_type, blur_type)


def structuring_element(kind, k):
    m = np.zeros((k, k), np.uint8)
    lib().lo_structuring_element(C.c_int(kind), C.c_int(k), _fp(m, C.c_uint8))
    return m


def morph(img, mask, op):
    img = np.ascontiguousarray(img, np.uint8); mask = np.ascontiguousarray(mask, np.uint8)
    out = np.empty_like(img)
    lib().lo_morph(_fp(img, C.c_uint8), _fp(out, C.c_uint8), C.c_int(img.shape[1]), C.c_int(img.shape[0]), _fp(mask, C.c_uint8),
                   C.c_int(mask.shape[0]), C.c_int(op))
    return out


def _img_op(name, img, *extra):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.empty_like(img)
    getattr(lib(), name)(_fp(img, C.c_uint8), _fp(out, C.c_uint8), C.c_int(img.shape[1]), C.c_int(img.shape[0]), *extra)
    return out


def median5(img):
    return _img_op("lo_median5", img)


def bilateral5(img, sigma_color=1.5, sigma_space=2.0):
    return _img_op("lo_bilateral5", img, C.c_double(sigma_color), C.c_double(sigma_space))


def gauss5(img):
    return _img_op("lo_gauss5", img)


def depth_fill(cam, depth):
    d = np.array(depth, np.uint8, copy=True, order="C")
    assert d.shape == (cam.height, cam.width)
    lib().lo_depth_fill(C.byref(cam), _fp(d, C.c_uint8))
    return d


def depth_splat(cam, xyzi, M):
    xyzi = np.ascontiguousarray(xyzi, np.float32).reshape(-1, 4); M = np.ascontiguousarray(M, np.float64).reshape(16)
    d = np.zeros((cam.height, cam.width), np.uint8)
    lib().lo_depth_splat(C.byref(cam), _fp(xyzi, C.c_float), C.c_int(len(xyzi)), _fp(M, C.c_double), _fp(d, C.c_uint8))
    return d


def backproject(cam, depth, bgr):
    depth = np.ascontiguousarray(depth, np.uint8); bgr = np.ascontiguousarray(bgr, np.uint8)
    out = np.zeros(cam.width * cam.height, PT_RGB)
    lib().lo_backproject.restype = C.c_int
    n = lib().lo_backproject(C.byref(cam), _fp(depth, C.c_uint8), _fp(bgr, C.c_uint8), out.ctypes.data_as(C.c_void_p))
    return out[:n].copy()


def associate_to_map(cam, xyzi, M, bgr, q, t):
    """-> (filled depth map [h, w] u8, camera-frame cloud, world-frame cloud) of one associateToMap call."""
    xyzi = np.ascontiguousarray(xyzi, np.float32).reshape(-1, 4); M = np.ascontiguousarray(M, np.float64).reshape(16)
    bgr = np.ascontiguousarray(bgr, np.uint8); q = np.ascontiguousarray(q, np.float64); t = np.ascontiguousarray(t, np.float64)
    assert bgr.shape == (cam.height, cam.width, 3)
    d = np.zeros((cam.height, cam.width), np.uint8)
    a = np.zeros(cam.width * cam.height, PT_RGB); b = np.zeros(cam.width * cam.height, PT_RGB)
    lib().lo_associate_to_map.restype = C.c_int
    n = lib().lo_associate_to_map(C.byref(cam), _fp(xyzi, C.c_float), C.c_int(len(xyzi)), _fp(M, C.c_double), _fp(bgr, C.c_uint8),
                                  _fp(q, C.c_double), _fp(t, C.c_double), _fp(d, C.c_uint8), a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p))
    return d, a[:n].copy(), b[:n].copy()


def lift_projective(cam, u, v):
    r = np.zeros(3)
    lib().lo_lift_projective(C.byref(cam), C.c_double(u), C.c_double(v), _fp(r, C.c_double))
    return r


def space_to_plane(cam, P):
    P = np.ascontiguousarray(P, np.float64); p = np.zeros(2)
    lib().lo_space_to_plane(C.byref(cam), _fp(P, C.c_double), _fp(p, C.c_double))
    return p


# ---- loop-closure pose graph (new feature; lo_posegraph.c) ----
def pose_graph_optimize(poses_tq, loops, loop_info, max_iter=5, ordering=0):
    """poses_tq [n,7] (t, q xyzw), loops [L,2] (old, current), loop_info [L,8] (KeyFrame.cc:630-633 layout).
    -> (optimised poses [n,7], dict(iterations, initial_cost, final_cost, bandwidth))."""
    P = np.ascontiguousarray(poses_tq, np.float64).reshape(-1, 7)
    lp = np.ascontiguousarray(loops, np.int32).reshape(-1, 2); li = np.ascontiguousarray(loop_info, np.float64).reshape(-1, 8)
    out = np.zeros_like(P); st = np.zeros(6)
    lib().lo_pose_graph_optimize.restype = C.c_int
    rc = lib().lo_pose_graph_optimize(C.c_int(len(P)), _fp(P, C.c_double), C.c_int(len(lp)), _fp(lp, C.c_int32), _fp(li, C.c_double),
                                      C.c_int(max_iter), C.c_int(ordering), _fp(out, C.c_double), _fp(st, C.c_double))
    if rc != 0:
        raise ValueError("lo_pose_graph_optimize failed")
    return out, dict(iterations=int(st[0]), initial_cost=st[1], final_cost=st[2], bandwidth=int(st[3]), accepted=int(st[4]), rejected=int(st[5]))


def q2ypr(q):
    q = np.ascontiguousarray(q, np.float64); o = np.zeros(3)
    lib().lo_pg_q2ypr(_fp(q, C.c_double), _fp(o, C.c_double))
    return o


def ypr2q(ypr):
    y = np.ascontiguousarray(ypr, np.float64); o = np.zeros(4)
    lib().lo_pg_ypr2q(_fp(y, C.c_double), _fp(o, C.c_double))
    return o


class PoseGraph:
    """The oracle's pose graph in the rounds the product runs (linearise a rank's share -> sum over ranks -> step)."""

    def __init__(self, poses_tq, loops, loop_info, ordering=0):
        P = np.ascontiguousarray(poses_tq, np.float64).reshape(-1, 7)
        lp = np.ascontiguousarray(loops, np.int32).reshape(-1, 2); li = np.ascontiguousarray(loop_info, np.float64).reshape(-1, 8)
        L = lib()
        L.lo_pg_create.restype = C.c_void_p
        L.lo_pg_reduce_count.restype = C.c_int64
        self.n = len(P)
        self.h = C.c_void_p(L.lo_pg_create(C.c_int(self.n), _fp(P, C.c_double), C.c_int(len(lp)), _fp(lp, C.c_int32), _fp(li, C.c_double), C.c_int(ordering)))
        if not self.h:
            raise ValueError("lo_pg_create failed")
        self.reduce_count = L.lo_pg_reduce_count(self.h)
        self.bandwidth = L.lo_pg_bandwidth(self.h)
        self.reduce_tensor = np.zeros(self.reduce_count)

    def _buf(self):
        t = self.reduce_tensor
        return t if isinstance(t, np.ndarray) else t.numpy()       # a CPU torch tensor shares its memory with the array

    def linearise(self, rank=0, world=1):
        lib().lo_pg_linearise(self.h, C.c_int(rank), C.c_int(world), _fp(self._buf(), C.c_double))

    def step(self, max_iter=5):
        return bool(lib().lo_pg_step(self.h, _fp(self._buf(), C.c_double), C.c_int(max_iter)))

    def result(self):
        out = np.zeros((self.n, 7)); st = np.zeros(6)
        lib().lo_pg_result(self.h, _fp(out, C.c_double), _fp(st, C.c_double))
        return out, dict(iterations=int(st[0]), initial_cost=st[1], final_cost=st[2], bandwidth=int(st[3]), accepted=int(st[4]), rejected=int(st[5]))

    def __del__(self):
        if getattr(self, "h", None):
            lib().lo_pg_free(self.h)
            self.h = None
