"""workloads/s1.py -- synthetic workload S1: HDL-64 scans of a ground plane + boxes + poles world along a
figure-8 trajectory.  Input plumbing shared by tests, __graft_entry__.smoke() and bench.py; it is neither the hot path
nor the oracle."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libs1synth.so")
_lib = None


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("s1_synth.c", "s1_synth.h")]
    if (not force and os.path.exists(_LIB_PATH) and all(os.path.exists(s) for s in src)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(s) for s in src)):
        return _LIB_PATH
    if all(os.path.exists(s) for s in src):
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.lo_synth_scan.restype = C.c_int
    return _lib


def _fp(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class World(C.Structure):
    _fields_ = [("n_boxes", C.c_int), ("boxes", C.POINTER(C.c_double)),
                ("n_cyls", C.c_int), ("cyls", C.POINTER(C.c_double)),
                ("ground_z", C.c_double),
                ("n_rings", C.c_int), ("elev_rad", C.POINTER(C.c_double)),
                ("n_az", C.c_int),
                ("range_sigma", C.c_double), ("dropout", C.c_double), ("max_range", C.c_double),
                ("seed", C.c_uint64),
                ("stray_frac", C.c_double), ("n_moving", C.c_int), ("moving", C.POINTER(C.c_double)), ("sector_drop", C.c_double)]


def hdl64_elevations_rad():
    """HDL-64E: upper bank +2 .. -8.33 deg step 1/3 deg (32 lasers), lower bank -8.83 .. -24.33 step 1/2."""
    up = 2.0 - np.arange(32) / 3.0
    lo = -8.83 - np.arange(32) / 2.0
    return np.deg2rad(np.concatenate([up, lo]))


class S1World:
    """Ground plane z=-1.73 + 40 boxes + 60 poles in a 200 m x 200 m area, numpy default_rng(20240)."""

    def __init__(self, seed=20240, n_az=2000, n_rings=64, range_sigma=0.02, dropout=0.05, max_range=120.0, clutter=False):
        """clutter=True (the stress sequence, tests/golden/s1_seq02_oracle.npz): + 200 small boxes (0.3 .. 1.5 m: kerbs, bins, parked things), 20 % of
        the returns at a random range in front of the surface, 8 cylinders moving at 1 .. 6 m/s, and 15 % of the (scan, ring)s lose an azimuth
        sector of 5 .. 15 % -- nothing the chain schedule or the search budgets were ever tuned on."""
        rng = np.random.default_rng(seed)
        boxes = []
        for _ in range(40):
            cx, cy = rng.uniform(-95, 95, 2)
            if abs(cx) < 6 and abs(cy) < 6:
                cx += 15.0
            sx, sy = rng.uniform(2, 20), rng.uniform(2, 20)
            h = rng.uniform(2.5, 12)
            boxes.append([cx - sx / 2, cy - sy / 2, -1.73, cx + sx / 2, cy + sy / 2, -1.73 + h])
        cyls = []
        for _ in range(60):
            cx, cy = rng.uniform(-95, 95, 2)
            cyls.append([cx, cy, 0.15, -1.73 + rng.uniform(3, 8)])
        self.moving = np.zeros((0, 6))
        stray, sector = 0.0, 0.0
        if clutter:
            for _ in range(200):
                cx, cy = rng.uniform(-95, 95, 2)
                sx, sy, h = rng.uniform(0.3, 1.5), rng.uniform(0.3, 1.5), rng.uniform(0.3, 1.8)
                boxes.append([cx - sx / 2, cy - sy / 2, -1.73, cx + sx / 2, cy + sy / 2, -1.73 + h])
            mv = []
            for _ in range(8):
                cx, cy = rng.uniform(-95, 95, 2)
                sp, hd = rng.uniform(1, 6), rng.uniform(0, 2 * np.pi)
                mv.append([cx, cy, sp * np.cos(hd), sp * np.sin(hd), rng.uniform(0.25, 0.9), -1.73 + rng.uniform(1.2, 3.0)])
            self.moving = np.array(mv, np.float64)
            stray, sector = 0.20, 0.15
        self.boxes = np.array(boxes, np.float64)
        self.cyls = np.array(cyls, np.float64)
        self.elev = hdl64_elevations_rad()[:n_rings].copy() if n_rings == 64 else np.deg2rad(np.linspace(15, -25, n_rings))
        self.n_az = n_az
        self.n_rings = n_rings
        self.w = World(len(boxes), _fp(self.boxes, C.c_double), len(cyls), _fp(self.cyls, C.c_double), -1.73,
                       n_rings, _fp(self.elev, C.c_double), n_az, range_sigma, dropout, max_range, seed,
                       stray, len(self.moving), _fp(self.moving, C.c_double) if len(self.moving) else None, sector)

    def trajectory(self, n_scans, dt=0.1, speed=8.0):
        """Planar figure-8 (Gerono lemniscate scaled so that |yaw rate| <= 0.3 rad/s), keeping clear of boxes is
        not attempted: the sensor may pass through obstacles, rays then start inside them (harmless)."""
        a = 60.0
        # arc-length parameterisation by numerical integration
        u = np.linspace(0, 2 * np.pi, 20001)
        x = a * np.sin(u); y = a * np.sin(u) * np.cos(u) * 0.9
        ds = np.hypot(np.diff(x), np.diff(y))
        s = np.concatenate([[0], np.cumsum(ds)])
        total = s[-1]
        sk = (np.arange(n_scans) * dt * speed) % total
        uk = np.interp(sk, s, u)
        xk = a * np.sin(uk); yk = a * np.sin(uk) * np.cos(uk) * 0.9
        dx = a * np.cos(uk); dy = a * 0.9 * (np.cos(uk) ** 2 - np.sin(uk) ** 2)
        yaw = np.unwrap(np.arctan2(dy, dx))
        return np.stack([xk, yk, np.zeros(n_scans), yaw], 1)

    def trajectory_clover(self, n_scans, dt=0.1):
        """Held-out trajectory (not the one the chain schedule was tuned on): a three-leaf clover r = 60 + 18 cos(3 phi) driven at a
        varying speed v(t) = 7 + 3 sin(2 pi t / 37 s) m/s (4 .. 10 m/s; the figure-8 runs at a constant 8), yaw along the tangent."""
        u = np.linspace(0, 2 * np.pi, 40001)
        r = 60.0 + 18.0 * np.cos(3 * u)
        x = r * np.cos(u); y = r * np.sin(u)
        ds = np.hypot(np.diff(x), np.diff(y))
        s = np.concatenate([[0], np.cumsum(ds)])
        total = s[-1]
        t = np.arange(n_scans) * dt
        # distance driven: integral of v(t)
        sk = (7.0 * t + 3.0 * 37.0 / (2 * np.pi) * (1.0 - np.cos(2 * np.pi * t / 37.0))) % total
        uk = np.interp(sk, s, u)
        rk = 60.0 + 18.0 * np.cos(3 * uk)
        xk = rk * np.cos(uk); yk = rk * np.sin(uk)
        drk = -54.0 * np.sin(3 * uk)
        dx = drk * np.cos(uk) - rk * np.sin(uk); dy = drk * np.sin(uk) + rk * np.cos(uk)
        yaw = np.unwrap(np.arctan2(dy, dx))
        return np.stack([xk, yk, np.zeros(n_scans), yaw], 1)

    def scans(self, poses, scan_id0=0):
        """Returns (xyzi [N,4] float32 concatenated, offsets int64 [n+1])."""
        poses = np.ascontiguousarray(poses, np.float64)
        n = len(poses)
        slot = self.n_rings * self.n_az * 4
        L = lib()
        step = 64
        buf = np.empty((min(step, n), slot), np.float32)      # reused across chunks
        out = np.empty((n * self.n_rings * self.n_az, 4), np.float32)   # upper bound, trimmed at the end (lazy pages)
        offsets = np.zeros(n + 1, np.int64)
        for c0 in range(0, n, step):
            m = min(step, n - c0)
            counts = np.zeros(m, np.int32)
            L.lo_synth_scans(C.byref(self.w), _fp(poses[c0:c0 + m], C.c_double), C.c_uint64(scan_id0 + c0), C.c_int(m),
                             _fp(buf, C.c_float), C.c_int64(slot), _fp(counts, C.c_int32))
            for i in range(m):
                o = offsets[c0 + i]
                out[o:o + counts[i]] = buf[i, :counts[i] * 4].reshape(-1, 4)
                offsets[c0 + i + 1] = o + counts[i]
        return out[:offsets[n]], offsets
