"""CPU checks of the oracle's LiDAR half (parity unpinned by the reference: known-answer and property tests)."""
import numpy as np
import pytest


def test_det_atan_accuracy(oracle):
    import ctypes as C
    # lo_atan / lo_atan2 are static inline; exercise them through the ring assignment instead:
    # a point at elevation exactly between two HDL-64 lasers must land on a deterministic side, and
    # points on the laser centres must map to ring k.
    elev = oracle.hdl64_elevations_rad()
    elev[0] = np.deg2rad(1.99)   # +2 deg sits exactly on the `angle > 2` discard boundary
    # add enough points per ring for the cloud to exist: 30 azimuths
    az = np.linspace(-3.0, 3.0, 30)
    cloud = []
    for r in range(64):
        for a in az:
            cloud.append([20 * np.cos(elev[r]) * np.cos(a), 20 * np.cos(elev[r]) * np.sin(a), 20 * np.sin(elev[r]), 0])
    out = oracle.scanreg(np.array(cloud, np.float32))
    rb = out["ring_begin"]
    # rings 0..50 keep their 30 points, rings 51..63 are discarded (A-LOAM keeps ids <= 50)
    assert list(np.diff(rb)[:51]) == [30] * 51
    assert out["info"].n_cloud == 51 * 30
    assert np.array_equal(out["cloud"][:, 3].astype(np.int32), np.repeat(np.arange(51), 30))


def test_scanreg_invariants(oracle, small_seq):
    xyzi, off = small_seq["xyzi"], small_seq["off"]
    r = oracle.scanreg(xyzi[off[0]:off[1]])
    info = r["info"]
    lab, cv = r["label"], r["curvature"]
    assert info.n_sharp == (lab == 2).sum()
    assert info.n_less_sharp == ((lab == 2) | (lab == 1)).sum()
    assert info.n_flat == (lab == -1).sum()
    assert (cv[lab > 0] > 0.1).all() and (cv[lab == -1] < 0.1).all()
    rb = r["ring_begin"]
    for ring in range(51):
        lo, hi = rb[ring], rb[ring + 1]
        if hi - lo < 17:
            continue
        S, E = lo + 5, hi - 6
        for j in range(6):
            sp = S + (E - S) * j // 6
            ep = S + (E - S) * (j + 1) // 6 - 1
            seg = lab[sp:ep + 1]
            assert (seg == 2).sum() <= 2 and (seg > 0).sum() <= 20 and (seg == -1).sum() <= 4
    # less-flat points are voxel centroids: fewer than the candidates, all inside the cloud's bounding box
    assert 0 < info.n_less_flat <= (lab <= 0).sum()
    assert r["less_flat"][:, :3].min() >= r["cloud"][:, :3].min() - 1e-3
    # sharp points are a subset of less-sharp points
    ls = {tuple(p) for p in r["less_sharp"][:, :3]}
    assert all(tuple(p) in ls for p in r["sharp"][:, :3])


def test_scanreg_curvature_known_answer(oracle):
    """Points on a straight line at constant spacing have zero curvature; a single displaced point has c = (10 d)^2."""
    n = 200
    elev = oracle.hdl64_elevations_rad()[10]
    x = np.full(n, 30.0); y = np.linspace(-5, 5, n); z = np.hypot(x, y) * np.tan(elev)
    pts = np.stack([x, y, z, np.zeros(n)], 1).astype(np.float32)
    r = oracle.scanreg(pts, min_range=1.0)
    assert r["info"].n_cloud == n
    assert np.abs(r["curvature"][5:-5]).max() < 1e-6      # a straight chord across one laser cone is not exactly collinear in z
    pts2 = pts.copy(); pts2[100, 0] += 0.5
    r2 = oracle.scanreg(pts2, min_range=1.0)
    order = np.argsort(r2["cloud"][:, 1])
    c = r2["curvature"][order]
    assert abs(c[100] - 25.0) < 1e-2 and abs(c[99] - 0.25) < 1e-2


def test_kdtree_equals_bruteforce(oracle, small_seq):
    xyzi, off = small_seq["xyzi"], small_seq["off"]
    r = oracle.scanreg(xyzi[off[0]:off[1]])
    rng = np.random.default_rng(1)
    q = r["less_flat"][rng.integers(0, len(r["less_flat"]), 800), :3] + rng.normal(0, 0.5, (800, 3)).astype(np.float32)
    q[:50] += 40.0     # far queries
    i1, d1 = oracle.nn(r["less_flat"], q, True)
    i2, d2 = oracle.nn(r["less_flat"], q, False)
    assert np.array_equal(i1, i2) and np.array_equal(d1, d2)
    # duplicate points: ties resolve to the lowest index in both
    dup = np.concatenate([r["less_sharp"][:100], r["less_sharp"][:100]])
    i1, _ = oracle.nn(dup, dup[:100, :3], True)
    i2, _ = oracle.nn(dup, dup[:100, :3], False)
    assert np.array_equal(i1, np.arange(100)) and np.array_equal(i2, np.arange(100))


def test_odometry_recovers_known_motion(oracle, small_seq):
    """Scan-to-scan increments follow the synthetic ground truth (0.8 m per scan forward) within LOAM accuracy."""
    res = oracle.run_sequence(small_seq["xyzi"], small_seq["off"])
    gt = oracle.gt_relative(small_seq["poses"])
    assert oracle.ate(res["poses"], gt) < 0.35
    fwd = res["incr"][2:, 4]
    assert (fwd > 0.6).all() and (fwd < 1.0).all()
    q = res["incr"][:, :4]
    assert np.abs(np.linalg.norm(q, axis=1) - 1).max() < 1e-12


def test_odometry_identity_on_identical_scans(oracle, small_seq):
    """Registering a scan against itself yields (nearly) the identity increment."""
    xyzi, off = small_seq["xyzi"], small_seq["off"]
    a = xyzi[off[0]:off[1]]
    two = np.concatenate([a, a]); o = np.array([0, len(a), 2 * len(a)], np.int64)
    res = oracle.run_sequence(two, o)
    assert np.abs(res["incr"][1] - np.array([0, 0, 0, 1, 0, 0, 0])).max() < 1e-3   # planes match voxel centroids, not the points themselves


def test_chain_sharding_converges_to_sequential(oracle, small_seq):
    seq = oracle.run_sequence(small_seq["xyzi"], small_seq["off"])
    sh = oracle.run_sequence(small_seq["xyzi"], small_seq["off"], n_chains=2, lead=3)
    # chain 1 owns scans 3..5 and re-converges from an identity warm start during its lead-in
    assert np.abs(seq["incr"][:3] - sh["incr"][:3]).max() == 0.0
    assert oracle.ate(seq["poses"], sh["poses"]) < 0.01      # north_star: ATE within 1 cm of the CPU path


def _np_lm(residual_blocks, x, max_iter=4):
    """Independent numpy restatement of Ceres' trust-region loop (LEVENBERG_MARQUARDT, defaults, Huber(0.1)) with a
    NUMERIC Jacobian of r(Plus(x, delta)) at delta = 0 -- cross-checks the oracle's dual-number Jacobians and its loop."""
    def plus(x, d):
        nd = np.linalg.norm(d[:3])
        out = x.copy()
        if nd > 0:
            dq = np.concatenate([np.sin(nd) / nd * d[:3], [np.cos(nd)]])
            ax, ay, az, aw = dq; bx, by, bz, bw = x[:4]
            out[:4] = [aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz,
                       aw * bz + az * bw + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz]
        out[4:] = x[4:] + d[3:]
        return out

    def huber(s):
        return (s, 1.0) if s <= 0.01 else (0.2 * np.sqrt(s) - 0.01, 0.1 / np.sqrt(s))

    def evaluate(x, jac):
        blocks = residual_blocks(x)
        cost = sum(0.5 * huber(b @ b)[0] for b in blocks)
        if not jac:
            return cost, None, None
        h = 1e-6
        Jn = []
        for k in range(6):
            d = np.zeros(6); d[k] = h
            bp = residual_blocks(plus(x, d)); bm = residual_blocks(plus(x, -d))
            Jn.append([(p - m) / (2 * h) for p, m in zip(bp, bm)])
        H = np.zeros((6, 6)); g = np.zeros(6)
        for bi, b in enumerate(blocks):
            sr = np.sqrt(huber(b @ b)[1])
            J = np.stack([Jn[k][bi] for k in range(6)], 1) * sr
            H += J.T @ J; g += J.T @ (b * sr)
        return cost, H, g

    radius, dec, reuse = 1e4, 2.0, False
    cost, H, g = evaluate(x, True)
    scale = 1.0 / (1.0 + np.sqrt(np.diag(H)))
    diag = None
    for _ in range(max_iter):
        Hs = H * np.outer(scale, scale); gs = g * scale
        if not reuse:
            diag = np.clip(np.diag(Hs), 1e-6, 1e32)
        step = -np.linalg.solve(Hs + np.diag(diag / radius), gs)
        model = -(step @ gs + 0.5 * step @ Hs @ step)
        cand = plus(x, step * scale)
        ccost, _, _ = evaluate(cand, False)
        if np.linalg.norm(x - cand) <= 1e-8 * (np.linalg.norm(x) + 1e-8) or abs(cost - ccost) <= 1e-6 * cost:
            break
        rel = (cost - ccost) / model
        if rel > 1e-3:
            x = cand
            cost, H, g = evaluate(x, True)
            radius = min(radius / max(1.0 / 3.0, 1.0 - (2 * rel - 1) ** 3), 1e16); dec = 2.0; reuse = False
        else:
            radius /= dec; dec *= 2.0; reuse = True
    return x, cost


def test_lm_matches_independent_numpy_restatement(oracle, small_seq):
    xyzi, off = small_seq["xyzi"], small_seq["off"]
    f0 = oracle.scanreg(xyzi[off[0]:off[1]]); f1 = oracle.scanreg(xyzi[off[1]:off[2]])
    q, t, st, corr = oracle.odom_step(f1["sharp"], f1["flat"], f0["less_sharp"], f0["less_flat"],
                                      np.array([0, 0, 0, 1.0]), np.zeros(3), want_corr=True)
    ns = len(f1["sharp"])
    cl, sl = f0["less_sharp"][:, :3].astype(np.float64), f0["less_flat"][:, :3].astype(np.float64)
    sh, fl = f1["sharp"][:, :3].astype(np.float64), f1["flat"][:, :3].astype(np.float64)

    def rot(qq, v):
        u, w = qq[:3], qq[3]
        uv = 2 * np.cross(u, v)
        return v + w * uv + np.cross(u, uv)

    def make_blocks(c):
        def blocks(x):
            out = []
            for i, (a, b, cc, kind) in enumerate(c):
                if kind == 1:
                    lp = rot(x[:4], sh[i]) + x[4:]
                    out.append(np.cross(lp - cl[a], lp - cl[b]) / np.linalg.norm(cl[a] - cl[b]))
                elif kind == 2:
                    lp = rot(x[:4], fl[i - ns]) + x[4:]
                    nrm = np.cross(sl[a] - sl[b], sl[a] - sl[cc]); nrm /= np.linalg.norm(nrm)
                    out.append(np.array([(lp - sl[a]) @ nrm]))
            return out
        return blocks

    x = np.array([0, 0, 0, 1.0, 0, 0, 0])
    x, c0 = _np_lm(make_blocks(corr[0]), x)
    assert abs(c0 - st.final_cost[0]) < 1e-6 * c0
    x, c1 = _np_lm(make_blocks(corr[1]), x)
    assert abs(c1 - st.final_cost[1]) < 1e-6 * c1
    assert np.abs(x[:4] - q).max() < 1e-7 and np.abs(x[4:] - t).max() < 1e-6
    assert st.final_cost[1] <= st.initial_cost[1]
    assert st.n_corner_corr[1] > 50 and st.n_plane_corr[1] > 200
