"""CPU oracle of A-LOAM laserMapping: building blocks against numpy, map bookkeeping, and the
physical behaviour on the synthetic sequence (scan-to-map refinement removes most of the odometry drift)."""
import numpy as np


def test_knn_equals_bruteforce(oracle):
    rng = np.random.default_rng(3)
    pts = np.zeros((3000, 4), np.float32); pts[:, :3] = rng.uniform(-20, 20, (3000, 3))
    pts[100] = pts[50]                                     # exact duplicate: ties resolve to the lower index first
    for q in rng.uniform(-22, 22, (40, 3)):
        idx, d2 = oracle.knn(pts, q, 5)
        qf = q.astype(np.float32)
        d = ((pts[:, 0] - qf[0]) ** 2 + (pts[:, 1] - qf[1]) ** 2) + (pts[:, 2] - qf[2]) ** 2
        order = np.lexsort((np.arange(len(pts)), d))[:5]
        assert list(idx) == list(order) and np.array_equal(d2, d[order])
    idx, _ = oracle.knn(pts[:3], [0, 0, 0], 5)
    assert len(idx) == 3


def test_sym_eig3_and_plane_fit_match_numpy(oracle):
    rng = np.random.default_rng(4)
    for _ in range(50):
        Z = rng.normal(0, [3.0, 0.3, 0.05], (5, 3)) @ np.linalg.qr(rng.normal(size=(3, 3)))[0]
        A = Z.T @ Z
        ev, vec = oracle.sym_eig3(A)
        ref = np.linalg.eigvalsh(A)
        assert np.allclose(ev, ref, rtol=1e-10, atol=1e-12 * ref.max())
        assert np.allclose(A @ vec, vec * ev, atol=1e-9 * ref.max()) and np.allclose(vec.T @ vec, np.eye(3), atol=1e-12)
        n0 = rng.normal(size=3); n0 /= np.linalg.norm(n0)
        P = rng.normal(0, 2.0, (5, 3)); P -= np.outer(P @ n0, n0); P += n0 * rng.uniform(3, 30) + rng.normal(0, 0.01, (5, 3))
        ok, n, d = oracle.plane_fit5(P)
        x = np.linalg.lstsq(P, -np.ones(5), rcond=None)[0]
        assert ok and np.allclose(n, x / np.linalg.norm(x), atol=1e-9) and abs(d - 1 / np.linalg.norm(x)) < 1e-9 * d
    ok, _, _ = oracle.plane_fit5(np.array([[1, 0, 5], [0, 1, 5], [-1, 0, 5], [0, -1, 5], [0, 0, 6.0]]))
    assert not ok                                            # one point 0.8 m off the plane of the other four


def test_first_frame_only_fills_the_map(oracle, small_seq):
    ref = oracle.run_sequence(small_seq["xyzi"], small_seq["off"])
    out = oracle.run_mapping(small_seq["xyzi"][:small_seq["off"][1]], small_seq["off"][:2], ref["poses"][:1])
    st = out["stats"][0]
    assert st.n_corner_map == 0 and st.n_surf_map == 0 and st.n_edge[0] == 0 and st.n_plane[0] == 0
    assert np.allclose(out["poses"][0], [0, 0, 0, 1, 0, 0, 0])


def test_map_cubes_hold_voxel_filtered_points_in_their_own_cube(oracle, small_seq):
    import ctypes as C
    x, off = small_seq["xyzi"], small_seq["off"]
    ref = oracle.run_sequence(x, off)
    m = oracle.Map()
    for k in range(3):
        n = int(off[k + 1] - off[k])
        cloud = np.zeros((n, 4), np.float32); curv = np.zeros(n, np.float32); label = np.zeros(n, np.int32)
        sh = np.zeros((n, 4), np.float32); ls = np.zeros((n, 4), np.float32); fl = np.zeros((n, 4), np.float32); lf = np.zeros((n, 4), np.float32)
        info = oracle.ScanregInfo()
        oracle.lib().lo_scanreg(x[off[k]:off[k + 1]].ctypes.data_as(C.c_void_p), n, 64, C.c_float(5.0), cloud.ctypes.data_as(C.c_void_p),
                                curv.ctypes.data_as(C.c_void_p), label.ctypes.data_as(C.c_void_p), sh.ctypes.data_as(C.c_void_p),
                                ls.ctypes.data_as(C.c_void_p), fl.ctypes.data_as(C.c_void_p), lf.ctypes.data_as(C.c_void_p), C.byref(info))
        q, t, st = m.process(ls[:info.n_less_sharp], lf[:info.n_less_flat], ref["poses"][k, :4], ref["poses"][k, 4:])
    assert st.n_edge[1] > 20 and st.n_plane[1] > 200
    cen = m.centre()
    assert list(cen) == [10, 10, 5]
    total = 0
    for which, leaf in ((0, 0.4), (1, 0.8)):
        for i in range(21):
            for j in range(21):
                for k in range(11):
                    c = m.cube(which, i, j, k)
                    if len(c) == 0:
                        continue
                    total += len(c)
                    lo = np.array([(i - cen[0]) * 50.0 - 25.0, (j - cen[1]) * 50.0 - 25.0, (k - cen[2]) * 50.0 - 25.0])
                    # centroids of points of one cube stay inside it (convexity); 1e-3 slack for float rounding
                    assert (c[:, :3] >= lo - 1e-3).all() and (c[:, :3] <= lo + 50.0 + 1e-3).all()
                    cells = np.floor(c[:, :3].astype(np.float32) * np.float32(1.0 / np.float32(leaf))).astype(np.int64)
                    # after the filter at most one point per voxel can remain ... except that a centroid may round
                    # into a neighbouring voxel; allow a handful
                    assert len(np.unique(cells, axis=0)) >= len(c) - max(3, len(c) // 100)
    assert total > 1000


def test_mapping_removes_most_of_the_odometry_drift(oracle):
    w = oracle.S1World(n_az=500)
    traj = w.trajectory(40)
    xyzi, off = w.scans(traj)
    ref = oracle.run_sequence(xyzi, off)
    out = oracle.run_mapping(xyzi, off, ref["poses"])
    gt = oracle.gt_relative(traj)
    e_odo = oracle.ate(ref["poses"], gt); e_map = oracle.ate(out["poses"], gt)
    assert e_map < e_odo and e_map < 0.15, (e_odo, e_map)
    st = out["stats"][-1]
    assert st.n_edge[1] > 50 and st.n_plane[1] > 500 and 1 <= st.lm_iters[1] <= 4
