"""Run by tests/test_lidar_gpu.py::test_search_formulations_agree_in_the_diagnostic_build in a child process whose LMONO_HIP_LIB points at
lmono_amd/lib/liblmono_hip_diag.so (-DLMONO_DIAG_SEARCH): the product library holds the default search only (k_corr_flat over the
(azimuth bin, scan line) index); the round-1 search (k_correspond on 1 m hash grids, an independent formulation of the same exact search) lives
in the diagnostic build, where the two are checked against each other and against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _register(ctx, xyzi, off, n_lines=64, min_range=5.0):
    import torch
    import lmono_amd
    dev = torch.from_numpy(xyzi).cuda()
    batch = lmono_amd.ScanBatch(ctx, len(off) - 1, len(xyzi))
    batch.scanreg(dev.data_ptr(), off, n_lines, min_range, keepalive=dev)
    return batch


def test_tile_search_equals_global_search(oracle, gpu_ctx, full_seq):
    import lmono_amd
    assert b"diagnostic build" in lmono_amd.load_library().lmono_version()
    """The flat search (k_corr_flat + the deferred list) and the hash-grid search (k_correspond) return the same
    correspondence indices for good and bad warm starts, and the same odometry bit for bit (same residual blocks, same solve)."""
    xyzi, off = full_seq["xyzi"], full_seq["off"]
    batch = _register(gpu_ctx, xyzi, off)
    poses = [(np.array([0.0, 0.0, 0.0, 1.0]), np.array([0.0, 0.0, 0.0])),            # identity: 0.8 m off, wide searches
             (np.array([0.0, 0.0, 0.01, 1.0]), np.array([0.7, 0.02, 0.0])),
             (np.array([0.002, -0.001, 0.02, 1.0]), np.array([0.85, -0.05, 0.01])),
             (np.array([0.0, 0.0, 0.3, 1.0]), np.array([3.0, 2.0, 0.5]))]            # far off: many features without partners
    try:
        for k in (1, 2):
            for q, t in poses:
                q = q / np.linalg.norm(q)
                gpu_ctx.set_option(gpu_ctx.OPT_CORR_TILE, 3)
                ref = batch.correspond(k, q, t)
                for mode in (0, 3):          # 0: 32 lanes per feature on the hash grid, 3: flattened sweeps (default)
                    gpu_ctx.set_option(gpu_ctx.OPT_CORR_TILE, mode)
                    gpu_ctx.timing_reset()
                    got = batch.correspond(k, q, t)
                    deferred = gpu_ctx.timing()[0]["deferred_features"]
                    print("scan %d t=%s mode %d: %d features, %d deferred to the list kernel" % (k, t, mode, len(ref), deferred))
                    assert np.array_equal(got, ref)
        # a batch registered under the default mode has no hash grids; the deferred-list kernel then searches through the line index alone
        gpu_ctx.set_option(gpu_ctx.OPT_CORR_TILE, 3)
        i0, p0 = batch.odometry(1, 0)
        for mode in (0, 3):
            gpu_ctx.set_option(gpu_ctx.OPT_CORR_TILE, mode)
            i1, p1 = batch.odometry(1, 0)
            assert np.array_equal(i0, i1) and np.array_equal(p0, p1)
    finally:
        gpu_ctx.set_option(gpu_ctx.OPT_CORR_TILE, 3)
    # the fall-back of the default search on a batch WITHOUT hash grids (registered under mode 3): every 5th feature is forced through
    # k_correspond_list, which then finds the nearest point by the arc sweep alone; same indices, same odometry
    fresh = _register(gpu_ctx, xyzi, off)
    try:
        gpu_ctx.set_option(gpu_ctx.OPT_DEFER_EVERY, 5)
        for q, t in poses[:3]:
            q = q / np.linalg.norm(q)
            gpu_ctx.timing_reset()
            got = fresh.correspond(2, q, t)
            assert gpu_ctx.timing()[0]["deferred_features"] >= len(got) // 5
            gpu_ctx.set_option(gpu_ctx.OPT_DEFER_EVERY, 0)
            assert np.array_equal(got, fresh.correspond(2, q, t))
            gpu_ctx.set_option(gpu_ctx.OPT_DEFER_EVERY, 5)
        i5, p5 = fresh.odometry(1, 0)
        assert np.array_equal(i5, i0) and np.array_equal(p5, p0)
    finally:
        gpu_ctx.set_option(gpu_ctx.OPT_DEFER_EVERY, 0)
    # against the oracle as well (index-exact), at full resolution
    f = [oracle.scanreg(xyzi[off[s]:off[s + 1]]) for s in range(3)]
    q, t = poses[1]
    q = q / np.linalg.norm(q)
    _, _, _, corr = oracle.odom_step(f[2]["sharp"], f[2]["flat"], f[1]["less_sharp"], f[1]["less_flat"], q, t, want_corr=True)
    assert np.array_equal(batch.correspond(2, q, t), corr[0])




@pytest.mark.parametrize("n_lines,min_range", [(16, 0.5), (32, 0.5), (64, 0.5)])
def test_odometry_other_sensors_and_near_points(oracle, gpu_ctx, n_lines, min_range):
    """The (scan line, azimuth bin) search on 16- / 32-line sensors and with points from 0.5 m on (feature points close to the sensor
    axis: their search balls cover every azimuth): correspondences of every search mode and the sequential odometry against the oracle."""
    w = oracle.S1World(n_az=600, n_rings=n_lines)
    xyzi, off = w.scans(w.trajectory(5))
    batch = _register(gpu_ctx, xyzi, off, n_lines, min_range)
    f = [oracle.scanreg(xyzi[off[s]:off[s + 1]], n_lines, min_range) for s in range(3)]
    q = np.array([0.0, 0.0, 0.008, 1.0]); q /= np.linalg.norm(q)
    t = np.array([0.75, -0.01, 0.0])
    for k in (1, 2):
        _, _, _, corr = oracle.odom_step(f[k]["sharp"], f[k]["flat"], f[k - 1]["less_sharp"], f[k - 1]["less_flat"], q, t, want_corr=True)
        try:
            for mode in (3, 0):
                gpu_ctx.set_option(gpu_ctx.OPT_CORR_TILE, mode)
                assert np.array_equal(batch.correspond(k, q, t), corr[0]), "mode %d, scan %d" % (mode, k)
        finally:
            gpu_ctx.set_option(gpu_ctx.OPT_CORR_TILE, 3)
    incr, poses = batch.odometry(1, 0)
    ref = oracle.run_sequence(xyzi, off, n_lines, min_range)
    assert np.abs(incr - ref["incr"]).max() < 1e-9 and np.abs(poses - ref["poses"]).max() < 1e-8


