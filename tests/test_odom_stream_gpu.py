"""Online laserOdometry (lmono_odom_stream_*, lmono_odom_step): one scan per call with the previous scan's feature
clouds and search index kept on the device -- what A-LOAM's nodes do per ROS callback -- against the batch entry point
run over the same scans with the strictly sequential schedule: the same increments bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scans(oracle):
    w = oracle.S1World(n_az=500)
    xyzi, off = w.scans(w.trajectory(24))
    return xyzi, off


def test_one_scan_per_call_equals_the_batch_bit_for_bit(gpu_ctx, scans):
    import torch
    import lmono_amd
    xyzi, off = scans
    n = len(off) - 1
    dev = torch.from_numpy(xyzi).cuda()
    batch = lmono_amd.ScanBatch(gpu_ctx, n, len(xyzi))
    batch.scanreg(dev.data_ptr(), off, 64, 5.0, keepalive=dev)
    ref_incr, ref_poses = batch.odometry(1, 0)
    ref_cnt = batch.counts()
    cap = int(np.diff(off).max()) + 1000
    # history 5: the six slots wrap four times over 24 scans (the "last" data move from the last slot to slot 0)
    for history, on_device in ((5, False), (30, True)):
        st = lmono_amd.OdomStream(gpu_ctx, cap, 64, 5.0, history=history)
        for k in range(n):
            a, e = int(off[k]), int(off[k + 1])
            if on_device:
                incr, pose, info = st.step(dev_ptr=dev.data_ptr() + a * 16, n_points=e - a)
            else:
                incr, pose, info = st.step(xyzi[a:e])
            assert (info[:6] == ref_cnt[k]).all(), "scan %d: counts / status differ" % k
            assert np.array_equal(incr, ref_incr[k]), "scan %d: increment differs from the batch run" % k
            assert np.abs(pose - ref_poses[k]).max() < 1e-11          # host accumulation vs the device's parallel prefix
            if k in (0, 7, n - 1):
                for which in (1, 2, 3, 4):
                    assert np.array_equal(st.cloud(which, e - a), batch.cloud(k, which, e - a))
        st.close()


def test_warm_start_override_and_limits(gpu_ctx, scans):
    import lmono_amd
    xyzi, off = scans
    st = lmono_amd.OdomStream(gpu_ctx, int(np.diff(off).max()), 64, 5.0, history=2)
    i0, p0, _ = st.step(xyzi[off[0]:off[1]])
    assert np.array_equal(i0, [0, 0, 0, 1, 0, 0, 0]) and np.array_equal(p0, i0)         # the first scan has no pair
    i1, _, info = st.step(xyzi[off[1]:off[2]])
    assert 0.5 < i1[4] < 1.1 and info[6] > 0 and info[7] > 100                          # ~0.8 m forward; LM ran on > 100 residual blocks
    # a caller-supplied warm start replaces para_q / para_t for that pair: far off -> another local solution path, still a valid pose
    i2, _, _ = st.step(xyzi[off[2]:off[3]], warm_start=[0, 0, 0, 1, 0.8, 0, 0])
    assert abs(np.linalg.norm(i2[:4]) - 1) < 1e-12 and 0.5 < i2[4] < 1.1
    with pytest.raises(lmono_amd.LmonoError):
        st.step(np.zeros((int(np.diff(off).max()) + 1, 4), np.float32))               # more points than a slot holds
    # an empty scan is registered (no features) and the next pair finds no correspondences: the increment stays the warm start
    ie, _, info = st.step(np.zeros((0, 4), np.float32))
    assert info[0] == 0 and info[7] == 0
    st.close()
    with pytest.raises(lmono_amd.LmonoError):
        lmono_amd.OdomStream(gpu_ctx, 1000, 48)
