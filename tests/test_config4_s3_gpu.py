"""BASELINE configs[4] at its named shape ("VLP-32 synthetic 10 Hz stream + fusion mapping colour projection"): S3
scans -- 32 rings x 1800 azimuth steps -- through scanRegistration (32 lines), sequential laserOdometry, the online per-scan entry and
the map builder's colour projection at the odometry's poses, every stage against the CPU oracle on the same scans."""
import numpy as np
import pytest

from tests import colour_cases as CC

pytestmark = pytest.mark.gpu
N = 10


@pytest.fixture(scope="module")
def s3(oracle):
    w = oracle.S1World(n_rings=32, n_az=1800)
    traj = w.trajectory(N)
    xyzi, off = w.scans(traj)
    return dict(xyzi=xyzi, off=off, traj=traj, ref=oracle.run_sequence(xyzi, off, n_lines=32, min_range=0.5))


def test_scanreg_and_odometry_at_32x1800(oracle, gpu_ctx, s3):
    import torch
    import lmono_amd
    xyzi, off, ref = s3["xyzi"], s3["off"], s3["ref"]
    dev = torch.from_numpy(xyzi).cuda()
    b = lmono_amd.ScanBatch(gpu_ctx, N, len(xyzi))
    b.scanreg(dev.data_ptr(), off, 32, 0.5, keepalive=dev)
    cnt = b.counts()
    assert (cnt[:, 5] == 0).all() and (cnt[:, 1:5] == ref["feat_counts"]).all()
    # feature clouds of one scan bit for bit (the whole-front-end check at 600 azimuth steps is tests/test_lidar_gpu.py)
    f = oracle.scanreg(xyzi[off[3]:off[4]], n_lines=32, min_range=0.5)
    n3 = int(off[4] - off[3])
    for which, key in ((1, "sharp"), (2, "less_sharp"), (3, "flat"), (4, "less_flat")):
        assert np.array_equal(b.cloud(3, which, n3), f[key]), key
    incr, poses = b.odometry(1, 0)
    assert np.abs(incr - ref["incr"]).max() < 1e-9 and np.abs(poses - ref["poses"]).max() < 1e-8
    assert 0.6 < incr[5, 4] < 1.0                                   # ~0.8 m forward per scan
    # the same scans one call at a time (10 Hz stream): same increments bit for bit
    st = lmono_amd.OdomStream(gpu_ctx, int(np.diff(off).max()), 32, 0.5, history=4)
    for k in range(N):
        i1, p1, info = st.step(xyzi[off[k]:off[k + 1]])
        assert np.array_equal(i1, incr[k]) and (info[1:5] == ref["feat_counts"][k]).all()
    st.close()
    s3["gpu_poses"] = poses


def test_colour_projection_of_the_stream_at_the_odometry_poses(oracle, gpu_ctx, s3):
    """MapBuilder::associateToMap + depthFill + accumulation over the S3 stream (1241 x 376 images, kitti_map_config_00.yaml settings):
    every frame's filled depth map and coloured clouds and the accumulated rgb_map equal the oracle's, byte for byte."""
    import lmono_amd
    xyzi, off = s3["xyzi"], s3["off"]
    poses = s3.get("gpu_poses", s3["ref"]["poses"])
    oc = oracle.kitti00_cam(1241, 376, 5, 0, 0, (0.0, 0.0, 0.0, 0.0))
    gc = lmono_amd.Camera(1241, 376, oc.fx, oc.fy, oc.cx, oc.cy, 0, 0, 0, 0, 5, 0, 0)
    M = CC.lidar_to_camera()
    mb = lmono_amd.MapBuilder(gpu_ctx, gc, max_cloud_points=int(np.diff(off).max()))
    acc = []
    total = 0
    for k in range(N):
        cloud = xyzi[off[k]:off[k + 1]]
        bgr = CC.noise_image(376, 1241, seed=10 + k)
        # camera pose in the camera-aligned world from the LiDAR odometry pose (map_build_node.cc:216-225 convention: T_wc = T_wl * T_lc)
        q_l, t_l = poses[k, :4], poses[k, 4:]
        n = mb.associate(cloud, M, bgr, q_l, t_l)
        d, a, b = oracle.associate_to_map(oc, cloud, M, bgr, q_l, t_l)
        assert (mb.depth() == d).all(), "frame %d: depth map" % k
        assert n == len(a) and mb.cloud(0).tobytes() == a.tobytes() and mb.cloud(1).tobytes() == b.tobytes(), "frame %d" % k
        acc.append(b); total += n
    assert total > 100000
    assert mb.map().tobytes() == np.concatenate(acc, 0).tobytes()
    mb.close()
