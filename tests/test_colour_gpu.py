"""Colour projection on the GPU (lmono_associate_to_map*) against the CPU oracle on the same scans and
images: the filled depth map, the camera-frame cloud and the world-frame cloud are bit-exact (byte / index work; the
double -> float conversions of the transforms are deterministic with contraction off on both sides)."""
import numpy as np
import pytest

from tests import colour_cases as CC

pytestmark = pytest.mark.gpu

DIST = (-1.5855983900634696e-01, 1.2994555880814793e-01, -6.0424265983630317e-04, 9.1268093157433972e-04)   # config/hk_cam00.yaml


def _cam(oracle, w=1241, h=376, k=5, kind=0, blur=0, dist=(0.0, 0.0, 0.0, 0.0), scale=1.0):
    import lmono_amd
    oc = oracle.kitti00_cam(w, h, k, kind, blur, dist)
    oc.fx *= scale; oc.fy *= scale; oc.cx *= scale; oc.cy *= scale
    gc = lmono_amd.Camera(w, h, oc.fx, oc.fy, oc.cx, oc.cy, dist[0], dist[1], dist[2], dist[3], k, kind, blur)
    return oc, gc


def _check(oracle, ctx, oc, gc, cloud, bgr, q, t, M=None):
    import lmono_amd
    M = CC.lidar_to_camera() if M is None else M
    mb = lmono_amd.MapBuilder(ctx, gc, max_cloud_points=max(len(cloud), 1))
    n = mb.associate(cloud, M, bgr, q, t)
    d, a, b = oracle.associate_to_map(oc, cloud, M, bgr, q, t)
    gd = mb.depth()
    assert (gd == d).all(), "depth map differs at %d pixels" % np.count_nonzero(gd != d)
    assert n == len(a)
    ga, gb = mb.cloud(0), mb.cloud(1)
    assert ga.tobytes() == a.tobytes() and gb.tobytes() == b.tobytes()
    assert mb.map().tobytes() == b.tobytes()
    mb.close()
    return n


@pytest.mark.parametrize("kind,k,blur", [(0, 5, 0), (1, 5, 0), (2, 5, 1), (0, 3, 1), (2, 7, 0), (0, 1, 0), (1, 9, 1), (2, 11, 0)])
def test_associate_matches_oracle_kitti_image(oracle, gpu_ctx, kind, k, blur):
    """The reference configuration (kitti_map_config_00.yaml: FULL / 5 / bilateral) and every other element / blur branch."""
    oc, gc = _cam(oracle, k=k, kind=kind, blur=blur)
    cloud = CC.s1_scan(n_az=1000)
    bgr = CC.noise_image(oc.height, oc.width)
    q = np.array([0.02, -0.03, 0.38, 0.0]); q[3] = np.sqrt(1 - (q[:3] ** 2).sum())
    n = _check(oracle, gpu_ctx, oc, gc, cloud, bgr, q, [12.5, -3.25, 0.75])
    assert n > 50000


def test_distortion_and_far_points(oracle, gpu_ctx):
    """Radial-tangential intrinsics (recursive lift) and a random cloud with points behind the camera and beyond 100 m
    (where the 8-bit depth wraps)."""
    oc, gc = _cam(oracle, dist=DIST)
    cloud = CC.random_cloud(150000)
    bgr = CC.noise_image(oc.height, oc.width, seed=4)
    q = np.array([0.0, 0.0, 0.0, 1.0])
    _check(oracle, gpu_ctx, oc, gc, cloud, bgr, q, [0.0, 0.0, 0.0])


@pytest.mark.parametrize("w,h", [(97, 53), (64, 32), (65, 33), (8, 8), (130, 9)])
def test_ragged_image_sizes(oracle, gpu_ctx, w, h):
    """Tile-edge cases of the fused fill kernel: images that are not multiples of the 64 x 32 tile, smaller than a tile or than the halo."""
    oc, gc = _cam(oracle, w=w, h=h, scale=w / 1241.0)
    oc.cy = gc.cy = h / 2.0
    cloud = CC.random_cloud(20000, seed=w * 1000 + h, zmax=80.0)
    bgr = CC.noise_image(h, w, seed=w)
    _check(oracle, gpu_ctx, oc, gc, cloud, bgr, [0.0, 0.0, 0.0, 1.0], [1.0, 2.0, 3.0])


def test_empty_cloud_and_dense_image(oracle, gpu_ctx):
    oc, gc = _cam(oracle, w=200, h=100, scale=200 / 1241.0)
    oc.cy = gc.cy = 50.0
    bgr = CC.noise_image(100, 200)
    assert _check(oracle, gpu_ctx, oc, gc, np.zeros((0, 4), np.float32), bgr, [0, 0, 0, 1.0], [0, 0, 0.0]) == 0
    # a wall 10 m ahead covering every pixel several times: each pixel keeps its last point
    rng = np.random.default_rng(0)
    n = 200000
    cloud = np.zeros((n, 4), np.float32)
    cloud[:, 0] = rng.uniform(8.0, 60.0, n); cloud[:, 1] = rng.uniform(-1, 1, n) * cloud[:, 0] * 0.9; cloud[:, 2] = rng.uniform(-1, 1, n) * cloud[:, 0] * 0.5
    assert _check(oracle, gpu_ctx, oc, gc, cloud, bgr, [0, 0, 0, 1.0], [0, 0, 0.0]) > 15000


def test_accumulation_clear_batch_and_capacity(oracle, gpu_ctx):
    """processMapping's accumulation (rgb_map += cloud; clear) over frames, the batched entry against the single one, and the
    loud failure when rgb_map is full."""
    import torch
    import lmono_amd
    oc, gc = _cam(oracle, w=320, h=96, scale=320 / 1241.0)
    oc.cy = gc.cy = 48.0
    M = CC.lidar_to_camera()
    frames = []
    for k in range(3):
        cloud = CC.s1_scan(n_rings=32, n_az=900, k=k)
        bgr = CC.noise_image(96, 320, seed=10 + k)
        q = np.array([0.0, 0.0, np.sin(0.1 * k), np.cos(0.1 * k)]); t = np.array([0.8 * k, 0.1 * k, 0.0])
        frames.append((cloud, bgr, q, t, oracle.associate_to_map(oc, cloud, M, bgr, q, t)))
    mb = lmono_amd.MapBuilder(gpu_ctx, gc, max_cloud_points=1 << 16, map_capacity_points=3 * 320 * 96)
    for cloud, bgr, q, t, ref in frames[:2]:
        assert mb.associate(cloud, M, bgr, q, t) == len(ref[1])
    assert mb.map().tobytes() == np.concatenate([frames[0][4][2], frames[1][4][2]]).tobytes()
    assert mb.cloud(1).tobytes() == frames[1][4][2].tobytes()
    mb.clear()
    assert len(mb.map()) == 0
    with pytest.raises(lmono_amd.LmonoError):
        mb.cloud(1)                               # the last world cloud left with the map
    # batched entry: three builders, device-resident inputs, one call
    mbs = [lmono_amd.MapBuilder(gpu_ctx, gc, max_cloud_points=16, map_capacity_points=320 * 96) for _ in range(3)]
    dc = [torch.from_numpy(f[0]).to("cuda:0") for f in frames]; di = [torch.from_numpy(f[1]).to("cuda:0") for f in frames]
    n = lmono_amd.MapBuilder.associate_batch(gpu_ctx, mbs, [x.data_ptr() for x in dc], [len(f[0]) for f in frames], [M] * 3,
                                             [x.data_ptr() for x in di], [f[2] for f in frames], [f[3] for f in frames])
    for s in range(3):
        d, a, b = frames[s][4]
        assert n[s] == len(a) and (mbs[s].depth() == d).all()
        assert mbs[s].cloud(0).tobytes() == a.tobytes() and mbs[s].map().tobytes() == b.tobytes()
    # a second frame does not fit a one-image rgb_map
    with pytest.raises(lmono_amd.LmonoError):
        mbs[0].associate(frames[0][0][:16], M, frames[0][1], frames[0][2], frames[0][3])
    with pytest.raises(lmono_amd.LmonoError):
        lmono_amd.MapBuilder(gpu_ctx, lmono_amd.Camera(320, 96, 100.0, 100.0, 160.0, 48.0, 0, 0, 0, 0, 4, 0, 0))    # even kernel
    for m in mbs + [mb]:
        m.close()


def test_cpp_map_build_writes_the_colour_map(oracle, tmp_path):
    """Host mirror MapBuilder (associateToMap + processMapping) over a sequence on disk through lmono_amd/host/map_build:
    rgb_map10.ply holds the world clouds of frames 0..9 (then the map is cleared), the last frame's products match the oracle."""
    import os
    import subprocess
    from lmono_amd import kitti_io as IO
    host = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lmono_amd", "host")
    subprocess.check_call(["make", "-s", "-C", host, "map_build"])
    W, H = 320, 96
    oc = oracle.kitti00_cam(W, H)
    oc.fx = oc.fy = 185.0; oc.cx = 156.5; oc.cy = 47.25
    seq = tmp_path / "seq"; out = tmp_path / "out"
    os.makedirs(seq / "velodyne"); os.makedirs(seq / "image_bgr"); os.makedirs(out)
    n = 11
    M = CC.lidar_to_camera()
    rng = np.random.default_rng(8)
    stamps = 50.0 + 0.1 * np.arange(n)
    qt = np.zeros((n, 7)); acc = []; ref_last = None
    for k in range(n):
        cloud = CC.s1_scan(n_rings=32, n_az=600, k=k)
        bgr = CC.noise_image(H, W, seed=100 + k)
        IO.write_velodyne_bin(IO.velodyne_path(str(seq), k), cloud)
        (seq / "image_bgr" / ("%06d.bgr" % k)).write_bytes(bgr.tobytes())
        # poses with exactly six decimals so that the text file carries them losslessly
        q = np.round(np.array([0.01 * k, -0.02, np.sin(0.05 * k), np.cos(0.05 * k)]), 6); t = np.round(rng.normal(0, 5, 3), 6)
        qt[k, :4] = q; qt[k, 4:] = t
        ref = oracle.associate_to_map(oc, cloud, M, bgr, q, t)
        if k < 10:
            acc.append(ref[2])
        ref_last = ref
    IO.write_trajectory(str(tmp_path / "traj.txt"), stamps, qt)
    txt = subprocess.check_output([os.path.join(host, "map_build"), str(seq), str(tmp_path / "traj.txt"), str(out), str(n), str(W), str(H),
                                   repr(oc.fx), repr(oc.fy), repr(oc.cx), repr(oc.cy)], text=True).strip().split("\n")
    assert len(txt) == n and txt[9].endswith("wrote " + IO.rgb_map_path(str(out), 10)) and "wrote" not in txt[10]
    ply = IO.read_ply_binary(IO.rgb_map_path(str(out), 10))
    want = np.concatenate(acc)
    assert ply.tobytes() == want.tobytes()
    d, a, b = ref_last
    assert (np.frombuffer((out / "depth_last.u8").read_bytes(), np.uint8).reshape(H, W) == d).all()
    assert (out / "cloud_cam_last.bin").read_bytes() == a.tobytes() and (out / "cloud_world_last.bin").read_bytes() == b.tobytes()
    rec = (out / "mapping_recorder.txt").read_text().split("\n")
    assert len(rec) == n + 1 and all(line.endswith(" ") and line.startswith("%f" % stamps[k]) for k, line in enumerate(rec[:n]))
