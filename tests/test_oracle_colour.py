"""CPU checks of the colour-projection oracle (oracle/lo_colour.c).  OpenCV is absent, so the image
filters are checked against scipy.ndimage / numpy restatements of the same documented definitions (a second, independent
implementation -- not a pin against OpenCV itself: parity stays 'unpinned' for those stages)."""
import numpy as np
import pytest
import scipy.ndimage as ndi

from oracle import oracle as O
from tests import colour_cases as CC


def _img(h, w, seed, density=0.15):
    rng = np.random.default_rng(seed)
    a = rng.integers(1, 101, (h, w)).astype(np.uint8)
    a[rng.random((h, w)) > density] = 0
    return a


def test_structuring_elements():
    assert (O.structuring_element(0, 5) == 1).all()
    cross = np.zeros((5, 5), np.uint8); cross[2, :] = 1; cross[:, 2] = 1
    assert (O.structuring_element(1, 5) == cross).all()
    ell = np.ones((5, 5), np.uint8); ell[0] = ell[4] = [0, 0, 1, 0, 0]     # cv::getStructuringElement(MORPH_ELLIPSE, 5x5)
    assert (O.structuring_element(2, 5) == ell).all()
    ell7 = O.structuring_element(2, 7)
    assert ell7[0].tolist() == [0, 0, 0, 1, 0, 0, 0] and ell7[3].all() and (ell7 == ell7[::-1]).all() and (ell7 == ell7[:, ::-1]).all()


@pytest.mark.parametrize("kind,k", [(0, 5), (1, 5), (2, 5), (0, 3), (2, 7)])
def test_morphology_matches_scipy(kind, k):
    a = _img(37, 53, 5)
    m = O.structuring_element(kind, k)
    assert (O.morph(a, m, 0) == ndi.grey_dilation(a, footprint=m.astype(bool), mode="constant", cval=0)).all()
    assert (O.morph(a, m, 1) == ndi.grey_erosion(a, footprint=m.astype(bool), mode="constant", cval=255)).all()


def test_median_and_gauss_match_scipy():
    a = _img(41, 67, 7, density=0.6)
    assert (O.median5(a) == ndi.median_filter(a, size=5, mode="nearest")).all()
    k = np.array([1, 4, 6, 4, 1], np.int64)
    s = ndi.convolve(a.astype(np.int64), np.outer(k, k), mode="mirror")
    assert (O.gauss5(a) == ((s + 128) >> 8).astype(np.uint8)).all()


def test_bilateral_matches_numpy_restatement():
    a = _img(33, 45, 9, density=0.7)
    h, w = a.shape
    pad = np.pad(a, 2, mode="reflect").astype(np.int32)
    cw = np.exp(np.arange(256, dtype=np.float64) ** 2 * (-0.5 / (1.5 * 1.5))).astype(np.float32)
    s = np.zeros((h, w), np.float32); ws = np.zeros((h, w), np.float32)
    for i in range(-2, 3):
        for j in range(-2, 3):
            r = np.sqrt(float(i * i + j * j))
            if r > 2:
                continue
            sw = np.float32(np.exp(r * r * (-0.5 / (2.0 * 2.0))))
            v = pad[2 + i:2 + i + h, 2 + j:2 + j + w]
            wt = sw * cw[np.abs(v - a.astype(np.int32))]
            s = s + v.astype(np.float32) * wt
            ws = ws + wt
    assert (O.bilateral5(a) == np.rint(s / ws).astype(np.uint8)).all()
    flat = np.full((9, 9), 42, np.uint8)
    assert (O.bilateral5(flat) == 42).all() and (O.gauss5(flat) == 42).all() and (O.median5(flat) == 42).all()


def test_pinhole_round_trip_and_splat():
    cam = O.kitti00_cam()
    p = O.space_to_plane(cam, [2.0, -1.0, 10.0])
    assert np.allclose(p, [718.856 * 0.2 + 607.1928, 718.856 * -0.1 + 185.2157], rtol=0, atol=1e-12)
    assert np.allclose(O.lift_projective(cam, *p), [0.2, -0.1, 1.0], atol=1e-15)
    camd = O.kitti00_cam(dist=(-1.5855983900634696e-01, 1.2994555880814793e-01, -6.0424265983630317e-04, 9.1268093157433972e-04))
    pd = O.space_to_plane(camd, [2.0, -1.0, 10.0])
    assert np.abs(pd - p).max() > 0.5                      # the distortion moves the pixel ...
    assert np.allclose(O.lift_projective(camd, *pd), [0.2, -0.1, 1.0], atol=1e-9)   # ... and the recursive model undoes it
    # splat: one point in front of the camera -> pixel (floor v, floor u) = trunc(100 - z); the later point of a pixel wins
    M = np.eye(4)
    pts = np.array([[2.0, -1.0, 10.0, 0], [2.0, -1.0, 30.5, 0], [0.0, 0.0, -5.0, 0], [0.0, 0.0, 104.3, 0]], np.float32)
    d = O.depth_splat(cam, pts[:1], M)
    assert d[int(p[1]), int(p[0])] == 90 and d.sum() == 90
    d = O.depth_splat(cam, pts, M)
    assert d[185, 607] == (-4) % 256                        # 100 - 104.3 -> -4 -> wraps (x86 double -> int -> uchar)
    q = O.space_to_plane(cam, [2.0, -1.0, 30.5])
    assert d[int(q[1]), int(q[0])] == 69
    assert np.count_nonzero(d) == 3                         # the point behind the camera is skipped


def test_associate_to_map_properties():
    cam = O.kitti00_cam()
    sc = CC.s1_scan(n_az=1000)
    M = CC.lidar_to_camera()
    bgr = CC.noise_image(cam.height, cam.width)
    d, a, b = O.associate_to_map(cam, sc, M, bgr, [0, 0, 0, 1.0], [1.0, 2.0, 3.0])
    assert len(a) == len(b) > 50000
    # every point lifts back to its pixel, in row-major order, with that pixel's colour and the filled depth
    u = a["x"].astype(np.float64) / a["z"] * cam.fx + cam.cx; v = a["y"].astype(np.float64) / a["z"] * cam.fy + cam.cy
    ui = np.rint(u).astype(int); vi = np.rint(v).astype(int)
    assert np.abs(u - ui).max() < 1e-3 and np.abs(v - vi).max() < 1e-3
    flat = vi * cam.width + ui
    assert (np.diff(flat) > 0).all()
    assert (100 - d[vi, ui].astype(int) == a["z"]).all() and a["z"].min() >= 1 and a["z"].max() <= 69
    px = bgr[vi, ui].astype(np.uint32)
    assert (a["bgra"] == (px[:, 0] | px[:, 1] << 8 | px[:, 2] << 16 | 0xff000000)).all()
    assert not ((np.abs(a["x"]) > 20) & (a["y"] > 1.8)).any()
    # identity rotation: world = camera + T (float arithmetic through double)
    assert np.array_equal(b["x"], (a["x"].astype(np.float64) + 1.0).astype(np.float32))
    assert np.array_equal(b["z"], (a["z"].astype(np.float64) + 3.0).astype(np.float32)) and (a["bgra"] == b["bgra"]).all()
    # stages compose: splat -> fill -> back-project
    d0 = O.depth_splat(cam, sc, M)
    assert (O.depth_fill(cam, d0) == d).all()
    assert np.array_equal(O.backproject(cam, d, bgr), a)
    # the hole filling only ever adds support
    assert (d[d0 > 0] > 0).all() and np.count_nonzero(d) > 5 * np.count_nonzero(d0)
