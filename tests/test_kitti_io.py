"""On-disk formats either side of the hot path: KITTI velodyne .bin / times / poses readers and the
reference's trajectory / timing-log printf formats, C++ (lmono_amd/host/kitti_io.cpp) against numpy (lmono_amd/kitti_io.py)."""
import os
import subprocess

import numpy as np
import pytest

from lmono_amd import kitti_io as IO

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "lmono_amd", "host")


def _make_sequence(tmp_path, n=3, seed=5):
    rng = np.random.default_rng(seed)
    seq = tmp_path / "seq"
    (seq / "velodyne").mkdir(parents=True)
    clouds = []
    for k in range(n):
        c = rng.normal(0, 20, (100 + 17 * k, 4)).astype(np.float32)
        IO.write_velodyne_bin(IO.velodyne_path(str(seq), k), c)
        clouds.append(c)
    stamps = 1317617735.0 + 0.1037 * np.arange(n)
    np.savetxt(seq / "times.txt", stamps, fmt="%.9f")
    poses = rng.normal(0, 5, (n, 12))
    np.savetxt(tmp_path / "poses.txt", poses, fmt="%.9e")
    return str(seq), str(tmp_path / "poses.txt"), clouds, stamps, poses


def test_numpy_round_trip_and_line_formats(tmp_path):
    seq, poses_file, clouds, stamps, poses = _make_sequence(tmp_path)
    xyzi, off, st = IO.load_scans(seq)
    assert list(np.diff(off)) == [len(c) for c in clouds]
    assert np.array_equal(xyzi, np.concatenate(clouds)) and np.allclose(st, stamps, atol=1e-6)
    assert np.allclose(IO.read_kitti_poses(poses_file).reshape(-1, 12), poses, rtol=1e-9)
    # printf("%f") semantics of the reference: six decimals, loam_odometry with a blank before the newline
    assert IO.format_pose_line(1.5, [1, -2.25, 3], [0, 0, 0.5, 1]) == "1.500000 1.000000 -2.250000 3.000000 0.000000 0.000000 0.500000 1.000000\n"
    assert IO.format_pose_line(0.1, [0, 0, 0], [0, 0, 0, 1], loam_style=True).endswith("1.000000 \n")
    assert IO.format_timing_line(2.0, 0.001, 0.002, 0.5) == "2.000000 0.001000 0.002000 0.500000\n"
    with pytest.raises(ValueError):
        (tmp_path / "bad.bin").write_bytes(b"\0" * 20)
        IO.read_velodyne_bin(str(tmp_path / "bad.bin"))
    # trajectory file: written from [q, t] rows, read back as stamp x y z qx qy qz qw
    qt = np.array([[0, 0, 0, 1, 1, 2, 3], [0, 0, 0.1, 0.99, 4, 5, 6]], float)
    IO.write_trajectory(str(tmp_path / "t.txt"), [0.0, 0.1], qt, loam_style=True)
    back = IO.read_trajectory(str(tmp_path / "t.txt"))
    assert np.allclose(back[:, 1:4], qt[:, 4:7]) and np.allclose(back[:, 4:8], qt[:, 0:4])


def test_cpp_reader_and_writers_match_numpy(tmp_path):
    subprocess.check_call(["make", "-s", "-C", HOST, "io_test"])
    seq, poses_file, clouds, stamps, poses = _make_sequence(tmp_path)
    out = tmp_path / "out"
    out.mkdir()
    txt = subprocess.check_output([os.path.join(HOST, "io_test"), seq, poses_file, str(out)], text=True).split("\n")
    assert txt[0] == "STAMPS %d POSES %d" % (len(stamps), len(poses))
    for k, c in enumerate(clouds):
        tag, idx, n, s = txt[1 + k].split()
        assert tag == "SCAN" and int(idx) == k and int(n) == len(c)
        assert abs(float(s) - float(c.astype(np.float64).sum())) <= 1e-6 * max(1.0, abs(float(c.astype(np.float64).sum())))
    assert txt[1 + len(clouds)] == "BAD -1"
    st = IO.read_times(os.path.join(seq, "times.txt"))
    P = IO.read_kitti_poses(poses_file)
    exp_new, exp_loam, exp_t = "", "", ""
    for k in range(len(st)):
        q = [0.0, 0.0, np.sin(0.05 * k), np.cos(0.05 * k)]
        exp_new += IO.format_pose_line(st[k], P[k][:, 3], q)
        exp_loam += IO.format_pose_line(st[k], P[k][:, 3], q, loam_style=True)
        exp_t += IO.format_timing_line(st[k], 0.001 * k, 0.002, 0.5 + k)
    assert (out / "new_odometry.txt").read_text() == exp_new
    assert (out / "loam_odometry.txt").read_text() == exp_loam
    assert (out / "times_recorder.txt").read_text() == exp_t


def test_ply_colour_map_format(tmp_path):
    """rgb_map<index>.ply (MapBuilder::processMapping, Map_Builder.cc:72-77): the C++ and the Python writer emit the same
    bytes, the PCL-style header is in place, and both readers return the points."""
    subprocess.check_call(["make", "-s", "-C", HOST, "io_test"])
    seq, poses_file, clouds, stamps, poses = _make_sequence(tmp_path)
    rng = np.random.default_rng(2)
    pts = np.zeros(1000, IO.POINT_RGB)
    pts["x"], pts["y"], pts["z"] = rng.normal(0, 30, (3, 1000)).astype(np.float32)
    pts["bgra"] = rng.integers(0, 1 << 24, 1000).astype(np.uint32) | 0xff000000
    (tmp_path / "pts.bin").write_bytes(pts.tobytes())
    out = tmp_path / "out2"
    out.mkdir()
    txt = subprocess.check_output([os.path.join(HOST, "io_test"), seq, poses_file, str(out), str(tmp_path / "pts.bin")], text=True)
    assert "PLY 1000 1000" in txt
    IO.write_ply_binary(str(tmp_path / "py.ply"), pts)
    raw = (out / "rgb_map10.ply").read_bytes()
    assert raw == (tmp_path / "py.ply").read_bytes()
    assert IO.rgb_map_path(str(out), 10) == str(out / "rgb_map10.ply")
    head = raw[:raw.index(b"end_header\n")].decode().split("\n")
    assert head[:4] == ["ply", "format binary_little_endian 1.0", "comment PCL generated", "element vertex 1000"]
    assert head[4:10] == ["property float x", "property float y", "property float z", "property uchar red", "property uchar green", "property uchar blue"]
    assert head[10] == "element camera 1" and len(head) == 33 and head[-1] == ""
    assert len(raw) == raw.index(b"end_header\n") + 11 + 1000 * 15 + 21 * 4
    assert (IO.read_ply_binary(str(out / "rgb_map10.ply")) == pts).all()
    assert (out / "mapping_recorder.txt").read_text() == IO.format_mapping_line(1.5, 2.25) + IO.format_mapping_line(2.5, 0.125)
    empty = np.zeros(0, IO.POINT_RGB)
    IO.write_ply_binary(str(tmp_path / "e.ply"), empty)
    assert len(IO.read_ply_binary(str(tmp_path / "e.ply"))) == 0
