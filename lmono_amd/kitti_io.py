"""On-disk formats either side of the hot path, numpy flavour of lmono_amd/host/kitti_io.{hpp,cpp}.

in : KITTI odometry layout -- <seq>/velodyne/%06d.bin (float32 x y z reflectance), <seq>/times.txt, poses/XX.txt
     (12 numbers per line: row-major 3x4 [R | t])                                              (upstream kittiHelper)
out: trajectory lines "stamp x y z qx qy qz qw" exactly as the reference prints them
     (mono_lidar_mapping/src/image_process/Estimator.cc:270-271 "loam_odometry" with its trailing blank, :642-643
     "new_odometry") and the timing log "stamp track_time laser_decode_time pred_time" (Estimator.cc:647)."""
import os

import numpy as np


def velodyne_path(sequence_dir, index):
    return os.path.join(sequence_dir, "velodyne", "%06d.bin" % index)


def read_velodyne_bin(path):
    """[n,4] float32 x y z reflectance; raises ValueError on a file that is not a whole number of 16-byte records."""
    raw = np.fromfile(path, dtype=np.float32)
    if raw.size % 4 != 0 or os.path.getsize(path) % 16 != 0:
        raise ValueError("%s: not a whole number of float32 x y z reflectance records" % path)
    return raw.reshape(-1, 4)


def write_velodyne_bin(path, xyzi):
    np.ascontiguousarray(xyzi, np.float32).tofile(path)


def read_times(path):
    return np.loadtxt(path, dtype=np.float64, ndmin=1)


def read_kitti_poses(path):
    """[n,3,4] float64."""
    return np.loadtxt(path, dtype=np.float64, ndmin=2).reshape(-1, 3, 4)


def load_scans(sequence_dir, first=0, count=None):
    """Concatenated scans of a sequence: (xyzi [total,4] float32, offsets int64 [n+1], stamps [n])."""
    stamps = read_times(os.path.join(sequence_dir, "times.txt"))
    last = len(stamps) if count is None else min(len(stamps), first + count)
    clouds = [read_velodyne_bin(velodyne_path(sequence_dir, k)) for k in range(first, last)]
    offsets = np.zeros(len(clouds) + 1, np.int64)
    offsets[1:] = np.cumsum([len(c) for c in clouds])
    xyzi = np.concatenate(clouds) if clouds else np.zeros((0, 4), np.float32)
    return xyzi, offsets, stamps[first:last]


def format_pose_line(stamp, p, q_xyzw, loam_style=False):
    """One trajectory line in the reference's printf format ("%f" = 6 decimals); loam_style adds the trailing blank of
    Estimator.cc:270."""
    line = "%f %f %f %f %f %f %f %f" % (stamp, p[0], p[1], p[2], q_xyzw[0], q_xyzw[1], q_xyzw[2], q_xyzw[3])
    return line + (" \n" if loam_style else "\n")


def write_trajectory(path, stamps, poses_qt, loam_style=False):
    """poses_qt: [n,7] = qx qy qz qw tx ty tz (the layout of lmono_odom_batch's poses)."""
    with open(path, "w") as f:
        for t, row in zip(stamps, poses_qt):
            f.write(format_pose_line(t, row[4:7], row[0:4], loam_style))


def read_trajectory(path):
    """[n,8] = stamp x y z qx qy qz qw."""
    return np.loadtxt(path, dtype=np.float64, ndmin=2)


def format_timing_line(stamp, track_time, laser_decode_time, pred_time):
    return "%f %f %f %f\n" % (stamp, track_time, laser_decode_time, pred_time)


POINT_RGB = np.dtype([("x", np.float32), ("y", np.float32), ("z", np.float32), ("bgra", np.uint32)])
_PLY_CAMERA = ["view_px", "view_py", "view_pz", "x_axisx", "x_axisy", "x_axisz", "y_axisx", "y_axisy", "y_axisz",
               "z_axisx", "z_axisy", "z_axisz", "focal", "scalex", "scaley", "centerx", "centery"]
_PLY_VERTEX = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])


def rgb_map_path(directory, map_index):
    """Map_Builder.cc:75: "<dir>/rgb_map<index>.ply"."""
    return os.path.join(directory, "rgb_map%d.ply" % map_index)


def write_ply_binary(path, pts):
    """pcl::io::savePLYFileBinary of an unorganised PointXYZRGB cloud (Map_Builder.cc:76): vertex x y z float + red green
    blue uchar, then PCL's one-record camera element.  pts: POINT_RGB records (bgra = b | g << 8 | r << 16 | a << 24)."""
    pts = np.ascontiguousarray(pts, POINT_RGB)
    n = len(pts)
    head = "ply\nformat binary_little_endian 1.0\ncomment PCL generated\nelement vertex %d\n" % n
    head += "property float x\nproperty float y\nproperty float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\n"
    head += "element camera 1\n" + "".join("property float %s\n" % p for p in _PLY_CAMERA)
    head += "property int viewportx\nproperty int viewporty\nproperty float k1\nproperty float k2\nend_header\n"
    v = np.zeros(n, _PLY_VERTEX)
    v["x"], v["y"], v["z"] = pts["x"], pts["y"], pts["z"]
    v["red"] = (pts["bgra"] >> 16) & 0xff; v["green"] = (pts["bgra"] >> 8) & 0xff; v["blue"] = pts["bgra"] & 0xff
    with open(path, "wb") as f:
        f.write(head.encode("ascii"))
        f.write(v.tobytes())
        f.write(np.array([0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 0, 0], "<f4").tobytes())
        f.write(np.array([n, 1], "<i4").tobytes())
        f.write(np.array([0, 0], "<f4").tobytes())


def read_ply_binary(path):
    with open(path, "rb") as f:
        raw = f.read()
    end = raw.index(b"end_header\n") + len(b"end_header\n")
    head = raw[:end].decode("ascii").split("\n")
    if "format binary_little_endian 1.0" not in head:
        raise ValueError("not a binary little-endian PLY: " + path)
    n = [int(l.split()[2]) for l in head if l.startswith("element vertex")][0]
    v = np.frombuffer(raw, _PLY_VERTEX, n, end)
    out = np.zeros(n, POINT_RGB)
    out["x"], out["y"], out["z"] = v["x"], v["y"], v["z"]
    out["bgra"] = v["blue"].astype(np.uint32) | v["green"].astype(np.uint32) << 8 | v["red"].astype(np.uint32) << 16 | 0xff000000
    return out


def format_mapping_line(stamp, toc_ms):
    """map_build_node.cc:230: fprintf(mapping_recorder, "%f %f \\n", stamp, toc)."""
    return "%f %f \n" % (stamp, toc_ms)
