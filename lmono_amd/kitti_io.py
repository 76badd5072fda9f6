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
