"""Shared inputs of the colour-projection tests: an S1/S3 scan, a noise image, the KITTI-like rig."""
import numpy as np

RLC = np.array([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]])   # camera axes (x right, y down, z forward) in the LiDAR frame
TLC = np.array([0.27, 0.0, -0.08])


def lidar_to_camera(rlc=RLC, tlc=TLC):
    M = np.eye(4)
    M[:3, :3] = rlc.T
    M[:3, 3] = (-1.0 * rlc.T) @ tlc
    return M


def noise_image(h, w, seed=1):
    return np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)


def random_cloud(n, seed=3, zmax=125.0):
    """Points spread over the camera frustum and beyond it (behind the camera, beyond 100 m where the 8-bit depth wraps)."""
    rng = np.random.default_rng(seed)
    p = np.zeros((n, 4), np.float32)
    p[:, 0] = rng.uniform(-20.0, zmax, n)      # LiDAR x = camera z
    p[:, 1] = rng.uniform(-60.0, 60.0, n)
    p[:, 2] = rng.uniform(-6.0, 12.0, n)
    p[:, 3] = rng.uniform(0, 1, n)
    return p


def s1_scan(n_rings=64, n_az=2000, k=0):
    from workloads.s1 import S1World
    w = S1World(n_rings=n_rings, n_az=n_az)
    poses = w.trajectory(k + 1)
    xyzi, off = w.scans(poses[k:k + 1], scan_id0=k)
    return xyzi[off[0]:off[1]]
