"""workloads/s4.py -- synthetic loop-closure pose graphs "S4" (SURVEY row 8f-2 / config 4 shape): keyframes along the S1
figure-8 driven for several laps, odometry = the true relative motion with noise integrated (so it drifts), loop
constraints between revisits of the same place in the loop_info layout of KeyFrame::findConnection
(mono_lidar_mapping/src/loop_detection/KeyFrame.cc:570-633: relative_t, relative_q (w x y z), relative_yaw in degrees).
Input plumbing for tests and bench.py; neither the hot path nor the oracle."""
import numpy as np


def _rot(ypr_deg):
    y, p, r = np.deg2rad(ypr_deg)
    Rz = np.array([[np.cos(y), -np.sin(y), 0], [np.sin(y), np.cos(y), 0], [0, 0, 1]])
    Ry = np.array([[np.cos(p), 0, np.sin(p)], [0, 1, 0], [-np.sin(p), 0, np.cos(p)]])
    Rx = np.array([[1, 0, 0], [0, np.cos(r), -np.sin(r)], [0, np.sin(r), np.cos(r)]])
    return Rz @ Ry @ Rx


def _quat(R):
    """Eigen::Quaterniond(Matrix3d) -> (x, y, z, w)."""
    tr = np.trace(R)
    if tr > 0:
        t = np.sqrt(tr + 1.0); w = 0.5 * t; t = 0.5 / t
        return np.array([(R[2, 1] - R[1, 2]) * t, (R[0, 2] - R[2, 0]) * t, (R[1, 0] - R[0, 1]) * t, w])
    i = int(np.argmax(np.diag(R))); j = (i + 1) % 3; k = (j + 1) % 3
    t = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
    q = np.zeros(4); q[i] = 0.5 * t; t = 0.5 / t
    q[3] = (R[k, j] - R[j, k]) * t; q[j] = (R[j, i] + R[i, j]) * t; q[k] = (R[k, i] + R[i, k]) * t
    return q


def _yaw_deg(R):
    return np.rad2deg(np.arctan2(R[1, 0], R[0, 0]))


def make_graph(n=400, laps=2.0, seed=7, sigma_t=0.02, sigma_yaw_deg=0.05, sigma_tilt_deg=0.2, loop_radius=3.0, loop_gap=50,
               loop_every=5, loop_noise_t=0.02, loop_noise_yaw_deg=0.05, outliers=0):
    """-> dict(truth [n,7], odom [n,7] (t, q xyzw), loops [L,2] int32 (old, current), loop_info [L,8])."""
    rng = np.random.default_rng(seed)
    a = 60.0
    u = np.linspace(0.0, 2.0 * np.pi * laps, n, endpoint=False)
    x = a * np.sin(u); y = a * np.sin(u) * np.cos(u) * 0.9
    dx = a * np.cos(u); dy = a * 0.9 * (np.cos(u) ** 2 - np.sin(u) ** 2)
    yaw = np.rad2deg(np.unwrap(np.arctan2(dy, dx)))
    z = 0.5 * np.sin(3.0 * u)
    tilt = rng.normal(0.0, sigma_tilt_deg, (n, 2))                       # small pitch / roll, observable and kept fixed by the graph
    Rt = [_rot([yaw[i], tilt[i, 0], tilt[i, 1]]) for i in range(n)]
    tt = np.stack([x, y, z], 1)
    truth = np.zeros((n, 7)); odom = np.zeros((n, 7))
    Ro, to = Rt[0].copy(), tt[0].copy()
    for i in range(n):
        if i > 0:
            dR = Rt[i - 1].T @ Rt[i]; dt = Rt[i - 1].T @ (tt[i] - tt[i - 1])
            dR = dR @ _rot([rng.normal(0, sigma_yaw_deg), 0.0, 0.0])     # yaw drift; pitch / roll stay observable (gravity)
            dt = dt + rng.normal(0, sigma_t, 3)
            to = to + Ro @ dt
            Ro = Ro @ dR
        truth[i, :3] = tt[i]; truth[i, 3:] = _quat(Rt[i])
        odom[i, :3] = to; odom[i, 3:] = _quat(Ro)
    loops, info = [], []
    for j in range(loop_gap, n, loop_every):
        d = np.linalg.norm(tt[:j - loop_gap + 1] - tt[j], axis=1)
        i = int(np.argmin(d))
        if d[i] > loop_radius:
            continue
        rel_t = Rt[i].T @ (tt[j] - tt[i]) + rng.normal(0, loop_noise_t, 3)
        rel_R = Rt[i].T @ Rt[j]
        q = _quat(rel_R)
        rel_yaw = _yaw_deg(Rt[j]) - _yaw_deg(Rt[i]) + rng.normal(0, loop_noise_yaw_deg)
        rel_yaw = (rel_yaw + 180.0) % 360.0 - 180.0
        loops.append((i, j)); info.append([rel_t[0], rel_t[1], rel_t[2], q[3], q[0], q[1], q[2], rel_yaw])
    for k in range(min(outliers, len(loops))):                            # gross false positives for the robust loss
        info[(k * 7919) % len(loops)][0] += 25.0
    return dict(truth=truth, odom=odom, loops=np.array(loops, np.int32).reshape(-1, 2), loop_info=np.array(info, np.float64).reshape(-1, 8))


def ate(est, truth):
    """RMS position error after aligning the first keyframe (both sequences share keyframe 0 by construction)."""
    return float(np.sqrt(np.mean(np.sum((est[:, :3] - truth[:, :3]) ** 2, 1))))
