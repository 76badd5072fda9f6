"""Seeded random inputs for the BA factor tests (shared by the CPU and GPU suites and by the golden generator)."""
import numpy as np


def rand_quat(rng, scale=0.3):
    q = np.concatenate([rng.normal(0, scale, 3), [1.0]])
    return q / np.linalg.norm(q)


def quat_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def rand_pose(rng, tscale=2.0, rscale=0.2):
    return np.concatenate([rng.normal(0, tscale, 3), rand_quat(rng, rscale)])


# laser_to_camera0 of kitti_config_05.yaml (mono_lidar_mapping/config/kitti_config_05.yaml:27-30)
def kitti_extrinsic():
    T = np.eye(4)
    T[:3, :3] = np.array([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]])   # camera z forward -> lidar x forward
    T[:3, 3] = [0.27, 0.0, -0.08]
    return T


def laser_cases(n, seed=1):
    rng = np.random.default_rng(seed)
    P = np.zeros((n, 14)); Cn = np.zeros((n, 24))
    for k in range(n):
        pi = rand_pose(rng); pj = pi.copy()
        pj[:3] += rng.normal([0.8, 0, 0], 0.2); pj[3:] = rand_quat(rng, 0.2)
        if k % 5 == 0:     # slightly non-unit quaternions as Ceres sees them mid-iteration never happens (Plus normalises); keep unit
            pass
        P[k] = np.concatenate([pi, pj])
        Ri = quat_R(rand_quat(rng)); Rj = quat_R(rand_quat(rng))
        if k == 1:         # trace <= 0 branch of the matrix -> quaternion conversion
            Ri = np.eye(3); Rj = np.diag([1.0, -1.0, -1.0])
        Cn[k] = np.concatenate([Ri.ravel(), Rj.ravel(), rng.normal(0, 3, 3), rng.normal(0, 3, 3)])
    if n > 0:              # identity case: zero residual
        P[0] = np.concatenate([[0, 0, 0, 0, 0, 0, 1.0], [0, 0, 0, 0, 0, 0, 1.0]])
        Cn[0] = np.concatenate([np.eye(3).ravel(), np.eye(3).ravel(), np.zeros(6)])
    return P, Cn, (3.0 * 1500.0) * np.eye(6)      # LASER_W * FACTOR_WEIGHT (kitti_config_05)


def mono_cases(n, seed=2):
    rng = np.random.default_rng(seed)
    T = kitti_extrinsic()
    from_R = T[:3, :3]
    P = np.zeros((n, 22)); Cn = np.zeros((n, 4))
    for k in range(n):
        # extrinsic near the KITTI one
        qx = _R_to_q(from_R @ quat_R(rand_quat(rng, 0.02)))
        ex = np.concatenate([T[:3, 3] + rng.normal(0, 0.02, 3), qx])
        pi = rand_pose(rng, 1.0, 0.05); pj = pi.copy(); pj[:3] += rng.normal([0.8, 0, 0], 0.3); pj[3:] = rand_quat(rng, 0.05)
        depth = rng.uniform(4, 40)
        pt_i = rng.uniform(-0.6, 0.6, 2) * [1.0, 0.3]
        # project the landmark into frame j to get a consistent pt_j, then add pixel noise
        Rx = quat_R(qx); Ri = quat_R(pi[3:]); Rj = quat_R(pj[3:])
        pw = Ri @ (Rx @ (depth * np.array([pt_i[0], pt_i[1], 1.0])) + ex[:3]) + pi[:3]
        pc = Rx.T @ (Rj.T @ (pw - pj[:3]) - ex[:3])
        pt_j = pc[:2] / pc[2] + rng.normal(0, 0.5 / 707.0, 2)
        P[k] = np.concatenate([ex, pi, pj, [1.0 / depth]])
        Cn[k] = np.concatenate([pt_i, pt_j])
    return P, Cn, 1500.0 * np.eye(2)


def _R_to_q(m):
    from oracle import ba_numpy as B
    return B.R_to_q(m)


def prior_cases(n, seed=3):
    rng = np.random.default_rng(seed)
    P = np.zeros((n, 7)); Cn = np.zeros((n, 16))
    for k in range(n):
        T = kitti_extrinsic()
        T[:3, :3] = T[:3, :3] @ quat_R(rand_quat(rng, 0.05)); T[:3, 3] += rng.normal(0, 0.05, 3)
        ex = np.concatenate([T[:3, 3] + rng.normal(0, 0.02, 3), _R_to_q(T[:3, :3] @ quat_R(rand_quat(rng, 0.02)))])
        P[k] = ex; Cn[k] = T.ravel()
    return P, Cn, np.array([1000.0, 1000.0])


def reproj_cases(n, seed=4):
    rng = np.random.default_rng(seed)
    Pm, Cm, _ = mono_cases(n, seed)
    P = np.zeros((n, 1)); Cn = np.zeros((n, 44))
    for k in range(n):
        ex, pi, pj = Pm[k, :7], Pm[k, 7:14], Pm[k, 14:21]
        EX = np.eye(4); EX[:3, :3] = quat_R(ex[3:]); EX[:3, 3] = ex[:3]
        P[k, 0] = Pm[k, 21] * rng.uniform(0.7, 1.3)
        Cn[k] = np.concatenate([Cm[k], quat_R(pi[3:]).ravel(), pi[:3], quat_R(pj[3:]).ravel(), pj[:3], EX.ravel()])
    return P, Cn, np.array([1500.0])


# --------------------------------------------------------------------------------------------
# Synthetic sliding windows "S2" (SURVEY.md 8d, config 3): KITTI-05 intrinsics / extrinsic / weights


# the S2 window generator is input plumbing shared with bench.py: workloads/s2.py
from workloads.s2 import make_window, FX, FY, CX, CY, W_IMG, H_IMG  # noqa: E402,F401
