"""Trajectory error metrics on laserOdometry pose arrays ([n,7]: q xyzw, t -- the layout of q_w_curr / t_w_curr).

ATE here is the RMS translation difference of two trajectories expressed in the same frame (both start at the identity at
scan 0; no alignment step, so a rotation error early in the sequence is charged with its full lever arm).  RPE is the
relative pose error over a fixed scan distance delta:  E_k = (Q_k^-1 Q_{k+delta})^-1 (P_k^-1 P_{k+delta}),  reported as the
RMS of |trans(E_k)| (m) and of the rotation angle of E_k (deg).  Evaluation plumbing used by bench.py and the tests; numpy only.
"""
import numpy as np


def _qmul(a, b):
    ax, ay, az, aw = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bx, by, bz, bw = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx,
                     aw * bw - ax * bx - ay * by - az * bz], -1)


def _qconj(q):
    return q * np.array([-1.0, -1.0, -1.0, 1.0])


def _qrot(q, v):
    u = q[..., :3]
    w = q[..., 3:4]
    uv = 2.0 * np.cross(u, v)
    return v + w * uv + np.cross(u, uv)


def relative(poses, delta=1):
    """T_k^-1 T_{k+delta} for k = 0 .. n-delta-1 -> [n-delta, 7]."""
    p = np.asarray(poses, np.float64)
    a, b = p[:-delta], p[delta:]
    qi = _qconj(a[:, :4])
    return np.concatenate([_qmul(qi, b[:, :4]), _qrot(qi, b[:, 4:] - a[:, 4:])], 1)


def ate(est, ref):
    """RMS of |t_est - t_ref| (m) over all poses (same frame, no alignment)."""
    d = np.asarray(est, np.float64)[:, 4:7] - np.asarray(ref, np.float64)[:, 4:7]
    return float(np.sqrt((d ** 2).sum(1).mean()))


def ate_sums(est, ref):
    """(sum of squared translation differences, count): partial sums a rank contributes to a sharded ATE."""
    d = np.asarray(est, np.float64)[:, 4:7] - np.asarray(ref, np.float64)[:, 4:7]
    return float((d ** 2).sum()), int(len(d))


def rpe_from_relative(rel_est, rel_ref):
    """-> (sum |trans E|^2, sum angle(E)^2 [rad^2], count) for already-relative poses (e.g. odometry increments, delta = 1)."""
    e = np.asarray(rel_est, np.float64)
    r = np.asarray(rel_ref, np.float64)
    ri = _qconj(r[:, :4])
    dq = _qmul(ri, e[:, :4])
    dt = _qrot(ri, e[:, 4:] - r[:, 4:])
    ang = 2.0 * np.arctan2(np.linalg.norm(dq[:, :3], axis=1), np.abs(dq[:, 3]))
    return float((dt ** 2).sum()), float((ang ** 2).sum()), int(len(e))


def rpe(est, ref, delta=1):
    """-> dict(trans_rmse_m, rot_rmse_deg, delta, pairs) over all pairs (k, k + delta)."""
    if len(est) <= delta:
        return dict(trans_rmse_m=0.0, rot_rmse_deg=0.0, delta=delta, pairs=0)
    st, sr, n = rpe_from_relative(relative(est, delta), relative(ref, delta))
    return dict(trans_rmse_m=float(np.sqrt(st / n)), rot_rmse_deg=float(np.degrees(np.sqrt(sr / n))), delta=delta, pairs=n)
