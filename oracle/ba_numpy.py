"""oracle/ba_numpy.py -- TEST INFRASTRUCTURE.  numpy restatement of lmono's BA factors, written independently of
oracle/lo_ba.c (matrix form here, scalar form there); agreement of the two to <= 1e-12 relative is the first gate
(SURVEY.md 8c).  PARITY UNPINNED: the reference has no tests or golden vectors and cannot be built here.

Every function cites the reference lines it restates (paths relative to /root/reference/mono_lidar_mapping).
Bug-compatible on purpose (SURVEY.md 8a notes): the LASERFactor rotation Jacobians take the bottom-right 3x3 of
vec-first 4x4 quaternion matrices, MonoProjectionFactor Frobenius-normalises a 3x3 product.
Quaternions are (x, y, z, w) in parameter blocks, exactly like para_pose (Estimator.cc:1029-1036)."""
import numpy as np


# ---- Eigen conventions -------------------------------------------------------------------------
def q_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx,
                     aw * bw - ax * bx - ay * by - az * bz])


def q_inv(q):
    """Eigen QuaternionBase::inverse(): conjugate / squaredNorm."""
    n2 = q @ q
    return np.array([-q[0], -q[1], -q[2], q[3]]) / n2


def q_rot(q, v):
    """Eigen QuaternionBase::_transformVector (valid as a rotation only for unit q; used as-is, like the reference)."""
    u = q[:3]
    uv = 2.0 * np.cross(u, v)
    return v + q[3] * uv + np.cross(u, uv)


def q_to_R(q):
    """Eigen QuaternionBase::toRotationMatrix()."""
    x, y, z, w = q
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return np.array([[1 - (tyy + tzz), txy - twz, txz + twy],
                     [txy + twz, 1 - (txx + tzz), tyz - twx],
                     [txz - twy, tyz + twx, 1 - (txx + tyy)]])


def q_normalized(q):
    return q / np.sqrt(q @ q)


def R_to_q(m):
    """Eigen Quaternion(Matrix3) constructor (internal::quaternionbase_assign_impl<Other,3,3>)."""
    t = m[0, 0] + m[1, 1] + m[2, 2]
    q = np.zeros(4)
    if t > 0:
        t = np.sqrt(t + 1.0)
        q[3] = 0.5 * t
        t = 0.5 / t
        q[0] = (m[2, 1] - m[1, 2]) * t
        q[1] = (m[0, 2] - m[2, 0]) * t
        q[2] = (m[1, 0] - m[0, 1]) * t
    else:
        i = 0
        if m[1, 1] > m[0, 0]:
            i = 1
        if m[2, 2] > m[i, i]:
            i = 2
        j = (i + 1) % 3
        k = (j + 1) % 3
        t = np.sqrt(m[i, i] - m[j, j] - m[k, k] + 1.0)
        q[i] = 0.5 * t
        t = 0.5 / t
        q[3] = (m[k, j] - m[j, k]) * t
        q[j] = (m[j, i] + m[i, j]) * t
        q[k] = (m[k, i] + m[i, k]) * t
    return q


def skew(v):
    """include/utils/math_utils.h:131-137"""
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def left_quat_matrix(q):
    """include/utils/math_utils.h:151-160 (vec-first 4x4)"""
    m = np.zeros((4, 4))
    m[:3, :3] = q[3] * np.eye(3) + skew(q[:3])
    m[3, :3] = -q[:3]
    m[:3, 3] = q[:3]
    m[3, 3] = q[3]
    return m


def right_quat_matrix(p):
    """include/utils/math_utils.h:163-172"""
    m = np.zeros((4, 4))
    m[:3, :3] = p[3] * np.eye(3) - skew(p[:3])
    m[3, :3] = -p[:3]
    m[:3, 3] = p[:3]
    m[3, 3] = p[3]
    return m


# ---- factors -----------------------------------------------------------------------------------
def laser_factor(pose_i, pose_j, L0_Ri, L0_Rj, L0_Pi, L0_Pj, sqrt_info):
    """LASERFactor ctor + Evaluate, include/factor/LaserFactor.h:29-100.  Returns r[6], J_i[6,7], J_j[6,7]."""
    delta_ij = R_to_q(L0_Ri.T @ L0_Rj)
    delta_pij = L0_Ri.T @ (L0_Pj - L0_Pi)
    Qi, Qj = pose_i[3:7], pose_j[3:7]
    Pi, Pj = pose_i[:3], pose_j[:3]
    Qi_inv = q_inv(Qi)
    r = np.zeros(6)
    r[:3] = q_rot(Qi_inv, Pj - Pi) - delta_pij
    r[3:] = 2 * q_mul(q_inv(delta_ij), q_mul(Qi_inv, Qj))[:3]
    r = sqrt_info @ r
    Ji = np.zeros((6, 7))
    Ji[:3, :3] = -q_to_R(Qi_inv)
    Ji[:3, 3:6] = skew(q_rot(Qi_inv, Pj - Pi))
    Ji[3:, 3:6] = -(left_quat_matrix(q_mul(q_inv(Qj), Qi)) @ right_quat_matrix(delta_ij))[1:, 1:]
    Ji = sqrt_info @ Ji
    Jj = np.zeros((6, 7))
    Jj[:3, :3] = q_to_R(Qi_inv)
    Jj[3:, 3:6] = left_quat_matrix(q_mul(q_mul(q_inv(delta_ij), Qi_inv), Qj))[1:, 1:]
    Jj = sqrt_info @ Jj
    return r, Ji, Jj


def mono_projection_factor(ex, pose_i, pose_j, inv_depth, pt_i, pt_j, sqrt_info):
    """MonoProjectionFactor::Evaluate, src/factor/MonoProjectionFactor.cc:40-174.
    Returns r[2], J_ex[2,7], J_i[2,7], J_j[2,7], J_depth[2]."""
    Qx, tx = ex[3:7], ex[:3]
    ti, tj = pose_i[:3], pose_j[:3]
    Qi, Qj = pose_i[3:7], pose_j[3:7]
    depth = 1.0 / inv_depth
    Ri = q_to_R(q_normalized(Qi))
    Rj = q_to_R(q_normalized(Qj))
    Rlc = q_to_R(q_normalized(Qx))
    p_i = np.array([pt_i[0], pt_i[1], 1.0])
    p_j = np.array([pt_j[0], pt_j[1], 1.0])
    pts_ci = depth * p_i
    pts_laser_i = q_rot(Qx, pts_ci) + tx
    pts_w = q_rot(Qi, pts_laser_i) + ti
    pt_l_j = q_rot(q_inv(Qj), pts_w - tj)
    pts_cj = q_rot(q_inv(Qx), pt_l_j - tx)
    dep_cj = pts_cj[2]
    r = sqrt_info @ ((pts_cj / dep_cj)[:2] - p_j[:2])
    reduce = np.array([[1.0 / dep_cj, 0, -pts_cj[0] / (dep_cj * dep_cj)],
                       [0, 1.0 / dep_cj, -pts_cj[1] / (dep_cj * dep_cj)]])
    reduce = sqrt_info @ reduce
    # extrinsic block (:118-134)
    RjtRi = Rj.T @ Ri
    fro = np.sqrt((RjtRi * RjtRi).sum())
    jaco = np.zeros((3, 6))
    jaco[:, :3] = Rlc.T @ (RjtRi / fro - np.eye(3))
    temp_r = Rlc.T @ Rj.T @ Ri @ Rlc
    jaco[:, 3:] = (-temp_r @ skew(pts_ci) + skew(temp_r @ pts_ci)
                   + skew(Rlc.T @ (Rj.T @ (Ri @ tx + ti - tj) - tx)))
    J_ex = np.zeros((2, 7)); J_ex[:, :6] = reduce @ jaco
    # pose i (:136-147)
    jaco_i = np.zeros((3, 6))
    jaco_i[:, :3] = Rlc.T @ Rj.T
    jaco_i[:, 3:] = Rlc.T @ Rj.T @ Ri @ (-skew(pts_laser_i))
    J_i = np.zeros((2, 7)); J_i[:, :6] = reduce @ jaco_i
    # pose j (:149-159)
    jaco_j = np.zeros((3, 6))
    jaco_j[:, :3] = -(Rlc.T @ Rj.T)
    jaco_j[:, 3:] = Rlc.T @ skew(pt_l_j)
    J_j = np.zeros((2, 7)); J_j[:, :6] = reduce @ jaco_j
    # inverse depth (:161-169)
    J_d = -(reduce @ Rlc.T @ Rj.T @ Ri @ Rlc @ p_i) * depth * depth
    return r, J_ex, J_i, J_j, J_d


def prior_factor(ex, transform, prior_t, prior_r):
    """PriorFactor ctor + Evaluate, include/factor/PriorFactor.h:29-69.  transform: 4x4.  Returns r[6], J[6,7]."""
    pos = transform[:3, 3]
    rot = R_to_q(transform[:3, :3])
    P, Q = ex[:3], ex[3:7]
    sqrt_info = np.diag([prior_t] * 3 + [prior_r] * 3)
    r = np.zeros(6)
    r[:3] = P - pos
    r[3:] = 2 * q_mul(q_inv(rot), Q)[:3]
    r = sqrt_info @ r
    jaco = np.eye(6)
    jaco[3:, 3:] = left_quat_matrix(q_mul(q_inv(Q), rot))[:3, :3]
    J = np.zeros((6, 7)); J[:, :6] = sqrt_info @ jaco
    return r, J


def reprojection_factor(inv_dep, pt_i, pt_j, Ri, Pi, Rj, Pj, EX, factor_weight):
    """ReprojectionFactor ctor + Evaluate, include/factor/ReprojectionFactor.h:16-78.  Returns r[2], J[2]."""
    Rlc, Tlc = EX[:3, :3], EX[:3, 3]
    sqrt_info = factor_weight * np.eye(2)
    p_i = np.array([pt_i[0], pt_i[1], 1.0]); p_j = np.array([pt_j[0], pt_j[1], 1.0])
    dep = 1.0 / inv_dep
    pts_ci = dep * p_i
    pts_li = Rlc @ pts_ci + Tlc
    pts_w = Ri @ pts_li + Pi
    pts_lj = Rj.T @ (pts_w - Pj)
    pts_cj = Rlc.T @ (pts_lj - Tlc)
    dep_cj = pts_cj[2]
    r = sqrt_info @ ((pts_cj / dep_cj)[:2] - p_j[:2])
    reduce = sqrt_info @ np.array([[1.0 / dep_cj, 0, -pts_cj[0] / (dep_cj * dep_cj)],
                                   [0, 1.0 / dep_cj, -pts_cj[1] / (dep_cj * dep_cj)]])
    J = -(reduce @ Rlc.T @ Rj.T @ Ri @ Rlc @ p_i) * dep * dep
    return r, J


def pose_plus(x, delta):
    """PoseLocalParameterization::Plus, src/factor/PoseLocalParameterization.cc:15-31; DeltaQ math_utils.h:117-128."""
    out = np.zeros(7)
    out[:3] = x[:3] + delta[:3]
    dq = np.array([delta[3] / 2.0, delta[4] / 2.0, delta[5] / 2.0, 1.0])
    out[3:] = q_normalized(q_mul(x[3:7], dq))
    return out


def cauchy(s, a=1.0):
    """ceres::CauchyLoss(a)::Evaluate."""
    b = a * a
    c = 1.0 / b
    u = 1.0 + s * c
    inv = 1.0 / u
    return np.array([b * np.log(u), max(np.finfo(float).tiny, inv), -c * (inv * inv)])


def corrector(r, Js, rho):
    """ResidualBlockInfo::Evaluate robust correction, src/factor/MarginalizationFactor.cc:34-66 (== ceres::Corrector)."""
    sq_norm = r @ r
    sqrt_rho1 = np.sqrt(rho[1])
    if sq_norm == 0.0 or rho[2] <= 0.0:
        residual_scaling, alpha_sq_norm = sqrt_rho1, 0.0
    else:
        D = 1.0 + 2.0 * sq_norm * rho[2] / rho[1]
        alpha = 1.0 - np.sqrt(D)
        residual_scaling = sqrt_rho1 / (1 - alpha)
        alpha_sq_norm = alpha / sq_norm
    Js2 = [sqrt_rho1 * (J - alpha_sq_norm * np.outer(r, r @ J)) for J in Js]
    return r * residual_scaling, Js2
