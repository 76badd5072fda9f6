// lmono_amd/csrc/ba.hip -- gfx950 kernels for lmono's sliding-window BA factors (fp64), bug-compatible with
// /root/reference/mono_lidar_mapping:
//   include/factor/LaserFactor.h:29-100         LASERFactor ctor + Evaluate          -> laser_factor
//   src/factor/MonoProjectionFactor.cc:40-174   MonoProjectionFactor::Evaluate       -> mono_factor
//   include/factor/PriorFactor.h:29-69          PriorFactor ctor + Evaluate          -> prior_factor
//   include/factor/ReprojectionFactor.h:16-78   ReprojectionFactor ctor + Evaluate   -> reproj_factor
//   include/utils/math_utils.h:117-172          DeltaQ, SkewSymmetric, Left/RightQuatMatrix
// The known quirks of the reference Jacobians are reproduced, not corrected.
// k_factor_eval: one thread per residual block, packed layouts of include/lmono_hip.h (Ceres cost-function layout:
// residuals, then row-major Jacobians of the global block size with a zero 7th pose column).
#include "common.hpp"

namespace lmono {
namespace ba {

__device__ __forceinline__ void q_mul(const double *a, const double *b, double *o)
{
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    o[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ void q_inv(const double *q, double *o)
{
    const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    o[0] = -q[0] / n2; o[1] = -q[1] / n2; o[2] = -q[2] / n2; o[3] = q[3] / n2;
}
__device__ __forceinline__ void q_rot(const double *q, const double *v, double *o)
{
    double uvx = q[1] * v[2] - q[2] * v[1], uvy = q[2] * v[0] - q[0] * v[2], uvz = q[0] * v[1] - q[1] * v[0];
    uvx += uvx; uvy += uvy; uvz += uvz;
    const double o0 = v[0] + q[3] * uvx + (q[1] * uvz - q[2] * uvy);
    const double o1 = v[1] + q[3] * uvy + (q[2] * uvx - q[0] * uvz);
    const double o2 = v[2] + q[3] * uvz + (q[0] * uvy - q[1] * uvx);
    o[0] = o0; o[1] = o1; o[2] = o2;
}
__device__ __forceinline__ void q_norm(const double *q, double *o)
{
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    o[0] = q[0] / n; o[1] = q[1] / n; o[2] = q[2] / n; o[3] = q[3] / n;
}
__device__ __forceinline__ void q_to_R(const double *q, double *R)
{
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
__device__ __forceinline__ void R_to_q(const double *m, double *q)
{
    double t = m[0] + m[4] + m[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (m[7] - m[5]) * t; q[1] = (m[2] - m[6]) * t; q[2] = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[i * 3 + i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0);
        double qq[4];
        qq[i] = 0.5 * t;
        t = 0.5 / t;
        qq[3] = (m[k * 3 + j] - m[j * 3 + k]) * t;
        qq[j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
        qq[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
        q[0] = qq[0]; q[1] = qq[1]; q[2] = qq[2]; q[3] = qq[3];
    }
}
__device__ __forceinline__ void mm(const double *A, const double *B, double *C)
{
    double T[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) T[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
#pragma unroll
    for (int k = 0; k < 9; k++) C[k] = T[k];
}
__device__ __forceinline__ void mT(const double *A, double *T)
{
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) T[i * 3 + j] = A[j * 3 + i];
}
__device__ __forceinline__ void mv(const double *A, const double *v, double *o)
{
    const double t0 = A[0] * v[0] + A[1] * v[1] + A[2] * v[2], t1 = A[3] * v[0] + A[4] * v[1] + A[5] * v[2], t2 = A[6] * v[0] + A[7] * v[1] + A[8] * v[2];
    o[0] = t0; o[1] = t1; o[2] = t2;
}
__device__ __forceinline__ void skew(const double *v, double *S)
{
    S[0] = 0; S[1] = -v[2]; S[2] = v[1]; S[3] = v[2]; S[4] = 0; S[5] = -v[0]; S[6] = -v[1]; S[7] = v[0]; S[8] = 0;
}
__device__ __forceinline__ void quat_matrix(const double *q, double sign, double *M)
{
    double S[9];
    skew(q, S);
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int j = 0; j < 3; j++) M[i * 4 + j] = (i == j ? q[3] : 0.0) + sign * S[i * 3 + j];
        M[12 + i] = -q[i];
        M[i * 4 + 3] = q[i];
    }
    M[15] = q[3];
}

// params[14] pose_i, pose_j; consts[24] L0_Ri, L0_Rj (row-major), L0_Pi, L0_Pj; info 6x6.  J: [2][6*7] or null.
__device__ __forceinline__ void laser_factor(const double *params, const double *consts, const double *info, double *r, double *J)
{
    const double *Ri0 = consts, *Rj0 = consts + 9, *Pi0 = consts + 18, *Pj0 = consts + 21;
    double RiT[9], Rrel[9], dq[4], dp[3], d[3];
    mT(Ri0, RiT); mm(RiT, Rj0, Rrel); R_to_q(Rrel, dq);
    _Pragma("unroll") for (int k = 0; k < 3; k++) d[k] = Pj0[k] - Pi0[k];
    mv(RiT, d, dp);
    const double *Pi = params, *Qi = params + 3, *Pj = params + 7, *Qj = params + 10;
    double Qi_inv[4], dPij[3], rp[3], dq_inv[4], qij[4], rq[4], res[6];
    q_inv(Qi, Qi_inv);
    _Pragma("unroll") for (int k = 0; k < 3; k++) dPij[k] = Pj[k] - Pi[k];
    q_rot(Qi_inv, dPij, rp);
    q_inv(dq, dq_inv); q_mul(Qi_inv, Qj, qij); q_mul(dq_inv, qij, rq);
    _Pragma("unroll") for (int k = 0; k < 3; k++) { res[k] = rp[k] - dp[k]; res[3 + k] = 2 * rq[k]; }
    _Pragma("unroll") for (int i = 0; i < 6; i++) { double s = 0; for (int k = 0; k < 6; k++) s += info[i * 6 + k] * res[k]; r[i] = s; }
    if (!J) return;
    double Rinv[9], S[9], Qj_inv[4], qa[4], L[16], Rm[16], qb[4], qc[4], L2[16];
    q_to_R(Qi_inv, Rinv); skew(rp, S);
    q_inv(Qj, Qj_inv); q_mul(Qj_inv, Qi, qa);
    quat_matrix(qa, 1.0, L); quat_matrix(dq, -1.0, Rm);
    q_mul(dq_inv, Qi_inv, qb); q_mul(qb, Qj, qc); quat_matrix(qc, 1.0, L2);
    double Ji[42], Jj[42];
    _Pragma("unroll") for (int k = 0; k < 42; k++) { Ji[k] = 0.0; Jj[k] = 0.0; }
    _Pragma("unroll") for (int i = 0; i < 3; i++)
        _Pragma("unroll") for (int j = 0; j < 3; j++) {
            double lr = 0;
            _Pragma("unroll") for (int k = 0; k < 4; k++) lr += L[(1 + i) * 4 + k] * Rm[k * 4 + 1 + j];
            Ji[i * 7 + j] = -Rinv[i * 3 + j];
            Ji[i * 7 + 3 + j] = S[i * 3 + j];
            Ji[(3 + i) * 7 + 3 + j] = -lr;
            Jj[i * 7 + j] = Rinv[i * 3 + j];
            Jj[(3 + i) * 7 + 3 + j] = L2[(1 + i) * 4 + 1 + j];
        }
    _Pragma("unroll") for (int i = 0; i < 6; i++)
        _Pragma("unroll") for (int j = 0; j < 7; j++) {
            double si = 0, sj = 0;
            _Pragma("unroll") for (int k = 0; k < 6; k++) { si += info[i * 6 + k] * Ji[k * 7 + j]; sj += info[i * 6 + k] * Jj[k * 7 + j]; }
            J[i * 7 + j] = si; J[42 + i * 7 + j] = sj;
        }
}

// params[22] ex, pose_i, pose_j, inv_depth; consts[4] pt_i.xy, pt_j.xy; info 2x2.  J: [44] or null.
__device__ void mono_factor(const double *params, const double *consts, const double *info, double *r, double *J)
{
    const double *tx = params, *Qx = params + 3, *ti = params + 7, *Qi = params + 10, *tj = params + 14, *Qj = params + 17;
    const double depth = 1.0 / params[21];
    const double p_i[3] = { consts[0], consts[1], 1.0 }, p_j[3] = { consts[2], consts[3], 1.0 };
    const double pts_ci[3] = { depth * p_i[0], depth * p_i[1], depth * p_i[2] };
    double pl[3], pw[3], plj[3], pcj[3], tmp[3], Qinv[4];
    q_rot(Qx, pts_ci, pl); for (int k = 0; k < 3; k++) pl[k] += tx[k];
    q_rot(Qi, pl, pw); for (int k = 0; k < 3; k++) pw[k] += ti[k];
    for (int k = 0; k < 3; k++) tmp[k] = pw[k] - tj[k];
    q_inv(Qj, Qinv); q_rot(Qinv, tmp, plj);
    for (int k = 0; k < 3; k++) tmp[k] = plj[k] - tx[k];
    q_inv(Qx, Qinv); q_rot(Qinv, tmp, pcj);
    const double dep = pcj[2];
    const double e0 = pcj[0] / dep - p_j[0], e1 = pcj[1] / dep - p_j[1];
    r[0] = info[0] * e0 + info[1] * e1;
    r[1] = info[2] * e0 + info[3] * e1;
    if (!J) return;
    double qn[4], Ri[9], Rj[9], Rlc[9];
    q_norm(Qi, qn); q_to_R(qn, Ri);
    q_norm(Qj, qn); q_to_R(qn, Rj);
    q_norm(Qx, qn); q_to_R(qn, Rlc);
    const double red0[6] = { 1.0 / dep, 0, -pcj[0] / (dep * dep), 0, 1.0 / dep, -pcj[1] / (dep * dep) };
    double red[6];
    for (int j = 0; j < 3; j++) { red[j] = info[0] * red0[j] + info[1] * red0[3 + j]; red[3 + j] = info[2] * red0[j] + info[3] * red0[3 + j]; }
    double RlcT[9], RjT[9], A[9], B[9], Cm[9], S[9], jaco[18];
    mT(Rlc, RlcT); mT(Rj, RjT);
    // extrinsic block (:118-134); (Rj^T Ri).normalized() is a Frobenius normalisation
    mm(RjT, Ri, A);
    double fro = 0;
    for (int k = 0; k < 9; k++) fro += A[k] * A[k];
    fro = sqrt(fro);
    for (int k = 0; k < 9; k++) B[k] = A[k] / fro - ((k % 4 == 0) ? 1.0 : 0.0);
    mm(RlcT, B, Cm);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) jaco[i * 6 + j] = Cm[i * 3 + j];
    double T[9], Tp[3], Sc[9], Sa[9], Sb[9], v[3], v2[3];
    mm(RlcT, RjT, T); mm(T, Ri, T); mm(T, Rlc, T);
    skew(pts_ci, Sc); mm(T, Sc, A);
    mv(T, pts_ci, Tp); skew(Tp, Sa);
    mv(Ri, tx, v); for (int k = 0; k < 3; k++) v[k] = v[k] + ti[k] - tj[k];
    mv(RjT, v, v2); for (int k = 0; k < 3; k++) v2[k] -= tx[k];
    mv(RlcT, v2, v); skew(v, Sb);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) jaco[i * 6 + 3 + j] = -A[i * 3 + j] + Sa[i * 3 + j] + Sb[i * 3 + j];
    for (int a = 0; a < 2; a++) { for (int j = 0; j < 6; j++) J[a * 7 + j] = red[a * 3] * jaco[j] + red[a * 3 + 1] * jaco[6 + j] + red[a * 3 + 2] * jaco[12 + j]; J[a * 7 + 6] = 0.0; }
    // pose i (:136-147)
    mm(RlcT, RjT, A);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) jaco[i * 6 + j] = A[i * 3 + j];
    mm(A, Ri, B); skew(pl, S);
    for (int k = 0; k < 9; k++) S[k] = -S[k];
    mm(B, S, Cm);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) jaco[i * 6 + 3 + j] = Cm[i * 3 + j];
    for (int a = 0; a < 2; a++) { for (int j = 0; j < 6; j++) J[14 + a * 7 + j] = red[a * 3] * jaco[j] + red[a * 3 + 1] * jaco[6 + j] + red[a * 3 + 2] * jaco[12 + j]; J[14 + a * 7 + 6] = 0.0; }
    // pose j (:149-159)
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) jaco[i * 6 + j] = -A[i * 3 + j];
    skew(plj, S); mm(RlcT, S, Cm);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) jaco[i * 6 + 3 + j] = Cm[i * 3 + j];
    for (int a = 0; a < 2; a++) { for (int j = 0; j < 6; j++) J[28 + a * 7 + j] = red[a * 3] * jaco[j] + red[a * 3 + 1] * jaco[6 + j] + red[a * 3 + 2] * jaco[12 + j]; J[28 + a * 7 + 6] = 0.0; }
    // inverse depth (:161-169)
    mv(T, p_i, v);
    for (int a = 0; a < 2; a++) J[42 + a] = -(red[a * 3] * v[0] + red[a * 3 + 1] * v[1] + red[a * 3 + 2] * v[2]) * depth * depth;
}

// params[7] ex; consts[16] 4x4 transform; info[2] = PRIOR_T, PRIOR_R.  J: [6*7] or null.
__device__ __forceinline__ void prior_factor(const double *params, const double *consts, const double *info, double *r, double *J)
{
    double Rm[9], rot[4], rinv[4], q[4], pos[3];
    _Pragma("unroll") for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) Rm[i * 3 + j] = consts[i * 4 + j]; pos[i] = consts[i * 4 + 3]; }
    R_to_q(Rm, rot);
    const double *P = params, *Q = params + 3;
    q_inv(rot, rinv); q_mul(rinv, Q, q);
    _Pragma("unroll") for (int k = 0; k < 3; k++) { r[k] = info[0] * (P[k] - pos[k]); r[3 + k] = info[1] * (2 * q[k]); }
    if (!J) return;
    double Qinv[4], qa[4], L[16];
    q_inv(Q, Qinv); q_mul(Qinv, rot, qa); quat_matrix(qa, 1.0, L);
    _Pragma("unroll") for (int k = 0; k < 42; k++) J[k] = 0.0;
    _Pragma("unroll") for (int i = 0; i < 3; i++) {
        J[i * 7 + i] = info[0];
        _Pragma("unroll") for (int j = 0; j < 3; j++) J[(3 + i) * 7 + 3 + j] = info[1] * L[i * 4 + j];
    }
}

// params[1] inv_dep; consts[44] pt_i.xy, pt_j.xy, Ri, Pi, Rj, Pj, EX(4x4); info[1] = FACTOR_WEIGHT.  J: [2] or null.
__device__ void reproj_factor(const double *params, const double *consts, const double *info, double *r, double *J)
{
    const double p_i[3] = { consts[0], consts[1], 1.0 }, p_j[3] = { consts[2], consts[3], 1.0 };
    const double *Ri = consts + 4, *Pi = consts + 13, *Rj = consts + 16, *Pj = consts + 25, *EX = consts + 28;
    double Rlc[9], Tlc[3];
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) Rlc[i * 3 + j] = EX[i * 4 + j]; Tlc[i] = EX[i * 4 + 3]; }
    const double dep = 1.0 / params[0], w = info[0];
    const double pc[3] = { dep * p_i[0], dep * p_i[1], dep * p_i[2] };
    double pl[3], pw[3], plj[3], pcj[3], tmp[3], RjT[9], RlcT[9];
    mT(Rj, RjT); mT(Rlc, RlcT);
    mv(Rlc, pc, pl); for (int k = 0; k < 3; k++) pl[k] += Tlc[k];
    mv(Ri, pl, pw); for (int k = 0; k < 3; k++) pw[k] += Pi[k];
    for (int k = 0; k < 3; k++) tmp[k] = pw[k] - Pj[k];
    mv(RjT, tmp, plj);
    for (int k = 0; k < 3; k++) tmp[k] = plj[k] - Tlc[k];
    mv(RlcT, tmp, pcj);
    const double d = pcj[2];
    r[0] = w * (pcj[0] / d - p_j[0]);
    r[1] = w * (pcj[1] / d - p_j[1]);
    if (!J) return;
    const double red[6] = { w * (1.0 / d), 0, w * (-pcj[0] / (d * d)), 0, w * (1.0 / d), w * (-pcj[1] / (d * d)) };
    double T[9], v[3];
    mm(RlcT, RjT, T); mm(T, Ri, T); mm(T, Rlc, T);
    mv(T, p_i, v);
    for (int a = 0; a < 2; a++) J[a] = -(red[a * 3] * v[0] + red[a * 3 + 1] * v[1] + red[a * 3 + 2] * v[2]) * dep * dep;
}

// PoseLocalParameterization::Plus (src/factor/PoseLocalParameterization.cc:15-31)
__device__ __forceinline__ void pose_plus(const double *x, const double *delta, double *out)
{
    const double dq[4] = { delta[3] / 2.0, delta[4] / 2.0, delta[5] / 2.0, 1.0 };
    double q[4];
    for (int k = 0; k < 3; k++) out[k] = x[k] + delta[k];
    q_mul(x + 3, dq, q);
    q_norm(q, out + 3);
}

} // namespace ba

struct FactorDims { int np, nc, ni, nr, nj; };
__host__ __device__ inline FactorDims factor_dims(int kind)
{
    switch (kind) {
    case 0: return { 14, 24, 36, 6, 84 };
    case 1: return { 22, 4, 4, 2, 44 };
    case 2: return { 7, 16, 2, 6, 42 };
    default: return { 1, 44, 1, 2, 2 };
    }
}

// first Jacobian element of parameter block k of a factor kind, in doubles; entry n_blocks = total (the packed layout of lmono_hip.h)
__device__ __forceinline__ int factor_block_begin(int kind, int k)
{
    switch (kind) {
    case 0: return k == 0 ? 0 : (k == 1 ? 42 : 84);                          // LASER: J_i [6x7], J_j [6x7]
    case 1: return k == 0 ? 0 : (k == 1 ? 14 : (k == 2 ? 28 : (k == 3 ? 42 : 44)));   // MONO: J_ex, J_i, J_j [2x7], J_depth [2x1]
    case 2: return k == 0 ? 0 : 42;                                          // PRIOR: J_ex [6x7]
    default: return k == 0 ? 0 : 2;                                          // REPROJ: J_depth [2x1]
    }
}
__device__ __forceinline__ int factor_blocks(int kind) { return kind == 0 ? 2 : (kind == 1 ? 4 : 1); }

// block_mask (optional, one byte per residual block): bit k set = the caller wants the Jacobian of parameter block k, as
// ceres::CostFunction::Evaluate is called with jacobians[k] != NULL; the other blocks of J are left untouched.  mask == 0 for a
// residual block = jacobians == NULL for it.
__global__ __launch_bounds__(64) void k_factor_eval(int kind, int count, const double *params, const double *consts, const double *info,
                                                    double *r, double *J, const unsigned char *block_mask)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const FactorDims d = factor_dims(kind);
    double p[22], c[44], inf[36], ro[6], jo[84];
    for (int k = 0; k < d.np; k++) p[k] = params[(size_t)i * d.np + k];
    for (int k = 0; k < d.nc; k++) c[k] = consts[(size_t)i * d.nc + k];
    for (int k = 0; k < d.ni; k++) inf[k] = info[k];
    const unsigned int mask = block_mask ? block_mask[i] : 0xffu;
    double *jp = (J && mask) ? jo : nullptr;
    switch (kind) {
    case 0: ba::laser_factor(p, c, inf, ro, jp); break;
    case 1: ba::mono_factor(p, c, inf, ro, jp); break;
    case 2: ba::prior_factor(p, c, inf, ro, jp); break;
    default: ba::reproj_factor(p, c, inf, ro, jp); break;
    }
    for (int k = 0; k < d.nr; k++) r[(size_t)i * d.nr + k] = ro[k];
    if (jp)
        for (int bk = 0; bk < factor_blocks(kind); bk++)
            if (mask & (1u << bk))
                for (int k = factor_block_begin(kind, bk); k < factor_block_begin(kind, bk + 1); k++) J[(size_t)i * d.nj + k] = jo[k];
}

} // namespace lmono
