"""MarginalizationInfo::marginalize / Marginalization::Evaluate restated (oracle/lo_marg.c) vs a numpy dense computation."""
import numpy as np

from tests import ba_cases as K


def test_marginalize_matches_numpy_dense(oracle):
    from oracle import ba_numpy as B
    w = K.make_window(5)
    J, r, m, x0, sel = oracle.marginalize(w)
    F0 = len(sel["feats"])
    assert m == 6 + F0 and F0 > 10
    n, pos = 66, m + 66
    H = np.zeros((pos, pos)); b = np.zeros(pos)
    idx_pose = lambda i: 0 if i == 0 else m + 6 + 6 * (i - 1)

    def add(blocks, res):
        for (ia, Ja) in blocks:
            for (ib, Jb) in blocks:
                H[ia:ia + Ja.shape[1], ib:ib + Jb.shape[1]] += Ja.T @ Jb
            b[ia:ia + Ja.shape[1]] += Ja.T @ res
    c = w["laser_consts"][0]
    rr, Ji, Jj = B.laser_factor(w["poses"][0], w["poses"][1], c[:9].reshape(3, 3), c[9:18].reshape(3, 3), c[18:21], c[21:24], w["laser_info"])
    add([(idx_pose(0), Ji[:, :6]), (idx_pose(1), Jj[:, :6])], rr)
    for o in range(len(sel["obs_j"])):
        f, j = sel["obs_feat"][o], sel["obs_j"][o]
        rr, Jx, Ja, Jb_, Jd = B.mono_projection_factor(w["ex"], w["poses"][0], w["poses"][j], sel["invd"][f], sel["pts"][o, :2], sel["pts"][o, 2:], w["mono_info"])
        rc, Js = B.corrector(rr, [Jx, Ja, Jb_, Jd.reshape(2, 1)], B.cauchy(rr @ rr))
        add([(m, Js[0][:, :6]), (idx_pose(0), Js[1][:, :6]), (idx_pose(j), Js[2][:, :6]), (6 + f, Js[3])], rc)
    Hmm = 0.5 * (H[:m, :m] + H[:m, :m].T)
    wv, V = np.linalg.eigh(Hmm)
    Hinv = V @ np.diag(np.where(wv > 1e-8, 1.0 / np.where(wv > 1e-8, wv, 1), 0)) @ V.T
    Hp = H[m:, m:] - H[m:, :m] @ Hinv @ H[:m, m:]
    bp = b[m:] - H[m:, :m] @ Hinv @ b[:m]
    wv2, V2 = np.linalg.eigh(Hp)
    keep = wv2 > 1e-8
    Hp_cut = (V2[:, keep] * wv2[keep]) @ V2[:, keep].T
    scale = np.abs(Hp_cut).max()
    assert np.abs(J.T @ J - Hp_cut).max() < 1e-8 * scale
    bp_cut = V2[:, keep] @ (V2[:, keep].T @ bp)
    assert np.abs(J.T @ r - bp_cut).max() < 1e-8 * (np.abs(bp_cut).max() + 1)
    # the 6-DoF gauge of the relative factors leaves a (numerically) rank-deficient prior: 66 - 6 strong directions
    assert (wv2 > 1e-6 * wv2.max()).sum() <= 66


def test_marginalization_evaluate(oracle):
    from oracle import ba_numpy as B
    w = K.make_window(6)
    J, r, m, x0, _ = oracle.marginalize(w)
    res, jac = oracle.marg_evaluate(J, r, x0, x0)
    assert np.abs(res - r).max() < 1e-9 * (np.abs(r).max() + 1)   # dx = 0 (to rounding) at the linearisation point
    for bk in range(11):
        assert np.array_equal(jac[bk][:, :6], J[:, 6 * bk:6 * bk + 6]) and (jac[bk][:, 6] == 0).all()
    # first-order behaviour: r(x0 (+) d) = r0 + J d for the parameterisation's own perturbation
    rng = np.random.default_rng(0)
    d = rng.normal(0, 1e-3, (11, 6))
    x = np.stack([B.pose_plus(x0[k], d[k]) for k in range(11)])
    res2, _ = oracle.marg_evaluate(J, r, x0, x, want_jac=False)
    lin = r + J @ d.ravel()
    assert np.abs(res2 - lin).max() < 1e-5 * (np.abs(lin).max() + 1)


def _second_new_numpy(J0, r0, x0, x, drop, oracle):
    """MarginalizationInfo::marginalize with the previous prior as the only factor (Estimator.cc:1406-1470), dense numpy."""
    nb = len(x)
    r = r0 + J0 @ oracle.prior_dx(x0, x)
    H, b = J0.T @ J0, J0.T @ r
    md = np.arange(6 * drop, 6 * drop + 6)
    kp = np.array([6 * k + c for k in range(nb) if k != drop for c in range(6)])
    Hmm = 0.5 * (H[np.ix_(md, md)] + H[np.ix_(md, md)].T)
    wv, V = np.linalg.eigh(Hmm)
    Hinv = V @ np.diag(np.where(wv > 1e-8, 1.0 / np.where(wv > 1e-8, wv, 1), 0)) @ V.T
    Hp = H[np.ix_(kp, kp)] - H[np.ix_(kp, md)] @ Hinv @ H[np.ix_(md, kp)]
    bp = b[kp] - H[np.ix_(kp, md)] @ Hinv @ b[md]
    wv2, V2 = np.linalg.eigh(Hp)
    keep = wv2 > 1e-8
    return (V2[:, keep] * wv2[keep]) @ V2[:, keep].T, V2[:, keep] @ (V2[:, keep].T @ bp)


def test_margin_second_new_matches_numpy_dense(oracle):
    """The MARGIN_SECOND_NEW branch: previous prior over [ex, pose0..pose9] (after the MARGIN_OLD address shift) loses pose9."""
    from oracle import ba_numpy as B
    w = K.make_window(7)
    J0, r0, m, x0, _ = oracle.marginalize(w)
    rng = np.random.default_rng(3)
    x = np.stack([B.pose_plus(x0[k], rng.normal(0, 2e-3, 6)) for k in range(11)])     # the window moved on since the prior was built
    J, r = oracle.marg_second_new(J0, r0, x0, x, 10)
    assert J.shape == (60, 60)
    H_ref, b_ref = _second_new_numpy(J0, r0, x0, x, 10, oracle)
    assert np.abs(J.T @ J - H_ref).max() < 1e-8 * np.abs(H_ref).max()
    assert np.abs(J.T @ r - b_ref).max() < 1e-8 * (np.abs(b_ref).max() + 1)
    # eliminating a block keeps the minimum over it: the new prior's cost at x equals min over pose9 of the old prior's cost
    c_new = 0.5 * r @ r
    r_old = r0 + J0 @ oracle.prior_dx(x0, x)
    Jd = J0[:, 60:66]
    step = np.linalg.lstsq(Jd, -r_old, rcond=None)[0]
    c_min = 0.5 * np.sum((r_old + Jd @ step) ** 2)
    # the eps-cut projections discard the part of b outside the kept eigenspace: agreement to the size of that part
    assert abs(c_new - c_min) < 1e-6 * (c_min + 1) + 0.5 * abs(np.sum(r_old ** 2) - np.sum((np.linalg.pinv(J0.T) @ (J0.T @ r_old)) ** 2))
    # a second elimination (any block index) is also supported
    J2, r2 = oracle.marg_second_new(J, r, np.delete(x, 10, 0), np.delete(x, 10, 0), 3)
    assert J2.shape == (54, 54)
