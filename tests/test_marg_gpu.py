"""k_marginalize / k_marg_evaluate (through the C ABI) against the CPU restatement (gauge-invariant products)."""
import numpy as np
import pytest

from tests import ba_cases as K

pytestmark = pytest.mark.gpu


def test_marginalize_matches_oracle(oracle, gpu_ctx):
    wins, refs = [], []
    for seed in (5, 6, 8):
        w = K.make_window(seed)
        J, r, m, x0, sel = oracle.marginalize(w)
        refs.append((J, r, x0))
        wins.append(dict(poses=w["poses"], ex=w["ex"], invd=sel["invd"], obs_feat=sel["obs_feat"], obs_j=sel["obs_j"], pts=sel["pts"],
                         laser01=w["laser_consts"][0], laser_info=w["laser_info"], mono_info=w["mono_info"]))
    Jg, rg, st = gpu_ctx.marginalize(wins)
    assert (st == 0).all()
    for k, (J, r, x0) in enumerate(refs):
        H_ref, b_ref = J.T @ J, J.T @ r
        H_gpu, b_gpu = Jg[k].T @ Jg[k], Jg[k].T @ rg[k]
        # structured pseudo-inverse + parallel Jacobi vs dense eigen pseudo-inverse: same prior to ~1e-8 relative
        assert np.abs(H_gpu - H_ref).max() < 1e-7 * np.abs(H_ref).max()
        assert np.abs(b_gpu - b_ref).max() < 1e-7 * (np.abs(b_ref).max() + 1)
        # the prior's cost at a perturbed state: 0.5 |r0 + J dx|^2 is what a solver would see
        from oracle import ba_numpy as B
        rng = np.random.default_rng(k)
        x = np.stack([B.pose_plus(x0[i], rng.normal(0, 1e-3, 6)) for i in range(11)])
        res_gpu = gpu_ctx.marg_evaluate(Jg[k:k + 1], rg[k:k + 1], x0[None], x[None])[0]
        res_self, _ = oracle.marg_evaluate(Jg[k], rg[k], x0, x, want_jac=False)
        assert np.abs(res_gpu - res_self).max() < 1e-9 * (np.abs(res_self).max() + 1)
        res_ref, _ = oracle.marg_evaluate(J, r, x0, x, want_jac=False)
        c_gpu, c_ref = 0.5 * res_gpu @ res_gpu, 0.5 * res_ref @ res_ref
        assert abs(c_gpu - c_ref) < 1e-6 * c_ref


def _prior_products(J, r):
    return J.T @ J, J.T @ r


def test_margin_chain_old_second_new_old(oracle, gpu_ctx):
    """MARGIN_OLD -> MARGIN_SECOND_NEW -> MARGIN_OLD as Estimator::margin() runs them (Estimator.cc:1307-1470),
    GPU against the oracle at every link.  Kept blocks after MARGIN_OLD: [ex, pose1..pose10] = after the address shift
    [ex, pose0..pose9]; MARGIN_SECOND_NEW drops the block aliasing para_pose[WINDOW_SIZE - 1] (index 10)."""
    from oracle import ba_numpy as B
    w = K.make_window(9)
    J0, r0, m, x0, sel = oracle.marginalize(w)
    win = dict(poses=w["poses"], ex=w["ex"], invd=sel["invd"], obs_feat=sel["obs_feat"], obs_j=sel["obs_j"], pts=sel["pts"],
               laser01=w["laser_consts"][0], laser_info=w["laser_info"], mono_info=w["mono_info"])
    Jg0, rg0, st = gpu_ctx.marginalize([win])
    assert st[0] == 0
    rng = np.random.default_rng(11)
    x = np.stack([B.pose_plus(x0[k], rng.normal(0, 2e-3, 6)) for k in range(11)])     # the next frame's state
    # link 2 on both sides from their OWN link-1 priors, and on the GPU from the oracle's prior (isolates the new kernel)
    J1, r1 = oracle.marg_second_new(J0, r0, x0, x, 10)
    Jg1, rg1, st1 = gpu_ctx.marg_second_new(Jg0, rg0, x0[None], x[None], 10)
    Jx1, rx1, _ = gpu_ctx.marg_second_new(J0[None], r0[None], x0[None], x[None], 10)
    assert st1[0] == 0 and Jg1.shape == (1, 60, 60)
    H_ref, b_ref = _prior_products(J1, r1)
    for Jt, rt, tol in ((Jx1[0], rx1[0], 1e-9), (Jg1[0], rg1[0], 1e-6)):
        H, b = _prior_products(Jt, rt)
        assert np.abs(H - H_ref).max() < tol * np.abs(H_ref).max()
        assert np.abs(b - b_ref).max() < tol * (np.abs(b_ref).max() + 1)
    # the new prior's cost at a further perturbed state (what a solver would see), new linearisation point = x without pose9
    xk = np.delete(x, 10, 0)
    xq = np.stack([B.pose_plus(xk[k], rng.normal(0, 1e-3, 6)) for k in range(10)])
    dx = oracle.prior_dx(xk, xq)
    c_ref = 0.5 * np.sum((r1 + J1 @ dx) ** 2); c_gpu = 0.5 * np.sum((rg1[0] + Jg1[0] @ dx) ** 2)
    assert abs(c_gpu - c_ref) < 1e-6 * c_ref
    # link 3: a keyframe again -- MARGIN_OLD builds a fresh prior (the previous one is never chained: `valid` stays false)
    w3 = K.make_window(10)
    J3, r3, m3, x03, sel3 = oracle.marginalize(w3)
    Jg3, rg3, st3 = gpu_ctx.marginalize([dict(poses=w3["poses"], ex=w3["ex"], invd=sel3["invd"], obs_feat=sel3["obs_feat"], obs_j=sel3["obs_j"],
                                              pts=sel3["pts"], laser01=w3["laser_consts"][0], laser_info=w3["laser_info"], mono_info=w3["mono_info"])])
    H_ref, b_ref = _prior_products(J3, r3); H, b = _prior_products(Jg3[0], rg3[0])
    assert st3[0] == 0 and np.abs(H - H_ref).max() < 1e-7 * np.abs(H_ref).max() and np.abs(b - b_ref).max() < 1e-7 * (np.abs(b_ref).max() + 1)


def test_marginalize_150_tracks_anchored_at_frame_0(oracle, gpu_ctx):
    """the tracker keeps up to MAX_CNT = 150 features per frame (FeatureTracker.cc:21) and all of them can be anchored at
    frame 0; lmono_marginalize takes them (capacity 160) and still matches the oracle."""
    w = K.make_window(5)
    J, r, m, x0, sel = oracle.marginalize(w)
    F0 = len(sel["invd"])
    # replicate the frame-0 tracks (slightly different depths / points) up to exactly 150
    rng = np.random.default_rng(2)
    reps = int(np.ceil(150 / F0))
    invd, of, oj, pts = [], [], [], []
    for rep in range(reps):
        for f in range(F0):
            if len(invd) == 150:
                break
            sel_o = np.nonzero(sel["obs_feat"] == f)[0]
            g = len(invd)
            invd.append(sel["invd"][f] * (1 + 0.01 * rep))
            for o in sel_o:
                of.append(g); oj.append(sel["obs_j"][o]); pts.append(sel["pts"][o] + rng.normal(0, 1e-4 * rep, 4))
    assert len(invd) == 150
    w2 = dict(w)
    poses, ex = w["poses"], w["ex"]
    lc = np.ascontiguousarray(w["laser_consts"][0], np.float64)
    invd = np.array(invd); of = np.array(of, np.int32); oj = np.array(oj, np.int32); pts = np.array(pts)
    import ctypes as C
    Jr = np.zeros((66, 66)); rr = np.zeros(66); mm = C.c_int(0)
    li = np.ascontiguousarray(w["laser_info"], np.float64); mi = np.ascontiguousarray(w["mono_info"], np.float64)
    fp = lambda a: a.ctypes.data_as(C.c_void_p)
    oracle.lib().lo_marginalize(fp(np.ascontiguousarray(poses)), fp(np.ascontiguousarray(ex)), C.c_int(150), fp(invd), C.c_int(len(of)), fp(of), fp(oj), fp(pts),
                                fp(lc), fp(li), fp(mi), fp(Jr), fp(rr), C.byref(mm))
    assert mm.value == 156
    Jg, rg, st = gpu_ctx.marginalize([dict(poses=poses, ex=ex, invd=invd, obs_feat=of, obs_j=oj, pts=pts, laser01=lc, laser_info=li, mono_info=mi)])
    H_ref, b_ref = Jr.T @ Jr, Jr.T @ rr
    H, b = Jg[0].T @ Jg[0], Jg[0].T @ rg[0]
    assert np.abs(H - H_ref).max() < 1e-7 * np.abs(H_ref).max() and np.abs(b - b_ref).max() < 1e-7 * (np.abs(b_ref).max() + 1)
    # one more than the capacity is a clean error, not a fault
    import lmono_amd
    big = dict(poses=poses, ex=ex, invd=np.full(161, 0.1), obs_feat=np.arange(161, dtype=np.int32), obs_j=np.full(161, 1, np.int32), pts=np.zeros((161, 4)),
               laser01=lc, laser_info=li, mono_info=mi)
    with pytest.raises(lmono_amd.LmonoError):
        gpu_ctx.marginalize([big])
