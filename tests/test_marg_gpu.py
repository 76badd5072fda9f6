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
