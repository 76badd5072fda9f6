"""BA factor kernels (through the C ABI) against the CPU oracle and the committed goldens."""
import os

import numpy as np
import pytest

from tests import ba_cases as K

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ba_factors.npz")
CASES = {0: K.laser_cases, 1: K.mono_cases, 2: K.prior_cases, 3: K.reproj_cases}


def _rel(a, b):
    return np.abs(a - b).max() / (np.abs(b).max() + 1.0)


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
def test_factor_eval_matches_oracle(oracle, gpu_ctx, kind):
    P, Cn, info = CASES[kind](300, seed=50 + kind)
    r, J = gpu_ctx.factor_eval(kind, P, Cn, info)
    ro, Jo = oracle.factor_eval(kind, P, Cn, info)
    # fp64 on both sides, same operation order up to compiler scheduling: 1e-12 relative
    assert _rel(r, ro) < 1e-12 and _rel(J, Jo) < 1e-12
    r2, _ = gpu_ctx.factor_eval(kind, P, Cn, info, want_jac=False)
    assert np.array_equal(r, r2)


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
def test_factor_eval_goldens(gpu_ctx, kind):
    g = np.load(GOLD)
    r, J = gpu_ctx.factor_eval(kind, g["P%d" % kind], g["C%d" % kind], g["I%d" % kind])
    assert _rel(r, g["r%d" % kind]) < 1e-12 and _rel(J, g["J%d" % kind]) < 1e-12


def test_factor_eval_device_resident_large_batch(oracle, gpu_ctx):
    """200 k MonoProjectionFactor blocks resident in HBM (the BA bench shape); spot-check against the oracle and
    check linearity in sqrt_info (a size-independent property: r(2 W) = 2 r(W), J(2 W) = 2 J(W))."""
    import torch
    P, Cn, info = K.mono_cases(2000, seed=9)
    reps = 100
    Pd = torch.from_numpy(np.tile(P, (reps, 1))).cuda(); Cd = torch.from_numpy(np.tile(Cn, (reps, 1))).cuda()
    n = Pd.shape[0]
    I1 = torch.from_numpy(info.ravel().copy()).cuda(); I2 = 2 * I1
    r1 = torch.zeros((n, 2), dtype=torch.float64, device="cuda"); J1 = torch.zeros((n, 44), dtype=torch.float64, device="cuda")
    r2 = torch.zeros_like(r1); J2 = torch.zeros_like(J1)
    gpu_ctx.factor_eval_d(1, n, Pd.data_ptr(), Cd.data_ptr(), I1.data_ptr(), r1.data_ptr(), J1.data_ptr())
    gpu_ctx.factor_eval_d(1, n, Pd.data_ptr(), Cd.data_ptr(), I2.data_ptr(), r2.data_ptr(), J2.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(r2, 2 * r1) and torch.equal(J2, 2 * J1)
    ro, Jo = oracle.factor_eval(1, P, Cn, info)
    assert _rel(r1[:2000].cpu().numpy(), ro) < 1e-12 and _rel(J1[-2000:].cpu().numpy(), Jo) < 1e-12


def test_per_block_null_jacobians(oracle, gpu_ctx):
    """ceres::CostFunction::Evaluate may be handed jacobians[k] == NULL for any subset of the parameter blocks (LaserFactor.h:45,
    MonoProjectionFactor.cc:40): lmono_factor_eval_blocks writes exactly the requested blocks and leaves the others untouched."""
    from tests import ba_cases as K
    bounds = {0: [0, 42, 84], 1: [0, 14, 28, 42, 44], 2: [0, 42], 3: [0, 2]}
    cases = {0: K.laser_cases(12), 1: K.mono_cases(12), 2: K.prior_cases(12), 3: K.reproj_cases(12)}
    rng = np.random.default_rng(4)
    for kind, (P, Cn, info) in cases.items():
        r_ref, J_ref = oracle.factor_eval(kind, P, Cn, info)
        nb = len(bounds[kind]) - 1
        mask = rng.integers(0, 1 << nb, len(P)).astype(np.uint8)
        mask[0] = 0; mask[1] = (1 << nb) - 1
        r, J = gpu_ctx.factor_eval_blocks(kind, P, Cn, info, mask)
        assert np.abs(r - r_ref).max() <= 1e-12 * (np.abs(r_ref).max() + 1)
        for i in range(len(P)):
            for b in range(nb):
                blk = slice(bounds[kind][b], bounds[kind][b + 1])
                if mask[i] >> b & 1:
                    assert np.abs(J[i, blk] - J_ref[i, blk]).max() <= 1e-12 * (np.abs(J_ref[i]).max() + 1)
                else:
                    assert np.isnan(J[i, blk]).all(), "block %d of residual %d was written although jacobians[%d] == NULL" % (b, i, b)
