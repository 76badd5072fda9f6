"""The headline configuration checked at its tolerance over the WHOLE configs[1] sequence: 4541 S1 scans,
the bench's (chains, lead), against the committed trajectory of the strictly sequential CPU oracle
(tests/golden/s1_seq00_oracle.npz, generator tests/golden/make_s1_trajectory.py).  Also: the GPU's own sequential schedule
(n_chains 1, lead 0) reproduces that trajectory, and the front end's feature counts are equal for every scan."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench_defaults():
    src = open(os.path.join(ROOT, "bench.py")).read()
    chains = int(re.search(r'"--chains", type=int, default=(\d+)', src).group(1))
    lead = int(re.search(r'"--lead", type=int, default=(\d+)', src).group(1))
    lead_full = int(re.search(r'"--lead-full", type=int, default=(-?\d+)', src).group(1))
    return chains, lead, lead_full


@pytest.fixture(scope="module")
def seq00(gpu_ctx):
    import torch
    import lmono_amd
    from workloads import s1 as S1
    gold = np.load(os.path.join(ROOT, "tests", "golden", "s1_seq00_oracle.npz"))
    n = len(gold["poses"])
    w = S1.S1World(n_az=2000)
    traj = w.trajectory(n)
    xyzi, off = w.scans(traj)
    assert (np.diff(off) == gold["n_points"]).all(), "the S1 generator no longer produces the fixture's scans"
    xd = torch.from_numpy(xyzi).cuda()
    del xyzi
    batch = lmono_amd.ScanBatch(gpu_ctx, n, int(off[-1]))
    batch.scanreg(xd.data_ptr(), off, 64, 5.0, keepalive=xd)
    return dict(batch=batch, gold=gold, n=n)


def test_front_end_feature_counts_equal_over_4541_scans(seq00):
    cnt = seq00["batch"].counts()
    assert (cnt[:, 5] == 0).all()
    assert (cnt[:, 1:5] == seq00["gold"]["feat_counts"]).all()


def test_sequential_schedule_reproduces_the_cpu_trajectory(seq00):
    incr, poses = seq00["batch"].odometry(1, 0)
    g = seq00["gold"]
    assert np.abs(incr - g["incr"]).max() < 1e-7
    assert np.abs(poses - g["poses"]).max() < 1e-5          # 4540 composed increments
    from lmono_amd import trajectory
    assert trajectory.ate(poses, g["poses"]) < 1e-6


def test_bench_schedule_meets_the_1cm_bar_over_the_whole_sequence(seq00):
    from lmono_amd import trajectory
    chains, lead, lead_full = _bench_defaults()
    ctx = seq00["batch"].ctx
    ctx.set_option(ctx.OPT_LEAD_FULL, lead_full)
    try:
        incr, poses = seq00["batch"].odometry(chains, lead)
    finally:
        ctx.set_option(ctx.OPT_LEAD_FULL, -1)
    g = seq00["gold"]
    ate = trajectory.ate(poses, g["poses"])
    r1 = trajectory.rpe(poses, g["poses"], 1)
    r100 = trajectory.rpe(poses, g["poses"], 100)
    print("chains %d lead %d (last %d lead-in pairs on all features): ATE %.5f m, RPE(1) %.2e m / %.2e deg, RPE(100) %.2e m / %.2e deg"
          % (chains, lead, lead_full, ate, r1["trans_rmse_m"], r1["rot_rmse_deg"], r100["trans_rmse_m"], r100["rot_rmse_deg"]))
    assert ate <= 0.01                                     # north_star: ATE within 1 cm of the reference path
    assert r1["trans_rmse_m"] <= 1e-3 and r1["rot_rmse_deg"] <= 1e-3


def test_chain_groups_and_lead_in_options_at_bench_scale(seq00):
    """256 chains: (1) LMONO_OPT_ODOM_STREAMS -- 1, 2 or 4 chain groups on their own HIP streams give the same increments bit for
    bit; (2) LMONO_OPT_LEAD_FULL -- a value >= lead changes nothing, a smaller one only moves the result within the tolerance."""
    from lmono_amd import trajectory
    chains, lead, _ = _bench_defaults()
    b = seq00["batch"]; ctx = b.ctx
    assert ctx.odom_chain_groups(chains) == 4 and ctx.odom_chain_groups(6) == 1
    ref_i, ref_p = b.odometry(chains, lead)
    try:
        for g in (1, 2):
            ctx.set_option(ctx.OPT_ODOM_STREAMS, g)
            assert ctx.odom_chain_groups(chains) == g
            i, p = b.odometry(chains, lead)
            assert np.array_equal(i, ref_i) and np.array_equal(p, ref_p)
        ctx.set_option(ctx.OPT_ODOM_STREAMS, 4)
        ctx.set_option(ctx.OPT_LEAD_FULL, lead)
        i, p = b.odometry(chains, lead)
        assert np.array_equal(i, ref_i)
        # a smaller value thins the early lead-in pairs: the unvalidated increments change, the validated ones stay inside the bar (the
        # repairs usually bring them back to the very same numbers)
        ctx.set_option(ctx.OPT_LEAD_FULL, 1)
        ctx.set_option(ctx.OPT_BOUNDARY_TOL, 0)
        i0, _ = b.odometry(chains, lead)
        ctx.set_option(ctx.OPT_LEAD_FULL, lead)
        i1, _ = b.odometry(chains, lead)
        assert not np.array_equal(i0, i1)
        ctx.set_option(ctx.OPT_BOUNDARY_TOL, 1000)
        ctx.set_option(ctx.OPT_LEAD_FULL, 1)
        i, p = b.odometry(chains, lead)
        assert trajectory.ate(p, seq00["gold"]["poses"]) <= 0.01
        # the strictly sequential schedule has no lead-in: the option does not touch it
        ctx.set_option(ctx.OPT_LEAD_FULL, 0)
        i1, _ = b.odometry(1, 0)
        assert np.abs(i1 - seq00["gold"]["incr"]).max() < 1e-7
    finally:
        ctx.set_option(ctx.OPT_ODOM_STREAMS, 4)
        ctx.set_option(ctx.OPT_LEAD_FULL, -1)
        ctx.set_option(ctx.OPT_BOUNDARY_TOL, 1000)


def test_context_on_a_caller_stream_gives_the_same_increments(seq00):
    """A context moved onto a caller-created stream runs at most 3 chain groups, all on the library's own streams (lmono_hip.hip:
    odom_run); the increments are the null-stream context's bit for bit."""
    import torch
    import lmono_amd
    from workloads import s1 as S1
    chains, lead, _ = _bench_defaults()
    n = 1024                                            # a quarter of the sequence is enough to run 3 groups of 85 chains
    w = S1.S1World(n_az=2000)
    xyzi, off = w.scans(w.trajectory(n))
    xd = torch.from_numpy(xyzi).cuda()
    del xyzi
    out = []
    for own_stream in (False, True):
        ctx = lmono_amd.Context(0)
        if own_stream:
            st = torch.cuda.Stream()
            ctx.set_stream(st.cuda_stream)
            assert ctx.odom_chain_groups(chains) == 3
        else:
            assert ctx.odom_chain_groups(chains) == 4
        b = lmono_amd.ScanBatch(ctx, n, int(off[-1]))
        torch.cuda.synchronize()
        b.scanreg(xd.data_ptr(), off, 64, 5.0, keepalive=xd)
        i, p = b.odometry(chains, lead)
        out.append((i, p))
        b.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
