"""The chained odometry schedule validates itself (LMONO_OPT_BOUNDARY_TOL, lmono_odom_boundary_report): every
chain's warm start is compared on the device with the increment the strictly sequential schedule would have warm-started from,
chains above the tolerance are re-started from it and re-run until their increments agree with the stored ones, in rounds.
Checked over chain layouts 128 .. 512 on BOTH bench-scale sequences -- the one the schedule was tuned on (seq 0, figure-8) and a
held-out one (seq 1: other world, clover trajectory at 4-10 m/s; tests/golden/s1_seq01_oracle.npz) -- and, from round 5, on the STRESS
sequence (seq 2: a cluttered world -- 200 small boxes, 20 % stray returns at random ranges, moving cylinders, rings with dropped sectors;
tests/golden/s1_seq02_oracle.npz) -- against the strictly sequential CPU oracle, with no per-layout tuning."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
LAYOUTS = (128, 192, 224, 256, 320, 512)


def _bench_defaults():
    src = open(os.path.join(ROOT, "bench.py")).read()
    lead = int(re.search(r'"--lead", type=int, default=(\d+)', src).group(1))
    lead_full = int(re.search(r'"--lead-full", type=int, default=(-?\d+)', src).group(1))
    return lead, lead_full


def _scans(seq, n):
    from workloads import s1 as S1
    if seq == 0:
        w = S1.S1World(n_az=2000)
        return w.scans(w.trajectory(n))
    if seq == 1:
        w = S1.S1World(seed=777, n_az=2000)
        return w.scans(w.trajectory_clover(n))
    w = S1.S1World(seed=4242, n_az=2000, clutter=True)          # the stress sequence: nothing was tuned against it
    return w.scans(w.trajectory(n))


@pytest.fixture(scope="module", params=[0, 1, 2], ids=["seq00_tuned", "seq01_held_out", "seq02_clutter"])
def seq(request, gpu_ctx):
    import torch
    import lmono_amd
    gold = np.load(os.path.join(ROOT, "tests", "golden", "s1_seq%02d_oracle.npz" % request.param))
    n = len(gold["poses"])
    xyzi, off = _scans(request.param, n)
    assert (np.diff(off) == gold["n_points"]).all(), "the S1 generator no longer produces the fixture's scans"
    xd = torch.from_numpy(xyzi).cuda()
    del xyzi
    batch = lmono_amd.ScanBatch(gpu_ctx, n, int(off[-1]))
    batch.scanreg(xd.data_ptr(), off, 64, 5.0, keepalive=xd)
    yield dict(batch=batch, gold=gold, n=n, id=request.param)
    batch.close()
    del xd
    torch.cuda.empty_cache()


def test_front_end_and_sequential_schedule_on_this_sequence(seq):
    cnt = seq["batch"].counts()
    assert (cnt[:, 5] == 0).all() and (cnt[:, 1:5] == seq["gold"]["feat_counts"]).all()
    incr, poses = seq["batch"].odometry(1, 0)
    assert np.abs(incr - seq["gold"]["incr"]).max() < 1e-7
    rep = seq["batch"].boundary_report()
    assert rep["n_chains"] == 1 and rep["flagged"] == 0 and rep["rounds"] == 0


def test_every_layout_meets_the_1cm_bar_without_tuning(seq):
    from lmono_amd import trajectory
    lead, lead_full = _bench_defaults()
    b = seq["batch"]; ctx = b.ctx
    ctx.set_option(ctx.OPT_LEAD_FULL, lead_full)
    try:
        for chains in LAYOUTS:
            incr, poses = b.odometry(chains, lead)
            rep = b.boundary_report()
            ate = trajectory.ate(poses, seq["gold"]["poses"])
            r1 = trajectory.rpe(poses, seq["gold"]["poses"], 1)
            print("seq %d chains %d lead %d: ATE %.6f m, RPE(1) %.2e m; boundaries flagged %d (chains re-run %d, pairs %d, rounds %d), "
                  "max residual %.2e, repair %.2f ms" % (seq["id"], chains, lead, ate, r1["trans_rmse_m"], rep["flagged"], rep["chains_rerun"],
                                                        rep["pairs_rerun"], rep["rounds"], rep["max_resid"], rep["repair_ms"]))
            assert rep["n_chains"] == chains and rep["unresolved"] == 0
            assert rep["tol"] == pytest.approx(1e-6)
            assert ate <= 0.01                                     # north_star: ATE within 1 cm of the reference path
            assert r1["trans_rmse_m"] <= 1e-3 and r1["rot_rmse_deg"] <= 1e-3
            # what the report says is what happened: flagged chains are exactly those whose first residual exceeds the tolerance
            first_flags = int((rep["resid"] > rep["tol"]).sum())
            assert rep["chains_rerun"] >= first_flags and (rep["rerun"] > 0).sum() == rep["chains_rerun"]
    finally:
        ctx.set_option(ctx.OPT_LEAD_FULL, -1)


def test_thinner_lead_ins_stay_inside_the_bar(seq):
    """LMONO_OPT_LEAD_FULL 1 and 0 (less accurate warm starts: every boundary is flagged).  With LEAD_FULL 1 the held-out sequence once came
    out 11 mm off although every boundary had passed: a repair chain had stopped at the first pair that agreed within the tolerance, and the
    NEXT pair amplified that 1e-6 difference to 1.9e-4 rad (a correspondence set on a knife edge).  A repair chain now has to reproduce two
    consecutive pairs (kRepairAgree, odometry.hip) before the rest of its chain stands."""
    from lmono_amd import trajectory
    lead, _ = _bench_defaults()
    b = seq["batch"]; ctx = b.ctx
    try:
        for lead_full in (1, 0):
            ctx.set_option(ctx.OPT_LEAD_FULL, lead_full)
            for ld in (lead, lead + 1):
                _, poses = b.odometry(256, ld)
                rep = b.boundary_report()
                ate = trajectory.ate(poses, seq["gold"]["poses"])
                print("seq %d, 256 chains, lead %d, LEAD_FULL %d: ATE %.6f m, %d chains re-run, %d pairs" % (seq["id"], ld, lead_full, ate, rep["chains_rerun"], rep["pairs_rerun"]))
                assert rep["unresolved"] == 0 and ate <= 0.002
    finally:
        ctx.set_option(ctx.OPT_LEAD_FULL, -1)


def test_lead_ins_seeded_from_the_neighbouring_chains_stay_inside_the_bar(seq):
    """LMONO_OPT_LEAD_SEED = 1 (round 6): after the first step of the pass the state of every chain still in its lead-in becomes the
    component-wise median of the first results of chains c - 1, c, c + 1 (a constant-velocity prior across 1.8 s).  The validation is the arbiter of the
    result whatever the lead-ins start from: on all three worlds the seeded schedule ends on the sequential oracle's trajectory like the unseeded one.
    (What it buys is measured in profiles/r6/lead_in_seeding_and_length_three_worlds.txt: nothing -- a first pair from the identity is biased, not an
    outlier, so its neighbours' first pairs are off the same way; the option ships off.)"""
    from lmono_amd import trajectory
    lead, lead_full = _bench_defaults()
    b = seq["batch"]; ctx = b.ctx
    ctx.set_option(ctx.OPT_LEAD_FULL, lead_full)
    try:
        out = {}
        for seed in (0, 1):
            ctx.set_option(ctx.OPT_LEAD_SEED, seed)
            _, poses = b.odometry(256, lead)
            rep = b.boundary_report()
            out[seed] = (trajectory.ate(poses, seq["gold"]["poses"]), rep["flagged"], rep["pairs_rerun"])
            assert rep["unresolved"] == 0 and out[seed][0] <= 0.002
        print("seq %d, 256 chains, lead %d: identity lead-ins ATE %.6f m, %d flagged, %d pairs re-run; seeded %.6f m, %d flagged, %d pairs"
              % ((seq["id"], lead) + out[0] + out[1]))
    finally:
        ctx.set_option(ctx.OPT_LEAD_SEED, 0)
        ctx.set_option(ctx.OPT_LEAD_FULL, -1)


def test_validation_off_is_the_round2_schedule_and_on_only_improves(seq):
    """tol 0 = no check (the round-2 behaviour: 224 chains x lead 7 missed the bar on seq 0 by one unlucky boundary)."""
    from lmono_amd import trajectory
    lead, lead_full = _bench_defaults()
    b = seq["batch"]; ctx = b.ctx
    ctx.set_option(ctx.OPT_LEAD_FULL, lead_full)
    try:
        ctx.set_option(ctx.OPT_BOUNDARY_TOL, 0)
        _, p_off = b.odometry(224, lead)
        rep = b.boundary_report()
        assert rep["flagged"] == 0 and rep["rounds"] == 0 and rep["pairs_rerun"] == 0
        ctx.set_option(ctx.OPT_BOUNDARY_TOL, 1000)
        _, p_on = b.odometry(224, lead)
        a_off, a_on = trajectory.ate(p_off, seq["gold"]["poses"]), trajectory.ate(p_on, seq["gold"]["poses"])
        print("seq %d, 224 chains: ATE without validation %.5f m, with %.6f m" % (seq["id"], a_off, a_on))
        assert a_on <= 0.01 and a_on <= a_off + 1e-4
    finally:
        ctx.set_option(ctx.OPT_BOUNDARY_TOL, 1000)
        ctx.set_option(ctx.OPT_LEAD_FULL, -1)


def test_chains_without_a_lead_in_are_repaired_to_the_sequential_result(gpu_ctx):
    """lead 0: every chain starts from the identity, every boundary is flagged; short chains make the repairs reach their chain's end, so
    the next boundary is flagged in the next round (cascade).  With a tight tolerance the result is the sequential schedule's."""
    import torch
    import lmono_amd
    n = 48
    xyzi, off = _scans(0, n)
    xd = torch.from_numpy(xyzi).cuda()
    b = lmono_amd.ScanBatch(gpu_ctx, n, int(off[-1]))
    b.scanreg(xd.data_ptr(), off, 64, 5.0, keepalive=xd)
    seq_incr, seq_poses = b.odometry(1, 0)
    try:
        gpu_ctx.set_option(gpu_ctx.OPT_BOUNDARY_TOL, 1)            # 1e-9
        for chains in (4, 16):
            incr, poses = b.odometry(chains, 0)
            rep = b.boundary_report()
            print("lead 0, %d chains of %d scans: flagged %d, pairs re-run %d, rounds %d, max |incr - sequential| %.2e"
                  % (chains, n // chains, rep["flagged"], rep["pairs_rerun"], rep["rounds"], np.abs(incr - seq_incr).max()))
            assert rep["unresolved"] == 0 and rep["chains_rerun"] == chains - 1
            assert rep["resid"][0] == 0 and (rep["resid"][1:] > 1e-3).all()        # identity vs 0.8 m of motion
            assert np.abs(incr - seq_incr).max() < 1e-7
            if chains == 16:
                assert rep["rounds"] > 1
    finally:
        gpu_ctx.set_option(gpu_ctx.OPT_BOUNDARY_TOL, 1000)
        b.close()


def test_shard_entry_validates_the_external_boundary(gpu_ctx):
    """Scan-range sharding: a rank's batch starts `lead` scans early; lmono_odom_shard_d + lmono_odom_shard_validate with the
    previous rank's last increment give the increments of the unsharded run on the owned scans."""
    import torch
    import lmono_amd
    n, cut, lead = 96, 40, 7
    xyzi, off = _scans(0, n)
    xd = torch.from_numpy(xyzi).cuda()
    whole = lmono_amd.ScanBatch(gpu_ctx, n, int(off[-1]))
    whole.scanreg(xd.data_ptr(), off, 64, 5.0, keepalive=xd)
    ref_incr, _ = whole.odometry(1, 0)
    whole.close()
    lo = int(off[cut - lead])
    off1 = (off[cut - lead:] - lo).astype(np.int64)
    x1 = xd[lo:].contiguous()
    shard = lmono_amd.ScanBatch(gpu_ctx, len(off1) - 1, int(off1[-1]))
    shard.scanreg(x1.data_ptr(), off1, 64, 5.0, keepalive=x1)
    incr_d = torch.zeros((len(off1) - 1, 7), dtype=torch.float64, device="cuda")
    try:
        gpu_ctx.set_option(gpu_ctx.OPT_BOUNDARY_TOL, 1)
        shard.odometry_shard_d(4, lead, lead, incr_d.data_ptr())
        gpu_ctx.synchronize()
        before = incr_d.cpu().numpy()[lead:]
        changed = shard.shard_validate(ref_incr[cut - 1], incr_d.data_ptr())
        gpu_ctx.synchronize()
        after = incr_d.cpu().numpy()[lead:]
        rep = shard.boundary_report()
        print("shard: external boundary residual -> flagged %d, pairs re-run %d; max |incr - unsharded| before %.2e after %.2e; changed_last %s"
              % (rep["flagged"], rep["pairs_rerun"], np.abs(before - ref_incr[cut:]).max(), np.abs(after - ref_incr[cut:]).max(), changed))
        assert np.abs(after - ref_incr[cut:]).max() < 1e-7
        assert rep["unresolved"] == 0
        # the deferred flow of bench.py --gpus N: main pass, (exchange,) ONE validation of the three inner boundaries and the external one
        incr_d.zero_()
        shard.odometry_shard_main_d(4, lead, lead, incr_d.data_ptr())
        shard.shard_validate(ref_incr[cut - 1], incr_d.data_ptr())
        gpu_ctx.synchronize()
        rep2 = shard.boundary_report()
        deferred = incr_d.cpu().numpy()[lead:]
        print("shard, deferred validation: flagged %d of %d chains, pairs re-run %d, rounds %d; max |incr - unsharded| %.2e"
              % (rep2["flagged"], rep2["n_chains"], rep2["pairs_rerun"], rep2["rounds"], np.abs(deferred - ref_incr[cut:]).max()))
        assert np.abs(deferred - ref_incr[cut:]).max() < 1e-7 and rep2["unresolved"] == 0
        print("   per chain: residual", rep2["resid"], "pairs re-run", rep2["rerun"], "| two-step flow:", rep["resid"], rep["rerun"])
        assert rep2["rounds"] <= rep["rounds"] + 1 and rep2["chains_rerun"] >= 1
        # the rank that owns the first scan: inner boundaries only; and without a pending validation a NULL boundary is refused
        shard.odometry_shard_main_d(4, 3, 0, incr_d.data_ptr())
        shard.shard_validate(None, incr_d.data_ptr())
        assert shard.boundary_report()["unresolved"] == 0
        with pytest.raises(lmono_amd.LmonoError):
            shard.shard_validate(None, incr_d.data_ptr())
    finally:
        gpu_ctx.set_option(gpu_ctx.OPT_BOUNDARY_TOL, 1000)
        shard.close()
