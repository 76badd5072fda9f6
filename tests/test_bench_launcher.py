"""bench.py --gpus N without a launcher around it starts N ranks itself.  CPU only: the ranks join a
gloo group and count themselves (--probe-ranks); nothing touches a GPU."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, env_extra=None):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_gpus_2_spawns_two_ranks():
    out = _run(["--gpus", "2", "--probe-ranks"])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["requested"] == 2


def test_gpus_1_stays_in_process():
    out = _run(["--gpus", "1", "--probe-ranks"])
    assert out["n_gpus"] == 1 and out["ranks_seen"] == 1


def test_launcher_environment_is_respected():
    """Started the way the driver starts it (torch.distributed.run sets WORLD_SIZE), bench.py must not spawn again."""
    env = dict(os.environ)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--probe-ranks"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["ranks_seen"] == 2


def test_strong_and_weak_shard_sizes():
    from lmono_amd import sharding
    # strong: 4541 scans in total over 8 ranks (configs[3]); weak: 4541 per rank
    tot = 0
    for r in range(8):
        lb, ob, oe = sharding.shard_range(4541, 8, r, 8)
        tot += oe - ob
        assert lb == max(ob - 8, 0)
    assert tot == 4541
    lb, ob, oe = sharding.shard_range(4541 * 8, 8, 7, 8)
    assert oe - ob == 4541


def test_trajectory_metrics():
    from lmono_amd import sharding, trajectory
    rng = np.random.default_rng(3)
    n = 200
    q = np.concatenate([rng.normal(0, 0.02, (n, 3)), np.ones((n, 1))], 1)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    incr = np.concatenate([q, rng.normal([0.8, 0, 0], 0.05, (n, 3))], 1)
    incr[0] = sharding.IDENTITY
    poses = sharding.prefix(incr)
    assert trajectory.ate(poses, poses) == 0.0
    r = trajectory.rpe(poses, poses, 1)
    assert r["trans_rmse_m"] < 1e-12 and r["rot_rmse_deg"] < 1e-9 and r["pairs"] == n - 1
    # relative(poses, 1) reproduces the increments
    assert np.abs(trajectory.relative(poses, 1) - incr[1:]).max() < 1e-12
    # a 1 cm shift of one increment: RPE(delta=1) sees it in exactly one pair, ATE in every later pose
    incr2 = incr.copy(); incr2[50, 4] += 0.01
    p2 = sharding.prefix(incr2)
    r = trajectory.rpe(p2, poses, 1)
    assert abs(r["trans_rmse_m"] - 0.01 / np.sqrt(n - 1)) < 1e-9
    assert abs(trajectory.ate(p2, poses) - 0.01 * np.sqrt(150 / 200)) < 1e-9
    # a small yaw error at pose 100 rotates the rest of the trajectory: rotation RPE sees one pair
    th = 1e-3
    incr3 = incr.copy(); incr3[100, :4] = sharding.quat_mul(incr[100, :4], np.array([0, 0, np.sin(th / 2), np.cos(th / 2)]))
    r = trajectory.rpe(sharding.prefix(incr3), poses, 1)
    assert abs(r["rot_rmse_deg"] - np.degrees(th) / np.sqrt(n - 1)) < 1e-9
    st, sr, cnt = trajectory.rpe_from_relative(incr3[1:], incr[1:])
    assert cnt == n - 1 and abs(np.sqrt(sr) - th) < 1e-12


def test_full_sequence_golden_is_consistent():
    """tests/golden/s1_seq00_oracle.npz: 4541 poses, increments compose to the poses, counts plausible."""
    from lmono_amd import sharding
    g = np.load(os.path.join(ROOT, "tests", "golden", "s1_seq00_oracle.npz"))
    assert g["poses"].shape == (4541, 7) and g["incr"].shape == (4541, 7) and g["feat_counts"].shape == (4541, 4)
    assert np.abs(sharding.prefix(g["incr"]) - g["poses"]).max() < 1e-9
    assert (g["n_points"] > 50000).all() and (g["feat_counts"][:, 0] <= 768).all()
