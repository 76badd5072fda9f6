"""Loop-closure pose graph on the GPU (lmono_pose_graph_*, SURVEY.md row 8f-2 -- a new feature: the checker is the CPU oracle's
statement of the same graph, not the reference) on S4 graphs: same trust-region path (iterations, accepted / rejected steps),
costs to 1e-9 relative and keyframes to 1e-7 (sin / cos and the elimination order differ in the last bits)."""
import numpy as np
import pytest

from workloads import s4

pytestmark = pytest.mark.gpu


def _compare(oracle, ctx, g, max_iter=5, tol=1e-7):
    import lmono_amd
    pg = lmono_amd.PoseGraph(ctx, g["odom"], g["loops"], g["loop_info"])
    out, st = pg.optimize(max_iter)
    ref, rs = oracle.pose_graph_optimize(g["odom"], g["loops"], g["loop_info"], max_iter=max_iter)
    assert (st["iterations"], st["accepted"], st["rejected"]) == (rs["iterations"], rs["accepted"], rs["rejected"]), (st, rs)
    assert abs(st["initial_cost"] - rs["initial_cost"]) <= 1e-9 * max(rs["initial_cost"], 1e-12)
    assert abs(st["final_cost"] - rs["final_cost"]) <= 1e-9 * max(rs["final_cost"], 1e-12)
    assert np.abs(out[:, :3] - ref[:, :3]).max() < tol and np.abs(np.abs(out[:, 3:]) - np.abs(ref[:, 3:])).max() < tol
    pg.close()
    return out, st


def test_matches_oracle_on_a_two_lap_graph(oracle, gpu_ctx):
    g = s4.make_graph(n=300)
    out, st = _compare(oracle, gpu_ctx, g)
    assert st["final_cost"] < 0.2 * st["initial_cost"] and st["bandwidth"] >= 4
    assert s4.ate(out, g["truth"]) < 0.4 * s4.ate(g["odom"], g["truth"])
    assert np.abs(out[0] - g["odom"][0]).max() < 1e-12


def test_configs3_graph_at_its_size(oracle, gpu_ctx):
    """BASELINE configs[3]'s graph at its named size: 4541 keyframes (KITTI seq 00), the graph `bench.py --workload posegraph` measures
    (2.2 laps, half bandwidth ~67 blocks) against oracle/lo_posegraph.c -- same accepted / rejected steps, costs to 1e-9, keyframes to 1e-7."""
    g = s4.make_graph(n=4541, laps=2.2)
    out, st = _compare(oracle, gpu_ctx, g)
    assert st["bandwidth"] > 40 and len(g["loops"]) > 100
    assert s4.ate(out, g["truth"]) < 0.5 * s4.ate(g["odom"], g["truth"])


def test_more_iterations_outliers_and_short_graphs(oracle, gpu_ctx):
    _compare(oracle, gpu_ctx, s4.make_graph(n=300, outliers=3), max_iter=12)
    _compare(oracle, gpu_ctx, s4.make_graph(n=120, loop_gap=30, seed=3), max_iter=8)
    _compare(oracle, gpu_ctx, s4.make_graph(n=6, loop_gap=3, loop_radius=100.0, loop_every=1, seed=5), max_iter=5)


def test_no_loops_and_bad_arguments(oracle, gpu_ctx):
    import lmono_amd
    g = s4.make_graph(n=80)
    pg = lmono_amd.PoseGraph(gpu_ctx, g["odom"], np.zeros((0, 2)), np.zeros((0, 8)))
    out, st = pg.optimize(5)
    assert st["iterations"] == 0 and st["bandwidth"] == 4 and np.abs(out[:, :3] - g["odom"][:, :3]).max() < 1e-12
    with pytest.raises(lmono_amd.LmonoError):
        lmono_amd.PoseGraph(gpu_ctx, g["odom"], [[0, 80]], np.zeros((1, 8)))          # loop index out of range
    with pytest.raises(lmono_amd.LmonoError):
        lmono_amd.PoseGraph(gpu_ctx, g["odom"][:1], np.zeros((0, 2)), np.zeros((0, 8)))


def test_rank_split_rounds_on_one_gpu(oracle, gpu_ctx):
    """The multi-GPU round structure with two rank objects on one device: the buffers each rank linearises (its own edges only)
    are summed the way the all-reduce would, both ranks step identically, and the result equals the single-rank solve."""
    import torch
    import lmono_amd
    from lmono_amd import sharding
    g = s4.make_graph(n=300)
    single, st = lmono_amd.PoseGraph(gpu_ctx, g["odom"], g["loops"], g["loop_info"]).optimize(5)
    world = 2
    ranks = [lmono_amd.PoseGraph(gpu_ctx, g["odom"], g["loops"], g["loop_info"]) for _ in range(world)]
    bufs = [torch.zeros(pg.reduce_count, dtype=torch.float64, device="cuda:0") for pg in ranks]
    for pg, b in zip(ranks, bufs):
        pg.use_reduce_tensor(b)
    for _ in range(6):
        for r, pg in enumerate(ranks):
            pg.linearise(r, world)
        gpu_ctx.synchronize()
        assert float(bufs[0].abs().sum()) > 0 and float(bufs[1].abs().sum()) > 0
        total = bufs[0] + bufs[1]
        for b in bufs:
            b.copy_(total)
        torch.cuda.synchronize()
        done = [pg.step(5) for pg in ranks]
        assert done[0] == done[1]
        if done[0]:
            break
    for pg in ranks:
        out, s2 = pg.result()
        assert s2["iterations"] == st["iterations"] and np.abs(out - single).max() < 1e-9
    # the driver used by bench.py / multi-GPU runs, world = 1
    pg = lmono_amd.PoseGraph(gpu_ctx, g["odom"], g["loops"], g["loop_info"])
    assert sharding.pose_graph_rounds(pg, 0, 1, max_iter=5) >= 2
    assert np.abs(pg.result()[0] - single).max() == 0.0
