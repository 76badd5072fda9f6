"""CPU checks of the pose-graph oracle (oracle/lo_posegraph.c; SURVEY.md row 8f-2 -- a new feature without a parity target in
the reference): helper conventions against the in-tree formulas, band vs dense elimination, the effect of loops on a drifting
trajectory, robustness to false loops, and the rank-split rounds."""
import numpy as np
import pytest

from oracle import oracle as O
from workloads import s4


def test_euler_helpers_follow_the_reference_conventions():
    # YawPitchRollToRotationMatrix (Loop_Detector.h:129-147) = Rz(y) Ry(p) Rx(r), degrees; R2ypr (math_utils.h:187-202) inverts it
    for ypr in ([30.0, 10.0, -5.0], [-170.0, 2.0, 1.0], [95.0, -20.0, 15.0], [0.0, 0.0, 0.0]):
        q = O.ypr2q(ypr)
        assert abs(np.linalg.norm(q) - 1) < 1e-12
        assert np.allclose(O.q2ypr(q), ypr, atol=1e-9)
        R = s4._rot(ypr)
        assert np.allclose(np.abs(q), np.abs(s4._quat(R)), atol=1e-12)


def test_no_loops_leaves_odometry_untouched():
    g = s4.make_graph(n=60)
    out, st = O.pose_graph_optimize(g["odom"], np.zeros((0, 2)), np.zeros((0, 8)))
    assert st["iterations"] == 0 and st["initial_cost"] < 1e-20 and st["bandwidth"] == 4
    assert np.abs(out[:, :3] - g["odom"][:, :3]).max() < 1e-12
    assert np.abs(np.abs(out[:, 3:]) - np.abs(g["odom"][:, 3:])).max() < 1e-9


def test_band_elimination_equals_dense_and_loops_remove_drift():
    g = s4.make_graph(n=240)
    assert len(g["loops"]) > 15
    out, st = O.pose_graph_optimize(g["odom"], g["loops"], g["loop_info"], max_iter=5)
    dense, st2 = O.pose_graph_optimize(g["odom"], g["loops"], g["loop_info"], max_iter=5, ordering=1)
    assert st["bandwidth"] < 80 and st2["bandwidth"] == 239
    assert st["iterations"] == st2["iterations"] and np.abs(out - dense).max() < 1e-9
    assert st["final_cost"] < 0.2 * st["initial_cost"]
    before, after = s4.ate(g["odom"], g["truth"]), s4.ate(out, g["truth"])
    assert after < 0.4 * before, (before, after)
    assert np.abs(out[0] - g["odom"][0]).max() < 1e-12               # keyframe 0 is the gauge
    # pitch / roll are not optimised: the rotation changes about the world z axis only
    for k in (50, 200):
        assert np.allclose(O.q2ypr(out[k, 3:])[1:], O.q2ypr(g["odom"][k, 3:])[1:], atol=1e-9)


def test_huber_loss_contains_false_loops():
    g = s4.make_graph(n=240, outliers=3)
    clean = s4.make_graph(n=240)
    out, st = O.pose_graph_optimize(g["odom"], g["loops"], g["loop_info"], max_iter=20)
    ref, _ = O.pose_graph_optimize(clean["odom"], clean["loops"], clean["loop_info"], max_iter=20)
    # three 25 m false positives among ~25 loops shift the solution by decimetres, not by metres
    assert s4.ate(out, g["truth"]) < s4.ate(ref, g["truth"]) + 0.6
    assert s4.ate(out, g["truth"]) < s4.ate(g["odom"], g["truth"])


@pytest.mark.parametrize("world", [2, 3])
def test_rank_split_rounds_equal_the_single_rank_solve(world):
    g = s4.make_graph(n=150, loop_gap=30)
    ref, st = O.pose_graph_optimize(g["odom"], g["loops"], g["loop_info"], max_iter=5)
    ranks = [O.PoseGraph(g["odom"], g["loops"], g["loop_info"]) for _ in range(world)]
    for _ in range(6):
        for r, pg in enumerate(ranks):
            pg.linearise(r, world)
        total = sum(pg.reduce_tensor for pg in ranks)
        done = []
        for pg in ranks:
            pg.reduce_tensor[:] = total
            done.append(pg.step(5))
        assert len(set(done)) == 1
        if done[0]:
            break
    for pg in ranks:
        out, s2 = pg.result()
        assert s2["iterations"] == st["iterations"] and np.abs(out - ref).max() < 1e-9
