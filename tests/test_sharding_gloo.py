"""N > 1 path on CPU: world_size-2 gloo run of the scan-range sharding + single all-gather pose exchange."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_total, lead, incr_all, ret):
    sys.path.insert(0, ROOT)
    from lmono_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lb, ob, oe = sharding.shard_range(n_total, world, rank, lead)
    # each rank "computes" the increments of its loaded range; lead-in rows are garbage on purpose
    local = incr_all[lb:oe].copy()
    local[: ob - lb] = 123.0
    poses = sharding.prefix(local, first=ob - lb)
    bases = sharding.gather_bases(torch.from_numpy(poses[-1].copy()))
    glob = sharding.rebase(bases[:rank].numpy(), poses)
    ret[rank] = (ob, oe, glob)
    dist.barrier()
    dist.destroy_process_group()


def _rand_incr(n, seed):
    rng = np.random.default_rng(seed)
    q = np.concatenate([rng.normal(0, 0.02, (n, 3)), np.ones((n, 1))], 1)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    t = rng.normal([0.8, 0, 0], 0.05, (n, 3))
    out = np.concatenate([q, t], 1)
    out[0] = [0, 0, 0, 1, 0, 0, 0]
    return out


def test_two_rank_pose_exchange_equals_single_process():
    from lmono_amd import sharding
    n_total, lead, world = 37, 5, 2
    incr = _rand_incr(n_total, 0)
    ref = sharding.prefix(incr)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, 29517, n_total, lead, incr, ret), nprocs=world, join=True)
    for rank in range(world):
        ob, oe, glob = ret[rank]
        assert np.abs(glob - ref[ob:oe]).max() < 1e-12


def test_shard_ranges_cover_sequence():
    from lmono_amd import sharding
    for n, w, lead in ((4541, 8, 5), (10, 4, 5), (7, 2, 0)):
        prev_end = 0
        for r in range(w):
            lb, ob, oe = sharding.shard_range(n, w, r, lead)
            assert ob == prev_end and lb == max(ob - lead, 0)
            prev_end = oe
        assert prev_end == n


def _pg_worker(rank, world, port, graph, max_iter, ret):
    """One rank of the pose-graph rounds: the product's driver (lmono_amd.sharding.pose_graph_rounds) over gloo, with the CPU
    oracle's graph object standing in for the GPU one (same linearise / reduce_tensor / step interface)."""
    sys.path.insert(0, ROOT)
    from lmono_amd import sharding
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pg = O.PoseGraph(graph["odom"], graph["loops"], graph["loop_info"])
    pg.reduce_tensor = torch.from_numpy(pg.reduce_tensor)          # shares memory with the array the oracle fills / reads
    calls = []

    def all_reduce(t):
        calls.append(t.numel())
        dist.all_reduce(t)
    rounds = sharding.pose_graph_rounds(pg, rank, world, max_iter=max_iter, all_reduce=all_reduce)
    out, st = pg.result()
    ret[rank] = (out, st, rounds, calls, float(pg.reduce_tensor.abs().sum()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_pose_graph_rounds_equal_single_process():
    """edges sharded by keyframe range, ONE all-reduce of the normal equations per round."""
    from oracle import oracle as O
    from workloads import s4
    g = s4.make_graph(n=160, loop_gap=30)
    ref, st = O.pose_graph_optimize(g["odom"], g["loops"], g["loop_info"], max_iter=5)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_pg_worker, args=(2, 29531, g, 5, ret), nprocs=2, join=True)
    for rank in range(2):
        out, s2, rounds, calls, _ = ret[rank]
        assert s2["iterations"] == st["iterations"] and abs(s2["final_cost"] - st["final_cost"]) < 1e-12
        assert np.abs(out - ref).max() < 1e-9
        assert len(calls) == rounds and rounds <= 6 and all(c == calls[0] for c in calls)      # one all-reduce per round
    assert np.array_equal(ret[0][0], ret[1][0])                     # both ranks hold bit-identical keyframes


def _vb_worker(rank, world, port, ret, deferred=False):
    """Rank-boundary validation rounds (lmono_amd.sharding.validate_rank_boundaries) over gloo with a stand-in for the GPU batch:
    `ws` = the rank's own lead-in estimate of the previous rank's last increment, a repair adopts the published one, and on rank 1 the
    repair reaches the end of the rank's range (its last increment changes -> rank 2 must validate again)."""
    sys.path.insert(0, ROOT)
    from lmono_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tol = 1e-6
    st = {"last": np.array([0, 0, 0, 1, 0.8 + 0.01 * rank, 0, 0.0]), "ws": np.array([0, 0, 0, 1, 0.8 + 0.01 * (rank - 1) + 1e-4, 0, 0.0]), "calls": 0, "repairs": 0}

    def validate(prev):
        st["calls"] += 1
        if prev is None:                            # deferred flow, rank 0: the rank's inner boundaries only
            st["inner_only"] = st.get("inner_only", 0) + 1
            return False
        if np.abs(prev - st["ws"]).max() <= tol:
            return False
        st["ws"] = prev.copy(); st["repairs"] += 1
        if rank == 1 and st["repairs"] == 1:
            st["last"] = st["last"] + 1e-3          # the repair ran to the end of this rank's range
            return True
        return False
    rounds = sharding.validate_rank_boundaries(lambda: torch.from_numpy(st["last"].copy()), validate, rank, world, deferred=deferred)
    ret[rank] = (rounds, st["calls"], st["repairs"], st["ws"], st["last"], st.get("inner_only", 0))
    dist.barrier()
    dist.destroy_process_group()


def test_rank_boundary_validation_rounds_over_gloo():
    world = 3
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_vb_worker, args=(world, 29541, ret), nprocs=world, join=True)
    rounds = [ret[r][0] for r in range(world)]
    assert rounds == [2, 2, 2]                       # round 1 repairs ranks 1 and 2, rank 1's change sends rank 2 round again
    assert ret[0][1] == 0 and ret[1][1] == 2 and ret[2][1] == 2
    assert ret[1][2] == 1 and ret[2][2] == 2
    for r in (1, 2):
        assert np.array_equal(ret[r][3], ret[r - 1][4])      # every rank's warm start is its predecessor's final last increment


def test_deferred_rank_boundary_validation_calls_every_rank_once():
    """Deferred flow (lmono_odom_shard_main_d): the first round's validate() is every rank's whole validation -- rank 0 gets prev = None
    exactly once -- and the later rounds are the rank boundaries' as before."""
    world = 3
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_vb_worker, args=(world, 29543, ret, True), nprocs=world, join=True)
    assert [ret[r][0] for r in range(world)] == [2, 2, 2]
    assert ret[0][1] == 1 and ret[0][5] == 1 and ret[0][2] == 0          # rank 0: one call, with None, no repair of an external boundary
    assert ret[1][1] == 2 and ret[2][1] == 2 and ret[1][5] == 0 and ret[2][5] == 0
    for r in (1, 2):
        assert np.array_equal(ret[r][3], ret[r - 1][4])
