"""Host logic for sharding a scan sequence over the GPUs of one node.

Each rank owns a contiguous range of scans plus a `lead`-scan lead-in; the only exchange step is ONE
all-gather (RCCL over xGMI, 56 B per rank) of every rank's cumulative transform, after which each rank re-bases
its poses locally.  The pose algebra below (numpy, fp64) mirrors k_pose_prefix / k_pose_rebase and follows
laserOdometry's accumulation t_w += q_w * t ; q_w = q_w * q (quaternions x,y,z,w)."""
import numpy as np


def shard_range(n_total, world, rank, lead):
    """Returns (load_begin, own_begin, own_end): rank processes scans [load_begin, own_end), owns [own_begin, own_end)."""
    own_begin = rank * n_total // world
    own_end = (rank + 1) * n_total // world
    load_begin = max(own_begin - lead, 0)
    return load_begin, own_begin, own_end


def quat_rotate(q, v):
    u = q[:3]; w = q[3]
    uv = 2.0 * np.cross(u, v)
    return v + w * uv + np.cross(u, uv)


def quat_mul(a, b):
    ax, ay, az, aw = a; bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx,
                     aw * bw - ax * bx - ay * by - az * bz])


def compose(a, b):
    """a (+) b for 7-vectors (q xyzw, t)."""
    out = np.empty(7)
    out[4:] = a[4:] + quat_rotate(a[:4], b[4:])
    out[:4] = quat_mul(a[:4], b[:4])
    return out


IDENTITY = np.array([0, 0, 0, 1, 0, 0, 0], np.float64)


def prefix(incr, first=0):
    """poses[k-first] = incr[first] (+) ... (+) incr[k]; incr[0] is the identity, so first = 0 is relative to scan 0
    and first > 0 relative to scan first-1 (the previous rank's last scan)."""
    n = len(incr) - first
    out = np.empty((n, 7))
    cur = IDENTITY.copy()
    for k in range(n):
        cur = compose(cur, incr[first + k])
        out[k] = cur
    return out


def rebase(bases, poses):
    base = IDENTITY.copy()
    for b in bases:
        base = compose(base, b)
    return np.stack([compose(base, p) for p in poses]) if len(poses) else poses


def gather_bases(my_base, group=None):
    """All-gather of each rank's cumulative transform: torch tensor [7] float64 (CPU for gloo, GPU for RCCL)
    -> [world, 7].  my_base = T(last scan of the previous rank -> my last scan) = prefix(...)[-1]."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    out = torch.empty(world * 7, dtype=torch.float64, device=my_base.device)
    dist.all_gather_into_tensor(out, my_base.contiguous().reshape(7), group=group)
    return out.view(world, 7)


def validate_rank_boundaries(last_incr, validate, rank, world, group=None, max_rounds=None, deferred=False):
    """The rank boundaries of a scan-range shard checked like the chain boundaries inside a rank (lmono_odom_shard_validate): every rank
    publishes its LAST increment (one all-gather of 7 doubles), rank r > 0 hands rank r-1's to `validate(prev_incr [7] numpy) -> bool`
    (True when its own last increment changed by the repair), and the round repeats while any rank reports a change (one all-reduce of a
    flag) -- a repair rarely reaches the end of a rank's range, so this is one round in practice.  last_incr() -> torch tensor [7]
    float64, the rank's current last increment (on the GPU for RCCL, on the CPU for gloo).  deferred: the ranks ran the main pass only
    (lmono_odom_shard_main_d); the first round's validate() is then every rank's WHOLE validation -- rank 0 calls validate(None) -- so that the
    rank boundary is repaired in the same rounds as the chain boundaries inside the rank.  Returns the number of rounds."""
    import torch
    import torch.distributed as dist
    rounds = 0
    limit = max_rounds if max_rounds is not None else world
    while rounds < limit:
        mine = last_incr().contiguous().reshape(7)
        last = torch.empty(world * 7, dtype=torch.float64, device=mine.device)
        dist.all_gather_into_tensor(last, mine, group=group)
        changed = False
        if rank > 0:
            changed = bool(validate(last.view(world, 7)[rank - 1].cpu().numpy()))
        elif deferred and rounds == 0:
            changed = bool(validate(None))
        flag = torch.tensor([1.0 if changed else 0.0], dtype=torch.float64, device=mine.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
        rounds += 1
        if float(flag.item()) == 0.0:
            break
    return rounds


def pose_graph_rounds(graph, rank, world, max_iter=5, all_reduce=None, force_collective=False):
    """Loop-closure pose graph over `world` ranks: every rank holds the same graph object (lmono_amd.PoseGraph
    on a GPU; anything with linearise(rank, world) / reduce_tensor / step(max_iter) works), linearises the edges it owns, ONE
    all-reduce sums the normal equations [H | g | cost] (RCCL over xGMI with the nccl backend), and every rank takes the same
    trust-region step.  force_collective: run the all-reduce with one rank as well (the RCCL path on a one-GPU box; the sum over one
    rank is the identity, so the result is the no-collective run's).  Returns the number of rounds run."""
    rounds = 0
    for _ in range(max_iter + 1):
        graph.linearise(rank, world)
        if world > 1 or force_collective:
            if all_reduce is None:
                import torch.distributed as dist
                dist.all_reduce(graph.reduce_tensor)
            else:
                all_reduce(graph.reduce_tensor)
        rounds += 1
        if graph.step(max_iter):
            break
    return rounds
