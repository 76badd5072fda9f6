"""Two ranks on ONE GPU (gloo rendezvous, both on cuda:0): the sharded odometry + rank-boundary validation + pose exchange path of
bench.py must reproduce, on each rank's owned range, the poses of the unsharded strictly sequential run (tight boundary tolerance).
Launch: python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P scripts/shard_check.py"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lmono_amd                      # noqa: E402
from lmono_amd import sharding        # noqa: E402
from workloads import s1 as O         # noqa: E402  (synthetic generator)


def main():
    rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_total, lead = 16, 3
    w = O.S1World(n_az=500)
    traj = w.trajectory(n_total)
    ctx = lmono_amd.Context(0)
    ctx.set_option(ctx.OPT_BOUNDARY_TOL, 1)          # 1e-9: the repaired shard must equal the sequential run
    dev = torch.device("cuda", 0)

    def run(lo, hi, first):
        xyzi, off = w.scans(traj[lo:hi], scan_id0=lo)
        d = torch.from_numpy(xyzi).to(dev)
        b = lmono_amd.ScanBatch(ctx, hi - lo, len(xyzi))
        b.scanreg(d.data_ptr(), off, keepalive=d)
        incr = torch.zeros((hi - lo, 7), dtype=torch.float64, device=dev)
        b.odometry_shard_main_d(2, lead, first, incr.data_ptr())
        torch.cuda.synchronize()
        rounds = sharding.validate_rank_boundaries(lambda: incr[-1].cpu(), lambda prev: b.shard_validate(prev, incr.data_ptr()), rank, world, deferred=True)
        rep = b.boundary_report()
        print("rank %d: %d rank-boundary round(s), %d chain(s) re-run, %d pair(s)" % (rank, rounds, rep["chains_rerun"], rep["pairs_rerun"]), flush=True)
        poses = torch.zeros((hi - lo - first, 7), dtype=torch.float64, device=dev)
        ctx.pose_prefix_d(incr.data_ptr(), first, hi - lo, poses.data_ptr())
        torch.cuda.synchronize()
        return incr, poses

    lb, ob, oe = sharding.shard_range(n_total, world, rank, lead)
    incr, poses = run(lb, oe, ob - lb)
    bases = sharding.gather_bases(poses[-1].clone().cpu()).to(dev)
    ctx.pose_rebase_d(bases.data_ptr(), rank, poses.data_ptr(), oe - ob)
    torch.cuda.synchronize()
    # reference on the same GPU: the whole sequence in one batch, the strictly sequential schedule
    xyzi, off = w.scans(traj)
    d = torch.from_numpy(xyzi).to(dev)
    b = lmono_amd.ScanBatch(ctx, n_total, len(xyzi))
    b.scanreg(d.data_ptr(), off, keepalive=d)
    _, ref = b.odometry(1, 0)
    err = np.abs(poses.cpu().numpy() - ref[ob:oe]).max()
    print("rank %d owns [%d,%d): max |pose - unsharded| = %.3e" % (rank, ob, oe, err), flush=True)
    ok = torch.tensor([1.0 if err < 1e-8 else 0.0])
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    sys.exit(0 if ok.item() > 0 else 1)


if __name__ == "__main__":
    main()
