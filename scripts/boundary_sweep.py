#!/usr/bin/env python3
"""Self-validating chain schedule (LMONO_OPT_BOUNDARY_TOL) over both bench-scale sequences: the tuned one (seq 0: figure-8,
tests/golden/s1_seq00_oracle.npz) and the held-out one (seq 1: other world, clover trajectory, s1_seq01_oracle.npz).  For every
(chains, lead, lead_full, tol) the odometry stage's wall time, ATE / RPE against the sequential CPU oracle and what the boundary
validation did.
    python scripts/boundary_sweep.py [seq ...] [--cfg chains,lead,lead_full,tol ...]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lmono_amd                          # noqa: E402
from lmono_amd import trajectory          # noqa: E402
from workloads import s1 as S1            # noqa: E402


def load(seq, n):
    if seq == 0:
        w = S1.S1World(n_az=2000); traj = w.trajectory(n)
    else:
        w = S1.S1World(seed=777, n_az=2000); traj = w.trajectory_clover(n)
    return w.scans(traj)


def main():
    seqs, cfgs = [], []
    it = iter(sys.argv[1:])
    for a in it:
        if a == "--cfg":
            cfgs = [tuple(int(v) for v in c.split(",")) for c in it]
        else:
            seqs.append(int(a))
    seqs = seqs or [0, 1]
    if not cfgs:
        cfgs = [(256, 7, 2, 0), (224, 7, 2, 0)]
        cfgs += [(ch, 7, 2, 1000) for ch in (128, 192, 224, 256, 320, 512)]
        cfgs += [(256, ld, 2, 1000) for ld in (3, 4, 5, 6)] + [(256, 5, -1, 1000), (256, 4, -1, 1000), (256, 6, 2, 300), (256, 6, 2, 3000), (256, 6, 2, 10000)]
    ctx = lmono_amd.Context(0)
    for seq in seqs:
        gold = np.load(os.path.join(ROOT, "tests", "golden", "s1_seq%02d_oracle.npz" % seq))
        n = len(gold["poses"])
        xyzi, off = load(seq, n)
        assert (np.diff(off) == gold["n_points"]).all()
        xd = torch.from_numpy(xyzi).cuda()
        del xyzi
        batch = lmono_amd.ScanBatch(ctx, n, int(off[-1]))
        batch.scanreg(xd.data_ptr(), off, 64, 5.0, keepalive=xd)
        cnt = batch.counts()
        print(json.dumps({"seq": seq, "feature_counts_equal": bool((cnt[:, 1:5] == gold["feat_counts"]).all()), "status_or": int(np.bitwise_or.reduce(cnt[:, 5]))}), flush=True)
        incr_d = torch.zeros((n, 7), dtype=torch.float64, device="cuda")
        poses_d = torch.zeros((n, 7), dtype=torch.float64, device="cuda")
        for chains, lead, lead_full, tol in cfgs:
            ctx.set_option(ctx.OPT_LEAD_FULL, lead_full)
            ctx.set_option(ctx.OPT_BOUNDARY_TOL, tol)
            batch.odometry_d(chains, lead, incr_d.data_ptr(), poses_d.data_ptr()); torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                batch.odometry_d(chains, lead, incr_d.data_ptr(), poses_d.data_ptr())
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / reps * 1e3
            p = poses_d.cpu().numpy()
            rep = batch.boundary_report()
            resid = rep.pop("resid"); rerun = rep.pop("rerun")
            r1 = trajectory.rpe(p, gold["poses"], 1)
            q = np.quantile(resid[1:], [0.5, 0.9, 0.99]) if len(resid) > 1 else [0, 0, 0]
            print(json.dumps({"seq": seq, "chains": chains, "lead": lead, "lead_full": lead_full, "tol_1e9": tol, "odometry_ms": round(ms, 2),
                              "ate_m": round(trajectory.ate(p, gold["poses"]), 6), "max_abs_pose_diff": float(np.abs(p - gold["poses"]).max()),
                              "rpe1_m": r1["trans_rmse_m"], "rpe1_deg": r1["rot_rmse_deg"], "report": rep,
                              "resid_q50_q90_q99": [float(v) for v in q], "max_rerun": int(rerun.max())}), flush=True)
        batch.close()
        del xd
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
