#!/usr/bin/env python3
"""Full-sequence tolerance of the chained odometry schedule: ATE / RPE of (chains, lead) runs over all 4541 S1 scans against
the committed sequential-oracle trajectory (tests/golden/s1_seq00_oracle.npz), with the odometry stage's wall time.
    python scripts/lead_sweep.py [chains,lead ...]      e.g. 256,5 256,8 128,6"""
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


def main():
    cfgs = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(256, l) for l in (4, 5, 6, 7, 8, 10)] + [(128, 6), (128, 8), (512, 8)]
    gold = np.load(os.path.join(ROOT, "tests", "golden", "s1_seq00_oracle.npz"))
    n = len(gold["poses"])
    w = S1.S1World(n_az=2000)
    xyzi, off = w.scans(w.trajectory(n))
    ctx = lmono_amd.Context(0)
    xd = torch.from_numpy(xyzi).cuda()
    del xyzi
    batch = lmono_amd.ScanBatch(ctx, n, int(off[-1]))
    batch.scanreg(xd.data_ptr(), off, 64, 5.0, keepalive=xd)
    cnt = batch.counts()
    print(json.dumps({"feature_counts_equal": bool((cnt[:, 1:5] == gold["feat_counts"]).all()), "status_or": int(np.bitwise_or.reduce(cnt[:, 5]))}), flush=True)
    incr_d = torch.zeros((n, 7), dtype=torch.float64, device="cuda")
    poses_d = torch.zeros((n, 7), dtype=torch.float64, device="cuda")
    for chains, lead in [(1, 0)] + cfgs:
        batch.odometry_d(chains, lead, incr_d.data_ptr(), poses_d.data_ptr()); torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 1 if chains == 1 else 3
        for _ in range(reps):
            batch.odometry_d(chains, lead, incr_d.data_ptr(), poses_d.data_ptr())
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        p = poses_d.cpu().numpy()
        r1 = trajectory.rpe(p, gold["poses"], 1); r100 = trajectory.rpe(p, gold["poses"], 100)
        print(json.dumps({"chains": chains, "lead": lead, "odometry_ms": round(ms, 2), "ate_m": round(trajectory.ate(p, gold["poses"]), 6),
                          "max_abs_pose_diff": float(np.abs(p - gold["poses"]).max()),
                          "rpe1_m": r1["trans_rmse_m"], "rpe1_deg": r1["rot_rmse_deg"], "rpe100_m": r100["trans_rmse_m"], "rpe100_deg": r100["rot_rmse_deg"]}), flush=True)


if __name__ == "__main__":
    main()
