"""Does the front end (HBM-bound) hide behind the odometry (gather-latency-bound) when both run at once?  Two resident batches, two
contexts on their own streams, two host threads: scanreg(B) alone, odometry(A) alone, both together.  usage: python scripts/overlap_probe.py [scans]"""
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import lmono_amd
from workloads import s1 as S1

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
chains = int(sys.argv[2]) if len(sys.argv) > 2 else 120
w = S1.S1World(n_az=2000)
traj = w.trajectory(n)
xyzi, off = w.scans(traj, scan_id0=0)
dev = torch.device("cuda:0")
xd = torch.from_numpy(xyzi).to(dev)
total = int(off[-1])
ctxs, batches, incrs = [], [], []
for k in range(2):
    c = lmono_amd.Context(0)
    c.use_own_stream()
    ctxs.append(c)
    batches.append(lmono_amd.ScanBatch(c, n, total))
    incrs.append(torch.zeros((n, 7), dtype=torch.float64, device=dev))


def reg(k):
    batches[k].scanreg(xd.data_ptr(), off, 64, 5.0, keepalive=xd)
    ctxs[k].synchronize()


def odo(k):
    batches[k].odometry_d(chains, 4, incrs[k].data_ptr(), None)
    ctxs[k].synchronize()


def timed(fns, reps=3):
    best = 1e9
    for _ in range(reps):
        th = [threading.Thread(target=f) for f in fns]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


reg(0); reg(1); odo(0); odo(1)
ref = incrs[0].clone()
t_reg = timed([lambda: reg(1)])
t_odo = timed([lambda: odo(0)])
t_both = timed([lambda: reg(1), lambda: odo(0)])
t_2reg = timed([lambda: reg(1), lambda: reg(0)])
odo(0); odo(1)
t_2odo = timed([lambda: odo(1), lambda: odo(0)])
print("scans %d chains %d: scanreg %.2f ms, odometry %.2f ms, sum %.2f, together %.2f ms; two scanregs %.2f, two odometries %.2f"
      % (n, chains, t_reg, t_odo, t_reg + t_odo, t_both, t_2reg, t_2odo))
print("odometry result unchanged:", bool(torch.equal(ref, incrs[0])))
