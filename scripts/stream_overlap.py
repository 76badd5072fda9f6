"""Experiment: the resident sequence cut into G scan ranges, each range on its own HIP stream (front end + odometry of one
range overlap with the other ranges' kernels).  usage: python scripts/stream_overlap.py <scans> <G> [chains]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lmono_amd
from lmono_amd import sharding
from workloads import s1 as S1

n = int(sys.argv[1]); G = int(sys.argv[2]); chains = int(sys.argv[3]) if len(sys.argv) > 3 else 256
lead = 5
w = S1.S1World(n_az=2000)
traj = w.trajectory(n)
groups = []
for g in range(G):
    lb, ob, oe = sharding.shard_range(n, G, g, lead)
    x, off = w.scans(traj[lb:oe], scan_id0=lb)
    ctx = lmono_amd.Context(0)
    st = torch.cuda.Stream()
    ctx.set_stream(st.cuda_stream)
    xd = torch.from_numpy(x).cuda()
    b = lmono_amd.ScanBatch(ctx, oe - lb, len(x))
    incr = torch.zeros((oe - lb, 7), dtype=torch.float64, device="cuda")
    groups.append((ctx, st, xd, off, b, incr, max(1, min(chains // G, oe - lb)), ob - lb))
torch.cuda.synchronize()

def step():
    for ctx, st, xd, off, b, incr, ch, lr in groups:
        b.scanreg(xd.data_ptr(), off, 64, 5.0, keepalive=xd)
    for ctx, st, xd, off, b, incr, ch, lr in groups:
        b.odometry_d(ch, lead, incr.data_ptr(), None)

step(); torch.cuda.synchronize()
t0 = time.perf_counter()
K = 3
for _ in range(K):
    step()
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / K
inc = np.concatenate([g[5].cpu().numpy()[g[7]:] for g in groups])
print("G=%d chains=%d: %.2f ms/step, %.0f scans/s, incr checksum %.9f" % (G, chains, el * 1e3, n / el, float(np.abs(inc).sum())), flush=True)
