"""laserMapping with the device-resident cube map over a synthetic S1 sequence: frames/s of lmono_mapper_process (one
stream), ATE of the odometry and of the mapped trajectory against ground truth, and the CPU oracle's mapping time on the
same input (test infrastructure, used here as the baseline only).   python scripts/bench_mapper.py [n_scans]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
import lmono_amd
from workloads import s1 as S1

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
w = S1.S1World(); traj = w.trajectory(n); x, off = w.scans(traj)
ctx = lmono_amd.Context(0)
xd = torch.from_numpy(x).cuda()
batch = lmono_amd.ScanBatch(ctx, n, len(x))
batch.scanreg(xd.data_ptr(), off, 64, 5.0, keepalive=xd)
_, odo = batch.odometry(n_chains=1, lead=0)
mapper = lmono_amd.Mapper(ctx)
got = np.zeros((n, 7))
ctx.synchronize()
t0 = time.perf_counter()
for k in range(n):
    q, t, st = mapper.process(batch, k, odo[k, :4], odo[k, 4:])
    got[k, :4] = q; got[k, 4:] = t
ctx.synchronize()
el = time.perf_counter() - t0
from oracle import oracle as O          # baseline / ground-truth helpers only
gt = O.gt_relative(traj)
print("%d scans: lmono_mapper_process %.2f ms per frame (%.0f frames/s, one stream, wall clock incl. the per-frame host round trips)" % (n, el / n * 1e3, n / el))
print("ATE vs ground truth: odometry %.3f m, mapped %.3f m; last frame blocks: %d edges, %d planes" % (O.ate(odo, gt), O.ate(got, gt), st[1], st[3]))
t0 = time.time(); ref = O.run_mapping(x, off, odo); cpu = time.time() - t0
print("CPU oracle (1 thread): scanreg %.0f ms + mapping %.1f ms per frame; max |pose difference| to the GPU mapper %.2e" % (ref["stage_ms"][0] / n, ref["stage_ms"][1] / n, np.abs(ref["poses"] - got).max()))
