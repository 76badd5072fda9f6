#!/usr/bin/env python3
"""Diagnostic: where a k_corr_flat workgroup spends its cycles.  Run on the GPU box through scripts/prof_flat.sh, which builds a
-DLMONO_TILE_PROF library into gpurun_out/ and points LMONO_HIP_LIB at it (s_memtime ticks of thread 0 of every workgroup, 100 MHz)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
assert os.environ.get("LMONO_HIP_LIB"), "run through scripts/prof_flat.sh"
sys.path.insert(0, ROOT)
import numpy as np, torch, lmono_amd
from workloads import s1 as S1
a = [int(v) for v in sys.argv[1:]]
n, chains, lead = (a + [4541, 256, 7])[:3] if len(a) < 3 else a[:3]
w = S1.S1World(n_az=2000)
xyzi, off = w.scans(w.trajectory(n))
ctx = lmono_amd.Context(0)
xd = torch.from_numpy(xyzi).cuda()
batch = lmono_amd.ScanBatch(ctx, n, int(off[-1]))
batch.scanreg(xd.data_ptr(), off, 64, 5.0, keepalive=xd)
incr = torch.zeros((n, 7), dtype=torch.float64, device="cuda")
batch.odometry_d(chains, lead, incr.data_ptr(), None)
ctx.timing_reset()
batch.odometry_d(chains, lead, incr.data_ptr(), None)
groups, _, _ = ctx.timing()
d = ctx.diag
wgs = max(d[0], 1)
names = ["1a requests", "1b resolve + prefix", "2 candidates", "3 decide + vote", "setup"]
tot = sum(d[1:6])
print("workgroups %d: NN rounds %.2f, walk rounds %.2f per workgroup; 4-point chunks per NN round %.0f, per walk round %.0f" %
      (wgs, d[7] / wgs, d[8] / wgs, d[9] / max(d[7], 1), d[10] / max(d[8], 1)))
for i, nm in enumerate(names):
    print("  %-22s %9.0f cycles / workgroup (%4.1f %%)" % (nm, d[1 + i] / wgs, 100 * d[1 + i] / tot))
print("  total %.0f cycles / workgroup" % (tot / wgs))
fine = ["2: chunk search", "2: addresses (LDS)", "2: gathers in flight", "2: arithmetic + flushes", "2: last flush"]
for i, nm in enumerate(fine):
    print("    %-24s %9.0f" % (nm, d[11 + i] / wgs))
print("    gather batches per workgroup (thread 0) %.1f, owner switches %.1f" % (d[16] / wgs, d[17] / wgs))
print("    rounds per workgroup as run (max nearest + max walk) %.2f; max over its features of (nearest + walk) %.2f; mean per feature %.2f" %
      ((d[7] + d[8]) / wgs, d[18] / wgs, d[19] / wgs / 128.0))
print({k: round(v, 3) for k, v in groups.items()})
