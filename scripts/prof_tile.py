#!/usr/bin/env python3
"""Diagnostic: where a k_corr_tile workgroup spends its cycles and why feature points are deferred.
Build (here):  hipcc ... -DLMONO_TILE_PROF -o lmono_amd/lib/liblmono_hip_prof.so   (python scripts/prof_tile.py --build)
Run (GPU box): LMONO_HIP_LIB=lmono_amd/lib/liblmono_hip_prof.so python scripts/prof_tile.py [scans] [chains] [lead]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "lmono_amd", "lib", "liblmono_hip_prof.so")
if "--build" in sys.argv:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
                           "-DLMONO_TILE_PROF", "-o", PROF, "lmono_hip.hip"], cwd=os.path.join(ROOT, "lmono_amd", "csrc"))
    sys.exit(0)
os.environ.setdefault("LMONO_HIP_LIB", PROF)
sys.path.insert(0, ROOT)
import numpy as np            # noqa: E402
import torch                  # noqa: E402
import lmono_amd              # noqa: E402
from workloads import s1 as S1   # noqa: E402

a = [int(v) for v in sys.argv[1:] if v.isdigit()]
n, chains, lead = (a + [1024, 256, 7])[:3] if len(a) < 3 else a[:3]
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
names = ["requests + geometry", "prefix + tables", "bucket table", "point copy", "feature binning", "search"]
print("workgroups %d, features served per workgroup %.1f, deferred %d of %d launches-pairs %d" % (wgs, d[14] / wgs, groups["deferred_features"], d[14], groups["odometry_launch_pairs"]))
tot = sum(d[1:7])
for i, nm in enumerate(names):
    print("  %-16s %9.0f cycles / workgroup  (%4.1f %%)" % (nm, d[1 + i] / wgs, 100 * d[1 + i] / tot))
print("  total %.0f cycles / workgroup" % (tot / wgs))
reasons = ["?", "nn: seeded arc outside window", "nn: first arc outside window", "nn: grown arc outside window", "walk r1 outside", "walk r2 outside", "walk 5 m outside"]
for i, nm in enumerate(reasons):
    print("  deferred %-32s %d" % (nm, d[7 + i]))
for nm, o in (("edge", 15), ("plane", 21)):
    nqd = max(d[o + 5], 1)
    print("  %-5s features %d: NN passes %.2f, NN lines %.1f, NN candidates %.1f, walk passes %.2f, walk candidates %.1f (per feature)" %
          (nm, d[o + 5], d[o + 0] / nqd, d[o + 4] / nqd, d[o + 1] / nqd, d[o + 2] / nqd, d[o + 3] / nqd))
hist = d[27:37]
print('  walk distance needed / r1 (quarters; last = no partner):', [round(float(x) / max(hist.sum(), 1), 3) for x in hist])
print({k: round(v, 3) for k, v in groups.items()})
