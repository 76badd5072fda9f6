#!/usr/bin/env python3
"""Phase stamps of k_odom_chain (diagnostic build -DLMONO_OC_PROF, LMONO_HIP_LIB): s_memtime ticks (100 MHz) of the search and solve phases
per outer iteration, summed over chains.   python scripts/prof_chain.py [scans] [chains] [lead]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lmono_amd                          # noqa: E402
from workloads import s1 as S1            # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
chains = int(sys.argv[2]) if len(sys.argv) > 2 else 32
lead = int(sys.argv[3]) if len(sys.argv) > 3 else 0
w = S1.S1World(n_az=2000)
xyzi, off = w.scans(w.trajectory(n))
ctx = lmono_amd.Context(0)
xd = torch.from_numpy(xyzi).cuda()
b = lmono_amd.ScanBatch(ctx, n, int(off[-1]))
b.scanreg(xd.data_ptr(), off, 64, 5.0, keepalive=xd)
ctx.set_option(ctx.OPT_BOUNDARY_TOL, 0)
for ch in (1, chains):
    b.odometry(ch, lead)
    ctx.timing_reset()
    b.odometry(ch, lead)
    t, _, _ = ctx.timing()
    d = ctx.diag
    outers = max(d[0], 1)
    print("chains %d: odometry %.2f ms; %d outer iterations; per outer: search %.1f us, solve %.1f us (s_memtime at 100 MHz)"
          % (ch, t["odometry_total"], outers, d[1] / outers / 100.0, d[2] / outers / 100.0))
