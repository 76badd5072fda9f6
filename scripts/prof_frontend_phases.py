"""Per-phase cycle counts of one k_voxel workgroup on an idle GPU: build the library with -DLMONO_VOX_PROF first
(hipcc ... -DLMONO_VOX_PROF -o lmono_amd/lib/liblmono_hip.so lmono_amd/csrc/lmono_hip.hip), run this, rebuild clean."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import lmono_amd
from workloads import s1 as S1
w = S1.S1World(n_az=2000); traj = w.trajectory(8); x, off = w.scans(traj)
ctx = lmono_amd.Context(0); xd = torch.from_numpy(x).cuda()
b = lmono_amd.ScanBatch(ctx, 8, len(x))
for _ in range(2):
    b.scanreg(xd.data_ptr(), off, 64, 5.0, keepalive=xd); ctx.synchronize()
