#!/bin/bash
# Phase cycles of one k_voxel workgroup (ring 20 of scan 3) in a -DLMONO_VOX_PROF build made on the box; idle GPU (8 scans) and loaded (512 scans).
set -e
mkdir -p gpurun_out/vox_prof
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -DLMONO_VOX_PROF -o gpurun_out/vox_prof/prof.so lmono_amd/csrc/lmono_hip.hip 2>/dev/null
export LMONO_HIP_LIB=$PWD/gpurun_out/vox_prof/prof.so      # a scratch library: the product library is never overwritten
timeout -k 10 120 python - <<'PY' | tee gpurun_out/vox_prof/phases.txt
import sys, numpy as np, torch
sys.path.insert(0, '.')
import lmono_amd
from workloads import s1 as S1
for n in (8, 512):
    w = S1.S1World(n_az=2000); traj = w.trajectory(n); x, off = w.scans(traj)
    ctx = lmono_amd.Context(0); xd = torch.from_numpy(x).cuda()
    b = lmono_amd.ScanBatch(ctx, n, len(x))
    print("scans", n, flush=True)
    for _ in range(2):
        b.scanreg(xd.data_ptr(), off, 64, 5.0, keepalive=xd); ctx.synchronize()
PY
rm -f gpurun_out/vox_prof/prof.so
