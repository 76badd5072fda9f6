"""Experiment: two-stage software pipeline over bench steps -- scanRegistration of step i+1 (stream F) beside laserOdometry of
step i (stream O and its chain-group streams), two ScanBatch working sets.  usage: python scripts/pipeline_probe.py [scans] [steps]"""
import os, sys, time
import ctypes as C
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lmono_amd
from workloads import s1 as S1

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4541
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
chains, lead = 256, 6
w = S1.S1World(n_az=2000)
traj = w.trajectory(n)
xyzi, off = w.scans(traj, scan_id0=0)
total = int(off[-1])
dev = torch.device("cuda", 0)
xd = torch.from_numpy(xyzi).to(dev)
del xyzi

def run(pipelined):
    sF, sO = torch.cuda.Stream(), torch.cuda.Stream()
    cF, cO = lmono_amd.Context(0), lmono_amd.Context(0)
    cO.set_option(cO.OPT_LEAD_FULL, 3)
    cF.set_stream(sF.cuda_stream)
    cO.set_stream(sO.cuda_stream if pipelined else sF.cuda_stream)
    nb = 2 if pipelined else 1
    bs = [lmono_amd.ScanBatch(cF, n, total) for _ in range(nb)]
    incr = [torch.zeros((n, 7), dtype=torch.float64, device=dev) for _ in range(nb)]
    poses = [torch.zeros((n, 7), dtype=torch.float64, device=dev) for _ in range(nb)]
    evF = [torch.cuda.Event() for _ in range(nb)]
    evO = [torch.cuda.Event() for _ in range(nb)]
    so = sO if pipelined else sF
    def step(i):
        k = i % nb
        b = bs[k]
        if pipelined and i >= nb:
            sF.wait_event(evO[k])
        b.scanreg(xd.data_ptr(), off, 64, 5.0, keepalive=xd)
        if pipelined:
            evF[k].record(sF); so.wait_event(evF[k])
        cO.check(cO.L.lmono_odom_batch_d(cO.h, b.h, chains, lead, C.c_void_p(incr[k].data_ptr()), None))
        cO.pose_prefix_d(incr[k].data_ptr(), 0, n, poses[k].data_ptr())
        if pipelined:
            evO[k].record(so)
    for i in range(2):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        step(i)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / K
    chk = [float(p[-1].abs().sum().item()) for p in poses]
    print("pipelined=%d: %.2f ms/step, %.0f scans/s, final pose checksum %s" % (pipelined, el * 1e3, n / el, chk), flush=True)

run(False)
run(True)
