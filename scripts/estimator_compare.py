#!/usr/bin/env python3
"""Frame-by-frame comparison of the host mirror's Estimator loop (GPU) with the CPU oracle on an S2 stream (diagnostic)."""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import estimator_stream as S
from workloads import s2
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
st = s2.make_stream(n, seed=3, stops=(40, 41, 77))
loops = [S.loop_event(st, 60), S.loop_event(st, 95, shift=(-0.03, 0.01, 0.06), yaw=-0.003)] if n > 100 else []
est, log = S.replay_oracle(st, loops)
d = tempfile.mkdtemp()
S.write_stream(os.path.join(d, "s.bin"), st, loops)
out = subprocess.run([os.path.join(ROOT, "lmono_amd", "host", "estimator_seq"), os.path.join(d, "s.bin")], capture_output=True, text=True)
frm = [ln.split()[1:] for ln in out.stdout.splitlines() if ln.startswith("FRM")]
odo = np.array([[float(v) for v in ln.split()[1:]] for ln in out.stdout.splitlines() if ln.startswith("ODO")])
ref = np.array(est.trajectory)
for k, (row, r) in enumerate(zip(frm, log)):
    dp = np.abs(odo[k - 10, 1:4] - ref[k - 10, 1:4]).max() if k >= 10 else 0.0
    print(k, "kf %s/%d st %s/%d it %s/%d term %s/%d cost %.9g/%.9g marg %s,%s/%d,%d feat %s/%d dP %.2e" %
          (row[1], r[0], row[3], r[2], row[4], r[3], row[5], r[4], float(row[6]), r[5], row[7], row[8], r[6], r[7], row[9], r[8], dp))
