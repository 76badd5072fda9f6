"""Generates tests/golden/ba_factors.npz: 20 seeded evaluations per BA factor kind (inputs + residuals + Jacobians)
from the numpy restatement oracle/ba_numpy.py of the reference's literal formulae (the reference itself cannot be
built or imported here: C++ with ROS/Ceres/Eigen dependencies).  Run from the repo root:
    python tests/golden/make_ba_goldens.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import ba_cases as K                      # noqa: E402
from tests.test_ba_factors_cpu import _np_eval, CASES   # noqa: E402

out = {}
for kind in range(4):
    P, Cn, info = CASES[kind](20, seed=100 + kind)
    r, J = _np_eval(kind, P, Cn, info)
    out.update({"P%d" % kind: P, "C%d" % kind: Cn, "I%d" % kind: np.asarray(info, np.float64), "r%d" % kind: r, "J%d" % kind: J})
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ba_factors.npz"), **out)
print("written", {k: v.shape for k, v in out.items()})
