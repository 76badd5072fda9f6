# debug helper: solve a set of test windows several times (fresh batch each time) and report whether the results agree with each other.
# usage: python scripts/dbg_ba_cluster.py <case> <K> [repeats]
import sys
sys.path.insert(0, '.')
import numpy as np
import lmono_amd
from tests import ba_cases as K
case, k = sys.argv[1], int(sys.argv[2])
rep = int(sys.argv[3]) if len(sys.argv) > 3 else 6
ctx = lmono_amd.Context(0)
if hasattr(ctx, "OPT_BA_CLUSTER"):
    ctx.set_option(ctx.OPT_BA_CLUSTER, k)
if case == "plain": ws = [K.make_window(0)]
elif case == "seven":
    ws = [K.make_window(s) for s in range(6)]; ws[4]["use_mono"] = False; ws[5]["ex_constant"] = True; ws.append(K.make_window(7, n_frames=6))
elif case == "six": ws = [K.make_window(s) for s in range(6)]
elif case == "two": ws = [K.make_window(0), K.make_window(1)]
first = None
for r in range(rep):
    b = lmono_amd.BaBatch(ctx, ws)
    b.solve(30)
    p, e, d, sm = b.read()
    key = (p.tobytes(), sm.tobytes())
    if first is None: first = key
    print(case, k, "run", r, "same as run 0" if key == first else "DIFFERENT", [int(x) for x in sm[:, 2]], [int(x) for x in sm[:, 3]])
