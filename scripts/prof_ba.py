# Per-phase cycle counts of k_ba_solve (window 0): build the library with -DLMONO_BA_PROF first, e.g.
#   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -DLMONO_BA_PROF -Iinclude lmono_amd/csrc/lmono_hip.hip -o lmono_amd/lib/liblmono_hip.so
# then: python scripts/prof_ba.py [n_windows]
import sys, numpy as np
sys.path.insert(0, '.')
import lmono_amd
from tests import ba_cases as K
ctx = lmono_amd.Context(0)
base = [K.make_window(s) for s in range(16)]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
b = lmono_amd.BaBatch(ctx, [base[k % 16] for k in range(n)])
b.solve(30); ctx.synchronize()
