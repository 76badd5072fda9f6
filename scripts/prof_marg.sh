#!/bin/bash
# Phase cycles of k_marginalize on the Estimator's own windows: the -DLMONO_MG_PROF build goes to a SCRATCH directory and is picked up through
# LD_LIBRARY_PATH (estimator_seq carries a RUNPATH, which LD_LIBRARY_PATH precedes) -- the in-tree library is never replaced; 600 frames of the S2 stream; one line per call: tracks anchored at frame 0, their observations, cycles of the factor
# pass / the Schur complement / the Jacobi eigen-decomposition (100 MHz constant clock x 24 = shader cycles at 2.4 GHz: the counter is s_memtime's).
set -e
O=gpurun_out/${1:-mg_prof}; mkdir -p $O
mkdir -p $O/lib
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -DLMONO_MG_PROF -o $O/lib/liblmono_hip.so lmono_amd/csrc/lmono_hip.hip 2>/dev/null
python3 - <<PY
import sys
sys.path.insert(0, '.')
from workloads import s2 as K
st = K.make_stream(600, seed=2, stops=())
K.write_stream('$O/stream600.bin', st)
PY
LD_LIBRARY_PATH=$PWD/$O/lib:$LD_LIBRARY_PATH lmono_amd/host/estimator_seq $O/stream600.bin - sync | grep "^MGPROF" > $O/marg_phases.txt
rm -rf $O/stream600.bin $O/lib
python3 - <<PY
import re
rows=[[int(x) for x in re.findall(r"\d+", l.replace("tred2", "tred"))][1:] for l in open("$O/marg_phases.txt")]      # ([0] is the 0 of the label "F0")
rows.sort(key=lambda r: r[0])
n=len(rows)
print("calls", n, "mean cycles: factor %.0f schur %.0f jacobi %.0f, total %.0f, max %d" % (tuple(sum(r[k] for r in rows)/n for k in (2,3,4)) + (sum(r[2]+r[3]+r[4] for r in rows)/n, max(r[2]+r[3]+r[4] for r in rows))))
for lo,hi in ((0,5),(5,10),(10,20),(20,40),(40,80),(80,161)):
    sel=[r for r in rows if lo<=r[0]<hi]
    if sel:
        m=lambda k: sum(r[k] for r in sel)/len(sel)
        ext = (" | tred2 %.0f accumulate %.0f ql %.0f sweeps %.0f rotations %.0f" % (m(5),m(6),m(7),m(8),m(9))) if len(sel[0]) >= 10 else ""
        ext += (" chain %.0f barrier %.0f" % (m(10), m(11))) if len(sel[0]) >= 12 else ""
        print("F0 in [%d,%d): %d calls, mean obs %.0f, factor %.0f schur %.0f eigen %.0f cycles" % (lo,hi,len(sel),m(1),m(2),m(3),m(4)) + ext)
PY
