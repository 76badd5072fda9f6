#!/bin/bash
# Phase cycles of k_ba_solve on the Estimator's own windows: the -DLMONO_BA_PROF build replaces the in-tree library ON THE GPU BOX's copy of the tree
# (estimator_seq loads it by rpath), 120 frames of the S2 stream, the last windows' PROF lines.
set -e
mkdir -p gpurun_out/ba_prof
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -DLMONO_BA_PROF -o lmono_amd/lib/liblmono_hip.so lmono_amd/csrc/lmono_hip.hip 2>/dev/null
python3 - <<'PY'
import sys, os
sys.path.insert(0, '.')
from workloads import s2 as K
st = K.make_stream(120, seed=2, stops=())
K.write_stream('gpurun_out/ba_prof/stream120.bin', st)
PY
lmono_amd/host/estimator_seq gpurun_out/ba_prof/stream120.bin - sync | grep "^PROF" | tail -5 > gpurun_out/ba_prof/phases_ba_seq.txt
rm -f gpurun_out/ba_prof/stream120.bin
cat gpurun_out/ba_prof/phases_ba_seq.txt
