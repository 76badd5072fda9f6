#!/bin/bash
# Phase cycles of k_ba_solve on the Estimator's own windows: the -DLMONO_BA_PROF build goes to a scratch directory and is loaded through
# LD_LIBRARY_PATH (it precedes estimator_seq's RUNPATH; the in-tree library is never replaced), 120 frames of the S2 stream, the last windows' PROF lines.
set -e
mkdir -p gpurun_out/ba_prof/lib
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -DLMONO_BA_PROF -o gpurun_out/ba_prof/lib/liblmono_hip.so lmono_amd/csrc/lmono_hip.hip 2>/dev/null
python3 - <<'PY'
import sys, os
sys.path.insert(0, '.')
from workloads import s2 as K
st = K.make_stream(120, seed=2, stops=())
K.write_stream('gpurun_out/ba_prof/stream120.bin', st)
PY
LD_LIBRARY_PATH=$PWD/gpurun_out/ba_prof/lib:$LD_LIBRARY_PATH lmono_amd/host/estimator_seq gpurun_out/ba_prof/stream120.bin - sync | grep "^PROF" | tail -5 > gpurun_out/ba_prof/phases_ba_seq.txt
rm -rf gpurun_out/ba_prof/stream120.bin gpurun_out/ba_prof/lib
cat gpurun_out/ba_prof/phases_ba_seq.txt
