#!/bin/bash
# Threads of the two marginalisation kernels (LMONO_MG_T) against the Estimator frame loop (ba-seq drives the C++ host binary, which finds the
# variant in a scratch directory through LD_LIBRARY_PATH / LMONO_HIP_LIB; the in-tree library is never replaced).  usage (GPU box): bash scripts/marg_threads_sweep.sh
set -u
mkdir -p gpurun_out/mg_t/lib
export LD_LIBRARY_PATH=$PWD/gpurun_out/mg_t/lib:${LD_LIBRARY_PATH:-}
export LMONO_HIP_LIB=$PWD/gpurun_out/mg_t/lib/liblmono_hip.so
: > gpurun_out/marg_threads_sweep.txt
for T in 256 512 1024; do
  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -DLMONO_MG_T=$T -o gpurun_out/mg_t/lib/liblmono_hip.so lmono_amd/csrc/lmono_hip.hip 2>/dev/null || { echo "T=$T build failed" >> gpurun_out/marg_threads_sweep.txt; continue; }
  timeout -k 10 300 python3 -m pytest tests/test_marg_gpu.py -m gpu -x -q 2>&1 | tail -1 >> gpurun_out/marg_threads_sweep.txt
  timeout -k 10 300 python3 bench.py --workload ba-seq --cpu-frames 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('LMONO_MG_T=$T:', d['value'], 'frames/s overlapped,', d['config']['inline_marginalisation']['frames_per_s'], 'inline')" >> gpurun_out/marg_threads_sweep.txt
done
rm -rf gpurun_out/mg_t
cat gpurun_out/marg_threads_sweep.txt
