#!/bin/bash
# experiment: k_map_factor with fast Jacobi rotations (scratch library on the GPU box): parity tests, bench line, kernel time
O=$PWD/gpurun_out/${1:-fastrot}; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -DLMONO_MAP_FAST_ROT=1 -o $O/fr.so lmono_amd/csrc/lmono_hip.hip 2>/dev/null || exit 1
LMONO_HIP_LIB=$O/fr.so timeout -k 10 300 python -m pytest tests/test_mapping_gpu.py -m gpu -q 2>&1 | tail -3
for lib in "" $O/fr.so; do
  LMONO_HIP_LIB=$lib python bench.py --workload map --scans 64 --streams 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib [$lib]', d['value'], 'max_pose_diff_vs_cpu', d['max_pose_diff_vs_cpu'])"
done
rm -f $O/fr.so
