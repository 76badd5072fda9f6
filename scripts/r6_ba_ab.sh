#!/bin/bash
# A / B on ONE box: the in-tree library against a build with the given -D switches (scratch directory, LMONO_HIP_LIB / LD_LIBRARY_PATH): BA single window, 1024
# windows, Estimator loop.  usage (GPU box): bash scripts/r6_ba_ab.sh "-DLMONO_BA_HS_FUSED=0" [tag]
O=gpurun_out/${2:-ba_ab}; mkdir -p $O/lib
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 $1 -o $O/lib/liblmono_hip.so lmono_amd/csrc/lmono_hip.hip 2>/dev/null || { echo "variant build failed"; exit 1; }
run() {
  for w in 1 1024; do timeout -k 10 200 python3 bench.py --workload ba --windows $w --steps $([ $w = 1 ] && echo 20 || echo 5) --warmup 2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ba', d['config']['workload'][-24:], d['value'], 'windows/s', d['ms_per_step'], 'ms')"; done
  timeout -k 10 300 python3 bench.py --workload ba-seq --frames-seq 600 --cpu-frames 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ba-seq', d['value'], 'frames/s, inline', d['config']['inline_marginalisation']['frames_per_s'])"
}
for rep in 1 2; do
echo "in-tree library (rep $rep):"; run
echo "variant $1 (rep $rep):"; LMONO_HIP_LIB=$PWD/$O/lib/liblmono_hip.so LD_LIBRARY_PATH=$PWD/$O/lib:${LD_LIBRARY_PATH:-} run
done
rm -rf $O/lib
