timeout -k 10 600 python -m pytest tests/test_mapping_gpu.py tests/test_config0_gpu.py -m gpu -x -q > gpurun_out/r4_msk_tests.txt 2>&1 || { tail -30 gpurun_out/r4_msk_tests.txt; exit 1; }
tail -2 gpurun_out/r4_msk_tests.txt
: > gpurun_out/r4_msk_sweep.txt
for K in 1 2 4 8; do for S in 1 64; do
  LMONO_MAP_SOLVE_K=$K timeout -k 10 150 python3 bench.py --workload map --scans 64 --streams $S 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K=$K streams $S:', d['value'], 'frames/s, max pose diff', d['max_pose_diff_vs_cpu'], 'ate', d['ate_vs_truth_m']['mapped'])" >> gpurun_out/r4_msk_sweep.txt || exit 1
done; done
cat gpurun_out/r4_msk_sweep.txt
