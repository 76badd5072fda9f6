#!/bin/bash
# BA cluster (K workgroups per window): parity tests, then single-window latency and small-batch rates for K = 1, 2, 4, 8 and the automatic choice
timeout -k 10 600 python -m pytest tests/test_ba_solve_gpu.py tests/test_ba_factors_gpu.py tests/test_estimator_loop_gpu.py tests/test_marg_gpu.py -m gpu -x -q > gpurun_out/ba_cluster_tests.log 2>&1; echo "ba tests rc=$?"; tail -4 gpurun_out/ba_cluster_tests.log
for k in 1 2 4 8 0; do
  for w in 1 8; do
    LMONO_BA_CLUSTER=$k timeout -k 10 200 python bench.py --workload ba --windows $w --steps 5 --warmup 1 > gpurun_out/ba_k${k}_w$w.json 2>/dev/null
    python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/ba_k${k}_w$w.json").read().strip().splitlines()[-1])
    print("K=$k windows=$w |", d["value"], d["unit"], "ms/step", d["ms_per_step"], "iters", d["config"].get("mean_iterations"), "cost diff vs cpu", d.get("final_cost_rel_diff_vs_cpu"), flush=True)
except Exception as e:
    print("K=$k windows=$w | failed", e, flush=True)
PY
  done
done
