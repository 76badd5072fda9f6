#!/bin/bash
# One GPU iteration on the marginalisation kernels (run through gpurun): parity tests, the Estimator loop (overlapped and inline), phase cycles.
TAG=${1:-mg}; O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_marg_gpu.py tests/test_estimator_loop_gpu.py tests/test_config0_gpu.py tests/test_ba_solve_gpu.py -m gpu -x -q > $O/tests.log 2>&1
rc=$?; tail -4 $O/tests.log
[ $rc -ne 0 ] && exit $rc
LMONO_HOST_TIMING=1 timeout -k 10 300 python bench.py --workload ba-seq --cpu-frames 0 > $O/baseq.json 2> $O/baseq.err && python - <<PY
import json
d=json.loads(open("$O/baseq.json").read().strip().splitlines()[-1]); print("ba-seq: frames/s", d["value"], "inline", d["config"].get("inline_marginalisation"))
PY
grep -i "phase\|ms per frame\|HOSTCLK\|clock" $O/baseq.err | tail -12
bash scripts/prof_marg.sh $TAG
