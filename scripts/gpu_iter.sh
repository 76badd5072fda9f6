#!/bin/bash
# One GPU iteration (run through gpurun): LiDAR parity tests, then the headline bench.  usage: bash scripts/gpu_iter.sh <tag> [bench args]
TAG=${1:-iter}; shift
mkdir -p gpurun_out/$TAG
timeout -k 10 300 python -m pytest tests/test_lidar_gpu.py tests/test_sequence_gpu.py -m gpu -x -q > gpurun_out/$TAG/tests.log 2>&1
rc=$?
tail -3 gpurun_out/$TAG/tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 250 python bench.py "$@" > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err || { tail -5 gpurun_out/$TAG/bench.err; exit 1; }
python - <<PY
import json
d=json.loads(open("gpurun_out/$TAG/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("ate_vs_cpu_m"), d["roofline"]["ms_per_launch"])
print(d["roofline"]["group_ms_per_step"])
PY
