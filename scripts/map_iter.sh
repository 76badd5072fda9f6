#!/bin/bash
# One GPU iteration on laserMapping (run through gpurun): parity tests, the single-stream and 64-stream bench lines.
TAG=${1:-map}; O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_mapping_gpu.py -m gpu -x -q > $O/tests.log 2>&1
rc=$?; tail -15 $O/tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --workload map --scans 64 --streams 1 > $O/map1.json 2> $O/map1.err && python - <<PY
import json
d=json.loads(open("$O/map1.json").read().strip().splitlines()[-1]); print("map 1 stream: frames/s", d["value"], {k: d["config"].get(k) for k in ("max_pose_diff_vs_cpu","ate_m") if k in d["config"]})
PY
LMONO_MAP_HOST_TABLES=1 timeout -k 10 300 python bench.py --workload map --scans 64 --streams 1 > $O/map1h.json 2> $O/map1h.err && python - <<PY
import json
d=json.loads(open("$O/map1h.json").read().strip().splitlines()[-1]); print("map 1 stream, host tables: frames/s", d["value"])
PY
if [ "$2" = full ]; then
timeout -k 10 300 python bench.py --workload map --scans 64 --streams 64 > $O/map64.json 2> $O/map64.err && python - <<PY
import json
d=json.loads(open("$O/map64.json").read().strip().splitlines()[-1]); print("map 64 streams: frames/s", d["value"])
PY
fi
