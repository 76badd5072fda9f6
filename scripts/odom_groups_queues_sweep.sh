#!/bin/bash
# Chain groups (LMONO_ODOM_STREAMS) x hardware queues (GPU_MAX_HW_QUEUES) of the headline pass.  usage (GPU box): bash scripts/odom_groups_queues_sweep.sh
set -u
OUT=$PWD/gpurun_out/odom_groups_queues
mkdir -p $OUT
: > $OUT/summary.txt
for Q in 4 8; do for G in 3 4 6 8; do
  GPU_MAX_HW_QUEUES=$Q LMONO_ODOM_STREAMS=$G timeout -k 10 280 python3 bench.py --steps 5 --warmup 1 --cpu-sample 0 > $OUT/b_q${Q}_g$G.json 2> $OUT/err_q${Q}_g$G.txt || { echo "queues $Q groups $G failed" >> $OUT/summary.txt; continue; }
  python3 - $Q $G $OUT/b_q${Q}_g$G.json >> $OUT/summary.txt <<'PY'
import json, sys
d = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
r = d.get("roofline", {})
print(f"hw queues {sys.argv[1]} groups {sys.argv[2]}: {d['value']:.0f} scans/s, {d['ms_per_step']:.2f} ms per pass, search {r.get('ms_per_launch')} ms per launch, stage ms {d.get('stage_ms_per_step', d.get('stage_ms'))}")
PY
done; done
cat $OUT/summary.txt
