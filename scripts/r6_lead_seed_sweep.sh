#!/bin/bash
# Chained schedule on the three worlds: lead-in length x lead-in seeding (LMONO_OPT_LEAD_SEED).  Per run: scans/s, odometry ms, boundaries flagged, pairs re-run,
# repair ms, residual quantiles at the first check, ATE against the sequential oracle.  usage (GPU box): bash scripts/r6_lead_seed_sweep.sh [tag] ["seq list"] ["lead,seed list"]
O=gpurun_out/${1:-lead_seed}; mkdir -p $O
SEQS=${2:-"2 0 1"}
CFGS=${3:-"6,0 6,1 4,1 8,0 8,1 10,0 12,0"}
for S in $SEQS; do
  for cfg in $CFGS; do
    L=${cfg%,*}; X=${cfg#*,}
    timeout -k 10 300 python3 bench.py --seq $S --no-extras --cpu-sample 0 --steps 5 --lead $L --lead-seed $X > $O/seq${S}_l${L}_s${X}.json 2> $O/seq${S}_l${L}_s${X}.err || { echo "seq $S lead $L seed $X failed"; tail -3 $O/seq${S}_l${L}_s${X}.err; continue; }
    python3 - <<PY
import json
d=json.loads(open("$O/seq${S}_l${L}_s${X}.json").read().strip().splitlines()[-1])
g=d["roofline"]["group_ms_per_step"]; v=d["boundary_validation"]
print("seq $S lead $L seed $X |", d["value"], "scans/s", d["ms_per_step"], "ms | ate", d.get("ate_vs_cpu_m"), "odo", g["odometry_total"], "repair ms", v["repair_ms_per_step"], "flagged", v["flagged"], "pairs", v["pairs_rerun"], "rounds", v.get("rounds"), "q50/90/99", v["residual_q50_q90_q99"], flush=True)
PY
  done
done
