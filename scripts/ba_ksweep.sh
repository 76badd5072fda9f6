#!/bin/bash
# single-window latency and Estimator loop rate for K = 1, 2, 4, 8 workgroups per window (LMONO_BA_CLUSTER).  usage: bash scripts/ba_ksweep.sh <tag>
O=gpurun_out/$1; mkdir -p $O
for k in 1 2 4 8; do
  LMONO_BA_CLUSTER=$k timeout -k 10 100 python bench.py --workload ba --windows 1 > $O/ba1_k$k.json 2>$O/ba1_k$k.err
  LMONO_BA_CLUSTER=$k timeout -k 10 150 python bench.py --workload ba-seq --frames-seq 1000 --cpu-frames 0 > $O/seq_k$k.json 2>$O/seq_k$k.err
  python - <<PY
import json
a=json.loads(open("$O/ba1_k$k.json").read().strip().splitlines()[-1]); b=json.loads(open("$O/seq_k$k.json").read().strip().splitlines()[-1])
print("K=$k  single window %.3f ms  | Estimator loop %.1f frames/s (inline marginalisation %.1f)" % (a["ms_per_step"], b["value"], b["config"]["inline_marginalisation"]["frames_per_s"]))
PY
done | tee $O/ksweep.txt
