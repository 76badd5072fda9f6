#!/bin/bash
# EstimatorBatch: lock-step frame time for streams x groups on ONE box (groups > 2: one workgroup per window forced, LMONO_BA_CLUSTER=1 -- several batches' clusters
# together would want more resident workgroups than the card has CUs).  usage (GPU box): bash scripts/r6_pool_groups.sh [frames]
O=gpurun_out/pool_groups; mkdir -p $O
python3 - <<PY
import sys
sys.path.insert(0, '.')
from workloads import s2 as K
for k in range(4):
    K.write_stream('$O/s%d.bin' % k, K.make_stream(${1:-300}, seed=2 + k, stops=()))
PY
B=lmono_amd/host/estimator_seq
for N in 256 512; do
  for G in 1 2 3 4; do
    echo "N=$N G=$G K=1: $(LMONO_BA_CLUSTER=1 $B $O/s0.bin - async streams=$N groups=$G digest $O/s1.bin $O/s2.bin $O/s3.bin 2>/dev/null | grep '^TIM')  digests: $(LMONO_BA_CLUSTER=1 $B $O/s0.bin - async streams=$N groups=$G digest $O/s1.bin $O/s2.bin $O/s3.bin 2>/dev/null | grep '^DIG' | awk '{print $3}' | sort -u | wc -l) distinct"
  done
done
rm -f $O/s*.bin
