#!/bin/bash
# EstimatorBatch: lock-step frame time for streams x groups on ONE box.  usage (GPU box): bash scripts/r6_pool_groups.sh [frames]
O=gpurun_out/pool_groups; mkdir -p $O
python3 - <<PY
import sys
sys.path.insert(0, '.')
from workloads import s2 as K
for k in range(4):
    K.write_stream('$O/s%d.bin' % k, K.make_stream(${1:-300}, seed=2 + k, stops=()))
PY
B=lmono_amd/host/estimator_seq
for N in 128 256 512; do
  for G in 1 2; do
    for T in 16 8; do
      echo "N=$N G=$G threads=$T: $(LMONO_HOST_THREADS=$T $B $O/s0.bin - async streams=$N groups=$G digest $O/s1.bin $O/s2.bin $O/s3.bin 2>/dev/null | grep '^TIM')"
    done
  done
done
rm -f $O/s*.bin
