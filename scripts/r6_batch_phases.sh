#!/bin/bash
# phase clocks of the lock-step frame (LMONO_HOST_TIMING=1) at N streams.  usage (GPU box): bash scripts/r6_batch_phases.sh "256 64" [frames] ["groups list"]
O=gpurun_out/batch_phases; mkdir -p $O
python3 - <<PY
import sys
sys.path.insert(0, '.')
from workloads import s2 as K
for k in range(4):
    K.write_stream('$O/s%d.bin' % k, K.make_stream(${2:-300}, seed=2 + k, stops=()))
PY
for N in ${1:-256}; do
  for G in ${3:-1}; do
  for mode in sync async; do
    LMONO_HOST_TIMING=1 lmono_amd/host/estimator_seq $O/s0.bin - $mode streams=$N groups=$G digest $O/s1.bin $O/s2.bin $O/s3.bin 2>&1 | grep -E "BATCHTIM|^TIM" | sed "s/^/N=$N G=$G $mode /"
  done
  done
done
rm -f $O/s*.bin
