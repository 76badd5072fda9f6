#!/bin/bash
# lock-step frame loop, A / B of environment switches on ONE box (host speed varies 2x box to box).  usage (GPU box): bash scripts/r6_batch_ab.sh "VAR=1" [streams] [frames]
O=gpurun_out/batch_ab; mkdir -p $O
python3 - <<PY
import sys
sys.path.insert(0, '.')
from workloads import s2 as K
for k in range(4):
    K.write_stream('$O/s%d.bin' % k, K.make_stream(${3:-300}, seed=2 + k, stops=()))
PY
N=${2:-256}
for rep in 1 2 3; do
  echo "default:  $(lmono_amd/host/estimator_seq $O/s0.bin - async streams=$N digest $O/s1.bin $O/s2.bin $O/s3.bin 2>/dev/null | grep '^TIM')"
  echo "$1: $(env $1 lmono_amd/host/estimator_seq $O/s0.bin - async streams=$N digest $O/s1.bin $O/s2.bin $O/s3.bin 2>/dev/null | grep '^TIM')"
done
rm -f $O/s*.bin
