#!/bin/bash
# EstimatorBatch, A / B of two prebuilt binaries on ONE box.  usage (GPU box): bash scripts/r6_pool_ab.sh binA binB [frames] [groups]   (the binaries: builds of two trees into gpurun_bin/, which is git-ignored and travels with gpurun)
O=gpurun_out/pool_ab; mkdir -p $O
python3 - <<PY
import sys
sys.path.insert(0, '.')
from workloads import s2 as K
for k in range(4):
    K.write_stream('$O/s%d.bin' % k, K.make_stream(${3:-300}, seed=2 + k, stops=()))
PY
for N in 64 256; do
  for rep in 1 2 3; do
    for b in $1 $2; do
      echo "$b N=$N G=${4:-1}: $(LMONO_HOST_TIMING=1 $b $O/s0.bin - async streams=$N groups=${4:-1} digest $O/s1.bin $O/s2.bin $O/s3.bin 2>$O/err.txt | grep '^TIM') | $(grep BATCHTIM $O/err.txt | head -1 | sed 's/.*frames: //; s/ (ms per.*//')"
    done
  done
done
rm -f $O/s*.bin
