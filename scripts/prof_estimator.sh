#!/bin/bash
# Kernel-time breakdown of the Estimator frame loop (estimator_seq over a 300-frame S2 stream) under rocprofv3.  usage: bash scripts/prof_estimator.sh
OUT=$PWD/gpurun_out/prof_est
mkdir -p $OUT
python3 - <<PY
import sys; sys.path.insert(0, '.')
from workloads import s2
st = s2.make_stream(300, seed=2)
s2.write_stream("$OUT/stream.bin", st)
PY
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- lmono_amd/host/estimator_seq $OUT/stream.bin > $OUT/out.txt 2> $OUT/err.txt
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
rm -rf $OUT/trace $OUT/stream.bin
grep TIM $OUT/out.txt
head -12 $OUT/kernel_stats.csv | cut -c1-150
