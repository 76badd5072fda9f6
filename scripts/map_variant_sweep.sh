#!/bin/bash
# Builds library variants (one set of -D flags per line of the file given as $1) into scratch files on the GPU box and runs the laserMapping bench
# (64 scans; 1 stream and 64 streams) with each.  The product library is never touched.  usage: bash scripts/map_variant_sweep.sh <variants file> <tag>
set -u
VAR=$1; TAG=$2
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
: > $OUT/summary.txt
i=0
while IFS= read -r flags || [ -n "$flags" ]; do
  i=$((i+1))
  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 $flags -o $OUT/v$i.so lmono_amd/csrc/lmono_hip.hip 2> $OUT/build$i.err || { echo "[$flags] build failed" >> $OUT/summary.txt; continue; }
  for S in 1 64; do
    LMONO_HIP_LIB=$OUT/v$i.so timeout -k 10 150 python3 bench.py --workload map --scans 64 --streams $S > $OUT/v${i}_s$S.json 2> $OUT/v${i}_s$S.err || { echo "[$flags] streams $S failed" >> $OUT/summary.txt; continue; }
    python3 - "$flags" $S $OUT/v${i}_s$S.json >> $OUT/summary.txt <<'PY'
import json, sys
d = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
print(f"[{sys.argv[1]}] streams {sys.argv[2]}: {d['value']:.1f} frames/s, max pose diff vs cpu {d['max_pose_diff_vs_cpu']:.2e}, ate mapped {d['ate_vs_truth_m']['mapped']}")
PY
  done
  rm -f $OUT/v$i.so
done < $VAR
cat $OUT/summary.txt
