#!/bin/bash
# build everything here, then one BA iteration on the GPU box (nothing runs on the box when the build fails).  usage: bash scripts/gpu_ba.sh <tag> [quick]
cd "$(dirname "$0")/.." || exit 1
python __graft_entry__.py > /tmp/build.log 2>&1 || { grep -E "error" -A5 /tmp/build.log | head -30; echo BUILD FAILED; exit 1; }
gpurun --timeout 900 -- "bash scripts/ba_iter.sh $1 $2" 2>&1 | tail -9
