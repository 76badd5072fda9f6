#!/bin/bash
# Stage stamps of k_corr_flat: -DLMONO_TILE_PROF build into a scratch library (the product library is not touched), then scripts/prof_flat.py.
#   usage: bash scripts/prof_flat.sh <tag> "<extra hipcc flags>" [scans chains lead]
TAG=${1:-flat}; FLAGS=$2; shift; shift
mkdir -p gpurun_out/prof_$TAG
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -DLMONO_TILE_PROF $FLAGS -o gpurun_out/prof_$TAG/prof.so lmono_amd/csrc/lmono_hip.hip 2>gpurun_out/prof_$TAG/build.err || { tail -5 gpurun_out/prof_$TAG/build.err; exit 1; }
LMONO_HIP_LIB=$PWD/gpurun_out/prof_$TAG/prof.so LMONO_LEAD_FULL=2 timeout -k 10 200 python scripts/prof_flat.py "$@" 2>&1 | tee gpurun_out/prof_$TAG/stages.txt
rm -f gpurun_out/prof_$TAG/prof.so
