#!/bin/bash
# Phase cycles of k_ba_solve (window 0) in a -DLMONO_BA_PROF build made on the box into a scratch library.  usage: bash scripts/prof_ba_phases.sh [n_windows]
set -e
mkdir -p gpurun_out/ba_prof
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -DLMONO_BA_PROF -o gpurun_out/ba_prof/prof.so lmono_amd/csrc/lmono_hip.hip 2>/dev/null
export LMONO_HIP_LIB=$PWD/gpurun_out/ba_prof/prof.so      # a scratch library: the product library is never overwritten
timeout -k 10 120 python scripts/prof_ba.py ${1:-1} | tee gpurun_out/ba_prof/phases.txt
rm -f gpurun_out/ba_prof/prof.so
