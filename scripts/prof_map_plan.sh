#!/bin/bash
# phase cycles of k_map_plan_update (a -DLMONO_MU_PROF scratch library on the GPU box): cycles between the kernel's workgroup barriers, frame by frame
O=$PWD/gpurun_out/${1:-mu_prof}; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -DLMONO_MU_PROF -o $O/prof.so lmono_amd/csrc/lmono_hip.hip 2>/dev/null || exit 1
LMONO_HIP_LIB=$O/prof.so timeout -k 10 200 python3 bench.py --workload map --scans 24 --streams 1 --steps 1 --warmup 0 --cpu-sample 0 2>/dev/null | grep MUPROF > $O/phases.txt
rm -f $O/prof.so
tail -6 $O/phases.txt
