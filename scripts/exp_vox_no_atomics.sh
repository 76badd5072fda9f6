set -u
mkdir -p gpurun_out/voxexp
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -DLMONO_VOX_TIMING_NO_ATOMICS -o gpurun_out/voxexp/noatom.so lmono_amd/csrc/lmono_hip.hip 2>/dev/null || exit 1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export LMONO_HIP_LIB=$PWD/gpurun_out/voxexp/noatom.so
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/voxexp/trace -- python3 bench.py --workload map --scans 64 --streams 1 > gpurun_out/voxexp/bench.json 2> gpurun_out/voxexp/err.txt
find gpurun_out/voxexp/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/voxexp/kernel_stats_noatomics.csv
rm -rf gpurun_out/voxexp/trace gpurun_out/voxexp/noatom.so
grep "k_vox" gpurun_out/voxexp/kernel_stats_noatomics.csv | cut -d, -f1-4
