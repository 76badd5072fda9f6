// lmono_amd/host/run_sequence.cpp -- KITTI-layout sequence -> scanRegistration + laserOdometry on the GPU -> trajectory
// file in the reference's "loam_odometry" format (Estimator.cc:270).  The C++ counterpart of examples/run_sequence.py:
//   run_sequence <sequence_dir> <out_trajectory> [first] [count] [n_chains] [lead] [out_mapped_trajectory]
// With the last argument laserMapping refines every pose (scan-to-map) and its trajectory is written too.
// n_chains = 0: ONLINE -- one scan per callback through LaserOdometryNode (lmono_odom_step), laserMapping behind it per scan, the
// per-scan latency printed ("LAT" lines: wall ms of the callback); same trajectories as n_chains = 1.
#include "kitti_io.hpp"
#include "lmono_host.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>

using namespace lmono_host;

int main(int argc, char **argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: run_sequence <sequence_dir> <out_trajectory> [first] [count] [n_chains] [lead]\n"); return 2; }
    const std::string seq = argv[1], out = argv[2];
    const int first = argc > 3 ? std::atoi(argv[3]) : 0;
    int count = argc > 4 ? std::atoi(argv[4]) : -1;
    const int n_chains = argc > 5 ? std::atoi(argv[5]) : 1, lead = argc > 6 ? std::atoi(argv[6]) : 0;
    const std::string out_mapped = argc > 7 ? argv[7] : "";
    try {
        std::vector<double> stamps;
        if (!read_times(seq + "/times.txt", stamps)) { std::fprintf(stderr, "%s/times.txt unreadable\n", seq.c_str()); return 1; }
        if (count < 0 || first + count > (int)stamps.size()) count = (int)stamps.size() - first;
        if (count <= 0) { std::fprintf(stderr, "no scans\n"); return 1; }
        std::vector<float> xyzi;
        std::vector<int64_t> off(1, 0);
        for (int k = 0; k < count; k++) {
            const long n = read_velodyne_bin(velodyne_path(seq, first + k), xyzi);
            if (n < 0) { std::fprintf(stderr, "%s unreadable\n", velodyne_path(seq, first + k).c_str()); return 1; }
            off.push_back(off.back() + n);
        }
        HipContext hip(0);
        if (n_chains == 0) {
            int64_t cap = 0;
            for (int k = 0; k < count; k++) cap = std::max(cap, off[(size_t)k + 1] - off[(size_t)k]);
            LaserOdometryNode node(hip, (int)cap);
            std::unique_ptr<LaserMapping> mapping;
            std::unique_ptr<TrajectoryWriter> wm;
            if (!out_mapped.empty()) { mapping.reset(new LaserMapping(hip)); wm.reset(new TrajectoryWriter(out_mapped, 1)); if (!wm->ok()) { std::fprintf(stderr, "cannot write %s\n", out_mapped.c_str()); return 1; } }
            TrajectoryWriter w(out, 1);
            if (!w.ok()) { std::fprintf(stderr, "cannot write %s\n", out.c_str()); return 1; }
            for (int k = 0; k < count; k++) {
                const auto t0 = std::chrono::steady_clock::now();
                node.laserCloudHandler(xyzi.data() + (size_t)off[(size_t)k] * 4, (int)(off[(size_t)k + 1] - off[(size_t)k]));
                const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                w.write(stamps[(size_t)(first + k)], node.t_w_curr, node.q_w_curr);
                std::printf("LAT %d %.3f features %d %d %d %d\n", k, ms, node.info[1], node.info[2], node.info[3], node.info[4]);
                if (mapping) {
                    double q[4], t[3];
                    mapping->process(node, q, t);
                    wm->write(stamps[(size_t)(first + k)], t, q);
                    std::printf("MAP %d edges %d %d planes %d %d iters %d %d\n", k, mapping->stats[0], mapping->stats[1], mapping->stats[2], mapping->stats[3],
                                mapping->stats[4], mapping->stats[5]);
                }
            }
            std::printf("DONE %d scans %lld points\n", count, (long long)off.back());
            return 0;
        }
        ScanRegistration reg(hip, count, off.back());
        reg.laserCloudHandlerHost(xyzi.data(), off.data(), count);
        LaserOdometry odo(hip);
        const std::vector<double> poses = odo.process(reg, n_chains, lead);
        TrajectoryWriter w(out, 1);
        if (!w.ok()) { std::fprintf(stderr, "cannot write %s\n", out.c_str()); return 1; }
        for (int k = 0; k < count; k++) w.write(stamps[(size_t)(first + k)], &poses[(size_t)k * 7 + 4], &poses[(size_t)k * 7]);
        if (!out_mapped.empty()) {
            LaserMapping mapping(hip);
            TrajectoryWriter wm(out_mapped, 1);
            if (!wm.ok()) { std::fprintf(stderr, "cannot write %s\n", out_mapped.c_str()); return 1; }
            for (int k = 0; k < count; k++) {
                double q[4], t[3];
                mapping.process(reg, k, &poses[(size_t)k * 7], &poses[(size_t)k * 7 + 4], q, t);
                wm.write(stamps[(size_t)(first + k)], t, q);
                std::printf("MAP %d edges %d %d planes %d %d iters %d %d\n", k, mapping.stats[0], mapping.stats[1], mapping.stats[2], mapping.stats[3],
                            mapping.stats[4], mapping.stats[5]);
            }
        }
        std::printf("DONE %d scans %lld points\n", count, (long long)off.back());
    } catch (const std::exception &e) {
        std::fprintf(stderr, "run_sequence: %s\n", e.what());
        return 1;
    }
    return 0;
}
