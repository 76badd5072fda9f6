// lmono_amd/host/estimator_seq.cpp -- replays a frame stream through the host mirror's Estimator::processImage (the reference's
// processEstimation without ROS and the image tracker, Estimator.cc:528-553) and prints, per frame, what the Python tests compare
// with the CPU oracle, then the trajectory of record in the reference's new_odometry.txt format (Estimator.cc:642-644).
//
// Stream file (doubles): n_frames, TLC[16], then per frame: header, L0_Pos[16], n_loop (0/1) [loop_time_stamp, old_T[3],
// old_Q[4] w x y z, correct_T[3], correct_Q[4] w x y z], n_features, n_features x (id, x_n, y_n, u, v).
// Usage: estimator_seq <stream.bin> [new_odometry.txt | -] [async]
// "async": marginalisation overlapped with the next frame (Estimator::setAsyncMargin); the PRI line (digest of the last prior) and
// everything else must come out the same bytes as without it.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "lmono_host.hpp"

using namespace lmono_host;

static std::vector<double> read_all(const char *path)
{
    FILE *f = std::fopen(path, "rb");
    if (!f) { std::perror(path); std::exit(2); }
    std::fseek(f, 0, SEEK_END); const long n = std::ftell(f); std::fseek(f, 0, SEEK_SET);
    std::vector<double> v((size_t)n / 8);
    if (std::fread(v.data(), 8, v.size(), f) != v.size()) std::exit(2);
    std::fclose(f);
    return v;
}

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    try {
        const std::vector<double> d = read_all(argv[1]);
        size_t k = 0;
        const int n_frames = (int)d[k++];
        HipContext hip(0);
        Params p;
        Estimator est(hip, p);
        if (argc > 3 && std::string(argv[3]) == "async") est.setAsyncMargin(true);
        for (int j = 0; j < 16; j++) est.TLC[j] = d[k++];
        double solve_ms = 0; int solves = 0;
        for (int f = 0; f < n_frames; f++) {
            const double header = d[k++];
            const double *L0 = &d[k]; k += 16;
            if ((int)d[k++]) {
                Estimator::LoopFrame lf;
                lf.loop_time_stamp = d[k++];
                for (int j = 0; j < 3; j++) lf.old_T[j] = d[k++];
                for (int j = 0; j < 4; j++) lf.old_Q[j] = d[k++];
                for (int j = 0; j < 3; j++) lf.correct_T[j] = d[k++];
                for (int j = 0; j < 4; j++) lf.correct_Q[j] = d[k++];
                est.setLoopFrame(lf);
            }
            const int nf = (int)d[k++];
            FeatureManager::Image image;
            for (int j = 0; j < nf; j++) { const int id = (int)d[k]; image[id] = { d[k + 1], d[k + 2], d[k + 3], d[k + 4] }; k += 5; }
            const bool was_inited = est.stage_flag == Estimator::INITED;
            const auto t0 = std::chrono::steady_clock::now();
            const bool keyframe = est.processImage(header, image, L0);
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (was_inited) { solve_ms += ms; solves++; }
            std::printf("FRM %d %d %d %d %d %d %.17g %d %d %zu\n", f, keyframe ? 1 : 0, (int)est.stage_flag, est.static_status ? 1 : 0, est.iterations, est.termination,
                        est.final_cost, est.margin_calls[0], est.margin_calls[1], est.feature_manager.feature.size());
        }
        for (const auto &r : est.new_odometry)
            std::printf("ODO %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]);
        est.marginWait();
        {
            const auto &mi = est.last_marginalization_info;
            double sj = 0, sr = 0;
            for (double v : mi.linearized_jacobians) sj += v * v;
            for (double v : mi.linearized_residuals) sr += v * v;
            std::printf("PRI %d %d %d %zu %.17g %.17g\n", mi.m, mi.n, mi.status, mi.parameter_blocks.size(), sj, sr);
        }
        std::printf("EXT");
        for (int j = 0; j < 16; j++) std::printf(" %.17g", est.TLC[j]);
        std::printf("\nTIM %d %.6f\n", solves, solves ? solve_ms / solves : 0.0);
        std::printf("FLP %.17g %ld\n", est.solve_flops, est.solve_obs);     // algorithmic flops of all window solves, projection blocks in all
        lmono_host::estimator_print_phase_clock();
        if (argc > 2 && std::string(argv[2]) != "-") {
            FILE *fo = std::fopen(argv[2], "w");
            if (!fo) { std::perror(argv[2]); return 2; }
            for (const auto &r : est.new_odometry) std::fprintf(fo, "%f %f %f %f %f %f %f %f\n", r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]);   // Estimator.cc:642
            std::fclose(fo);
        }
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "estimator_seq: %s\n", e.what());
        return 1;
    }
}
