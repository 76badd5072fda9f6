// lmono_amd/host/host_test.cpp -- drives the host mirror (Estimator::optimization / outliersRejection / slideWindow /
// FeatureManager::triangulate) on a window fixture written by tests/test_host_cpp.py and prints the results as text
// for the Python test to compare with the CPU oracle.  Usage: host_test <fixture.bin>
#include <cstdio>
#include <cstdlib>
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
        HipContext hip(0);
        Params p;
        Estimator est(hip, p);
        const int n_tracks = (int)d[k++];
        est.frame_count = WINDOW_SIZE; est.first_refine = p.FINE_TIMES;     // prior active, as after the first solve
        est.stage_flag = Estimator::INITED;
        for (int i = 0; i <= WINDOW_SIZE; i++) { for (int j = 0; j < 9; j++) est.Rs[i].m[j] = d[k++]; for (int j = 0; j < 3; j++) est.Ps[i].v[j] = d[k++]; }
        for (int i = 0; i <= WINDOW_SIZE; i++) { for (int j = 0; j < 9; j++) est.L0_R[i].m[j] = d[k++]; for (int j = 0; j < 3; j++) est.L0_T[i].v[j] = d[k++]; }
        for (int j = 0; j < 16; j++) est.TLC[j] = d[k++];
        for (int t = 0; t < n_tracks; t++) {
            FeaturePerId f;
            f.feature_id = t; f.start_frame = (int)d[k++];
            const int n = (int)d[k++];
            f.estimated_depth = d[k++];
            for (int o = 0; o < n; o++) { FeaturePerFrame ff; ff.pt[0] = d[k++]; ff.pt[1] = d[k++]; f.feature_per_frame.push_back(ff); }
            est.feature_manager.feature.push_back(f);
        }
        est.feature_manager.triangulate(WINDOW_SIZE, est.Rs, est.Ps, est.TLC);
        std::printf("TRI");
        for (auto &f : est.feature_manager.feature) std::printf(" %.17g", f.estimated_depth);
        std::printf("\n");
        const bool conv = est.optimization();
        std::printf("OPT %d %.17g %.17g %d %d\n", conv ? 1 : 0, est.initial_cost, est.final_cost, est.iterations, est.termination);
        std::printf("POS");
        for (int i = 0; i <= WINDOW_SIZE; i++) for (int j = 0; j < 3; j++) std::printf(" %.17g", est.Ps[i].v[j]);
        std::printf("\nROT");
        for (int i = 0; i <= WINDOW_SIZE; i++) for (int j = 0; j < 9; j++) std::printf(" %.17g", est.Rs[i].m[j]);
        std::printf("\n");
        est.marginalization_flag = Estimator::MARGIN_OLD;
        est.margin();
        {
            // gauge-invariant digest of the prior: trace(J^T J) and |J^T r|^2
            const auto &J = est.last_marginalization_info.linearized_jacobians; const auto &r = est.last_marginalization_info.linearized_residuals;
            double tr = 0, g2 = 0;
            for (int c = 0; c < 66; c++) { double g = 0; for (int k = 0; k < 66; k++) { tr += J[k * 66 + c] * J[k * 66 + c]; g += J[k * 66 + c] * r[k]; } g2 += g * g; }
            std::printf("MRG %d %.17g %.17g %d\n", est.last_marginalization_info.m, tr, g2, est.last_marginalization_info.status);
        }
        std::set<int> rm;
        est.outliersRejection(rm, p.OUTLIER_T);
        std::printf("OUT %zu\n", rm.size());
        est.feature_manager.removeOutlier(rm);
        const size_t before = est.feature_manager.feature.size();
        est.marginalization_flag = Estimator::MARGIN_OLD;
        est.slideWindow();
        std::printf("SLD %zu %zu", before, est.feature_manager.feature.size());
        int anchored0 = 0;
        for (auto &f : est.feature_manager.feature) if (f.start_frame == 0) anchored0++;
        std::printf(" %d\n", anchored0);
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "host_test: %s\n", e.what());
        return 1;
    }
}
