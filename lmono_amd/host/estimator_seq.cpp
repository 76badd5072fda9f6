// lmono_amd/host/estimator_seq.cpp -- replays a frame stream through the host mirror's Estimator::processImage (the reference's
// processEstimation without ROS and the image tracker, Estimator.cc:528-553) and prints, per frame, what the Python tests compare
// with the CPU oracle, then the trajectory of record in the reference's new_odometry.txt format (Estimator.cc:642-644).
//
// Stream file (doubles): n_frames, TLC[16], then per frame: header, L0_Pos[16], n_loop (0/1) [loop_time_stamp, old_T[3],
// old_Q[4] w x y z, correct_T[3], correct_Q[4] w x y z], n_features, n_features x (id, x_n, y_n, u, v).
// Usage: estimator_seq <stream.bin> [new_odometry.txt | -] [sync | async] [streams=N [groups=G] [digest] [more stream files ...]]
// "async": marginalisation overlapped with the next frame (Estimator::setAsyncMargin); the PRI line (digest of the last prior) and
// everything else must come out the same bytes as without it.
// "streams=N": N independent Estimators stepped in lock-step by EstimatorBatch (one batched C-ABI call per numeric step); stream s replays
// file s mod (number of files given).  Every stream's lines are printed behind a "STR s" line and are, byte for byte, the lines of the
// single-stream run of its file (and "DIG s <hash>" = FNV-1a of those lines; "digest": print only the DIG lines -- 256 streams x 2761 frames
// of text is 70 MB).  The single-stream run prints its own "DIG 0 <hash>" over the same lines.
// "groups=G": the N streams as G EstimatorBatches of N / G streams, each on its own context (own HIP stream) and host thread: the sequences are
// independent, so the groups need not wait for each other -- one group's host passes run beside another group's kernels.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
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

struct Frame {
    double header; double L0[16];
    bool has_loop = false; Estimator::LoopFrame loop;
    FeatureManager::Image image;
};
struct Stream { double TLC[16]; std::vector<Frame> frames; };

static Stream parse_stream(const char *path)
{
    const std::vector<double> d = read_all(path);
    size_t k = 0;
    Stream st;
    const int n_frames = (int)d[k++];
    for (int j = 0; j < 16; j++) st.TLC[j] = d[k++];
    st.frames.resize((size_t)n_frames);
    for (int f = 0; f < n_frames; f++) {
        Frame &fr = st.frames[(size_t)f];
        fr.header = d[k++];
        for (int j = 0; j < 16; j++) fr.L0[j] = d[k++];
        if ((int)d[k++]) {
            fr.has_loop = true;
            Estimator::LoopFrame &lf = fr.loop;
            lf.loop_time_stamp = d[k++];
            for (int j = 0; j < 3; j++) lf.old_T[j] = d[k++];
            for (int j = 0; j < 4; j++) lf.old_Q[j] = d[k++];
            for (int j = 0; j < 3; j++) lf.correct_T[j] = d[k++];
            for (int j = 0; j < 4; j++) lf.correct_Q[j] = d[k++];
        }
        const int nf = (int)d[k++];
        for (int j = 0; j < nf; j++) { const int id = (int)d[k]; fr.image[id] = { d[k + 1], d[k + 2], d[k + 3], d[k + 4] }; k += 5; }
    }
    return st;
}

// the lines of one stream (FRM per frame, then ODO / PRI / EXT) and their running digest
struct Lines {
    std::string text;
    unsigned long long h = 1469598103934665603ull;
    bool keep = true;
    void add(const char *s)
    {
        for (const char *p = s; *p; p++) { h ^= (unsigned char)*p; h *= 1099511628211ull; }
        if (keep) text += s;
    }
};
static void frm_line(Lines &out, int f, bool keyframe, const Estimator &est)
{
    char buf[256];
    std::snprintf(buf, sizeof buf, "FRM %d %d %d %d %d %d %.17g %d %d %zu\n", f, keyframe ? 1 : 0, (int)est.stage_flag, est.static_status ? 1 : 0, est.iterations, est.termination,
                  est.final_cost, est.margin_calls[0], est.margin_calls[1], est.feature_manager.feature.size());
    out.add(buf);
}
static void tail_lines(Lines &out, const Estimator &est)
{
    char buf[1024];
    for (const auto &r : est.new_odometry) {
        std::snprintf(buf, sizeof buf, "ODO %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]);
        out.add(buf);
    }
    const auto &mi = est.last_marginalization_info;
    double sj = 0, sr = 0;
    for (double v : mi.linearized_jacobians) sj += v * v;
    for (double v : mi.linearized_residuals) sr += v * v;
    std::snprintf(buf, sizeof buf, "PRI %d %d %d %zu %.17g %.17g\n", mi.m, mi.n, mi.status, mi.parameter_blocks.size(), sj, sr);
    out.add(buf);
    std::string e = "EXT";
    for (int j = 0; j < 16; j++) { std::snprintf(buf, sizeof buf, " %.17g", est.TLC[j]); e += buf; }
    e += "\n";
    out.add(e.c_str());
}

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    try {
        int n_streams = 0, n_groups = 1; bool digest_only = false, async = false;
        std::vector<const char *> files{ argv[1] };
        for (int a = 3; a < argc; a++) {
            const std::string s = argv[a];
            if (s == "async") async = true;
            else if (s == "sync") async = false;
            else if (s.rfind("streams=", 0) == 0) n_streams = std::atoi(s.c_str() + 8);
            else if (s.rfind("groups=", 0) == 0) n_groups = std::max(1, std::atoi(s.c_str() + 7));
            else if (s == "digest") digest_only = true;
            else files.push_back(argv[a]);
        }
        HipContext hip(0);
        Params p;
        if (n_streams <= 0) {
            const Stream st = parse_stream(argv[1]);
            Estimator est(hip, p);
            if (async) est.setAsyncMargin(true);
            std::memcpy(est.TLC, st.TLC, sizeof(st.TLC));
            Lines out;
            double solve_ms = 0; int solves = 0;
            for (size_t f = 0; f < st.frames.size(); f++) {
                const Frame &fr = st.frames[f];
                if (fr.has_loop) est.setLoopFrame(fr.loop);
                const bool was_inited = est.stage_flag == Estimator::INITED;
                const auto t0 = std::chrono::steady_clock::now();
                const bool keyframe = est.processImage(fr.header, fr.image, fr.L0);
                const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                if (was_inited) { solve_ms += ms; solves++; }
                frm_line(out, (int)f, keyframe, est);
            }
            est.marginWait();
            tail_lines(out, est);
            std::fputs(out.text.c_str(), stdout);
            std::printf("DIG 0 %016llx\n", out.h);
            std::printf("TIM %d %.6f\n", solves, solves ? solve_ms / solves : 0.0);
            std::printf("FLP %.17g %ld\n", est.solve_flops, est.solve_obs);     // algorithmic flops of all window solves, projection blocks in all
            lmono_host::estimator_print_phase_clock();
            if (argc > 2 && std::string(argv[2]) != "-") {
                FILE *fo = std::fopen(argv[2], "w");
                if (!fo) { std::perror(argv[2]); return 2; }
                for (const auto &r : est.new_odometry) std::fprintf(fo, "%f %f %f %f %f %f %f %f\n", r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]);   // Estimator.cc:642
                std::fclose(fo);
            }
            return 0;
        }
        // ---- N streams in lock-step
        std::vector<Stream> src;
        for (const char *f : files) src.push_back(parse_stream(f));
        const size_t n_frames = src[0].frames.size();
        for (const Stream &s : src) if (s.frames.size() != n_frames) { std::fprintf(stderr, "estimator_seq: the stream files must hold the same number of frames\n"); return 2; }
        const int N = n_streams, G = std::min(n_groups, N);
        std::vector<Lines> out((size_t)N);
        for (int s = 0; s < N; s++) out[(size_t)s].keep = !digest_only;
        // group g: streams [s0, s1) as one EstimatorBatch on its own context / HIP stream (G = 1: the context above).  ONE thread drives all groups, frame by
        // frame: finish(g, frame f - 1), begin(g, frame f) for g = 0 .. G - 1 -- a group's solve is in flight while the thread runs the other groups' host
        // passes (threads per group were measured first: the HIP runtime serialises the submitting threads, 9.1 -> 10.4 ms per lock-step frame at G = 2)
        struct Group {
            int s0 = 0, n = 0;
            std::unique_ptr<HipContext> own;
            std::unique_ptr<EstimatorBatch> eb;
            std::vector<double> headers;
            std::vector<const FeatureManager::Image *> img;
            std::vector<std::array<double, 16>> L0;
            std::unique_ptr<bool[]> kf;
            size_t cur_f = 0;           // the frame the batch is working on (begin sets it; the frame hook prints it)
        };
        std::vector<Group> grp((size_t)G);
        for (int g = 0; g < G; g++) {
            Group &q = grp[(size_t)g];
            q.s0 = (int)((long long)g * N / G); q.n = (int)((long long)(g + 1) * N / G) - q.s0;
            if (G > 1) { q.own.reset(new HipContext(0)); q.own->useOwnStream(); }
            q.eb.reset(new EstimatorBatch(G > 1 ? *q.own : hip, p, q.n));
            if (async) q.eb->setAsyncMargin(true);
            for (int s = 0; s < q.n; s++) std::memcpy(q.eb->stream(s).TLC, src[(size_t)(q.s0 + s) % src.size()].TLC, 128);
            q.headers.resize((size_t)q.n); q.img.resize((size_t)q.n); q.L0.resize((size_t)q.n); q.kf.reset(new bool[(size_t)q.n]);
            // a stream's FRM line is written at the end of its frame by the thread that ran the stream's last pass (EstimatorBatch::setFrameHook): no serial
            // loop over the streams on the driving thread
            Group *qp = &q;
            q.eb->setFrameHook([qp, &out](int s, const Estimator &e) { frm_line(out[(size_t)(qp->s0 + s)], (int)qp->cur_f, qp->kf[(size_t)s], e); });
        }
        auto begin = [&](Group &q, size_t f) {
            q.cur_f = f;
            for (int s = 0; s < q.n; s++) {
                const Frame &fr = src[(size_t)(q.s0 + s) % src.size()].frames[f];
                q.headers[(size_t)s] = fr.header; q.img[(size_t)s] = &fr.image; std::memcpy(q.L0[(size_t)s].data(), fr.L0, 128);
                if (fr.has_loop) q.eb->stream(s).setLoopFrame(fr.loop);
            }
            q.eb->processImageBegin(q.headers.data(), q.img.data(), reinterpret_cast<const double (*)[16]>(q.L0.data()), q.kf.get());
        };
        auto finish = [&](Group &q, size_t) { q.eb->processImageFinish(); };
        double solve_ms = 0; int solves = 0;
        for (size_t f = 0; f < n_frames; f++) {
            const bool was_inited = grp[0].eb->stream(0).stage_flag == Estimator::INITED;
            const auto t0 = std::chrono::steady_clock::now();
            for (int g = 0; g < G; g++) {
                if (G > 1 && f > 0) finish(grp[(size_t)g], f - 1);
                begin(grp[(size_t)g], f);
                if (G == 1) finish(grp[(size_t)g], f);
            }
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            // (G > 1: an iteration finishes frame f - 1 and begins frame f of every group: one lock-step frame of work, counted when frame f - 1 was an INITED one)
            if (G == 1 ? was_inited : (f > 0 && was_inited)) { solve_ms += ms; solves++; }
        }
        if (G > 1) for (int g = 0; g < G; g++) finish(grp[(size_t)g], n_frames - 1);
        double flops = 0; long obs = 0;
        for (int g = 0; g < G; g++) {
            Group &q = grp[(size_t)g];
            q.eb->marginWait();
            for (int s = 0; s < q.n; s++) {
                tail_lines(out[(size_t)(q.s0 + s)], q.eb->stream(s));
                flops += q.eb->stream(s).solve_flops; obs += q.eb->stream(s).solve_obs;
            }
        }
        for (int s = 0; s < N; s++) {
            if (!digest_only) { std::printf("STR %d\n", s); std::fputs(out[(size_t)s].text.c_str(), stdout); }
            std::printf("DIG %d %016llx\n", s, out[(size_t)s].h);
        }
        grp.clear();          // (the batches print their phase clocks as they go)
        // TIM: lock-step frames timed (INITED), milliseconds per lock-step frame (= N stream frames), streams
        std::printf("TIM %d %.6f %d %d\n", solves, solves ? solve_ms / solves : 0.0, N, G);
        std::printf("FLP %.17g %ld\n", flops, obs);
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "estimator_seq: %s\n", e.what());
        return 1;
    }
}
