// lmono_amd/host/lmono_host.cpp -- see lmono_host.hpp.  Window / track bookkeeping on the host, numerics in the HIP library.
#include "lmono_host.hpp"
#include <exception>
#include <thread>
#include <condition_variable>
#include <mutex>
#include <chrono>
#include <cstdlib>
#include <cstdio>
#include "kitti_io.hpp"
#include <algorithm>
#include <cmath>
#include <cstring>

namespace lmono_host {

// host-side phase clocks of the INITED frame loop (LMONO_HOST_TIMING=1: printed to stderr by ~Estimator): where a frame's wall time goes beside the kernels
namespace {
struct PhaseClock {
    double ms[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    long frames = 0;
    const bool on = std::getenv("LMONO_HOST_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t0;
    void start() { if (on) t0 = std::chrono::steady_clock::now(); }
    void lap(int k) { if (on) { const auto t1 = std::chrono::steady_clock::now(); ms[k] += std::chrono::duration<double, std::milli>(t1 - t0).count(); t0 = t1; } }
} g_clock;
}
void estimator_print_phase_clock()
{
    if (!g_clock.on || g_clock.frames == 0) return;
    static const char *name[8] = { "featureCheck+loop", "triangulate", "pack+update", "solve+read", "double2Matrix+margin", "outliers", "slideWindow", "row" };
    std::fprintf(stderr, "HOSTTIM frames %ld:", g_clock.frames);
    for (int k = 0; k < 8; k++) std::fprintf(stderr, " %s %.4f", name[k], g_clock.ms[k] / g_clock.frames);
    std::fprintf(stderr, " (ms per frame)\n");
}

static void mat_mul(const double *A, const double *B, double *C) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j]; }
static void mat_vec(const double *A, const double *v, double *o) { for (int i = 0; i < 3; i++) o[i] = A[i * 3] * v[0] + A[i * 3 + 1] * v[1] + A[i * 3 + 2] * v[2]; }

// Eigen::Quaterniond(Matrix3d) / q.normalized().toRotationMatrix() as used by matrix2Double / double2Matrix
static void R_to_q(const double *m, double *q)
{
    double t = m[0] + m[4] + m[8];
    if (t > 0) { t = std::sqrt(t + 1.0); q[3] = 0.5 * t; t = 0.5 / t; q[0] = (m[7] - m[5]) * t; q[1] = (m[2] - m[6]) * t; q[2] = (m[3] - m[1]) * t; }
    else {
        int i = 0; if (m[4] > m[0]) i = 1; if (m[8] > m[i * 3 + i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0);
        q[i] = 0.5 * t; t = 0.5 / t; q[3] = (m[k * 3 + j] - m[j * 3 + k]) * t; q[j] = (m[j * 3 + i] + m[i * 3 + j]) * t; q[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
    }
}
static void q_to_R(const double *qin, double *R)
{
    const double n = std::sqrt(qin[0] * qin[0] + qin[1] * qin[1] + qin[2] * qin[2] + qin[3] * qin[3]);
    const double x = qin[0] / n, y = qin[1] / n, z = qin[2] / n, w = qin[3] / n;
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z, twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy; R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx; R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}

// ---- FeatureManager ---------------------------------------------------------------------------------------------
int FeatureManager::getFeatureCount()
{
    int cnt = 0;
    for (auto &it : feature) { it.used_num = (int)it.feature_per_frame.size(); if (it.used_num >= params->TRACK_CNT) cnt++; }
    return cnt;
}
std::vector<double> FeatureManager::getDepthVector()
{
    std::vector<double> dep;
    for (auto &it : feature) { it.used_num = (int)it.feature_per_frame.size(); if (it.used_num < params->TRACK_CNT) continue; dep.push_back(1.0 / it.estimated_depth); }
    return dep;
}
void FeatureManager::setDepth(const std::vector<double> &x)
{
    int idx = -1;
    for (auto &it : feature) {
        it.used_num = (int)it.feature_per_frame.size();
        if (it.used_num < params->TRACK_CNT) continue;
        it.estimated_depth = 1.0 / x[++idx];
        it.solve_flag = (it.estimated_depth < 0.1 || it.estimated_depth > 300) ? 2 : 1;
    }
}
void FeatureManager::removeFailures()
{
    for (auto it = feature.begin(); it != feature.end();) { if (it->solve_flag == 2) it = feature.erase(it); else ++it; }
}
void FeatureManager::removeOutlier(const std::set<int> &ids)
{
    for (auto it = feature.begin(); it != feature.end();) { if (ids.count(it->feature_id)) it = feature.erase(it); else ++it; }
}
void FeatureManager::pack(std::vector<int> &start, std::vector<int> &off, std::vector<double> &pts, std::vector<double> &depth, bool all_tracks)
{
    start.clear(); off.assign(1, 0); pts.clear(); depth.clear();
    for (auto &it : feature) {
        it.used_num = (int)it.feature_per_frame.size();
        if (!all_tracks && it.used_num < params->TRACK_CNT) continue;
        start.push_back(it.start_frame);
        for (auto &f : it.feature_per_frame) { pts.push_back(f.pt[0]); pts.push_back(f.pt[1]); }
        off.push_back((int)pts.size() / 2);
        depth.push_back(it.estimated_depth);
    }
}
void FeatureManager::triangulate(int, const Mat3 Rs[], const Vec3 Ps[], const double tlc[16])
{
    std::vector<int> start, off; std::vector<double> pts, depth;
    pack(start, off, pts, depth, true);      // tracks below TRACK_CNT are skipped inside the kernels
    if (start.empty()) return;
    const int feat_off[2] = { 0, (int)start.size() };
    std::vector<double> R(99, 0.0), P(33, 0.0);
    for (int i = 0; i <= WINDOW_SIZE; i++) { std::memcpy(&R[9 * i], Rs[i].m, 72); std::memcpy(&P[3 * i], Ps[i].v, 24); }
    std::vector<int> flag(start.size(), 0);
    hip->check(lmono_triangulate(hip->get(), 1, feat_off, R.data(), P.data(), tlc, start.data(), off.data(), pts.data(), depth.data(), flag.data(),
                                 params->TRACK_CNT, WINDOW_SIZE, params->FACTOR_WEIGHT, 50), "lmono_triangulate");
    size_t k = 0;
    for (auto &it : feature) { it.estimated_depth = depth[k]; if (it.used_num >= params->TRACK_CNT) it.solve_flag = flag[k]; k++; }
}
void FeatureManager::removeBackShiftDepth(const Mat3 &R0, const Vec3 &P0, const Mat3 &R1, const Vec3 &P1, const double tlc[16])
{
    std::vector<double> pt, dep;
    for (auto &it : feature) if (it.start_frame == 0 && it.feature_per_frame.size() >= 3) { pt.push_back(it.feature_per_frame[0].pt[0]); pt.push_back(it.feature_per_frame[0].pt[1]); dep.push_back(it.estimated_depth); }
    std::vector<double> out(dep.size());
    if (!dep.empty()) hip->check(lmono_shift_depth(hip->get(), R0.m, P0.v, R1.m, P1.v, tlc, (int)dep.size(), pt.data(), dep.data(), out.data()), "lmono_shift_depth");
    size_t k = 0;
    for (auto it = feature.begin(); it != feature.end();) {
        if (it->start_frame != 0) { it->start_frame--; ++it; continue; }
        const bool keep = it->feature_per_frame.size() >= 3;      // after erasing the first observation: size >= 2
        it->feature_per_frame.erase(it->feature_per_frame.begin());
        if (!keep) { it = feature.erase(it); continue; }
        it->estimated_depth = out[k++];
        ++it;
    }
}
void FeatureManager::removeBack()
{
    for (auto it = feature.begin(); it != feature.end();) {
        if (it->start_frame != 0) { it->start_frame--; ++it; continue; }
        it->feature_per_frame.erase(it->feature_per_frame.begin());
        if (it->feature_per_frame.empty()) it = feature.erase(it); else ++it;
    }
}
void FeatureManager::removeFront(int frame_count)
{
    for (auto it = feature.begin(); it != feature.end();) {
        if (it->start_frame == frame_count) { it->start_frame--; ++it; continue; }
        const int j = WINDOW_SIZE - 1 - it->start_frame;
        if (it->endFrame() < frame_count - 1) { ++it; continue; }
        it->feature_per_frame.erase(it->feature_per_frame.begin() + j);
        if (it->feature_per_frame.empty()) it = feature.erase(it); else ++it;
    }
}

void FeatureManager::clearDepth()
{
    for (auto &it : feature) { it.solve_flag = 0; it.estimated_depth = INIT_DEPTH; }
}
double FeatureManager::computeParallax(const FeaturePerId &it_per_id, int frame_count) const
{
    const FeaturePerFrame &frame_i = it_per_id.feature_per_frame[frame_count - 2 - it_per_id.start_frame];
    const FeaturePerFrame &frame_j = it_per_id.feature_per_frame[frame_count - 1 - it_per_id.start_frame];
    const double du = frame_i.uv[0] - frame_j.uv[0], dv = frame_i.uv[1] - frame_j.uv[1];
    return std::max(0.0, std::sqrt(du * du + dv * dv));
}
bool FeatureManager::featureCheck(int frame_count, const Image &image, double)
{
    double parallax_sum = 0;
    int parallax_num = 0;
    last_track_num = 0; long_track_num = 0; new_feature_num = 0;
    for (const auto &id_pts : image) {
        FeaturePerFrame f_per_fra;
        f_per_fra.pt[0] = id_pts.second[0]; f_per_fra.pt[1] = id_pts.second[1]; f_per_fra.uv[0] = id_pts.second[2]; f_per_fra.uv[1] = id_pts.second[3];
        const int feature_id = id_pts.first;
        auto it = std::find_if(feature.begin(), feature.end(), [feature_id](const FeaturePerId &f) { return f.feature_id == feature_id; });
        if (it == feature.end()) {
            FeaturePerId f; f.feature_id = feature_id; f.start_frame = frame_count;
            f.feature_per_frame.push_back(f_per_fra);
            feature.push_back(f);
            new_feature_num++;
        } else {
            it->feature_per_frame.push_back(f_per_fra);
            last_track_num++;
            if ((int)it->feature_per_frame.size() >= params->TRACK_CNT) long_track_num++;
        }
    }
    if (frame_count < 2 || last_track_num < 20 || new_feature_num > 0.5 * last_track_num) return true;
    for (auto &it : feature)
        if (it.start_frame <= frame_count - 2 && it.start_frame + (int)it.feature_per_frame.size() - 1 >= frame_count - 1) {
            parallax_sum += computeParallax(it, frame_count);
            parallax_num++;
        }
    if (parallax_num == 0) return true;
    return parallax_sum / parallax_num >= params->FEATURE_THRESHOLD;
}

// ---- Estimator ----------------------------------------------------------------------------------------------------
Estimator::Estimator(HipContext &hip, const Params &p) : hip_(hip), p_(p)
{
    for (auto &R : Rs) { std::memset(R.m, 0, sizeof(R.m)); R.m[0] = R.m[4] = R.m[8] = 1.0; }
    for (auto &P : Ps) std::memset(P.v, 0, sizeof(P.v));
    std::memset(TLC, 0, sizeof(TLC)); TLC[0] = TLC[5] = TLC[10] = TLC[15] = 1.0;
    feature_manager.params = &p_; feature_manager.hip = &hip_;
}
struct Estimator::MarginWorker {
    std::mutex mu;
    std::condition_variable cv;
    std::function<void()> job;          // pending job (empty: none)
    bool busy = false, quit = false;
    std::exception_ptr error;
    std::thread th;
    MarginWorker() : th([this] { run(); }) {}
    ~MarginWorker() { { std::lock_guard<std::mutex> g(mu); quit = true; } cv.notify_all(); th.join(); }
    void run()
    {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [this] { return quit || (bool)job; });
            if (!job) return;                                  // quit with nothing pending
            std::function<void()> j;
            j.swap(job);
            lk.unlock();
            std::exception_ptr e;
            try { j(); } catch (...) { e = std::current_exception(); }
            lk.lock();
            if (e) error = e;
            busy = false;
            cv.notify_all();
        }
    }
    void submit(std::function<void()> j)
    {
        { std::lock_guard<std::mutex> g(mu); job = std::move(j); busy = true; }
        cv.notify_all();
    }
    void wait()                                                // rethrows what the job threw
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [this] { return !busy; });
        if (error) { std::exception_ptr e = error; error = nullptr; std::rethrow_exception(e); }
    }
};
Estimator::~Estimator()
{
    try { marginWait(); } catch (...) {}
    margin_worker_.reset();
    if (ba_batch_) lmono_ba_batch_destroy(ba_batch_);
}
void Estimator::marginWait() { if (margin_worker_) margin_worker_->wait(); }     // rethrows what the worker threw
void Estimator::marginSubmit(std::function<void()> job)
{
    if (!async_margin_) { job(); return; }
    if (!margin_worker_) margin_worker_.reset(new MarginWorker());
    margin_worker_->submit(std::move(job));
}
void Estimator::setAsyncMargin(bool on)
{
    marginWait();
    if (on && !margin_hip_) { margin_hip_.reset(new HipContext(hip_.device())); margin_hip_->useOwnStream(); }
    async_margin_ = on;
}
void Estimator::matrix2Double()
{
    for (int i = 0; i <= WINDOW_SIZE; i++) { std::memcpy(para_pose[i], Ps[i].v, 24); R_to_q(Rs[i].m, para_pose[i] + 3); }
    double R[9];
    for (int i = 0; i < 3; i++) { para_ex[0][i] = TLC[i * 4 + 3]; for (int j = 0; j < 3; j++) R[i * 3 + j] = TLC[i * 4 + j]; }
    R_to_q(R, para_ex[0] + 3);
    para_depth_inv = feature_manager.getDepthVector();
}
void Estimator::double2Matrix()
{
    double R00[9], rot_diff[9];
    q_to_R(para_pose[0] + 3, R00);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) rot_diff[i * 3 + j] = Rs[0].m[i * 3] * R00[j * 3] + Rs[0].m[i * 3 + 1] * R00[j * 3 + 1] + Rs[0].m[i * 3 + 2] * R00[j * 3 + 2];
    const Vec3 origin_t0 = Ps[0];
    for (int i = 0; i <= WINDOW_SIZE; i++) {
        const double t[3] = { para_pose[i][0] - para_pose[0][0], para_pose[i][1] - para_pose[0][1], para_pose[i][2] - para_pose[0][2] };
        double R[9], rt[3];
        q_to_R(para_pose[i] + 3, R);
        mat_vec(rot_diff, t, rt);
        for (int k = 0; k < 3; k++) Ps[i].v[k] = rt[k] + origin_t0.v[k];
        mat_mul(rot_diff, R, Rs[i].m);
    }
    double Rx[9];
    q_to_R(para_ex[0] + 3, Rx);
    for (int i = 0; i < 3; i++) { TLC[i * 4 + 3] = para_ex[0][i]; for (int j = 0; j < 3; j++) TLC[i * 4 + j] = Rx[i * 3 + j]; }
    feature_manager.setDepth(para_depth_inv);
    feature_manager.removeFailures();
    loop_closure = false;                                   // :1111-1120
}
bool Estimator::optimization()
{
    matrix2Double();
    // residual blocks exactly as Estimator.cc:1155-1215 adds them
    std::vector<int> obs_feat, obs_i, obs_j; std::vector<double> obs_pts;
    const bool use_mono = p_.ESTIMATE_LASER && !static_status;
    int feature_index = -1;
    for (auto &it : feature_manager.feature) {
        it.used_num = (int)it.feature_per_frame.size();
        if (it.used_num < p_.TRACK_CNT) continue;
        ++feature_index;
        const int i = it.start_frame; int j = i - 1;
        for (auto &f : it.feature_per_frame) {
            j++;
            if (i == j) continue;
            obs_feat.push_back(feature_index); obs_i.push_back(i); obs_j.push_back(j);
            obs_pts.insert(obs_pts.end(), { it.feature_per_frame[0].pt[0], it.feature_per_frame[0].pt[1], f.pt[0], f.pt[1] });
        }
    }
    const int F = feature_index + 1;
    const int feat_off[2] = { 0, F }, obs_off[2] = { 0, (int)obs_feat.size() };
    int use_prior = 0;
    if (first_refine >= p_.FINE_TIMES) use_prior = 1; else first_refine++;
    const int flags[4] = { frame_count + 1, use_prior, p_.ESTIMATE_LASER == 0 ? 1 : 0, use_mono ? 1 : 0 };
    std::vector<double> poses(77, 0.0), laser(240, 0.0);
    for (int i = 0; i <= WINDOW_SIZE; i++) std::memcpy(&poses[7 * i], para_pose[i], 56);
    for (int i = 0; i < frame_count; i++) {
        double *c = &laser[24 * i];
        std::memcpy(c, L0_R[i].m, 72); std::memcpy(c + 9, L0_R[i + 1].m, 72); std::memcpy(c + 18, L0_T[i].v, 24); std::memcpy(c + 21, L0_T[i + 1].v, 24);
    }
    double laser_info[36] = { 0 }, mono_info[4] = { p_.FACTOR_WEIGHT, 0, 0, p_.FACTOR_WEIGHT }, prior_w[2] = { p_.PRIOR_T, p_.PRIOR_R };
    for (int k = 0; k < 6; k++) laser_info[k * 7] = p_.LASER_W * p_.FACTOR_WEIGHT;
    lmono_ba_desc d{};
    d.n_windows = 1; d.feat_off = feat_off; d.obs_off = obs_off; d.flags = flags; d.poses = poses.data(); d.ex = para_ex[0];
    d.inv_depth = para_depth_inv.data(); d.obs_feat = obs_feat.data(); d.obs_i = obs_i.data(); d.obs_j = obs_j.data(); d.obs_pts = obs_pts.data();
    d.laser_consts = laser.data(); d.prior_T = TLC; d.laser_info = laser_info; d.mono_info = mono_info; d.prior_w = prior_w;
    // the reference builds a new ceres::Problem per call (Estimator.cc:1017); here the device arrays of the previous frame's
    // problem are loaded again in place, so the steady-state frame loop does not allocate
    if (!ba_batch_) {
        ba_batch_ = lmono_ba_batch_create(hip_.get(), &d);
        if (!ba_batch_) throw std::runtime_error(std::string("lmono_ba_batch_create: ") + lmono_last_error(hip_.get()));
    } else
        hip_.check(lmono_ba_batch_update(hip_.get(), ba_batch_, &d), "lmono_ba_batch_update");
    g_clock.lap(2);
    lmono_ba_batch *b = ba_batch_;
    hip_.check(lmono_ba_solve(hip_.get(), b, p_.NUM_ITERATIONS), "lmono_ba_solve");
    double summary[6];
    std::vector<double> invd((size_t)std::max(F, 1));
    hip_.check(lmono_ba_batch_read(hip_.get(), b, poses.data(), para_ex[0], invd.data(), summary), "lmono_ba_batch_read");
    g_clock.lap(3);
    for (int i = 0; i <= WINDOW_SIZE; i++) std::memcpy(para_pose[i], &poses[7 * i], 56);
    if (use_mono) para_depth_inv.assign(invd.begin(), invd.begin() + F);
    initial_cost = summary[0]; final_cost = summary[1]; iterations = (int)summary[2]; termination = (int)summary[3];
    solve_flops += (double)iterations * (2000.0 * (use_mono ? (double)obs_feat.size() : 0.0) + 72.0 * 72.0 * 72.0 / 3.0);
    solve_obs += (long)obs_feat.size();
    double2Matrix();
    if (frame_count < WINDOW_SIZE) return false;
    if (p_.ESTIMATE_LASER) margin();                         // Estimator.cc:1288-1291
    g_clock.lap(4);
    return termination == 0 || final_cost < 5e-3;            // Estimator.cc:1293
}
void Estimator::outliersRejection(std::set<int> &removeIndex, const double &error)
{
    std::vector<int> start, off; std::vector<double> pts, depth;
    feature_manager.pack(start, off, pts, depth, true);
    if (start.empty()) return;
    const int feat_off[2] = { 0, (int)start.size() };
    std::vector<double> R(99, 0.0), P(33, 0.0), score(start.size());
    for (int i = 0; i <= WINDOW_SIZE; i++) { std::memcpy(&R[9 * i], Rs[i].m, 72); std::memcpy(&P[3 * i], Ps[i].v, 24); }
    hip_.check(lmono_outlier_scores(hip_.get(), 1, feat_off, R.data(), P.data(), TLC, start.data(), off.data(), pts.data(), depth.data(),
                                    p_.TRACK_CNT, p_.FACTOR_WEIGHT, score.data()), "lmono_outlier_scores");
    size_t k = 0;
    for (auto &it : feature_manager.feature) { if (score[k] >= 0 && score[k] > error) removeIndex.insert(it.feature_id); k++; }
}
void Estimator::margin()
{
    marginWait();                                             // the previous prior is this call's input (and its storage is reused)
    HipContext *h = async_margin_ ? margin_hip_.get() : &hip_;
    MarginalizationInfo &mi = last_marginalization_info;
    if (marginalization_flag != MARGIN_OLD) {
        // :1406-1470: the previous prior, as the only factor, loses the block that aliases para_pose[WINDOW_SIZE - 1]
        if (!mi.present) return;
        const auto it = std::find(mi.parameter_blocks.begin(), mi.parameter_blocks.end(), WINDOW_SIZE - 1);
        if (it == mi.parameter_blocks.end()) return;
        const int drop = (int)(it - mi.parameter_blocks.begin()), nb = (int)mi.parameter_blocks.size();
        matrix2Double();
        std::vector<double> x((size_t)nb * 7);
        for (int k = 0; k < nb; k++) std::memcpy(&x[7 * (size_t)k], mi.parameter_blocks[k] < 0 ? para_ex[0] : para_pose[mi.parameter_blocks[k]], 56);
        margin_calls[1]++;
        auto job = [h, &mi, drop, nb, x]() mutable {
            const int n = 6 * (nb - 1);
            std::vector<double> J((size_t)n * n), r((size_t)n);
            int status = 0;
            h->check(lmono_marg_second_new(h->get(), 1, nb, drop, mi.linearized_jacobians.data(), mi.linearized_residuals.data(),
                                           mi.keep_block_data.data(), x.data(), J.data(), r.data(), &status), "lmono_marg_second_new");
            mi.linearized_jacobians.swap(J); mi.linearized_residuals.swap(r);
            x.erase(x.begin() + 7 * (size_t)drop, x.begin() + 7 * (size_t)(drop + 1));
            mi.keep_block_data.swap(x);                            // parameter_block_data: the values at this marginalisation
            mi.parameter_blocks.erase(mi.parameter_blocks.begin() + drop);   // addr_shift :1442-1455: pose i -> pose i (i < 9), pose 10 -> pose 9 is not a block
            mi.m = 6; mi.n = n; mi.status = status; mi.valid = false;
        };
        marginSubmit(std::move(job));
        return;
    }
    matrix2Double();
    struct Pack {
        std::vector<int> obs_feat, obs_j; std::vector<double> obs_pts, invd;
        int f0 = 0;
        double poses[77], ex[7], laser01[24], laser_info[36] = { 0 }, mono_info[4];
    };
    auto pk = std::make_shared<Pack>();
    int feature_index = -1;
    for (auto &it : feature_manager.feature) {
        it.used_num = (int)it.feature_per_frame.size();
        if (it.used_num < p_.TRACK_CNT) continue;
        ++feature_index;
        if (it.start_frame != 0) continue;
        int j = -1;
        for (auto &f : it.feature_per_frame) {
            j++;
            if (j == 0) continue;
            pk->obs_feat.push_back(pk->f0); pk->obs_j.push_back(j);
            pk->obs_pts.insert(pk->obs_pts.end(), { it.feature_per_frame[0].pt[0], it.feature_per_frame[0].pt[1], f.pt[0], f.pt[1] });
        }
        pk->invd.push_back(para_depth_inv[feature_index]);
        pk->f0++;
    }
    for (int i = 0; i <= WINDOW_SIZE; i++) std::memcpy(pk->poses + 7 * i, para_pose[i], 56);
    std::memcpy(pk->ex, para_ex[0], 56);
    std::memcpy(pk->laser01, L0_R[0].m, 72); std::memcpy(pk->laser01 + 9, L0_R[1].m, 72);
    std::memcpy(pk->laser01 + 18, L0_T[0].v, 24); std::memcpy(pk->laser01 + 21, L0_T[1].v, 24);
    for (int k = 0; k < 6; k++) pk->laser_info[k * 7] = p_.LASER_W * p_.FACTOR_WEIGHT;
    pk->mono_info[0] = pk->mono_info[3] = p_.FACTOR_WEIGHT; pk->mono_info[1] = pk->mono_info[2] = 0;
    margin_calls[0]++;
    auto job = [h, &mi, pk]() {
        const int feat_off[2] = { 0, pk->f0 }, obs_off[2] = { 0, (int)pk->obs_feat.size() };
        mi.linearized_jacobians.assign(66 * 66, 0.0); mi.linearized_residuals.assign(66, 0.0);
        const int dummy = 0; const double dzero = 0.0;
        h->check(lmono_marginalize(h->get(), 1, feat_off, obs_off, pk->poses, pk->ex, pk->invd.empty() ? &dzero : pk->invd.data(),
                                   pk->obs_feat.empty() ? &dummy : pk->obs_feat.data(), pk->obs_j.empty() ? &dummy : pk->obs_j.data(),
                                   pk->obs_pts.empty() ? &dzero : pk->obs_pts.data(), pk->laser01, pk->laser_info, pk->mono_info,
                                   mi.linearized_jacobians.data(), mi.linearized_residuals.data(), &mi.status), "lmono_marginalize");
        mi.m = 6 + pk->f0; mi.n = 66;
        mi.keep_block_data.assign(77, 0.0);
        std::memcpy(mi.keep_block_data.data(), pk->ex, 56);
        for (int i = 1; i <= WINDOW_SIZE; i++) std::memcpy(mi.keep_block_data.data() + 7 * i, pk->poses + 7 * i, 56);
        // addr_shift :1390-1396: the kept blocks ex, pose1 .. pose10 alias para_ex[0], para_pose[0] .. para_pose[9] from now on
        mi.parameter_blocks.assign(1, -1);
        for (int i = 0; i < WINDOW_SIZE; i++) mi.parameter_blocks.push_back(i);
        mi.present = true;
        mi.valid = false;      // never set by the reference either
    };
    marginSubmit(std::move(job));
}

void Estimator::slideWindow()
{
    if (marginalization_flag == MARGIN_OLD) {
        back_R0 = Rs[0]; back_P0 = Ps[0];
        if (frame_count == WINDOW_SIZE) {
            for (int i = 0; i < frame_count; i++) {
                Header[i] = Header[i + 1];
                std::swap(Rs[i], Rs[i + 1]); std::swap(Ps[i], Ps[i + 1]);
                std::swap(L0_R[i], L0_R[i + 1]); std::swap(L0_T[i], L0_T[i + 1]);     // all_image_frame.erase(begin()): slot 10 is rewritten by the next frame
            }
            Rs[WINDOW_SIZE] = Rs[WINDOW_SIZE - 1]; Ps[WINDOW_SIZE] = Ps[WINDOW_SIZE - 1]; Header[WINDOW_SIZE] = Header[WINDOW_SIZE - 1];
            if (stage_flag == NOT_INITED) feature_manager.removeBack();                    // slideWindowOld :744-768
            else feature_manager.removeBackShiftDepth(back_R0, back_P0, Rs[0], Ps[0], TLC);
        }
    } else if (frame_count == WINDOW_SIZE) {
        Header[frame_count - 1] = Header[frame_count];
        Ps[frame_count - 1] = Ps[frame_count]; Rs[frame_count - 1] = Rs[frame_count];
        // all_image_frame.erase(all_image_frame.end() - 1) (:735) removes the LAST element -- the newest frame's LiDAR pose -- while slot
        // WINDOW_SIZE - 1 takes the newest Ps / Rs / Header.  Reproduced as written: L0_R / L0_T[WINDOW_SIZE - 1] keep the second-newest
        // frame's pose and slot WINDOW_SIZE is rewritten by the next frame.
        feature_manager.removeFront(frame_count);                                          // slideWindowNew
    }
}

// ---- frame loop -------------------------------------------------------------------------------------------------------
void Estimator::processCompactData(const double L0_Pos[16])
{
    const double d[3] = { L0_Pos[3] - last_laser_t.v[0], L0_Pos[7] - last_laser_t.v[1], L0_Pos[11] - last_laser_t.v[2] };
    static_status = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) < 0.1;             // :259-265
    last_laser_t.v[0] = L0_Pos[3]; last_laser_t.v[1] = L0_Pos[7]; last_laser_t.v[2] = L0_Pos[11];
}
bool Estimator::runInitialization()
{
    // :986-1012 (the structure-from-motion block above it is commented out in the reference)
    double rlcT[9], tlc[3] = { TLC[3], TLC[7], TLC[11] };
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) rlcT[i * 3 + j] = TLC[j * 4 + i];
    for (int i = 0; i <= frame_count; i++) {
        mat_mul(rlcT, L0_R[i].m, Rs[i].m);
        const double dv[3] = { L0_T[i].v[0] - tlc[0], L0_T[i].v[1] - tlc[1], L0_T[i].v[2] - tlc[2] };
        mat_vec(rlcT, dv, Ps[i].v);
    }
    feature_manager.clearDepth();
    feature_manager.triangulate(frame_count, Rs, Ps, TLC);
    std::set<int> removeIndex;
    outliersRejection(removeIndex, 100.0);
    feature_manager.removeOutlier(removeIndex);
    return true;
}
void Estimator::loopCorrection()
{
    if (loop_buf.empty()) return;
    const LoopFrame lf = loop_buf.back();                 // the while loop of :312-317 keeps the last one
    loop_buf.clear();
    int idx = -1;
    for (int i = 0; i < WINDOW_SIZE; i++) if (lf.loop_time_stamp == Header[i]) idx = i;
    if (idx < 0) return;
    loop_closure = true;
    // Eigen::Quaterniond(w, x, y, z).toRotationMatrix(): the message's quaternion as it is (not normalised)
    const double w = lf.correct_Q[0], x = lf.correct_Q[1], y = lf.correct_Q[2], z = lf.correct_Q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z, twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    const double Rc[9] = { 1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1 - (txx + tyy) };
    const Mat3 Ri = Rs[idx]; const Vec3 Pi = Ps[idx];
    double RiT[9];
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) RiT[a * 3 + b] = Ri.m[b * 3 + a];
    for (int i = 0; i <= WINDOW_SIZE; i++) {
        if (i == idx) continue;
        double rel_r[9], rel_t[3], rt[3];
        mat_mul(RiT, Rs[i].m, rel_r);
        const double dp[3] = { Pi.v[0] - Ps[i].v[0], Pi.v[1] - Ps[i].v[1], Pi.v[2] - Ps[i].v[2] };
        mat_vec(RiT, dp, rel_t);
        mat_mul(Rc, rel_r, Rs[i].m);
        mat_vec(Rc, rel_t, rt);
        for (int k = 0; k < 3; k++) Ps[i].v[k] = lf.correct_T[k] - rt[k];
    }
    std::memcpy(Rs[idx].m, Rc, 72);
    for (int k = 0; k < 3; k++) Ps[idx].v[k] = lf.correct_T[k];
}
bool Estimator::processImage(double header, const FeatureManager::Image &image, const double transform_to_init[16])
{
    g_clock.start();
    processCompactData(transform_to_init);
    const bool keyframe = feature_manager.featureCheck(frame_count, image, header);       // :383-394
    marginalization_flag = keyframe ? MARGIN_OLD : MARGIN_SECOND_NEW;
    Header[frame_count] = header;
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) L0_R[frame_count].m[i * 3 + j] = transform_to_init[i * 4 + j]; L0_T[frame_count].v[i] = transform_to_init[i * 4 + 3]; }   // all_image_frame.push_back
    if (stage_flag == NOT_INITED) {
        if (frame_count == WINDOW_SIZE) {
            if (p_.ESTIMATE_LASER != 2 && runInitialization()) {
                optimization();
                stage_flag = INITED;
                std::set<int> removeIndex;
                outliersRejection(removeIndex, 3);
                feature_manager.removeOutlier(removeIndex);
                slideWindow();
            } else slideWindow();
        }
        if (frame_count < WINDOW_SIZE) {
            frame_count++;
            Ps[frame_count] = Ps[frame_count - 1]; Rs[frame_count] = Rs[frame_count - 1]; Header[frame_count] = Header[frame_count - 1];
        }
    } else {
        loopCorrection();
        g_clock.lap(0);
        feature_manager.triangulate(frame_count, Rs, Ps, TLC);
        g_clock.lap(1);
        std::set<int> removeIndex;
        optimization();
        outliersRejection(removeIndex, p_.OUTLIER_T);
        feature_manager.removeOutlier(removeIndex);
        g_clock.lap(5);
        slideWindow();
        g_clock.lap(6);
        g_clock.frames++;
    }
    if (stage_flag == INITED) {                             // new_odometry.txt row, :634-645
        std::array<double, 8> row;
        row[0] = Header[WINDOW_SIZE];
        for (int k = 0; k < 3; k++) row[1 + k] = Ps[WINDOW_SIZE].v[k];
        R_to_q(Rs[WINDOW_SIZE].m, &row[4]);
        new_odometry.push_back(row);
    }
    return keyframe;
}

// ---- A-LOAM nodes ---------------------------------------------------------------------------------------------------
ScanRegistration::ScanRegistration(HipContext &hip, int n_scans_cap, int64_t points_cap, int N_SCANS, float MINIMUM_RANGE)
    : hip_(hip), batch_(lmono_batch_create(hip.get(), n_scans_cap, points_cap)), N_SCANS_(N_SCANS), MINIMUM_RANGE_(MINIMUM_RANGE), cap_(points_cap)
{
    if (!batch_) throw std::runtime_error(std::string("lmono_batch_create: ") + lmono_last_error(hip.get()));
}
ScanRegistration::~ScanRegistration() { lmono_batch_destroy(batch_); }
void ScanRegistration::laserCloudHandler(const float *xyzi_d, const int64_t *offsets_h, int n_scans)
{
    hip_.check(lmono_scanreg_batch(hip_.get(), batch_, xyzi_d, offsets_h, n_scans, N_SCANS_, MINIMUM_RANGE_), "lmono_scanreg_batch");
    n_ = n_scans;
}
void ScanRegistration::laserCloudHandlerHost(const float *xyzi_h, const int64_t *offsets_h, int n_scans)
{
    hip_.check(lmono_scanreg_batch_h(hip_.get(), batch_, xyzi_h, offsets_h, n_scans, N_SCANS_, MINIMUM_RANGE_), "lmono_scanreg_batch_h");
    n_ = n_scans;
}
std::vector<float> ScanRegistration::cloud(int scan, int which)
{
    std::vector<float> out((size_t)cap_ * 4);
    const int n = lmono_batch_get_cloud(hip_.get(), batch_, scan, which, out.data(), (int)cap_);
    hip_.check(n, "lmono_batch_get_cloud");
    out.resize((size_t)n * 4);
    return out;
}
std::vector<double> LaserOdometry::process(ScanRegistration &reg, int n_chains, int lead)
{
    std::vector<double> poses((size_t)reg.n_scans() * 7);
    hip_.check(lmono_odom_batch(hip_.get(), reg.batch(), n_chains, lead, nullptr, poses.data()), "lmono_odom_batch");
    return poses;
}

LaserOdometryNode::LaserOdometryNode(HipContext &hip, int max_points_per_scan, int n_lines, float minimum_range, int history)
    : hip_(hip), stream_(lmono_odom_stream_create(hip.get(), max_points_per_scan, n_lines, minimum_range, history))
{
    if (!stream_) throw std::runtime_error(std::string("lmono_odom_stream_create: ") + lmono_last_error(hip.get()));
}
LaserOdometryNode::~LaserOdometryNode() { lmono_odom_stream_destroy(stream_); }
void LaserOdometryNode::laserCloudHandler(const float *xyzi, int n_points)
{
    int32_t st[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    hip_.check(lmono_odom_step(hip_.get(), stream_, xyzi, n_points, 0, 0, q_last_curr, t_last_curr, q_w_curr, t_w_curr, st), "lmono_odom_step");
    for (int k = 0; k < 8; k++) info[k] = st[k];
}
lmono_scan_batch *LaserOdometryNode::batch() const
{
    lmono_scan_batch *b = nullptr;
    (void)lmono_odom_stream_scan(stream_, &b, nullptr);
    return b;
}
int LaserOdometryNode::scan() const
{
    int s = -1;
    (void)lmono_odom_stream_scan(stream_, nullptr, &s);
    return s;
}

// ---- laserMapping ---------------------------------------------------------------------------------------------------
LaserMapping::LaserMapping(HipContext &hip, float lineRes, float planeRes) : hip_(hip), mapper_(lmono_mapper_create(hip.get(), lineRes, planeRes))
{
    if (!mapper_) throw std::runtime_error(std::string("lmono_mapper_create: ") + lmono_last_error(hip.get()));
}
LaserMapping::~LaserMapping() { lmono_mapper_destroy(mapper_); }
void LaserMapping::process(ScanRegistration &reg, int scan, const double q_wodom_curr[4], const double t_wodom_curr[3], double q_w_curr[4], double t_w_curr[3])
{
    int32_t st[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    hip_.check(lmono_mapper_process(hip_.get(), mapper_, reg.batch(), scan, q_wodom_curr, t_wodom_curr, q_w_curr, t_w_curr, st), "lmono_mapper_process");
    for (int k = 0; k < 8; k++) stats[k] = st[k];
}
void LaserMapping::process(LaserOdometryNode &node, double q_w_curr[4], double t_w_curr[3])
{
    int32_t st[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    hip_.check(lmono_mapper_process(hip_.get(), mapper_, node.batch(), node.scan(), node.q_w_curr, node.t_w_curr, q_w_curr, t_w_curr, st), "lmono_mapper_process");
    for (int k = 0; k < 8; k++) stats[k] = st[k];
}
std::vector<float> LaserMapping::cube(int which, int i, int j, int k)
{
    const int n = lmono_mapper_cube(hip_.get(), mapper_, which, i, j, k, nullptr, 0);
    hip_.check(n < 0 ? n : 0, "lmono_mapper_cube");
    std::vector<float> out((size_t)(n > 0 ? n : 1) * 4);
    if (n > 0) hip_.check(lmono_mapper_cube(hip_.get(), mapper_, which, i, j, k, out.data(), n) < 0 ? -1 : 0, "lmono_mapper_cube");
    out.resize((size_t)n * 4);
    return out;
}

// ---- map builder (colour projection) -----------------------------------------------------------------------------------
MapBuilder::MapBuilder(HipContext &hip, const lmono_camera &camera, bool save_map, const std::string &map_dir, int max_cloud_points)
    : hip_(hip), mb_(lmono_map_builder_create(hip.get(), &camera, max_cloud_points, (int64_t)10 * camera.width * camera.height)), cam_(camera),
      save_map_(save_map), map_dir_(map_dir)
{
    if (!mb_) throw std::runtime_error(std::string("lmono_map_builder_create: ") + lmono_last_error(hip.get()));
}
MapBuilder::~MapBuilder() { lmono_map_builder_destroy(mb_); }
int MapBuilder::associateToMap(const double Q[4], const double T[3], const float *xyzi, int n_points, const double rlc[9], const double tlc[3],
                               const uint8_t *frame, double)
{
    // map_build_node.cc:216-220: transformation = [rlc^T | (-1) * rlc^T * tlc]
    double M[16] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1 };
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) M[4 * i + j] = rlc[3 * j + i];
        M[4 * i + 3] = (-1.0 * rlc[i]) * tlc[0] + (-1.0 * rlc[3 + i]) * tlc[1] + (-1.0 * rlc[6 + i]) * tlc[2];
    }
    int n = 0;
    hip_.check(lmono_associate_to_map(hip_.get(), mb_, xyzi, n_points, M, frame, Q, T, &n), "lmono_associate_to_map");
    pending_ = true;
    return n;
}
std::string MapBuilder::processMapping()
{
    std::string written;
    if (!pending_) return written;
    pending_ = false;
    map_index++;                                                        // Map_Builder.cc:64
    if (map_index > 0 && map_index % 10 == 0) {                         // :72
        if (save_map_) {
            const std::vector<lmono_point_rgb> m = rgbMap();
            written = rgb_map_path(map_dir_, map_index);
            static_assert(sizeof(PointRgb) == sizeof(lmono_point_rgb), "point layout");
            if (!write_ply_binary(written, reinterpret_cast<const PointRgb *>(m.data()), m.size())) throw std::runtime_error("cannot write " + written);
        }
        hip_.check(lmono_map_builder_clear(hip_.get(), mb_), "lmono_map_builder_clear");   // :81
    }
    return written;
}
std::vector<lmono_point_rgb> MapBuilder::rgbCloud(int which)
{
    const int n = lmono_map_builder_cloud(hip_.get(), mb_, which, nullptr, 0);
    hip_.check(n < 0 ? n : 0, "lmono_map_builder_cloud");
    std::vector<lmono_point_rgb> out((size_t)n);
    if (n > 0) hip_.check(lmono_map_builder_cloud(hip_.get(), mb_, which, out.data(), n) < 0 ? -1 : 0, "lmono_map_builder_cloud");
    return out;
}
std::vector<lmono_point_rgb> MapBuilder::rgbMap()
{
    const int64_t n = lmono_map_builder_map(hip_.get(), mb_, nullptr, 0);
    hip_.check(n < 0 ? (int)n : 0, "lmono_map_builder_map");
    std::vector<lmono_point_rgb> out((size_t)n);
    if (n > 0) hip_.check(lmono_map_builder_map(hip_.get(), mb_, out.data(), n) < 0 ? -1 : 0, "lmono_map_builder_map");
    return out;
}
std::vector<uint8_t> MapBuilder::depthMap()
{
    std::vector<uint8_t> d((size_t)cam_.width * cam_.height);
    hip_.check(lmono_map_builder_depth(hip_.get(), mb_, d.data()), "lmono_map_builder_depth");
    return d;
}

// ---- loop-closure pose graph (new feature) ------------------------------------------------------------------------------
int PoseGraph::optimize4DoF(std::vector<double> &poses_tq, const std::vector<Loop> &loops, int max_iter)
{
    const int n = (int)(poses_tq.size() / 7);
    std::vector<int32_t> idx(loops.size() * 2);
    std::vector<double> info(loops.size() * 8);
    for (size_t k = 0; k < loops.size(); k++) {
        idx[2 * k] = loops[k].old_index; idx[2 * k + 1] = loops[k].cur_index;
        for (int q = 0; q < 8; q++) info[8 * k + (size_t)q] = loops[k].loop_info[q];
    }
    lmono_pose_graph *g = lmono_pose_graph_create(hip_.get(), n, poses_tq.data(), (int)loops.size(), idx.data(), info.data());
    if (!g) throw std::runtime_error(std::string("lmono_pose_graph_create: ") + lmono_last_error(hip_.get()));
    double st[6] = { 0, 0, 0, 0, 0, 0 };
    int rc = lmono_pose_graph_optimize(hip_.get(), g, max_iter);
    if (rc == 0) rc = lmono_pose_graph_result(hip_.get(), g, poses_tq.data(), st);
    lmono_pose_graph_destroy(g);
    hip_.check(rc, "lmono_pose_graph_optimize");
    initial_cost = st[1]; final_cost = st[2]; bandwidth = (int)st[3];
    return (int)st[0];
}

} // namespace lmono_host
