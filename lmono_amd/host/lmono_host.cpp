// lmono_amd/host/lmono_host.cpp -- see lmono_host.hpp.  Window / track bookkeeping on the host, numerics in the HIP library.
#include "lmono_host.hpp"
#include <exception>
#include <thread>
#include <condition_variable>
#include <mutex>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstdio>
#include "kitti_io.hpp"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <unordered_map>
#include <iterator>

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
    triangulateApply(depth.data(), flag.data());
}
void FeatureManager::triangulateApply(const double *depth, const int *flag)
{
    size_t k = 0;
    for (auto &it : feature) { it.estimated_depth = depth[k]; if (it.used_num >= params->TRACK_CNT) it.solve_flag = flag[k]; k++; }
}
void FeatureManager::shiftDepthPack(std::vector<double> &pt, std::vector<double> &dep) const
{
    pt.clear(); dep.clear();
    for (auto &it : feature) if (it.start_frame == 0 && it.feature_per_frame.size() >= 3) { pt.push_back(it.feature_per_frame[0].pt[0]); pt.push_back(it.feature_per_frame[0].pt[1]); dep.push_back(it.estimated_depth); }
}
void FeatureManager::shiftDepthApply(const double *out)
{
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
void FeatureManager::removeBackShiftDepth(const Mat3 &R0, const Vec3 &P0, const Mat3 &R1, const Vec3 &P1, const double tlc[16])
{
    std::vector<double> pt, dep;
    shiftDepthPack(pt, dep);
    std::vector<double> out(dep.size());
    if (!dep.empty()) hip->check(lmono_shift_depth(hip->get(), R0.m, P0.v, R1.m, P1.v, tlc, (int)dep.size(), pt.data(), dep.data(), out.data()), "lmono_shift_depth");
    shiftDepthApply(out.data());
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
    // the reference looks every observation up with std::find_if over the list (:330-333: first match in list order); an index of the ids -- first
    // occurrence wins, entries appended as the tracks are -- finds the same element without the O(tracks) walk per observation
    std::unordered_map<int, std::list<FeaturePerId>::iterator> index;
    index.reserve(feature.size() * 2 + image.size());
    for (auto it = feature.begin(); it != feature.end(); ++it) index.emplace(it->feature_id, it);
    for (const auto &id_pts : image) {
        FeaturePerFrame f_per_fra;
        f_per_fra.pt[0] = id_pts.second[0]; f_per_fra.pt[1] = id_pts.second[1]; f_per_fra.uv[0] = id_pts.second[2]; f_per_fra.uv[1] = id_pts.second[3];
        const int feature_id = id_pts.first;
        const auto found = index.find(feature_id);
        auto it = found == index.end() ? feature.end() : found->second;
        if (it == feature.end()) {
            FeaturePerId f; f.feature_id = feature_id; f.start_frame = frame_count;
            f.feature_per_frame.push_back(f_per_fra);
            feature.push_back(f);
            index.emplace(feature_id, std::prev(feature.end()));
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
class MarginWorker {
public:
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
void Estimator::packSolve(SolvePack &sp)
{
    matrix2Double();
    // residual blocks exactly as Estimator.cc:1155-1215 adds them
    sp.obs_feat.clear(); sp.obs_i.clear(); sp.obs_j.clear(); sp.obs_pts.clear();
    sp.use_mono = p_.ESTIMATE_LASER && !static_status;
    int feature_index = -1;
    for (auto &it : feature_manager.feature) {
        it.used_num = (int)it.feature_per_frame.size();
        if (it.used_num < p_.TRACK_CNT) continue;
        ++feature_index;
        const int i = it.start_frame; int j = i - 1;
        for (auto &f : it.feature_per_frame) {
            j++;
            if (i == j) continue;
            sp.obs_feat.push_back(feature_index); sp.obs_i.push_back(i); sp.obs_j.push_back(j);
            sp.obs_pts.insert(sp.obs_pts.end(), { it.feature_per_frame[0].pt[0], it.feature_per_frame[0].pt[1], f.pt[0], f.pt[1] });
        }
    }
    sp.F = feature_index + 1;
    int use_prior = 0;
    if (first_refine >= p_.FINE_TIMES) use_prior = 1; else first_refine++;
    sp.flags[0] = frame_count + 1; sp.flags[1] = use_prior; sp.flags[2] = p_.ESTIMATE_LASER == 0 ? 1 : 0; sp.flags[3] = sp.use_mono ? 1 : 0;
    std::memset(sp.poses, 0, sizeof(sp.poses)); std::memset(sp.laser, 0, sizeof(sp.laser));
    for (int i = 0; i <= WINDOW_SIZE; i++) std::memcpy(&sp.poses[7 * i], para_pose[i], 56);
    for (int i = 0; i < frame_count; i++) {
        double *c = &sp.laser[24 * i];
        std::memcpy(c, L0_R[i].m, 72); std::memcpy(c + 9, L0_R[i + 1].m, 72); std::memcpy(c + 18, L0_T[i].v, 24); std::memcpy(c + 21, L0_T[i + 1].v, 24);
    }
}
void Estimator::unpackSolve(const SolvePack &sp, const double *poses77, const double *ex7, const double *invd, const double *summary)
{
    for (int i = 0; i <= WINDOW_SIZE; i++) std::memcpy(para_pose[i], &poses77[7 * i], 56);
    std::memcpy(para_ex[0], ex7, 56);
    if (sp.use_mono) para_depth_inv.assign(invd, invd + sp.F);
    initial_cost = summary[0]; final_cost = summary[1]; iterations = (int)summary[2]; termination = (int)summary[3];
    solve_flops += (double)iterations * (2000.0 * (sp.use_mono ? (double)sp.obs_feat.size() : 0.0) + 72.0 * 72.0 * 72.0 / 3.0);
    solve_obs += (long)sp.obs_feat.size();
    double2Matrix();
}
// the sqrt_info statics of Estimator::setParameter (Estimator.cc:94-95) and the PriorFactor weights, as the C ABI takes them
static void solve_infos(const Params &p, double laser_info[36], double mono_info[4], double prior_w[2])
{
    std::memset(laser_info, 0, 36 * sizeof(double));
    for (int k = 0; k < 6; k++) laser_info[k * 7] = p.LASER_W * p.FACTOR_WEIGHT;
    mono_info[0] = mono_info[3] = p.FACTOR_WEIGHT; mono_info[1] = mono_info[2] = 0;
    prior_w[0] = p.PRIOR_T; prior_w[1] = p.PRIOR_R;
}
bool Estimator::optimization()
{
    SolvePack sp;
    packSolve(sp);
    const int feat_off[2] = { 0, sp.F }, obs_off[2] = { 0, (int)sp.obs_feat.size() };
    double laser_info[36], mono_info[4], prior_w[2];
    solve_infos(p_, laser_info, mono_info, prior_w);
    lmono_ba_desc d{};
    d.n_windows = 1; d.feat_off = feat_off; d.obs_off = obs_off; d.flags = sp.flags; d.poses = sp.poses; d.ex = para_ex[0];
    d.inv_depth = para_depth_inv.data(); d.obs_feat = sp.obs_feat.data(); d.obs_i = sp.obs_i.data(); d.obs_j = sp.obs_j.data(); d.obs_pts = sp.obs_pts.data();
    d.laser_consts = sp.laser; d.prior_T = TLC; d.laser_info = laser_info; d.mono_info = mono_info; d.prior_w = prior_w;
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
    double summary[6], poses[77], ex[7];
    std::vector<double> invd((size_t)std::max(sp.F, 1));
    hip_.check(lmono_ba_batch_read(hip_.get(), b, poses, ex, invd.data(), summary), "lmono_ba_batch_read");
    g_clock.lap(3);
    unpackSolve(sp, poses, ex, invd.data(), summary);
    if (frame_count < WINDOW_SIZE) return false;
    if (p_.ESTIMATE_LASER) margin();                         // Estimator.cc:1288-1291
    g_clock.lap(4);
    return termination == 0 || final_cost < 5e-3;            // Estimator.cc:1293
}
bool Estimator::packTracks(TrackPack &tp)
{
    feature_manager.pack(tp.start, tp.off, tp.pts, tp.depth, true);      // tracks below TRACK_CNT are skipped inside the kernels
    std::memset(tp.R, 0, sizeof(tp.R)); std::memset(tp.P, 0, sizeof(tp.P));
    for (int i = 0; i <= WINDOW_SIZE; i++) { std::memcpy(&tp.R[9 * i], Rs[i].m, 72); std::memcpy(&tp.P[3 * i], Ps[i].v, 24); }
    return !tp.start.empty();
}
void Estimator::applyOutlierScores(const double *score, double error, std::set<int> &removeIndex)
{
    size_t k = 0;
    for (auto &it : feature_manager.feature) { if (score[k] >= 0 && score[k] > error) removeIndex.insert(it.feature_id); k++; }
}
void Estimator::outliersRejection(std::set<int> &removeIndex, const double &error)
{
    TrackPack tp;
    if (!packTracks(tp)) return;
    const int feat_off[2] = { 0, (int)tp.start.size() };
    std::vector<double> score(tp.start.size());
    hip_.check(lmono_outlier_scores(hip_.get(), 1, feat_off, tp.R, tp.P, TLC, tp.start.data(), tp.off.data(), tp.pts.data(), tp.depth.data(),
                                    p_.TRACK_CNT, p_.FACTOR_WEIGHT, score.data()), "lmono_outlier_scores");
    applyOutlierScores(score.data(), error, removeIndex);
}
void Estimator::packMargin(MargPack &mp)
{
    MarginalizationInfo &mi = last_marginalization_info;
    mp = MargPack();
    if (marginalization_flag != MARGIN_OLD) {
        // :1406-1470: the previous prior, as the only factor, loses the block that aliases para_pose[WINDOW_SIZE - 1]
        if (!mi.present) return;
        const auto it = std::find(mi.parameter_blocks.begin(), mi.parameter_blocks.end(), WINDOW_SIZE - 1);
        if (it == mi.parameter_blocks.end()) return;
        mp.kind = 2;
        mp.drop = (int)(it - mi.parameter_blocks.begin()); mp.nb = (int)mi.parameter_blocks.size();
        matrix2Double();
        mp.x.resize((size_t)mp.nb * 7);
        for (int k = 0; k < mp.nb; k++) std::memcpy(&mp.x[7 * (size_t)k], mi.parameter_blocks[k] < 0 ? para_ex[0] : para_pose[mi.parameter_blocks[k]], 56);
        margin_calls[1]++;
        return;
    }
    matrix2Double();
    mp.kind = 1;
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
            mp.obs_feat.push_back(mp.f0); mp.obs_j.push_back(j);
            mp.obs_pts.insert(mp.obs_pts.end(), { it.feature_per_frame[0].pt[0], it.feature_per_frame[0].pt[1], f.pt[0], f.pt[1] });
        }
        mp.invd.push_back(para_depth_inv[feature_index]);
        mp.f0++;
    }
    for (int i = 0; i <= WINDOW_SIZE; i++) std::memcpy(mp.poses + 7 * i, para_pose[i], 56);
    std::memcpy(mp.ex, para_ex[0], 56);
    std::memcpy(mp.laser01, L0_R[0].m, 72); std::memcpy(mp.laser01 + 9, L0_R[1].m, 72);
    std::memcpy(mp.laser01 + 18, L0_T[0].v, 24); std::memcpy(mp.laser01 + 21, L0_T[1].v, 24);
    margin_calls[0]++;
}
void Estimator::applyMarginOld(const MargPack &mp, const double *lin_J, const double *lin_r, int status)
{
    MarginalizationInfo &mi = last_marginalization_info;
    mi.linearized_jacobians.assign(lin_J, lin_J + 66 * 66); mi.linearized_residuals.assign(lin_r, lin_r + 66);
    mi.status = status;
    mi.m = 6 + mp.f0; mi.n = 66;
    mi.keep_block_data.assign(77, 0.0);
    std::memcpy(mi.keep_block_data.data(), mp.ex, 56);
    for (int i = 1; i <= WINDOW_SIZE; i++) std::memcpy(mi.keep_block_data.data() + 7 * i, mp.poses + 7 * i, 56);
    // addr_shift :1390-1396: the kept blocks ex, pose1 .. pose10 alias para_ex[0], para_pose[0] .. para_pose[9] from now on
    mi.parameter_blocks.assign(1, -1);
    for (int i = 0; i < WINDOW_SIZE; i++) mi.parameter_blocks.push_back(i);
    mi.present = true;
    mi.valid = false;      // never set by the reference either
}
void Estimator::applyMarginSecond(MargPack &mp, const double *lin_J, const double *lin_r, int status)
{
    MarginalizationInfo &mi = last_marginalization_info;
    const int n = 6 * (mp.nb - 1);
    mi.linearized_jacobians.assign(lin_J, lin_J + (size_t)n * n); mi.linearized_residuals.assign(lin_r, lin_r + n);
    mp.x.erase(mp.x.begin() + 7 * (size_t)mp.drop, mp.x.begin() + 7 * (size_t)(mp.drop + 1));
    mi.keep_block_data.swap(mp.x);                             // parameter_block_data: the values at this marginalisation
    mi.parameter_blocks.erase(mi.parameter_blocks.begin() + mp.drop);   // addr_shift :1442-1455: pose i -> pose i (i < 9), pose 10 -> pose 9 is not a block
    mi.m = 6; mi.n = n; mi.status = status; mi.valid = false;
}
void Estimator::margin()
{
    marginWait();                                             // the previous prior is this call's input (and its storage is reused)
    HipContext *h = async_margin_ ? margin_hip_.get() : &hip_;
    auto mp = std::make_shared<MargPack>();
    packMargin(*mp);
    if (mp->kind == 0) return;
    double laser_info[36], mono_info[4], prior_w[2];
    solve_infos(p_, laser_info, mono_info, prior_w);
    std::array<double, 40> info;
    std::memcpy(info.data(), laser_info, 36 * sizeof(double)); std::memcpy(info.data() + 36, mono_info, 4 * sizeof(double));
    auto job = [this, h, mp, info]() {
        MarginalizationInfo &mi = last_marginalization_info;
        int status = 0;
        if (mp->kind == 2) {
            const int n = 6 * (mp->nb - 1);
            std::vector<double> J((size_t)n * n), r((size_t)n);
            h->check(lmono_marg_second_new(h->get(), 1, mp->nb, mp->drop, mi.linearized_jacobians.data(), mi.linearized_residuals.data(),
                                           mi.keep_block_data.data(), mp->x.data(), J.data(), r.data(), &status), "lmono_marg_second_new");
            applyMarginSecond(*mp, J.data(), r.data(), status);
            return;
        }
        const int feat_off[2] = { 0, mp->f0 }, obs_off[2] = { 0, (int)mp->obs_feat.size() };
        std::vector<double> J(66 * 66, 0.0), r(66, 0.0);
        const int dummy = 0; const double dzero = 0.0;
        h->check(lmono_marginalize(h->get(), 1, feat_off, obs_off, mp->poses, mp->ex, mp->invd.empty() ? &dzero : mp->invd.data(),
                                   mp->obs_feat.empty() ? &dummy : mp->obs_feat.data(), mp->obs_j.empty() ? &dummy : mp->obs_j.data(),
                                   mp->obs_pts.empty() ? &dzero : mp->obs_pts.data(), mp->laser01, info.data(), info.data() + 36,
                                   J.data(), r.data(), &status), "lmono_marginalize");
        applyMarginOld(*mp, J.data(), r.data(), status);
    };
    marginSubmit(std::move(job));
}

bool Estimator::slideWindowBegin(ShiftPack &sp)
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
            else {
                feature_manager.shiftDepthPack(sp.pt, sp.dep);                             // removeBackShiftDepth(back_R0, back_P0, Rs[0], Ps[0]) with TLC
                std::memcpy(sp.frames, back_R0.m, 72); std::memcpy(sp.frames + 9, back_P0.v, 24); std::memcpy(sp.frames + 12, Rs[0].m, 72);
                std::memcpy(sp.frames + 21, Ps[0].v, 24); std::memcpy(sp.frames + 24, TLC, 128);
                return true;
            }
        }
    } else if (frame_count == WINDOW_SIZE) {
        Header[frame_count - 1] = Header[frame_count];
        Ps[frame_count - 1] = Ps[frame_count]; Rs[frame_count - 1] = Rs[frame_count];
        // all_image_frame.erase(all_image_frame.end() - 1) (:735) removes the LAST element -- the newest frame's LiDAR pose -- while slot
        // WINDOW_SIZE - 1 takes the newest Ps / Rs / Header.  Reproduced as written: L0_R / L0_T[WINDOW_SIZE - 1] keep the second-newest
        // frame's pose and slot WINDOW_SIZE is rewritten by the next frame.
        feature_manager.removeFront(frame_count);                                          // slideWindowNew
    }
    return false;
}
void Estimator::slideWindowFinish(const double *depth_out) { feature_manager.shiftDepthApply(depth_out); }
void Estimator::slideWindow()
{
    ShiftPack sp;
    if (!slideWindowBegin(sp)) return;
    std::vector<double> out(sp.dep.size());
    if (!sp.dep.empty())
        hip_.check(lmono_shift_depth(hip_.get(), sp.frames, sp.frames + 9, sp.frames + 12, sp.frames + 21, sp.frames + 24, (int)sp.dep.size(), sp.pt.data(),
                                     sp.dep.data(), out.data()), "lmono_shift_depth");
    slideWindowFinish(out.data());
}

// ---- frame loop -------------------------------------------------------------------------------------------------------
void Estimator::processCompactData(const double L0_Pos[16])
{
    const double d[3] = { L0_Pos[3] - last_laser_t.v[0], L0_Pos[7] - last_laser_t.v[1], L0_Pos[11] - last_laser_t.v[2] };
    static_status = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) < 0.1;             // :259-265
    last_laser_t.v[0] = L0_Pos[3]; last_laser_t.v[1] = L0_Pos[7]; last_laser_t.v[2] = L0_Pos[11];
}
void Estimator::initialPoses()
{
    double rlcT[9], tlc[3] = { TLC[3], TLC[7], TLC[11] };
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) rlcT[i * 3 + j] = TLC[j * 4 + i];
    for (int i = 0; i <= frame_count; i++) {
        mat_mul(rlcT, L0_R[i].m, Rs[i].m);
        const double dv[3] = { L0_T[i].v[0] - tlc[0], L0_T[i].v[1] - tlc[1], L0_T[i].v[2] - tlc[2] };
        mat_vec(rlcT, dv, Ps[i].v);
    }
    feature_manager.clearDepth();
}
bool Estimator::runInitialization()
{
    // :986-1012 (the structure-from-motion block above it is commented out in the reference)
    initialPoses();
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
void Estimator::preFrame(double header, const FeatureManager::Image &image, const double transform_to_init[16], bool *keyframe_out)
{
    processCompactData(transform_to_init);
    const bool keyframe = feature_manager.featureCheck(frame_count, image, header);       // :383-394
    marginalization_flag = keyframe ? MARGIN_OLD : MARGIN_SECOND_NEW;
    Header[frame_count] = header;
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) L0_R[frame_count].m[i * 3 + j] = transform_to_init[i * 4 + j]; L0_T[frame_count].v[i] = transform_to_init[i * 4 + 3]; }   // all_image_frame.push_back
    if (keyframe_out) *keyframe_out = keyframe;
}
void Estimator::pushOdometryRow()
{
    std::array<double, 8> row;
    row[0] = Header[WINDOW_SIZE];
    for (int k = 0; k < 3; k++) row[1 + k] = Ps[WINDOW_SIZE].v[k];
    R_to_q(Rs[WINDOW_SIZE].m, &row[4]);
    new_odometry.push_back(row);
}
bool Estimator::processImage(double header, const FeatureManager::Image &image, const double transform_to_init[16])
{
    g_clock.start();
    bool keyframe = false;
    preFrame(header, image, transform_to_init, &keyframe);
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
    if (stage_flag == INITED) pushOdometryRow();             // new_odometry.txt row, :634-645
    return keyframe;
}

// ---- EstimatorBatch: N Estimators in lock-step ---------------------------------------------------------------------------
// worker threads for the per-stream host halves (pack / unpack: list walks, a few hundred microseconds per stream and frame in all)
class HostPool {
public:
    explicit HostPool(int n_threads) : T_(std::max(1, n_threads)), cur_((size_t)std::max(1, n_threads))
    {
        for (int t = 1; t < T_; t++) th_.emplace_back([this, t] { work(t); });
    }
    ~HostPool()
    {
        { std::lock_guard<std::mutex> g(mu_); quit_ = true; gen_.fetch_add(1, std::memory_order_release); }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    // fn(i) for i in [0, n), the caller takes part; rethrows the first exception.  Every thread OWNS a fixed share of the items -- stream s is always walked
    // by the same thread, so its lists stay in that core's cache and its allocations in that thread's malloc arena (handing the items out first come, first
    // served, the per-stream passes of 256 streams ran no faster on 16 threads than on 4: every free was another thread's block) -- and a thread that has
    // finished its own share takes items from the others' (a worker that wakes up late does not hold the pass up).  A worker spins for a few tens of
    // microseconds for the next pass before it sleeps: a lock-step frame is eight passes a few hundred microseconds apart.
    void run(int n, const std::function<void(int)> &fn)
    {
        if (n <= 0) return;
        if (th_.empty() || n == 1) { for (int i = 0; i < n; i++) fn(i); return; }
        std::lock_guard<std::mutex> one_pass(run_mu_);          // (batches that share the pool from different threads take turns)
        {
            std::lock_guard<std::mutex> g(mu_);
            fn_ = &fn; n_ = n; err_ = nullptr;
            for (int t = 0; t < T_; t++) cur_[(size_t)t].v.store(lo(t, n), std::memory_order_relaxed);
            left_.store(n, std::memory_order_relaxed);
            gen_.fetch_add(1, std::memory_order_release);
        }
        cv_.notify_all();
        take(0, &fn, n);
        // the last items are still running on the workers: spin, they are microseconds long
        for (int spin = 0; left_.load(std::memory_order_acquire) != 0; spin++) {
            if (spin < 4096) cpu_relax();
            else std::this_thread::yield();
        }
        { std::lock_guard<std::mutex> g(mu_); fn_ = nullptr; }                   // no worker can join this pass any more ...
        while (active_.load(std::memory_order_acquire) != 0) cpu_relax();       // ... and the ones that did have left it (they hold a pointer to fn)
        std::lock_guard<std::mutex> g(mu_);
        if (err_) { std::exception_ptr e = err_; err_ = nullptr; std::rethrow_exception(e); }
    }
private:
    struct alignas(64) Cursor { std::atomic<int> v{ 0 }; };
    int lo(int t, int n) const { return (int)((long long)t * n / T_); }
    static void cpu_relax()
    {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#else
        std::this_thread::yield();
#endif
    }
    void take(int me, const std::function<void(int)> *fn, int n)
    {
        for (int k = 0; k < T_; k++) {                   // own share first, then the others'
            const int t = (me + k) % T_, hi = lo(t + 1, n);
            for (;;) {
                const int i = cur_[(size_t)t].v.fetch_add(1, std::memory_order_relaxed);
                if (i >= hi) break;
                try { (*fn)(i); } catch (...) { std::lock_guard<std::mutex> g(mu_); if (!err_) err_ = std::current_exception(); }
                left_.fetch_sub(1, std::memory_order_acq_rel);
            }
        }
    }
    void work(int me)
    {
        unsigned long seen = 0;
        for (;;) {
            // a short spin for the next pass, then sleep
            bool got = false;
            for (int spin = 0; spin < 8192; spin++) {
                if (gen_.load(std::memory_order_acquire) != seen) { got = true; break; }
                cpu_relax();
            }
            const std::function<void(int)> *fn; int n;
            {
                std::unique_lock<std::mutex> lk(mu_);
                if (!got) cv_.wait(lk, [&] { return gen_.load(std::memory_order_acquire) != seen; });
                seen = gen_.load(std::memory_order_acquire);
                if (quit_) return;
                fn = fn_; n = n_;
                if (fn) active_.fetch_add(1, std::memory_order_acq_rel);      // joined under the lock: run() does not return before this worker has left take()
            }
            // (a worker that wakes up late finds fn_ already cleared: the pass is over)
            if (fn) { take(me, fn, n); active_.fetch_sub(1, std::memory_order_acq_rel); }
        }
    }
    const int T_;
    std::vector<std::thread> th_;
    std::mutex mu_, run_mu_;
    std::condition_variable cv_;
    const std::function<void(int)> *fn_ = nullptr;
    int n_ = 0;
    std::vector<Cursor> cur_;
    std::atomic<int> left_{ 0 }, active_{ 0 };
    std::atomic<unsigned long> gen_{ 0 };
    bool quit_ = false;
    std::exception_ptr err_;
};

// one pool per thread count and process: several EstimatorBatches (estimator_seq groups=G, driven by one thread) share their workers instead of
// each keeping its own set spinning / sleeping beside the others'
static std::shared_ptr<HostPool> shared_host_pool(int n_threads)
{
    static std::mutex mu;
    static std::weak_ptr<HostPool> cur;
    static int cur_n = 0;
    std::lock_guard<std::mutex> g(mu);
    std::shared_ptr<HostPool> p = cur.lock();
    if (p && cur_n == n_threads) return p;
    p = std::make_shared<HostPool>(n_threads);
    cur = p; cur_n = n_threads;
    return p;
}

// LMONO_HOST_TIMING=1: wall time of the lock-step frame's phases (INITED frames), printed by ~EstimatorBatch
namespace {
struct BatchClock {
    static constexpr int kN = 12;
    double ms[kN] = { 0 };
    long frames = 0;
    const bool on = std::getenv("LMONO_HOST_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t0;
    void start() { if (on) t0 = std::chrono::steady_clock::now(); }
    void lap(int k) { if (on) { const auto t1 = std::chrono::steady_clock::now(); ms[k] += std::chrono::duration<double, std::milli>(t1 - t0).count(); t0 = t1; } }
    void print(int n_streams) const
    {
        if (!on || frames == 0) return;
        static const char *name[kN] = { "pre+pack tracks", "-", "triangulate call", "apply+pack solve", "concat solve", "ba update", "solve+read", "unpack+pack margin/tracks", "margin submit", "outliers call+apply+slide begin", "shift call+finish+rows", "-" };
        std::fprintf(stderr, "BATCHTIM %d streams, %ld lock-step frames:", n_streams, frames);
        for (int k = 0; k < kN; k++) std::fprintf(stderr, " %s %.3f", name[k], ms[k] / frames);
        std::fprintf(stderr, " (ms per lock-step frame)\n");
    }
} g_bclock;
}

// The lock-step frame is a chain of [per-stream host pass on the pool] -> [one batched C-ABI call]; what a call needs is packed by the pass in front of it
// and what it returns is applied by the pass behind it, so an INITED frame is five passes and four calls (plus the marginalisation, which is handed
// to its worker).  The packs and the concatenated call arrays live here from frame to frame (their vectors keep their capacity).
struct EstimatorBatch::Work {
    std::vector<TrackPack> tp;
    std::vector<SolvePack> sp;
    std::vector<ShiftPack> shp;
    std::vector<char> due;
    // concatenated arrays of the calls
    std::vector<int> foff, ooff, start, off, flag, flags, obs_feat, obs_i, obs_j, shoff;
    std::vector<double> R, P, tlc, pts, depth, score, poses, ex, invd, obs_pts, laser, prior_T, summary, frames, shpt, shdep, shout;
};

EstimatorBatch::EstimatorBatch(HipContext &hip, const Params &p, int n_streams, int host_threads) : hip_(hip), p_(p)
{
    if (n_streams < 1) throw std::invalid_argument("EstimatorBatch: n_streams must be >= 1");
    for (int s = 0; s < n_streams; s++) est_.emplace_back(new Estimator(hip, p));
    int nt = host_threads;
    if (nt <= 0) { if (const char *e = std::getenv("LMONO_HOST_THREADS")) nt = std::atoi(e); }
    if (nt <= 0) nt = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    pool_ = shared_host_pool(std::min(nt, n_streams));
    work_.reset(new Work());
}
EstimatorBatch::~EstimatorBatch()
{
    try { marginWait(); } catch (...) {}
    margin_worker_.reset();
    if (ba_batch_) lmono_ba_batch_destroy(ba_batch_);
    g_bclock.print(size());
}
void EstimatorBatch::marginWait() { if (margin_worker_) margin_worker_->wait(); }
void EstimatorBatch::setAsyncMargin(bool on)
{
    marginWait();
    if (on && !margin_hip_) { margin_hip_.reset(new HipContext(hip_.device())); margin_hip_->useOwnStream(); }
    async_margin_ = on;
}

namespace {
template <typename T> void fit(std::vector<T> &v, size_t n) { if (v.size() < n) v.resize(n); }
}

// the tracks of every stream (w.tp, packed by the pass before) as one set of windows: w.foff / w.ooff and the concatenated arrays
void EstimatorBatch::concatTracks()
{
    Work &w = *work_;
    const int N = size();
    w.foff.assign((size_t)N + 1, 0); w.ooff.assign((size_t)N + 1, 0);
    for (int s = 0; s < N; s++) { w.foff[(size_t)s + 1] = w.foff[(size_t)s] + (int)w.tp[(size_t)s].start.size(); w.ooff[(size_t)s + 1] = w.ooff[(size_t)s] + (int)w.tp[(size_t)s].pts.size() / 2; }
    const int TF = w.foff[(size_t)N], TO = w.ooff[(size_t)N];
    fit(w.R, (size_t)N * 99); fit(w.P, (size_t)N * 33); fit(w.tlc, (size_t)N * 16); fit(w.pts, (size_t)TO * 2 + 2); fit(w.depth, (size_t)TF + 1);
    fit(w.start, (size_t)TF + 1); fit(w.off, (size_t)TF + 1); fit(w.flag, (size_t)TF + 1); fit(w.score, (size_t)TF + 1);
    pool_->run(N, [&](int s) {
        const TrackPack &t = w.tp[(size_t)s];
        std::memcpy(&w.R[(size_t)s * 99], t.R, sizeof(t.R)); std::memcpy(&w.P[(size_t)s * 33], t.P, sizeof(t.P)); std::memcpy(&w.tlc[(size_t)s * 16], est_[(size_t)s]->TLC, 128);
        const int f0 = w.foff[(size_t)s], o0 = w.ooff[(size_t)s];
        for (size_t k = 0; k < t.start.size(); k++) { w.start[(size_t)f0 + k] = t.start[k]; w.off[(size_t)f0 + k] = o0 + t.off[k]; w.depth[(size_t)f0 + k] = t.depth[k]; }
        if (!t.pts.empty()) std::memcpy(&w.pts[(size_t)o0 * 2], t.pts.data(), t.pts.size() * sizeof(double));
    });
    w.off[(size_t)TF] = TO;
}
// FeatureManager::triangulate of every stream: one lmono_triangulate over N windows (results: w.depth, w.flag by w.foff)
void EstimatorBatch::callTriangulate()
{
    Work &w = *work_;
    concatTracks();
    const int N = size();
    if (w.foff[(size_t)N] == 0) return;
    std::fill(w.flag.begin(), w.flag.begin() + w.foff[(size_t)N], 0);
    hip_.check(lmono_triangulate(hip_.get(), N, w.foff.data(), w.R.data(), w.P.data(), w.tlc.data(), w.start.data(), w.off.data(), w.pts.data(), w.depth.data(), w.flag.data(),
                                 p_.TRACK_CNT, WINDOW_SIZE, p_.FACTOR_WEIGHT, 50), "lmono_triangulate");
}
void EstimatorBatch::applyTriangulate(int s)
{
    Work &w = *work_;
    if (w.foff[(size_t)s + 1] > w.foff[(size_t)s]) est_[(size_t)s]->feature_manager.triangulateApply(&w.depth[(size_t)w.foff[(size_t)s]], &w.flag[(size_t)w.foff[(size_t)s]]);
}
// Estimator::outliersRejection's statistic for every stream: one lmono_outlier_scores over N windows (results: w.score by w.foff)
void EstimatorBatch::callOutliers()
{
    Work &w = *work_;
    concatTracks();
    const int N = size();
    if (w.foff[(size_t)N] == 0) return;
    hip_.check(lmono_outlier_scores(hip_.get(), N, w.foff.data(), w.R.data(), w.P.data(), w.tlc.data(), w.start.data(), w.off.data(), w.pts.data(), w.depth.data(),
                                    p_.TRACK_CNT, p_.FACTOR_WEIGHT, w.score.data()), "lmono_outlier_scores");
}
void EstimatorBatch::applyOutliers(int s, double error)
{
    Work &w = *work_;
    if (w.foff[(size_t)s + 1] == w.foff[(size_t)s]) return;
    std::set<int> removeIndex;
    est_[(size_t)s]->applyOutlierScores(&w.score[(size_t)w.foff[(size_t)s]], error, removeIndex);
    est_[(size_t)s]->feature_manager.removeOutlier(removeIndex);
}
// Estimator::optimization's solve for every stream (w.sp, packed by the pass before): N windows in one lmono_ba_batch_update + lmono_ba_solve + lmono_ba_batch_read
void EstimatorBatch::callSolve()
{
    Work &w = *work_;
    const int N = size();
    w.foff.assign((size_t)N + 1, 0); w.ooff.assign((size_t)N + 1, 0);
    for (int s = 0; s < N; s++) { w.foff[(size_t)s + 1] = w.foff[(size_t)s] + w.sp[(size_t)s].F; w.ooff[(size_t)s + 1] = w.ooff[(size_t)s] + (int)w.sp[(size_t)s].obs_feat.size(); }
    const int TF = w.foff[(size_t)N], TO = w.ooff[(size_t)N];
    fit(w.flags, (size_t)N * 4); fit(w.obs_feat, (size_t)TO + 1); fit(w.obs_i, (size_t)TO + 1); fit(w.obs_j, (size_t)TO + 1);
    fit(w.poses, (size_t)N * 77); fit(w.ex, (size_t)N * 7); fit(w.invd, (size_t)TF + 1); fit(w.obs_pts, (size_t)TO * 4 + 4); fit(w.laser, (size_t)N * 240); fit(w.prior_T, (size_t)N * 16);
    fit(w.summary, (size_t)N * 6);
    pool_->run(N, [&](int s) {
        const SolvePack &q = w.sp[(size_t)s];
        Estimator &e = *est_[(size_t)s];
        std::memcpy(&w.flags[(size_t)s * 4], q.flags, sizeof(q.flags));
        std::memcpy(&w.poses[(size_t)s * 77], q.poses, sizeof(q.poses)); std::memcpy(&w.ex[(size_t)s * 7], e.para_ex[0], 56);
        std::memcpy(&w.laser[(size_t)s * 240], q.laser, sizeof(q.laser)); std::memcpy(&w.prior_T[(size_t)s * 16], e.TLC, 128);
        if (q.F > 0) std::memcpy(&w.invd[(size_t)w.foff[(size_t)s]], e.para_depth_inv.data(), (size_t)q.F * sizeof(double));
        const size_t o0 = (size_t)w.ooff[(size_t)s], no = q.obs_feat.size();
        if (no) {
            std::memcpy(&w.obs_feat[o0], q.obs_feat.data(), no * sizeof(int)); std::memcpy(&w.obs_i[o0], q.obs_i.data(), no * sizeof(int));
            std::memcpy(&w.obs_j[o0], q.obs_j.data(), no * sizeof(int)); std::memcpy(&w.obs_pts[o0 * 4], q.obs_pts.data(), no * 4 * sizeof(double));
        }
    });
    g_bclock.lap(4);
    double laser_info[36], mono_info[4], prior_w[2];
    solve_infos(p_, laser_info, mono_info, prior_w);
    lmono_ba_desc d{};
    d.n_windows = N; d.feat_off = w.foff.data(); d.obs_off = w.ooff.data(); d.flags = w.flags.data(); d.poses = w.poses.data(); d.ex = w.ex.data();
    d.inv_depth = w.invd.data(); d.obs_feat = w.obs_feat.data(); d.obs_i = w.obs_i.data(); d.obs_j = w.obs_j.data(); d.obs_pts = w.obs_pts.data();
    d.laser_consts = w.laser.data(); d.prior_T = w.prior_T.data(); d.laser_info = laser_info; d.mono_info = mono_info; d.prior_w = prior_w;
    if (!ba_batch_) {
        ba_batch_ = lmono_ba_batch_create(hip_.get(), &d);
        if (!ba_batch_) throw std::runtime_error(std::string("lmono_ba_batch_create: ") + lmono_last_error(hip_.get()));
    } else
        hip_.check(lmono_ba_batch_update(hip_.get(), ba_batch_, &d), "lmono_ba_batch_update");
    g_bclock.lap(5);
    hip_.check(lmono_ba_solve(hip_.get(), ba_batch_, p_.NUM_ITERATIONS), "lmono_ba_solve");       // asynchronous on the context stream
}
void EstimatorBatch::readSolve()
{
    Work &w = *work_;
    hip_.check(lmono_ba_batch_read(hip_.get(), ba_batch_, w.poses.data(), w.ex.data(), w.invd.data(), w.summary.data()), "lmono_ba_batch_read");
    g_bclock.lap(6);
}
void EstimatorBatch::applySolve(int s)
{
    Work &w = *work_;
    est_[(size_t)s]->unpackSolve(w.sp[(size_t)s], &w.poses[(size_t)s * 77], &w.ex[(size_t)s * 7], &w.invd[(size_t)w.foff[(size_t)s]], &w.summary[(size_t)s * 6]);
}

// Estimator::margin of every stream (packs filled by the pass before): the MARGIN_OLD streams in one lmono_marginalize, the MARGIN_SECOND_NEW ones in one
// lmono_marg_second_new -- on the marginalisation context and its worker thread when overlapped (the concatenation happens there too)
void EstimatorBatch::submitMargin(std::shared_ptr<std::vector<MargPack>> packs)
{
    const int N = size();
    HipContext *h = async_margin_ ? margin_hip_.get() : &hip_;
    auto job = [this, h, packs, N]() {
        std::vector<int> olds, seconds;
        for (int s = 0; s < N; s++) { const int k = (*packs)[(size_t)s].kind; if (k == 1) olds.push_back(s); else if (k == 2) seconds.push_back(s); }
        double laser_info[36], mono_info[4], prior_w[2];
        solve_infos(p_, laser_info, mono_info, prior_w);
        if (!olds.empty()) {
            const int W = (int)olds.size();
            std::vector<int> foff((size_t)W + 1, 0), ooff((size_t)W + 1, 0);
            for (int w = 0; w < W; w++) { const MargPack &m = (*packs)[(size_t)olds[(size_t)w]]; foff[(size_t)w + 1] = foff[(size_t)w] + m.f0; ooff[(size_t)w + 1] = ooff[(size_t)w] + (int)m.obs_feat.size(); }
            const int TF = foff[(size_t)W], TO = ooff[(size_t)W];
            std::vector<double> poses((size_t)W * 77), ex((size_t)W * 7), invd((size_t)TF + 1, 0.0), obs_pts((size_t)TO * 4 + 4, 0.0), laser01((size_t)W * 24);
            std::vector<int> obs_feat((size_t)TO + 1, 0), obs_j((size_t)TO + 1, 0), status((size_t)W, 0);
            for (int w = 0; w < W; w++) {
                const MargPack &m = (*packs)[(size_t)olds[(size_t)w]];
                std::memcpy(&poses[(size_t)w * 77], m.poses, sizeof(m.poses)); std::memcpy(&ex[(size_t)w * 7], m.ex, sizeof(m.ex)); std::memcpy(&laser01[(size_t)w * 24], m.laser01, sizeof(m.laser01));
                if (m.f0) std::memcpy(&invd[(size_t)foff[(size_t)w]], m.invd.data(), (size_t)m.f0 * sizeof(double));
                const size_t no = m.obs_feat.size(), o0 = (size_t)ooff[(size_t)w];
                if (no) { std::memcpy(&obs_feat[o0], m.obs_feat.data(), no * sizeof(int)); std::memcpy(&obs_j[o0], m.obs_j.data(), no * sizeof(int)); std::memcpy(&obs_pts[o0 * 4], m.obs_pts.data(), no * 4 * sizeof(double)); }
            }
            std::vector<double> J((size_t)W * 66 * 66, 0.0), r((size_t)W * 66, 0.0);
            h->check(lmono_marginalize(h->get(), W, foff.data(), ooff.data(), poses.data(), ex.data(), invd.data(), obs_feat.data(), obs_j.data(), obs_pts.data(),
                                       laser01.data(), laser_info, mono_info, J.data(), r.data(), status.data()), "lmono_marginalize");
            for (int w = 0; w < W; w++)
                est_[(size_t)olds[(size_t)w]]->applyMarginOld((*packs)[(size_t)olds[(size_t)w]], &J[(size_t)w * 66 * 66], &r[(size_t)w * 66], status[(size_t)w]);
        }
        if (!seconds.empty()) {
            // (a prior that still holds para_pose[WINDOW_SIZE - 1]'s block is the one a MARGIN_OLD pass built: 11 blocks, the dropped one last -- the same
            // shape for every stream; anything else is grouped by shape)
            std::map<std::pair<int, int>, std::vector<int>> by_shape;
            for (int s : seconds) by_shape[{ (*packs)[(size_t)s].nb, (*packs)[(size_t)s].drop }].push_back(s);
            for (auto &grp : by_shape) {
                const int nb = grp.first.first, drop = grp.first.second, W = (int)grp.second.size();
                const size_t n0 = 6 * (size_t)nb, n = n0 - 6;
                std::vector<double> J0((size_t)W * n0 * n0), r0((size_t)W * n0), x0((size_t)W * nb * 7), x((size_t)W * nb * 7), J((size_t)W * n * n), r((size_t)W * n);
                std::vector<int> status((size_t)W, 0);
                for (int w = 0; w < W; w++) {
                    const Estimator::MarginalizationInfo &mi = est_[(size_t)grp.second[(size_t)w]]->last_marginalization_info;
                    std::memcpy(&J0[(size_t)w * n0 * n0], mi.linearized_jacobians.data(), n0 * n0 * sizeof(double));
                    std::memcpy(&r0[(size_t)w * n0], mi.linearized_residuals.data(), n0 * sizeof(double));
                    std::memcpy(&x0[(size_t)w * nb * 7], mi.keep_block_data.data(), (size_t)nb * 7 * sizeof(double));
                    std::memcpy(&x[(size_t)w * nb * 7], (*packs)[(size_t)grp.second[(size_t)w]].x.data(), (size_t)nb * 7 * sizeof(double));
                }
                h->check(lmono_marg_second_new(h->get(), W, nb, drop, J0.data(), r0.data(), x0.data(), x.data(), J.data(), r.data(), status.data()), "lmono_marg_second_new");
                for (int w = 0; w < W; w++)
                    est_[(size_t)grp.second[(size_t)w]]->applyMarginSecond((*packs)[(size_t)grp.second[(size_t)w]], &J[(size_t)w * n * n], &r[(size_t)w * n], status[(size_t)w]);
            }
        }
    };
    if (!async_margin_) { job(); return; }
    if (!margin_worker_) margin_worker_.reset(new MarginWorker());
    margin_worker_->submit(std::move(job));
}

// the depth shifts of every stream whose slide is removeBackShiftDepth (w.due / w.shp, filled by the pass before): one lmono_shift_depth_batch
void EstimatorBatch::callShift()
{
    Work &w = *work_;
    const int N = size();
    w.shoff.assign((size_t)N + 1, 0);
    for (int s = 0; s < N; s++) w.shoff[(size_t)s + 1] = w.shoff[(size_t)s] + (w.due[(size_t)s] ? (int)w.shp[(size_t)s].dep.size() : 0);
    const int T = w.shoff[(size_t)N];
    fit(w.frames, (size_t)N * 40); fit(w.shpt, (size_t)T * 2 + 2); fit(w.shdep, (size_t)T + 1); fit(w.shout, (size_t)T + 1);
    for (int s = 0; s < N; s++) {
        if (!w.due[(size_t)s]) { std::memset(&w.frames[(size_t)s * 40], 0, 40 * sizeof(double)); continue; }
        std::memcpy(&w.frames[(size_t)s * 40], w.shp[(size_t)s].frames, sizeof(w.shp[(size_t)s].frames));
        const size_t n = w.shp[(size_t)s].dep.size();
        if (n) { std::memcpy(&w.shpt[(size_t)w.shoff[(size_t)s] * 2], w.shp[(size_t)s].pt.data(), n * 2 * sizeof(double)); std::memcpy(&w.shdep[(size_t)w.shoff[(size_t)s]], w.shp[(size_t)s].dep.data(), n * sizeof(double)); }
    }
    if (T > 0) hip_.check(lmono_shift_depth_batch(hip_.get(), N, w.frames.data(), w.shoff.data(), w.shpt.data(), w.shdep.data(), w.shout.data()), "lmono_shift_depth_batch");
}

// Estimator::processImage (Estimator.cc:367-499) for every stream, the numeric steps batched
void EstimatorBatch::processImage(const double *headers, const FeatureManager::Image *images, const double (*transform_to_init)[16], bool *keyframe)
{
    std::vector<const FeatureManager::Image *> ptr((size_t)size());
    for (int s = 0; s < size(); s++) ptr[(size_t)s] = images + s;
    processImage(headers, ptr.data(), transform_to_init, keyframe);
}
void EstimatorBatch::processImage(const double *headers, const FeatureManager::Image *const *images, const double (*transform_to_init)[16], bool *keyframe)
{
    processImageBegin(headers, images, transform_to_init, keyframe);
    processImageFinish();
}
// The frame in two halves around the window solve, which is the one long GPU step and is launched asynchronously: Begin runs everything up to the launch and
// returns; Finish waits for the solve and runs the rest.  A caller that drives several EstimatorBatches (each on its own context / stream) from ONE thread
// interleaves them -- finish(A), begin(A, next frame), finish(B), begin(B, next frame) ... -- so that one batch's host passes run under another's solve
// (estimator_seq groups=G); processImage = Begin + Finish.
void EstimatorBatch::processImageBegin(const double *headers, const FeatureManager::Image *const *images, const double (*transform_to_init)[16], bool *keyframe)
{
    if (pending_ != 0) throw std::logic_error("EstimatorBatch::processImageBegin: the previous frame was not finished");
    const int N = size();
    Work &w = *work_;
    if ((int)w.tp.size() != N) { w.tp.resize((size_t)N); w.sp.resize((size_t)N); w.shp.resize((size_t)N); w.due.assign((size_t)N, 0); }
    std::vector<char> kf((size_t)N, 0);
    g_bclock.start();
    const Estimator &e0 = *est_[0];
    for (int s = 1; s < N; s++)
        if (est_[(size_t)s]->stage_flag != e0.stage_flag || est_[(size_t)s]->frame_count != e0.frame_count)
            throw std::logic_error("EstimatorBatch: the streams are not at the same frame of their sequences");
    auto pre = [&](int s) { bool k = false; est_[(size_t)s]->preFrame(headers[s], *images[s], transform_to_init[s], &k); kf[(size_t)s] = k ? 1 : 0; };
    if (e0.stage_flag == Estimator::NOT_INITED) {
        if (e0.frame_count == WINDOW_SIZE && p_.ESTIMATE_LASER != 2) {
            // runInitialization :986-1012, then optimization
            pool_->run(N, [&](int s) { pre(s); est_[(size_t)s]->initialPoses(); est_[(size_t)s]->packTracks(w.tp[(size_t)s]); });
            callTriangulate();
            pool_->run(N, [&](int s) { applyTriangulate(s); est_[(size_t)s]->packTracks(w.tp[(size_t)s]); });
            callOutliers();
            pool_->run(N, [&](int s) { applyOutliers(s, 100.0); est_[(size_t)s]->packSolve(w.sp[(size_t)s]); });
            callSolve();
            pending_ = 2;
        } else {
            pool_->run(N, [&](int s) {
                pre(s);
                Estimator &e = *est_[(size_t)s];
                if (e.frame_count == WINDOW_SIZE) e.slideWindow();              // (ESTIMATE_LASER == 2: NOT_INITED slides are list surgery only, removeBack)
                if (e.frame_count < WINDOW_SIZE) {
                    e.frame_count++;
                    e.Ps[e.frame_count] = e.Ps[e.frame_count - 1]; e.Rs[e.frame_count] = e.Rs[e.frame_count - 1]; e.Header[e.frame_count] = e.Header[e.frame_count - 1];
                }
                // (a frame without a solve ends here: the caller's keyframe flag first, then the hook)
                if (keyframe) keyframe[s] = kf[(size_t)s] != 0;
                if (frame_hook_) frame_hook_(s, e);
            });
        }
    } else {
        pool_->run(N, [&](int s) { pre(s); est_[(size_t)s]->loopCorrection(); est_[(size_t)s]->packTracks(w.tp[(size_t)s]); });
        g_bclock.lap(0);
        callTriangulate();
        g_bclock.lap(2);
        pool_->run(N, [&](int s) { applyTriangulate(s); est_[(size_t)s]->packSolve(w.sp[(size_t)s]); });
        g_bclock.lap(3);
        callSolve();
        pending_ = 1;
    }
    if (keyframe) for (int s = 0; s < N; s++) keyframe[s] = kf[(size_t)s] != 0;
}
void EstimatorBatch::processImageFinish()
{
    if (pending_ == 0) return;
    const int N = size();
    Work &w = *work_;
    const bool init_frame = pending_ == 2;
    pending_ = 0;
    g_bclock.start();
    readSolve();
    // Estimator::optimization behind the solve: double2Matrix, then margin() (frame_count == WINDOW_SIZE here) -- and the tracks for the outlier scores
    std::shared_ptr<std::vector<MargPack>> packs;
    const bool do_margin = p_.ESTIMATE_LASER != 0;
    if (do_margin) { marginWait(); packs = std::make_shared<std::vector<MargPack>>((size_t)N); }
    pool_->run(N, [&](int s) {
        applySolve(s);
        if (do_margin) est_[(size_t)s]->packMargin((*packs)[(size_t)s]);
        est_[(size_t)s]->packTracks(w.tp[(size_t)s]);
    });
    g_bclock.lap(7);
    // (overlapped marginalisation: few streams hand the packs to the worker right here, the reference's place -- the job is short and is over before the next
    // frame's first call; from 64 streams on they go at the END of the frame.  Both placements were measured again with the owner-share pool, one box, pairs of
    // runs: 8 streams 3.92 / 3.94 ms here vs 4.08 / 4.06 at the end; 64 streams 4.62 / 4.66 vs 4.64 / 4.63; 256 streams 7.18 / 6.13 vs 6.52 / 6.25 -- and over
    // 2750 frames 64 streams wait 0.4 ms per frame for a job handed over here.  LMONO_BATCH_MARGIN_EARLY=1 forces this place.)
    static const bool force_early = std::getenv("LMONO_BATCH_MARGIN_EARLY") != nullptr;
    const bool early = force_early || N < 64;
    if (do_margin && (!async_margin_ || early)) { submitMargin(packs); packs.reset(); }
    g_bclock.lap(8);
    if (init_frame) for (auto &e : est_) e->stage_flag = Estimator::INITED;
    const double outlier_error = init_frame ? 3.0 : p_.OUTLIER_T;
    callOutliers();
    pool_->run(N, [&](int s) { applyOutliers(s, outlier_error); w.due[(size_t)s] = est_[(size_t)s]->slideWindowBegin(w.shp[(size_t)s]) ? 1 : 0; });
    g_bclock.lap(9);
    callShift();
    pool_->run(N, [&](int s) {
        if (w.due[(size_t)s]) est_[(size_t)s]->slideWindowFinish(&w.shout[(size_t)w.shoff[(size_t)s]]);
        est_[(size_t)s]->pushOdometryRow();
        if (frame_hook_) frame_hook_(s, *est_[(size_t)s]);
    });
    g_bclock.lap(10);
    if (packs && async_margin_) submitMargin(packs);
    if (!init_frame) g_bclock.frames++;
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
