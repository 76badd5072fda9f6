// lmono_amd/csrc/host_workers.hpp -- worker threads of the library for host work that is per-stream over many streams.  Host code only (no HIP): included
// by lmono_hip.hip and, on its own, by the ThreadSanitizer test (tests/test_host_workers_cpu.py).
#pragma once
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

// Worker threads for the host halves that are per-stream work over many streams (lmono_mapper_process_batch's update plan): fn(item, thread) for
// item in [0, n).  Every thread owns a contiguous share of the items and takes from the others' when it is done; the caller takes part.
class HostWorkers {
public:
    explicit HostWorkers(int n_threads) : T_(n_threads < 1 ? 1 : n_threads), cur_((size_t)(n_threads < 1 ? 1 : n_threads))
    {
        for (int t = 1; t < T_; t++) th_.emplace_back([this, t] { work(t); });
    }
    ~HostWorkers()
    {
        { std::lock_guard<std::mutex> g(mu_); quit_ = true; gen_.fetch_add(1, std::memory_order_release); }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    int threads() const { return T_; }
    bool run(int n, const std::function<void(int, int)> &fn)
    {
        if (n <= 0) return true;
        if (th_.empty() || n == 1) {
            try { for (int i = 0; i < n; i++) fn(i, 0); } catch (...) { return false; }
            return true;
        }
        {
            std::lock_guard<std::mutex> g(mu_);
            fn_ = &fn; n_ = n; failed_.store(false, std::memory_order_relaxed);
            for (int t = 0; t < T_; t++) cur_[(size_t)t].v.store(lo(t, n), std::memory_order_relaxed);
            left_.store(n, std::memory_order_relaxed);
            gen_.fetch_add(1, std::memory_order_release);
        }
        cv_.notify_all();
        take(0, &fn, n);
        while (left_.load(std::memory_order_acquire) != 0) relax();
        { std::lock_guard<std::mutex> g(mu_); fn_ = nullptr; }                 // nobody joins this pass any more ...
        while (active_.load(std::memory_order_acquire) != 0) relax();         // ... and those who did have left it
        return !failed_.load(std::memory_order_relaxed);
    }
private:
    struct alignas(64) Cursor { std::atomic<int> v{ 0 }; };
    int lo(int t, int n) const { return (int)((long long)t * n / T_); }
    static void relax() { __builtin_ia32_pause(); }
    void take(int me, const std::function<void(int, int)> *fn, int n)
    {
        for (int k = 0; k < T_; k++) {
            const int t = (me + k) % T_, hi = lo(t + 1, n);
            for (;;) {
                const int i = cur_[(size_t)t].v.fetch_add(1, std::memory_order_relaxed);
                if (i >= hi) break;
                try { (*fn)(i, me); } catch (...) { failed_.store(true, std::memory_order_relaxed); }      // (an item reports its errors through its own result slot; what is
                                                                                                      // caught here is an allocation failure: run() returns false)
                left_.fetch_sub(1, std::memory_order_acq_rel);
            }
        }
    }
    void work(int me)
    {
        unsigned long seen = 0;
        for (;;) {
            bool got = false;
            for (int spin = 0; spin < 8192; spin++) { if (gen_.load(std::memory_order_acquire) != seen) { got = true; break; } relax(); }
            const std::function<void(int, int)> *fn; int n;
            {
                std::unique_lock<std::mutex> lk(mu_);
                if (!got) cv_.wait(lk, [&] { return gen_.load(std::memory_order_acquire) != seen; });
                seen = gen_.load(std::memory_order_acquire);
                if (quit_) return;
                fn = fn_; n = n_;
                if (fn) active_.fetch_add(1, std::memory_order_acq_rel);
            }
            if (fn) { take(me, fn, n); active_.fetch_sub(1, std::memory_order_acq_rel); }
        }
    }
    const int T_;
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_;
    const std::function<void(int, int)> *fn_ = nullptr;
    int n_ = 0;
    std::vector<Cursor> cur_;
    std::atomic<int> left_{ 0 }, active_{ 0 };
    std::atomic<bool> failed_{ false };
    std::atomic<unsigned long> gen_{ 0 };
    bool quit_ = false;
};
