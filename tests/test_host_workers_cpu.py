"""The library's host worker threads (lmono_amd/csrc/host_workers.hpp: lmono_mapper_process_batch plans the streams' map updates on them) under
ThreadSanitizer: every item of every pass runs exactly once, on whatever thread; an item that throws fails the pass, not the process.  CPU only."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HARNESS = r'''
#include "host_workers.hpp"
#include <cstdio>
#include <stdexcept>
int main()
{
    int bad = 0;
    for (int T : { 1, 2, 3, 8 }) {
        HostWorkers w(T);
        std::vector<int> hits, who;
        for (int pass = 0; pass < 400; pass++) {
            const int n = pass % 7 == 0 ? 1 : (pass * 37) % 300 + 1;
            hits.assign((size_t)n, 0); who.assign((size_t)n, -1);
            const bool ok = w.run(n, [&](int i, int tid) { hits[(size_t)i] += 1; who[(size_t)i] = tid; });      // (an item owns its slots: no two threads take the same item)
            if (!ok) bad++;
            for (int i = 0; i < n; i++) if (hits[(size_t)i] != 1 || who[(size_t)i] < 0 || who[(size_t)i] >= T) bad++;
            if (pass % 50 == 0) std::this_thread::sleep_for(std::chrono::milliseconds(2));      // the workers fall asleep in between: the wake-up path
        }
        // an item that throws: the pass reports failure, every other item still ran, the pool is usable afterwards
        std::vector<int> ran(64, 0);
        const bool ok = w.run(64, [&](int i, int) { if (i == 13) throw std::runtime_error("x"); ran[(size_t)i] = 1; });
        if (ok && T > 0) bad++;
        int cnt = 0; for (int v : ran) cnt += v;
        if (T > 1 && cnt != 63) bad++;
        if (!w.run(10, [&](int, int) {})) bad++;
    }
    std::printf("bad %d\n", bad);
    return bad ? 1 : 0;
}
'''


def test_host_workers_under_thread_sanitizer(tmp_path):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    src = tmp_path / "hw.cpp"
    src.write_text(HARNESS)
    exe = str(tmp_path / "hw")
    b = subprocess.run(["g++", "-O1", "-g", "-fsanitize=thread", "-std=c++17", "-pthread", "-I" + os.path.join(ROOT, "lmono_amd", "csrc"), str(src), "-o", exe],
                       capture_output=True, text=True)
    if b.returncode != 0 and ("tsan" in b.stderr or "sanitize" in b.stderr):
        pytest.skip("no ThreadSanitizer runtime: " + b.stderr[-200:])
    assert b.returncode == 0, b.stderr[-2000:]
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66"))
    assert out.returncode == 0 and "ThreadSanitizer" not in out.stderr and "bad 0" in out.stdout, (out.stdout[-500:], out.stderr[-3000:])
