// scripts/fill_microbench.hip -- stand-alone timing of k_depth_fill<5> (no torch, no oracle): B frames of 1241 x 376 with a
// sparse random splat image; prints the average launch time.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
// -I include -I lmono_amd/csrc scripts/fill_microbench.hip -o /tmp/fill_mb   (development aid for profiles/, not a test)
#include <hip/hip_runtime.h>
#include "../include/lmono_hip.h"
#include "colour.hip"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
using namespace lmono;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char **argv)
{
    const int B = argc > 1 ? std::atoi(argv[1]) : 64, iters = argc > 2 ? std::atoi(argv[2]) : 20, W = 1241, H = 376;
    std::vector<ColourJob> jobs((size_t)B);
    std::vector<unsigned int> key((size_t)W * H);
    unsigned int s = 12345u;
    for (auto &k : key) { s = s * 1664525u + 1013904223u; k = (s >> 8) % 100u < 4u ? ((s >> 16) % 95u + 3u) | 0x100u : 0u; }
    BilateralTab t{};
    for (int i = 0; i < 256; i++) t.color[i] = (float)std::exp(i * i * (-0.5 / 2.25));
    int n = 0;
    for (int i = -2; i <= 2; i++) for (int j = -2; j <= 2; j++) { const double r = std::sqrt((double)i * i + j * j); if (r > 2) continue; t.space[n] = (float)std::exp(r * r * -0.125); t.di[n] = i; t.dj[n] = j; n++; }
    BilateralTab *tab; CK(hipMalloc(&tab, sizeof t)); CK(hipMemcpy(tab, &t, sizeof t, hipMemcpyHostToDevice));
    for (int b = 0; b < B; b++) {
        ColourJob &j = jobs[(size_t)b];
        j = ColourJob{};
        j.cam.w = W; j.cam.h = H; j.cam.ksize = 5; j.cam.blur = argc > 3 ? std::atoi(argv[3]) : 0;
        for (int i = 0; i < 25; i++) j.cam.mask[i] = 1;
        j.cam.mask_rect = 1;
        CK(hipMalloc(&j.key, key.size() * 4)); CK(hipMemcpy(j.key, key.data(), key.size() * 4, hipMemcpyHostToDevice));
        CK(hipMalloc(&j.depth, key.size()));
    }
    ColourJob *dj; CK(hipMalloc(&dj, sizeof(ColourJob) * B)); CK(hipMemcpy(dj, jobs.data(), sizeof(ColourJob) * B, hipMemcpyHostToDevice));
    const dim3 grid((unsigned)(((W + kFillTW - 1) / kFillTW) * ((H + kFillTH - 1) / kFillTH)), (unsigned)B);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) k_depth_fill<5><<<grid, kColT>>>(dj, tab);
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; i++) k_depth_fill<5><<<grid, kColT>>>(dj, tab);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned char> d(key.size());
    CK(hipMemcpy(d.data(), jobs[0].depth, d.size(), hipMemcpyDeviceToHost));
    unsigned long long sum = 0; for (auto v : d) sum = sum * 31u + v;
    std::printf("k_depth_fill<5> B=%d: %.1f us / launch, %.2f us / frame, checksum %llx\n", B, ms * 1e3 / iters, ms * 1e3 / iters / B, sum);
    return 0;
}
