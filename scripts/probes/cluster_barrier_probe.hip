// Cost of a barrier + 28-double exchange between K workgroups on ONE XCD (blocks 0, 8, 16, ... of the grid), three ways.  Stand-alone:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/cbp scripts/probes/cluster_barrier_probe.hip && /tmp/cbp
// V0: wave 0 releases at agent scope, lane 0 arrives / polls relaxed, acquire at agent scope, plain loads of the partials (k_map_solve's barrier).
// V1: no fences: the partials are written and read with relaxed agent-scope ATOMIC stores / loads (they bypass the L1), s_waitcnt before the arrival.
// V2: like V1 but plain stores for the partials (the L1 is write-through) and atomic loads to read them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int kRounds = 2000, kT = 512;

template <int V>
__global__ __launch_bounds__(kT) void k_probe(int K, double *part, unsigned int *ctr, double *out, int *bad)
{
    if (blockIdx.x % 8 != 0) return;
    const int rank = blockIdx.x / 8, tid = threadIdx.x;
    __shared__ double s_sum[28];
    double acc = 0.0;
    for (int r = 0; r < kRounds; r++) {
        if (tid < 64) {
            double *mine = part + ((size_t)(r & 1) * 8 + rank) * 32;          // two banks in turn: a round's partials are not overwritten while a neighbour still reads them
            const double v = (double)(r + 1) * (rank + 1) + tid;
            if (V == 0) {
                if (tid < 28) mine[tid] = v;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            } else if (V == 1) {
                if (tid < 28) __hip_atomic_store(mine + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_s_waitcnt(0);
            } else {
                if (tid < 28) mine[tid] = v;
                __builtin_amdgcn_s_waitcnt(0);
            }
            if (tid == 0) {
                __hip_atomic_fetch_add(ctr + r, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int spins = 0;
                while (__hip_atomic_load(ctr + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned int)K) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1 << 24)) { *bad = 1; break; }
                }
            }
            if (V == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            double t = 0.0;
            if (tid < 28) {
                for (int w = 0; w < K; w++) {
                    const double *p = part + ((size_t)(r & 1) * 8 + w) * 32 + tid;
                    t += V == 0 ? *p : __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                s_sum[tid] = t;
                const double want = (double)(r + 1) * (K * (K + 1) / 2) + (double)K * tid;
                if (t != want) *bad = 2;
            }
        }
        __syncthreads();
        acc += s_sum[tid % 28];
        __syncthreads();
    }
    if (rank == 0 && tid == 0) *out = acc;
}

template <int V> static void run(int K)
{
    double *part, *out; unsigned int *ctr; int *bad;
    hipMalloc(&part, 2 * 8 * 32 * sizeof(double)); hipMalloc(&out, 8); hipMalloc(&ctr, kRounds * 4); hipMalloc(&bad, 4);
    float best = 1e30f;
    int hb = 0;
    for (int rep = 0; rep < 4; rep++) {
        hipMemset(ctr, 0, kRounds * 4); hipMemset(bad, 0, 4); hipMemset(part, 0, 2 * 8 * 32 * sizeof(double));
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_probe<V>, dim3(8 * K), dim3(kT), 0, 0, K, part, ctr, out, bad);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
        int b; hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost); hb |= b;
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
    printf("variant %d, K = %d: %.2f us per round%s\n", V, K, best * 1e3f / kRounds, hb ? (hb & 2 ? "  WRONG SUMS" : "  TIMED OUT") : "");
    hipFree(part); hipFree(out); hipFree(ctr); hipFree(bad);
}

int main()
{
    for (int K : { 1, 2, 4, 8 }) { run<0>(K); run<1>(K); run<2>(K); }
    return 0;
}
