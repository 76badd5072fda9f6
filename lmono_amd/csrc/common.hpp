// lmono_amd/csrc/common.hpp -- shared device helpers for the gfx950 kernels (wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>
#include <float.h>
#include "../../include/lmono_hip.h"

namespace lmono {

constexpr int kWave = 64;
constexpr int kMaxRings = LMONO_MAX_RINGS;
constexpr int kSectors = 6;
constexpr int kRingCap = LMONO_RING_CAP;
constexpr int kMaxSharp = LMONO_MAX_SHARP;
constexpr int kMaxLessSharp = LMONO_MAX_LESS_SHARP;
constexpr int kMaxFlat = LMONO_MAX_FLAT;
constexpr int kMaxQueries = kMaxSharp + kMaxFlat;

// status bits per scan
constexpr int kStatusRingOverflow = 1;   // a ring holds more than kRingCap points
constexpr int kStatusGridOverflow = 2;   // a "last" cloud does not fit its hash grid
constexpr int kStatusWalkOverflow = 8;    // more than 256 features share one scan line (walk truncated)
constexpr int kStatusIrregularLines = 4; // scan-line ids of a feature cloud too disordered for the windowed walk (array-order walk used)

#define LM_PI 3.14159265358979323846
#define LM_PI_2 1.57079632679489661923

// Deterministic arctangent from IEEE +,-,*,/ and sqrt only (bit-identical to the CPU oracle's
// restatement; the file is compiled with -ffp-contract=off).  Three half-angle reductions and a
// 12-term odd Taylor series.
__device__ __forceinline__ double det_atan(double x)
{
    const bool neg = x < 0.0;
    double a = neg ? -x : x;
    const bool inv = a > 1.0;
    if (inv) a = 1.0 / a;
    a = a / (1.0 + sqrt(1.0 + a * a));
    a = a / (1.0 + sqrt(1.0 + a * a));
    a = a / (1.0 + sqrt(1.0 + a * a));
    const double z = a * a;
    double s = 1.0 / 23.0;
    s = 1.0 / 21.0 - z * s;
    s = 1.0 / 19.0 - z * s;
    s = 1.0 / 17.0 - z * s;
    s = 1.0 / 15.0 - z * s;
    s = 1.0 / 13.0 - z * s;
    s = 1.0 / 11.0 - z * s;
    s = 1.0 / 9.0 - z * s;
    s = 1.0 / 7.0 - z * s;
    s = 1.0 / 5.0 - z * s;
    s = 1.0 / 3.0 - z * s;
    s = 1.0 - z * s;
    double r = 8.0 * (a * s);
    if (inv) r = LM_PI_2 - r;
    return neg ? -r : r;
}

__device__ __forceinline__ double det_atan2(double y, double x)
{
    if (x > 0.0) return det_atan(y / x);
    if (x < 0.0) {
        const double r = det_atan(y / x);
        return (y >= 0.0) ? r + LM_PI : r - LM_PI;
    }
    if (y > 0.0) return LM_PI_2;
    if (y < 0.0) return -LM_PI_2;
    return 0.0;
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ int wave_min_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ int wave_max_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long w = __shfl_xor(v, o);
        v = w < v ? w : v;
    }
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// inclusive prefix sum across the wave
__device__ __forceinline__ int wave_scan_incl(int v)
{
    const int l = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int w = __shfl_up(v, o);
        if (l >= o) v += w;
    }
    return v;
}

// squared distance exactly as the reference evaluates it in float: (dx*dx + dy*dy) + dz*dz
__device__ __forceinline__ float dist2f(float ax, float ay, float az, float bx, float by, float bz)
{
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    return dx * dx + dy * dy + dz * dz;
}

// order-preserving pack of a non-negative float and a 32-bit payload: ascending u64 == ascending (f, payload)
__device__ __forceinline__ unsigned long long pack_fu(float f, unsigned int payload)
{
    return ((unsigned long long)__float_as_uint(f) << 32) | payload;
}

// in-LDS bitonic sort of n (power of two) 64-bit keys by all threads of the block
__device__ __forceinline__ void bitonic_sort_u64(unsigned long long *keys, int n)
{
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n; i += blockDim.x) {
                const int p = i ^ j;
                if (p > i) {
                    const unsigned long long a = keys[i], b = keys[p];
                    const bool asc = (i & k) == 0;
                    if ((a > b) == asc) { keys[i] = b; keys[p] = a; }
                }
            }
            __syncthreads();
        }
    }
}

__device__ __forceinline__ int next_pow2(int v)
{
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

} // namespace lmono
