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
constexpr int kStatusDenseCell = 16;     // a 1 m grid cell of a "last" cloud holds > 32767 points (generic search path used)
constexpr int kStatusIrregularLines = 4; // scan-line ids of a feature cloud too disordered for the windowed walk (array-order walk used)

#define LM_PI 3.14159265358979323846
#define LM_PI_2 1.57079632679489661923

// Deterministic arctangent from IEEE +,-,*,/ only (bit-identical to the CPU oracle's restatement; the file is compiled
// with -ffp-contract=off): |x| > 1 -> 1/|x|, nearest breakpoint c = k/16, t = (a - c) / (1 + a c) with |t| <= 1/32,
// atan(a) = atan(c) from a table of correctly rounded constants + odd Taylor series in t up to t^15 (two divisions).
__device__ __constant__ const double kAtanTab[17] = {
    0,
    0.06241880999595735,
    0.12435499454676144,
    0.18534794999569476,
    0.24497866312686414,
    0.30288486837497142,
    0.35877067027057225,
    0.41241044159738732,
    0.46364760900080609,
    0.51238946031073773,
    0.55859931534356244,
    0.60228734613496415,
    0.64350110879328437,
    0.68231655487474807,
    0.71882999962162453,
    0.75315128096219441,
    0.78539816339744828
};

// tab: kAtanTab, or a copy of it in LDS (a per-lane look-up in the constant table is a global load that is waited for on the spot)
__device__ __forceinline__ double det_atan(double x, const double *tab = kAtanTab)
{
    const bool neg = x < 0.0;
    double a = neg ? -x : x;
    const bool inv = a > 1.0;
    if (inv) a = 1.0 / a;
    const int k = (int)(a * 16.0 + 0.5);
    const double c = (double)k * 0.0625;
    const double t = (a - c) / (1.0 + a * c);
    const double z = t * t;
    double s = 1.0 / 15.0;
    s = 1.0 / 13.0 - z * s;
    s = 1.0 / 11.0 - z * s;
    s = 1.0 / 9.0 - z * s;
    s = 1.0 / 7.0 - z * s;
    s = 1.0 / 5.0 - z * s;
    s = 1.0 / 3.0 - z * s;
    s = 1.0 - z * s;
    double r = tab[k] + t * s;
    if (inv) r = LM_PI_2 - r;
    return neg ? -r : r;
}

__device__ __forceinline__ double det_atan2(double y, double x, const double *tab = kAtanTab)
{
    if (x > 0.0) return det_atan(y / x, tab);
    if (x < 0.0) {
        const double r = det_atan(y / x, tab);
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
// sum over the wave, valid in lane 63: row_shr 1 / 2 / 4 / 8 inside every 16-lane row, row_bcast 15 / 31 across the rows (VALU moves; the LDS-crossbar
// butterfly of wave_sum_d above is two ds_bpermute per stage: it cost k_marginalize's factor pass 160 k cycles per round of observations); a lane without a source adds 0
template <int kCtrl, int kRowMask>
__device__ __forceinline__ double dpp_mov_f64(double v)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned int lo = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)(unsigned int)u, kCtrl, kRowMask, 0xf, false);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)(unsigned int)(u >> 32), kCtrl, kRowMask, 0xf, false);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double wave_sum_d_lane63(double v)
{
    v += dpp_mov_f64<0x111, 0xf>(v);
    v += dpp_mov_f64<0x112, 0xf>(v);
    v += dpp_mov_f64<0x114, 0xf>(v);
    v += dpp_mov_f64<0x118, 0xf>(v);
    v += dpp_mov_f64<0x142, 0xa>(v);
    v += dpp_mov_f64<0x143, 0xc>(v);
    return v;
}
// Jacobi rotation that annihilates a_pq: t = sgn(d) 2 a_pq / (|d| + sqrt(d^2 + 4 a_pq^2)) with d = a_qq - a_pp (the textbook tangent with numerator and
// denominator multiplied by 2 |a_pq|: one square root and one reciprocal instead of a division, a square root and a division), c = 1 / sqrt(1 + t^2),
// s = t c.  v_rsq_f64 / v_rcp_f64 with Newton corrections: accurate to rounding, not correctly rounded -- the iteration converges to the same
// eigen-decomposition, and a round waits for 33 lanes to finish this chain.
__device__ __forceinline__ double mg_rsqrt(double d)
{
    double y = __builtin_amdgcn_rsq(d);
    const double e = __builtin_fma(-d * y, y, 1.0);
    return __builtin_fma(y * e, __builtin_fma(e, 0.375, 0.5), y);
}
__device__ __forceinline__ void jacobi_cs(double app, double aqq, double apq, double &c, double &s)
{
    const double d = aqq - app, a2 = 2.0 * apq;
    const double x = __builtin_fma(d, d, a2 * a2);
    const double h = x * mg_rsqrt(x);
    const double den = fabs(d) + h;
    double r = __builtin_amdgcn_rcp(den);
    r = r * __builtin_fma(-den, r, 2.0);
    r = r * __builtin_fma(-den, r, 2.0);
    const double t = (d >= 0 ? a2 : -a2) * r;
    c = mg_rsqrt(__builtin_fma(t, t, 1.0));
    s = t * c;
}

__device__ __forceinline__ void marg_corrector(double *r, double *J, int nc, const double rho1, const double rho2)
{
    // ResidualBlockInfo::Evaluate / ceres::Corrector on one 2-row Jacobian block with leading dimension nc (in place, r untouched)
    const double sq = r[0] * r[0] + r[1] * r[1];
    const double sr = sqrt(rho1);
    double alpha_sq = 0.0;
    if (!(sq == 0.0 || rho2 <= 0.0)) { const double Dd = 1.0 + 2.0 * sq * rho2 / rho1; alpha_sq = (1.0 - sqrt(Dd)) / sq; }
    for (int j = 0; j < nc; j++) {
        const double rj = r[0] * J[j] + r[1] * J[nc + j];
        J[j] = sr * (J[j] - alpha_sq * r[0] * rj);
        J[nc + j] = sr * (J[nc + j] - alpha_sq * r[1] * rj);
    }
}

// ---- DPP reductions with a wave-uniform result (no LDS crossbar, one VALU instruction per stage): row_shr 1/2/4/8 inside
// every 16-lane row, row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2-3; lane 63 then holds the reduction of the
// wave and lane 31 that of lanes 0..31.  max / min are idempotent, so lanes without a source simply keep their own value.
// lanes without a source (or outside kRowMask) receive `ident`, the identity of the reduction, which lets the compiler
// fold the move into the consuming v_min_u32 / v_max_u32 as a DPP operand
template <int kCtrl, int kRowMask>
__device__ __forceinline__ unsigned int dpp_mov_u32(unsigned int v, unsigned int ident)
{
    return (unsigned int)__builtin_amdgcn_update_dpp((int)ident, (int)v, kCtrl, kRowMask, 0xf, false);
}
__device__ __forceinline__ unsigned int wave_max_u32_uniform(unsigned int v)
{
    v = max(v, dpp_mov_u32<0x111, 0xf>(v, 0u));
    v = max(v, dpp_mov_u32<0x112, 0xf>(v, 0u));
    v = max(v, dpp_mov_u32<0x114, 0xf>(v, 0u));
    v = max(v, dpp_mov_u32<0x118, 0xf>(v, 0u));
    v = max(v, dpp_mov_u32<0x142, 0xa>(v, 0u));
    v = max(v, dpp_mov_u32<0x143, 0xc>(v, 0u));
    return (unsigned int)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned int wave_min_u32_uniform(unsigned int v)
{
    v = min(v, dpp_mov_u32<0x111, 0xf>(v, ~0u));
    v = min(v, dpp_mov_u32<0x112, 0xf>(v, ~0u));
    v = min(v, dpp_mov_u32<0x114, 0xf>(v, ~0u));
    v = min(v, dpp_mov_u32<0x118, 0xf>(v, ~0u));
    v = min(v, dpp_mov_u32<0x142, 0xa>(v, ~0u));
    v = min(v, dpp_mov_u32<0x143, 0xc>(v, ~0u));
    return (unsigned int)__builtin_amdgcn_readlane((int)v, 63);
}
// arg-max / arg-min of a (hi, lo) key in lexicographic order, two 32-bit reductions; identity: (0, 0) / (~0, ~0)
__device__ __forceinline__ unsigned long long wave_max_key_uniform(unsigned long long k)
{
    const unsigned int hi = (unsigned int)(k >> 32), lo = (unsigned int)k;
    const unsigned int mh = wave_max_u32_uniform(hi);
    const unsigned int ml = wave_max_u32_uniform(hi == mh ? lo : 0u);
    return ((unsigned long long)mh << 32) | ml;
}
__device__ __forceinline__ unsigned long long wave_min_key_uniform(unsigned long long k)
{
    const unsigned int hi = (unsigned int)(k >> 32), lo = (unsigned int)k;
    const unsigned int mh = wave_min_u32_uniform(hi);
    const unsigned int ml = wave_min_u32_uniform(hi == mh ? lo : ~0u);
    return ((unsigned long long)mh << 32) | ml;
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

// in-LDS bitonic sort of n (power of two) 64-bit keys by the 4 waves of a 256-thread block.  Each wave owns a
// contiguous quarter of the array; compare-exchange stages whose partner distance j stays inside a quarter need no
// workgroup barrier (LDS operations of one wave execute in order), only the few stages with j >= n/4 do.
__device__ __forceinline__ void bitonic_sort_u64_lds(unsigned long long *keys, int n)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int P = n >= 8 ? n / 4 : n;            // elements per wave (tiny arrays: wave 0 alone)
    const int nw = n >= 8 ? 4 : 1;
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= P) {
                __syncthreads();
                for (int q = threadIdx.x; q < n / 2; q += 256) {
                    const int i = 2 * j * (q / j) + (q % j), p = i + j;
                    const unsigned long long a = keys[i], b = keys[p];
                    const bool asc = (i & k) == 0;
                    if ((a > b) == asc) { keys[i] = b; keys[p] = a; }
                }
                __syncthreads();
            } else if (wave < nw) {
                const int base = wave * P;
                for (int q = lane; q < P / 2; q += 64) {
                    const int i = base + 2 * j * (q / j) + (q % j), p = i + j;
                    const unsigned long long a = keys[i], b = keys[p];
                    const bool asc = (i & k) == 0;
                    if ((a > b) == asc) { keys[i] = b; keys[p] = a; }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    __syncthreads();
}

// value of lane (l ^ J), J = 1, 2, 4, 8, 16, 32.  Partner distances 1, 2 and 8 are one DPP move per dword (quad_perm,
// row_ror:8), 4 is two (row_half_mirror then the quad reversed); 16 and 32 leave the row and go through ds_bpermute.
template <int kCtrl>
__device__ __forceinline__ unsigned long long dpp_perm_u64(unsigned long long v)
{
    const int l = (int)(unsigned int)v, h = (int)(unsigned int)(v >> 32);
    const unsigned int lo = (unsigned int)__builtin_amdgcn_update_dpp(l, l, kCtrl, 0xf, 0xf, false);   // every lane has a source
    const unsigned int hi = (unsigned int)__builtin_amdgcn_update_dpp(h, h, kCtrl, 0xf, 0xf, false);
    return ((unsigned long long)hi << 32) | lo;
}
template <int J>
__device__ __forceinline__ unsigned long long lane_xor_u64(unsigned long long v)
{
    if constexpr (J == 1) return dpp_perm_u64<0xB1>(v);                          // quad_perm [1,0,3,2]
    else if constexpr (J == 2) return dpp_perm_u64<0x4E>(v);                     // quad_perm [2,3,0,1]
    else if constexpr (J == 4) return dpp_perm_u64<0x1B>(dpp_perm_u64<0x141>(v));  // row_half_mirror (i -> 7 - i), then quad_perm [3,2,1,0]
    else if constexpr (J == 8) return dpp_perm_u64<0x128>(v);                    // row_ror:8
    else return __shfl_xor(v, J);
}
// one compare-exchange stage of the register bitonic sort with partner distance J < 64 (lane exchange)
template <int R, int J>
__device__ __forceinline__ void bitonic_lane_stage(unsigned long long (&v)[R], int base, int lane, int k)
{
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int i = base + 64 * r + lane;
        const unsigned long long o = lane_xor_u64<J>(v[r]);
        const bool keep_min = ((lane & J) == 0) == ((i & k) == 0);
        // keep the smaller of (own, partner) on the low side: one 64-bit compare decides (equal keys: either)
        v[r] = ((o < v[r]) == keep_min) ? o : v[r];
    }
}

// Register-resident variant for n = 256 R keys (R = 1, 2, 4, 8): every wave keeps its quarter of the array in registers
// (lane l holds elements base + 64 r + l).  Of the log2(n) (log2(n) + 1) / 2 compare-exchange stages only three have a
// partner in another wave (through LDS, with barriers); partner distances 64..n/8 are register swaps inside a lane and
// distances < 64 are lane exchanges (ds_bpermute) -- no LDS addressing, no bank conflicts, no barriers.
template <int R>
__device__ __forceinline__ void bitonic_sort_u64_reg(unsigned long long *keys)
{
    constexpr int P = 64 * R, n = 4 * P;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, base = wave * P;
    unsigned long long v[R];
#pragma unroll
    for (int r = 0; r < R; r++) v[r] = keys[base + 64 * r + lane];
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= P) {
                __syncthreads();
#pragma unroll
                for (int r = 0; r < R; r++) keys[base + 64 * r + lane] = v[r];
                __syncthreads();
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int i = base + 64 * r + lane;
                    const unsigned long long o = keys[i ^ j];
                    const bool keep_min = ((i & j) == 0) == ((i & k) == 0);
                    v[r] = ((o < v[r]) == keep_min) ? o : v[r];
                }
            } else if (j >= 64) {
#pragma unroll
                for (int dr = R / 2; dr >= 1; dr >>= 1) {
                    if (j != 64 * dr) continue;
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        if (r & dr) continue;
                        const bool asc = ((base + 64 * r + lane) & k) == 0;
                        const unsigned long long a = v[r], b = v[r | dr];
                        const bool sw = (a > b) == asc;
                        v[r] = sw ? b : a; v[r | dr] = sw ? a : b;
                    }
                }
            } else {
                switch (j) {
                case 1: bitonic_lane_stage<R, 1>(v, base, lane, k); break;
                case 2: bitonic_lane_stage<R, 2>(v, base, lane, k); break;
                case 4: bitonic_lane_stage<R, 4>(v, base, lane, k); break;
                case 8: bitonic_lane_stage<R, 8>(v, base, lane, k); break;
                case 16: bitonic_lane_stage<R, 16>(v, base, lane, k); break;
                default: bitonic_lane_stage<R, 32>(v, base, lane, k); break;
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; r++) keys[base + 64 * r + lane] = v[r];
    __syncthreads();
}

// sort n (power of two, <= 2048) 64-bit keys in LDS with the 256 threads of the block
__device__ __forceinline__ void bitonic_sort_u64(unsigned long long *keys, int n)
{
    switch (n) {
    case 256: bitonic_sort_u64_reg<1>(keys); break;
    case 512: bitonic_sort_u64_reg<2>(keys); break;
    case 1024: bitonic_sort_u64_reg<4>(keys); break;
    case 2048: bitonic_sort_u64_reg<8>(keys); break;
    default: bitonic_sort_u64_lds(keys, n); break;
    }
}

__device__ __forceinline__ int next_pow2(int v)
{
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

} // namespace lmono
