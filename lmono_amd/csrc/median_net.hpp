// median_net.hpp -- median of a 5 x 5 window as a fixed network of min / max / median-of-3 operations (the median stage of
// k_depth_fill, colour.hip).  ~150 three-input VALU operations per pixel instead of the ~430 of an 8-step bisection on the
// value range.  Correctness rests on the 0-1 principle (the network is built from monotone operations only):
// tests/test_median_net.py compiles this header for the host and checks all 2^25 binary inputs.
#pragma once

#if defined(__HIPCC__)
#define MN_FN __device__ __forceinline__
#else
#define MN_FN inline
#endif

namespace mednet {

MN_FN int mn2(int a, int b) { return a < b ? a : b; }
MN_FN int mx2(int a, int b) { return a > b ? a : b; }
MN_FN int mn3(int a, int b, int c) { return mn2(mn2(a, b), c); }
MN_FN int mx3(int a, int b, int c) { return mx2(mx2(a, b), c); }
MN_FN int md3(int a, int b, int c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
#else
    return mx2(mn2(a, b), mn2(mx2(a, b), c));
#endif
}
MN_FN void ce(int &a, int &b) { const int lo = mn2(a, b), hi = mx2(a, b); a = lo; b = hi; }
MN_FN void sort3(int &a, int &b, int &c) { const int lo = mn3(a, b, c), mi = md3(a, b, c), hi = mx3(a, b, c); a = lo; b = mi; c = hi; }
// ascending sort of 5: sort3 + sort2, the two extremes, and a sort3 of what is left (12 three-input operations)
MN_FN void sort5(int &a, int &b, int &c, int &d, int &e)
{
    sort3(a, b, c); ce(d, e);
    const int o0 = mn2(a, d), o4 = mx2(c, e);
    int x = mx2(a, d), y = b, z = mn2(c, e);
    sort3(x, y, z);
    a = o0; b = x; c = y; d = z; e = o4;
}

// median of the 25 values in[row * 5 + col]
MN_FN int median25(const int *in)
{
    int m[5][5];
#pragma unroll
    for (int r = 0; r < 5; r++)
#pragma unroll
        for (int c = 0; c < 5; c++) m[r][c] = in[r * 5 + c];
#pragma unroll
    for (int c = 0; c < 5; c++) sort5(m[0][c], m[1][c], m[2][c], m[3][c], m[4][c]);     // columns ascending down the rows
#pragma unroll
    for (int r = 0; r < 5; r++) sort5(m[r][0], m[r][1], m[r][2], m[r][3], m[r][4]);     // rows ascending (columns stay sorted)
    // entry (r, k) now has at least (r + 1)(k + 1) - 1 values below it and (5 - r)(5 - k) - 1 above it: six entries can only lie
    // below the median, six only above, and the median of the remaining 13 is the median of the 25
    int s[8] = { m[0][3], m[0][4], m[1][2], m[1][3], m[2][1], m[2][2], m[3][0], m[3][1] };   // four pairs already in order
    const int rest[5] = { m[1][4], m[2][3], m[3][2], m[4][0], m[4][1] };
    // forgetful selection of the 7th of 13: drop the minimum and the maximum of the working set, take the next value in
    ce(s[0], s[2]); ce(s[0], s[4]); ce(s[0], s[6]);          // s0 = min of 8
    ce(s[1], s[7]); ce(s[3], s[7]); ce(s[5], s[7]);          // s7 = max of 8
    int t[7] = { s[1], s[2], s[3], s[4], s[5], s[6], rest[0] };
    ce(t[0], t[1]); ce(t[2], t[3]); ce(t[4], t[5]);
    ce(t[0], t[2]); ce(t[0], t[4]); ce(t[0], t[6]);          // t0 = min of 7
    ce(t[1], t[6]); ce(t[3], t[6]); ce(t[5], t[6]);          // t6 = max of 7
    int u[6] = { t[1], t[2], t[3], t[4], t[5], rest[1] };
    ce(u[0], u[1]); ce(u[2], u[3]); ce(u[4], u[5]);
    ce(u[0], u[2]); ce(u[0], u[4]);
    ce(u[1], u[5]); ce(u[3], u[5]);
    int p[5] = { u[1], u[2], u[3], u[4], rest[2] };
    ce(p[0], p[1]); ce(p[2], p[3]);
    ce(p[0], p[2]); ce(p[0], p[4]);
    ce(p[1], p[4]); ce(p[3], p[4]);
    int q[4] = { p[1], p[2], p[3], rest[3] };
    ce(q[0], q[1]); ce(q[2], q[3]);
    ce(q[0], q[2]); ce(q[1], q[3]);
    return md3(q[1], q[2], rest[4]);
}

}  // namespace mednet
