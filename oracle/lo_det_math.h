/*
 * oracle/lo_det_math.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Deterministic arctangent built only from IEEE-754 +,-,*,/ and sqrt (all
 * correctly rounded on x86-64 and on gfx950), so that the CPU oracle and the
 * HIP path produce bit-identical ring ids / azimuths.  The reference
 * (A-LOAM scanRegistration, source absent) calls
 * libm atan/atan2 on float data; libm differs from this function by a few
 * ulp of *double*, i.e. it changes the float-rounded value with probability
 * ~1e-8 per point.  Compile with -ffp-contract=off.
 */
#ifndef LO_DET_MATH_H
#define LO_DET_MATH_H
#include <math.h>

#define LO_PI 3.14159265358979323846
#define LO_PI_2 1.57079632679489661923

static inline double lo_atan(double x)
{
    int neg = x < 0.0;
    double a = neg ? -x : x;
    int inv = a > 1.0;
    if (inv) a = 1.0 / a;
    /* three half-angle reductions: atan(a) = 2 atan(a / (1 + sqrt(1 + a^2))) */
    a = a / (1.0 + sqrt(1.0 + a * a));
    a = a / (1.0 + sqrt(1.0 + a * a));
    a = a / (1.0 + sqrt(1.0 + a * a));
    /* |a| <= tan(pi/32) ~ 0.0985: odd Taylor series, 12 terms (a^25 term < 1e-26) */
    double z = a * a;
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
    if (inv) r = LO_PI_2 - r;
    return neg ? -r : r;
}

static inline double lo_atan2(double y, double x)
{
    if (x > 0.0) return lo_atan(y / x);
    if (x < 0.0) {
        double r = lo_atan(y / x);
        return (y >= 0.0) ? r + LO_PI : r - LO_PI;
    }
    if (y > 0.0) return LO_PI_2;
    if (y < 0.0) return -LO_PI_2;
    return 0.0;
}

#endif
