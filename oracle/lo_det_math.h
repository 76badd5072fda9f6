/*
 * oracle/lo_det_math.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Deterministic arctangent built only from IEEE-754 +,-,*,/ (all correctly rounded on x86-64 and on gfx950), so that
 * the CPU oracle and the HIP path produce bit-identical ring ids / azimuths.  The reference (A-LOAM scanRegistration,
 * source absent) calls libm atan/atan2 on float data; libm differs from this function by
 * a few ulp of *double*, i.e. it changes the float-rounded value with probability ~1e-8 per point.
 * Algorithm: |x| > 1 -> 1/|x|; nearest breakpoint c = k/16; t = (a - c) / (1 + a c), |t| <= 1/32; atan(a) =
 * atan(c) (table of correctly rounded constants) + odd Taylor series in t up to t^15.  Compile with -ffp-contract=off.
 */
#ifndef LO_DET_MATH_H
#define LO_DET_MATH_H
#include <math.h>

#define LO_PI 3.14159265358979323846
#define LO_PI_2 1.57079632679489661923

static const double lo_atan_tab[17] = {
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

static inline double lo_atan(double x)
{
    int neg = x < 0.0;
    double a = neg ? -x : x;
    int inv = a > 1.0;
    if (inv) a = 1.0 / a;
    int k = (int)(a * 16.0 + 0.5);
    double c = (double)k * 0.0625;
    double t = (a - c) / (1.0 + a * c);
    double z = t * t;
    double s = 1.0 / 15.0;
    s = 1.0 / 13.0 - z * s;
    s = 1.0 / 11.0 - z * s;
    s = 1.0 / 9.0 - z * s;
    s = 1.0 / 7.0 - z * s;
    s = 1.0 / 5.0 - z * s;
    s = 1.0 / 3.0 - z * s;
    s = 1.0 - z * s;
    double r = lo_atan_tab[k] + t * s;
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
