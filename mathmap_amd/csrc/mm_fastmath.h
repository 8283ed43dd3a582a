// sin / cos of a float argument, result rounded to float: what the reference's generated C
// computes as (float)sin((double)x) with glibc (opmacros: `sin` is the double function, the
// assignment to a float compvar rounds).  Every argument of a MathMap math op is a float
// promoted to double, so the double function only ever sees 2^32 distinct inputs; this
// evaluates them with a table-driven reduction in double that is ~4x shorter than OCML's
// general-purpose double sin, and tools/verify_fastmath.c checks the result against glibc
// for EVERY float below MMF_LIMIT in magnitude (same C source, host build with -mfma).
// Arguments at or above the limit, infinities and NaN go to the platform's double function.
//
// x = k*(pi/32) + y, |y| <= pi/64:   sin x = S[k] cos y + C[k] sin y,  cos x = C[k] cos y - S[k] sin y
// with k*(pi/32) removed in three exact steps (HI/MID carry 27 bits: k*HI, k*MID exact for
// |k| < 2^26), S/C = correctly rounded sin/cos(k pi/32) and Taylor polynomials in y.
//
// This header is C and C++ (host verifier and device prelude include the same text).
#ifndef MM_FASTMATH_H
#define MM_FASTMATH_H

#include "mm_fastmath_tables.h"

#ifndef MMF_FN
#define MMF_FN static inline
#endif
#ifndef MMF_CONST_TABLE
#define MMF_CONST_TABLE static const
#endif
#ifndef MMF_FMA
#define MMF_FMA(a, b, c) fma((a), (b), (c))
#endif
#ifndef MMF_RINT
#define MMF_RINT(a) rint((a))
#endif
#ifndef MMF_FABSF
#define MMF_FABSF(a) fabsf((a))
#define MMF_FABS(a) fabs((a))
#endif
#ifndef MMF_SIN_SLOW
#define MMF_SIN_SLOW(a) sin((a))
#define MMF_COS_SLOW(a) cos((a))
#endif

#define MMF_LIMIT 4194304.0f /* 2^22: |k| < 2^25.4 */

MMF_CONST_TABLE double mmf_sincos_table[128] = MMF_SINCOS_TABLE;

// Reduction and the two polynomials; returns k.  *sy = sin(y), *cm1 = cos(y) - 1.
MMF_FN int mmf_reduce(double xd, double *sy, double *cm1) {
    const double kd = MMF_RINT(xd * MMF_INV_PIO32);
    double y = MMF_FMA(kd, -MMF_PIO32_HI, xd);
    y = MMF_FMA(kd, -MMF_PIO32_MID, y);
    y = MMF_FMA(kd, -MMF_PIO32_LO, y);
    const double y2 = y * y;
    double ps = MMF_FMA(y2, MMF_S5, MMF_S4);
    ps = MMF_FMA(y2, ps, MMF_S3);
    ps = MMF_FMA(y2, ps, MMF_S2);
    ps = MMF_FMA(y2, ps, MMF_S1);
    *sy = MMF_FMA(y * y2, ps, y);
    double pc = MMF_FMA(y2, MMF_C5, MMF_C4);
    pc = MMF_FMA(y2, pc, MMF_C3);
    pc = MMF_FMA(y2, pc, MMF_C2);
    pc = MMF_FMA(y2, pc, MMF_C1);
    *cm1 = y2 * pc;
    return (int)kd;
}

MMF_FN float mmf_sin_f32(float x) {
    if (!(MMF_FABSF(x) < MMF_LIMIT)) return (float)MMF_SIN_SLOW((double)x);
    double sy, cm1;
    const int k = mmf_reduce((double)x, &sy, &cm1) & 63;
    const double s = mmf_sincos_table[2 * k], c = mmf_sincos_table[2 * k + 1];
    // s*cos(y) + c*sin(y) = s + (s*(cos y - 1) + c*sin y)
    const double r = s + MMF_FMA(s, cm1, c * sy);
    return x == 0.0f ? x : (float)r;        // keeps the sign of a zero argument
}

MMF_FN float mmf_cos_f32(float x) {
    if (!(MMF_FABSF(x) < MMF_LIMIT)) return (float)MMF_COS_SLOW((double)x);
    double sy, cm1;
    const int k = mmf_reduce((double)x, &sy, &cm1) & 63;
    const double s = mmf_sincos_table[2 * k], c = mmf_sincos_table[2 * k + 1];
    // c*cos(y) - s*sin(y) = c + (c*(cos y - 1) - s*sin y)
    const double r = c + MMF_FMA(c, cm1, -(s * sy));
    return (float)r;
}

// Both, in double, for the float-complex functions (which evaluate in double and round once):
// the same reduction, accurate to about half an ulp of double for any |xd| < MMF_LIMIT.
MMF_FN void mmf_sincos_d(double xd, double *sn, double *cs) {
    if (!(MMF_FABS(xd) < (double)MMF_LIMIT)) {
        *sn = MMF_SIN_SLOW(xd);
        *cs = MMF_COS_SLOW(xd);
        return;
    }
    double sy, cm1;
    const int k = mmf_reduce(xd, &sy, &cm1) & 63;
    const double s = mmf_sincos_table[2 * k], c = mmf_sincos_table[2 * k + 1];
    const double rs = s + MMF_FMA(s, cm1, c * sy);
    *sn = xd == 0.0 ? xd : rs;
    *cs = c + MMF_FMA(c, cm1, -(s * sy));
}

#endif  // MM_FASTMATH_H
