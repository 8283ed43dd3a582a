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

// sin and cos of the same argument (the polar -> cartesian conversion of every `ra` filter): one
// reduction, the same operations per result as the two functions above -- identical values.
typedef struct { float s, c; } mmf_sincos_t;
MMF_FN mmf_sincos_t mmf_sincos_f32(float x) {
    mmf_sincos_t o;
    if (!(MMF_FABSF(x) < MMF_LIMIT)) {
        o.s = (float)MMF_SIN_SLOW((double)x);
        o.c = (float)MMF_COS_SLOW((double)x);
        return o;
    }
    double sy, cm1;
    const int k = mmf_reduce((double)x, &sy, &cm1) & 63;
    const double s = mmf_sincos_table[2 * k], c = mmf_sincos_table[2 * k + 1];
    const double rs = s + MMF_FMA(s, cm1, c * sy);
    const double rc = c + MMF_FMA(c, cm1, -(s * sy));
    o.s = x == 0.0f ? x : (float)rs;
    o.c = (float)rc;
    return o;
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

// ---- exp and log of a float argument ---------------------------------------------------------
// (float)exp((double)x) and (float)log((double)x), verified like sin/cos against glibc for every
// float in the fast range (exp: |x| <= 700, the rest under/overflows even in double and goes to
// the platform function; log: every positive finite float).  Tables are double-double.
#ifndef MMF_LDEXP
#define MMF_LDEXP(a, e) ldexp((a), (e))
#define MMF_EXP_SLOW(a) exp((a))
#define MMF_LOG_SLOW(a) log((a))
#define MMF_POW_SLOW(a, b) pow((a), (b))
#endif

MMF_CONST_TABLE double mmf_exp_table[128] = MMF_EXP_TABLE;
MMF_CONST_TABLE double mmf_log_table[387] = MMF_LOG_TABLE;

// exp(xd) for |xd| <= 700: x = k ln2/64 + r, |r| <= ln2/128; 2^(k/64) from the table, exp(r) - 1
// by its Taylor polynomial (r^7/5040 < 2^-65 of the result).
MMF_FN double mmf_exp_d(double xd) {
    const double kd = MMF_RINT(xd * MMF_INV_LN2O64);
    const int k = (int)kd;
    double r = MMF_FMA(kd, -MMF_LN2O64_HI, xd);
    r = MMF_FMA(kd, -MMF_LN2O64_LO, r);
    double q = MMF_FMA(r, 1.0 / 720.0, 1.0 / 120.0);
    q = MMF_FMA(r, q, 1.0 / 24.0);
    q = MMF_FMA(r, q, 1.0 / 6.0);
    q = MMF_FMA(r, q, 0.5);
    const double p = MMF_FMA(r * r, q, r);                   // exp(r) - 1
    const double th = mmf_exp_table[2 * (k & 63)], tl = mmf_exp_table[2 * (k & 63) + 1];
    const double res = th + MMF_FMA(th, p, tl);
    return MMF_LDEXP(res, k >> 6);
}

MMF_FN float mmf_exp_f32(float x) {
    if (!(MMF_FABSF(x) <= 700.0f)) return (float)MMF_EXP_SLOW((double)x);
    return (float)mmf_exp_d((double)x);
}

// log(xd), xd a positive, finite, normal double: xd = 2^e m, m in [1, 2); i = round((m-1) 128),
// c = 1 + i/128 (i = 128 folds into the next binade, so x near 1 has c = 1 and r = x - 1
// exactly); log m = -log(invc) + log1p(m invc - 1) with invc = fl(1/c) and the table holding
// -log(invc) for that rounded value.
MMF_FN double mmf_log_d(double xd) {
    union { double d; unsigned long long u; } b;
    b.d = xd;
    int e = (int)((b.u >> 52) & 0x7ff) - 1023;
    b.u = (b.u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m = b.d;
    int i = (int)MMF_RINT((m - 1.0) * 128.0);
    if (i == 128) {
        m *= 0.5;
        e += 1;
        i = 0;
    }
    const double invc = mmf_log_table[3 * i], lch = mmf_log_table[3 * i + 1], lcl = mmf_log_table[3 * i + 2];
    const double r = MMF_FMA(m, invc, -1.0);
    // log1p(r) - r = r^2 (-1/2 + r (1/3 + r (-1/4 + r (1/5 + r (-1/6 + r/7)))))
    double q = MMF_FMA(r, 1.0 / 7.0, -1.0 / 6.0);
    q = MMF_FMA(r, q, 0.2);
    q = MMF_FMA(r, q, -0.25);
    q = MMF_FMA(r, q, 1.0 / 3.0);
    q = MMF_FMA(r, q, -0.5);
    const double ed = (double)e;
    const double s = ed * MMF_LN2_HI;                         // exact: 42 x 8 bits
    const double t = s + lch;
    const double err = (s - t) + lch;                         // Fast2Sum: |s| >= |lch| whenever e != 0
    const double lo = MMF_FMA(r * r, q, MMF_FMA(ed, MMF_LN2_LO, lcl) + err);
    return t + (lo + r);
}

MMF_FN float mmf_log_f32(float x) {
    // positive finite floats only (denormals included: they are normal doubles)
    if (!(x > 0.0f && x <= 3.40282346638528859812e38f)) return (float)MMF_LOG_SLOW((double)x);
    return (float)mmf_log_d((double)x);
}

// ---- pow of two float arguments --------------------------------------------------------------
// (float)pow((double)x, (double)y) = exp(y log x) with log x carried as a double-double: for a
// float x the reduced argument r = m*invc - 1 is exact (invc has 20 bits), the table and
// e*ln2 are double-double, so log x is good to about 2^-68; the product with y keeps its
// rounding residual and exp() takes the high part, the low part enters as a final (1 + pl).
// Two arguments cannot be enumerated: tools/verify_fastmath.c compares 4e9 random and
// structured pairs with glibc (profiles/r01_verify_fastmath.json records the count and the
// mismatches, 0 so far) -- OCML's pow differs from glibc's result in ~0.1 % of float roundings.
MMF_FN void mmf_log_dd(double xd, double *hi, double *lo) {
    union { double d; unsigned long long u; } b;
    b.d = xd;
    int e = (int)((b.u >> 52) & 0x7ff) - 1023;
    b.u = (b.u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m = b.d;
    int i = (int)MMF_RINT((m - 1.0) * 128.0);
    if (i == 128) {
        m *= 0.5;
        e += 1;
        i = 0;
    }
    const double invc = mmf_log_table[3 * i], lch = mmf_log_table[3 * i + 1], lcl = mmf_log_table[3 * i + 2];
    const double r = MMF_FMA(m, invc, -1.0);                 // exact for a float-valued m
    double q = MMF_FMA(r, 1.0 / 9.0, -0.125);
    q = MMF_FMA(r, q, 1.0 / 7.0);
    q = MMF_FMA(r, q, -1.0 / 6.0);
    q = MMF_FMA(r, q, 0.2);
    q = MMF_FMA(r, q, -0.25);
    q = MMF_FMA(r, q, 1.0 / 3.0);
    q = MMF_FMA(r, q, -0.5);
    const double ed = (double)e;
    const double s = ed * MMF_LN2_HI;                         // exact
    const double t1 = s + lch;
    const double e1 = (s - t1) + lch;                         // Fast2Sum (|s| >= |lch| or s == 0)
    const double t2 = t1 + r;                                 // TwoSum
    const double bb = t2 - t1;
    const double e2 = (t1 - (t2 - bb)) + (r - bb);
    // tail: everything below t2 (the polynomial part is up to 2^-9 of t2, so renormalise)
    const double tail = MMF_FMA(r * r, q, (e1 + e2) + MMF_FMA(ed, MMF_LN2_LO, lcl));
    const double h = t2 + tail;
    *hi = h;
    *lo = (t2 - h) + tail;                                    // Fast2Sum: |t2| >= |tail|
}

MMF_FN float mmf_pow_pos_f32(float x, float y);
MMF_FN float mmf_pow_f32(float x, float y) {
    // a negative base with an integer exponent is +-|x|^y (pow's definition), and |x|^y has to come out exact where it is
    // exactly representable -- (-154.5)^3 = -3687953.625 sits on a float rounding tie, the platform's pow is an ulp off
    // there (found by tools/libm_exceptions.py: 2094 such pairs in 4.3e9 samples)
    if (x < 0.0f && x >= -3.40282346638528859812e38f && MMF_FABSF(y) <= 3.40282346638528859812e38f && (double)y == MMF_RINT((double)y)) {
        const float m = mmf_pow_pos_f32(-x, y);
        const double half = (double)y * 0.5;
        return half == MMF_RINT(half) ? m : -m;            // even / odd exponent (every float >= 2^24 is even)
    }
    return mmf_pow_pos_f32(x, y);
}
MMF_FN float mmf_pow_pos_f32(float x, float y) {
    if (!(x > 0.0f && x <= 3.40282346638528859812e38f) || !(MMF_FABSF(y) <= 3.40282346638528859812e38f))
        return (float)MMF_POW_SLOW((double)x, (double)y);
    const double xd = (double)x, yd = (double)y;
    // Integer exponents whose power is exactly representable (x^2 always is: 48 bits) must come
    // out exact -- an exact result can sit on a float rounding tie (11^7 = 19487171 needs 25
    // bits), where being one double ulp off changes the float.  Square-and-multiply with an fma
    // residual per product proves exactness; anything inexact continues below.
    if (yd == MMF_RINT(yd) && MMF_FABS(yd) <= 64.0) {
        int n = (int)MMF_FABS(yd);
        double base = xd, acc = 1.0;
        int exact = 1;
        while (n) {
            if (n & 1) {
                const double p = acc * base;
                exact = exact && MMF_FMA(acc, base, -p) == 0.0;
                acc = p;
            }
            n >>= 1;
            if (n) {
                const double p = base * base;
                exact = exact && MMF_FMA(base, base, -p) == 0.0;
                base = p;
            }
        }
        if (exact && acc <= 1.7976931348623157e308 && acc >= 2.2250738585072014e-308) {
            if (yd >= 0.0) return (float)acc;
            const double inv = 1.0 / acc;          // exact iff acc is a power of two; else one rounding of an inexact value
            if (MMF_FMA(inv, acc, -1.0) == 0.0) return (float)inv;
        }
    }
    double lh, ll;
    mmf_log_dd(xd, &lh, &ll);
    const double ph = yd * lh;
    if (!(MMF_FABS(ph) <= 700.0)) return (float)MMF_POW_SLOW(xd, yd);
    const double pl = MMF_FMA(yd, lh, -ph) + yd * ll;
    const double res = mmf_exp_d(ph);
    const double r = MMF_FMA(res, pl, res);
    // A result within a few double ulps of a float rounding tie or of an exact float (low 29
    // mantissa bits near 0x10000000 or 0) is decided by the platform pow, which is exact on exact
    // cases; this happens about once in 2^26 calls.
    union { double d; unsigned long long u; } b;
    b.d = r;
    const unsigned low = (unsigned)(b.u & 0x1fffffffULL);
    const unsigned d_tie = low > 0x10000000u ? low - 0x10000000u : 0x10000000u - low;
    if (d_tie <= 4u) return (float)MMF_POW_SLOW(xd, yd);
    return (float)r;
}

// ---- one-argument functions the platform computes: checked against glibc for every float, with an exception list ----
// (float)f((double)x) for tan, asin, acos, atan, sinh, cosh, tanh, asinh, acosh, atanh (and sin / cos beyond the
// table-driven range) is the device's double function (OCML) rounded to float.  OCML and glibc both carry an error of
// well under an ulp of *double*, so their float roundings can differ only where the double result lies next to a float
// rounding tie.  tools/libm_exceptions.py evaluates every float on the GPU and compares with the host's glibc
// (oracle/libm_ref.c): for ROCm 7.2's OCML against glibc 2.35 the two agree on all 2^32 arguments of every one of these
// functions except asinh (4 arguments) and acosh (2) -- mm_libm_exceptions.h lists those with glibc's float, and the
// two functions look an argument up when the platform result is within MMF_TIE_BAND double ulps of a tie (one call in
// 2^22).  tests/test_gpu_parity.py::test_unary_libm_equals_glibc_for_every_float repeats the enumeration: 0 differences.
typedef struct { unsigned x, r; } mmf_exc_t;
#include "mm_libm_exceptions.h"
// low 29 bits of a double's mantissa within the band around 0x10000000: its float rounding could go either way
#define MMF_TIE_BAND 64u
MMF_FN int mmf_near_float_tie(double r) {
    union { double d; unsigned long long u; } b;
    b.d = r;
    return (((unsigned)b.u + (MMF_TIE_BAND - 0x10000000u)) & 0x1fffffffu) <= 2u * MMF_TIE_BAND;
}
MMF_FN float mmf_exc_lookup(const mmf_exc_t *t, int n, float x, float fallback) {
    union { float f; unsigned u; } b;
    b.f = x;
    int lo = 0, hi = n - 1;
    while (lo <= hi) {                 /* sorted by argument bits */
        const int mid = (lo + hi) >> 1;
        if (t[mid].x == b.u) { b.u = t[mid].r; return b.f; }
        if (t[mid].x < b.u) lo = mid + 1; else hi = mid - 1;
    }
    return fallback;
}
#ifndef MMF_ASINH_SLOW
#define MMF_ASINH_SLOW(a) asinh((a))
#define MMF_ACOSH_SLOW(a) acosh((a))
#endif
#define MMF_UNARY_WITH_EXCEPTIONS(name, PLATFORM)                                                        \
    MMF_FN float mmf_##name##_f32(float x) {                                                            \
        const double r = PLATFORM((double)x);                                                           \
        if (mmf_near_float_tie(r)) return mmf_exc_lookup(mmf_exc_##name, MMF_EXC_N_##name, x, (float)r); \
        return (float)r;                                                                                \
    }
MMF_UNARY_WITH_EXCEPTIONS(asinh, MMF_ASINH_SLOW)
MMF_UNARY_WITH_EXCEPTIONS(acosh, MMF_ACOSH_SLOW)

// ---- hypot of two float arguments ---------------------------------------------------------------------
// (float)hypot((double)x, (double)y) as glibc 2.35 computes it (sysdeps/ieee754/dbl-64/e_hypot.c; the x86-64 build has no
// FMA variant: its kernel is h = sqrt(ax^2 + ay^2) followed by one correction step in plain double arithmetic).  For
// floats widened to double the squares are exact and neither overflow nor underflow, so none of its scaling branches
// is reachable; what remains is kernel(ax, ay) -- and ax + ay when ay <= 2^-54 ax.  The correction moves h by about an
// ulp of double and matters for the float only when h lies within a few ulps of a float rounding boundary (once in
// 2^26 calls): everywhere else (float)h is already the result, and the division is skipped.
#ifndef MMF_HYPOT_SLOW
#define MMF_HYPOT_SLOW(a, b) hypot((a), (b))
#endif
#ifndef MMF_SQRT
#define MMF_SQRT(a) sqrt((a))
#endif
MMF_FN float mmf_hypot_f32(float x, float y) {
    const double fx = (double)x, fy = (double)y;
    // ax^2 + ay^2 in either order is the same sum; with ay <= 2^-54 ax it is ax^2 and its root ax = ax + ay rounded:
    // glibc's early exit needs no test here
    double h = MMF_SQRT(fx * fx + fy * fy);
    union { double d; unsigned long long u; } b;
    b.d = h;
    const unsigned low = (unsigned)b.u & 0x1fffffffu;
    const unsigned d_tie = low > 0x10000000u ? low - 0x10000000u : 0x10000000u - low;
    if (d_tie <= 8u || !(h <= 1.7976931348623157e308)) {
        if (!(h <= 1.7976931348623157e308)) return (float)MMF_HYPOT_SLOW(fx, fy);      // inf, NaN
        const double gx = MMF_FABS(fx), gy = MMF_FABS(fy);
        const double ax = gx < gy ? gy : gx, ay = gx < gy ? gx : gy;
        if (ax >= ay * 0x1p54) return (float)(ax + ay);    // glibc: ax >= ay / EPS (a power of two: the same comparison)
        // the kernel's correction (no contraction: the reference's build has none either)
        double t1, t2;
        if (h <= 2.0 * ay) {
            const double delta = h - ay;
            t1 = ax * (2.0 * delta - ax);
            t2 = (delta - 2.0 * (ax - ay)) * delta;
        } else {
            const double delta = h - ax;
            t1 = 2.0 * delta * (ax - 2.0 * ay);
            t2 = (4.0 * delta - ay) * ay + delta * delta;
        }
        h -= (t1 + t2) / (2.0 * h);
    }
    return (float)h;
}

// Double-in, double-out variants with the range tests, for the float-complex functions.
MMF_FN double mmf_exp_any(double xd) { return (MMF_FABS(xd) <= 700.0) ? mmf_exp_d(xd) : MMF_EXP_SLOW(xd); }
MMF_FN double mmf_log_any(double xd) {
    return (xd >= 2.2250738585072014e-308 && xd <= 1.7976931348623157e308) ? mmf_log_d(xd) : MMF_LOG_SLOW(xd);
}

#endif  // MM_FASTMATH_H
