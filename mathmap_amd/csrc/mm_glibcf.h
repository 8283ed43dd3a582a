// glibc 2.35's float libm and float-complex functions, restated.
//
// The reference's generated C calls glibc's `float _Complex` functions for every complex op
// (ops.lisp:194-213: csqrtf csinf ccosf ctanf casinf cacosf catanf cpowf cexpf clogf cargf csinhf
// ccoshf ctanhf casinhf cacoshf catanhf).  Those are 1-2 float ulps away from the exact value, so
// "computed in double and rounded once" is NOT what the reference produces; to return the
// reference's bits the device has to run glibc's own algorithms.  glibc is a third-party
// dependency that is not part of the reference tree: this file restates the published algorithms
// of glibc 2.35 (the version of the image, `ldd --version`; Ubuntu GLIBC 2.35-0ubuntu3.x):
//
//   sysdeps/ieee754/flt-32/e_expf.c, e_logf.c, s_sincosf.c (+ sincosf.h, sysdeps/x86/fpu/
//   sincosf_poly.h), e_exp2f_data.c, e_logf_data.c, s_sincosf_data.c      (ARM optimized routines)
//   sysdeps/ieee754/flt-32/s_atanf.c, e_atan2f.c, s_log1pf.c, s_expm1f.c, e_sinhf.c, e_coshf.c,
//   e_hypotf.c, x2y2m1f.c, s_scalbnf.c                                     (fdlibm, float)
//   math/s_cexp_template.c, s_clog_template.c, s_cpow_template.c, s_csqrt_template.c,
//   s_csin/ccos/ctan/csinh/ccosh/ctanh_template.c, k_casinh_template.c,
//   s_casin/cacos/casinh/cacosh/catan/catanh_template.c                    (complex templates)
//
// x86-64 selects the FMA builds of expf, logf and sincosf at load time (ifunc: every CPU with
// FMA + AVX2, i.e. this container and the GPU boxes' hosts); where those builds contract a
// multiply-add the same contraction is written out here with MMQ_FMA, everything else is
// evaluated operation by operation (the translation unit must be compiled with fp contraction
// off).  tools/verify_glibcf.c builds this same text for the host and compares every function with
// the host's libm: exhaustively over all 2^32 floats for the one-argument functions, on > 10^9
// structured and random pairs for the two-argument and complex ones.
//
// This header is C and C++ (host verifier, CPU tests and the device prelude include the same text).
#ifndef MM_GLIBCF_H
#define MM_GLIBCF_H

#ifndef MMQ_FN
#define MMQ_FN static inline
#endif
#ifndef MMQ_TABLE
#define MMQ_TABLE static const
#endif
#ifndef MMQ_FMA
#define MMQ_FMA(a, b, c) fma((a), (b), (c))
#endif
#ifndef MMQ_SQRT
#define MMQ_SQRT(a) sqrt((a))      /* double, correctly rounded */
#define MMQ_SQRTF(a) sqrtf((a))    /* float, correctly rounded */
#endif

#ifndef MMQ_UNLIKELY
#define MMQ_UNLIKELY(c) __builtin_expect(!!(c), 0)
#endif

// Two forms of the hot functions.  mmq_ref_* is the restatement, branch for branch as glibc has it.  On a GPU
// every `if` around a few instructions costs more than the instructions (exec-mask bookkeeping, hazard nops --
// scalar instructions take issue slots like vector ones), so the functions the complex ops run per pixel have a
// second form: the common case evaluated straight through with the SAME operations in the same order
// (alternatives picked by selects, ranges merged where the arithmetic coincides), everything unusual
// (NaN, infinities, zeros, huge / tiny arguments) behind one unlikely test that calls the reference form.
// Both forms go through tools/verify_glibcf.c, so "same bits" is checked, not argued.
typedef struct { float re, im; } mmq_cf;
typedef struct { float s, c; } mmq_sc;

MMQ_FN unsigned mmq_asuint(float f) { union { float f; unsigned u; } c; c.f = f; return c.u; }
MMQ_FN float mmq_asfloat(unsigned u) { union { float f; unsigned u; } c; c.u = u; return c.f; }
MMQ_FN unsigned long long mmq_asuint64(double f) { union { double f; unsigned long long u; } c; c.f = f; return c.u; }
MMQ_FN double mmq_asdouble(unsigned long long u) { union { double f; unsigned long long u; } c; c.u = u; return c.f; }
MMQ_FN float mmq_fabsf(float x) { return mmq_asfloat(mmq_asuint(x) & 0x7fffffffu); }
MMQ_FN float mmq_copysignf(float x, float s) { return mmq_asfloat((mmq_asuint(x) & 0x7fffffffu) | (mmq_asuint(s) & 0x80000000u)); }
MMQ_FN int mmq_isnanf(float x) { return (mmq_asuint(x) & 0x7fffffffu) > 0x7f800000u; }
MMQ_FN int mmq_isinff(float x) { return (mmq_asuint(x) & 0x7fffffffu) == 0x7f800000u; }
MMQ_FN int mmq_isfinitef(float x) { return (mmq_asuint(x) & 0x7fffffffu) < 0x7f800000u; }
MMQ_FN int mmq_issignalingf(float x) { return mmq_isnanf(x) && !(mmq_asuint(x) & 0x00400000u); }
MMQ_FN int mmq_signbitf(float x) { return (int)(mmq_asuint(x) >> 31); }
MMQ_FN float mmq_nanf(void) { return mmq_asfloat(0x7fc00000u); }
MMQ_FN float mmq_inff(void) { return mmq_asfloat(0x7f800000u); }
#define MMQ_FLT_MAX 3.40282346638528859812e+38f
#define MMQ_FLT_MIN 1.17549435082228750797e-38f
#define MMQ_FLT_EPSILON 1.1920928955078125e-07f

// ---- expf (e_expf.c, __expf_fma) ------------------------------------------------------------------
// tab[i] = bits(2^(i/32)) - (i << 47), e_exp2f_data.c
MMQ_TABLE unsigned long long mmq_exp2f_tab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull,
    0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull,
    0x3feedea64c123422ull, 0x3feece086061892dull, 0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull,
    0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull, 0x3feee89f995ad3adull,
    0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
    0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};

MMQ_FN float mmq_expf(float x) {
    const unsigned abstop = (mmq_asuint(x) >> 20) & 0x7ff;
    const double xd = (double)x;
    if (abstop >= 0x42b) {                       /* |x| >= 88 or x is nan */
        if (mmq_asuint(x) == 0xff800000u) return 0.0f;
        if (abstop >= 0x7f8) return x + x;
        if (x > 0x1.62e42ep6f) return mmq_inff();            /* overflow */
        if (x < -0x1.9fe368p6f) return 0.0f;                 /* underflow */
        if (x < -0x1.9d1d9ep6f) return 0x1p-149f;            /* __math_may_uflowf: 0x1.4p-75f squared */
    }
    /* x*N/Ln2 = k + r, r in [-1/2, 1/2], N = 32 */
    const double InvLn2N = 0x1.71547652b82fep+5, Shift = 0x1.8p+52;
    double kd = MMQ_FMA(InvLn2N, xd, Shift);
    const unsigned long long ki = mmq_asuint64(kd);
    kd -= Shift;
    const double r = MMQ_FMA(InvLn2N, xd, -kd);
    unsigned long long t = mmq_exp2f_tab[ki & 31];
    t += ki << 47;
    const double s = mmq_asdouble(t);
    const double z = MMQ_FMA(0x1.c6af84b912394p-20, r, 0x1.ebfce50fac4f3p-13);
    const double r2 = r * r;
    double y = MMQ_FMA(0x1.62e42ff0c52d6p-6, r, 1.0);
    y = MMQ_FMA(z, r2, y);
    y = y * s;
    return (float)y;
}

// ---- logf (e_logf.c, __logf_fma) -----------------------------------------------------------------
MMQ_TABLE double mmq_logf_tab[32] = {   /* {invc, logc} x 16, e_logf_data.c */
    0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2, 0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2,
    0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2, 0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3,
    0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3, 0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3,
    0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4, 0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4,
    0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5, 0x1p+0, 0x0p+0,
    0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5, 0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4,
    0x1.b2036576afce6p-1, 0x1.526e57720db08p-3, 0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3,
    0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2, 0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2};

MMQ_FN float mmq_ref_logf(float x) {
    unsigned ix = mmq_asuint(x);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {      /* x < 0x1p-126 or inf or nan */
        if (ix * 2 == 0) return -mmq_inff();
        if (ix == 0x7f800000u) return x;
        if ((ix & 0x80000000u) || ix * 2 >= 0xff000000u) return mmq_nanf();
        ix = mmq_asuint(x * 0x1p23f);                          /* subnormal: normalise */
        ix -= 23u << 23;
    }
    const unsigned tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) & 15);
    const int k = (int)tmp >> 23;
    const unsigned iz = ix - (tmp & 0xff800000u);
    const double invc = mmq_logf_tab[2 * i], logc = mmq_logf_tab[2 * i + 1];
    const double z = (double)mmq_asfloat(iz);
    const double r = MMQ_FMA(z, invc, -1.0);
    const double y0 = MMQ_FMA((double)k, 0x1.62e42fefa39efp-1, logc);
    const double r2 = r * r;
    double y = MMQ_FMA(0x1.5575b0be00b6ap-2, r, -0x1.ffffef20a4123p-2);
    y = MMQ_FMA(-0x1.00ea348b88334p-2, r2, y);
    y = MMQ_FMA(y, r2, y0 + r);
    return (float)y;
}

MMQ_FN float mmq_logf(float x) {
    const unsigned ix = mmq_asuint(x);
    // zero, negative, subnormal, inf, nan -> reference form.  (x == 1 needs no case of its own: its table entry is
    // invc = 1, logc = 0, which makes r, y0 and the result +0.)
    if (MMQ_UNLIKELY(ix - 0x00800000u >= 0x7f800000u - 0x00800000u)) return mmq_ref_logf(x);
    const unsigned tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) & 15);
    const int k = (int)tmp >> 23;
    const unsigned iz = ix - (tmp & 0xff800000u);
    const double invc = mmq_logf_tab[2 * i], logc = mmq_logf_tab[2 * i + 1];
    const double z = (double)mmq_asfloat(iz);
    const double r = MMQ_FMA(z, invc, -1.0);
    const double y0 = MMQ_FMA((double)k, 0x1.62e42fefa39efp-1, logc);
    const double r2 = r * r;
    double y = MMQ_FMA(0x1.5575b0be00b6ap-2, r, -0x1.ffffef20a4123p-2);
    y = MMQ_FMA(-0x1.00ea348b88334p-2, r2, y);
    y = MMQ_FMA(y, r2, y0 + r);
    return (float)y;
}

// ---- sincosf (s_sincosf.c, __sincosf_fma with the x86 vector polynomial) --------------------
MMQ_TABLE unsigned mmq_inv_pio4[24] = {
    0xa2, 0xa2f9, 0xa2f983, 0xa2f9836e, 0xf9836e4e, 0x836e4e44, 0x6e4e4415, 0x4e441529, 0x441529fc, 0x1529fc27,
    0x29fc2757, 0xfc2757d1, 0x2757d1f5, 0x57d1f534, 0xd1f534dd, 0xf534ddc0, 0x34ddc0db, 0xddc0db62, 0xc0db6295,
    0xdb629599, 0x6295993c, 0x95993c43, 0x993c4390, 0x3c439041};

// sin and cos of the reduced argument x (x2 = x*x of the UNSIGNED reduced argument), quadrant n; `neg`
// selects the negated cosine polynomial (table[1])
MMQ_FN mmq_sc mmq_sincosf_poly(double x, double x2, int neg, int n) {
    const double c0 = neg ? -1.0 : 1.0, c1 = neg ? 0x1.ffffffd0c621cp-2 : -0x1.ffffffd0c621cp-2;
    const double c2 = neg ? -0x1.55553e1068f19p-5 : 0x1.55553e1068f19p-5, c3 = neg ? 0x1.6c087e89a359dp-10 : -0x1.6c087e89a359dp-10;
    const double c4 = neg ? -0x1.99343027bf8c3p-16 : 0x1.99343027bf8c3p-16;
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    const double x3 = x2 * x, x4 = x2 * x2;
    const double s1c = MMQ_FMA(x2, s3, s2), c2c = MMQ_FMA(x2, c4, c3);
    const double c1c = MMQ_FMA(x2, c1, c0);
    const double x5 = x3 * x2, x6 = x4 * x2;
    double s = MMQ_FMA(x3, s1, x), c = MMQ_FMA(x4, c2, c1c);
    s = MMQ_FMA(x5, s1c, s);
    c = MMQ_FMA(x6, c2c, c);
    mmq_sc r;
    if (n & 1) { r.s = (float)c; r.c = (float)s; }
    else { r.s = (float)s; r.c = (float)c; }
    return r;
}

MMQ_FN mmq_sc mmq_ref_sincosf(float y) {
    const unsigned top = (mmq_asuint(y) >> 20) & 0x7ff;
    double x = (double)y;
    mmq_sc r;
    if (top < 0x3f4) {                                /* |y| < pi/4 */
        if (top < 0x398) { r.s = y; r.c = 1.0f; return r; }       /* |y| < 2^-12 */
        return mmq_sincosf_poly(x, x * x, 0, 0);
    }
    if (top < 0x42f) {                                /* |y| < 120: reduce_fast */
        const double rr = x * 0x1.45f306dc9c883p+23;
        const int n = ((int)rr + 0x800000) >> 24;
        x = MMQ_FMA(-(double)n, 0x1.921fb54442d18p+0, x);
        const double s = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
        return mmq_sincosf_poly(x * s, x * x, (n & 2) != 0, n);
    }
    if (top < 0x7f8) {                                /* reduce_large */
        unsigned xi = mmq_asuint(y);
        const int sign = (int)(xi >> 31);
        const unsigned *arr = &mmq_inv_pio4[(xi >> 26) & 15];
        const int shift = (int)((xi >> 23) & 7);
        unsigned long long n, res0, res1, res2;
        xi = (xi & 0xffffffu) | 0x800000u;
        xi <<= shift;
        res0 = (unsigned)(xi * arr[0]);
        res1 = (unsigned long long)xi * arr[4];
        res2 = (unsigned long long)xi * arr[8];
        res0 = (res2 >> 32) | (res0 << 32);
        res0 += res1;
        n = (res0 + (1ull << 61)) >> 62;
        res0 -= n << 62;
        x = (double)(long long)res0 * 0x1.921fb54442d18p-62;
        const int q = (int)n + sign;
        const double s = ((q & 3) == 1 || (q & 3) == 2) ? -1.0 : 1.0;
        return mmq_sincosf_poly(x * s, x * x, (q & 2) != 0, (int)n);
    }
    r.s = r.c = y - y;                                /* inf or nan */
    return r;
}

MMQ_FN mmq_sc mmq_sincosf(float y) {
    const unsigned top = (mmq_asuint(y) >> 20) & 0x7ff;
    if (MMQ_UNLIKELY(top >= 0x42f)) return mmq_ref_sincosf(y);       // |y| >= 120, inf, nan
    // |y| < 120: reduce_fast.  Below pi/4 it yields n = 0 and x unchanged (fma(-0, hpi, x) = x), i.e. exactly the
    // reference's direct polynomial; below 2^-12 the polynomial's value rounds to (y, 1) like the reference's shortcut.
    double x = (double)y;
    const double rr = x * 0x1.45f306dc9c883p+23;
    const int n = ((int)rr + 0x800000) >> 24;
    x = MMQ_FMA(-(double)n, 0x1.921fb54442d18p+0, x);
    const double s = ((n + 1) & 2) ? -1.0 : 1.0;                     // quadrants 1 and 2
    mmq_sc r = mmq_sincosf_poly(x * s, x * x, (n & 2) != 0, n);
    if (top < 0x398) { r.s = y; r.c = 1.0f; }                        // |y| < 2^-12 (keeps -0 and the subnormals bit for bit)
    return r;
}

// ---- atanf (s_atanf.c), atan2f (e_atan2f.c) ---------------------------------------------
MMQ_FN float mmq_ref_atanf(float x) {
    const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
    const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
    const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f, aT3 = -1.1111110449e-01f,
                aT4 = 9.0908870101e-02f, aT5 = -7.6918758452e-02f, aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f,
                aT8 = 4.9768779427e-02f, aT9 = -3.6531571299e-02f, aT10 = 1.6285819933e-02f;
    const int hx = (int)mmq_asuint(x), ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x4c000000) {                           /* |x| >= 2^25 */
        if (ix > 0x7f800000) return x + x;
        if (hx > 0) return atanhi[3] + atanlo[3];
        return -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3ee00000) {                            /* |x| < 0.4375 */
        if (ix < 0x31000000) return x;                /* |x| < 2^-29 */
        id = -1;
    } else {
        x = mmq_fabsf(x);
        if (ix < 0x3f980000) {
            if (ix < 0x3f300000) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); }
            else { id = 1; x = (x - 1.0f) / (x + 1.0f); }
        } else {
            if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); }
            else { id = 3; x = -1.0f / x; }
        }
    }
    const float z = x * x, w = z * z;
    const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    if (id < 0) return x - x * (s1 + s2);
    const float r = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return hx < 0 ? -r : r;
}

MMQ_FN float mmq_ref_atan2f(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f,
                pi_lo = -8.7422776573e-08f;
    const int hx = (int)mmq_asuint(x), hy = (int)mmq_asuint(y), ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    float z;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return mmq_ref_atanf(y);
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) {
        switch (m) {
            case 0: case 1: return y;
            case 2: return pi + tiny;
            default: return -pi - tiny;
        }
    }
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) {
            switch (m) {
                case 0: return pi_o_4 + tiny;
                case 1: return -pi_o_4 - tiny;
                case 2: return 3.0f * pi_o_4 + tiny;
                default: return -3.0f * pi_o_4 - tiny;
            }
        } else {
            switch (m) {
                case 0: return 0.0f;
                case 1: return -0.0f;
                case 2: return pi + tiny;
                default: return -pi - tiny;
            }
        }
    }
    if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int k = (iy - ix) >> 23;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = mmq_ref_atanf(mmq_fabsf(y / x));
    switch (m) {
        case 0: return z;
        case 1: return mmq_asfloat(mmq_asuint(z) ^ 0x80000000u);
        case 2: return pi - (z - pi_lo);
        default: return (z - pi_lo) - pi;
    }
}

// atanf without branches: the five ranges differ in how the argument is reduced -- a quotient num / den (x / 1 for
// the first range: exact) -- and in the constants added back; both are picked by selects, one division is made.
MMQ_FN float mmq_atanf(float x) {
    const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f, aT3 = -1.1111110449e-01f,
                aT4 = 9.0908870101e-02f, aT5 = -7.6918758452e-02f, aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f,
                aT8 = 4.9768779427e-02f, aT9 = -3.6531571299e-02f, aT10 = 1.6285819933e-02f;
    const int hx = (int)mmq_asuint(x), ix = hx & 0x7fffffff;
    const float ax = mmq_fabsf(x);
    const int r0 = ix < 0x3ee00000, r1 = ix < 0x3f300000, r2 = ix < 0x3f980000, r3 = ix < 0x401c0000;
    const float num = r0 ? x : r1 ? 2.0f * ax - 1.0f : r2 ? ax - 1.0f : r3 ? ax - 1.5f : -1.0f;
    const float den = r0 ? 1.0f : r1 ? 2.0f + ax : r2 ? ax + 1.0f : r3 ? 1.0f + 1.5f * ax : ax;
    const float hi = r1 ? 4.6364760399e-01f : r2 ? 7.8539812565e-01f : r3 ? 9.8279368877e-01f : 1.5707962513e+00f;
    const float lo = r1 ? 5.0121582440e-09f : r2 ? 3.7748947079e-08f : r3 ? 3.4473217170e-08f : 7.5497894159e-08f;
    const float xr = num / den;
    const float z = xr * xr, w = z * z;
    const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    const float small = ix < 0x31000000 ? x : xr - xr * (s1 + s2);    /* |x| < 0.4375; below 2^-29 the reference returns x (keeps -0) */
    const float big = hi - ((xr * (s1 + s2) - lo) - xr);
    float r = r0 ? small : (hx < 0 ? -big : big);
    if (ix >= 0x4c000000) {                                           /* |x| >= 2^25, inf, nan: constants */
        const float lim = 1.5707962513e+00f + 7.5497894159e-08f;
        r = ix > 0x7f800000 ? x + x : (hx > 0 ? lim : -1.5707962513e+00f - 7.5497894159e-08f);
    }
    return r;
}

MMQ_FN float mmq_atan2f(float y, float x) {
    const float pi = 3.1415927410e+00f, pi_o_2 = 1.5707963705e+00f, pi_lo = -8.7422776573e-08f;
    const int hx = (int)mmq_asuint(x), hy = (int)mmq_asuint(y), ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    // zeros, infinities and NaNs: the reference form.  (x == 1 is not special here: atanf is odd in this
    // implementation, so atanf(y) equals the general path's +-atanf(|y / 1|).)
    if (MMQ_UNLIKELY(ix - 1u >= 0x7f7fffffu || iy - 1u >= 0x7f7fffffu)) return mmq_ref_atan2f(y, x);
    const int k = (iy - ix) >> 23;
    float z = mmq_atanf(mmq_fabsf(y / x));
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    const float t = z - pi_lo;
    const float neg = mmq_asfloat(mmq_asuint(z) ^ 0x80000000u);
    return hx >= 0 ? (hy >= 0 ? z : neg) : (hy >= 0 ? pi - t : t - pi);
}

// ---- hypotf (e_hypotf.c, 2.35), x2y2m1f, scalbnf ---------------------------------------------
MMQ_FN float mmq_hypotf(float x, float y) {
    if (!mmq_isfinitef(x) || !mmq_isfinitef(y)) {
        if ((mmq_isinff(x) || mmq_isinff(y)) && !mmq_issignalingf(x) && !mmq_issignalingf(y)) return mmq_inff();
        return x + y;
    }
    return (float)MMQ_SQRT((double)x * (double)x + (double)y * (double)y);
}
MMQ_FN float mmq_x2y2m1f(float x, float y) {
    const double dx = x, dy = y;
    return (float)((dx - 1.0) * (dx + 1.0) + dy * dy);
}
MMQ_FN float mmq_scalbnf(float x, int n) {      /* exact scaling as used by the complex templates (no over/underflow there) */
    return (float)((double)x * mmq_asdouble((unsigned long long)(1023 + n) << 52));
}

// ---- log1pf (s_log1pf.c) ------------------------------------------------------------------
MMQ_FN float mmq_log1pf(float x) {
    const float ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f, two25 = 3.355443200e+07f;
    const float Lp1 = 6.6666668653e-01f, Lp2 = 4.0000000596e-01f, Lp3 = 2.8571429849e-01f, Lp4 = 2.2222198546e-01f,
                Lp5 = 1.8183572590e-01f, Lp6 = 1.5313838422e-01f, Lp7 = 1.4798198640e-01f;
    float hfsq, f = 0.0f, c = 0.0f, s, z, R, u;
    int k = 1, hu = 0;
    const int hx = (int)mmq_asuint(x), ax = hx & 0x7fffffff;
    if (hx < 0x3ed413d7) {                           /* x < 0.41422 */
        if (ax >= 0x3f800000) {                       /* x <= -1.0 */
            if (x == -1.0f) return -two25 / 0.0f;
            return (x - x) / (x - x);
        }
        if (ax < 0x31000000) {                        /* |x| < 2^-29 */
            if (ax < 0x24800000) return x;            /* |x| < 2^-54 */
            return x - x * x * 0.5f;
        }
        if (hx > 0 || hx <= (int)0xbe95f61f) { k = 0; f = x; hu = 1; }   /* -0.2929 < x < 0.41422 */
    }
    if (hx >= 0x7f800000) return x + x;
    if (k != 0) {
        if (hx < 0x5a000000) {
            u = 1.0f + x;
            hu = (int)mmq_asuint(u);
            k = (hu >> 23) - 127;
            c = (k > 0) ? 1.0f - (u - x) : x - (u - 1.0f);   /* correction term */
            c /= u;
        } else {
            u = x;
            hu = (int)mmq_asuint(u);
            k = (hu >> 23) - 127;
            c = 0;
        }
        hu &= 0x007fffff;
        if (hu < 0x3504f7) {
            u = mmq_asfloat((unsigned)hu | 0x3f800000u);       /* normalise u */
        } else {
            k += 1;
            u = mmq_asfloat((unsigned)hu | 0x3f000000u);       /* normalise u/2 */
            hu = (0x00800000 - hu) >> 2;
        }
        f = u - 1.0f;
    }
    hfsq = 0.5f * f * f;
    if (hu == 0) {                                    /* |f| < 2^-20 */
        if (f == 0.0f) {
            if (k == 0) return 0.0f;
            c += k * ln2_lo;
            return k * ln2_hi + c;
        }
        R = hfsq * (1.0f - 0.66666666666666666f * f);
        if (k == 0) return f - R;
        return k * ln2_hi - ((R - (k * ln2_lo + c)) - f);
    }
    s = f / (2.0f + f);
    z = s * s;
    R = z * (Lp1 + z * (Lp2 + z * (Lp3 + z * (Lp4 + z * (Lp5 + z * (Lp6 + z * Lp7))))));
    if (k == 0) return f - (hfsq - s * (hfsq + R));
    return k * ln2_hi - ((hfsq - (s * (hfsq + R) + (k * ln2_lo + c))) - f);
}

// ---- complex templates --------------------------------------------------------------------------------
MMQ_FN mmq_cf mmq_cmake(float re, float im) { mmq_cf c; c.re = re; c.im = im; return c; }

// fpclassify-based tests of the templates: "finite" = FP_ZERO / FP_SUBNORMAL / FP_NORMAL
MMQ_FN mmq_cf mmq_ref_cexpf(mmq_cf x) {
    mmq_cf r;
    const int rfin = mmq_isfinitef(x.re), ifin = mmq_isfinitef(x.im);
    if (rfin) {
        if (ifin) {
            const int t = 88;                      /* (int)((FLT_MAX_EXP - 1) * M_LN2) */
            float sinix, cosix;
            if (mmq_fabsf(x.im) > MMQ_FLT_MIN) { const mmq_sc sc = mmq_sincosf(x.im); sinix = sc.s; cosix = sc.c; }
            else { sinix = x.im; cosix = 1.0f; }
            float re = x.re;
            if (re > t) {
                const float exp_t = mmq_expf((float)t);
                re -= t; sinix *= exp_t; cosix *= exp_t;
                if (re > t) { re -= t; sinix *= exp_t; cosix *= exp_t; }
            }
            if (re > t) {                           /* overflow (original real part > 3t) */
                r.re = MMQ_FLT_MAX * cosix;
                r.im = MMQ_FLT_MAX * sinix;
            } else {
                const float exp_val = mmq_expf(re);
                r.re = exp_val * cosix;
                r.im = exp_val * sinix;
            }
            return r;
        }
        r.re = r.im = mmq_nanf();                  /* imaginary part inf or nan */
        return r;
    }
    if (mmq_isinff(x.re)) {
        if (ifin) {
            const float value = mmq_signbitf(x.re) ? 0.0f : mmq_inff();
            if (x.im == 0.0f) { r.re = value; r.im = x.im; return r; }
            float sinix, cosix;
            if (mmq_fabsf(x.im) > MMQ_FLT_MIN) { const mmq_sc sc = mmq_sincosf(x.im); sinix = sc.s; cosix = sc.c; }
            else { sinix = x.im; cosix = 1.0f; }
            r.re = mmq_copysignf(value, cosix);
            r.im = mmq_copysignf(value, sinix);
            return r;
        }
        if (!mmq_signbitf(x.re)) { r.re = mmq_inff(); r.im = x.im - x.im; return r; }
        r.re = 0.0f;
        r.im = mmq_copysignf(0.0f, x.im);
        return r;
    }
    r.re = mmq_nanf();                              /* real part nan */
    r.im = (x.im == 0.0f) ? x.im : mmq_nanf();
    return r;
}

MMQ_FN mmq_cf mmq_ref_clogf(mmq_cf x) {
    mmq_cf r;
    const int rnan = mmq_isnanf(x.re), inan = mmq_isnanf(x.im);
    if (x.re == 0.0f && x.im == 0.0f) {
        r.im = mmq_signbitf(x.re) ? 3.14159265358979323846f : 0.0f;
        r.im = mmq_copysignf(r.im, x.im);
        r.re = -1.0f / mmq_fabsf(x.re);
        return r;
    }
    if (!rnan && !inan) {
        float absx = mmq_fabsf(x.re), absy = mmq_fabsf(x.im);
        int scale = 0;
        if (absx < absy) { const float t = absx; absx = absy; absy = t; }
        if (absx > MMQ_FLT_MAX / 2.0f) {
            scale = -1;
            absx = mmq_scalbnf(absx, scale);
            absy = (absy >= MMQ_FLT_MIN * 2.0f ? mmq_scalbnf(absy, scale) : 0.0f);
        } else if (absx < MMQ_FLT_MIN && absy < MMQ_FLT_MIN) {
            scale = 24;
            absx = mmq_scalbnf(absx, scale);
            absy = mmq_scalbnf(absy, scale);
        }
        if (absx == 1.0f && scale == 0) {
            r.re = mmq_log1pf(absy * absy) / 2.0f;
        } else if (absx > 1.0f && absx < 2.0f && absy < 1.0f && scale == 0) {
            float d2m1 = (absx - 1.0f) * (absx + 1.0f);
            if (absy >= MMQ_FLT_EPSILON) d2m1 += absy * absy;
            r.re = mmq_log1pf(d2m1) / 2.0f;
        } else if (absx < 1.0f && absx >= 0.5f && absy < MMQ_FLT_EPSILON / 2.0f && scale == 0) {
            const float d2m1 = (absx - 1.0f) * (absx + 1.0f);
            r.re = mmq_log1pf(d2m1) / 2.0f;
        } else if (absx < 1.0f && absx >= 0.5f && scale == 0 && absx * absx + absy * absy >= 0.5f) {
            const float d2m1 = mmq_x2y2m1f(absx, absy);
            r.re = mmq_log1pf(d2m1) / 2.0f;
        } else {
            const float d = mmq_hypotf(absx, absy);
            r.re = mmq_logf(d) - scale * 0.69314718055994530942f;
        }
        r.im = mmq_atan2f(x.im, x.re);
        return r;
    }
    r.im = mmq_nanf();
    r.re = (mmq_isinff(x.re) || mmq_isinff(x.im)) ? mmq_inff() : mmq_nanf();
    return r;
}

MMQ_FN mmq_cf mmq_cexpf(mmq_cf x) {
    // both parts finite and the real part <= 88: sincos, exp, two products.  Everything else (overflow staging,
    // infinities, NaN) is the reference form.
    if (MMQ_UNLIKELY(!mmq_isfinitef(x.re) || !mmq_isfinitef(x.im) || x.re > 88.0f)) return mmq_ref_cexpf(x);
    mmq_sc sc = mmq_sincosf(x.im);
    if (!(mmq_fabsf(x.im) > MMQ_FLT_MIN)) { sc.s = x.im; sc.c = 1.0f; }
    const float exp_val = mmq_expf(x.re);
    mmq_cf r;
    r.re = exp_val * sc.c;
    r.im = exp_val * sc.s;
    return r;
}

MMQ_FN mmq_cf mmq_clogf(mmq_cf x) {
    float absx = mmq_fabsf(x.re), absy = mmq_fabsf(x.im);
    if (absx < absy) { const float t = absx; absx = absy; absy = t; }
    // NaN, both zero, or a modulus that the reference rescales (huge, or both parts subnormal): reference form
    if (MMQ_UNLIKELY(!(absx <= MMQ_FLT_MAX / 2.0f) || !(absy == absy) || absx < MMQ_FLT_MIN)) return mmq_ref_clogf(x);
    mmq_cf r;
    // scale == 0 from here.  Near |z| = 1 the reference goes through log1pf of |z|^2 - 1, formed in one of four ways
    const int c1 = absx == 1.0f;
    const int c2 = absx > 1.0f && absx < 2.0f && absy < 1.0f;
    const int c34 = absx < 1.0f && absx >= 0.5f;
    const int c3 = c34 && absy < MMQ_FLT_EPSILON / 2.0f;
    const int c4 = c34 && absx * absx + absy * absy >= 0.5f;
    if (c1 | c2 | c3 | c4) {
        const float y2 = absy * absy;
        const float t = (absx - 1.0f) * (absx + 1.0f);
        const float t2 = absy >= MMQ_FLT_EPSILON ? t + y2 : t;
        const float d2m1 = c1 ? y2 : c2 ? t2 : c3 ? t : mmq_x2y2m1f(absx, absy);
        r.re = mmq_log1pf(d2m1) / 2.0f;
    } else {
        const float d = mmq_hypotf(absx, absy);
        r.re = mmq_logf(d);                       /* (scale is 0: the reference subtracts 0 * ln2) */
    }
    r.im = mmq_atan2f(x.im, x.re);
    return r;
}

MMQ_FN float mmq_cargf(mmq_cf x) { return mmq_atan2f(x.im, x.re); }

// gcc's complex multiplication: the two products and, when both parts come out NaN, C99 Annex G's
// recovery of infinities (libgcc __mulsc3)
MMQ_FN mmq_cf mmq_cmulf(mmq_cf p, mmq_cf q) {
    float a = p.re, b = p.im, c = q.re, d = q.im;
    const float ac = a * c, bd = b * d, ad = a * d, bc = b * c;
    mmq_cf r;
    r.re = ac - bd;
    r.im = ad + bc;
    if (mmq_isnanf(r.re) && mmq_isnanf(r.im)) {
        int recalc = 0;
        if (mmq_isinff(a) || mmq_isinff(b)) {
            a = mmq_copysignf(mmq_isinff(a) ? 1.0f : 0.0f, a);
            b = mmq_copysignf(mmq_isinff(b) ? 1.0f : 0.0f, b);
            if (mmq_isnanf(c)) c = mmq_copysignf(0.0f, c);
            if (mmq_isnanf(d)) d = mmq_copysignf(0.0f, d);
            recalc = 1;
        }
        if (mmq_isinff(c) || mmq_isinff(d)) {
            c = mmq_copysignf(mmq_isinff(c) ? 1.0f : 0.0f, c);
            d = mmq_copysignf(mmq_isinff(d) ? 1.0f : 0.0f, d);
            if (mmq_isnanf(a)) a = mmq_copysignf(0.0f, a);
            if (mmq_isnanf(b)) b = mmq_copysignf(0.0f, b);
            recalc = 1;
        }
        if (!recalc && (mmq_isinff(ac) || mmq_isinff(bd) || mmq_isinff(ad) || mmq_isinff(bc))) {
            if (mmq_isnanf(a)) a = mmq_copysignf(0.0f, a);
            if (mmq_isnanf(b)) b = mmq_copysignf(0.0f, b);
            if (mmq_isnanf(c)) c = mmq_copysignf(0.0f, c);
            if (mmq_isnanf(d)) d = mmq_copysignf(0.0f, d);
            recalc = 1;
        }
        if (recalc) {
            r.re = mmq_inff() * (a * c - b * d);
            r.im = mmq_inff() * (a * d + b * c);
        }
    }
    return r;
}

MMQ_FN mmq_cf mmq_cpowf(mmq_cf x, mmq_cf c) { return mmq_cexpf(mmq_cmulf(c, mmq_clogf(x))); }

// ---- expm1f (s_expm1f.c), sinhf (e_sinhf.c), coshf (e_coshf.c) ------------------------------------------
#define MMQ_HAVE_BATCH2 1
MMQ_FN float mmq_expm1f(float x) {
    const float huge = 1.0e+30f, tiny = 1.0e-30f, o_threshold = 8.8721679688e+01f, ln2_hi = 6.9313812256e-01f,
                ln2_lo = 9.0580006145e-06f, invln2 = 1.4426950216e+00f, Q1 = -3.3333335072e-02f, Q2 = 1.5873016091e-03f,
                Q3 = -7.9365076090e-05f, Q4 = 4.0082177293e-06f, Q5 = -2.0109921195e-07f;
    float y, hi, lo, c = 0.0f, t, e, hxs, hfx, r1;
    int k;
    unsigned hx = mmq_asuint(x);
    const unsigned xsb = hx & 0x80000000u;
    hx &= 0x7fffffffu;
    if (hx >= 0x4195b844u) {                      /* |x| >= 27 ln2 */
        if (hx >= 0x42b17218u) {                  /* |x| >= 88.721... */
            if (hx > 0x7f800000u) return x + x;
            if (hx == 0x7f800000u) return xsb == 0 ? x : -1.0f;
            if (x > o_threshold) return huge * huge;
        }
        if (xsb != 0) return tiny - 1.0f;         /* x < -27 ln2 */
    }
    if (hx > 0x3eb17218u) {                       /* |x| > 0.5 ln2 */
        if (hx < 0x3F851592u) {                   /* and |x| < 1.5 ln2 */
            if (xsb == 0) { hi = x - ln2_hi; lo = ln2_lo; k = 1; }
            else { hi = x + ln2_hi; lo = -ln2_lo; k = -1; }
        } else {
            k = (int)(invln2 * x + (xsb == 0 ? 0.5f : -0.5f));
            t = (float)k;
            hi = x - t * ln2_hi;
            lo = t * ln2_lo;
        }
        x = hi - lo;
        c = (hi - x) - lo;
    } else if (hx < 0x33000000u) {                /* |x| < 2^-25 */
        t = huge + x;
        return x - (t - (huge + x));
    } else
        k = 0;
    hfx = 0.5f * x;
    hxs = x * hfx;
    r1 = 1.0f + hxs * (Q1 + hxs * (Q2 + hxs * (Q3 + hxs * (Q4 + hxs * Q5))));
    t = 3.0f - r1 * hfx;
    e = hxs * ((r1 - t) / (6.0f - x * t));
    if (k == 0) return x - (x * e - hxs);
    e = (x * (e - c) - c);
    e -= hxs;
    if (k == -1) return 0.5f * (x - e) - 0.5f;
    if (k == 1) {
        if (x < -0.25f) return -2.0f * (e - (x + 0.5f));
        return 1.0f + 2.0f * (x - e);
    }
    if (k <= -2 || k > 56) {                      /* suffice to return exp(x) - 1 */
        y = 1.0f - (e - x);
        y = mmq_asfloat(mmq_asuint(y) + ((unsigned)k << 23));
        return y - 1.0f;
    }
    if (k < 23) {
        t = mmq_asfloat(0x3f800000u - (0x1000000u >> k));      /* 1 - 2^-k */
        y = t - (e - x);
        y = mmq_asfloat(mmq_asuint(y) + ((unsigned)k << 23));
    } else {
        t = mmq_asfloat((unsigned)(0x7f - k) << 23);           /* 2^-k */
        y = x - (e + t);
        y += 1.0f;
        y = mmq_asfloat(mmq_asuint(y) + ((unsigned)k << 23));
    }
    return y;
}

MMQ_FN float mmq_sinhf(float x) {
    const float shuge = 1.0e37f;
    float t, w, h;
    const int jx = (int)mmq_asuint(x), ix = jx & 0x7fffffff;
    if (ix >= 0x7f800000) return x + x;
    h = 0.5f;
    if (jx < 0) h = -h;
    if (ix < 0x41b00000) {                        /* |x| < 22 */
        if (ix < 0x31800000) return x;            /* |x| < 2^-28 */
        t = mmq_expm1f(mmq_fabsf(x));
        if (ix < 0x3f800000) return h * (2.0f * t - t * t / (t + 1.0f));
        return h * (t + t / (t + 1.0f));
    }
    if (ix <= 0x42b1717f) return h * mmq_expf(mmq_fabsf(x));      /* |x| < 88.7216796875 (the binary's bound) */
    if (ix <= 0x42b2d4fc) {
        w = mmq_expf(0.5f * mmq_fabsf(x));
        t = h * w;
        return t * w;
    }
    return x * shuge;
}

MMQ_FN float mmq_coshf(float x) {
    const float huge = 1.0e30f;
    float t, w;
    const int ix = (int)(mmq_asuint(x) & 0x7fffffffu);
    if (ix < 0x41b00000) {                        /* |x| < 22 */
        if (ix < 0x3eb17218) {                    /* |x| < 0.5 ln2 */
            if (ix < 0x24000000) return 1.0f;
            t = mmq_expm1f(mmq_fabsf(x));
            w = 1.0f + t;
            return 1.0f + (t * t) / (w + w);
        }
        t = mmq_expf(mmq_fabsf(x));
        return 0.5f * t + 0.5f / t;
    }
    if (ix <= 0x42b1717f) return 0.5f * mmq_expf(mmq_fabsf(x));
    if (ix <= 0x42b2d4fc) {
        w = mmq_expf(0.5f * mmq_fabsf(x));
        t = 0.5f * w;
        return t * w;
    }
    if (ix >= 0x7f800000) return x * x;
    return huge * huge;
}

// ---- csqrtf, csinhf, ccoshf, csinf, ccosf, ctanhf, ctanf (math/s_c*_template.c) ---------------------------
MMQ_FN mmq_cf mmq_csqrtf(mmq_cf x) {
    mmq_cf res;
    const int rnan = mmq_isnanf(x.re), inan = mmq_isnanf(x.im), rinf = mmq_isinff(x.re), iinf = mmq_isinff(x.im);
    if (rnan || inan || rinf || iinf) {
        if (iinf) { res.re = mmq_inff(); res.im = x.im; }
        else if (rinf) {
            if (x.re < 0.0f) { res.re = inan ? mmq_nanf() : 0.0f; res.im = mmq_copysignf(mmq_inff(), x.im); }
            else { res.re = x.re; res.im = inan ? mmq_nanf() : mmq_copysignf(0.0f, x.im); }
        } else { res.re = mmq_nanf(); res.im = mmq_nanf(); }
        return res;
    }
    if (x.im == 0.0f) {
        if (x.re < 0.0f) { res.re = 0.0f; res.im = mmq_copysignf(MMQ_SQRTF(-x.re), x.im); }
        else { res.re = mmq_fabsf(MMQ_SQRTF(x.re)); res.im = mmq_copysignf(0.0f, x.im); }
        return res;
    }
    if (x.re == 0.0f) {
        float r;
        if (mmq_fabsf(x.im) >= 2.0f * MMQ_FLT_MIN) r = MMQ_SQRTF(0.5f * mmq_fabsf(x.im));
        else r = 0.5f * MMQ_SQRTF(2.0f * mmq_fabsf(x.im));
        res.re = r;
        res.im = mmq_copysignf(r, x.im);
        return res;
    }
    float d, r, s;
    int scale = 0;
    if (mmq_fabsf(x.re) > MMQ_FLT_MAX / 4.0f) {
        scale = 1;
        x.re = mmq_scalbnf(x.re, -2);
        x.im = mmq_scalbnf(x.im, -2);
    } else if (mmq_fabsf(x.im) > MMQ_FLT_MAX / 4.0f) {
        scale = 1;
        if (mmq_fabsf(x.re) >= 4.0f * MMQ_FLT_MIN) x.re = mmq_scalbnf(x.re, -2);
        else x.re = 0.0f;
        x.im = mmq_scalbnf(x.im, -2);
    } else if (mmq_fabsf(x.re) < 2.0f * MMQ_FLT_MIN && mmq_fabsf(x.im) < 2.0f * MMQ_FLT_MIN) {
        scale = -((24 + 1) / 2);
        x.re = mmq_scalbnf(x.re, -2 * scale);
        x.im = mmq_scalbnf(x.im, -2 * scale);
    }
    d = mmq_hypotf(x.re, x.im);
    if (x.re > 0.0f) {
        r = MMQ_SQRTF(0.5f * (d + x.re));
        if (scale == 1 && mmq_fabsf(x.im) < 1.0f) {
            s = x.im / r;
            r = mmq_scalbnf(r, scale);
            scale = 0;
        } else
            s = 0.5f * (x.im / r);
    } else {
        s = MMQ_SQRTF(0.5f * (d - x.re));
        if (scale == 1 && mmq_fabsf(x.im) < 1.0f) {
            r = mmq_fabsf(x.im / s);
            s = mmq_scalbnf(s, scale);
            scale = 0;
        } else
            r = mmq_fabsf(0.5f * (x.im / s));
    }
    if (scale) {
        r = mmq_scalbnf(r, scale);
        s = mmq_scalbnf(s, scale);
    }
    res.re = r;
    res.im = mmq_copysignf(s, x.im);
    return res;
}

MMQ_FN void mmq_sincos_or_tiny(float v, float *s, float *c) {
    if (mmq_fabsf(v) > MMQ_FLT_MIN) { const mmq_sc sc = mmq_sincosf(v); *s = sc.s; *c = sc.c; }
    else { *s = v; *c = 1.0f; }
}

MMQ_FN mmq_cf mmq_csinhf(mmq_cf x) {
    mmq_cf r;
    const int negate = mmq_signbitf(x.re);
    const int rfin = mmq_isfinitef(x.re), ifin = mmq_isfinitef(x.im);
    x.re = mmq_fabsf(x.re);
    if (rfin) {
        if (ifin) {
            const int t = 88;
            float sinix, cosix;
            mmq_sincos_or_tiny(x.im, &sinix, &cosix);
            if (negate) cosix = -cosix;
            if (mmq_fabsf(x.re) > t) {
                const float exp_t = mmq_expf((float)t);
                float rx = mmq_fabsf(x.re);
                rx -= t;
                sinix *= exp_t / 2.0f;
                cosix *= exp_t / 2.0f;
                if (rx > t) { rx -= t; sinix *= exp_t; cosix *= exp_t; }
                if (rx > t) { r.re = MMQ_FLT_MAX * cosix; r.im = MMQ_FLT_MAX * sinix; }
                else { const float ev = mmq_expf(rx); r.re = ev * cosix; r.im = ev * sinix; }
            } else {
                r.re = mmq_sinhf(x.re) * cosix;
                r.im = mmq_coshf(x.re) * sinix;
            }
            return r;
        }
        if (x.re == 0.0f) { r.re = mmq_copysignf(0.0f, negate ? -1.0f : 1.0f); r.im = x.im - x.im; }
        else { r.re = mmq_nanf(); r.im = mmq_nanf(); }
        return r;
    }
    if (mmq_isinff(x.re)) {
        if (ifin && x.im != 0.0f) {
            float sinix, cosix;
            mmq_sincos_or_tiny(x.im, &sinix, &cosix);
            r.re = mmq_copysignf(mmq_inff(), cosix);
            r.im = mmq_copysignf(mmq_inff(), sinix);
            if (negate) r.re = -r.re;
        } else if (x.im == 0.0f) {
            r.re = negate ? -mmq_inff() : mmq_inff();
            r.im = x.im;
        } else {
            r.re = mmq_inff();
            r.im = x.im - x.im;
        }
        return r;
    }
    r.re = mmq_nanf();
    r.im = x.im == 0.0f ? x.im : mmq_nanf();
    return r;
}

MMQ_FN mmq_cf mmq_ccoshf(mmq_cf x) {
    mmq_cf r;
    const int rfin = mmq_isfinitef(x.re), ifin = mmq_isfinitef(x.im);
    if (rfin) {
        if (ifin) {
            const int t = 88;
            float sinix, cosix;
            mmq_sincos_or_tiny(x.im, &sinix, &cosix);
            if (mmq_fabsf(x.re) > t) {
                const float exp_t = mmq_expf((float)t);
                float rx = mmq_fabsf(x.re);
                if (mmq_signbitf(x.re)) sinix = -sinix;
                rx -= t;
                sinix *= exp_t / 2.0f;
                cosix *= exp_t / 2.0f;
                if (rx > t) { rx -= t; sinix *= exp_t; cosix *= exp_t; }
                if (rx > t) { r.re = MMQ_FLT_MAX * cosix; r.im = MMQ_FLT_MAX * sinix; }
                else { const float ev = mmq_expf(rx); r.re = ev * cosix; r.im = ev * sinix; }
            } else {
                r.re = mmq_coshf(x.re) * cosix;
                r.im = mmq_sinhf(x.re) * sinix;
            }
            return r;
        }
        r.im = x.re == 0.0f ? 0.0f : mmq_nanf();
        r.re = x.im - x.im;
        return r;
    }
    if (mmq_isinff(x.re)) {
        if (ifin && x.im != 0.0f) {
            float sinix, cosix;
            mmq_sincos_or_tiny(x.im, &sinix, &cosix);
            r.re = mmq_copysignf(mmq_inff(), cosix);
            r.im = mmq_copysignf(mmq_inff(), sinix) * mmq_copysignf(1.0f, x.re);
        } else if (x.im == 0.0f) {
            r.re = mmq_inff();
            r.im = x.im * mmq_copysignf(1.0f, x.re);
        } else {
            r.re = mmq_inff();
            r.im = x.im - x.im;
        }
        return r;
    }
    r.re = mmq_nanf();
    r.im = x.im == 0.0f ? x.im : mmq_nanf();
    return r;
}

MMQ_FN mmq_cf mmq_csinf(mmq_cf x) {
    mmq_cf r;
    const int negate = mmq_signbitf(x.re);
    const int rfin = mmq_isfinitef(x.re), ifin = mmq_isfinitef(x.im);
    x.re = mmq_fabsf(x.re);
    if (ifin) {
        if (rfin) {
            const int t = 88;
            float sinix, cosix;
            if (x.re > MMQ_FLT_MIN) { const mmq_sc sc = mmq_sincosf(x.re); sinix = sc.s; cosix = sc.c; }
            else { sinix = x.re; cosix = 1.0f; }
            if (negate) sinix = -sinix;
            if (mmq_fabsf(x.im) > t) {
                const float exp_t = mmq_expf((float)t);
                float ix = mmq_fabsf(x.im);
                if (mmq_signbitf(x.im)) cosix = -cosix;
                ix -= t;
                sinix *= exp_t / 2.0f;
                cosix *= exp_t / 2.0f;
                if (ix > t) { ix -= t; sinix *= exp_t; cosix *= exp_t; }
                if (ix > t) { r.re = MMQ_FLT_MAX * sinix; r.im = MMQ_FLT_MAX * cosix; }
                else { const float ev = mmq_expf(ix); r.re = ev * sinix; r.im = ev * cosix; }
            } else {
                r.re = mmq_coshf(x.im) * sinix;
                r.im = mmq_sinhf(x.im) * cosix;
            }
            return r;
        }
        if (x.im == 0.0f) { r.re = x.re - x.re; r.im = x.im; }
        else { r.re = mmq_nanf(); r.im = mmq_nanf(); }
        return r;
    }
    if (mmq_isinff(x.im)) {
        if (x.re == 0.0f) {
            r.re = mmq_copysignf(0.0f, negate ? -1.0f : 1.0f);
            r.im = x.im;
        } else if (rfin) {
            float sinix, cosix;
            if (x.re > MMQ_FLT_MIN) { const mmq_sc sc = mmq_sincosf(x.re); sinix = sc.s; cosix = sc.c; }
            else { sinix = x.re; cosix = 1.0f; }
            r.re = mmq_copysignf(mmq_inff(), sinix);
            r.im = mmq_copysignf(mmq_inff(), cosix);
            if (negate) r.re = -r.re;
            if (mmq_signbitf(x.im)) r.im = -r.im;
        } else {
            r.re = x.re - x.re;
            r.im = mmq_inff();
        }
        return r;
    }
    r.re = (x.re == 0.0f) ? mmq_copysignf(0.0f, negate ? -1.0f : 1.0f) : mmq_nanf();
    r.im = mmq_nanf();
    return r;
}

MMQ_FN mmq_cf mmq_ccosf(mmq_cf x) { return mmq_ccoshf(mmq_cmake(-x.im, x.re)); }

MMQ_FN mmq_cf mmq_ctanhf(mmq_cf x) {
    mmq_cf res;
    if (!mmq_isfinitef(x.re) || !mmq_isfinitef(x.im)) {
        if (mmq_isinff(x.re)) {
            res.re = mmq_copysignf(1.0f, x.re);
            if (mmq_isfinitef(x.im) && mmq_fabsf(x.im) > 1.0f) {
                const mmq_sc sc = mmq_sincosf(x.im);
                res.im = mmq_copysignf(0.0f, sc.s * sc.c);
            } else
                res.im = mmq_copysignf(0.0f, x.im);
        } else if (x.im == 0.0f) {
            res = x;
        } else {
            res.re = (x.re == 0.0f) ? x.re : mmq_nanf();
            res.im = mmq_nanf();
        }
        return res;
    }
    float sinix, cosix, den;
    const int t = 44;                       /* (int)((FLT_MAX_EXP - 1) * M_LN2 / 2) */
    mmq_sincos_or_tiny(x.im, &sinix, &cosix);
    if (mmq_fabsf(x.re) > t) {
        const float exp_2t = mmq_expf((float)(2 * t));
        res.re = mmq_copysignf(1.0f, x.re);
        res.im = 4.0f * sinix * cosix;
        x.re = mmq_fabsf(x.re);
        x.re -= t;
        res.im /= exp_2t;
        if (x.re > t) res.im /= exp_2t;
        else res.im /= mmq_expf(2.0f * x.re);
    } else {
        float sinhrx, coshrx;
        if (mmq_fabsf(x.re) > MMQ_FLT_MIN) { sinhrx = mmq_sinhf(x.re); coshrx = mmq_coshf(x.re); }
        else { sinhrx = x.re; coshrx = 1.0f; }
        if (mmq_fabsf(sinhrx) > mmq_fabsf(cosix) * MMQ_FLT_EPSILON) den = sinhrx * sinhrx + cosix * cosix;
        else den = cosix * cosix;
        res.re = sinhrx * coshrx / den;
        res.im = sinix * cosix / den;
    }
    return res;
}

MMQ_FN mmq_cf mmq_ctanf(mmq_cf x) {
    mmq_cf res;
    if (!mmq_isfinitef(x.re) || !mmq_isfinitef(x.im)) {
        if (mmq_isinff(x.im)) {
            if (mmq_isfinitef(x.re) && mmq_fabsf(x.re) > 1.0f) {
                const mmq_sc sc = mmq_sincosf(x.re);
                res.re = mmq_copysignf(0.0f, sc.s * sc.c);
            } else
                res.re = mmq_copysignf(0.0f, x.re);
            res.im = mmq_copysignf(1.0f, x.im);
        } else if (x.re == 0.0f) {
            res = x;
        } else {
            res.re = mmq_nanf();
            res.im = (x.im == 0.0f) ? x.im : mmq_nanf();
        }
        return res;
    }
    float sinrx, cosrx, den;
    const int t = 44;
    mmq_sincos_or_tiny(x.re, &sinrx, &cosrx);
    if (mmq_fabsf(x.im) > t) {
        const float exp_2t = mmq_expf((float)(2 * t));
        res.im = mmq_copysignf(1.0f, x.im);
        res.re = 4.0f * sinrx * cosrx;
        x.im = mmq_fabsf(x.im);
        x.im -= t;
        res.re /= exp_2t;
        if (x.im > t) res.re /= exp_2t;
        else res.re /= mmq_expf(2.0f * x.im);
    } else {
        float sinhix, coshix;
        if (mmq_fabsf(x.im) > MMQ_FLT_MIN) { sinhix = mmq_sinhf(x.im); coshix = mmq_coshf(x.im); }
        else { sinhix = x.im; coshix = 1.0f; }
        if (mmq_fabsf(sinhix) > mmq_fabsf(cosrx) * MMQ_FLT_EPSILON) den = cosrx * cosrx + sinhix * sinhix;
        else den = cosrx * cosrx;
        res.re = sinrx * cosrx / den;
        res.im = sinhix * coshix / den;
    }
    return res;
}

// ---- casinhf / casinf / cacosf / cacoshf (math/k_casinh_template.c, s_casinh/casin/cacos/cacosh_template.c),
//      catanf / catanhf (s_catan_template.c, s_catanh_template.c) -----------------------------------------------
#define MMQ_HAVE_BATCH3 1
#define MMQ_PI_F 3.14159265358979323846f
#define MMQ_PI_2_F 1.57079632679489661923f
#define MMQ_PI_4_F 0.78539816339744830962f
#define MMQ_LN2_F 0.69314718055994530942f

MMQ_FN mmq_cf mmq_kernel_casinhf(mmq_cf x, int adj) {
    mmq_cf res, y;
    const float eps = MMQ_FLT_EPSILON;
    const float rx = mmq_fabsf(x.re), ix = mmq_fabsf(x.im);
    if (rx >= 1.0f / eps || ix >= 1.0f / eps) {
        y.re = rx;
        y.im = ix;
        if (adj) { const float t = y.re; y.re = mmq_copysignf(y.im, x.im); y.im = t; }
        res = mmq_clogf(y);
        res.re += MMQ_LN2_F;
    } else if (rx >= 0.5f && ix < eps / 8.0f) {
        const float s = mmq_hypotf(1.0f, rx);
        res.re = mmq_logf(rx + s);
        res.im = adj ? mmq_atan2f(s, x.im) : mmq_atan2f(ix, s);
    } else if (rx < eps / 8.0f && ix >= 1.5f) {
        const float s = MMQ_SQRTF((ix + 1.0f) * (ix - 1.0f));
        res.re = mmq_logf(ix + s);
        res.im = adj ? mmq_atan2f(rx, mmq_copysignf(s, x.im)) : mmq_atan2f(s, rx);
    } else if (ix > 1.0f && ix < 1.5f && rx < 0.5f) {
        if (rx < eps * eps) {
            const float ix2m1 = (ix + 1.0f) * (ix - 1.0f);
            const float s = MMQ_SQRTF(ix2m1);
            res.re = mmq_log1pf(2.0f * (ix2m1 + ix * s)) / 2.0f;
            res.im = adj ? mmq_atan2f(rx, mmq_copysignf(s, x.im)) : mmq_atan2f(s, rx);
        } else {
            const float ix2m1 = (ix + 1.0f) * (ix - 1.0f);
            const float rx2 = rx * rx;
            const float f = rx2 * (2.0f + rx2 + 2.0f * ix * ix);
            const float d = MMQ_SQRTF(ix2m1 * ix2m1 + f);
            const float dp = d + ix2m1;
            const float dm = f / dp;
            const float r1 = MMQ_SQRTF((dm + rx2) / 2.0f);
            const float r2 = rx * ix / r1;
            res.re = mmq_log1pf(rx2 + dp + 2.0f * (rx * r1 + ix * r2)) / 2.0f;
            res.im = adj ? mmq_atan2f(rx + r1, mmq_copysignf(ix + r2, x.im)) : mmq_atan2f(ix + r2, rx + r1);
        }
    } else if (ix == 1.0f && rx < 0.5f) {
        if (rx < eps / 8.0f) {
            res.re = mmq_log1pf(2.0f * (rx + MMQ_SQRTF(rx))) / 2.0f;
            res.im = adj ? mmq_atan2f(MMQ_SQRTF(rx), mmq_copysignf(1.0f, x.im)) : mmq_atan2f(1.0f, MMQ_SQRTF(rx));
        } else {
            const float d = rx * MMQ_SQRTF(4.0f + rx * rx);
            const float s1 = MMQ_SQRTF((d + rx * rx) / 2.0f);
            const float s2 = MMQ_SQRTF((d - rx * rx) / 2.0f);
            res.re = mmq_log1pf(rx * rx + d + 2.0f * (rx * s1 + s2)) / 2.0f;
            res.im = adj ? mmq_atan2f(rx + s1, mmq_copysignf(1.0f + s2, x.im)) : mmq_atan2f(1.0f + s2, rx + s1);
        }
    } else if (ix < 1.0f && rx < 0.5f) {
        if (ix >= eps) {
            if (rx < eps * eps) {
                const float onemix2 = (1.0f + ix) * (1.0f - ix);
                const float s = MMQ_SQRTF(onemix2);
                res.re = mmq_log1pf(2.0f * rx / s) / 2.0f;
                res.im = adj ? mmq_atan2f(s, x.im) : mmq_atan2f(ix, s);
            } else {
                const float onemix2 = (1.0f + ix) * (1.0f - ix);
                const float rx2 = rx * rx;
                const float f = rx2 * (2.0f + rx2 + 2.0f * ix * ix);
                const float d = MMQ_SQRTF(onemix2 * onemix2 + f);
                const float dp = d + onemix2;
                const float dm = f / dp;
                const float r1 = MMQ_SQRTF((dp + rx2) / 2.0f);
                const float r2 = rx * ix / r1;
                res.re = mmq_log1pf(rx2 + dm + 2.0f * (rx * r1 + ix * r2)) / 2.0f;
                res.im = adj ? mmq_atan2f(rx + r1, mmq_copysignf(ix + r2, x.im)) : mmq_atan2f(ix + r2, rx + r1);
            }
        } else {
            const float s = mmq_hypotf(1.0f, rx);
            res.re = mmq_log1pf(2.0f * rx * (rx + s)) / 2.0f;
            res.im = adj ? mmq_atan2f(s, x.im) : mmq_atan2f(ix, s);
        }
    } else {
        y.re = (rx - ix) * (rx + ix) + 1.0f;
        y.im = 2.0f * rx * ix;
        y = mmq_csqrtf(y);
        y.re += rx;
        y.im += ix;
        if (adj) { const float t = y.re; y.re = mmq_copysignf(y.im, x.im); y.im = t; }
        res = mmq_clogf(y);
    }
    res.re = mmq_copysignf(res.re, x.re);
    res.im = mmq_copysignf(res.im, adj ? 1.0f : x.im);
    return res;
}

MMQ_FN mmq_cf mmq_casinhf(mmq_cf x) {
    mmq_cf res;
    const int rnan = mmq_isnanf(x.re), inan = mmq_isnanf(x.im), rinf = mmq_isinff(x.re), iinf = mmq_isinff(x.im);
    if (rnan || inan || rinf || iinf) {
        if (iinf) {
            res.re = mmq_copysignf(mmq_inff(), x.re);
            if (rnan) res.im = mmq_nanf();
            else res.im = mmq_copysignf(!rinf ? MMQ_PI_2_F : MMQ_PI_4_F, x.im);
        } else if (rnan || rinf) {
            res.re = x.re;
            if ((rinf && !inan) || (rnan && x.im == 0.0f)) res.im = mmq_copysignf(0.0f, x.im);
            else res.im = mmq_nanf();
        } else {
            res.re = mmq_nanf();
            res.im = mmq_nanf();
        }
        return res;
    }
    if (x.re == 0.0f && x.im == 0.0f) return x;
    return mmq_kernel_casinhf(x, 0);
}

MMQ_FN mmq_cf mmq_casinf(mmq_cf x) {
    mmq_cf res;
    if (mmq_isnanf(x.re) || mmq_isnanf(x.im)) {
        if (x.re == 0.0f) return x;
        if (mmq_isinff(x.re) || mmq_isinff(x.im)) { res.re = mmq_nanf(); res.im = mmq_copysignf(mmq_inff(), x.im); }
        else { res.re = mmq_nanf(); res.im = mmq_nanf(); }
        return res;
    }
    const mmq_cf y = mmq_casinhf(mmq_cmake(-x.im, x.re));
    res.re = y.im;
    res.im = -y.re;
    return res;
}

MMQ_FN mmq_cf mmq_cacosf(mmq_cf x) {
    mmq_cf y, res;
    const int special = !mmq_isfinitef(x.re) || !mmq_isfinitef(x.im) || (x.re == 0.0f && x.im == 0.0f);
    if (special) {
        y = mmq_casinf(x);
        res.re = MMQ_PI_2_F - y.re;
        if (res.re == 0.0f) res.re = 0.0f;
        res.im = -y.im;
    } else {
        y = mmq_kernel_casinhf(mmq_cmake(-x.im, x.re), 1);
        res.re = y.im;
        res.im = y.re;
    }
    return res;
}

MMQ_FN mmq_cf mmq_cacoshf(mmq_cf x) {
    mmq_cf res;
    const int rnan = mmq_isnanf(x.re), inan = mmq_isnanf(x.im), rinf = mmq_isinff(x.re), iinf = mmq_isinff(x.im);
    if (rnan || inan || rinf || iinf) {
        if (iinf) {
            res.re = mmq_inff();
            if (rnan) res.im = mmq_nanf();
            else res.im = mmq_copysignf(rinf ? (x.re < 0.0f ? MMQ_PI_F - MMQ_PI_4_F : MMQ_PI_4_F) : MMQ_PI_2_F, x.im);
        } else if (rinf) {
            res.re = mmq_inff();
            if (!inan) res.im = mmq_copysignf(mmq_signbitf(x.re) ? MMQ_PI_F : 0.0f, x.im);
            else res.im = mmq_nanf();
        } else {
            res.re = mmq_nanf();
            res.im = (x.re == 0.0f) ? MMQ_PI_2_F : mmq_nanf();
        }
        return res;
    }
    if (x.re == 0.0f && x.im == 0.0f) {
        res.re = 0.0f;
        res.im = mmq_copysignf(MMQ_PI_2_F, x.im);
        return res;
    }
    const mmq_cf y = mmq_kernel_casinhf(mmq_cmake(-x.im, x.re), 1);
    if (mmq_signbitf(x.im)) { res.re = y.re; res.im = -y.im; }
    else { res.re = -y.re; res.im = y.im; }
    return res;
}

// the denominator 1 - |z|^2 shared by catanf and catanhf (absx >= absy already ordered by the caller)
MMQ_FN float mmq_catan_den(float absx, float absy) {
    float den;
    if (absy < MMQ_FLT_EPSILON / 2.0f) {
        den = (1.0f - absx) * (1.0f + absx);
        if (den == 0.0f) den = 0.0f;
    } else if (absx >= 1.0f)
        den = (1.0f - absx) * (1.0f + absx) - absy * absy;
    else if (absx >= 0.75f || absy >= 0.5f)
        den = -mmq_x2y2m1f(absx, absy);
    else
        den = (1.0f - absx) * (1.0f + absx) - absy * absy;
    return den;
}

MMQ_FN mmq_cf mmq_catanf(mmq_cf x) {
    mmq_cf res;
    const int rnan = mmq_isnanf(x.re), inan = mmq_isnanf(x.im), rinf = mmq_isinff(x.re), iinf = mmq_isinff(x.im);
    const float eps = MMQ_FLT_EPSILON;
    if (rnan || inan || rinf || iinf) {
        if (rinf) {
            res.re = mmq_copysignf(MMQ_PI_2_F, x.re);
            res.im = mmq_copysignf(0.0f, x.im);
        } else if (iinf) {
            res.re = !rnan ? mmq_copysignf(MMQ_PI_2_F, x.re) : mmq_nanf();
            res.im = mmq_copysignf(0.0f, x.im);
        } else if (x.im == 0.0f) {
            res.re = mmq_nanf();
            res.im = mmq_copysignf(0.0f, x.im);
        } else {
            res.re = mmq_nanf();
            res.im = mmq_nanf();
        }
        return res;
    }
    if (x.re == 0.0f && x.im == 0.0f) return x;
    if (mmq_fabsf(x.re) >= 16.0f / eps || mmq_fabsf(x.im) >= 16.0f / eps) {
        res.re = mmq_copysignf(MMQ_PI_2_F, x.re);
        if (mmq_fabsf(x.re) <= 1.0f) res.im = 1.0f / x.im;
        else if (mmq_fabsf(x.im) <= 1.0f) res.im = x.im / x.re / x.re;
        else {
            const float h = mmq_hypotf(x.re / 2.0f, x.im / 2.0f);
            res.im = x.im / h / h / 4.0f;
        }
        return res;
    }
    float absx = mmq_fabsf(x.re), absy = mmq_fabsf(x.im);
    if (absx < absy) { const float t = absx; absx = absy; absy = t; }
    float den = mmq_catan_den(absx, absy);
    res.re = 0.5f * mmq_atan2f(2.0f * x.re, den);
    if (mmq_fabsf(x.im) == 1.0f && mmq_fabsf(x.re) < eps * eps)
        res.im = mmq_copysignf(0.5f, x.im) * (MMQ_LN2_F - mmq_logf(mmq_fabsf(x.re)));
    else {
        float r2 = 0.0f, num, f;
        if (mmq_fabsf(x.re) >= eps * eps) r2 = x.re * x.re;
        num = x.im + 1.0f;
        num = r2 + num * num;
        den = x.im - 1.0f;
        den = r2 + den * den;
        f = num / den;
        if (f < 0.5f) res.im = 0.25f * mmq_logf(f);
        else {
            num = 4.0f * x.im;
            res.im = 0.25f * mmq_log1pf(num / den);
        }
    }
    return res;
}

MMQ_FN mmq_cf mmq_catanhf(mmq_cf x) {
    mmq_cf res;
    const int rnan = mmq_isnanf(x.re), inan = mmq_isnanf(x.im), rinf = mmq_isinff(x.re), iinf = mmq_isinff(x.im);
    const float eps = MMQ_FLT_EPSILON;
    if (rnan || inan || rinf || iinf) {
        if (iinf) {
            res.re = mmq_copysignf(0.0f, x.re);
            res.im = mmq_copysignf(MMQ_PI_2_F, x.im);
        } else if (rinf || x.re == 0.0f) {
            res.re = mmq_copysignf(0.0f, x.re);
            res.im = !inan ? mmq_copysignf(MMQ_PI_2_F, x.im) : mmq_nanf();
        } else {
            res.re = mmq_nanf();
            res.im = mmq_nanf();
        }
        return res;
    }
    if (x.re == 0.0f && x.im == 0.0f) return x;
    if (mmq_fabsf(x.re) >= 16.0f / eps || mmq_fabsf(x.im) >= 16.0f / eps) {
        res.im = mmq_copysignf(MMQ_PI_2_F, x.im);
        if (mmq_fabsf(x.im) <= 1.0f) res.re = 1.0f / x.re;
        else if (mmq_fabsf(x.re) <= 1.0f) res.re = x.re / x.im / x.im;
        else {
            const float h = mmq_hypotf(x.re / 2.0f, x.im / 2.0f);
            res.re = x.re / h / h / 4.0f;
        }
        return res;
    }
    if (mmq_fabsf(x.re) == 1.0f && mmq_fabsf(x.im) < eps * eps)
        res.re = mmq_copysignf(0.5f, x.re) * (MMQ_LN2_F - mmq_logf(mmq_fabsf(x.im)));
    else {
        float i2 = 0.0f;
        if (mmq_fabsf(x.im) >= eps * eps) i2 = x.im * x.im;
        float num = 1.0f + x.re;
        num = i2 + num * num;
        float den = 1.0f - x.re;
        den = i2 + den * den;
        const float f = num / den;
        if (f < 0.5f) res.re = 0.25f * mmq_logf(f);
        else {
            num = 4.0f * x.re;
            res.re = 0.25f * mmq_log1pf(num / den);
        }
    }
    float absx = mmq_fabsf(x.re), absy = mmq_fabsf(x.im);
    if (absx < absy) { const float t = absx; absx = absy; absy = t; }
    const float den2 = mmq_catan_den(absx, absy);
    res.im = 0.5f * mmq_atan2f(2.0f * x.im, den2);
    return res;
}

#endif  /* MM_GLIBCF_H */
