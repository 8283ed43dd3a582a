// The three operators the reference implements with GSL and GLib, restated from the published
// algorithms -- GSL and GLib are not in this image and not vendored by the reference, and its
// test-suite has no golden for a filter that uses them: PARITY UNPINNED.
//
//   SOLVE_LINEAR_2 / _3  opmacros.h:84-96   gsl_linalg_HH_solve (GSL linalg/hh.c): x = b, then
//                        Householder transformations of A applied to x, back substitution; with
//                        the error handler off (mathmap.c:353) a singular matrix returns x as far
//                        as it got, which is what this does too.
//   ELL_JAC              opmacros.h:118-126 gsl_sf_elljac_e (GSL specfunc/elljac.c): descending
//                        Landen / AGM, N = 16; |m| > 1 yields sn = cn = dn = 0.
//   RAND                 opmacros.h:127     g_random_double_range(a, b) = a + u (b - a) with u from
//                        GLib's process-global Mersenne Twister seeded from /dev/urandom -- the
//                        reference's output is not reproducible run to run, and depends on pixel
//                        order.  Here u comes from a counter-based hash of (column, row, frame,
//                        call number within the pixel), built like g_rand_double from two 32-bit
//                        words (u = (w0 2^-32 + w1) 2^-32, retried while >= 1), so a render is
//                        deterministic and independent of how the frame is split into launches.
//
// C and C++ (the oracle and the device prelude include the same text).
#ifndef MM_GSLMATH_H
#define MM_GSLMATH_H

#ifndef MMG_FN
#define MMG_FN static inline
#endif

#define MMG_DBL_EPSILON 2.2204460492503131e-16

// x (N floats in/out as doubles), A (N*N, row major), both overwritten.
MMG_FN void mmg_hh_svx(int N, double *A, double *x) {
    double d[3];
    int i, j, k;
    for (i = 0; i < N; i++) {
        const double aii = A[i * N + i];
        double alpha, f, ak, max_norm = 0.0, r = 0.0;
        for (k = i; k < N; k++) {
            const double aki = A[k * N + i];
            r += aki * aki;
        }
        if (r == 0.0) return;                       /* rank deficient */
        alpha = sqrt(r) * (aii >= 0.0 ? 1.0 : -1.0);   /* GSL_SIGN */
        ak = 1.0 / (r + alpha * aii);
        A[i * N + i] = aii + alpha;
        d[i] = -alpha;
        for (k = i + 1; k < N; k++) {
            double norm = 0.0;
            f = 0.0;
            for (j = i; j < N; j++) {
                const double ajk = A[j * N + k], aji = A[j * N + i];
                norm += ajk * ajk;
                f += ajk * aji;
            }
            max_norm = max_norm > norm ? max_norm : norm;
            f *= ak;
            for (j = i; j < N; j++) A[j * N + k] = A[j * N + k] - f * A[j * N + i];
        }
        if (fabs(alpha) < 2.0 * MMG_DBL_EPSILON * sqrt(max_norm)) return;   /* apparent singularity */
        f = 0.0;
        for (j = i; j < N; j++) f += x[j] * A[j * N + i];
        f *= ak;
        for (j = i; j < N; j++) x[j] = x[j] - f * A[j * N + i];
    }
    for (i = N; i-- > 0;) {
        double sum = 0.0;
        for (k = i + 1; k < N; k++) sum += A[i * N + k] * x[k];
        x[i] = (x[i] - sum) / d[i];
    }
}

MMG_FN void mmg_elljac(double u, double m, double *sn, double *cn, double *dn) {
    if (fabs(m) > 1.0) {
        *sn = 0.0;
        *cn = 0.0;
        *dn = 0.0;
    } else if (fabs(m) < 2.0 * MMG_DBL_EPSILON) {
        *sn = sin(u);
        *cn = cos(u);
        *dn = 1.0;
    } else if (fabs(m - 1.0) < 2.0 * MMG_DBL_EPSILON) {
        *sn = tanh(u);
        *cn = 1.0 / cosh(u);
        *dn = *cn;
    } else {
        double mu[16], nu[16], c[16], d[16];
        double sin_umu, cos_umu, t, r;
        int n = 0;
        mu[0] = 1.0;
        nu[0] = sqrt(1.0 - m);
        while (fabs(mu[n] - nu[n]) > 4.0 * MMG_DBL_EPSILON * fabs(mu[n] + nu[n])) {
            mu[n + 1] = 0.5 * (mu[n] + nu[n]);
            nu[n + 1] = sqrt(mu[n] * nu[n]);
            ++n;
            if (n >= 15) break;
        }
        sin_umu = sin(u * mu[n]);
        cos_umu = cos(u * mu[n]);
        /* sin(u mu) can be zero: when |sin| < |cos| work with sn(K-u), cn(K-u), dn(K-u) */
        if (fabs(sin_umu) < fabs(cos_umu)) {
            t = sin_umu / cos_umu;
            c[n] = mu[n] * t;
            d[n] = 1.0;
            while (n > 0) {
                n--;
                c[n] = d[n + 1] * c[n + 1];
                r = (c[n + 1] * c[n + 1]) / mu[n + 1];
                d[n] = (r + nu[n]) / (r + mu[n]);
            }
            *dn = sqrt(1.0 - m) / d[n];
            *cn = (*dn) * (cos_umu >= 0.0 ? 1.0 : -1.0) / hypot(1.0, c[n]);
            *sn = (*cn) * c[n] / sqrt(1.0 - m);
        } else {
            t = cos_umu / sin_umu;
            c[n] = mu[n] * t;
            d[n] = 1.0;
            while (n > 0) {
                --n;
                c[n] = d[n + 1] * c[n + 1];
                r = (c[n + 1] * c[n + 1]) / mu[n + 1];
                d[n] = (r + nu[n]) / (r + mu[n]);
            }
            *dn = d[n];
            *sn = (sin_umu >= 0.0 ? 1.0 : -1.0) / hypot(1.0, c[n]);
            *cn = c[n] * (*sn);
        }
    }
}

// 32-bit finaliser (the "lowbias32" constants), applied to a combination of the four keys.
MMG_FN unsigned mmg_hash32(unsigned x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

// u in [0, 1) for call number `ctr` of pixel (col, row) of frame `frame`.
MMG_FN double mmg_rand_unit(int col, int row, int frame, unsigned ctr) {
    unsigned key = mmg_hash32((unsigned)col * 0x9e3779b1U ^ mmg_hash32((unsigned)row ^ mmg_hash32((unsigned)frame + 0x85ebca6bU)));
    unsigned n = ctr * 4u;
    for (;;) {
        const unsigned w0 = mmg_hash32(key + n), w1 = mmg_hash32(key + n + 1u);
        double v = (double)w0 * 2.3283064365386962890625e-10;       /* G_RAND_DOUBLE_TRANSFORM */
        v = (v + (double)w1) * 2.3283064365386962890625e-10;
        if (v < 1.0) return v;
        n += 2u;          /* "very bad rounding luck": draw again, like g_rand_double */
    }
}

#endif  // MM_GSLMATH_H
