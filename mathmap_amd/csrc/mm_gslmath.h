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
//   ELL_INT_*            opmacros.h:102-117 gsl_sf_ellint_{Kcomp,Ecomp,F,E,P,D,RC,RD,RF,RJ} with
//                        GSL_PREC_SINGLE (GSL specfunc/ellint.c): Carlson's duplication algorithms
//                        (the SLATEC RC/RD/RF/RJ routines) iterated until the relative deviations drop
//                        below errtol = 0.03 and finished with the 5th-order series; the Legendre forms
//                        reduce phi to (-pi/2, pi/2] by nc = floor(phi/pi + 0.5) and add 2 nc times the
//                        complete integral; complete K and E switch to the Abramowitz-Stegun 17.3.34 /
//                        17.3.36 polynomials for 1 - k^2 < sqrt(DBL_EPSILON).  Domain errors return NaN
//                        (the error handler is off, mathmap.c:353).  D takes the GSL >= 2 form without n.
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

// ---- elliptic integrals ------------------------------------------------------------------------
#define MMG_NAN (0.0 / 0.0)
#define MMG_ELL_ERRTOL 0.03          /* GSL_PREC_SINGLE */
#define MMG_ELL_NMAX 10000
#define MMG_PI 3.14159265358979323846

MMG_FN double mmg_max3(double a, double b, double c) { a = a > b ? a : b; return a > c ? a : c; }

MMG_FN double mmg_ellint_RC(double x, double y) {
    const double lolim = 1.1125369292536007e-307, uplim = 3.5953862697246315e+307;   /* 5 DBL_MIN, 0.2 DBL_MAX */
    double xn = x, yn = y, mu, sn, lamda, s;
    int n = 0;
    if (x < 0.0 || y < 0.0 || x + y < lolim) return MMG_NAN;
    if (!((x > y ? x : y) < uplim)) return MMG_NAN;
    for (;;) {
        mu = (xn + yn + yn) / 3.0;
        sn = (yn + mu) / mu - 2.0;
        if (fabs(sn) < MMG_ELL_ERRTOL) break;
        lamda = 2.0 * sqrt(xn) * sqrt(yn) + yn;
        xn = (xn + lamda) * 0.25;
        yn = (yn + lamda) * 0.25;
        if (++n == MMG_ELL_NMAX) return MMG_NAN;
    }
    s = sn * sn * (0.3 + sn * (1.0 / 7.0 + sn * (0.375 + sn * (9.0 / 22.0))));
    return (1.0 + s) / sqrt(mu);
}

MMG_FN double mmg_ellint_RD(double x, double y, double z) {
    const double lolim = 6.2789393636470293e-206, uplim = 2.6293670456603617e+203;
    const double c1 = 3.0 / 14.0, c2 = 1.0 / 6.0, c3 = 9.0 / 22.0, c4 = 3.0 / 26.0;
    double xn = x, yn = y, zn = z, sigma = 0.0, power4 = 1.0, mu, xndev, yndev, zndev;
    double ea, eb, ec, ed, ef, s1, s2;
    int n = 0;
    if ((x < y ? x : y) < 0.0 || (x + y < z ? x + y : z) < lolim) return MMG_NAN;
    if (!(mmg_max3(x, y, z) < uplim)) return MMG_NAN;
    for (;;) {
        double xnroot, ynroot, znroot, lamda, epslon;
        mu = (xn + yn + 3.0 * zn) * 0.2;
        xndev = (mu - xn) / mu;
        yndev = (mu - yn) / mu;
        zndev = (mu - zn) / mu;
        epslon = mmg_max3(fabs(xndev), fabs(yndev), fabs(zndev));
        if (epslon < MMG_ELL_ERRTOL) break;
        xnroot = sqrt(xn);
        ynroot = sqrt(yn);
        znroot = sqrt(zn);
        lamda = xnroot * (ynroot + znroot) + ynroot * znroot;
        sigma += power4 / (znroot * (zn + lamda));
        power4 *= 0.25;
        xn = (xn + lamda) * 0.25;
        yn = (yn + lamda) * 0.25;
        zn = (zn + lamda) * 0.25;
        if (++n == MMG_ELL_NMAX) return MMG_NAN;
    }
    ea = xndev * yndev;
    eb = zndev * zndev;
    ec = ea - eb;
    ed = ea - 6.0 * eb;
    ef = ed + ec + ec;
    s1 = ed * (-c1 + 0.25 * c3 * ed - 1.5 * c4 * zndev * ef);
    s2 = zndev * (c2 * ef + zndev * (-c3 * ec + zndev * c4 * ea));
    return 3.0 * sigma + power4 * (1.0 + s1 + s2) / (mu * sqrt(mu));
}

MMG_FN double mmg_ellint_RF(double x, double y, double z) {
    const double lolim = 1.1125369292536007e-307, uplim = 3.5953862697246315e+307;
    const double c1 = 1.0 / 24.0, c2 = 3.0 / 44.0, c3 = 1.0 / 14.0;
    double xn = x, yn = y, zn = z, mu, xndev, yndev, zndev, e2, e3, s;
    int n = 0;
    if (x < 0.0 || y < 0.0 || z < 0.0) return MMG_NAN;
    if (x + y < lolim || x + z < lolim || y + z < lolim) return MMG_NAN;
    if (!(mmg_max3(x, y, z) < uplim)) return MMG_NAN;
    for (;;) {
        double epslon, lamda, xnroot, ynroot, znroot;
        mu = (xn + yn + zn) / 3.0;
        xndev = 2.0 - (mu + xn) / mu;
        yndev = 2.0 - (mu + yn) / mu;
        zndev = 2.0 - (mu + zn) / mu;
        epslon = mmg_max3(fabs(xndev), fabs(yndev), fabs(zndev));
        if (epslon < MMG_ELL_ERRTOL) break;
        xnroot = sqrt(xn);
        ynroot = sqrt(yn);
        znroot = sqrt(zn);
        lamda = xnroot * (ynroot + znroot) + ynroot * znroot;
        xn = (xn + lamda) * 0.25;
        yn = (yn + lamda) * 0.25;
        zn = (zn + lamda) * 0.25;
        if (++n == MMG_ELL_NMAX) return MMG_NAN;
    }
    e2 = xndev * yndev - zndev * zndev;
    e3 = xndev * yndev * zndev;
    s = 1.0 + (c1 * e2 - 0.1 - c2 * e3) * e2 + c3 * e3;
    return s / sqrt(mu);
}

MMG_FN double mmg_ellint_RJ(double x, double y, double z, double p) {
    const double lolim = 4.8095540743117414e-103, uplim = 9.9015482149165372e+101;
    const double c1 = 3.0 / 14.0, c2 = 1.0 / 3.0, c3 = 3.0 / 22.0, c4 = 3.0 / 26.0;
    double xn = x, yn = y, zn = z, pn = p, sigma = 0.0, power4 = 1.0, mu, xndev, yndev, zndev, pndev;
    double ea, eb, ec, e2, e3, s1, s2, s3;
    int n = 0;
    if (x < 0.0 || y < 0.0 || z < 0.0) return MMG_NAN;
    if (x + y < lolim || x + z < lolim || y + z < lolim || p < lolim) return MMG_NAN;
    { const double m3 = mmg_max3(x, y, z); if (!((m3 > p ? m3 : p) < uplim)) return MMG_NAN; }
    for (;;) {
        double xnroot, ynroot, znroot, lamda, alfa, beta, epslon, rc;
        mu = (xn + yn + zn + pn + pn) * 0.2;
        xndev = (mu - xn) / mu;
        yndev = (mu - yn) / mu;
        zndev = (mu - zn) / mu;
        pndev = (mu - pn) / mu;
        epslon = mmg_max3(fabs(xndev), fabs(yndev), fabs(zndev));
        if (fabs(pndev) > epslon) epslon = fabs(pndev);
        if (epslon < MMG_ELL_ERRTOL) break;
        xnroot = sqrt(xn);
        ynroot = sqrt(yn);
        znroot = sqrt(zn);
        lamda = xnroot * (ynroot + znroot) + ynroot * znroot;
        alfa = pn * (xnroot + ynroot + znroot) + xnroot * ynroot * znroot;
        alfa = alfa * alfa;
        beta = pn * (pn + lamda) * (pn + lamda);
        rc = mmg_ellint_RC(alfa, beta);
        if (rc != rc) return MMG_NAN;
        sigma += power4 * rc;
        power4 *= 0.25;
        xn = (xn + lamda) * 0.25;
        yn = (yn + lamda) * 0.25;
        zn = (zn + lamda) * 0.25;
        pn = (pn + lamda) * 0.25;
        if (++n == MMG_ELL_NMAX) return MMG_NAN;
    }
    ea = xndev * (yndev + zndev) + yndev * zndev;
    eb = xndev * yndev * zndev;
    ec = pndev * pndev;
    e2 = ea - 3.0 * ec;
    e3 = eb + 2.0 * pndev * (ea - ec);
    s1 = 1.0 + e2 * (-c1 + 0.75 * c3 * e2 - 1.5 * c4 * e3);
    s2 = eb * (0.5 * c2 + pndev * (-c3 - c3 + pndev * c4));
    s3 = pndev * ea * (c2 - pndev * c3) - c2 * pndev * ec;
    return 3.0 * sigma + power4 * (s1 + s2 + s3) / (mu * sqrt(mu));
}

MMG_FN double mmg_ellint_Kcomp(double k) {
    if (k * k >= 1.0) return MMG_NAN;
    if (k * k >= 1.0 - 1.4901161193847656e-08) {            /* Abramowitz + Stegun 17.3.34 */
        const double y = 1.0 - k * k;
        const double ta = 1.38629436112 + y * (0.09666344259 + y * 0.03590092383);
        const double tb = -log(y) * (0.5 + y * (0.12498593597 + y * 0.06880248576));
        return ta + tb;
    }
    return mmg_ellint_RF(0.0, 1.0 - k * k, 1.0);
}

MMG_FN double mmg_ellint_Ecomp(double k) {
    if (k * k >= 1.0) return MMG_NAN;
    if (k * k >= 1.0 - 1.4901161193847656e-08) {            /* Abramowitz + Stegun 17.3.36 */
        const double y = 1.0 - k * k;
        const double ta = 1.0 + y * (0.44325141463 + y * (0.06260601220 + 0.04757383546 * y));
        const double tb = -y * log(y) * (0.24998368310 + y * (0.09200180037 + 0.04069697526 * y));
        return ta + tb;
    } else {
        const double y = 1.0 - k * k;
        return mmg_ellint_RF(0.0, y, 1.0) - k * k / 3.0 * mmg_ellint_RD(0.0, y, 1.0);
    }
}

MMG_FN double mmg_ellint_Pcomp(double k, double n) {
    if (k * k >= 1.0) return MMG_NAN;
    { const double y = 1.0 - k * k; return mmg_ellint_RF(0.0, y, 1.0) - (n / 3.0) * mmg_ellint_RJ(0.0, y, 1.0, 1.0 + n); }
}

MMG_FN double mmg_ellint_Dcomp(double k) {
    if (k * k >= 1.0) return MMG_NAN;
    return (1.0 / 3.0) * mmg_ellint_RD(0.0, 1.0 - k * k, 1.0);
}

MMG_FN double mmg_ellint_F(double phi, double k) {
    const double nc = floor(phi / MMG_PI + 0.5);
    const double sin_phi = sin(phi - nc * MMG_PI), sin2_phi = sin_phi * sin_phi;
    double val = sin_phi * mmg_ellint_RF(1.0 - sin2_phi, 1.0 - k * k * sin2_phi, 1.0);
    if (nc != 0.0) val += 2 * nc * mmg_ellint_Kcomp(k);
    return val;
}

MMG_FN double mmg_ellint_E(double phi, double k) {
    const double nc = floor(phi / MMG_PI + 0.5);
    const double sin_phi = sin(phi - nc * MMG_PI), sin2_phi = sin_phi * sin_phi;
    const double x = 1.0 - sin2_phi, y = 1.0 - k * k * sin2_phi;
    if (x < MMG_DBL_EPSILON) {
        const double re = mmg_ellint_Ecomp(k);
        return 2 * nc * re + (sin_phi >= 0.0 ? 1.0 : -1.0) * re;       /* GSL_SIGN */
    } else {
        const double sin3_phi = sin2_phi * sin_phi;
        double val = sin_phi * mmg_ellint_RF(x, y, 1.0) - k * k / 3.0 * sin3_phi * mmg_ellint_RD(x, y, 1.0);
        if (nc != 0.0) val += 2 * nc * mmg_ellint_Ecomp(k);
        return val;
    }
}

MMG_FN double mmg_ellint_P(double phi, double k, double n) {
    const double nc = floor(phi / MMG_PI + 0.5);
    const double sin_phi = sin(phi - nc * MMG_PI), sin2_phi = sin_phi * sin_phi, sin3_phi = sin2_phi * sin_phi;
    const double x = 1.0 - sin2_phi, y = 1.0 - k * k * sin2_phi;
    double val = sin_phi * mmg_ellint_RF(x, y, 1.0) - n / 3.0 * sin3_phi * mmg_ellint_RJ(x, y, 1.0, 1.0 + n * sin2_phi);
    if (nc != 0.0) val += 2 * nc * mmg_ellint_Pcomp(k, n);
    return val;
}

MMG_FN double mmg_ellint_D(double phi, double k) {
    const double nc = floor(phi / MMG_PI + 0.5);
    const double sin_phi = sin(phi - nc * MMG_PI), sin2_phi = sin_phi * sin_phi, sin3_phi = sin2_phi * sin_phi;
    double val = sin3_phi / 3.0 * mmg_ellint_RD(1.0 - sin2_phi, 1.0 - k * k * sin2_phi, 1.0);
    if (nc != 0.0) val += 2 * nc * mmg_ellint_Dcomp(k);
    return val;
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
