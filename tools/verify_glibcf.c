/* Host check of mathmap_amd/csrc/mm_glibcf.h against the host's glibc (2.35), bit for bit:
 *   one-argument functions (expf logf sincosf atanf log1pf expm1f sinhf coshf): EVERY float;
 *   two-argument functions (atan2f hypotf) and the float-complex ones: a structured grid of special and
 *   near-special values (all pairs) plus N random pairs per thread drawn from several distributions.
 * NaN results compare equal to any NaN.  The device compiles the same text (IEEE float/double ops, fma,
 * correctly rounded division and square root), so a clean run here makes the HIP path return glibc's bits.
 *
 * build: gcc -O2 -mfma -ffp-contract=off -pthread tools/verify_glibcf.c -o /tmp/verify_glibcf -lm
 * usage: verify_glibcf [stride [pairs_per_thread]]     stride 1 = every float
 * Test infrastructure; not linked into the product.
 */
#define _GNU_SOURCE
#include <complex.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../mathmap_amd/csrc/mm_glibcf.h"

enum { NTHREADS = 8, NF1 = 16, NF2 = 24 };
static uint32_t stride = 1;
static uint64_t pairs = 20000000;

static inline uint64_t splitmix(uint64_t *s) {
    uint64_t z = (*s += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
static inline uint32_t bits_of(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float float_of(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline int same(float a, float b) { return (a != a && b != b) || bits_of(a) == bits_of(b); }

typedef struct { const char *name; uint64_t checked, bad; uint32_t first_a, first_b; } stat_t;
typedef struct { int tid; stat_t f1[NF1], f2[NF2]; } acc_t;

static const char *F1_NAMES[NF1] = {"expf", "logf", "sinf(sincosf)", "cosf(sincosf)", "atanf", "log1pf", "expm1f", "sinhf", "coshf"};

static void check1(stat_t *s, const char *name, uint32_t xb, float want, float got) {
    s->name = name;
    ++s->checked;
    if (!same(want, got)) { if (!s->bad) s->first_a = xb; ++s->bad; }
}
static void check2(stat_t *s, const char *name, uint32_t ab, uint32_t bb, float want_re, float want_im, float got_re, float got_im) {
    s->name = name;
    ++s->checked;
    if (!same(want_re, got_re) || !same(want_im, got_im)) { if (!s->bad) { s->first_a = ab; s->first_b = bb; } ++s->bad; }
}

static void one_arg(acc_t *a, uint32_t u) {
    const float x = float_of(u);
    float s, c;
    check1(&a->f1[0], F1_NAMES[0], u, expf(x), mmq_expf(x));
    check1(&a->f1[1], F1_NAMES[1], u, logf(x), mmq_logf(x));
    sincosf(x, &s, &c);
    const mmq_sc sc = mmq_sincosf(x);
    check1(&a->f1[2], F1_NAMES[2], u, s, sc.s);
    check1(&a->f1[3], F1_NAMES[3], u, c, sc.c);
    check1(&a->f1[4], F1_NAMES[4], u, atanf(x), mmq_atanf(x));
    check1(&a->f1[5], F1_NAMES[5], u, log1pf(x), mmq_log1pf(x));
#ifdef MMQ_HAVE_BATCH2
    check1(&a->f1[6], F1_NAMES[6], u, expm1f(x), mmq_expm1f(x));
    check1(&a->f1[7], F1_NAMES[7], u, sinhf(x), mmq_sinhf(x));
    check1(&a->f1[8], F1_NAMES[8], u, coshf(x), mmq_coshf(x));
#endif
}

static void two_arg(acc_t *a, float p, float q) {
    const uint32_t pb = bits_of(p), qb = bits_of(q);
    check2(&a->f2[0], "atan2f", pb, qb, atan2f(p, q), 0, mmq_atan2f(p, q), 0);
    check2(&a->f2[1], "hypotf", pb, qb, hypotf(p, q), 0, mmq_hypotf(p, q), 0);
    const float _Complex z = CMPLXF(p, q);
    const mmq_cf zz = mmq_cmake(p, q);
    float _Complex w;
    mmq_cf g;
    w = cexpf(z); g = mmq_cexpf(zz);
    check2(&a->f2[2], "cexpf", pb, qb, crealf(w), cimagf(w), g.re, g.im);
    w = clogf(z); g = mmq_clogf(zz);
    check2(&a->f2[3], "clogf", pb, qb, crealf(w), cimagf(w), g.re, g.im);
    {   /* cpowf with a few fixed exponents and with the pair itself as exponent of a fixed base */
        static const float ex[4][2] = {{1.3f, 0.4f}, {2.0f, 0.0f}, {-0.5f, 1.25f}, {0.0f, 1.0f}};
        for (int i = 0; i < 4; ++i) {
            w = cpowf(z, CMPLXF(ex[i][0], ex[i][1])); g = mmq_cpowf(zz, mmq_cmake(ex[i][0], ex[i][1]));
            check2(&a->f2[4], "cpowf(z, c)", pb, qb, crealf(w), cimagf(w), g.re, g.im);
        }
        w = cpowf(CMPLXF(0.25f, -0.75f), z); g = mmq_cpowf(mmq_cmake(0.25f, -0.75f), zz);
        check2(&a->f2[5], "cpowf(c, z)", pb, qb, crealf(w), cimagf(w), g.re, g.im);
    }
#ifdef MMQ_HAVE_BATCH2
    w = csqrtf(z); g = mmq_csqrtf(zz);
    check2(&a->f2[6], "csqrtf", pb, qb, crealf(w), cimagf(w), g.re, g.im);
    w = csinf(z); g = mmq_csinf(zz);
    check2(&a->f2[7], "csinf", pb, qb, crealf(w), cimagf(w), g.re, g.im);
    w = ccosf(z); g = mmq_ccosf(zz);
    check2(&a->f2[8], "ccosf", pb, qb, crealf(w), cimagf(w), g.re, g.im);
    w = ctanf(z); g = mmq_ctanf(zz);
    check2(&a->f2[9], "ctanf", pb, qb, crealf(w), cimagf(w), g.re, g.im);
    w = csinhf(z); g = mmq_csinhf(zz);
    check2(&a->f2[10], "csinhf", pb, qb, crealf(w), cimagf(w), g.re, g.im);
    w = ccoshf(z); g = mmq_ccoshf(zz);
    check2(&a->f2[11], "ccoshf", pb, qb, crealf(w), cimagf(w), g.re, g.im);
    w = ctanhf(z); g = mmq_ctanhf(zz);
    check2(&a->f2[12], "ctanhf", pb, qb, crealf(w), cimagf(w), g.re, g.im);
#endif
#ifdef MMQ_HAVE_BATCH3
    w = casinf(z); g = mmq_casinf(zz);
    check2(&a->f2[13], "casinf", pb, qb, crealf(w), cimagf(w), g.re, g.im);
    w = cacosf(z); g = mmq_cacosf(zz);
    check2(&a->f2[14], "cacosf", pb, qb, crealf(w), cimagf(w), g.re, g.im);
    w = catanf(z); g = mmq_catanf(zz);
    check2(&a->f2[15], "catanf", pb, qb, crealf(w), cimagf(w), g.re, g.im);
    w = casinhf(z); g = mmq_casinhf(zz);
    check2(&a->f2[16], "casinhf", pb, qb, crealf(w), cimagf(w), g.re, g.im);
    w = cacoshf(z); g = mmq_cacoshf(zz);
    check2(&a->f2[17], "cacoshf", pb, qb, crealf(w), cimagf(w), g.re, g.im);
    w = catanhf(z); g = mmq_catanhf(zz);
    check2(&a->f2[18], "catanhf", pb, qb, crealf(w), cimagf(w), g.re, g.im);
#endif
}

/* special and near-special values: every pair of them is tried */
static int special_values(float *v) {
    static const uint32_t base[] = {0x00000000, 0x00000001, 0x00000002, 0x007fffff, 0x00800000, 0x00800001, 0x00ffffff,
                                    0x01000000, 0x0c000000, 0x24800000, 0x31000000, 0x33ffffff, 0x34000000, 0x38800000,
                                    0x39800000, 0x3a000000, 0x3c23d70a, 0x3e800000, 0x3ee00000, 0x3effffff, 0x3f000000, 0x3f000001,
                                    0x3f300000, 0x3f3504f3, 0x3f490fdb, 0x3f7fffff, 0x3f800000, 0x3f800001, 0x3f980000, 0x3fb504f3,
                                    0x3fc90fdb, 0x3fffffff, 0x40000000, 0x401c0000, 0x40490fdb, 0x40c90fdb, 0x41200000, 0x42b00000,
                                    0x42b17217, 0x42b17218, 0x42b2d4fc, 0x42cff1b5, 0x42efffff, 0x42f00000, 0x43000000, 0x43800000,
                                    0x44000000, 0x47000000, 0x4b800000, 0x4c000000, 0x4f000000, 0x5a000000, 0x5f000000, 0x7e800000,
                                    0x7effffff, 0x7f000000, 0x7f7fffff, 0x7f800000, 0x7fc00000};
    int n = 0;
    for (unsigned i = 0; i < sizeof base / sizeof base[0]; ++i) {
        v[n++] = float_of(base[i]);
        v[n++] = float_of(base[i] | 0x80000000u);
    }
    return n;
}

static float random_float(uint64_t *seed, int kind) {
    const uint64_t r = splitmix(seed);
    switch (kind) {
        case 0: return float_of((uint32_t)r);                                              /* any bit pattern */
        case 1: return (float)((double)(int64_t)(r >> 11) * 0x1p-53 * 16.0 - 8.0);           /* uniform in [-8, 8) */
        case 2: return (float)((double)(r >> 11) * 0x1p-53 * 4.0 - 2.0);                     /* uniform in [-2, 2) */
        case 3: { float f = float_of(0x3f000000u + (uint32_t)(r & 0x00ffffff)); return (r >> 63) ? -f : f; }   /* |x| in [0.5, 2) */
        case 4: { float f = float_of(0x30000000u + (uint32_t)(r % 0x20000000u)); return (r >> 63) ? -f : f; }  /* 2^-31 .. 2^33 */
        default: return (float)((double)(r >> 11) * 0x1p-53 * 200.0 - 100.0);                /* uniform in [-100, 100) */
    }
}

static void *worker(void *p) {
    acc_t *a = p;
    for (uint64_t u = (uint64_t)a->tid * stride; u <= 0xffffffffull; u += (uint64_t)NTHREADS * stride) one_arg(a, (uint32_t)u);
    static float sv[256];
    const int nsv = special_values(sv);
    for (int i = a->tid; i < nsv; i += NTHREADS)
        for (int j = 0; j < nsv; ++j) two_arg(a, sv[i], sv[j]);
    uint64_t seed = 0x1234567ull * (a->tid + 1);
    for (uint64_t n = 0; n < pairs; ++n) {
        const int k = (int)(n % 12);
        static const int kinds[12][2] = {{0, 0}, {1, 1}, {2, 2}, {3, 3}, {4, 4}, {5, 5}, {1, 5}, {5, 1}, {3, 2}, {2, 3}, {4, 1}, {0, 2}};
        float x = random_float(&seed, kinds[k][0]), y = random_float(&seed, kinds[k][1]);
        if (n % 97 == 0) y = sv[splitmix(&seed) % nsv];      /* one component special */
        if (n % 89 == 0) x = sv[splitmix(&seed) % nsv];
        two_arg(a, x, y);
    }
    return NULL;
}

int main(int argc, char **argv) {
    if (argc > 1) stride = (uint32_t)strtoul(argv[1], NULL, 0);
    if (argc > 2) pairs = strtoull(argv[2], NULL, 0);
    if (!stride) stride = 1;
    pthread_t th[NTHREADS];
    static acc_t acc[NTHREADS];
    for (int t = 0; t < NTHREADS; ++t) { acc[t].tid = t; pthread_create(&th[t], NULL, worker, &acc[t]); }
    for (int t = 0; t < NTHREADS; ++t) pthread_join(th[t], NULL);
    uint64_t total_bad = 0;
    printf("{\"glibc\": \"%s\", \"stride\": %u, \"pairs_per_thread\": %llu, \"functions\": {", "host libm", stride, (unsigned long long)pairs);
    int first = 1;
    for (int pass = 0; pass < 2; ++pass)
        for (int f = 0; f < (pass ? NF2 : NF1); ++f) {
            stat_t s = {0};
            for (int t = 0; t < NTHREADS; ++t) {
                const stat_t *q = pass ? &acc[t].f2[f] : &acc[t].f1[f];
                if (!q->name) continue;
                s.name = q->name;
                s.checked += q->checked;
                if (q->bad && !s.bad) { s.first_a = q->first_a; s.first_b = q->first_b; }
                s.bad += q->bad;
            }
            if (!s.name) continue;
            printf("%s\"%s\": {\"checked\": %llu, \"mismatches\": %llu", first ? "" : ", ", s.name, (unsigned long long)s.checked,
                   (unsigned long long)s.bad);
            if (s.bad) printf(", \"first\": [\"0x%08x\", \"0x%08x\"]", s.first_a, s.first_b);
            printf("}");
            first = 0;
            total_bad += s.bad;
        }
    printf("}, \"total_mismatches\": %llu}\n", (unsigned long long)total_bad);
    return total_bad ? 1 : 0;
}
