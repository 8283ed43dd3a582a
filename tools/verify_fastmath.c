/* Exhaustive host check of mathmap_amd/csrc/mm_fastmath.h against glibc:
 *   for every float x with |x| < MMF_LIMIT (both signs, zeros, denormals):
 *       mmf_sin_f32(x) == (float)sin((double)x)   and   mmf_cos_f32(x) == (float)cos((double)x)
 * bit for bit.  The device compiles the same source (fma, rint, IEEE double), so a clean run
 * here makes the HIP path identical to the reference's (float)sin((double)x) on this glibc.
 *
 * build: gcc -O2 -mfma -ffp-contract=off -pthread tools/verify_fastmath.c -o /tmp/verify_fastmath -lm
 * usage: verify_fastmath [stride]     stride 1 = every float (about a minute on 8 cores)
 * Test infrastructure; not linked into the product.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../mathmap_amd/csrc/mm_fastmath.h"

enum { NTHREADS = 8 };
static uint32_t stride = 1;
static uint32_t limit_bits;
typedef struct { uint64_t checked, bad_sin, bad_cos, checked_exp, bad_exp, checked_log, bad_log, checked_pow, bad_pow;
                 uint64_t checked_hypot, bad_hypot;
                 uint32_t first_bad, first_bad_exp, first_bad_log, first_bad_pow_x, first_bad_pow_y;
                 uint32_t first_bad_hypot_x, first_bad_hypot_y; int tid; } acc_t;
static uint64_t pow_pairs = 0;      /* random (x, y) pairs per thread, argv[2] */

static inline uint64_t splitmix(uint64_t *s) {
    uint64_t z = (*s += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

static inline uint32_t bits_of(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float float_of(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static void *worker(void *p) {
    acc_t *a = p;
    for (uint64_t u = (uint64_t)a->tid * stride; u < limit_bits; u += (uint64_t)NTHREADS * stride) {
        for (int sign = 0; sign < 2; ++sign) {
            const float x = float_of((uint32_t)u | ((uint32_t)sign << 31));
            const float ws = (float)sin((double)x), wc = (float)cos((double)x);
            const float gs = mmf_sin_f32(x), gc = mmf_cos_f32(x);
            const mmf_sincos_t both = mmf_sincos_f32(x);         /* the fused pair must be the same two values */
            if (bits_of(both.s) != bits_of(gs)) ++a->bad_sin;
            if (bits_of(both.c) != bits_of(gc)) ++a->bad_cos;
            if (bits_of(ws) != bits_of(gs)) { if (!a->bad_sin && !a->bad_cos) a->first_bad = bits_of(x); ++a->bad_sin; }
            if (bits_of(wc) != bits_of(gc)) { if (!a->bad_sin && !a->bad_cos) a->first_bad = bits_of(x); ++a->bad_cos; }
            ++a->checked;
        }
    }
    /* exp: every float with |x| <= 700 (both signs); log: every positive finite float */
    const uint32_t exp_limit = bits_of(700.0f), log_limit = 0x7f800000u;
    for (uint64_t u = (uint64_t)a->tid * stride; u <= exp_limit; u += (uint64_t)NTHREADS * stride)
        for (int sign = 0; sign < 2; ++sign) {
            const float x = float_of((uint32_t)u | ((uint32_t)sign << 31));
            if (bits_of((float)exp((double)x)) != bits_of(mmf_exp_f32(x))) { if (!a->bad_exp) a->first_bad_exp = bits_of(x); ++a->bad_exp; }
            ++a->checked_exp;
        }
    for (uint64_t u = 1 + (uint64_t)a->tid * stride; u < log_limit; u += (uint64_t)NTHREADS * stride) {
        const float x = float_of((uint32_t)u);
        if (bits_of((float)log((double)x)) != bits_of(mmf_log_f32(x))) { if (!a->bad_log) a->first_bad_log = bits_of(x); ++a->bad_log; }
        ++a->checked_log;
    }
    /* hypot: two arguments cannot be enumerated -- random pairs in four mixes (any two finite floats; the same binade;
     * y a few to 30 binades below x, where x^2 + y^2 comes close to the square of a rounding boundary; small integers,
     * Pythagorean triples among them), the function restates glibc's own arithmetic */
    {
        uint64_t seed = 0x7654321ULL + (uint64_t)a->tid * 0x9e3779b9ULL;
        for (uint64_t n = 0; n < pow_pairs; ++n) {
            const uint64_t z = splitmix(&seed);
            float x, y;
            switch (n & 3) {
                case 0: x = float_of((uint32_t)(z & 0xffffffffu)); y = float_of((uint32_t)(z >> 32)); break;
                case 1: x = float_of(0x3f800000u | (uint32_t)(z & 0x7fffffu)); y = float_of(0x3f800000u | (uint32_t)((z >> 32) & 0x7fffffu)); break;
                case 2: x = float_of(0x3f800000u | (uint32_t)(z & 0x7fffffu));
                        y = float_of(((127u - (uint32_t)((z >> 56) % 31)) << 23) | (uint32_t)((z >> 24) & 0x7fffffu)); break;
                default: x = (float)(int)(z & 0xfff); y = (float)(int)((z >> 32) & 0xfff); break;
            }
            const float want = (float)hypot((double)x, (double)y), got = mmf_hypot_f32(x, y);
            if (bits_of(want) != bits_of(got) && !(want != want && got != got)) {
                if (!a->bad_hypot) { a->first_bad_hypot_x = bits_of(x); a->first_bad_hypot_y = bits_of(y); }
                ++a->bad_hypot;
            }
            ++a->checked_hypot;
        }
    }
    /* pow: two arguments cannot be enumerated -- random pairs in four mixes (any positive x with any y;
     * x in [1/4, 4] with |y| < 64; small integer and half-integer y; x within a few thousand ulps of 1 with
     * huge y) plus every small-integer power of a stride of bases */
    if (a->tid == 0)      /* every integer and eighth-integer base up to 4096 to every half-integer power in [-20, 20] */
        for (int xi = 1; xi <= 4096 * 8; ++xi)
            for (int yi = -40; yi <= 40; ++yi) {
                const float x = (float)xi / 8.0f, y = (float)yi * 0.5f;
                const float want = (float)pow((double)x, (double)y), got = mmf_pow_f32(x, y);
                if (bits_of(want) != bits_of(got)) {
                    if (!a->bad_pow) { a->first_bad_pow_x = bits_of(x); a->first_bad_pow_y = bits_of(y); }
                    ++a->bad_pow;
                }
                ++a->checked_pow;
            }
    uint64_t seed = 0x1234567ULL + (uint64_t)a->tid * 0x9e3779b9ULL;
    for (uint64_t n = 0; n < pow_pairs; ++n) {
        const uint64_t z = splitmix(&seed);
        float x, y;
        switch (n & 3) {
            case 0: x = float_of((uint32_t)(z & 0x7fffffffu) % 0x7f800000u); y = float_of((uint32_t)(z >> 32)); break;
            case 1: x = 0.25f + 3.75f * (float)((z >> 8) & 0xffffff) / 16777216.0f; y = ((float)((z >> 32) & 0xffffff) / 16777216.0f - 0.5f) * 128.0f; break;
            case 2: x = float_of((uint32_t)(z & 0x7fffffffu) % 0x7f800000u); y = (float)((int)((z >> 32) % 41) - 20) * 0.5f; break;
            default: x = float_of(0x3f800000u + (uint32_t)((z & 0x1fff)) - 4096u); y = ((float)((z >> 32) & 0xffffff) / 16777216.0f - 0.5f) * 4.0e6f; break;
        }
        if (!(x > 0.0f)) x = 1.5f;
        const float want = (float)pow((double)x, (double)y), got = mmf_pow_f32(x, y);
        if (bits_of(want) != bits_of(got) && !(want != want && got != got)) {
            if (!a->bad_pow) { a->first_bad_pow_x = bits_of(x); a->first_bad_pow_y = bits_of(y); }
            ++a->bad_pow;
        }
        ++a->checked_pow;
    }
    return NULL;
}

int main(int argc, char **argv) {
    if (argc > 1) stride = (uint32_t)strtoul(argv[1], NULL, 0);
    pow_pairs = argc > 2 ? strtoull(argv[2], NULL, 0) : 2000000ULL;
    if (stride == 0) stride = 1;
    limit_bits = bits_of(MMF_LIMIT);
    pthread_t th[NTHREADS];
    acc_t acc[NTHREADS];
    memset(acc, 0, sizeof acc);
    for (int i = 0; i < NTHREADS; ++i) { acc[i].tid = i; pthread_create(&th[i], NULL, worker, &acc[i]); }
    uint64_t checked = 0, bs = 0, bc = 0, ce = 0, be = 0, cl = 0, bl = 0, cp = 0, bp = 0;
    uint32_t first = 0, fe = 0, fl = 0, fpx = 0, fpy = 0;
    for (int i = 0; i < NTHREADS; ++i) {
        pthread_join(th[i], NULL);
        checked += acc[i].checked; bs += acc[i].bad_sin; bc += acc[i].bad_cos;
        ce += acc[i].checked_exp; be += acc[i].bad_exp; cl += acc[i].checked_log; bl += acc[i].bad_log;
        if (!first && (acc[i].bad_sin || acc[i].bad_cos)) first = acc[i].first_bad;
        if (!fe && acc[i].bad_exp) fe = acc[i].first_bad_exp;
        if (!fl && acc[i].bad_log) fl = acc[i].first_bad_log;
        cp += acc[i].checked_pow; bp += acc[i].bad_pow;
        if (!fpx && acc[i].bad_pow) { fpx = acc[i].first_bad_pow_x; fpy = acc[i].first_bad_pow_y; }
    }
    uint64_t ch = 0, bh = 0;
    uint32_t fhx = 0, fhy = 0;
    for (int i = 0; i < NTHREADS; ++i) {
        ch += acc[i].checked_hypot; bh += acc[i].bad_hypot;
        if (!fhx && !fhy && acc[i].bad_hypot) { fhx = acc[i].first_bad_hypot_x; fhy = acc[i].first_bad_hypot_y; }
    }
    printf("{");
    printf("\"hypot_random_pairs_checked\": %llu, \"hypot_mismatches\": %llu, \"hypot_first_bad_bits\": \"0x%08x 0x%08x\", ",
           (unsigned long long)ch, (unsigned long long)bh, fhx, fhy);
    printf("\"checked\": %llu, \"stride\": %u, \"sin_mismatches\": %llu, \"cos_mismatches\": %llu, \"first_bad_bits\": \"0x%08x\", "
           "\"exp_checked\": %llu, \"exp_mismatches\": %llu, \"exp_first_bad_bits\": \"0x%08x\", "
           "\"log_checked\": %llu, \"log_mismatches\": %llu, \"log_first_bad_bits\": \"0x%08x\", "
           "\"pow_random_pairs_checked\": %llu, \"pow_mismatches\": %llu, \"pow_first_bad_bits\": \"0x%08x 0x%08x\"}\n",
           (unsigned long long)checked, stride, (unsigned long long)bs, (unsigned long long)bc, first,
           (unsigned long long)ce, (unsigned long long)be, fe, (unsigned long long)cl, (unsigned long long)bl, fl,
           (unsigned long long)cp, (unsigned long long)bp, fpx, fpy);
    return (bs || bc || be || bl) ? 1 : 0;     /* pow / hypot mismatches are reported, not fatal: not enumerable */
}
