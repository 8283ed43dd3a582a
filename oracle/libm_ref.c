/* Test infrastructure (oracle side): glibc's (float)f((double)x) for runs of consecutive float bit patterns, the value
 * the reference's generated C computes for a real math op on a float (ops.lisp:126-147 print the double libm names,
 * the assignment to a float compvar rounds).  Used by tests/ and tools/libm_exceptions.py to compare the device's
 * functions with the host libm for EVERY float.  Not linked into the product.
 *
 * build: gcc -O2 -fPIC -shared -pthread -fno-builtin oracle/libm_ref.c -o oracle/_build/libmm_libm_ref.so -lm */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <string.h>

enum { OP_SIN, OP_COS, OP_TAN, OP_ASIN, OP_ACOS, OP_ATAN, OP_EXP, OP_LOG, OP_SINH, OP_COSH, OP_TANH, OP_ASINH, OP_ACOSH, OP_ATANH, OP_COUNT };

static float eval(int op, float x) {
    const double d = (double)x;
    switch (op) {
        case OP_SIN: return (float)sin(d);
        case OP_COS: return (float)cos(d);
        case OP_TAN: return (float)tan(d);
        case OP_ASIN: return (float)asin(d);
        case OP_ACOS: return (float)acos(d);
        case OP_ATAN: return (float)atan(d);
        case OP_EXP: return (float)exp(d);
        case OP_LOG: return (float)log(d);
        case OP_SINH: return (float)sinh(d);
        case OP_COSH: return (float)cosh(d);
        case OP_TANH: return (float)tanh(d);
        case OP_ASINH: return (float)asinh(d);
        case OP_ACOSH: return (float)acosh(d);
        case OP_ATANH: return (float)atanh(d);
        default: return 0.0f;
    }
}

typedef struct { int op; uint32_t first; uint64_t lo, hi; const uint32_t *got; uint64_t bad; uint32_t bad_x[64], bad_want[64]; } job_t;

static void *worker(void *p) {
    job_t *j = p;
    for (uint64_t i = j->lo; i < j->hi; ++i) {
        const uint32_t bits = j->first + (uint32_t)i;
        float x, w;
        memcpy(&x, &bits, 4);
        w = eval(j->op, x);
        uint32_t wb, gb = j->got[i];
        memcpy(&wb, &w, 4);
        float g;
        memcpy(&g, &gb, 4);
        if (wb != gb && !(w != w && g != g)) {       /* any NaN equals any NaN */
            if (j->bad < 64) { j->bad_x[j->bad] = bits; j->bad_want[j->bad] = wb; }
            ++j->bad;
        }
    }
    return 0;
}

/* Compares got[i] (the device's result bits for argument bits first + i) with glibc for i < count on `threads` threads.
 * Returns the number of mismatches; the first up to `cap` are written as (argument bits, glibc's result bits) pairs. */
uint64_t mmo_libm_compare(int op, uint32_t first, uint64_t count, const uint32_t *got, int threads, uint32_t *bad_pairs, int cap) {
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    pthread_t th[64];
    job_t jobs[64];
    for (int t = 0; t < threads; ++t) {
        memset(&jobs[t], 0, sizeof jobs[t]);
        jobs[t].op = op; jobs[t].first = first; jobs[t].got = got;
        jobs[t].lo = count * t / threads; jobs[t].hi = count * (t + 1) / threads;
        pthread_create(&th[t], 0, worker, &jobs[t]);
    }
    uint64_t bad = 0;
    int n = 0;
    for (int t = 0; t < threads; ++t) {
        pthread_join(th[t], 0);
        for (uint64_t k = 0; k < jobs[t].bad && k < 64 && n < cap; ++k, ++n) { bad_pairs[2 * n] = jobs[t].bad_x[k]; bad_pairs[2 * n + 1] = jobs[t].bad_want[k]; }
        bad += jobs[t].bad;
    }
    return bad;
}

/* ---- two-argument ops: pseudo-random pairs (the device generates the same ones from the same counter) ---- */
enum { OP2_ATAN2, OP2_HYPOT, OP2_POW, OP2_FMOD, OP2_COUNT };

static uint64_t splitmix(uint64_t z) {
    z += 0x9e3779b97f4a7c15ULL;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

/* pair number n of run `seed`: four mixes -- any two floats; both within a few binades of 1 (either sign); the second a
 * few to 30 binades below the first; small integers and halves */
void mmo_pair(uint64_t seed, uint64_t n, float *x, float *y) {
    const uint64_t z = splitmix(seed * 0x100000001b3ULL + n);
    uint32_t a, b;
    switch (n & 3) {
        case 0: a = (uint32_t)z; b = (uint32_t)(z >> 32); break;
        case 1: a = ((uint32_t)z & 0x80ffffffu) | ((124u + (uint32_t)((z >> 24) & 7)) << 23);
                b = ((uint32_t)(z >> 32) & 0x80ffffffu) | ((124u + (uint32_t)((z >> 56) & 7)) << 23); break;
        case 2: a = ((uint32_t)z & 0x807fffffu) | (127u << 23);
                b = ((uint32_t)(z >> 32) & 0x807fffffu) | ((127u - (uint32_t)((z >> 56) % 31)) << 23); break;
        default: { float fa = (float)((int)(z & 0x3ff) - 512) * 0.5f, fb = (float)((int)((z >> 32) & 0x3ff) - 512) * 0.5f;
                   memcpy(&a, &fa, 4); memcpy(&b, &fb, 4); }
    }
    memcpy(x, &a, 4);
    memcpy(y, &b, 4);
}

static float eval2(int op, float x, float y) {
    switch (op) {
        case OP2_ATAN2: return (float)atan2((double)x, (double)y);
        case OP2_HYPOT: return (float)hypot((double)x, (double)y);
        case OP2_POW: return (float)pow((double)x, (double)y);
        default: return (float)fmod((double)x, (double)y);
    }
}

typedef struct { int op; uint64_t seed, lo, hi; const uint32_t *got; uint64_t bad; uint32_t bad_x[16], bad_y[16]; } job2_t;

static void *worker2(void *p) {
    job2_t *j = p;
    for (uint64_t i = j->lo; i < j->hi; ++i) {
        float x, y, w, g;
        mmo_pair(j->seed, i, &x, &y);
        w = eval2(j->op, x, y);
        uint32_t wb, gb = j->got[i];
        memcpy(&wb, &w, 4);
        memcpy(&g, &gb, 4);
        if (wb != gb && !(w != w && g != g)) {
            if (j->bad < 16) { memcpy(&j->bad_x[j->bad], &x, 4); memcpy(&j->bad_y[j->bad], &y, 4); }
            ++j->bad;
        }
    }
    return 0;
}

/* got[i] = the device's result bits for pair i of run `seed`; returns the number of mismatches, the first few pairs in bad_pairs */
uint64_t mmo_libm_compare2(int op, uint64_t seed, uint64_t count, const uint32_t *got, int threads, uint32_t *bad_pairs, int cap) {
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    pthread_t th[64];
    job2_t jobs[64];
    for (int t = 0; t < threads; ++t) {
        memset(&jobs[t], 0, sizeof jobs[t]);
        jobs[t].op = op; jobs[t].seed = seed; jobs[t].got = got;
        jobs[t].lo = count * t / threads; jobs[t].hi = count * (t + 1) / threads;
        pthread_create(&th[t], 0, worker2, &jobs[t]);
    }
    uint64_t bad = 0;
    int n = 0;
    for (int t = 0; t < threads; ++t) {
        pthread_join(th[t], 0);
        for (uint64_t k = 0; k < jobs[t].bad && k < 16 && n < cap; ++k, ++n) { bad_pairs[2 * n] = jobs[t].bad_x[k]; bad_pairs[2 * n + 1] = jobs[t].bad_y[k]; }
        bad += jobs[t].bad;
    }
    return bad;
}
