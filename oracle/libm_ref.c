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
