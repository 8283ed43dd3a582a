// Device side of the every-float check of the real one-argument math ops (test scaffolding: linked into
// tests/libmathmap_hip_selftest.so, not into the product library).  The kernel evaluates what a JIT kernel computes for
// `op` of a float -- the functions hipgen.cpp names -- on runs of consecutive float bit patterns; the caller compares with
// glibc's double function rounded to float (oracle/libm_ref.c, same op numbering).  No host run can exercise the platform's
// (OCML's) functions, which most of these ops are; this does, for every argument.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#define MMF_FN static __device__ __forceinline__
#define MMF_CONST_TABLE static __device__ const
#define MMF_FMA(a, b, c) __builtin_fma((a), (b), (c))
#define MMF_RINT(a) __builtin_rint((a))
#define MMF_FABSF(a) __builtin_fabsf((a))
#define MMF_FABS(a) __builtin_fabs((a))
#define MMF_SIN_SLOW(a) sin((a))
#define MMF_COS_SLOW(a) cos((a))
#define MMF_LDEXP(a, e) __builtin_ldexp((a), (e))
#define MMF_EXP_SLOW(a) exp((a))
#define MMF_LOG_SLOW(a) log((a))
#define MMF_POW_SLOW(a, b) pow((a), (b))
#define MMF_SQRT(a) __builtin_sqrt((a))
#define MMF_HYPOT_SLOW(a, b) hypot((a), (b))
#define MMF_ASINH_SLOW(a) asinh((a))
#define MMF_ACOSH_SLOW(a) acosh((a))
#include "mm_fastmath.h"

// ---- every real one-argument math op on a run of consecutive float bit patterns -------------------------------
// What a kernel computes for `op` of a float (the functions hipgen.cpp names), for arguments first, first + 1, ...; the
// caller compares with glibc (oracle/libm_ref.c, same op numbering).
namespace {
enum { OP_SIN, OP_COS, OP_TAN, OP_ASIN, OP_ACOS, OP_ATAN, OP_EXP, OP_LOG, OP_SINH, OP_COSH, OP_TANH, OP_ASINH, OP_ACOSH, OP_ATANH };

__device__ __forceinline__ float eval_unary(int op, float x) {
    switch (op) {
        case OP_SIN: return mmf_sin_f32(x);                     // table-driven below 2^22, the platform's beyond
        case OP_COS: return mmf_cos_f32(x);
        case OP_EXP: return mmf_exp_f32(x);
        case OP_LOG: return mmf_log_f32(x);
        case OP_ASINH: return mmf_asinh_f32(x);                 // the platform's + exception list
        case OP_ACOSH: return mmf_acosh_f32(x);
        case OP_TAN: return (float)tan((double)x);              // the platform's double function, as mm_device.h calls it
        case OP_ASIN: return (float)asin((double)x);
        case OP_ACOS: return (float)acos((double)x);
        case OP_ATAN: return (float)atan((double)x);
        case OP_SINH: return (float)sinh((double)x);
        case OP_COSH: return (float)cosh((double)x);
        case OP_TANH: return (float)tanh((double)x);
        case OP_ATANH: return (float)atanh((double)x);
        default: return 0.0f;
    }
}

__global__ void __launch_bounds__(256) k_eval_unary(int op, unsigned first, unsigned long long count, unsigned *out) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (unsigned long long)gridDim.x * 256)
        out[i] = __float_as_uint(eval_unary(op, __uint_as_float(first + (unsigned)i)));
}
}  // namespace

// Result bits of `op` for the `count` arguments whose bits are first, first + 1, ... into host memory.  Returns 0.
extern "C" int mmhip_selftest_eval_unary(int op, unsigned first, unsigned long long count, unsigned *out_host) {
    static unsigned *d = nullptr;
    static unsigned long long cap = 0;
    if (count > cap) {
        if (d) (void)hipFree(d);
        d = nullptr;
        if (hipMalloc((void **)&d, count * sizeof(unsigned)) != hipSuccess) { cap = 0; return -1; }
        cap = count;
    }
    k_eval_unary<<<4096, 256>>>(op, first, count, d);
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    if (hipMemcpy(out_host, d, count * sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess) return -3;
    return 0;
}

// ---- two-argument ops on pseudo-random pairs (oracle/libm_ref.c mmo_pair generates the same pairs) ------------------
namespace {
enum { OP2_ATAN2, OP2_HYPOT, OP2_POW, OP2_FMOD };

__device__ __forceinline__ unsigned long long splitmix_d(unsigned long long z) {
    z += 0x9e3779b97f4a7c15ULL;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

__device__ __forceinline__ void pair_d(unsigned long long seed, unsigned long long n, float *x, float *y) {
    const unsigned long long z = splitmix_d(seed * 0x100000001b3ULL + n);
    unsigned a, b;
    switch (n & 3) {
        case 0: a = (unsigned)z; b = (unsigned)(z >> 32); break;
        case 1: a = ((unsigned)z & 0x80ffffffu) | ((124u + (unsigned)((z >> 24) & 7)) << 23);
                b = ((unsigned)(z >> 32) & 0x80ffffffu) | ((124u + (unsigned)((z >> 56) & 7)) << 23); break;
        case 2: a = ((unsigned)z & 0x807fffffu) | (127u << 23);
                b = ((unsigned)(z >> 32) & 0x807fffffu) | ((127u - (unsigned)((z >> 56) % 31)) << 23); break;
        default: a = __float_as_uint((float)((int)(z & 0x3ff) - 512) * 0.5f); b = __float_as_uint((float)((int)((z >> 32) & 0x3ff) - 512) * 0.5f);
    }
    *x = __uint_as_float(a);
    *y = __uint_as_float(b);
}

__global__ void __launch_bounds__(256) k_eval_binary(int op, unsigned long long seed, unsigned long long count, unsigned *out) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (unsigned long long)gridDim.x * 256) {
        float x, y, r;
        pair_d(seed, i, &x, &y);
        switch (op) {
            case OP2_ATAN2: r = (float)atan2((double)x, (double)y); break;      // the platform's, as mm_device.h calls it
            case OP2_HYPOT: r = mmf_hypot_f32(x, y); break;
            case OP2_POW: r = mmf_pow_f32(x, y); break;
            default: r = (float)fmod((double)x, (double)y); break;
        }
        out[i] = __float_as_uint(r);
    }
}
}  // namespace

extern "C" int mmhip_selftest_eval_binary(int op, unsigned long long seed, unsigned long long count, unsigned *out_host) {
    unsigned *d = nullptr;
    if (hipMalloc((void **)&d, count * sizeof(unsigned)) != hipSuccess) return -1;
    k_eval_binary<<<4096, 256>>>(op, seed, count, d);
    int rc = 0;
    if (hipDeviceSynchronize() != hipSuccess) rc = -2;
    else if (hipMemcpy(out_host, d, count * sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess) rc = -3;
    (void)hipFree(d);
    return rc;
}
