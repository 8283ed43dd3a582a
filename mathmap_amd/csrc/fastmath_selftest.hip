// Device-side exhaustive check of the float-argument acos / asin (test scaffolding: linked into
// tests/libmathmap_hip_selftest.so, not into the product library).
//
// mm_fastmath.h has two layers for these functions: the table forms (mmf_acos_f32, mmf_asin_f32), which the host verifier
// compares with glibc for every float in [-1, 1], and the fast forms the kernels call (mmf_acos_fast_f32,
// mmf_asin_fast_f32): the platform's double function, falling back to the table form next to a float rounding tie.  The
// platform function on the device is OCML's, which no host run can exercise -- so the device enumerates every float in
// [-1, 1] (and the first floats beyond) and counts where the two layers disagree.  0 means: on this GPU the fast forms
// return glibc's (float)acos((double)x) / (float)asin((double)x) for every argument.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#define MMF_FN static __device__ __forceinline__
#define MMF_COLD_FN static __device__ __attribute__((noinline))
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
#define MMF_ACOS_SLOW(a) acos((a))
#define MMF_ASIN_SLOW(a) asin((a))
#define MMF_SQRT(a) __builtin_sqrt((a))
#define MMF_HYPOT_SLOW(a, b) hypot((a), (b))
#include "mm_fastmath.h"

namespace {

struct Counts { unsigned long long checked, bad_acos, bad_asin, fallbacks; unsigned first_bad_acos, first_bad_asin; };

__global__ void __launch_bounds__(256) k_check(Counts *out, unsigned limit_bits) {
    unsigned long long bad_a = 0, bad_s = 0, n = 0, fb = 0;
    for (unsigned long long u = (unsigned long long)blockIdx.x * 256 + threadIdx.x; u <= limit_bits; u += (unsigned long long)gridDim.x * 256)
        for (unsigned sign = 0; sign < 2; ++sign) {
            const unsigned bits = (unsigned)u | (sign << 31);
            const float x = __uint_as_float(bits);
            const float fa = mmf_acos_fast_f32(x), ta = mmf_acos_f32(x), fs = mmf_asin_fast_f32(x), ts = mmf_asin_f32(x);
            const bool na = fa != fa && ta != ta, ns = fs != fs && ts != ts;         // NaN on both sides (|x| > 1)
            if (__float_as_uint(fa) != __float_as_uint(ta) && !na) { if (!bad_a) atomicCAS(&out->first_bad_acos, 0u, bits); ++bad_a; }
            if (__float_as_uint(fs) != __float_as_uint(ts) && !ns) { if (!bad_s) atomicCAS(&out->first_bad_asin, 0u, bits); ++bad_s; }
            fb += mmf_near_float_tie(acos((double)x)) ? 1 : 0;
            ++n;
        }
    atomicAdd(&out->checked, n);
    if (bad_a) atomicAdd(&out->bad_acos, bad_a);
    if (bad_s) atomicAdd(&out->bad_asin, bad_s);
    if (fb) atomicAdd(&out->fallbacks, fb);
}

}  // namespace

// out[0..5] = values checked, acos mismatches, asin mismatches, arguments that took acos's fallback, first bad argument bits
// (acos, asin).  Returns 0 when the kernel ran.
extern "C" int mmhip_selftest_acos_asin_exhaustive(unsigned long long *out) {
    Counts *d = nullptr, h;
    if (hipMalloc((void **)&d, sizeof(Counts)) != hipSuccess) return -1;
    if (hipMemset(d, 0, sizeof(Counts)) != hipSuccess) return -1;
    k_check<<<256 * 64, 256>>>(d, 0x3f800010u);
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    if (hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return -3;
    (void)hipFree(d);
    out[0] = h.checked; out[1] = h.bad_acos; out[2] = h.bad_asin; out[3] = h.fallbacks; out[4] = h.first_bad_acos; out[5] = h.first_bad_asin;
    return 0;
}
