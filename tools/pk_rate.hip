// Microbenchmark: issue rate of v_pk_add_f32 / v_pk_mul_f32 against v_add_f32 / v_mul_f32 (no FMA
// contraction), dependent chains, 8 waves per SIMD -- what bounds hipgen.cpp's pair mode.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/pk_rate.hip -o /tmp/pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int CHAINS>
__global__ void k_scalar(float *out, float a, float b, int iters) {
    float x[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x * 1e-3f + c;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) { x[c] = x[c] * b; x[c] = x[c] + a; }
    }
    float s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CHAINS>
__global__ void k_packed(float *out, float a, float b, int iters) {
    f2 x[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) x[c] = f2{threadIdx.x * 1e-3f + c, threadIdx.x * 2e-3f + c};
    const f2 va = {a, a * 1.5f}, vb = {b, b * 1.0000001f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) { x[c] = x[c] * vb; x[c] = x[c] + va; }
    }
    f2 s = {0, 0};
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

// Which instruction classes share an issue slot?  Four independent chains per wave, 8 waves per SIMD.
template <int OP>
__global__ void k_class(float *out, float a, float b, int iters) {
    float x[4];
    int n[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { x[c] = threadIdx.x * 1e-3f + c; n[c] = threadIdx.x + c; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (OP == 0) { x[c] = __builtin_fmaf(x[c], b, a); x[c] = __builtin_fmaf(x[c], b, a); }          // v_fma_f32
                if (OP == 1) { n[c] = n[c] + (int)threadIdx.x; n[c] = n[c] ^ i; }                                  // v_add_u32 / v_xor_b32
                if (OP == 2) { x[c] = __builtin_fmaxf(x[c], a); x[c] = __builtin_fminf(x[c], b); }                 // v_max_f32 / v_min_f32
                if (OP == 3) { n[c] = (int)x[c]; x[c] = (float)n[c] * b; }                                         // v_cvt_i32_f32, v_cvt_f32_i32 + v_mul_f32 (3 instructions)
                if (OP == 4) { x[c] = x[c] * b; n[c] = n[c] + (int)threadIdx.x; }                                  // v_mul_f32 next to v_add_u32
                if (OP == 5) { x[c] = x[c] > a ? x[c] * b : x[c] + a; }                                            // v_cmp + v_mul + v_add + v_cndmask (4 instructions)
            }
    }
    float s = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) s += x[c] + n[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float *out;
    hipMalloc(&out, (size_t)256 * 8 * 256 * sizeof(float));       // the largest launch: 2048 x 256 lanes
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    float ms;
#define RUN(KERNEL, CH, WPS, LABEL)                                                                              \
    for (int rep = 0; rep < 2; ++rep) {                                                                          \
        hipEventRecord(e0);                                                                                      \
        KERNEL<CH><<<256 * WPS, 256>>>(out, 1e-9f, 1.0000001f, iters);                                           \
        hipEventRecord(e1);                                                                                      \
        hipEventSynchronize(e1);                                                                                 \
    }                                                                                                            \
    hipEventElapsedTime(&ms, e0, e1);                                                                            \
    printf("%-28s chains=%d waves/SIMD=%d: %.2f cycles per wave-instruction per SIMD (2.4 GHz)\n", LABEL, CH, WPS, \
           ms * 1e-3 * 2.4e9 / ((double)iters * 16 * CH * WPS));
    RUN(k_scalar, 1, 1, "v_mul_f32 / v_add_f32")
    RUN(k_scalar, 2, 1, "v_mul_f32 / v_add_f32")
    RUN(k_scalar, 4, 1, "v_mul_f32 / v_add_f32")
    RUN(k_scalar, 8, 1, "v_mul_f32 / v_add_f32")
    RUN(k_scalar, 1, 2, "v_mul_f32 / v_add_f32")
    RUN(k_scalar, 1, 4, "v_mul_f32 / v_add_f32")
    RUN(k_scalar, 1, 8, "v_mul_f32 / v_add_f32")
    RUN(k_scalar, 2, 8, "v_mul_f32 / v_add_f32")
    RUN(k_scalar, 4, 8, "v_mul_f32 / v_add_f32")
    RUN(k_scalar, 2, 4, "v_mul_f32 / v_add_f32")
#define RUNC(OP, NINSTR, LABEL)                                                                                  \
    for (int rep = 0; rep < 2; ++rep) {                                                                          \
        hipEventRecord(e0);                                                                                      \
        k_class<OP><<<256 * 8, 256>>>(out, 1e-9f, 1.0000001f, iters);                                            \
        hipEventRecord(e1);                                                                                      \
        hipEventSynchronize(e1);                                                                                 \
    }                                                                                                            \
    hipEventElapsedTime(&ms, e0, e1);                                                                            \
    printf("%-44s 4 chains, 8 waves/SIMD: %.2f cycles per wave-instruction per SIMD (2.4 GHz)\n", LABEL,          \
           ms * 1e-3 * 2.4e9 / ((double)iters * 8 * 4 * NINSTR * 8));
    RUNC(0, 2, "v_fma_f32")
    RUNC(1, 2, "v_add_u32 / v_xor_b32")
    RUNC(2, 2, "v_max_f32 / v_min_f32")
    RUNC(3, 3, "v_cvt_i32_f32 / v_cvt_f32_i32 / v_mul_f32")
    RUNC(4, 2, "v_mul_f32 next to v_add_u32")
    RUNC(5, 4, "v_cmp_gt_f32 / v_mul / v_add / v_cndmask")
    RUN(k_packed, 1, 1, "v_pk_mul_f32 / v_pk_add_f32")
    RUN(k_packed, 4, 1, "v_pk_mul_f32 / v_pk_add_f32")
    RUN(k_packed, 1, 8, "v_pk_mul_f32 / v_pk_add_f32")
    RUN(k_packed, 4, 8, "v_pk_mul_f32 / v_pk_add_f32")
    return 0;
}
