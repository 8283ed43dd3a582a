// Microbenchmark: what bounds the blur's recurrence step?  The interior block of k_iir_causal
// (16 steps x (9 v_mul_f64 + 9 v_add_f64 + 1 cvt)) with its inputs (a) synthesised in registers,
// (b) loaded from a float map like the kernel does, at 1 and 2 waves per SIMD; plus v_add_f64 with
// two VGPR operands (the microbenchmark in f64_rate.hip only has the SGPR + VGPR form).
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/f64_iir_rate.hip -o /tmp/f64_iir_rate
#include <hip/hip_runtime.h>
#include <cstdio>

struct Coef { double n[5], d[5]; };

__device__ __forceinline__ double step(double s0, double s1, double s2, double s3, double s4, double v1, double v2, double v3,
                                       double v4, const double *n, const double *d) {
    double acc = 0.0;
    acc += n[0] * s0;
    acc += n[1] * s1 - d[1] * v1;
    acc += n[2] * s2 - d[2] * v2;
    acc += n[3] * s3 - d[3] * v3;
    acc += n[4] * s4 - d[4] * v4;
    return acc;
}

template <int MODE>
__global__ void __launch_bounds__(256) k_iir(const float *in, double *out, Coef c, int blocks16, long stride) {
    const long L = (long)blockIdx.x * 256 + threadIdx.x;
    double s1 = 0, s2 = 0, s3 = 0, s4 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0;
    float cur[16], nxt[16];
    const float *p = in + L % stride;
#pragma unroll
    for (int u = 0; u < 16; ++u) cur[u] = MODE == 0 ? (float)(threadIdx.x + u) * 1e-3f : p[(long)u * stride];
    for (int b = 0; b < blocks16; ++b) {
        if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 16; ++u) nxt[u] = p[((long)(b + 1) * 16 + u) * stride];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const double s0 = (double)cur[u];
            const double acc = step(s0, s1, s2, s3, s4, v1, v2, v3, v4, c.n, c.d);
            s4 = s3; s3 = s2; s2 = s1; s1 = s0;
            v4 = v3; v3 = v2; v2 = v1; v1 = acc;
        }
        if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 16; ++u) cur[u] = nxt[u];
        } else {
#pragma unroll
            for (int u = 0; u < 16; ++u) cur[u] += 1e-3f;
        }
    }
    out[L] = v1 + v2 + v3 + v4;
}

template <int CHAINS>
__global__ void k_addvv(double *out, int iters) {
    double x[CHAINS], y[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) { x[c] = threadIdx.x * 1e-3 + c; y[c] = threadIdx.x * 1e-9 + c * 1e-7; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) x[c] = x[c] + y[(c + r) % CHAINS];
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    const long stride = 16384L * 4;
    const int blocks16 = 1024;
    float *in;
    double *out;
    hipMalloc(&in, (size_t)stride * (blocks16 + 2) * 16 * sizeof(float));
    hipMemset(in, 0, (size_t)stride * (blocks16 + 2) * 16 * sizeof(float));
    hipMalloc(&out, (size_t)2048 * 256 * sizeof(double));      // the largest launch below: 2048 x 256 lanes
    Coef c;
    for (int i = 0; i < 5; ++i) { c.n[i] = 0.01 * (i + 1); c.d[i] = i == 0 ? 0.0 : (i & 1 ? -0.9 : 0.3) / i; }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms;
    for (int wps = 1; wps <= 2; ++wps) {
        for (int mode = 0; mode < 2; ++mode) {
            const int grid = 256 * wps;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) k_iir<0><<<grid, 256>>>(in, out, c, blocks16, stride);
                else k_iir<1><<<grid, 256>>>(in, out, c, blocks16, stride);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            hipEventElapsedTime(&ms, e0, e1);
            const double instr = (double)blocks16 * 16 * 19;
            printf("iir interior block, %s, %d wave(s)/SIMD: %.3f ms, %.2f cycles per f64 instruction per SIMD (2.4 GHz)\n",
                   mode ? "inputs from HBM (16384 lines)" : "inputs in registers", wps, ms, ms * 1e-3 * 2.4e9 / instr / wps);
        }
    }
    for (int rep = 0; rep < 2; ++rep) { hipEventRecord(e0); k_addvv<1><<<256, 256>>>(out, 20000); hipEventRecord(e1); hipEventSynchronize(e1); }
    hipEventElapsedTime(&ms, e0, e1);
    printf("v_add_f64 vgpr+vgpr dependent, 1 wave/SIMD: %.2f cycles\n", ms * 1e-3 * 2.4e9 / (20000.0 * 16));
    for (int rep = 0; rep < 2; ++rep) { hipEventRecord(e0); k_addvv<8><<<256, 256>>>(out, 20000); hipEventRecord(e1); hipEventSynchronize(e1); }
    hipEventElapsedTime(&ms, e0, e1);
    printf("v_add_f64 vgpr+vgpr 8 chains, 1 wave/SIMD: %.2f cycles\n", ms * 1e-3 * 2.4e9 / (20000.0 * 16 * 8));
    for (int rep = 0; rep < 2; ++rep) { hipEventRecord(e0); k_addvv<8><<<2048, 256>>>(out, 20000); hipEventRecord(e1); hipEventSynchronize(e1); }
    hipEventElapsedTime(&ms, e0, e1);
    printf("v_add_f64 vgpr+vgpr 8 chains, 8 waves/SIMD: %.2f cycles\n", ms * 1e-3 * 2.4e9 / (20000.0 * 16 * 8 * 8));
    return 0;
}
