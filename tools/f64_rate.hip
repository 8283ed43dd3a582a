// Microbenchmark: issue rate and dependent latency of v_add_f64 / v_mul_f64 / v_fma_f64 on one
// wave per SIMD (the occupancy the blur's scan kernels run at) and on 8.
// hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/f64_rate.hip -o /tmp/f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE, int CHAINS>
__global__ void k(double *out, double a, double b, int iters) {
    double x[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x * 1e-3 + c;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if (MODE == 0) x[c] = x[c] + a;
                else if (MODE == 1) x[c] = x[c] * b;
                else x[c] = __builtin_fma(x[c], b, a);
            }
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int CHAINS>
void run(const char *name, int waves_per_simd, double *out) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int blocks = 256 * waves_per_simd;       // 256 CUs, 256 threads = 4 waves = 1 per SIMD
    k<MODE, CHAINS><<<blocks, 256>>>(out, 1e-9, 1.0000001, 10);
    hipEventRecord(e0);
    k<MODE, CHAINS><<<blocks, 256>>>(out, 1e-9, 1.0000001, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_wave = (double)iters * 16 * CHAINS;
    const double ns_per_instr = ms * 1e6 / instr_per_wave / waves_per_simd;
    printf("%-10s chains=%d waves/SIMD=%d: %.2f ns per wave-instruction per SIMD (%.1f cycles at 2.4 GHz)\n", name, CHAINS,
           waves_per_simd, ns_per_instr, ns_per_instr * 2.4);
}

int main() {
    double *out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(double));
    run<0, 1>("add dep", 1, out);
    run<1, 1>("mul dep", 1, out);
    run<2, 1>("fma dep", 1, out);
    run<0, 8>("add ilp8", 1, out);
    run<1, 8>("mul ilp8", 1, out);
    run<2, 8>("fma ilp8", 1, out);
    run<0, 2>("add ilp2", 1, out);
    run<0, 4>("add ilp4", 1, out);
    run<0, 1>("add dep", 8, out);
    run<0, 1>("add dep", 2, out);
    run<0, 1>("add dep", 4, out);
    run<0, 2>("add ilp2", 2, out);
    run<0, 2>("add ilp2", 4, out);
    run<0, 8>("add ilp8", 8, out);
    run<2, 8>("fma ilp8", 8, out);
    return 0;
}
