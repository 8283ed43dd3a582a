// Native filters (the reference's native-filters/*.c) as hand-written HIP kernels.
// Called by the runtime between the prologue and the pixel kernel; inputs and
// outputs stay in HBM.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "mm_host_abi.h"

namespace mm {

// Scratch buffers reused across calls (sized for the largest map seen).
struct NativeWorkspace {
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    void *reserve(size_t bytes);
    void release();
    // cached hipFFT plans of the FFT native filters (native_fft.hip)
    void *fft_fwd = nullptr, *fft_inv = nullptr;
    int fft_w = 0, fft_h = 0, fft_batch = 0;
    bool fft_valid = false;
};

// Runs native filter `func` with the arguments recorded by the prologue kernel.
// `out_map` is a float[h][w][4] device buffer.  Returns 0 on success.
// `row_lo, row_hi`: the rows of the map the caller is going to read.  Filters whose output
// rows depend on a bounded neighbourhood of input rows (gaussian_blur: a halo of ceil(22.7 sigma)
// rows, below which the recurrences' start-up error is under 1e-17 of the value) may fill only
// those rows plus their halo; [0, render_h) asks for the whole map.
int run_native_filter(const std::string &func, const HNativeRec &rec, const std::vector<HImageDesc> &images,
                      int render_w, int render_h, float *out_map, NativeWorkspace &ws, hipStream_t stream,
                      std::string *err, int *row_lo, int *row_hi);   // in: rows wanted, out: rows filled

// native_fft.hip: convolve / half_convolve / visualize_fft (native-filters/convolve.c)
int fft_native_filter(const std::string &func, const HNativeRec &rec, const std::vector<HImageDesc> &images, int render_w,
                      int render_h, float *out_map, NativeWorkspace &ws, hipStream_t stream, std::string *err);
int native_input_map(const char *who, const HImage &img, const std::vector<HImageDesc> &images, int w, int h, float *dst,
                     const float **map, hipStream_t s, std::string *err);
void fft_release_plans(NativeWorkspace &ws);

void launch_supersample_combine(const unsigned char *longs, const unsigned char *shorts, unsigned char *out, int w, int h,
                                int bpp, int out_stride, hipStream_t s);

}  // namespace mm
