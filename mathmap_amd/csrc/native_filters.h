// Native filters (the reference's native-filters/*.c) as hand-written HIP kernels.
// Called by the runtime between the prologue and the pixel kernel; inputs and
// outputs stay in HBM.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "mm_host_abi.h"

namespace mm {

// What render_image's pixel fetch reads from the invocation (builtins.c:120-159: get_orig_val_pixel
// is hard-wired there -- nearest, never bilinear -- but it still honours the invocation's
// supersampling flag, edge behaviours and edge colours).
struct NativeEnv {
    int supersampling = 0;
    int edge_x = 0, edge_y = 0;                 // 0 colour, 1 wrap, 2 reflect, 3 rotate
    uint32_t edge_color_x = 0, edge_color_y = 0;
};

// Scratch buffers reused across calls (sized for the largest map seen).
struct NativeWorkspace {
    NativeEnv env;                              // set by the runtime before every run_native_filter
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    void *reserve(size_t bytes);
    void release();
    // per-kernel timing of a native filter's own launches (mmhip_enable_timing): one HIP event pair per
    // launch on the launch stream, read back by mmhip_drain_native_kernel_ms
    struct Timed { const char *name; hipEvent_t a, b; };
    bool timing = false;
    std::vector<Timed> timed;                   // launches since the last drain, in launch order
    std::vector<std::pair<hipEvent_t, hipEvent_t>> timed_free;
    // brackets `launch()` with an event pair when timing is on
    template <class F> void timed_launch(const char *name, hipStream_t s, F &&launch) {
        if (!timing || timed.size() >= 4096) { launch(); return; }
        Timed t{name, nullptr, nullptr};
        if (!timed_free.empty()) { t.a = timed_free.back().first; t.b = timed_free.back().second; timed_free.pop_back(); }
        else if (hipEventCreate(&t.a) != hipSuccess || hipEventCreate(&t.b) != hipSuccess) { launch(); return; }
        (void)hipEventRecord(t.a, s);
        launch();
        (void)hipEventRecord(t.b, s);
        timed.push_back(t);
    }
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
// Optional direct output of a native filter (new_template.c.in:279-293 applied to the filter's own
// result): when the calling filter's pixel is just this result sampled at the pixel centre, the
// filter's last kernel also writes the RGBA8 pixels of rows [first_row, first_row + num_rows) x
// columns [region_x, region_x + region_w) and the pixel kernel is not launched.  `written` reports
// whether the path taken supports it (the recursive Gaussian does).
struct NativeDirectOut {
    void *out = nullptr;          // first requested row, first requested column
    int row_stride = 0;           // bytes
    int first_row = 0, num_rows = 0, region_x = 0, region_w = 0;
    bool skip_map = false;        // in: the caller will not read the float map (it is not memoised): do not write it
    bool written = false;         // out; with skip_map the map's contents are then undefined
};

int run_native_filter(const std::string &func, const HNativeRec &rec, const std::vector<HImageDesc> &images,
                      int render_w, int render_h, float *out_map, NativeWorkspace &ws, hipStream_t stream,
                      std::string *err, int *row_lo, int *row_hi,    // in: rows wanted, out: rows filled
                      NativeDirectOut *direct = nullptr);

// native_fft.hip: convolve / half_convolve / visualize_fft (native-filters/convolve.c)
int fft_native_filter(const std::string &func, const HNativeRec &rec, const std::vector<HImageDesc> &images, int render_w,
                      int render_h, float *out_map, NativeWorkspace &ws, hipStream_t stream, std::string *err);
int native_input_map(const char *who, const HImage &img, const std::vector<HImageDesc> &images, const NativeEnv &env, int w,
                     int h, float *dst, const float **map, hipStream_t s, std::string *err);
void fft_release_plans(NativeWorkspace &ws);
// render_image, drawable branch (builtins.c:303-343): `dst` = float[h][w][4]
void launch_render_drawable(const HImageDesc &in, const HImage &img, const NativeEnv &env, float *dst, int w, int h, hipStream_t s);

void launch_supersample_combine(const unsigned char *longs, const unsigned char *shorts, unsigned char *out, int w, int h,
                                int bpp, int out_stride, hipStream_t s);

}  // namespace mm
