// IR -> HIP C++ kernel string.  The MI355X counterpart of the reference's
// backends/cc.c (which fills new_template.c.in and shells out to gcc).
#pragma once
#include <string>
#include <vector>

#include "ir.h"

namespace mm {

struct KernelOptions {
    int intersample = 1;      // bilinear input fetch (CLI flag -i)
    int supersampling = 0;
    int edge_x = 0, edge_y = 0;
    int pixel_inc = 1;        // stride of the image sources the bilinear fetch interpolates over (the GIMP preview's fast source)
    int tile_w = 0;           // pixels per workgroup row (0: chosen from the body, auto_tile_w); tile_h = 256 / tile_w
    int unroll = 0;           // pixels evaluated back to back per work-item step; 0 = choose (hipgen.cpp auto_unroll)
    bool hoist = true;        // evaluate frame-constant code once per frame in a prologue kernel
    bool fast_math_exact = true;   // use f32 paths only where bit-identical to the double path
};

// One native-filter (or render) call found in the frame-constant code.
struct NativeCall {
    std::string func;             // "native_filter_gaussian_blur", "RENDER", ...
    std::vector<Ty> arg_types;
    int record_offset = 0;        // byte offset of its record in the frame-constant buffer
    bool in_loop = false;         // a call site inside a `while': its calls are numbered as they happen and recorded in
                                  // the dynamic entries (below); its own record stays unexecuted
    bool dynamic = false;         // one of the MM_NATIVE_DYN_CALLS entries behind the call sites: the n-th call made
                                  // from any in-loop site of this frame (the record's `index' names the site)
};

enum { MM_NATIVE_REC_BYTES = 320, MM_NATIVE_ARG_BYTES = 32, MM_NATIVE_MAX_ARGS = 9, MM_NATIVE_DYN_CALLS = 16 };

struct KernelSource {
    std::string source;           // full translation unit (prelude + 2 kernels)
    std::string prologue_name, pixel_name;
    int xy_bytes = 0;             // size of the frame-constant buffer
    bool has_prologue = false;
    std::vector<NativeCall> natives;      // the call sites, then -- if a site sits in a loop -- MM_NATIVE_DYN_CALLS dynamic entries
    int native_sites = 0;         // number of call sites among `natives'
    int native_ctr_offset = -1;   // int[4] in the frame-constant buffer: calls made so far (program order), dynamic
                                  // entries taken, "more in-loop calls than dynamic entries" flag
    int tile_w = 16, tile_h = 16;
    int unroll = 1;               // MM_UNROLL of the pixel kernel; the launch's rows per work-item is a multiple
    bool prologue_uses_time = true;   // frame-constant code reads t or frame: re-run it for every frame
    bool single_pixel = false;    // kernel renders exactly one pixel per work-item: launch with ppt = 1
    int row_values = 0;           // > 0: kernel `mm_rows` fills that many per-row values (mm_args.rowtab) before the pixel kernel
    std::string rows_name;
    int direct_native = -1;       // index into natives: the pixel is that result sampled at (x, y) and nothing else
    std::string key;              // cache key (hash of source)
};

// `functions_of`: the code whose `functions` (filter_$name bodies) `code` may call; null = its own
KernelSource generate_hip(FilterCode &code, const KernelOptions &opt, FilterCode *functions_of = nullptr);
const char *device_prelude();
const char *device_noise_prelude();   // mm_noise_device.h
const char *device_fastmath_prelude();   // mm_fastmath.h + tables
const char *noise_table_text();        // nullptr when the libnoise table was not available at build time

}  // namespace mm
