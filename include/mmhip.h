/*
 * mmhip.h -- C ABI of the MI355X-native MathMap pixel engine (libmathmap_hip.so).
 *
 * Standalone tier: compile a .mm filter, bind user values / input images, render
 * rows on the GPU.  Mirrors the life cycle of the reference's session API
 * (mathmap.h:274-299, mathmap_common.c):
 *
 *   mmhip_compile        ~ compile_mathmap        (mathmap_common.c:503-582)
 *   mmhip_invoke         ~ invoke_mathmap         (mathmap_common.c:746-795)
 *   mmhip_set_*          ~ -D name=value handling (mathmap_cmdline.c:756-796)
 *   mmhip_render         ~ invocation_new_frame + call_invocation_parallel_and_join
 *                          (mathmap_common.c:797-1018); one launch renders the row band
 *   mmhip_unload         ~ unload_mathmap / free_invocation
 *
 * Plain pointers and sizes only; no C++ or torch types cross this boundary.
 * Every function returning int returns 0 on success and a negative value on
 * error; mmhip_last_error() then holds the message (the reference's error_string,
 * exprtree.c:40).  The reference-ABI tier (gen_and_load_hip_code) is declared in
 * mathmap_hip_backend.h.
 */
#ifndef MMHIP_H
#define MMHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mmhip_filter mmhip_filter;
typedef struct mmhip_invocation mmhip_invocation;

/* user value kinds (userval.h:34-42) */
enum { MMHIP_UV_INT = 0, MMHIP_UV_FLOAT = 1, MMHIP_UV_BOOL = 2, MMHIP_UV_COLOR = 3,
       MMHIP_UV_CURVE = 4, MMHIP_UV_GRADIENT = 5, MMHIP_UV_IMAGE = 6 };

/* edge behaviours (mathmap.h:134-147) */
enum { MMHIP_EDGE_COLOR = 0, MMHIP_EDGE_WRAP = 1, MMHIP_EDGE_REFLECT = 2, MMHIP_EDGE_ROTATE = 3 };

typedef struct mmhip_options {
    int intersample;      /* 1 = bilinear input sampling (CLI -i), 0 = nearest */
    int supersampling;    /* affects the nearest fetch only (builtins.c:154-158) */
    int edge_behaviour_x, edge_behaviour_y;
    int tile_w;           /* workgroup tile width in pixels: 8,16,32,64,128,256 (0 = default) */
    int specialize_uservals; /* 1 = JIT a kernel variant per set of scalar user values with the values baked
                                in as literals and the reference's literal folds applied (x*0 -> 0, x+0 -> x,
                                dead branches); off by default */
    int pixel_inc;        /* drawable_get_pixel_inc (mathmap.c:1320-1327): the stride of the preview's image source
                             (fast_image_source_scale) the bilinear fetch interpolates over (builtins.c:186-216);
                             0 or 1 = full-resolution sources, the CLI's and every final render's case */
    int reserved[6];
} mmhip_options;

typedef struct mmhip_userval_info {
    int kind;
    int index;
    char name[64];
    int int_min, int_max, int_default;
    float float_min, float_max, float_default;
    int bool_default;
    unsigned image_flags;
} mmhip_userval_info;

const char *mmhip_last_error(void);
const char *mmhip_version(void);

/* ---- compile (no GPU needed up to mmhip_filter_load) ---- */
void mmhip_default_options(mmhip_options *o);
mmhip_filter *mmhip_compile(const char *source, const mmhip_options *opts);
/* like mmhip_compile, with n scalar user values (index, value) baked in as literals */
mmhip_filter *mmhip_compile_specialized(const char *source, const mmhip_options *opts, int n, const int *indices,
                                        const double *values);
/* the variant of a compiled filter (from source text or from an IR dump) with n scalar user values baked in:
   what a render with those values runs when the filter was compiled with specialize_uservals */
mmhip_filter *mmhip_filter_specialized(const mmhip_filter *f, int n, const int *indices, const double *values);
/* builds a filter from an IR dump (the JSON of mmhip_filter_ir_json): IR-level entry point */
mmhip_filter *mmhip_compile_ir_json(const char *ir_json, const mmhip_options *opts);
void mmhip_filter_free(mmhip_filter *f);
const char *mmhip_filter_name(const mmhip_filter *f);
int mmhip_filter_num_uservals(const mmhip_filter *f);
int mmhip_filter_userval_info(const mmhip_filter *f, int index, mmhip_userval_info *out);
const char *mmhip_filter_ir_json(mmhip_filter *f);        /* IR dump after the optimisation passes */
/* IR dump straight out of lowering (or the reference-ABI importer), before constant specialisation,
   copy propagation / DCE, loop-carried CSE and frame-constant hoisting: the input of
   mmhip_compile_ir_json and of the test oracle (oracle/ccgen.py) */
const char *mmhip_filter_ir_json_raw(mmhip_filter *f);
const char *mmhip_filter_kernel_source(mmhip_filter *f);  /* the HIP C++ handed to hiprtc */
int mmhip_filter_num_native_calls(const mmhip_filter *f);
/* hiprtc-compiles for gfx950 and (if a device is present) loads the module.
   load_module = 0 only compiles (usable without a GPU).  Returns code size. */
long mmhip_filter_jit(mmhip_filter *f, int load_module);
double mmhip_filter_jit_seconds(const mmhip_filter *f);

/* ---- invocation ---- */
mmhip_invocation *mmhip_invoke(mmhip_filter *f, int img_width, int img_height);
void mmhip_invocation_free(mmhip_invocation *inv);
int mmhip_set_int(mmhip_invocation *inv, int index, int value);
int mmhip_set_float(mmhip_invocation *inv, int index, float value);
int mmhip_set_bool(mmhip_invocation *inv, int index, int value);
int mmhip_set_color(mmhip_invocation *inv, int index, float r, float g, float b, float a);
/* curve = 1024 samples of the transfer curve over [0,1]; gradient = 1024 packed 0xRRGGBBAA colours
   (USER_CURVE_POINTS / USER_GRADIENT_POINTS, userval.h:36-37).  Defaults: identity ramp, grey ramp. */
/* Row-striped frames with native-filter calls: allow native filters to fill only the rows a
   stripe render reads (+- margin rows, plus the filter's own halo).  -1 = whole map (default). */
int mmhip_set_native_row_margin(mmhip_invocation *inv, int margin);
int mmhip_set_curve(mmhip_invocation *inv, int index, const float *values1024);
int mmhip_set_gradient(mmhip_invocation *inv, int index, const uint32_t *rgba1024);
int mmhip_set_by_name(mmhip_invocation *inv, const char *name, const char *value);  /* -Dname=value */
/* Input image from host memory: channels = 3 (RGB8, alpha forced to 255 as
   mathmap_cmdline.c:183 does) or 4 (RGBA8).  Uploaded once, stays in HBM. */
int mmhip_set_image_host(mmhip_invocation *inv, int index, const uint8_t *pixels, int width, int height, int channels);
/* Input image already resident in HBM as packed 0xRRGGBBAA uint32 per pixel. */
int mmhip_set_image_device(mmhip_invocation *inv, int index, const void *device_rgba32, int width, int height);
int mmhip_set_edge_colors(mmhip_invocation *inv, uint32_t color_x, uint32_t color_y);
int mmhip_set_render_size(mmhip_invocation *inv, int render_width, int render_height);
/* sub-pixel sampling offset of the slice (mathmap.h:219; -0.5 for the second supersampling pass) */
int mmhip_set_sampling_offset(mmhip_invocation *inv, float offset_x, float offset_y);

/* Renders rows [first_row, last_row) of region (region_x, region_y, region_w, region_h)
   at animation parameter t / frame into device memory `out_device` (row 0 of the
   band at out_device; bpp bytes per pixel, row_stride bytes per row; floatmap != 0
   writes float[4] per pixel instead -- its rows are the *frame's* render width apart,
   16 * render_width bytes, whatever the region's width and row_stride, like the
   reference's float-map bands (new_template.c.in:297); the region's columns sit at the
   start of each row).  `stream` is a hipStream_t (NULL = the invocation's own stream).
   Asynchronous. */
int mmhip_render(mmhip_invocation *inv, int frame, float t, int region_x, int region_y, int region_w, int region_h,
                 int first_row, int last_row, void *out_device, int row_stride, int bpp, int floatmap, void *stream);
/* The CLI's -o: supersampled render of a region (two slices + 1-1-2-1-1 / 6 byte combine,
   call_invocation, mathmap_common.c:880-927).  Compile the filter with supersampling = 1. */
int mmhip_render_supersampled(mmhip_invocation *inv, int frame, float t, int region_x, int region_y, int region_w,
                              int region_h, void *out_device, int row_stride, int bpp, void *stream);
/* Convenience: whole frame to a host RGBA8 buffer (width*height*4 bytes); synchronous. */
int mmhip_render_host(mmhip_invocation *inv, int frame, float t, uint8_t *out_rgba);
int mmhip_sync(mmhip_invocation *inv);
/* Average device time (ms) of the last pixel-kernel launch as measured with HIP
   events on the launch stream; requires mmhip_enable_timing(inv, 1). */
int mmhip_enable_timing(mmhip_invocation *inv, int on);
double mmhip_last_kernel_ms(mmhip_invocation *inv);
/* Durations of the pixel kernel of all timed launches since the last drain (oldest first, waits
   for the last one): lets a caller queue many launches without a synchronisation per launch. */
int mmhip_drain_kernel_ms(mmhip_invocation *inv, double *out_ms, int cap);
/* Durations (ms) of the kernels native filters launched themselves (gaussian_blur's scan kernels) since the last
   drain, in launch order; names receives a 64-byte label per entry.  Requires mmhip_enable_timing(inv, 1). */
int mmhip_drain_native_kernel_ms(mmhip_invocation *inv, char *names, double *out_ms, int cap);
/* Launches of this invocation whose pixels a native filter wrote itself -- a filter like
   examples/Blur/Gaussian Blur.mm, whose pixel is the blurred map sampled at the pixel centre: the
   blur's last kernel packs the output (new_template.c.in:279-293) and the pixel kernel is skipped. */
long mmhip_direct_native_launches(mmhip_invocation *inv);

/* device memory helpers for callers without their own allocator */
void *mmhip_device_alloc(size_t bytes);
void mmhip_device_free(void *p);
int mmhip_copy_to_host(void *dst_host, const void *src_device, size_t bytes);
int mmhip_copy_to_device(void *dst_device, const void *src_host, size_t bytes);
int mmhip_device_count(void);
/* One process per GPU: selects the device (ordinal among the visible ones) the calling thread's later
   mmhip_* calls use -- invocations, their streams, modules and buffers live on the device that is
   current when they are created.  Returns 0, or -1 with mmhip_last_error(). */
int mmhip_set_device(int ordinal);

#ifdef __cplusplus
}
#endif
#endif
