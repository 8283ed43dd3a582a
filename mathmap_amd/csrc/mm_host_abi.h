// Host mirrors of the POD structs the JIT kernels read (mm_device.h).  Field
// order and types must match the device definitions exactly; sizes are
// static_assert-ed below and re-checked against the device compiler in
// tests/test_codegen.py (offsetof dump kernel).
#pragma once
#include <cstdint>

namespace mm {

struct HImage { int idx; int pw; int ph; float xf; float yf; int resized; };

enum { IMG_DRAWABLE = 0, IMG_FLOATMAP = 1, IMG_NULL = 2 };

struct HImageDesc {
    const void *data;
    int w, h;
    int kind;
    int num_frames;
    float scale_x, scale_y, middle_x, middle_y;
    float ax, bx, ay, by;
};

union HUserval { int i; float f; uint32_t c; int image; };

struct HArgs {
    int img_width, img_height;
    int render_width, render_height;
    int frame_render_width, frame_render_height;
    float t;
    int frame;
    float R;
    int region_x, region_y, region_width, region_height;
    float sampling_offset_x, sampling_offset_y;
    int first_row, num_rows;
    int output_bpp;
    int row_stride;
    int floatmap;
    uint32_t edge_color_x, edge_color_y;
    const HUserval *uservals;
    const HImageDesc *images;
    const void *curves;
    const void *gradients;
    void *out;
    int native_slot_base;
    int ppt;
    float *xtab;
    float *ytab;
    uint32_t tiles_magic;     // ceil(2^32 / tiles_x) when workgroup id / tiles_x is exact by multiply-high, else 0
    uint32_t num_images;      // entries of `images`
    float *rowtab;            // per-row values of the launch (mm_rows), [value][row]
};

// workgroup id -> (tile row, tile column) is one division by the number of tile columns per work-item start-up; a
// multiply-high by ceil(2^32 / d) gives the same quotient for every n with n * d < 2^32 (the error of the product is
// below n / 2^32 < 1 / d, the smallest distance of n / d to the next integer)
static inline uint32_t tile_division_magic(long tiles_x, long nwg) {
    if (tiles_x < 2 || nwg * tiles_x >= (1L << 32)) return 0;
    return (uint32_t)(((1UL << 32) + (unsigned long)tiles_x - 1) / (unsigned long)tiles_x);
}

struct HNativeArg { int kind; int i; float f; HImage img; };
struct HNativeRec { int executed; int index; int nargs; int pad; HNativeArg args[4]; };

static_assert(sizeof(HImage) == 24, "mm_image layout");
static_assert(sizeof(HImageDesc) == 56, "mm_image_desc layout");
static_assert(sizeof(HNativeArg) == 36, "mm_narg_t layout");
static_assert(sizeof(HArgs) == 168, "mm_args layout");

}  // namespace mm
