/*
 * mm_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's per-pixel runtime, used only by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker the HIP
 * path is compared with.  Nothing under mathmap_amd/ may include, link or call it.
 *
 * What it restates (reference file:line):
 *   op macros                opmacros.h:30-218, new_template.c.in:50-77
 *   coordinates / output     opmacros.h:156-157, new_template.c.in:208-312
 *   pixel fetch              builtins/builtins.c:40-265, color.h:36-54,
 *                            mathmap.c:1195-1209, mathmap_cmdline.c:131-184
 *   image scale values       userval.c:262-280, floatmap.c:30-46
 *   render_image             builtins/builtins.c:267-346
 *   gaussian_blur            native-filters/gauss.c:38-262,641-670
 *   convolve/half_convolve/visualize_fft   native-filters/convolve.c (mm_oracle_fft.c)
 *
 * Pinning: tests/test_oracle_golden.py checks it against the reference's own golden
 * PNGs (tests/golden/*.png, copied from the reference's tests/ directory).
 * The generated per-filter C (oracle/ccgen.py) is compiled exactly like the
 * reference's cc backend compiles its output: gcc -O2 -fPIC (Makefile:58), and
 * calls the host's glibc for every libm / complex function, as the reference does.
 */
#ifndef MM_ORACLE_H
#define MM_ORACLE_H

#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned int color_t;

typedef struct { float v[2]; } mmo_tup2;
typedef struct { float v[3]; } mmo_tup3;
typedef struct { float v[4]; } mmo_tup4;
typedef struct { float v[9]; } mmo_tup9;

enum { MMO_IMG_DRAWABLE = 0, MMO_IMG_FLOATMAP = 1, MMO_IMG_NULL = 2, MMO_IMG_CLOSURE = 3 };

/* an image table entry: 8-bit RGB input drawable or float map */
typedef struct {
    const void *data;     /* drawable: RGB8 rows, 3*(w*y+x) (mathmap_cmdline.c:182); floatmap: float[h][w][4] */
    int w, h;
    int kind;
    int num_frames;
    int channels;         /* 3 (alpha forced 255) or 4 */
    float scale_x, scale_y, middle_x, middle_y;
    float ax, bx, ay, by;
} mmo_image_desc;

/* image *value* (image_t* in the reference, incl. the IMAGE_RESIZE wrapper) */
typedef struct { int idx; int pw; int ph; float xf; float yf; int resized; } mmo_image;

typedef union { int i; float f; color_t c; int image; } mmo_userval;

typedef struct mmo_native_memo {
    int valid;
    int func;
    int in_idx, in2_idx;
    float a1, a2;
    float *map;
    int w, h;
} mmo_native_memo;

typedef struct {
    int img_width, img_height;
    int render_width, render_height;
    int frame_render_width, frame_render_height;
    float t;
    int frame;
    float R;
    int region_x, region_y, region_width, region_height;
    float sampling_offset_x, sampling_offset_y;
    int output_bpp;
    int row_stride;
    int floatmap;
    int intersample;
    int supersampling;
    int edge_behaviour_x, edge_behaviour_y;
    color_t edge_color_x, edge_color_y;
    const mmo_userval *uservals;
    mmo_image_desc *images;      /* table; native results are appended at native_slot_base */
    int num_images;
    int native_slot_base;
    mmo_native_memo *memo;       /* one per native slot */
    const float *curves;
    const color_t *gradients;
    int closure_base;            /* image-table slot of closure image #0 rendered for native filters, or -1 */
    int pixel_inc;               /* drawable_get_pixel_inc (mathmap.c:1320-1327): fast_image_source_scale while previewing, else 1 (0 = 1) */
    int memo_sites, memo_cap;    /* memo[0 .. memo_sites): one entry per native call site; [memo_sites .. memo_cap): further argument
                                    sets of sites that run more than once per frame (mmo_memo_slot) */
} mmo_args;

/* ---- op macros (opmacros.h) ---- */
#ifndef MIN
#define MIN(a, b) (((a) < (b)) ? (a) : (b))
#endif
#ifndef MAX
#define MAX(a, b) (((a) < (b)) ? (b) : (a))
#endif
#define NOP() (0.0)
#define INT2FLOAT(x) ((float)(x))
#define FLOAT2INT(x) ((int)(x))
#define INT2COMPLEX(x) ((float _Complex)(x))
#define FLOAT2COMPLEX(x) ((float _Complex)(x))
#define ADD(a, b) ((a) + (b))
#define SUB(a, b) ((a) - (b))
#define NEG(a) (-(a))
#define MUL(a, b) ((a) * (b))
#define DIV(a, b) ((float)(a) / (float)(b))
#define MOD(a, b) (fmod((a), (b)))
#define GAMMA(a) (((a) > 171.0) ? 0.0 : tgamma((a)))
#define EQ(a, b) ((a) == (b))
#define LESS(a, b) ((a) < (b))
#define LEQ(a, b) ((a) <= (b))
#define NOT(a) (!(a))
#define PRINT_FLOAT(a) (0)
#define NEWLINE() (0)
#define COMPLEX(r, i) ((r) + (i)*I)
#define CLAMP01(x) (MAX(0, MIN(1, (x))))

#define MAKE_RGBA_COLOR(r, g, b, a) \
    ((((color_t)(r)) << 24) | (((color_t)(g)) << 16) | (((color_t)(b)) << 8) | ((color_t)(a)))
#define RED(c) ((c) >> 24)
#define GREEN(c) (((c) >> 16) & 0xff)
#define BLUE(c) (((c) >> 8) & 0xff)
#define ALPHA(c) ((c)&0xff)
#define RED_FLOAT(c) (RED(c) / 255.0)
#define GREEN_FLOAT(c) (GREEN(c) / 255.0)
#define BLUE_FLOAT(c) (BLUE(c) / 255.0)
#define ALPHA_FLOAT(c) (ALPHA(c) / 255.0)
#define MAKE_COLOR(r, g, b, a) \
    (MAKE_RGBA_COLOR(CLAMP01((r)) * 255, CLAMP01((g)) * 255, CLAMP01((b)) * 255, CLAMP01((a)) * 255))

#define CALC_VIRTUAL_X(pxl, size, sampl_off) (((pxl) - ((size)-1) / 2.0 + (sampl_off)) / (((size)-1) / 2.0))
#define CALC_VIRTUAL_Y(pxl, size, sampl_off) ((-(pxl) + ((size)-1) / 2.0 - (sampl_off)) / (((size)-1) / 2.0))

#define TUPLE_NTH(t, n) ((t).v[(n)])
/* tree vectors (tree_vectors.c:89-150): a float array of static length, indices clamped, a write makes a new vector */
static inline float mmo_tv_nth(int i, const float *v, int n) { return v[i < 0 ? 0 : (i >= n ? n - 1 : i)]; }
static inline void mmo_tv_set(int i, float *v, int n, float x) { v[i < 0 ? 0 : (i >= n ? n - 1 : i)] = x; }

#define USERVAL_INT_ACCESS(n) (A->uservals[(n)].i)
#define USERVAL_FLOAT_ACCESS(n) (A->uservals[(n)].f)
#define USERVAL_BOOL_ACCESS(n) (A->uservals[(n)].i)
#define USERVAL_COLOR_ACCESS(n) (A->uservals[(n)].c)
#define USERVAL_CURVE_ACCESS(n) (A->uservals[(n)].i)
#define USERVAL_GRADIENT_ACCESS(n) (A->uservals[(n)].i)
#define USERVAL_IMAGE_ACCESS(n) (mmo_image_from_table(A, A->uservals[(n)].image))

#define IMAGE_PIXEL_WIDTH(i) ((i).pw)
#define IMAGE_PIXEL_HEIGHT(i) ((i).ph)
#define RESIZE_IMAGE(i, xf, yf) (mmo_resize_image((i), (xf), (yf)))
#define STRIP_RESIZE(i) (mmo_strip_resize((i)))
#define ORIG_VAL(x, y, i, f) (mmo_orig_val(A, (x), (y), (i), (f)))
#define UNINITED_IMAGE (mmo_null_image())

#define USER_CURVE_POINTS 1024
#define APPLY_CURVE(c, p) (A->curves[(c)*USER_CURVE_POINTS + (int)(CLAMP01((p)) * (USER_CURVE_POINTS - 1))])
#define APPLY_GRADIENT(g, p) \
    (mmo_tuple_from_color(A->gradients[(g)*USER_CURVE_POINTS + (int)(CLAMP01((p)) * (USER_CURVE_POINTS - 1))]))

mmo_image mmo_image_from_table(const mmo_args *A, int idx);
mmo_image mmo_null_image(void);
mmo_image mmo_closure_image(const mmo_args *A, int closure_id);
mmo_image mmo_resize_image(mmo_image i, float xf, float yf);
mmo_image mmo_strip_resize(mmo_image i);
mmo_tup4 mmo_tuple_from_color(color_t c);
mmo_tup4 mmo_orig_val(const mmo_args *A, float x, float y, mmo_image img, float frame);
color_t mmo_get_orig_val_pixel(const mmo_args *A, float x, float y, const mmo_image_desc *d, int frame);
color_t mmo_get_orig_val_intersample_pixel(const mmo_args *A, float x, float y, const mmo_image_desc *d, int frame);
mmo_tup4 mmo_get_floatmap_pixel(const mmo_image_desc *d, float x, float y);
void mmo_fill_drawable_desc(mmo_image_desc *d, const void *data, int w, int h, int channels);
void mmo_fill_floatmap_desc(mmo_image_desc *d, float *data, int w, int h);
void mmo_store_pixel(const mmo_args *A, unsigned char *p, float *fp, const float rt[4]);

/* native filters; `slot` = index of the call site (result lands in images[native_slot_base+slot]) */
mmo_image mmo_native_gaussian_blur(const mmo_args *A, int slot, mmo_image in, float hdev, float vdev);
mmo_image mmo_native_convolve(const mmo_args *A, int slot, mmo_image in, mmo_image filter, float normalize, float copy_alpha);
mmo_image mmo_native_half_convolve(const mmo_args *A, int slot, mmo_image in, mmo_image filter, float copy_alpha);
mmo_image mmo_native_visualize_fft(const mmo_args *A, int slot, mmo_image in, float ignore_alpha);
mmo_image mmo_render(const mmo_args *A, int slot, mmo_image in, int w, int h);
int mmo_memo_slot(const mmo_args *A, int site, int func, int in, int in2, float a1, float a2);
const mmo_image_desc *mmo_desc_of(const mmo_args *A, int idx);   /* &A->images[idx], or an unbound image's descriptor for a handle outside the table */
void mmo_render_image(const mmo_args *A, const mmo_image_desc *src, mmo_image srcv, int w, int h, float *out);
void mmo_gauss_iir(float *map, int width, int height, float hdev, float vdev);
void mmo_gauss_rle(float *map, int width, int height, float hdev, float vdev);
void mmo_find_iir_constants(double *n_p, double *n_m, double *d_p, double *d_m, double *bd_p, double *bd_m, float std_dev);
void mmo_free_memo(mmo_args *A, int nslots);
float _Complex cgamma(float _Complex z);

/* GSL / GLib operators: GSL and GLib are not in the image -- the same restatement of the published
 * algorithms the device uses (mathmap_amd/csrc/mm_gslmath.h is included as text: a checker of the
 * plumbing around them, not an independent oracle).  PARITY UNPINNED for these three. */
#include "../mathmap_amd/csrc/mm_gslmath.h"
static inline mmo_tup2 mmo_solve_linear_2(mmo_tup4 m, mmo_tup2 v) {
    double A4[4], x[2];
    int i;
    mmo_tup2 r;
    for (i = 0; i < 4; ++i) A4[i] = m.v[i];
    x[0] = v.v[0]; x[1] = v.v[1];
    mmg_hh_svx(2, A4, x);
    r.v[0] = x[0]; r.v[1] = x[1];
    return r;
}
static inline mmo_tup3 mmo_solve_linear_3(mmo_tup9 m, mmo_tup3 v) {
    double A9[9], x[3];
    int i;
    mmo_tup3 r;
    for (i = 0; i < 9; ++i) A9[i] = m.v[i];
    for (i = 0; i < 3; ++i) x[i] = v.v[i];
    mmg_hh_svx(3, A9, x);
    for (i = 0; i < 3; ++i) r.v[i] = x[i];
    return r;
}
static inline mmo_tup3 mmo_ell_jac(double u, double m) {
    double sn, cn, dn;
    mmo_tup3 r;
    mmg_elljac(u, m, &sn, &cn, &dn);
    r.v[0] = sn; r.v[1] = cn; r.v[2] = dn;
    return r;
}
#define SOLVE_LINEAR_2(m, v) (mmo_solve_linear_2((m), (v)))
#define SOLVE_LINEAR_3(m, v) (mmo_solve_linear_3((m), (v)))
#define ELL_JAC(u, m) (mmo_ell_jac((u), (m)))
#define ELL_INT_K_COMP(k) (mmg_ellint_Kcomp((k)))
#define ELL_INT_E_COMP(k) (mmg_ellint_Ecomp((k)))
#define ELL_INT_F(phi, k) (mmg_ellint_F((phi), (k)))
#define ELL_INT_E(phi, k) (mmg_ellint_E((phi), (k)))
#define ELL_INT_P(phi, k, n) (mmg_ellint_P((phi), (k), (n)))
#define ELL_INT_D(phi, k, n) (mmg_ellint_D((phi), (k)))
#define ELL_INT_RC(x, y) (mmg_ellint_RC((x), (y)))
#define ELL_INT_RD(x, y, z) (mmg_ellint_RD((x), (y), (z)))
#define ELL_INT_RF(x, y, z) (mmg_ellint_RF((x), (y), (z)))
#define ELL_INT_RJ(x, y, z, p) (mmg_ellint_RJ((x), (y), (z), (p)))
#define RAND(a, b) \
    (mmg_rand_unit(col + A->region_x, row + A->region_y, A->frame, mm_rand_ctr++) * ((double)(b) - (double)(a)) + (double)(a))
#define gsl_sf_beta(a, b) (exp(lgamma((a)) + lgamma((b)) - lgamma((a) + (b))))   /* GSL absent: parity unpinned */

#endif
