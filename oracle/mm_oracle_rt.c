/*
 * mm_oracle_rt.c -- CPU ORACLE runtime (test infrastructure, NOT product code).
 * See mm_oracle.h for scope and the reference lines each function follows.
 */
#include "mm_oracle.h"

#include <assert.h>
#include <stdio.h>

/* ---- images --------------------------------------------------------------- */

mmo_image mmo_image_from_table(const mmo_args *A, int idx) {
    mmo_image im;
    im.idx = idx;
    im.pw = A->images[idx].w;
    im.ph = A->images[idx].h;
    im.xf = im.yf = 1.0f;
    im.resized = 0;
    return im;
}

mmo_image mmo_null_image(void) {
    mmo_image im = {-1, 0, 0, 1.0f, 1.0f, 0};
    return im;
}

/* ALLOC_CLOSURE_IMAGE + cc.c:189: a closure image has the canvas pixel size.  A closure that a native
 * filter takes as its input was rendered beforehand (render_image's closure branch, builtins.c:273-298:
 * the closure's own calc_lines over the frame, floatmap = 1, frame 0, t = 0) by the harness, which runs
 * the closure's code as a filter of its own (oracle/ccgen.py CpuFilter.subs); its float map sits in the
 * image table at closure_base + closure_id. */
mmo_image mmo_closure_image(const mmo_args *A, int closure_id) {
    mmo_image im = {-1, 0, 0, 1.0f, 1.0f, 0};
    im.pw = A->img_width;
    im.ph = A->img_height;
    /* (a render's own code evaluates the main filter's code once more: closures from this one on are not rendered for it --
     * their consumers are dead code there -- and read as no image) */
    if (closure_id >= 0 && A->closure_base >= 0 && A->closure_base + closure_id < A->num_images) im.idx = A->closure_base + closure_id;
    return im;
}

/* drawable.c:211-228 */
mmo_image mmo_resize_image(mmo_image i, float xf, float yf) {
    i.xf = xf;
    i.yf = yf;
    i.resized = 1;
    return i;
}

/* opmacros.h:197 */
mmo_image mmo_strip_resize(mmo_image i) {
    i.xf = i.yf = 1.0f;
    i.resized = 0;
    return i;
}

/* userval.c:262-280 */
void mmo_fill_drawable_desc(mmo_image_desc *d, const void *data, int w, int h, int channels) {
    memset(d, 0, sizeof *d);
    d->data = data;
    d->w = w;
    d->h = h;
    d->kind = MMO_IMG_DRAWABLE;
    d->num_frames = 1;
    d->channels = channels;
    d->scale_x = (w - 1) / 2.0;
    d->scale_y = (h - 1) / 2.0;
    d->middle_x = 1.0;
    d->middle_y = 1.0;
}

/* floatmap.c:30-46 */
void mmo_fill_floatmap_desc(mmo_image_desc *d, float *data, int w, int h) {
    memset(d, 0, sizeof *d);
    d->data = data;
    d->w = w;
    d->h = h;
    d->kind = MMO_IMG_FLOATMAP;
    d->num_frames = 1;
    d->ax = d->bx = (float)(w - 1) / 2.0;
    d->ay = d->by = (float)(h - 1) / 2.0;
    d->ay *= -1.0;
}

/* opmacros.h:176-181 */
mmo_tup4 mmo_tuple_from_color(color_t c) {
    mmo_tup4 t;
    t.v[0] = RED_FLOAT(c);
    t.v[1] = GREEN_FLOAT(c);
    t.v[2] = BLUE_FLOAT(c);
    t.v[3] = ALPHA_FLOAT(c);
    return t;
}

/* ---- pixel fetch ------------------------------------------------------------ */

/* builtins.c:40-119 */
static void apply_edge_behaviour(const mmo_args *A, int *_x, int *_y, int width, int height) {
    int x = *_x, y = *_y;
    switch (A->edge_behaviour_x) {
        case 1:
            if (x < 0) x = x % width + width;
            else if (x >= width) x %= width;
            break;
        case 2:
            if (x < 0) x = -x % width;
            else if (x >= width) x = (width - 1) - (x % width);
            break;
        case 3:
            if (x < 0) { x = -x % width; y = (height - 1) - y; }
            else if (x >= width) { x = (width - 1) - (x % width); y = (height - 1) - y; }
            break;
        default: break;
    }
    switch (A->edge_behaviour_y) {
        case 1:
            if (y < 0) y = y % height + height;
            else if (y >= height) y %= height;
            break;
        case 2:
            if (y < 0) y = -y % height;
            else if (y >= height) y = (height - 1) - (y % height);
            break;
        case 3:
            if (y < 0) { x = (width - 1) - x; y = -y % height; }
            else if (y >= height) { x = (width - 1) - x; y = (height - 1) - (y % height); }
            break;
        default: break;
    }
    *_x = x;
    *_y = y;
}

/* builtins.c:121-130 -> mathmap.c:1195-1209 -> mathmap_cmdline.c:131-184 */
static color_t get_pixel(const mmo_args *A, int x, int y, const mmo_image_desc *d, int frame) {
    const unsigned char *p;
    if (d == NULL || d->kind == MMO_IMG_NULL) return MAKE_RGBA_COLOR(255, 255, 255, 255);
    apply_edge_behaviour(A, &x, &y, d->w, d->h);
    if (x < 0 || x >= d->w) return A->edge_color_x;
    if (y < 0 || y >= d->h) return A->edge_color_y;
    if (frame < 0 || frame >= d->num_frames) return MAKE_RGBA_COLOR(255, 255, 255, 255);
    p = (const unsigned char *)d->data + (size_t)d->channels * ((size_t)d->w * y + x);
    return MAKE_RGBA_COLOR(p[0], p[1], p[2], d->channels == 4 ? p[3] : 255);
}

/* builtins.c:132-146 */
static void to_pixel_coords(const mmo_image_desc *d, float *x, float *y) {
    *x = (*x + d->middle_x) * d->scale_x;
    *y = -((*y - d->middle_y) * d->scale_y);
}

/* builtins.c:148-161 */
color_t mmo_get_orig_val_pixel(const mmo_args *A, float x, float y, const mmo_image_desc *d, int frame) {
    if (d != NULL && d->kind != MMO_IMG_NULL) to_pixel_coords(d, &x, &y);
    if (!A->supersampling) {
        x += 0.5;
        y += 0.5;
    }
    return get_pixel(A, floor(x), floor(y), d, frame);
}

/* builtins.c:163-245, color.h:48-54 */
color_t mmo_get_orig_val_intersample_pixel(const mmo_args *A, float x, float y, const mmo_image_desc *d, int frame) {
    int x1, x2, y1, y2;
    float x2fact, y2fact, x1fact, y1fact, p1fact, p2fact, p3fact, p4fact;
    color_t pixel1, pixel2, pixel3, pixel4;
    float r, g, b, a;
    int pixel_inc_x = A->pixel_inc > 1 ? A->pixel_inc : 1, pixel_inc_y = pixel_inc_x;     /* builtins.c:182-184 */
    if (d != NULL && d->kind != MMO_IMG_NULL) to_pixel_coords(d, &x, &y);
    if (pixel_inc_x > 1) {       /* builtins.c:186-194: the preview's strided source */
        x -= pixel_inc_x / 2.0;
        x1 = floor(x / pixel_inc_x) * pixel_inc_x;
        x2 = x1 + pixel_inc_x;
        x2fact = (x - x1) / pixel_inc_x;
    } else {
        x1 = floor(x);
        x2 = x1 + 1;
        x2fact = x - x1;
    }
    if (pixel_inc_y > 1) {       /* builtins.c:203-211 */
        y -= pixel_inc_y / 2.0;
        y1 = floor(y / pixel_inc_y) * pixel_inc_y;
        y2 = y1 + pixel_inc_y;
        y2fact = (y - y1) / pixel_inc_y;
    } else {
        y1 = floor(y);
        y2 = y1 + 1;
        y2fact = y - y1;
    }
    x1fact = 1.0 - x2fact;
    y1fact = 1.0 - y2fact;
    p1fact = x1fact * y1fact;
    p2fact = x1fact * y2fact;
    p3fact = x2fact * y1fact;
    p4fact = x2fact * y2fact;
    pixel1 = get_pixel(A, x1, y1, d, frame);
    pixel2 = get_pixel(A, x1, y2, d, frame);
    pixel3 = get_pixel(A, x2, y1, d, frame);
    pixel4 = get_pixel(A, x2, y2, d, frame);
    r = RED(pixel1) * p1fact;   g = GREEN(pixel1) * p1fact;   b = BLUE(pixel1) * p1fact;   a = ALPHA(pixel1) * p1fact;
    r = r + RED(pixel2) * p2fact; g = g + GREEN(pixel2) * p2fact; b = b + BLUE(pixel2) * p2fact; a = a + ALPHA(pixel2) * p2fact;
    r = r + RED(pixel3) * p3fact; g = g + GREEN(pixel3) * p3fact; b = b + BLUE(pixel3) * p3fact; a = a + ALPHA(pixel3) * p3fact;
    r = r + RED(pixel4) * p4fact; g = g + GREEN(pixel4) * p4fact; b = b + BLUE(pixel4) * p4fact; a = a + ALPHA(pixel4) * p4fact;
    return MAKE_RGBA_COLOR((color_t)rintf(r) & 0xff, (color_t)rintf(g) & 0xff, (color_t)rintf(b) & 0xff,
                           (color_t)rintf(a) & 0xff);
}

/* builtins.c:247-265 */
mmo_tup4 mmo_get_floatmap_pixel(const mmo_image_desc *d, float x, float y) {
    mmo_tup4 t = {{0.0f, 0.0f, 0.0f, 0.0f}};
    int ix = (int)lrintf(d->ax * x + d->bx);
    int iy = (int)lrintf(d->ay * y + d->by);
    if (ix < 0 || ix >= d->w || iy < 0 || iy >= d->h) return t;
    memcpy(t.v, (const float *)d->data + ((size_t)iy * d->w + ix) * 4, sizeof(float) * 4);
    return t;
}

/* opmacros.h:199-216 */
mmo_tup4 mmo_orig_val(const mmo_args *A, float x, float y, mmo_image img, float f) {
    const mmo_image_desc *d;
    if (img.resized) {
        x *= img.xf;
        y *= img.yf;
    }
    if (img.idx < 0) {
        mmo_tup4 t = {{1.0f, 1.0f, 1.0f, 1.0f}};
        return t;
    }
    d = &A->images[img.idx];
    if (d->kind == MMO_IMG_FLOATMAP) return mmo_get_floatmap_pixel(d, x, y);
    if (A->intersample) return mmo_tuple_from_color(mmo_get_orig_val_intersample_pixel(A, x, y, d, (int)f));
    return mmo_tuple_from_color(mmo_get_orig_val_pixel(A, x, y, d, (int)f));
}

/* new_template.c.in:272-293 */
void mmo_store_pixel(const mmo_args *A, unsigned char *p, float *fp, const float rt[4]) {
    int output_bpp = A->output_bpp;
    int is_bw = output_bpp == 1 || output_bpp == 2;
    int need_alpha = output_bpp == 2 || output_bpp == 4;
    int alpha_index = output_bpp - 1;
    if (A->floatmap) {
        int i;
        for (i = 0; i < 4; ++i) fp[i] = rt[i];
        return;
    }
    if (is_bw)
        p[0] = (CLAMP01(rt[0]) * 0.299 + CLAMP01(rt[1]) * 0.587 + CLAMP01(rt[2]) * 0.114) * 255.0;
    else {
        p[0] = CLAMP01(rt[0]) * 255.0;
        p[1] = CLAMP01(rt[1]) * 255.0;
        p[2] = CLAMP01(rt[2]) * 255.0;
    }
    if (need_alpha) p[alpha_index] = CLAMP01(rt[3]) * 255.0;
}

/* ---- render_image, drawable branch (builtins.c:303-343) ------------------------ */
void mmo_render_image(const mmo_args *A, const mmo_image_desc *src, mmo_image srcv, int width, int height, float *out) {
    mmo_image_desc fm;
    float ax, bx, ay, by;
    int x, y;
    float *p = out;
    mmo_args B = *A;
    B.intersample = 0; /* get_orig_val_pixel is hard-wired (builtins.c:306) */
    (void)src;
    mmo_fill_floatmap_desc(&fm, out, width, height);
    ax = fm.ax; bx = fm.bx; ay = fm.ay; by = fm.by;
    for (y = 0; y < height; ++y) {
        float fy = ((float)y - by) / ay;
        for (x = 0; x < width; ++x) {
            float fx = ((float)x - bx) / ax;
            mmo_tup4 t = mmo_orig_val(&B, fx, fy, srcv, 0.0);
            memcpy(p, t.v, sizeof(float) * 4);
            p += 4;
        }
    }
}

/* ---- gaussian blur (native-filters/gauss.c) -------------------------------------- */

/* gauss.c:38-115 */
void mmo_find_iir_constants(double *n_p, double *n_m, double *d_p, double *d_m, double *bd_p, double *bd_m, float std_dev) {
    int i;
    double x0, x1, x2, x3, x4, x5, x6, x7, div;
    div = sqrt(2 * M_PI) * std_dev;
    x0 = -1.783 / std_dev;
    x1 = -1.723 / std_dev;
    x2 = 0.6318 / std_dev;
    x3 = 1.997 / std_dev;
    x4 = 1.6803 / div;
    x5 = 3.735 / div;
    x6 = -0.6803 / div;
    x7 = -0.2598 / div;
    n_p[0] = x4 + x6;
    n_p[1] = (exp(x1) * (x7 * sin(x3) - (x6 + 2 * x4) * cos(x3)) + exp(x0) * (x5 * sin(x2) - (2 * x6 + x4) * cos(x2)));
    n_p[2] = (2 * exp(x0 + x1) * ((x4 + x6) * cos(x3) * cos(x2) - x5 * cos(x3) * sin(x2) - x7 * cos(x2) * sin(x3)) +
              x6 * exp(2 * x0) + x4 * exp(2 * x1));
    n_p[3] = (exp(x1 + 2 * x0) * (x7 * sin(x3) - x6 * cos(x3)) + exp(x0 + 2 * x1) * (x5 * sin(x2) - x4 * cos(x2)));
    n_p[4] = 0.0;
    d_p[0] = 0.0;
    d_p[1] = -2 * exp(x1) * cos(x3) - 2 * exp(x0) * cos(x2);
    d_p[2] = 4 * cos(x3) * cos(x2) * exp(x0 + x1) + exp(2 * x1) + exp(2 * x0);
    d_p[3] = -2 * cos(x2) * exp(x0 + 2 * x1) - 2 * cos(x3) * exp(x1 + 2 * x0);
    d_p[4] = exp(2 * x0 + 2 * x1);
    for (i = 0; i <= 4; i++) d_m[i] = d_p[i];
    n_m[0] = 0.0;
    for (i = 1; i <= 4; i++) n_m[i] = n_p[i] - d_p[i] * n_p[0];
    {
        double sum_n_p = 0.0, sum_n_m = 0.0, sum_d = 0.0, a, b;
        for (i = 0; i <= 4; i++) {
            sum_n_p += n_p[i];
            sum_n_m += n_m[i];
            sum_d += d_p[i];
        }
        a = sum_n_p / (1.0 + sum_d);
        b = sum_n_m / (1.0 + sum_d);
        for (i = 0; i <= 4; i++) {
            bd_p[i] = d_p[i] * a;
            bd_m[i] = d_m[i] * b;
        }
    }
}

/* one line of gauss.c:161-198 / :209-249; src/dest have `n` elements */
static void iir_line(const float *src, float *dest, int n, const double *n_p, const double *n_m, const double *d_p,
                     const double *d_m, const double *bd_p, const double *bd_m, double *val_p, double *val_m) {
    const float *sp_p = src, *sp_m = src + (n - 1);
    double *vp = val_p, *vm = val_m + (n - 1);
    float initial_p = sp_p[0], initial_m = sp_m[0];
    int k, i, j, terms;
    memset(val_p, 0, n * sizeof(double));
    memset(val_m, 0, n * sizeof(double));
    for (k = 0; k < n; k++) {
        terms = (k < 4) ? k : 4;
        for (i = 0; i <= terms; i++) {
            *vp += n_p[i] * sp_p[-i] - d_p[i] * vp[-i];
            *vm += n_m[i] * sp_m[i] - d_m[i] * vm[i];
        }
        for (j = i; j <= 4; j++) {
            *vp += (n_p[j] - bd_p[j]) * initial_p;
            *vm += (n_m[j] - bd_m[j]) * initial_m;
        }
        sp_p++;
        sp_m--;
        vp++;
        vm--;
    }
    for (k = 0; k < n; k++) dest[k] = val_p[k] + val_m[k];
}

/* gauss.c:126-262: vertical pass over all four channels, then horizontal; in place */
void mmo_gauss_iir(float *map, int width, int height, float hdev, float vdev) {
    double n_p[5], n_m[5], d_p[5], d_m[5], bd_p[5], bd_m[5];
    int m = MAX(width, height);
    double *val_p = malloc(m * sizeof(double)), *val_m = malloc(m * sizeof(double));
    float *src = malloc(m * sizeof(float)), *dest = malloc(m * sizeof(float));
    int channel, col, row, i;
    mmo_find_iir_constants(n_p, n_m, d_p, d_m, bd_p, bd_m, vdev);
    for (channel = 0; channel < 4; ++channel)
        for (col = 0; col < width; col++) {
            for (i = 0; i < height; ++i) src[i] = map[((size_t)i * width + col) * 4 + channel];
            iir_line(src, dest, height, n_p, n_m, d_p, d_m, bd_p, bd_m, val_p, val_m);
            for (i = 0; i < height; ++i) map[((size_t)i * width + col) * 4 + channel] = dest[i];
        }
    mmo_find_iir_constants(n_p, n_m, d_p, d_m, bd_p, bd_m, hdev);
    for (channel = 0; channel < 4; ++channel)
        for (row = 0; row < height; row++) {
            for (i = 0; i < width; ++i) src[i] = map[((size_t)row * width + i) * 4 + channel];
            iir_line(src, dest, width, n_p, n_m, d_p, d_m, bd_p, bd_m, val_p, val_m);
            for (i = 0; i < width; ++i) map[((size_t)row * width + i) * 4 + channel] = dest[i];
        }
    free(val_p);
    free(val_m);
    free(src);
    free(dest);
}

/* TEST SUPPORT for frames too large to blur whole in test time (BASELINE config 4: 16384 x 16384, a
 * 4.3 GB float map): rows `rows[0..nrows)` of mmo_gauss_iir(mmo_render_image(drawable)) -- the same
 * fetch (mmo_orig_val, nearest) and the same iir_line as above, organised so that the caller can
 * spread the columns of the vertical pass over threads and nothing but the sampled rows is kept.
 *   step 1, per column range: vertical pass (gauss.c:155-201) of columns [col_lo, col_hi), all four
 *           channels; the results of the sampled rows go to vmid[nrows][width][4];
 *   step 2: horizontal pass (gauss.c:203-252) of those rows, in place.
 * `img` is a width x height drawable of `channels` bytes per pixel bound to a `stretched image`
 * argument of a stretched filter (no resize factors), like examples/Blur/Gaussian Blur.mm. */
void mmo_gauss_rows_vertical(const unsigned char *img, int channels, int width, int height, float vdev_uv, int col_lo,
                             int col_hi, const int *rows, int nrows, float *vmid) {
    double n_p[5], n_m[5], d_p[5], d_m[5], bd_p[5], bd_m[5];
    double *val_p = malloc(height * sizeof(double)), *val_m = malloc(height * sizeof(double));
    float *src = malloc((size_t)height * 4 * sizeof(float)), *dest = malloc(height * sizeof(float));
    float *line = malloc(height * sizeof(float));
    mmo_args B;
    mmo_image_desc d, fm;
    mmo_image in;
    int col, channel, i, k;
    memset(&B, 0, sizeof B);
    mmo_fill_drawable_desc(&d, img, width, height, channels);
    B.images = &d;
    B.num_images = 1;
    in.idx = 0; in.pw = width; in.ph = height; in.xf = in.yf = 1.0f; in.resized = 0;
    mmo_fill_floatmap_desc(&fm, NULL, width, height);
    mmo_find_iir_constants(n_p, n_m, d_p, d_m, bd_p, bd_m, fabs(vdev_uv * fm.ay));      /* gauss.c:659-660 */
    for (col = col_lo; col < col_hi; ++col) {
        float fx = ((float)col - fm.bx) / fm.ax;                                        /* builtins.c:324-333 */
        for (i = 0; i < height; ++i) {
            float fy = ((float)i - fm.by) / fm.ay;
            mmo_tup4 t = mmo_orig_val(&B, fx, fy, in, 0.0);
            memcpy(src + (size_t)i * 4, t.v, 4 * sizeof(float));
        }
        for (channel = 0; channel < 4; ++channel) {
            for (i = 0; i < height; ++i) line[i] = src[(size_t)i * 4 + channel];
            iir_line(line, dest, height, n_p, n_m, d_p, d_m, bd_p, bd_m, val_p, val_m);
            for (k = 0; k < nrows; ++k) vmid[((size_t)k * width + col) * 4 + channel] = dest[rows[k]];
        }
    }
    free(val_p); free(val_m); free(src); free(dest); free(line);
}

void mmo_gauss_rows_horizontal(float *vmid, int width, int height, int nrows, float hdev_uv) {
    double n_p[5], n_m[5], d_p[5], d_m[5], bd_p[5], bd_m[5];
    double *val_p = malloc(width * sizeof(double)), *val_m = malloc(width * sizeof(double));
    float *src = malloc(width * sizeof(float)), *dest = malloc(width * sizeof(float));
    mmo_image_desc fm;
    int k, channel, i;
    mmo_fill_floatmap_desc(&fm, NULL, width, height);
    mmo_find_iir_constants(n_p, n_m, d_p, d_m, bd_p, bd_m, fabs(hdev_uv * fm.ax));
    for (channel = 0; channel < 4; ++channel)
        for (k = 0; k < nrows; ++k) {
            for (i = 0; i < width; ++i) src[i] = vmid[((size_t)k * width + i) * 4 + channel];
            iir_line(src, dest, width, n_p, n_m, d_p, d_m, bd_p, bd_m, val_p, val_m);
            for (i = 0; i < width; ++i) vmid[((size_t)k * width + i) * 4 + channel] = dest[i];
        }
    free(val_p); free(val_m); free(src); free(dest);
}

/* The descriptor behind an image handle.  A handle outside the table (the null image, idx -1: what a closure that was not
 * rendered for this code reads as -- mmo_closure_image) is an unbound image, not whatever lies next to the table. */
const mmo_image_desc *mmo_desc_of(const mmo_args *A, int idx) {
    static mmo_image_desc nul;      /* (written with the same bytes by every caller) */
    if (idx >= 0 && idx < A->num_images) return &A->images[idx];
    memset(&nul, 0, sizeof nul);
    nul.kind = MMO_IMG_NULL;
    return &nul;
}

/* The reference caches native-filter results per invocation under (filter, arguments) (native-filters/cache.c:110-156): every
 * distinct argument set has an image of its own, whichever call site asked.  Here: entry `site` for the first argument set a
 * call site sees; a site that runs again with other arguments (inside a loop -- each pixel runs the loop anew) finds or takes
 * one of the entries behind the sites'.  Returns the entry (= its image slot behind native_slot_base). */
int mmo_memo_slot(const mmo_args *A, int site, int func, int in, int in2, float a1, float a2) {
    int i;
    for (i = 0; i < A->memo_cap; ++i) {
        const mmo_native_memo *m = &A->memo[i];
        if (m->valid && m->func == func && m->in_idx == in && m->in2_idx == in2 && m->a1 == a1 && m->a2 == a2) return i;
    }
    if (A->memo_cap <= A->memo_sites || !A->memo[site].valid) return site;
    for (i = A->memo_sites; i < A->memo_cap; ++i)
        if (!A->memo[i].valid) return i;
    fprintf(stderr, "mm_oracle: more than %d native-filter results in one frame\n", A->memo_cap);
    abort();
}

/* native_filter_gaussian_blur, gauss.c:641-670, with the per-invocation memo of
 * native-filters/cache.c:110-156 (keyed by input image and the two float args). */
mmo_image mmo_native_gaussian_blur(const mmo_args *A, int slot, mmo_image in, float hdev, float vdev) {
    mmo_native_memo *m = &A->memo[slot = mmo_memo_slot(A, slot, 1, in.idx, 0, hdev, vdev)];
    mmo_image_desc *dst = &A->images[A->native_slot_base + slot];
    mmo_image out;
    int w = A->render_width, h = A->render_height;
    if (!(m->valid && m->func == 1 && m->in_idx == in.idx && m->a1 == hdev && m->a2 == vdev && m->w == w && m->h == h)) {
        const mmo_image_desc *src = mmo_desc_of(A, in.idx);
        float horizontal_std_dev, vertical_std_dev;
        if (m->map == NULL || m->w != w || m->h != h) {
            free(m->map);
            m->map = malloc((size_t)w * h * 4 * sizeof(float));
        }
        if (src->kind == MMO_IMG_FLOATMAP) {
            w = src->w;
            h = src->h;
            memcpy(m->map, src->data, (size_t)w * h * 4 * sizeof(float));
        } else
            mmo_render_image(A, src, in, w, h, m->map);
        mmo_fill_floatmap_desc(dst, m->map, w, h);
        horizontal_std_dev = fabs(hdev * dst->ax);
        vertical_std_dev = fabs(vdev * dst->ay);
        if (horizontal_std_dev < 0.5 || vertical_std_dev < 0.5)
            mmo_gauss_rle(m->map, w, h, horizontal_std_dev, vertical_std_dev);
        else
            mmo_gauss_iir(m->map, w, h, horizontal_std_dev, vertical_std_dev);
        m->valid = 1;
        m->func = 1;
        m->in_idx = in.idx;
        m->in2_idx = 0;
        m->a1 = hdev;
        m->a2 = vdev;
        m->w = w;
        m->h = h;
    }
    out.idx = A->native_slot_base + slot;
    out.pw = dst->w;
    out.ph = dst->h;
    out.xf = out.yf = 1.0f;
    out.resized = 0;
    return out;
}

/* RENDER op on a drawable (builtins.c:267-346, force = 0) */
mmo_image mmo_render(const mmo_args *A, int slot, mmo_image in, int w, int h) {
    mmo_native_memo *m;
    mmo_image_desc *dst;
    mmo_image out;
    const mmo_image_desc *src = mmo_desc_of(A, in.idx);
    /* a closure the harness rendered beforehand: render_image made a new, plain float map of it
     * (no resize wrapper on the result, builtins.c:270-271,345) */
    if (src->kind == MMO_IMG_FLOATMAP && A->closure_base >= 0 && in.idx >= A->closure_base) {
        in.pw = src->w;
        in.ph = src->h;
        in.xf = in.yf = 1.0f;
        in.resized = 0;
        return in;
    }
    /* builtins.c:273-274: a *plain* float map is returned as it is.  One behind a resize wrapper (what filter code hands to
     * render(): every image value is RESIZE_IMAGE(STRIP_RESIZE(..)), drawable.c:213-227) is of type IMAGE_RESIZE and takes
     * the sampling branch like a drawable does: a new map, each pixel ORIG_VAL at the new map's own unit coordinates --
     * scaled by the wrapper's factors, nearest texel, zeros outside (builtins.c:303-343, 247-265) */
    if (src->kind == MMO_IMG_FLOATMAP && !in.resized) return in;
    slot = mmo_memo_slot(A, slot, 2, in.idx, 0, in.resized ? in.xf : 1.0f, in.resized ? in.yf : 1.0f);
    m = &A->memo[slot];
    dst = &A->images[A->native_slot_base + slot];
    if (!(m->valid && m->func == 2 && m->in_idx == in.idx && m->w == w && m->h == h)) {
        if (m->map == NULL || m->w != w || m->h != h) {
            free(m->map);
            m->map = malloc((size_t)w * h * 4 * sizeof(float));
        }
        mmo_render_image(A, src, in, w, h, m->map);
        mmo_fill_floatmap_desc(dst, m->map, w, h);
        m->valid = 1;
        m->func = 2;
        m->in_idx = in.idx;
        m->in2_idx = 0;
        m->a1 = in.resized ? in.xf : 1.0f;
        m->a2 = in.resized ? in.yf : 1.0f;
        m->w = w;
        m->h = h;
    }
    out.idx = A->native_slot_base + slot;
    out.pw = w;
    out.ph = h;
    out.xf = out.yf = 1.0f;
    out.resized = 0;
    return out;
}

void mmo_free_memo(mmo_args *A, int nslots) {
    int i;
    for (i = 0; i < nslots; ++i) {
        free(A->memo[i].map);
        memset(&A->memo[i], 0, sizeof A->memo[i]);
    }
}

/* ---- FIR path for sigma < 0.5 px (gauss.c:264-639) -------------------------------- */

/* gauss.c:264-306 */
static void make_rle_curve(double sigma, float **p_curve, int *p_length, float **p_sum, float *p_total) {
    const double sigma2 = 2 * sigma * sigma;
    const double l = sqrt(-sigma2 * log(1.0 / 255.0));
    int i, n, length;
    float *sum, *curve;
    n = ceil(l) * 2;
    if ((n % 2) == 0) n += 1;
    curve = malloc(sizeof(float) * n);
    length = n / 2;
    curve += length;
    curve[0] = 1.0;
    for (i = 1; i <= length; i++) {
        float temp = exp(-(i * i) / sigma2);
        curve[-i] = temp;
        curve[i] = temp;
    }
    sum = malloc(sizeof(float) * (2 * length + 1));
    sum[0] = 0;
    for (i = 1; i <= length * 2; i++) sum[i] = curve[i - length - 1] + sum[i - 1];
    sum += length;
    *p_total = sum[length] - sum[-length];
    *p_curve = curve;
    *p_sum = sum;
    *p_length = length;
}

/* gauss.c:315-378: pix[-border .. width+border-1] = edge-replicated line, rle = run lengths */
static int run_length_encode(const float *src, int *rle, float *pix, int dist, int width, int border) {
    float last;
    int count = 0, i, same = 0;
    src += dist * (width - 1);
    rle += width + border - 1;
    pix += width + border - 1;
    last = *src;
    for (i = 0; i < border; i++) {
        count++;
        *pix-- = last;
        *rle-- = count;
    }
    for (i = 0; i < width; i++) {
        float c = *src;
        src -= dist;
        if (c == last) {
            count++;
            *pix-- = last;
            *rle-- = count;
            same++;
        } else {
            count = 1;
            last = c;
            *pix-- = last;
            *rle-- = count;
        }
    }
    for (i = 0; i < border; i++) {
        count++;
        *pix-- = last;
        *rle-- = count;
    }
    return same;
}

/* gauss.c:380-420 (note the int-typed `s2` and `ctotal`: kept on purpose) */
static void do_encoded_lre(const int *enc, const float *src, float *dest, int width, int length, int dist,
                           int ctotal, const float *csum) {
    int col;
    for (col = 0; col < width; col++, dest += dist) {
        const int *rpt;
        const float *pix;
        int nb, i;
        float s1, val = 0.0;
        int start = -length;
        rpt = &enc[col + start];
        pix = &src[col + start];
        s1 = csum[start];
        nb = rpt[0];
        i = start + nb;
        while (i <= length) {
            int s2 = csum[i];
            val += pix[0] * (s2 - s1);
            s1 = s2;
            rpt = &rpt[nb];
            pix = &pix[nb];
            nb = rpt[0];
            i += nb;
        }
        val += pix[0] * (csum[length] - s1);
        val = val / ctotal;
        *dest = val;
    }
}

/* gauss.c:422-498 (the unrolled loops add the same terms in the same order) */
static void do_full_lre(const float *src, float *dest, int width, int length, int dist, const float *curve, float ctotal) {
    int col;
    for (col = 0; col < width; col++, dest += dist) {
        const float *x1, *x2, *c = &curve[0];
        int i;
        float val = 0.0;
        x1 = x2 = &src[col];
        val += x1[0] * c[0];
        c += 1;
        x1 += 1;
        x2 -= 1;
        i = length;
        while (i >= 1) {
            val += (x1[0] + x2[-0]) * c[0];
            c += 1;
            x1 += 1;
            x2 -= 1;
            i -= 1;
        }
        val = val / ctotal;
        *dest = val;
    }
}

static void rle_pass(float *map, int width, int height, float std_dev, int vertical) {
    float *curve, *sum, total;
    int length, line, b, i;
    int n = vertical ? height : width, lines = vertical ? width : height;
    int *rle;
    float *pix, *src, *dest;
    make_rle_curve(std_dev, &curve, &length, &sum, &total);
    rle = malloc(sizeof(int) * (n + 2 * length));
    rle += length;
    pix = malloc(sizeof(float) * (n + 2 * length));
    pix += length;
    src = malloc(sizeof(float) * n * 4);
    dest = malloc(sizeof(float) * n * 4);
    for (line = 0; line < lines; line++) {
        for (i = 0; i < n; ++i) {
            size_t off = vertical ? ((size_t)i * width + line) * 4 : ((size_t)line * width + i) * 4;
            memcpy(src + i * 4, map + off, sizeof(float) * 4);
        }
        for (b = 0; b < 4; b++) {
            int same = run_length_encode(src + b, rle, pix, 4, n, length);
            if (same > (3 * n) / 4) do_encoded_lre(rle, pix, dest + b, n, length, 4, total, sum);
            else do_full_lre(pix, dest + b, n, length, 4, curve, total);
        }
        for (i = 0; i < n; ++i) {
            size_t off = vertical ? ((size_t)i * width + line) * 4 : ((size_t)line * width + i) * 4;
            memcpy(map + off, dest + i * 4, sizeof(float) * 4);
        }
    }
    free(rle - length);
    free(pix - length);
    free(src);
    free(dest);
    free(sum - length);
    free(curve - length);
}

/* gauss.c:500-639 */
void mmo_gauss_rle(float *map, int width, int height, float hdev, float vdev) {
    if (vdev > 0.0) rle_pass(map, width, height, vdev, 1);
    if (hdev > 0.0) rle_pass(map, width, height, hdev, 0);
}

/* ---- cgamma (builtins/spec_func.c:35-64; Luke's approximation, double-complex inside) ---- */
float _Complex cgamma(float _Complex z) {
    static const double coeff[7] = {41.624436916439068, -51.224241022374774, 11.338755813488977, -0.747732687772388,
                                    0.008782877493061, -1.899030264e-6, 1.946335e-9};
    double _Complex s, H, w;
    int n;
    if (creal(z) < 0.0) {
        double _Complex denom = 1.0;
        int flr = -floor(creal(z));
        for (n = 0; n < flr; ++n) denom = denom * (z + n);
        return cgamma(z + flr) / denom;
    }
    w = z - 1.0;
    s = coeff[0];
    H = 1.0;
    for (n = 1; n < 7; n++) {
        H *= (w + 1 - n) / (w + n);
        s += coeff[n] * H;
    }
    return (2.506628274631 * cexp(-w - 5.5) * cpow(w + 5.5, w + 0.5) * s);
}
