/*
 * mm_oracle_fft.c -- CPU ORACLE (test infrastructure, NOT product code): the three
 * FFT-based native filters of the reference's native-filters/convolve.c.
 *
 *   convolve        convolve.c:67-174
 *   half_convolve   convolve.c:176-272
 *   visualize_fft   convolve.c:274-357
 *
 * The reference calls FFTW 3 (fftw_plan_dft_r2c_2d / c2r_2d, FFTW_ESTIMATE), which is not
 * in this image and not vendored in the reference tree.  What FFTW computes is the
 * unnormalised DFT  Y[ky][kx] = sum_y sum_x X[y][x] e^{-2 pi i (ky y/h + kx x/w)}  for
 * kx < w/2+1, and for c2r the inverse with the Hermitian half implied (the imaginary parts
 * of the kx = 0 and kx = w/2 row terms do not contribute).  This file evaluates exactly
 * those sums directly (O(n*(w+h)), long double accumulation, twiddles from cosl/sinl), so
 * it is a more accurate value of the same quantity than either FFTW or hipFFT produces;
 * FFT results are compared with a tolerance (tests state it), everything around the
 * transforms follows the reference line by line.  PARITY of the transforms themselves is
 * pinned only by the reference's own golden image for visualize_fft
 * (tests/golden/utilities_visualize_fft.png).
 */
#include "mm_oracle.h"

#include <complex.h>
#include <stdio.h>

typedef struct { long double re, im; } ldc;

static ldc *make_twiddles(int n) {
    ldc *t = malloc(sizeof(ldc) * (size_t)n);
    const long double two_pi = 6.283185307179586476925286766559005768L;
    int m;
    for (m = 0; m < n; ++m) {
        t[m].re = cosl(two_pi * m / n);
        t[m].im = sinl(two_pi * m / n);
    }
    return t;
}

/* out[h][cw] = r2c DFT of in[h][w] (what fftw_plan_dft_r2c_2d(h, w, ...) computes) */
void mmo_dft_r2c_2d(const double *in, double _Complex *out, int w, int h) {
    const int cw = w / 2 + 1;
    ldc *tw = make_twiddles(w), *th = make_twiddles(h);
    ldc *rows = malloc(sizeof(ldc) * (size_t)h * cw);
    int x, y, k;
    for (y = 0; y < h; ++y)
        for (k = 0; k < cw; ++k) {
            long double sr = 0, si = 0;
            long m = 0;
            for (x = 0; x < w; ++x) {
                sr += in[(size_t)y * w + x] * tw[m].re;
                si -= in[(size_t)y * w + x] * tw[m].im;
                m += k;
                if (m >= w) m -= w;
            }
            rows[(size_t)y * cw + k].re = sr;
            rows[(size_t)y * cw + k].im = si;
        }
    for (k = 0; k < cw; ++k)
        for (int ky = 0; ky < h; ++ky) {
            long double sr = 0, si = 0;
            long m = 0;
            for (y = 0; y < h; ++y) {
                const ldc v = rows[(size_t)y * cw + k];
                /* v * (c - i s) */
                sr += v.re * th[m].re + v.im * th[m].im;
                si += v.im * th[m].re - v.re * th[m].im;
                m += ky;
                if (m >= h) m -= h;
            }
            out[(size_t)ky * cw + k] = (double)sr + (double)si * I;
        }
    free(rows);
    free(tw);
    free(th);
}

/* out[h][w] = unnormalised c2r inverse of in[h][cw] (fftw_plan_dft_c2r_2d): complex inverse
 * along the columns, then the real inverse of each row. */
void mmo_dft_c2r_2d(const double _Complex *in, double *out, int w, int h) {
    const int cw = w / 2 + 1;
    ldc *tw = make_twiddles(w), *th = make_twiddles(h);
    ldc *cols = malloc(sizeof(ldc) * (size_t)h * cw);
    int x, y, k;
    for (k = 0; k < cw; ++k)
        for (y = 0; y < h; ++y) {
            long double sr = 0, si = 0;
            long m = 0;
            for (int ky = 0; ky < h; ++ky) {
                const long double vr = creal(in[(size_t)ky * cw + k]), vi = cimag(in[(size_t)ky * cw + k]);
                /* v * (c + i s) */
                sr += vr * th[m].re - vi * th[m].im;
                si += vi * th[m].re + vr * th[m].im;
                m += y;
                if (m >= h) m -= h;
            }
            cols[(size_t)y * cw + k].re = sr;
            cols[(size_t)y * cw + k].im = si;
        }
    for (y = 0; y < h; ++y)
        for (x = 0; x < w; ++x) {
            long double s = 0;
            long m = 0;
            for (k = 0; k < cw; ++k) {
                const ldc v = cols[(size_t)y * cw + k];
                const long double term = v.re * tw[m].re - v.im * tw[m].im;
                s += (k == 0 || 2 * k == w) ? term : 2 * term;
                m += x;
                if (m >= w) m -= w;
            }
            out[(size_t)y * w + x] = (double)s;
        }
    free(cols);
    free(tw);
    free(th);
}

/* convolve.c:36-64 */
static void copy(double *dest, const float *src, int n) {
    int i;
    for (i = 0; i < n; ++i) dest[i] = src[i * 4];
}

static double copy_and_add(double *dest, const float *src, int n) {
    int half;
    if (n <= 0) return 0.0;
    if (n == 1) return dest[0] = src[0];
    if (n == 2) {
        double d1, d2;
        d1 = dest[0] = src[0];
        d2 = dest[1] = src[4];
        return d1 + d2;
    }
    half = n / 2;
    return copy_and_add(dest, src, half) + copy_and_add(dest + half, src + half * 4, n - half);
}

/* the argument image as a float map of the given size: used as is when it already is one,
 * otherwise render_image (convolve.c:88-95) */
static const float *as_floatmap(const mmo_args *A, mmo_image img, int w, int h, float **owned) {
    const mmo_image_desc *src = mmo_desc_of(A, img.idx);
    *owned = NULL;
    if (src->kind == MMO_IMG_FLOATMAP && src->w == w && src->h == h) return (const float *)src->data;
    *owned = malloc((size_t)w * h * 4 * sizeof(float));
    mmo_render_image(A, src, img, w, h, *owned);
    return *owned;
}

static int memo_hit(mmo_native_memo *m, int func, int in, int in2, float a1, float a2, int w, int h) {
    return m->valid && m->func == func && m->in_idx == in && m->in2_idx == in2 && m->a1 == a1 && m->a2 == a2 &&
           m->w == w && m->h == h;
}

static mmo_image finish(const mmo_args *A, int slot, mmo_native_memo *m, int func, int in, int in2, float a1, float a2,
                        int w, int h) {
    mmo_image out;
    mmo_fill_floatmap_desc(&A->images[A->native_slot_base + slot], m->map, w, h);
    m->valid = 1;
    m->func = func;
    m->in_idx = in;
    m->in2_idx = in2;
    m->a1 = a1;
    m->a2 = a2;
    m->w = w;
    m->h = h;
    out.idx = A->native_slot_base + slot;
    out.pw = w;
    out.ph = h;
    out.xf = out.yf = 1.0f;
    out.resized = 0;
    return out;
}

static void memo_map(mmo_native_memo *m, int w, int h) {
    if (m->map == NULL || m->w != w || m->h != h) {
        free(m->map);
        m->map = malloc((size_t)w * h * 4 * sizeof(float));
    }
}

/* convolve.c:67-174 */
mmo_image mmo_native_convolve(const mmo_args *A, int slot, mmo_image in, mmo_image filter, float normalize_f,
                              float copy_alpha_f) {
    mmo_native_memo *m = &A->memo[slot = mmo_memo_slot(A, slot, 3, in.idx, filter.idx, normalize_f, copy_alpha_f)];
    const int normalize = normalize_f != 0.0, copy_alpha = copy_alpha_f != 0.0;
    int w = A->render_width, h = A->render_height;
    if (mmo_desc_of(A, in.idx)->kind == MMO_IMG_FLOATMAP) {
        w = mmo_desc_of(A, in.idx)->w;
        h = mmo_desc_of(A, in.idx)->h;
    }
    if (!memo_hit(m, 3, in.idx, filter.idx, normalize_f, copy_alpha_f, w, h)) {
        float *own_in, *own_filter;
        const float *in_data = as_floatmap(A, in, w, h, &own_in);
        const float *filter_data = as_floatmap(A, filter, w, h, &own_filter);
        const int n = h * w, nhalf = w * (h / 2) + w / 2, cn = h * (w / 2 + 1);
        double *fftw_in = malloc(sizeof(double) * n);
        double _Complex *image_out = malloc(sizeof(double _Complex) * cn);
        double _Complex *filter_out = malloc(sizeof(double _Complex) * cn);
        const int num_channels = copy_alpha ? 3 : 4;
        int i, channel;
        memo_map(m, w, h);
        for (channel = 0; channel < num_channels; ++channel) {
            for (i = 0; i < n; ++i) fftw_in[i] = in_data[i * 4 + channel];
            mmo_dft_r2c_2d(fftw_in, image_out, w, h);
            if (normalize) {
                double d1 = copy_and_add(fftw_in, filter_data + channel + (n - nhalf) * 4, nhalf);
                double d2 = copy_and_add(fftw_in + nhalf, filter_data + channel, n - nhalf);
                double factor = 1.0 / (d1 + d2);
                for (i = 0; i < n; ++i) fftw_in[i] *= factor;
            } else {
                copy(fftw_in, filter_data + channel + (n - nhalf) * 4, nhalf);
                copy(fftw_in + nhalf, filter_data + channel, n - nhalf);
            }
            mmo_dft_r2c_2d(fftw_in, filter_out, w, h);
            for (i = 0; i < cn; ++i) image_out[i] *= filter_out[i];
            mmo_dft_c2r_2d(image_out, fftw_in, w, h);
            for (i = 0; i < n; ++i) m->map[i * 4 + channel] = fftw_in[i] / n;
        }
        if (copy_alpha)
            for (i = 0; i < n; ++i) m->map[i * 4 + 3] = in_data[i * 4 + 3];
        free(fftw_in);
        free(image_out);
        free(filter_out);
        free(own_in);
        free(own_filter);
    }
    return finish(A, slot, m, 3, in.idx, filter.idx, normalize_f, copy_alpha_f, w, h);
}

/* convolve.c:176-272 */
mmo_image mmo_native_half_convolve(const mmo_args *A, int slot, mmo_image in, mmo_image filter, float copy_alpha_f) {
    mmo_native_memo *m = &A->memo[slot = mmo_memo_slot(A, slot, 4, in.idx, filter.idx, copy_alpha_f, 0.0f)];
    const int copy_alpha = copy_alpha_f != 0.0;
    int w = A->render_width, h = A->render_height;
    if (mmo_desc_of(A, in.idx)->kind == MMO_IMG_FLOATMAP) {
        w = mmo_desc_of(A, in.idx)->w;
        h = mmo_desc_of(A, in.idx)->h;
    }
    if (!memo_hit(m, 4, in.idx, filter.idx, copy_alpha_f, 0.0f, w, h)) {
        float *own_in, *own_filter;
        const float *in_data = as_floatmap(A, in, w, h, &own_in);
        const float *filter_data = as_floatmap(A, filter, w, h, &own_filter);
        const int n = h * w, nhalf = w * (h / 2) + w / 2, cw = w / 2 + 1, cn = h * cw;
        double *fftw_in = malloc(sizeof(double) * n);
        double _Complex *image_out = malloc(sizeof(double _Complex) * cn);
        const int num_channels = copy_alpha ? 3 : 4;
        int i, x, y, channel;
        memo_map(m, w, h);
        for (channel = 0; channel < num_channels; ++channel) {
            for (i = 0; i < n; ++i) fftw_in[i] = in_data[i * 4 + channel];
            mmo_dft_r2c_2d(fftw_in, image_out, w, h);
            for (y = 0; y < h; ++y)
                for (x = 0; x < cw; ++x) {
                    int out_idx = x + y * w;
                    int in_idx = out_idx + nhalf;
                    if (in_idx >= n) in_idx -= n;
                    image_out[x + y * cw] *= filter_data[in_idx * 4 + channel];
                }
            mmo_dft_c2r_2d(image_out, fftw_in, w, h);
            for (i = 0; i < n; ++i) m->map[i * 4 + channel] = fftw_in[i] / n;
        }
        if (copy_alpha)
            for (i = 0; i < n; ++i) m->map[i * 4 + 3] = in_data[i * 4 + 3];
        free(fftw_in);
        free(image_out);
        free(own_in);
        free(own_filter);
    }
    return finish(A, slot, m, 4, in.idx, filter.idx, copy_alpha_f, 0.0f, w, h);
}

/* convolve.c:274-357 */
mmo_image mmo_native_visualize_fft(const mmo_args *A, int slot, mmo_image in, float ignore_alpha_f) {
    mmo_native_memo *m = &A->memo[slot = mmo_memo_slot(A, slot, 5, in.idx, -1, ignore_alpha_f, 0.0f)];
    const int ignore_alpha = ignore_alpha_f != 0.0;
    int w = A->render_width, h = A->render_height;
    if (mmo_desc_of(A, in.idx)->kind == MMO_IMG_FLOATMAP) {
        w = mmo_desc_of(A, in.idx)->w;
        h = mmo_desc_of(A, in.idx)->h;
    }
    if (!memo_hit(m, 5, in.idx, -1, ignore_alpha_f, 0.0f, w, h)) {
        float *own_in;
        const float *in_data = as_floatmap(A, in, w, h, &own_in);
        const int n = h * w, cw = w / 2 + 1, cn = h * cw;
        const double sqrtn = sqrt(n);
        double *fftw_in = malloc(sizeof(double) * n);
        double _Complex *image_out = malloc(sizeof(double _Complex) * cn);
        const int num_channels = ignore_alpha ? 3 : 4;
        int i, x, y, channel;
        memo_map(m, w, h);
        memset(m->map, 0, sizeof(float) * (size_t)n * 4);
        for (channel = 0; channel < num_channels; ++channel) {
            for (i = 0; i < n; ++i) fftw_in[i] = in_data[i * 4 + channel];
            mmo_dft_r2c_2d(fftw_in, image_out, w, h);
            for (y = 0; y < h; ++y) {
                int out_y = y + h / 2;
                if (out_y >= h) out_y -= h;
                for (x = 0; x < cw; ++x) {
                    int out_x1 = cw - 1 - x;
                    int out_x2 = x + w - cw;
                    double val = cabs(image_out[x + y * cw]) / sqrtn;
                    m->map[(out_x1 + out_y * w) * 4 + channel] = val;
                    m->map[(out_x2 + out_y * w) * 4 + channel] = val;
                }
            }
        }
        if (ignore_alpha)
            for (i = 0; i < n; ++i) m->map[i * 4 + 3] = 1.0;
        free(fftw_in);
        free(image_out);
        free(own_in);
    }
    return finish(A, slot, m, 5, in.idx, -1, ignore_alpha_f, 0.0f, w, h);
}
