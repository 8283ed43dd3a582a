// Native filters as hand-written HIP kernels for gfx950.
//
// gaussian_blur restates native-filters/gauss.c of the reference:
//   render_image (drawable -> float4 map, nearest)      builtins/builtins.c:303-343
//   gauss_iir: 4th-order recursive Gaussian, vertical pass over all four channels,
//   then horizontal (gauss.c:126-262), double accumulators, float store per pass
//   (transfer_pixels, gauss.c:117-124); coefficients gauss.c:38-115.
// The per-line arithmetic keeps the reference's operation order exactly (this file is
// compiled with -ffp-contract=off), so results are bit-identical to the CPU code.
//
// Data layout: float maps are float[h][w][4] interleaved in HBM (floatmap.c:30-46), a
// 16384^2 map is 4.29 GB; all intermediates stay on the device.
#include "native_filters.h"

#include <cmath>
#include <cstring>

namespace mm {

void *NativeWorkspace::reserve(size_t bytes) {
    if (bytes > scratch_bytes) {
        if (scratch) (void)hipFree(scratch);
        scratch = nullptr;
        if (hipMalloc(&scratch, bytes) != hipSuccess) { scratch_bytes = 0; return nullptr; }
        scratch_bytes = bytes;
    }
    return scratch;
}

void NativeWorkspace::release() {
    if (scratch) (void)hipFree(scratch);
    scratch = nullptr;
    scratch_bytes = 0;
}

namespace {

struct IirCoef { double n_p[5], n_m[5], d_p[5], d_m[5], bd_p[5], bd_m[5]; };

// gauss.c:38-115 (host, double libm -- identical to what the reference computes)
void find_iir_constants(IirCoef &c, float std_dev) {
    double div = sqrt(2 * M_PI) * std_dev;
    double x0 = -1.783 / std_dev, x1 = -1.723 / std_dev, x2 = 0.6318 / std_dev, x3 = 1.997 / std_dev;
    double x4 = 1.6803 / div, x5 = 3.735 / div, x6 = -0.6803 / div, x7 = -0.2598 / div;
    c.n_p[0] = x4 + x6;
    c.n_p[1] = (exp(x1) * (x7 * sin(x3) - (x6 + 2 * x4) * cos(x3)) + exp(x0) * (x5 * sin(x2) - (2 * x6 + x4) * cos(x2)));
    c.n_p[2] = (2 * exp(x0 + x1) * ((x4 + x6) * cos(x3) * cos(x2) - x5 * cos(x3) * sin(x2) - x7 * cos(x2) * sin(x3)) +
                x6 * exp(2 * x0) + x4 * exp(2 * x1));
    c.n_p[3] = (exp(x1 + 2 * x0) * (x7 * sin(x3) - x6 * cos(x3)) + exp(x0 + 2 * x1) * (x5 * sin(x2) - x4 * cos(x2)));
    c.n_p[4] = 0.0;
    c.d_p[0] = 0.0;
    c.d_p[1] = -2 * exp(x1) * cos(x3) - 2 * exp(x0) * cos(x2);
    c.d_p[2] = 4 * cos(x3) * cos(x2) * exp(x0 + x1) + exp(2 * x1) + exp(2 * x0);
    c.d_p[3] = -2 * cos(x2) * exp(x0 + 2 * x1) - 2 * cos(x3) * exp(x1 + 2 * x0);
    c.d_p[4] = exp(2 * x0 + 2 * x1);
    for (int i = 0; i <= 4; i++) c.d_m[i] = c.d_p[i];
    c.n_m[0] = 0.0;
    for (int i = 1; i <= 4; i++) c.n_m[i] = c.n_p[i] - c.d_p[i] * c.n_p[0];
    double sum_n_p = 0.0, sum_n_m = 0.0, sum_d = 0.0;
    for (int i = 0; i <= 4; i++) {
        sum_n_p += c.n_p[i];
        sum_n_m += c.n_m[i];
        sum_d += c.d_p[i];
    }
    double a = sum_n_p / (1.0 + sum_d), b = sum_n_m / (1.0 + sum_d);
    for (int i = 0; i <= 4; i++) {
        c.bd_p[i] = c.d_p[i] * a;
        c.bd_m[i] = c.d_m[i] * b;
    }
}

// ---- K2: render_image of a drawable (builtins.c:303-343) -----------------------------------
// fx = ((float)x - bx) / ax, ORIG_VAL with the *nearest* fetch, TUPLE_FROM_COLOR.
__global__ void __launch_bounds__(256) k_render_drawable(const uint32_t *__restrict__ src, int sw, int sh,
                                                         float scale_x, float scale_y, float middle_x, float middle_y,
                                                         int resized, float xf, float yf, uint32_t edge_x,
                                                         uint32_t edge_y, int supersampling, float4 *__restrict__ out,
                                                         int w, int h) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)w * h) return;
    const int px = (int)(i % w), py = (int)(i / w);
    const float ax = (float)((float)(w - 1) / 2.0), bx = ax;
    const float by = (float)((float)(h - 1) / 2.0);
    const float ay = (float)(by * -1.0);
    float x = ((float)px - bx) / ax;
    float y = ((float)py - by) / ay;
    if (resized) { x *= xf; y *= yf; }
    x = (x + middle_x) * scale_x;
    y = -((y - middle_y) * scale_y);
    if (!supersampling) { x += 0.5; y += 0.5; }
    const int ix = (int)floor((double)x), iy = (int)floor((double)y);
    uint32_t c;
    if (ix < 0 || ix >= sw) c = edge_x;
    else if (iy < 0 || iy >= sh) c = edge_y;
    else c = src[(long)iy * sw + ix];
    float4 t;
    t.x = (c >> 24) / 255.0;
    t.y = ((c >> 16) & 0xff) / 255.0;
    t.z = ((c >> 8) & 0xff) / 255.0;
    t.w = (c & 0xff) / 255.0;
    out[i] = t;
}

// ---- K3/K4 (first version): one lane per line of one channel ------------------------------------
// Line L has `n` elements at map[base(L) + k*estride].  Causal sweep stores vp (double) to
// scratch, anticausal sweep adds vm and stores the float result in place.  History of the
// last four inputs / outputs lives in registers.
struct LineGeom { long n_lines; int n; long estride; int vertical; int w; };

__device__ __forceinline__ long line_base(const LineGeom &g, long L) {
    // vertical: L = col*4+ch -> base L ; horizontal: L = row*4+ch -> row*w*4 + ch
    return g.vertical ? L : (L >> 2) * (long)g.w * 4 + (L & 3);
}

__global__ void __launch_bounds__(256) k_iir_lines(float *__restrict__ map, double *__restrict__ scratch, LineGeom g,
                                                   IirCoef c) {
    const long L = (long)blockIdx.x * 256 + threadIdx.x;
    if (L >= g.n_lines) return;
    float *p = map + line_base(g, L);
    double *sc = scratch + L;             // scratch[k][n_lines]: coalesced across lanes
    const int n = g.n;
    const long es = g.estride;
    // ---- causal ----
    {
        const float initial = p[0];
        double s1 = 0, s2 = 0, s3 = 0, s4 = 0;   // inputs k-1..k-4
        double v1 = 0, v2 = 0, v3 = 0, v4 = 0;   // outputs k-1..k-4
        for (int k = 0; k < n; ++k) {
            const double s0 = (double)p[(long)k * es];
            double acc = 0.0;
            acc += c.n_p[0] * s0 - c.d_p[0] * acc;
            if (k >= 1) acc += c.n_p[1] * s1 - c.d_p[1] * v1; else acc += (c.n_p[1] - c.bd_p[1]) * initial;
            if (k >= 2) acc += c.n_p[2] * s2 - c.d_p[2] * v2; else acc += (c.n_p[2] - c.bd_p[2]) * initial;
            if (k >= 3) acc += c.n_p[3] * s3 - c.d_p[3] * v3; else acc += (c.n_p[3] - c.bd_p[3]) * initial;
            if (k >= 4) acc += c.n_p[4] * s4 - c.d_p[4] * v4; else acc += (c.n_p[4] - c.bd_p[4]) * initial;
            sc[(long)k * g.n_lines] = acc;
            s4 = s3; s3 = s2; s2 = s1; s1 = s0;
            v4 = v3; v3 = v2; v2 = v1; v1 = acc;
        }
    }
    // ---- anticausal ----
    {
        const float initial = p[(long)(n - 1) * es];
        double s1 = 0, s2 = 0, s3 = 0, s4 = 0;   // inputs k+1..k+4
        double v1 = 0, v2 = 0, v3 = 0, v4 = 0;
        for (int k = n - 1, j = 0; k >= 0; --k, ++j) {
            const double s0 = (double)p[(long)k * es];
            double acc = 0.0;
            acc += c.n_m[0] * s0 - c.d_m[0] * acc;
            if (j >= 1) acc += c.n_m[1] * s1 - c.d_m[1] * v1; else acc += (c.n_m[1] - c.bd_m[1]) * initial;
            if (j >= 2) acc += c.n_m[2] * s2 - c.d_m[2] * v2; else acc += (c.n_m[2] - c.bd_m[2]) * initial;
            if (j >= 3) acc += c.n_m[3] * s3 - c.d_m[3] * v3; else acc += (c.n_m[3] - c.bd_m[3]) * initial;
            if (j >= 4) acc += c.n_m[4] * s4 - c.d_m[4] * v4; else acc += (c.n_m[4] - c.bd_m[4]) * initial;
            const double vp = sc[(long)k * g.n_lines];
            p[(long)k * es] = (float)(vp + acc);
            s4 = s3; s3 = s2; s2 = s1; s1 = s0;
            v4 = v3; v3 = v2; v2 = v1; v1 = acc;
        }
    }
}

int gaussian_blur(const HNativeRec &rec, const std::vector<HImageDesc> &images, int rw, int rh, float *out_map,
                  NativeWorkspace &ws, hipStream_t s, std::string *err) {
    const HImage &img = rec.args[0].img;
    float hdev = rec.args[1].f, vdev = rec.args[2].f;
    if (img.idx < 0 || img.idx >= (int)images.size()) { *err = "gaussian_blur: input is not a bitmap image"; return -1; }
    const HImageDesc &in = images[img.idx];
    int w = rw, h = rh;
    if (in.kind == IMG_FLOATMAP) {
        w = in.w;
        h = in.h;
        if (w != rw || h != rh) { *err = "gaussian_blur: float-map input of a different size is not supported"; return -1; }
        if (hipMemcpyAsync(out_map, in.data, (size_t)w * h * 16, hipMemcpyDeviceToDevice, s) != hipSuccess) {
            *err = "gaussian_blur: copy failed";
            return -1;
        }
    } else if (in.kind == IMG_DRAWABLE) {
        long n = (long)w * h;
        k_render_drawable<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(
            (const uint32_t *)in.data, in.w, in.h, in.scale_x, in.scale_y, in.middle_x, in.middle_y, img.resized, img.xf,
            img.yf, 0u, 0u, 0, (float4 *)out_map, w, h);
    } else {
        *err = "gaussian_blur: input image is not bound";
        return -1;
    }
    // gauss.c:659-660 (float products, fabs)
    const float ax = (float)((float)(w - 1) / 2.0);
    float ay = (float)((float)(h - 1) / 2.0);
    ay *= -1.0f;
    float hs = (float)fabs(hdev * ax), vs = (float)fabs(vdev * ay);
    if (hs < 0.5f || vs < 0.5f) {
        *err = "gaussian_blur: sigma < 0.5 px (the reference's RLE/FIR path, gauss.c:500-639) is not implemented on the GPU yet";
        return -1;
    }
    size_t lines = (size_t)std::max(w, h) * 4;
    double *scratch = (double *)ws.reserve((size_t)w * h * 4 * sizeof(double));
    (void)lines;
    if (!scratch) { *err = "gaussian_blur: out of device memory for the scan scratch"; return -1; }
    IirCoef c;
    // vertical pass first (gauss.c:155-201)
    find_iir_constants(c, vs);
    LineGeom gv{(long)w * 4, h, (long)w * 4, 1, w};
    k_iir_lines<<<(unsigned)((gv.n_lines + 255) / 256), 256, 0, s>>>(out_map, scratch, gv, c);
    // horizontal pass (gauss.c:203-252)
    find_iir_constants(c, hs);
    LineGeom gh{(long)h * 4, w, 4, 0, w};
    k_iir_lines<<<(unsigned)((gh.n_lines + 255) / 256), 256, 0, s>>>(out_map, scratch, gh, c);
    if (hipGetLastError() != hipSuccess) { *err = "gaussian_blur: kernel launch failed"; return -1; }
    return 0;
}

}  // namespace

int run_native_filter(const std::string &func, const HNativeRec &rec, const std::vector<HImageDesc> &images,
                      int render_w, int render_h, float *out_map, NativeWorkspace &ws, hipStream_t stream,
                      std::string *err) {
    if (func == "native_filter_gaussian_blur") return gaussian_blur(rec, images, render_w, render_h, out_map, ws, stream, err);
    *err = "native filter " + func + " is not implemented in the HIP backend yet";
    return -1;
}

}  // namespace mm
