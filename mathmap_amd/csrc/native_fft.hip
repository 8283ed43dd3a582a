// FFT-based native filters for gfx950: convolve, half_convolve, visualize_fft
// (the reference's native-filters/convolve.c, which calls FFTW 3 in double precision).
//
// The transforms themselves are plain library FFTs and go to hipFFT/rocFFT (double, real <->
// half-complex, all channels batched in one plan); everything around them -- channel
// de-interleave with the filter's half-image rotation, the normalisation sum, the spectral
// products, the spectrum visualisation and the re-interleave -- is hand-written here and
// stays in HBM.  libhipfft.so is loaded on the first FFT filter call (dlopen), so filters
// that never use these three do not pay for it; a missing library is a hard error.
//
// HBM layout (one workspace allocation, n = w*h, cn = h*(w/2+1), C = channels transformed):
//   [in map float4[n]] [filter map float4[n]] [planes double[C][n]] [specA double2[C][cn]]
//   [specB double2[C][cn]] [partial sums] [factor double[4]]
// A 16384^2 frame needs about 30 GB of workspace, sized for the 288 GB of an MI355X.
//
// FFT results are not bit-identical to FFTW's (different factorisation, same O(eps log n)
// error); tests compare against a direct long-double DFT with a stated tolerance.
#include <dlfcn.h>
#include <hipfft/hipfft.h>

#include <algorithm>
#include <cmath>
#include <mutex>

#include "native_filters.h"

namespace mm {
namespace {

struct FftApi {
    void *lib = nullptr;
    hipfftResult (*PlanMany)(hipfftHandle *, int, int *, int *, int, int, int *, int, int, hipfftType, int) = nullptr;
    hipfftResult (*SetStream)(hipfftHandle, hipStream_t) = nullptr;
    hipfftResult (*ExecD2Z)(hipfftHandle, hipfftDoubleReal *, hipfftDoubleComplex *) = nullptr;
    hipfftResult (*ExecZ2D)(hipfftHandle, hipfftDoubleComplex *, hipfftDoubleReal *) = nullptr;
    hipfftResult (*Destroy)(hipfftHandle) = nullptr;
    std::string error;
};

FftApi &fft_api() {
    static FftApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"libhipfft.so", "libhipfft.so.0", "/opt/rocm/lib/libhipfft.so"};
        for (const char *n : names)
            if ((api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
        if (!api.lib) { api.error = std::string("cannot load libhipfft.so: ") + dlerror(); return; }
        auto sym = [&](const char *n) {
            void *p = dlsym(api.lib, n);
            if (!p && api.error.empty()) api.error = std::string("libhipfft.so lacks ") + n;
            return p;
        };
        api.PlanMany = (decltype(api.PlanMany))sym("hipfftPlanMany");
        api.SetStream = (decltype(api.SetStream))sym("hipfftSetStream");
        api.ExecD2Z = (decltype(api.ExecD2Z))sym("hipfftExecD2Z");
        api.ExecZ2D = (decltype(api.ExecZ2D))sym("hipfftExecZ2D");
        api.Destroy = (decltype(api.Destroy))sym("hipfftDestroy");
    });
    return api;
}

// planes[c][i] = (double)src[(i + shift) mod n].c  (* factor[c])        convolve.c:38-44,118-143
// shift = 0 for the image; n - nhalf for the filter, whose two halves trade places.
__global__ void __launch_bounds__(256) k_fft_load(const float4 *__restrict__ src, double *__restrict__ planes, long n,
                                                  long shift, const double *__restrict__ factor, int nch) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    long j = i + shift;
    if (j >= n) j -= n;
    const float4 v = src[j];
    const float c[4] = {v.x, v.y, v.z, v.w};
    for (int k = 0; k < nch; ++k) {
        double d = (double)c[k];
        if (factor) d *= factor[k];
        planes[(size_t)k * n + i] = d;
    }
}

// Per-channel sums of a float map in double: block partials, then one block folds them.
// (The reference adds in a recursive-halving order, convolve.c:46-64; any pairwise order of
// float addends in double agrees with it to an ulp or two of the sum.)
__global__ void __launch_bounds__(256) k_fft_chan_partial(const float4 *__restrict__ src, long n, double *__restrict__ partial) {
    __shared__ double red[4][256];
    double s[4] = {0, 0, 0, 0};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float4 v = src[i];
        s[0] += (double)v.x;
        s[1] += (double)v.y;
        s[2] += (double)v.z;
        s[3] += (double)v.w;
    }
    for (int k = 0; k < 4; ++k) red[k][threadIdx.x] = s[k];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off)
            for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x < 4) partial[(size_t)blockIdx.x * 4 + threadIdx.x] = red[threadIdx.x][0];
}

__global__ void __launch_bounds__(256) k_fft_chan_factor(const double *__restrict__ partial, int nblocks, double *__restrict__ factor) {
    __shared__ double red[4][256];
    double s[4] = {0, 0, 0, 0};
    for (int b = threadIdx.x; b < nblocks; b += 256)
        for (int k = 0; k < 4; ++k) s[k] += partial[(size_t)b * 4 + k];
    for (int k = 0; k < 4; ++k) red[k][threadIdx.x] = s[k];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off)
            for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x < 4) factor[threadIdx.x] = 1.0 / red[threadIdx.x][0];      // convolve.c:124
}

// image_out[i] *= filter_out[i]                                            convolve.c:146-147
__global__ void __launch_bounds__(256) k_fft_spec_mul(double2 *__restrict__ a, const double2 *__restrict__ b, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const double2 p = a[i], q = b[i];
    double2 r;
    r.x = p.x * q.x - p.y * q.y;
    r.y = p.x * q.y + p.y * q.x;
    a[i] = r;
}

// image_out[x + y*cw] *= filter[(x + y*w + nhalf) mod n].channel           convolve.c:236-246
__global__ void __launch_bounds__(256) k_fft_spec_half_mul(double2 *__restrict__ a, const float4 *__restrict__ filt, int w,
                                                           int h, int cw, long nhalf, int nch) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long cn = (long)h * cw, n = (long)w * h;
    if (i >= cn) return;
    const int y = (int)(i / cw), x = (int)(i - (long)y * cw);
    long in_idx = (long)x + (long)y * w + nhalf;
    if (in_idx >= n) in_idx -= n;
    const float4 v = filt[in_idx];
    const float c[4] = {v.x, v.y, v.z, v.w};
    for (int k = 0; k < nch; ++k) {
        double2 p = a[(size_t)k * cn + i];
        p.x *= (double)c[k];
        p.y *= (double)c[k];
        a[(size_t)k * cn + i] = p;
    }
}

// out[i].c = fftw_in[i] / n, alpha copied from the input when only 3 channels were
// transformed                                                               convolve.c:150-159
__global__ void __launch_bounds__(256) k_fft_store(const double *__restrict__ planes, const float4 *__restrict__ in,
                                                   float4 *__restrict__ out, long n, int nch) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float c[4];
    const double dn = (double)(int)n;
    for (int k = 0; k < nch; ++k) c[k] = (float)(planes[(size_t)k * n + i] / dn);
    if (nch < 4) c[3] = in[i].w;
    out[i] = make_float4(c[0], c[1], c[2], c[3]);
}

// Spectrum magnitude, centred, mirrored to both halves                     convolve.c:315-350
// The reference writes columns out_x1 = cw-1-x and out_x2 = x+w-cw for x = 0..cw-1 in
// order, so where two x land on one column the larger x wins: gathered here per output.
__global__ void __launch_bounds__(256) k_fft_visualize(const double2 *__restrict__ spec, float4 *__restrict__ out, int w,
                                                       int h, int cw, int nch, double sqrtn) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long n = (long)w * h, cn = (long)h * cw;
    if (i >= n) return;
    const int oy = (int)(i / w), ox = (int)(i - (long)oy * w);
    int y = oy - h / 2;               // out_y = y + h/2 (mod h)
    if (y < 0) y += h;
    const int xa = cw - 1 - ox, xb = ox - (w - cw);
    int x = -1;
    if (xa >= 0 && xa < cw) x = xa;
    if (xb >= 0 && xb < cw && xb > x) x = xb;
    float c[4] = {0.f, 0.f, 0.f, 0.f};
    if (x >= 0)
        for (int k = 0; k < nch; ++k) {
            const double2 v = spec[(size_t)k * cn + (size_t)y * cw + x];
            c[k] = (float)(hypot(v.x, v.y) / sqrtn);
        }
    if (nch < 4) c[3] = 1.0f;
    out[i] = make_float4(c[0], c[1], c[2], c[3]);
}

inline unsigned blocks_for(long n) { return (unsigned)((n + 255) / 256); }

int ensure_plans(NativeWorkspace &ws, int w, int h, int batch, hipStream_t s, std::string *err) {
    FftApi &api = fft_api();
    if (!api.error.empty()) { *err = "FFT native filters need hipFFT: " + api.error; return -1; }
    if (!(ws.fft_valid && ws.fft_w == w && ws.fft_h == h && ws.fft_batch == batch)) {
        fft_release_plans(ws);
        int dims[2] = {h, w};
        hipfftHandle fwd = nullptr, inv = nullptr;
        if (api.PlanMany(&fwd, 2, dims, nullptr, 1, 0, nullptr, 1, 0, HIPFFT_D2Z, batch) != HIPFFT_SUCCESS) {
            *err = "hipfftPlanMany(D2Z) failed";
            return -1;
        }
        if (api.PlanMany(&inv, 2, dims, nullptr, 1, 0, nullptr, 1, 0, HIPFFT_Z2D, batch) != HIPFFT_SUCCESS) {
            api.Destroy(fwd);
            *err = "hipfftPlanMany(Z2D) failed";
            return -1;
        }
        ws.fft_fwd = fwd;
        ws.fft_inv = inv;
        ws.fft_w = w;
        ws.fft_h = h;
        ws.fft_batch = batch;
        ws.fft_valid = true;
    }
    if (api.SetStream((hipfftHandle)ws.fft_fwd, s) != HIPFFT_SUCCESS || api.SetStream((hipfftHandle)ws.fft_inv, s) != HIPFFT_SUCCESS) {
        *err = "hipfftSetStream failed";
        return -1;
    }
    return 0;
}

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

void fft_release_plans(NativeWorkspace &ws) {
    if (!ws.fft_valid) return;
    FftApi &api = fft_api();
    if (api.Destroy) {
        api.Destroy((hipfftHandle)ws.fft_fwd);
        api.Destroy((hipfftHandle)ws.fft_inv);
    }
    ws.fft_fwd = ws.fft_inv = nullptr;
    ws.fft_valid = false;
}

int fft_native_filter(const std::string &func, const HNativeRec &rec, const std::vector<HImageDesc> &images, int render_w,
                      int render_h, float *out_map, NativeWorkspace &ws, hipStream_t s, std::string *err) {
    const bool is_conv = func == "native_filter_convolve", is_half = func == "native_filter_half_convolve";
    const char *who = is_conv ? "convolve" : is_half ? "half_convolve" : "visualize_fft";
    const HImage &img = rec.args[0].img;
    if (img.idx < 0 || img.idx >= (int)images.size()) { *err = std::string(who) + ": input is not a bitmap image"; return -1; }
    // a float-map input keeps its own size (convolve.c:88-90); the runtime's output map is
    // render-sized, so other sizes are refused rather than silently cropped
    const int w = render_w, h = render_h;
    if (images[img.idx].kind == IMG_FLOATMAP && (images[img.idx].w != w || images[img.idx].h != h)) {
        *err = std::string(who) + ": float-map input of a different size is not supported";
        return -1;
    }
    if ((long)w * h > 0x7fffffffL) { *err = std::string(who) + ": image too large"; return -1; }
    // flag arguments arrive as bool_const floats (convolve.c:73-74,183,282)
    bool normalize = false, drop_alpha;
    if (is_conv) {
        normalize = rec.args[2].f != 0.0f;
        drop_alpha = rec.args[3].f != 0.0f;       // copy_alpha
    } else if (is_half)
        drop_alpha = rec.args[2].f != 0.0f;       // copy_alpha
    else
        drop_alpha = rec.args[1].f != 0.0f;       // ignore_alpha
    const int nch = drop_alpha ? 3 : 4;
    const long n = (long)w * h;
    const int cw = w / 2 + 1;
    const long cn = (long)h * cw;
    const long nhalf = (long)w * (h / 2) + w / 2;
    const int sum_blocks = (int)std::min<long>(1024, (n + 255) / 256);

    const size_t map_bytes = align256((size_t)n * 16);
    const size_t planes_bytes = align256((size_t)n * 8 * nch);
    const size_t spec_bytes = align256((size_t)cn * 16 * nch);
    const size_t partial_bytes = align256((size_t)sum_blocks * 4 * 8);
    const size_t total = 2 * map_bytes + planes_bytes + 2 * spec_bytes + partial_bytes + 256;
    char *base = (char *)ws.reserve(total);
    if (!base) { *err = std::string(who) + ": out of device memory for the FFT workspace"; return -1; }
    float *in_tmp = (float *)base;
    float *filt_tmp = (float *)(base + map_bytes);
    double *planes = (double *)(base + 2 * map_bytes);
    double2 *spec_a = (double2 *)(base + 2 * map_bytes + planes_bytes);
    double2 *spec_b = (double2 *)(base + 2 * map_bytes + planes_bytes + spec_bytes);
    double *partial = (double *)(base + 2 * map_bytes + planes_bytes + 2 * spec_bytes);
    double *factor = (double *)(base + 2 * map_bytes + planes_bytes + 2 * spec_bytes + partial_bytes);

    const float *in_map = nullptr, *filt_map = nullptr;
    if (native_input_map(who, img, images, ws.env, w, h, in_tmp, &in_map, s, err) != 0) return -1;
    if ((is_conv || is_half) && native_input_map(who, rec.args[1].img, images, ws.env, w, h, filt_tmp, &filt_map, s, err) != 0) return -1;
    if (ensure_plans(ws, w, h, nch, s, err) != 0) return -1;
    FftApi &api = fft_api();
    hipfftHandle fwd = (hipfftHandle)ws.fft_fwd, inv = (hipfftHandle)ws.fft_inv;

    // FFT of the input image, all channels in one batched plan
    k_fft_load<<<blocks_for(n), 256, 0, s>>>((const float4 *)in_map, planes, n, 0, nullptr, nch);
    if (api.ExecD2Z(fwd, planes, (hipfftDoubleComplex *)spec_a) != HIPFFT_SUCCESS) { *err = std::string(who) + ": forward FFT failed"; return -1; }

    if (is_conv) {
        const double *fac = nullptr;
        if (normalize) {
            k_fft_chan_partial<<<sum_blocks, 256, 0, s>>>((const float4 *)filt_map, n, partial);
            k_fft_chan_factor<<<1, 256, 0, s>>>(partial, sum_blocks, factor);
            fac = factor;
        }
        k_fft_load<<<blocks_for(n), 256, 0, s>>>((const float4 *)filt_map, planes, n, n - nhalf, fac, nch);
        if (api.ExecD2Z(fwd, planes, (hipfftDoubleComplex *)spec_b) != HIPFFT_SUCCESS) { *err = "convolve: filter FFT failed"; return -1; }
        k_fft_spec_mul<<<blocks_for(cn * nch), 256, 0, s>>>(spec_a, spec_b, cn * nch);
    } else if (is_half) {
        k_fft_spec_half_mul<<<blocks_for(cn), 256, 0, s>>>(spec_a, (const float4 *)filt_map, w, h, cw, nhalf, nch);
    } else {
        k_fft_visualize<<<blocks_for(n), 256, 0, s>>>(spec_a, (float4 *)out_map, w, h, cw, nch, sqrt((double)(int)n));
        if (hipGetLastError() != hipSuccess) { *err = "visualize_fft: kernel launch failed"; return -1; }
        return 0;
    }
    if (api.ExecZ2D(inv, (hipfftDoubleComplex *)spec_a, planes) != HIPFFT_SUCCESS) { *err = std::string(who) + ": inverse FFT failed"; return -1; }
    k_fft_store<<<blocks_for(n), 256, 0, s>>>(planes, (const float4 *)in_map, (float4 *)out_map, n, nch);
    if (hipGetLastError() != hipSuccess) { *err = std::string(who) + ": kernel launch failed"; return -1; }
    return 0;
}

}  // namespace mm
