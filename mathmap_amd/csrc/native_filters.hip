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

#include <algorithm>
#include <cmath>
#include <cstring>
#include <type_traits>
#include <vector>

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
    fft_release_plans(*this);
    for (Timed &t : timed) timed_free.push_back({t.a, t.b});
    timed.clear();
    for (auto &p : timed_free) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    timed_free.clear();
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
                                                         int resized, float xf, float yf, NativeEnv env,
                                                         float4 *__restrict__ out, int w, int h) {
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
    if (!env.supersampling) { x += 0.5; y += 0.5; }     // get_orig_val_pixel, builtins.c:153-157
    // cvttsd2si semantics for NaN / out-of-range (see mm_f2i in mm_device.h): INT_MIN, i.e. outside
    const double fx = floor((double)x), fy = floor((double)y);
    int ix = (fx >= -2147483648.0 && fx < 2147483648.0) ? (int)fx : (int)0x80000000;
    int iy = (fy >= -2147483648.0 && fy < 2147483648.0) ? (int)fy : (int)0x80000000;
    // apply_edge_behaviour (builtins.c:40-119); INT_MIN negates to itself and INT_MIN % n is C's
    switch (env.edge_x) {
        case 1: if (ix < 0) ix = ix % sw + sw; else if (ix >= sw) ix %= sw; break;
        case 2: if (ix < 0) ix = (int)(0u - (unsigned)ix) % sw; else if (ix >= sw) ix = (sw - 1) - (ix % sw); break;
        case 3:
            if (ix < 0) { ix = (int)(0u - (unsigned)ix) % sw; iy = (int)((unsigned)(sh - 1) - (unsigned)iy); }
            else if (ix >= sw) { ix = (sw - 1) - (ix % sw); iy = (int)((unsigned)(sh - 1) - (unsigned)iy); }
            break;
        default: break;
    }
    switch (env.edge_y) {
        case 1: if (iy < 0) iy = iy % sh + sh; else if (iy >= sh) iy %= sh; break;
        case 2: if (iy < 0) iy = (int)(0u - (unsigned)iy) % sh; else if (iy >= sh) iy = (sh - 1) - (iy % sh); break;
        case 3:
            if (iy < 0) { ix = (int)((unsigned)(sw - 1) - (unsigned)ix); iy = (int)(0u - (unsigned)iy) % sh; }
            else if (iy >= sh) { ix = (int)((unsigned)(sw - 1) - (unsigned)ix); iy = (sh - 1) - (iy % sh); }
            break;
        default: break;
    }
    uint32_t c;
    if (ix < 0 || ix >= sw) c = env.edge_color_x;
    else if (iy < 0 || iy >= sh) c = env.edge_color_y;
    else c = src[(long)iy * sw + ix];
    float4 t;
    t.x = (c >> 24) / 255.0;
    t.y = ((c >> 16) & 0xff) / 255.0;
    t.z = ((c >> 8) & 0xff) / 255.0;
    t.w = (c & 0xff) / 255.0;
    out[i] = t;
}

// render_image of a float map behind a resize wrapper (builtins.c:303-343 with ORIG_VAL's float-map branch, 247-265):
// the new map's pixel at its own unit coordinates, scaled by the wrapper's factors, nearest texel of the source, zeros outside
__global__ void __launch_bounds__(256) k_render_floatmap(const float4 *__restrict__ src, int sw, int sh, float sax, float sbx, float say,
                                                         float sby, float xf, float yf, float4 *__restrict__ out, int w, int h) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)w * h) return;
    const int px = (int)(i % w), py = (int)(i / w);
    const float ax = (float)((float)(w - 1) / 2.0), bx = ax;
    const float by = (float)((float)(h - 1) / 2.0);
    const float ay = (float)(by * -1.0);
    float x = ((float)px - bx) / ax;
    float y = ((float)py - by) / ay;
    x *= xf;
    y *= yf;
    const int ix = (int)lrintf(sax * x + sbx), iy = (int)lrintf(say * y + sby);
    float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (ix >= 0 && ix < sw && iy >= 0 && iy < sh) v = src[(long)iy * sw + ix];
    out[i] = v;
}

void render_drawable_impl(const HImageDesc &in, const HImage &img, const NativeEnv &env, float *dst, int w, int h, hipStream_t s) {
    const long n = (long)w * h;
    k_render_drawable<<<(unsigned)((n + 255) / 256), 256, 0, s>>>((const uint32_t *)in.data, in.w, in.h, in.scale_x, in.scale_y,
                                                                 in.middle_x, in.middle_y, img.resized, img.xf, img.yf, env,
                                                                 (float4 *)dst, w, h);
}

// ---- K3/K4: recursive Gaussian along one axis, output written transposed ---------------------
// Input layout  in[k][line][ch]   (element k of line `line`: ((k*lines + line)*4 + ch) floats)
// Output layout out[line][k][ch]  (transposed), so running the same kernel pair twice does
// vertical (lines = columns) then horizontal (lines = rows of the original) and ends in the
// original layout -- no separate transpose pass, and every global access is coalesced:
//   * one lane per (line, channel): consecutive lanes read consecutive floats of a row of `in`;
//   * the causal sweep streams its f64 partials to scratch[k][line*4+ch] (coalesced);
//   * the anticausal sweep stages 16 finished steps x 16 lines per wave in LDS and writes
//     them as 256-byte runs of the transposed map.
// The sweeps are sequential along k by definition (4th-order recurrences in the reference's
// exact operation order); the only parallelism is lines x channels, so memory latency is
// hidden by prefetching IIR_U steps ahead in registers rather than by occupancy.
#define IIR_U 16
#ifndef IIR_PF
#define IIR_PF 3          // blocks a causal sweep's loads run ahead of its arithmetic
#endif

// seg: segment length (a multiple of IIR_U, >= n when the lines are not split), halo: warm-up
// steps a segment's sweeps take before they reach it (a multiple of IIR_U), lane_blocks: 256-lane
// workgroups per segment.
struct LineArgs { int n; int lines; int seg; int halo; unsigned lane_blocks; };

__device__ __forceinline__ double iir_step(double s0, double s1, double s2, double s3, double s4, double v1, double v2,
                                           double v3, double v4, const double *n, const double *d) {
    // gauss.c:181-185 with terms = 4: acc starts at 0 and adds (n[i]*s[k-i] - d[i]*v[k-i]) for i = 0..4.
    // i = 0 reads d[0]*acc with acc = +0 and d[0] = 0.0 (find_iir_constants sets d_p[0] = d_m[0] = 0):
    // that product is +0 and x - (+0) is x for every x (signed zeros, NaN included), so the term is
    // n[0]*s0; the addition to the +0 accumulator stays (it turns a -0 product into +0).
    double acc = 0.0;
    acc += n[0] * s0;
    acc += n[1] * s1 - d[1] * v1;
    acc += n[2] * s2 - d[2] * v2;
    acc += n[3] * s3 - d[3] * v3;
    acc += n[4] * s4 - d[4] * v4;
    return acc;
}

// The i = 0 term when more is known about the input s0:
//  * causal, s0 never -0 (bytes / 255): n_p[0] = 1 / (sqrt(2 pi) sigma) > 0, so the product is -0 only for
//    s0 = -0 and the addition to the +0 accumulator changes nothing else -- drop the addition;
//  * anticausal, s0 finite: n_m[0] is +0 (find_iir_constants), the product is +-0 and the accumulator
//    stays +0 -- drop the term.  (An infinite or NaN s0 would make it NaN: float-map inputs keep it.)
template <bool S0_NOT_NEG_ZERO>
__device__ __forceinline__ double iir_step_causal(double s0, double s1, double s2, double s3, double s4, double v1, double v2,
                                                  double v3, double v4, const double *n, const double *d) {
    if (!S0_NOT_NEG_ZERO) return iir_step(s0, s1, s2, s3, s4, v1, v2, v3, v4, n, d);
    double acc = n[0] * s0;
    acc += n[1] * s1 - d[1] * v1;
    acc += n[2] * s2 - d[2] * v2;
    acc += n[3] * s3 - d[3] * v3;
    acc += n[4] * s4 - d[4] * v4;
    return acc;
}
template <bool S0_FINITE>
__device__ __forceinline__ double iir_step_anticausal(double s0, double s1, double s2, double s3, double s4, double v1, double v2,
                                                      double v3, double v4, const double *n, const double *d) {
    if (!S0_FINITE) return iir_step(s0, s1, s2, s3, s4, v1, v2, v3, v4, n, d);
    double acc = 0.0;
    acc += n[1] * s1 - d[1] * v1;
    acc += n[2] * s2 - d[2] * v2;
    acc += n[3] * s3 - d[3] * v3;
    acc += n[4] * s4 - d[4] * v4;
    return acc;
}

// steps 0..3 of a sweep (gauss.c:178-190: fewer than 4 predecessors, the rest uses the edge value)
__device__ __forceinline__ double iir_edge_step(int j, double s0, double s1, double s2, double s3, double v1, double v2,
                                                double v3, const double *n, const double *d, const double *bd,
                                                float initial) {
    double acc = 0.0;
    acc += n[0] * s0;        // d[0] = 0: see iir_step
    if (j >= 1) acc += n[1] * s1 - d[1] * v1; else acc += (n[1] - bd[1]) * initial;
    if (j >= 2) acc += n[2] * s2 - d[2] * v2; else acc += (n[2] - bd[2]) * initial;
    if (j >= 3) acc += n[3] * s3 - d[3] * v3; else acc += (n[3] - bd[3]) * initial;
    acc += (n[4] - bd[4]) * initial;
    return acc;
}

// Element source of a scan.  fetch() is the bare load (its result is not touched until the
// element is consumed, so a block's 16 prefetches stay in flight while the previous block is
// computed); decode() turns the loaded word into the float the recurrence sees.
// A float map laid out in[k][line][ch] ...
//
// Addressing: element k of a lane is base[k * stride + lane] with `base`, `stride` and (in the
// sweeps) k wave-uniform, so the row address is scalar arithmetic and the load takes the scalar
// base + 32-bit lane offset form.  row()/at() let a block of 16 consecutive k share one row address:
// with a single wave on a SIMD every instruction, scalar ones included, takes an issue slot from
// the recurrence, and per-element 64-bit index arithmetic was a third of a block's instructions.
template <bool FINITE>
struct MapSrcT {
    typedef float raw_t;
    static constexpr bool not_neg_zero = false;    // a float map may hold -0 (so may a pass's own output: tiny negative sums round to it)
    static constexpr bool finite = FINITE;         // the first pass's output when its input was a drawable
    const float *p;        // the map (uniform)
    unsigned stride;       // lines * 4
    unsigned lane;         // line * 4 + channel
    __device__ __forceinline__ raw_t fetch(int k) const { return p[(size_t)k * stride + lane]; }
    __device__ __forceinline__ const raw_t *row(int k) const { return p + (size_t)k * stride; }
    __device__ __forceinline__ raw_t at(const raw_t *r, int u) const {       // scalar base + 32-bit byte offset
        return *(const raw_t *)((const char *)r + (((unsigned)u * stride + lane) << 2));
    }
    __device__ __forceinline__ float decode(raw_t v) const { return v; }
    __device__ __forceinline__ MapSrcT for_lane(long L) const { return MapSrcT{p, stride, (unsigned)L}; }
};
typedef MapSrcT<false> MapSrc;
typedef MapSrcT<true> FiniteMapSrc;
// ... or the input drawable itself when render_image's coordinate mapping (builtins.c:303-343)
// is the identity -- a drawable of the render size, which the host establishes by evaluating
// the mapping for every row and column (same float operations) before it picks this source.
// render_image is then fused into the first pass: 4 instead of 16 B/px read, no 4.3 GB map.
struct DrawableSrc {
    typedef uint32_t raw_t;
    static constexpr bool not_neg_zero = true, finite = true;     // byte / 255
    const uint32_t *p;     // the drawable's window (uniform)
    unsigned sw;           // source pitch in pixels
    unsigned lane;         // column
    int shift;             // 24 - 8*channel
    __device__ __forceinline__ raw_t fetch(int k) const { return p[(size_t)k * sw + lane]; }
    __device__ __forceinline__ const raw_t *row(int k) const { return p + (size_t)k * sw; }
    __device__ __forceinline__ raw_t at(const raw_t *r, int u) const {
        return *(const raw_t *)((const char *)r + (((unsigned)u * sw + lane) << 2));
    }
    // TUPLE_FROM_COLOR: (float)((c >> shift & 0xff) / 255.0) -- the correctly rounded quotient k / 255 for all 256 bytes,
    // and so is one Newton step on the f32 product (q = k r, q' = q + (k - 255 q) r, the residual exact in an fma;
    // enumerated in tests/test_cpu_suite.py, mm_bytes_to_unit in mm_device.h): three f32 instructions instead of a
    // conversion to f64, an f64 product and a conversion back, on a wave that has the SIMD to itself
    __device__ __forceinline__ float decode(raw_t c) const {
        const float k = (float)((c >> shift) & 0xff), r = 1.0f / 255.0f;
        const float q = k * r;
        return __builtin_fmaf(__builtin_fmaf(-255.0f, q, k), r, q);
    }
    __device__ __forceinline__ DrawableSrc for_lane(long L) const { return DrawableSrc{p, sw, (unsigned)(L >> 2), 24 - 8 * (int)(L & 3)}; }
};

// Causal sweep.  Nothing but the recurrence state at every IIR_U-th step leaves the kernel:
// ckpt[b][i][lane] = v(kb-1-i), i = 0..3, kb = b*IIR_U, b >= 1 -- 2 bytes per element instead of
// the 8 a full f64 partial-sum map would take.  The anticausal kernel re-runs the causal
// recurrence block by block from these checkpoints (same operations, same order: identical
// values) right before it needs them.
// The segment [s0, s1) of the line: the sweep starts at k0 = max(0, s0 - halo), at k0 > 0 as if k0
// were the line's edge (see the segment note at k_iir_causal), and stores the checkpoints of the
// segment's own blocks only.
// Where the second pass also writes the packed output pixels (NativeDirectOut), in its own
// coordinates: lines are rows of the window, steps are columns.
struct PackOut { unsigned char *out; long row_stride; int line_lo, line_hi, k_lo, k_hi; int write_map; };

// new_template.c.in:279-293 for output_bpp 4: CLAMP01 in float, the product in double, the conversion to a byte
// truncates: the byte is floor(255 c), exactly what one f32 fma under round-toward-zero leaves in the low mantissa bits
// of 255 c + 2^23 (mm_pack_rgba8 in mm_device.h has the argument).  Only the single-precision rounding mode is
// switched (MODE bits 1:0), inside one asm statement; the f64 recurrences around it are not affected.
__device__ __forceinline__ unsigned pack_rgba8(float4 v) {
    const float r = __builtin_amdgcn_fmed3f(v.x, 0.0f, 1.0f), g = __builtin_amdgcn_fmed3f(v.y, 0.0f, 1.0f);
    const float b = __builtin_amdgcn_fmed3f(v.z, 0.0f, 1.0f), a = __builtin_amdgcn_fmed3f(v.w, 0.0f, 1.0f);
    unsigned ur, ug, ub, ua;
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
                 "v_fma_f32 %0, %4, %8, %9\n\t"
                 "v_fma_f32 %1, %5, %8, %9\n\t"
                 "v_fma_f32 %2, %6, %8, %9\n\t"
                 "v_fma_f32 %3, %7, %8, %9\n\t"
                 "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
                 : "=&v"(ur), "=&v"(ug), "=&v"(ub), "=&v"(ua)
                 : "v"(r), "v"(g), "v"(b), "v"(a), "s"(255.0f), "v"(8388608.0f));
    const unsigned lo = __builtin_amdgcn_perm(ug, ur, 0x0c0c0400u);      // r | g << 8
    const unsigned hi = __builtin_amdgcn_perm(ua, ub, 0x0c0c0400u);      // b | a << 8
    return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
}

struct IirState { double s1, s2, s3, s4, v1, v2, v3, v4; };

// IIR_PF + 1 consecutive blocks starting at kb0, ring slot j holding block kb0 + j*IIR_U.  A block's
// 16 loads are issued IIR_PF blocks before the block is computed: one block of arithmetic (~0.8 us) is
// shorter than a trip to HBM under load, and with one wave per SIMD nothing else hides the
// difference.  The ring is indexed statically, so it lives in registers (a lone wave may use all
// 512).  FAST: every block of the group and every block prefetched by it is a full interior block of
// the segment -- straight-line code without a branch, which is also what lets the compiler count
// the outstanding loads exactly instead of waiting for all of them at a join.
template <bool FAST, class Src>
__device__ __forceinline__ void causal_group(const Src &src, typename Src::raw_t (&ring)[IIR_PF + 1][IIR_U], IirState &st,
                                             double *__restrict__ ck, unsigned stride, unsigned lane, int kb0, int k0, int s0,
                                             int s1, float initial, const IirCoef &c) {
    double s1_ = st.s1, s2 = st.s2, s3 = st.s3, s4 = st.s4, v1 = st.v1, v2 = st.v2, v3 = st.v3, v4 = st.v4;
#pragma unroll
    for (int j = 0; j <= IIR_PF; ++j) {
        const int kb = kb0 + j * IIR_U;
        if (FAST || kb < s1) {
            typename Src::raw_t *cur = ring[j], *far = ring[(j + IIR_PF) % (IIR_PF + 1)];
            const int kf = kb + IIR_PF * IIR_U;
            if (FAST || kf + IIR_U <= s1) {
                const typename Src::raw_t *r = src.row(kf);
#pragma unroll
                for (int u = 0; u < IIR_U; ++u) far[u] = src.at(r, u);
            } else {
#pragma unroll
                for (int u = 0; u < IIR_U; ++u) far[u] = src.fetch(kf + u < s1 ? kf + u : k0);   // past the end: unused
            }
            if (FAST || (kb > 0 && kb >= s0)) {
                double *q = ck + (size_t)(kb / IIR_U) * 4 * stride;
                // (read back by the anticausal kernel after this whole sweep: non-temporal, like the map)
                __builtin_nontemporal_store(v1, &q[lane]);
                __builtin_nontemporal_store(v2, &q[stride + lane]);
                __builtin_nontemporal_store(v3, &q[2 * stride + lane]);
                __builtin_nontemporal_store(v4, &q[3 * stride + lane]);
            }
            if (FAST || (kb > k0 && kb + IIR_U <= s1)) {
                // interior block: no edge steps, no bounds; the state "shifts" are register renames
#pragma unroll
                for (int u = 0; u < IIR_U; ++u) {
                    const double s0v = (double)src.decode(cur[u]);
                    const double acc = iir_step_causal<Src::not_neg_zero>(s0v, s1_, s2, s3, s4, v1, v2, v3, v4, c.n_p, c.d_p);
                    s4 = s3; s3 = s2; s2 = s1_; s1_ = s0v;
                    v4 = v3; v3 = v2; v2 = v1; v1 = acc;
                }
            } else {
#pragma unroll
                for (int u = 0; u < IIR_U; ++u) {
                    const int k = kb + u;
                    if (k < s1) {
                        const double s0v = (double)src.decode(cur[u]);
                        const double acc = (kb == k0 && u < 4) ? iir_edge_step(u, s0v, s1_, s2, s3, v1, v2, v3, c.n_p, c.d_p, c.bd_p, initial)
                                                                : iir_step(s0v, s1_, s2, s3, s4, v1, v2, v3, v4, c.n_p, c.d_p);
                        s4 = s3; s3 = s2; s2 = s1_; s1_ = s0v;
                        v4 = v3; v3 = v2; v2 = v1; v1 = acc;
                    }
                }
            }
        }
    }
    st = IirState{s1_, s2, s3, s4, v1, v2, v3, v4};
}

template <class Src>
__device__ __forceinline__ void causal_sweep(const Src &src, double *__restrict__ ck, unsigned stride, unsigned lane, int k0,
                                             int s0, int s1, const IirCoef &c) {
    const float initial = src.decode(src.fetch(k0));
    IirState st{0, 0, 0, 0, 0, 0, 0, 0};
    typename Src::raw_t ring[IIR_PF + 1][IIR_U];
#pragma unroll
    for (int j = 0; j < IIR_PF; ++j)
#pragma unroll
        for (int u = 0; u < IIR_U; ++u) {
            const int k = k0 + j * IIR_U + u;
            ring[j][u] = src.fetch(k < s1 ? k : k0);
        }
    const int G = (IIR_PF + 1) * IIR_U;
    int kb0 = k0;
    do {        // the group with the edge steps, and a segment's warm-up groups (no checkpoints)
        causal_group<false>(src, ring, st, ck, stride, lane, kb0, k0, s0, s1, initial, c);
        kb0 += G;
    } while (kb0 < s0 && kb0 < s1);
    for (; kb0 + (2 * IIR_PF + 1) * IIR_U <= s1; kb0 += G) causal_group<true>(src, ring, st, ck, stride, lane, kb0, k0, s0, s1, initial, c);
    for (; kb0 < s1; kb0 += G) causal_group<false>(src, ring, st, ck, stride, lane, kb0, k0, s0, s1, initial, c);
}

// One block of the causal recurrence re-run from its checkpoint (block 0: from the edge).
template <bool GUARD, class Src>
__device__ __forceinline__ void rerun_block(const Src &src, int b, int n, const typename Src::raw_t *in, const typename Src::raw_t *r4,
                                            const double *ckv, const IirCoef &c, float initial_p, double *vc) {
    const int kb = b * IIR_U;
    double cs1 = src.decode(r4[3]), cs2 = src.decode(r4[2]), cs3 = src.decode(r4[1]), cs4 = src.decode(r4[0]);
    double cv1 = ckv[0], cv2 = ckv[1], cv3 = ckv[2], cv4 = ckv[3];
#pragma unroll
    for (int u = 0; u < IIR_U; ++u) {
        if (!GUARD) {
            const double s0 = (double)src.decode(in[u]);
            const double acc = iir_step_causal<Src::not_neg_zero>(s0, cs1, cs2, cs3, cs4, cv1, cv2, cv3, cv4, c.n_p, c.d_p);
            vc[u] = acc;
            cs4 = cs3; cs3 = cs2; cs2 = cs1; cs1 = s0;
            cv4 = cv3; cv3 = cv2; cv2 = cv1; cv1 = acc;
        } else {
            vc[u] = 0.0;
            if (kb + u < n) {
                const double s0 = (double)src.decode(in[u]);
                const double acc = (b == 0 && u < 4) ? iir_edge_step(u, s0, cs1, cs2, cs3, cv1, cv2, cv3, c.n_p, c.d_p, c.bd_p, initial_p)
                                                     : iir_step(s0, cs1, cs2, cs3, cs4, cv1, cv2, cv3, cv4, c.n_p, c.d_p);
                vc[u] = acc;
                cs4 = cs3; cs3 = cs2; cs2 = cs1; cs1 = s0;
                cv4 = cv3; cv3 = cv2; cv2 = cv1; cv1 = acc;
            }
        }
    }
}

// Anticausal sweep + causal re-run + sum, written transposed (see the layout note above).
// Software-pipelined three deep: while block b takes its anticausal steps, block b-1's causal
// values are re-run -- two independent dependency chains interleaved step by step, which is
// what fills the f64 pipeline of a SIMD that holds a single wave -- and block b-2 is in flight
// from memory.
// Segment [sg0, sg1) of the line: the anticausal recurrence starts at k1 - 1, k1 = min(n, sg1 + halo)
// (at k1 < n as if k1 were the line's end) and runs unrecorded down to sg1 first.
template <class Src>
__device__ __forceinline__ void anticausal_sweep(const Src &src, const double *__restrict__ ck, unsigned stride, unsigned clane,
                                                 int n, int sg0, int sg1, int k1,
                                                 const IirCoef &c, float *tw, int lane, long line0, int lines,
                                                 float *__restrict__ outT, const PackOut &po, bool active) {
    typedef typename Src::raw_t raw_t;
    const int ll = lane >> 2, ch = lane & 3;                // line within the wave's 16, channel
    const float initial_m = src.decode(src.fetch(k1 - 1)), initial_p = src.decode(src.fetch(0));
    double s1 = 0, s2 = 0, s3 = 0, s4 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0;      // anticausal state
    const int bbeg = sg0 / IIR_U, nblocks = (sg1 + IIR_U - 1) / IIR_U;          // this segment's blocks
    if (k1 > sg1) {
        raw_t cur_s[IIR_U], nxt_s[IIR_U];
        // warm-up over [sg1, k1): sg1 is a block boundary here (only the last segment ends elsewhere, and
        // it has no warm-up)
        const int wend = (k1 + IIR_U - 1) / IIR_U;
#pragma unroll
        for (int u = 0; u < IIR_U; ++u) cur_s[u] = src.fetch((wend - 1) * IIR_U + u < k1 ? (wend - 1) * IIR_U + u : sg1);
        for (int b = wend - 1; b >= nblocks; --b) {
            const int kb = b * IIR_U;
#pragma unroll
            for (int u = 0; u < IIR_U; ++u) nxt_s[u] = src.fetch(b > nblocks ? kb - IIR_U + u : sg1);
            if (kb + 2 * IIR_U <= k1) {
#pragma unroll
                for (int u = IIR_U - 1; u >= 0; --u) {
                    const double s0 = (double)src.decode(cur_s[u]);
                    const double acc = iir_step(s0, s1, s2, s3, s4, v1, v2, v3, v4, c.n_m, c.d_m);
                    s4 = s3; s3 = s2; s2 = s1; s1 = s0;
                    v4 = v3; v3 = v2; v2 = v1; v1 = acc;
                }
            } else {
#pragma unroll
                for (int u = IIR_U - 1; u >= 0; --u) {
                    const int k = kb + u;
                    if (k < k1) {
                        const int j = k1 - 1 - k;
                        const double s0 = (double)src.decode(cur_s[u]);
                        double acc;
                        if (j < 4) acc = iir_edge_step(j, s0, s1, s2, s3, v1, v2, v3, c.n_m, c.d_m, c.bd_m, initial_m);
                        else acc = iir_step(s0, s1, s2, s3, s4, v1, v2, v3, v4, c.n_m, c.d_m);
                        s4 = s3; s3 = s2; s2 = s1; s1 = s0;
                        v4 = v3; v3 = v2; v2 = v1; v1 = acc;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < IIR_U; ++u) cur_s[u] = nxt_s[u];
        }
    }
    // Pipeline slots.  At step b one slot is `cur` (block b: inputs and re-run causal values), one is
    // `nxt` (block b-1: inputs, the 4 inputs before it and its checkpoint; its causal values are re-run
    // during the step) and one is `far` (block b-2, being loaded).  The roles rotate through three
    // statically indexed slots -- three steps are one group of straight-line code -- so the
    // rotation costs no register moves.
    struct Slot { raw_t s[IIR_U]; raw_t r4[4]; double ck[4]; double vc[IIR_U]; };
    Slot slot[3];
    auto load_block = [&](auto fast, int blk, Slot &o) {
        const int kb = blk * IIR_U;
        if (decltype(fast)::value || (blk > 0 && kb + IIR_U <= n)) {
            const raw_t *r = src.row(kb - 4);
            const double *q = ck + (size_t)blk * 4 * stride;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                o.r4[i] = src.at(r, i);
                o.ck[i] = *(const double *)((const char *)q + (((unsigned)i * stride + clane) << 3));
            }
#pragma unroll
            for (int u = 0; u < IIR_U; ++u) o.s[u] = src.at(r, 4 + u);
        } else {
#pragma unroll
            for (int u = 0; u < IIR_U; ++u) o.s[u] = src.fetch(kb + u < n ? kb + u : 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                o.r4[i] = src.fetch(blk > 0 ? kb - 4 + i : 0);
                o.ck[i] = blk > 0 ? ck[((size_t)blk * 4 + i) * stride + clane] : 0.0;
            }
        }
    };
    auto interior_causal = [&](int blk) { return blk > 0 && (blk + 1) * IIR_U <= n; };
    // One step: anticausal steps of block b (slot cur) next to the causal re-run of block b-1 (slot nxt),
    // loads of block b-2 (slot far), 16 x 16 transposed write.  FAST: b-2 is a full block with a
    // checkpoint and block b lies at least a block inside [.., k1) -- no edge steps, no bounds.
    auto step = [&](auto fast, int b, Slot &cur, Slot &nxt, Slot &far) {
        constexpr bool FAST = decltype(fast)::value;
        const int kb = b * IIR_U;
        if (FAST || b - 2 >= bbeg) load_block(fast, b - 2, far);
        if (FAST || (b > bbeg && kb + 2 * IIR_U <= k1 && interior_causal(b - 1))) {
            // both chains unguarded: causal step u of block b-1 next to anticausal step 15-u of block b
            double cs1 = src.decode(nxt.r4[3]), cs2 = src.decode(nxt.r4[2]), cs3 = src.decode(nxt.r4[1]), cs4 = src.decode(nxt.r4[0]);
            double cv1 = nxt.ck[0], cv2 = nxt.ck[1], cv3 = nxt.ck[2], cv4 = nxt.ck[3];
#pragma unroll
            for (int u = 0; u < IIR_U; ++u) {
                {
                    const double s0 = (double)src.decode(nxt.s[u]);
                    const double acc = iir_step_causal<Src::not_neg_zero>(s0, cs1, cs2, cs3, cs4, cv1, cv2, cv3, cv4, c.n_p, c.d_p);
                    nxt.vc[u] = acc;
                    cs4 = cs3; cs3 = cs2; cs2 = cs1; cs1 = s0;
                    cv4 = cv3; cv3 = cv2; cv2 = cv1; cv1 = acc;
                }
                {
                    const int ua = IIR_U - 1 - u;
                    const double s0 = (double)src.decode(cur.s[ua]);
                    const double acc = iir_step_anticausal<Src::finite>(s0, s1, s2, s3, s4, v1, v2, v3, v4, c.n_m, c.d_m);
                    tw[ll * (IIR_U * 4 + 4) + ua * 4 + ch] = (float)(cur.vc[ua] + acc);   // transfer_pixels, gauss.c:117-124
                    s4 = s3; s3 = s2; s2 = s1; s1 = s0;
                    v4 = v3; v3 = v2; v2 = v1; v1 = acc;
                }
            }
        } else {
            if (b > bbeg) {
                if (interior_causal(b - 1)) rerun_block<false>(src, b - 1, n, nxt.s, nxt.r4, nxt.ck, c, initial_p, nxt.vc);
                else rerun_block<true>(src, b - 1, n, nxt.s, nxt.r4, nxt.ck, c, initial_p, nxt.vc);
            }
#pragma unroll
            for (int u = IIR_U - 1; u >= 0; --u) {
                const int k = kb + u;
                if (k < n) {
                    const int j = k1 - 1 - k;                    // steps from the end (of the line, or of the warm-up)
                    const double s0 = (double)src.decode(cur.s[u]);
                    double acc;
                    if (j < 4) acc = iir_edge_step(j, s0, s1, s2, s3, v1, v2, v3, c.n_m, c.d_m, c.bd_m, initial_m);
                    else acc = iir_step(s0, s1, s2, s3, s4, v1, v2, v3, v4, c.n_m, c.d_m);
                    tw[ll * (IIR_U * 4 + 4) + u * 4 + ch] = (float)(cur.vc[u] + acc);   // transfer_pixels, gauss.c:117-124
                    s4 = s3; s3 = s2; s2 = s1; s1 = s0;
                    v4 = v3; v3 = v2; v2 = v1; v1 = acc;
                }
            }
        }
        // the tile is private to this wave: a wave-level fence suffices
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // write 16 lines x 16 steps transposed: lane -> (line = i*4 + lane/16, step = lane%16)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int tl = i * 4 + (lane >> 4), kk = lane & 15;
            const long line = line0 + tl;
            const int k = kb + kk;
            if (line < lines && (FAST || k < n)) {
                const float4 v = *(const float4 *)&tw[tl * (IIR_U * 4 + 4) + kk * 4];
                if (po.write_map) {      // written once, read by the next pass long after it has left the caches: non-temporal
                    typedef float mm_f4 __attribute__((ext_vector_type(4)));
                    __builtin_nontemporal_store(mm_f4{v.x, v.y, v.z, v.w}, (mm_f4 *)&outT[(line * (long)n + k) * 4]);
                }
                if (po.out && line >= po.line_lo && line < po.line_hi && k >= po.k_lo && k < po.k_hi)
                    __builtin_nontemporal_store(pack_rgba8(v), (unsigned *)(po.out + (line - po.line_lo) * po.row_stride + (long)(k - po.k_lo) * 4));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto group = [&](auto fast, int b) {
        constexpr bool FAST = decltype(fast)::value;
        step(fast, b, slot[0], slot[1], slot[2]);
        if (FAST || b - 1 >= bbeg) step(fast, b - 1, slot[1], slot[2], slot[0]);
        if (FAST || b - 2 >= bbeg) step(fast, b - 2, slot[2], slot[0], slot[1]);
    };
    // prologue: the last block's inputs and causal values, the one before it on its way
    load_block(std::false_type{}, nblocks - 1, slot[0]);
    if (nblocks - 1 > bbeg) load_block(std::false_type{}, nblocks - 2, slot[1]);
    rerun_block<true>(src, nblocks - 1, n, slot[0].s, slot[0].r4, slot[0].ck, c, initial_p, slot[0].vc);
    int b = nblocks - 1;
    const int fast_floor = (bbeg > 1 ? bbeg : 1) + 4;         // lowest b whose group is all-FAST: b-4 >= max(bbeg, 1)
    for (; b >= bbeg && b * IIR_U + 2 * IIR_U > k1; b -= 3) group(std::false_type{}, b);      // the group at the line's end
    for (; b >= fast_floor; b -= 3) group(std::true_type{}, b);
    for (; b >= bbeg; b -= 3) group(std::false_type{}, b);
    (void)active;
}

// Segments (optional, see plan_segments).  A frame offers lines x 4 independent recurrences -- one
// wave per SIMD at 16384^2, an eighth of the machine at 2048^2.  A line can be split into segments
// that are swept concurrently: a segment's sweeps start `halo` steps before (causal) / after
// (anticausal) it, from the start-up the reference uses at a line's edge.  The recurrences' poles
// are exp(-1.783/sigma) and exp(-1.723/sigma) (gauss.c:57-58), so after halo = 22.7 sigma steps the
// influence of the different start has decayed by e^-39 ~ 1e-17 before the first value that is
// kept.  The true ends of a line keep the reference's start-up.
template <class Src>
__global__ void __launch_bounds__(256) k_iir_causal(Src in, double *__restrict__ ckpt, LineArgs g, IirCoef c) {
    const unsigned sgi = blockIdx.x / g.lane_blocks, lb = blockIdx.x % g.lane_blocks;
    const long L = (long)lb * 256 + threadIdx.x;
    const long stride = (long)g.lines * 4;
    if (L >= stride) return;
    const int s0 = (int)sgi * g.seg, s1 = min(g.n, s0 + g.seg), k0 = max(0, s0 - g.halo);
    causal_sweep(in.for_lane(L), ckpt, (unsigned)stride, (unsigned)L, k0, s0, s1, c);
}

template <class Src>
__global__ void __launch_bounds__(256) k_iir_anticausal_T(Src in, const double *__restrict__ ckpt, float *__restrict__ outT,
                                                          LineArgs g, IirCoef c, PackOut po) {
    // wave-private staging tile: 16 lines x IIR_U steps x 4 channels, line stride padded by 4 floats
    __shared__ float tile[4][16 * (IIR_U * 4 + 4)];
    const unsigned sgi = blockIdx.x / g.lane_blocks, lb = blockIdx.x % g.lane_blocks;
    const long L = (long)lb * 256 + threadIdx.x;
    const long stride = (long)g.lines * 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long line0 = ((long)lb * 256 + (long)wave * 64) >> 2;   // first line of this wave
    const bool active = L < stride;
    const long Lc = active ? L : 0;      // idle lanes of the last wave shadow lane 0; their tile rows are never written out
    const int s0 = (int)sgi * g.seg, s1 = min(g.n, s0 + g.seg), k1 = s1 < g.n ? min(g.n, s1 + g.halo) : g.n;
    anticausal_sweep(in.for_lane(Lc), ckpt, (unsigned)stride, (unsigned)Lc, g.n, s0, s1, k1, c, tile[wave], lane, line0, g.lines, outT, po, active);
}

// How to split lines of n steps, `lines` of them, for a recurrence of standard deviation sigma.
// Off by default -- one segment, every value identical to the reference's: a warmed-up segment start
// agrees with the full sweep only to the recurrence's own rounding-noise floor (~1e-14 relative in
// f64: rounding differences are re-amplified by the 4th-order recurrence and never die out
// completely), so 1e-7 (sigma 7 px) to 3e-6 (sigma 20 px) of the float32 results then round the other
// way by one ulp.  MMHIP_GAUSS_SEGMENTS=auto picks the count by the model below (a sweep costs segment
// + halo steps, at a rate set by the waves a SIMD holds: 1024 SIMDs, and a second wave hides the
// first one's scalar and memory instructions), =N forces N.  Measured at 16384^2, sigma 20 px: 8.85 ->
// 8.13 ms with two segments; the gain is large only for frames too small to give every SIMD a wave.
LineArgs plan_segments(int n, int lines, float sigma) {
    LineArgs g{n, lines, ((n + IIR_U - 1) / IIR_U) * IIR_U, 0, (unsigned)(((long)lines * 4 + 255) / 256)};
    const char *env = getenv("MMHIP_GAUSS_SEGMENTS");
    if (!env || !*env) return g;
    const int forced = strcmp(env, "auto") ? atoi(env) : 0;
    if (strcmp(env, "auto") && forced <= 1) return g;
    const int halo = (((int)ceil(22.7 * (double)sigma) + 2 + IIR_U - 1) / IIR_U) * IIR_U;
    const double waves = (double)g.lane_blocks * 4.0;
    double best = 0.0;
    for (int ns = 1; ns <= 64; ++ns) {
        const int seg = (((n + ns - 1) / ns + IIR_U - 1) / IIR_U) * IIR_U;
        if (ns > 1 && (seg < halo || (long)seg * (ns - 1) >= n)) break;     // too short to pay / empty last segment
        const double per_simd = ceil(waves * ns / 1024.0);
        const double cost = (double)(seg + (ns > 1 ? halo : 0)) * std::max(1.0, 0.53 * per_simd);
        if (forced > 0 ? ns == forced : (ns == 1 || cost < best * 0.97)) {
            best = cost;
            g.seg = seg;
            g.halo = ns > 1 ? halo : 0;
        }
    }
    return g;
}

unsigned segment_count(const LineArgs &g) { return (unsigned)((g.n + g.seg - 1) / g.seg); }

// ---- K5: FIR path for sigma < 0.5 px on either axis (gauss.c:264-639) ----------------------------
// make_rle_curve on the host (double exp, float taps), then per axis: a statistics kernel
// (how many elements of a line repeat their successor -> run_length_encode's `same`) and a
// stencil kernel that evaluates either do_full_lre or do_encoded_lre -- including the
// latter's int-truncated cumulative sums (gauss.c:405) -- per line, like the reference's
// `same > 3n/4` switch.  Lines are edge-replicated (gauss.c:333-375).
struct FirArgs { int w, h, vertical, length; float total; int ctotal; };

// gauss.c:264-306
void make_rle_curve(double sigma, std::vector<float> &curve, std::vector<float> &sum, int &length, float &total) {
    const double sigma2 = 2 * sigma * sigma;
    const double l = sqrt(-sigma2 * log(1.0 / 255.0));
    int n = (int)(ceil(l) * 2);
    if ((n % 2) == 0) n += 1;
    length = n / 2;
    curve.assign(n, 0.f);            // curve[length + i], i in [-length, length]
    curve[length] = 1.0f;
    for (int i = 1; i <= length; i++) {
        float temp = (float)exp(-(i * i) / sigma2);
        curve[length - i] = temp;
        curve[length + i] = temp;
    }
    sum.assign(2 * length + 1, 0.f); // sum[length + i]
    for (int i = 1; i <= length * 2; i++) sum[i] = curve[i - 1] + sum[i - 1];
    total = sum[2 * length] - sum[0];
}

__device__ __forceinline__ long fir_index(const FirArgs &g, int line, int k, int ch) {
    return g.vertical ? ((long)k * g.w + line) * 4 + ch : ((long)line * g.w + k) * 4 + ch;
}

__global__ void __launch_bounds__(256) k_fir_same(const float *__restrict__ in, int *__restrict__ flags, FirArgs g) {
    const long L = (long)blockIdx.x * 256 + threadIdx.x;
    const int lines = g.vertical ? g.w : g.h, n = g.vertical ? g.h : g.w;
    if (L >= (long)lines * 4) return;
    const int line = (int)(L >> 2), ch = (int)(L & 3);
    float last = in[fir_index(g, line, n - 1, ch)];
    int same = 0;
    for (int k = n - 1; k >= 0; --k) {
        const float c = in[fir_index(g, line, k, ch)];
        if (c == last) same++;
        else last = c;
    }
    flags[L] = same > (3 * n) / 4;
}

__global__ void __launch_bounds__(256) k_fir_apply(const float *__restrict__ in, float *__restrict__ out,
                                                   const int *__restrict__ flags, const float *__restrict__ curve,
                                                   const float *__restrict__ csum, FirArgs g) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)g.w * g.h * 4) return;
    const int ch = (int)(e & 3);
    const long px = e >> 2;
    const int col = (int)(px % g.w), row = (int)(px / g.w);
    const int line = g.vertical ? col : row, k = g.vertical ? row : col;
    const int n = g.vertical ? g.h : g.w;
    const int length = g.length;
    const float *c = curve + length, *cs = csum + length;     // centred
    auto pix = [&](int i) {                                    // edge-replicated line element
        int kk = k + i;
        kk = kk < 0 ? 0 : (kk >= n ? n - 1 : kk);
        return in[fir_index(g, line, kk, ch)];
    };
    float val = 0.0f;
    if (!flags[line * 4 + ch]) {
        // do_full_lre, gauss.c:422-498
        val += pix(0) * c[0];
        for (int i = 1; i <= length; ++i) val += (pix(i) + pix(-i)) * c[i];
        val = val / g.total;
    } else {
        // do_encoded_lre, gauss.c:380-420: walk the runs of equal values inside the window
        int pos = -length;                 // window-relative position of the current run start
        auto runlen = [&](int p) {         // equal elements from p onward (as far as the buffer goes)
            const float v = pix(p);
            int r = 1;
            while (k + p + r < n + length && pix(p + r) == v) ++r;
            return r;
        };
        float s1 = cs[-length];
        int nb = runlen(pos);
        int i = -length + nb;
        while (i <= length) {
            const int s2 = (int)cs[i];
            val += pix(pos) * (s2 - s1);
            s1 = (float)s2;
            pos += nb;
            nb = runlen(pos);
            i += nb;
        }
        val += pix(pos) * (cs[length] - s1);
        val = val / g.ctotal;
    }
    out[e] = val;
}

int gauss_rle(float *map, float *tmp, int w, int h, float hs, float vs, NativeWorkspace &ws, char *aux, hipStream_t s,
              std::string *err) {
    // aux: device scratch for flags (max(w,h)*4 ints) + curve + csum
    float *src = map, *dst = tmp;
    for (int pass = 0; pass < 2; ++pass) {
        const int vertical = pass == 0;
        const float sd = vertical ? vs : hs;
        if (!(sd > 0.0f)) continue;
        std::vector<float> curve, sum;
        int length;
        float total;
        make_rle_curve(sd, curve, sum, length, total);
        const int lines = vertical ? w : h;
        int *flags = (int *)aux;
        float *d_curve = (float *)(aux + (size_t)std::max(w, h) * 4 * sizeof(int));
        float *d_sum = d_curve + curve.size();
        if (hipMemcpyAsync(d_curve, curve.data(), curve.size() * 4, hipMemcpyHostToDevice, s) != hipSuccess ||
            hipMemcpyAsync(d_sum, sum.data(), sum.size() * 4, hipMemcpyHostToDevice, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) {      // the host vectors die at the end of this iteration
            *err = "gaussian_blur: tap upload failed";
            return -1;
        }
        FirArgs g{w, h, vertical, length, total, (int)total};
        k_fir_same<<<(unsigned)(((long)lines * 4 + 255) / 256), 256, 0, s>>>(src, flags, g);
        k_fir_apply<<<(unsigned)(((long)w * h * 4 + 255) / 256), 256, 0, s>>>(src, dst, flags, d_curve, d_sum, g);
        std::swap(src, dst);
    }
    if (src != map && hipMemcpyAsync(map, src, (size_t)w * h * 16, hipMemcpyDeviceToDevice, s) != hipSuccess) {
        *err = "gaussian_blur: copy failed";
        return -1;
    }
    (void)ws;
    return 0;
}

int gaussian_blur(const HNativeRec &rec, const std::vector<HImageDesc> &images, int rw, int rh, float *out_map,
                  NativeWorkspace &ws, hipStream_t s, std::string *err, int *rows_lo, int *rows_hi, NativeDirectOut *direct) {
    const int row_lo = *rows_lo, row_hi = *rows_hi;
    *rows_lo = 0;             // paths that fill the whole map leave it so; the windowed IIR path narrows it
    *rows_hi = rh;
    const HImage &img = rec.args[0].img;
    float hdev = rec.args[1].f, vdev = rec.args[2].f;
    if (img.idx < 0 || img.idx >= (int)images.size()) { *err = "gaussian_blur: input is not a bitmap image"; return -1; }
    const HImageDesc &in = images[img.idx];
    int w = rw, h = rh;
    if (in.kind == IMG_FLOATMAP) {
        w = in.w;
        h = in.h;
        if (w != rw || h != rh) { *err = "gaussian_blur: float-map input of a different size is not supported"; return -1; }
    } else if (in.kind != IMG_DRAWABLE) {
        *err = "gaussian_blur: input image is not bound";
        return -1;
    }
    // gauss.c:659-660 (float products, fabs)
    const float ax = (float)((float)(w - 1) / 2.0);
    float ay = (float)((float)(h - 1) / 2.0);
    ay *= -1.0f;
    float hs = (float)fabs(hdev * ax), vs = (float)fabs(vdev * ay);
    // workspace: the scans' checkpoints (4 doubles per line*channel every IIR_U steps, the larger of
    // the two passes; also covers the FIR path's flags and taps) followed by the intermediate map
    const size_t ck_v = (size_t)((h + IIR_U - 1) / IIR_U) * 4 * ((size_t)w * 4) * sizeof(double);
    const size_t ck_h = (size_t)((w + IIR_U - 1) / IIR_U) * 4 * ((size_t)h * 4) * sizeof(double);
    const size_t scratch_bytes = ((std::max(ck_v, ck_h) + (size_t)std::max(w, h) * 64 + (size_t)(w + h) * sizeof(int) + 65536) + 255) & ~(size_t)255;
    const size_t map_bytes = (size_t)w * h * 4 * sizeof(float);
    char *wsp = (char *)ws.reserve(scratch_bytes + map_bytes);
    if (!wsp) { *err = "gaussian_blur: out of device memory for the scan workspace"; return -1; }
    double *scratch = (double *)wsp;
    float *mapT = (float *)(wsp + scratch_bytes);
    // materialise the input map only where a kernel needs it whole (the FIR path)
    auto render_input = [&]() -> int {
        if (in.kind == IMG_FLOATMAP) {
            if (hipMemcpyAsync(out_map, in.data, (size_t)w * h * 16, hipMemcpyDeviceToDevice, s) != hipSuccess) {
                *err = "gaussian_blur: copy failed";
                return -1;
            }
        } else {
            render_drawable_impl(in, img, ws.env, out_map, w, h, s);
        }
        return 0;
    };
    if (hs < 0.5f || vs < 0.5f) {     // gauss.c:662-665
        if (render_input() != 0) return -1;
        return gauss_rle(out_map, mapT, w, h, hs, vs, ws, wsp, s, err);
    }
    IirCoef c;
    // Does render_image map output pixel (x, y) to texel (x, y)?  Evaluated on the host with the
    // operations of k_render_drawable (IEEE float: same results), once per row and column.
    bool identity = in.kind == IMG_DRAWABLE && in.w == w && in.h == h;
    if (identity) {
        const float ax = (float)((float)(w - 1) / 2.0), bx = ax;
        const float by = (float)((float)(h - 1) / 2.0);
        const float ay2 = (float)(by * -1.0);
        for (int i = 0; i < w && identity; ++i) {
            float x = ((float)i - bx) / ax;
            if (img.resized) x *= img.xf;
            x = (x + in.middle_x) * in.scale_x;
            if (!ws.env.supersampling) x += 0.5;
            identity = (int)floor((double)x) == i;
        }
        for (int i = 0; i < h && identity; ++i) {
            float y = ((float)i - by) / ay2;
            if (img.resized) y *= img.yf;
            y = -((y - in.middle_y) * in.scale_y);
            if (!ws.env.supersampling) y += 0.5;
            identity = (int)floor((double)y) == i;
        }
    }
    if (in.kind == IMG_DRAWABLE && !identity && render_input() != 0) return -1;
    // Row window: a caller that only reads rows [row_lo, row_hi) (one GPU's stripe of a striped
    // frame; inputs are replicated on every GPU, so the halo is computed locally, not exchanged)
    // gets those rows plus a halo of ceil(22.7 sigma_v) rows -- the recurrences' poles are
    // exp(-1.783/sigma) and exp(-1.723/sigma) (gauss.c:57-58), so the different start-up at the
    // window edge has decayed below 1e-17 of the value, under half an ulp of the f64 sums, by the
    // time the sweep reaches a row that is read.  True image edges keep the reference's start-up.
    int y0 = 0, y1 = h;
    if (row_lo > 0 || row_hi < h) {
        const int halo = (int)ceil(22.7 * (double)vs) + 2;
        y0 = std::max(0, row_lo - halo);
        y1 = std::min(h, row_hi + halo);
        *rows_lo = std::max(0, row_lo);        // valid: the rows asked for (the halo rows are scratch)
        *rows_hi = std::min(h, row_hi);
        if (y1 <= y0) return 0;
    }
    const int hn = y1 - y0;
    // the scan kernels index a block of rows with 32-bit element offsets (MapSrc::at)
    if ((long)std::max(w, hn) * 4 * 20 * 8 >= (1L << 32)) { *err = "gaussian_blur: frame too large for the scan kernels"; return -1; }
    // vertical pass first (gauss.c:155-201): lines = columns, n = rows of the window; result transposed
    // into mapT[w][hn][4].  Its two sweeps read the input where it lies: the float map, or --
    // identity mapping -- the drawable itself: no intermediate map, 4 instead of 16 B/px.
    find_iir_constants(c, vs);
    {
        const LineArgs g = plan_segments(hn, w, vs);
        const unsigned blocks = g.lane_blocks * segment_count(g);
        if (in.kind == IMG_FLOATMAP || !identity) {
            const MapSrc src{(in.kind == IMG_FLOATMAP ? (const float *)in.data : out_map) + (long)y0 * w * 4, (unsigned)w * 4u, 0u};
            ws.timed_launch("iir_causal_vertical", s, [&] { k_iir_causal<<<blocks, 256, 0, s>>>(src, scratch, g, c); });
            ws.timed_launch("iir_anticausal_vertical", s, [&] { k_iir_anticausal_T<<<blocks, 256, 0, s>>>(src, scratch, mapT, g, c, PackOut{nullptr, 0, 0, 0, 0, 0, 1}); });
        } else {
            const DrawableSrc src{(const uint32_t *)in.data + (long)y0 * in.w, (unsigned)in.w, 0u, 0};
            ws.timed_launch("iir_causal_vertical", s, [&] { k_iir_causal<<<blocks, 256, 0, s>>>(src, scratch, g, c); });
            ws.timed_launch("iir_anticausal_vertical", s, [&] { k_iir_anticausal_T<<<blocks, 256, 0, s>>>(src, scratch, mapT, g, c, PackOut{nullptr, 0, 0, 0, 0, 0, 1}); });
        }
    }
    // horizontal pass (gauss.c:203-252): in mapT the window's rows are the "columns"; transposing again
    // restores the original layout, written to rows [y0, y1) of out_map
    find_iir_constants(c, hs);
    {
        const LineArgs g = plan_segments(w, hn, hs);
        const unsigned blocks = g.lane_blocks * segment_count(g);
        // lines of this pass are rows of the window [y0, y1), steps are columns
        PackOut po{nullptr, 0, 0, 0, 0, 0, 1};
        if (direct && direct->out && direct->first_row >= y0 && direct->first_row + direct->num_rows <= y1) {
            po = PackOut{(unsigned char *)direct->out, (long)direct->row_stride, direct->first_row - y0,
                         direct->first_row + direct->num_rows - y0, direct->region_x, direct->region_x + direct->region_w,
                         direct->skip_map ? 0 : 1};
            direct->written = true;
        }
        if (in.kind == IMG_FLOATMAP) {
            const MapSrc src{mapT, (unsigned)hn * 4u, 0u};
            ws.timed_launch("iir_causal_horizontal", s, [&] { k_iir_causal<<<blocks, 256, 0, s>>>(src, scratch, g, c); });
            ws.timed_launch("iir_anticausal_horizontal", s, [&] { k_iir_anticausal_T<<<blocks, 256, 0, s>>>(src, scratch, out_map + (long)y0 * w * 4, g, c, po); });
        } else {      // the first pass read bytes: its output is finite
            const FiniteMapSrc src{mapT, (unsigned)hn * 4u, 0u};
            ws.timed_launch("iir_causal_horizontal", s, [&] { k_iir_causal<<<blocks, 256, 0, s>>>(src, scratch, g, c); });
            ws.timed_launch("iir_anticausal_horizontal", s, [&] { k_iir_anticausal_T<<<blocks, 256, 0, s>>>(src, scratch, out_map + (long)y0 * w * 4, g, c, po); });
        }
    }
    if (hipGetLastError() != hipSuccess) { *err = "gaussian_blur: kernel launch failed"; return -1; }
    return 0;
}

}  // namespace

void launch_render_drawable(const HImageDesc &in, const HImage &img, const NativeEnv &env, float *dst, int w, int h, hipStream_t s) {
    render_drawable_impl(in, img, env, dst, w, h, s);
}

// The argument image of a native filter as a float map of w x h pixels: its own data when
// it already is one of that size, otherwise render_image into `dst` (convolve.c:88-95).
int native_input_map(const char *who, const HImage &img, const std::vector<HImageDesc> &images, const NativeEnv &env, int w,
                     int h, float *dst, const float **map, hipStream_t s, std::string *err) {
    if (img.idx < 0 || img.idx >= (int)images.size()) { *err = std::string(who) + ": input is not a bitmap image"; return -1; }
    const HImageDesc &in = images[img.idx];
    if (in.kind == IMG_FLOATMAP && in.w == w && in.h == h) { *map = (const float *)in.data; return 0; }
    if (in.kind != IMG_DRAWABLE) { *err = std::string(who) + ": input image is not bound (or a float map of another size)"; return -1; }
    launch_render_drawable(in, img, env, dst, w, h, s);
    *map = dst;
    return 0;
}

// Supersampling combine of call_invocation (mathmap_common.c:880-927): per output byte
// (l1[c] + l1[c+1] + 2*l2[c] + l3[c] + l3[c+1]) / 6 in integer arithmetic, where l1/l3 are
// rows r and r+1 of the "long" slice (width+1, offset -0.5) and l2 row r of the short slice.
__global__ void __launch_bounds__(256) k_supersample_combine(const unsigned char *__restrict__ longs,
                                                             const unsigned char *__restrict__ shorts,
                                                             unsigned char *__restrict__ out, int w, int h, int bpp,
                                                             int out_stride) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)w * h * bpp) return;
    const int b = (int)(i % bpp);
    const long px = i / bpp;
    const int col = (int)(px % w), row = (int)(px / w);
    const long lstride = (long)(w + 1) * bpp;
    const unsigned char *l1 = longs + row * lstride;
    // the reference's last iteration renders no new line3 (calc_lines clips at the slice end),
    // so line3 still equals line1 there
    const unsigned char *l3 = longs + (row + 1 < h ? row + 1 : row) * lstride;
    const unsigned char *l2 = shorts + (long)row * w * bpp;
    const int v = (l1[col * bpp + b] + l1[(col + 1) * bpp + b] + 2 * l2[col * bpp + b] + l3[col * bpp + b] +
                   l3[(col + 1) * bpp + b]) / 6;
    out[(long)row * out_stride + (long)col * bpp + b] = (unsigned char)v;
}

void launch_supersample_combine(const unsigned char *longs, const unsigned char *shorts, unsigned char *out, int w, int h,
                                int bpp, int out_stride, hipStream_t s) {
    const long n = (long)w * h * bpp;
    k_supersample_combine<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(longs, shorts, out, w, h, bpp, out_stride);
}

int run_native_filter(const std::string &func, const HNativeRec &rec, const std::vector<HImageDesc> &images,
                      int render_w, int render_h, float *out_map, NativeWorkspace &ws, hipStream_t stream,
                      std::string *err, int *row_lo, int *row_hi, NativeDirectOut *direct) {
    if (func == "native_filter_gaussian_blur")
        return gaussian_blur(rec, images, render_w, render_h, out_map, ws, stream, err, row_lo, row_hi, direct);
    *row_lo = 0;              // every other native filter produces the whole map
    *row_hi = render_h;
    if (func == "RENDER") {   // render_image (builtins.c:267-346), drawable / float-map branches
        const HImage &img = rec.args[0].img;
        if (img.idx < 0 || img.idx >= (int)images.size()) { *err = "render(): rendering a filter closure is not supported by the HIP backend yet"; return -1; }
        const HImageDesc &in = images[img.idx];
        const int w = render_w, h = render_h;
        if (in.kind == IMG_FLOATMAP) {
            if (!img.resized) {       // builtins.c:273-274: a plain float map is the result itself (here: its copy)
                if (in.w != w || in.h != h) { *err = "render(): float-map input of a different size"; return -1; }
                if (hipMemcpyAsync(out_map, in.data, (size_t)w * h * 16, hipMemcpyDeviceToDevice, stream) != hipSuccess) { *err = "render(): copy failed"; return -1; }
                return 0;
            }
            // behind a resize wrapper -- what filter code hands to render() -- the image is of type IMAGE_RESIZE, not
            // IMAGE_FLOATMAP: render_image samples it into a new map like it does a drawable
            if ((const void *)in.data == (const void *)out_map) { *err = "render(): internal: source and result share a map"; return -1; }
            const long n = (long)w * h;
            k_render_floatmap<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>((const float4 *)in.data, in.w, in.h, in.ax, in.bx, in.ay, in.by,
                                                                            img.xf, img.yf, (float4 *)out_map, w, h);
            return 0;
        }
        if (in.kind != IMG_DRAWABLE) { *err = "render(): input image is not bound"; return -1; }
        launch_render_drawable(in, img, ws.env, out_map, w, h, stream);
        return 0;
    }
    if (func == "native_filter_convolve" || func == "native_filter_half_convolve" || func == "native_filter_visualize_fft")
        return fft_native_filter(func, rec, images, render_w, render_h, out_map, ws, stream, err);
    *err = "native filter " + func + " is not implemented in the HIP backend";
    return -1;
}

}  // namespace mm
