// Device-side runtime of JIT-compiled MathMap pixel kernels (gfx950).
//
// This file is prepended verbatim to every kernel string handed to hiprtc
// (hipgen.cpp).  It restates, as __device__ code, the semantics of the reference's
// op macros (opmacros.h:30-218), colour helpers (color.h:36-54) and pixel fetch
// runtime (builtins/builtins.c:40-265).  Arithmetic rules that must hold for
// parity with the cc backend:
//   * the generated code is compiled with -ffp-contract=off (gcc for baseline
//     x86-64 never fuses a*b+c);
//   * real libm ops are the *double* functions applied to the promoted operand
//     (C has no overloads: sin(float) calls sin(double)), rounded once on
//     assignment -- hence the mm_* wrappers taking double;
//   * float literals are printed as double literals of the float value, so C's
//     usual arithmetic conversions decide float-vs-double exactly as in gcc.
//
// Compile-time switches provided by the generator (#define before this text):
//   MM_INTERSAMPLE   1 = bilinear fetch (get_orig_val_intersample_pixel), 0 = nearest
//   MM_SUPERSAMPLING 1 = no +0.5 in the nearest fetch (builtins.c:154-158)
//   MM_EDGE_X / MM_EDGE_Y   edge behaviour: 0 colour, 1 wrap, 2 reflect, 3 rotate
#ifndef MM_DEVICE_H
#define MM_DEVICE_H

#ifndef MM_INTERSAMPLE
#define MM_INTERSAMPLE 0
#endif
#ifndef MM_SUPERSAMPLING
#define MM_SUPERSAMPLING 0
#endif
#ifndef MM_EDGE_X
#define MM_EDGE_X 0
#endif
#ifndef MM_EDGE_Y
#define MM_EDGE_Y 0
#endif

#define MM_DEV static __device__ __forceinline__

typedef unsigned int color_t;

struct mm_complex { float re, im; };
template <int N> struct mm_tup { float v[N]; };

// An image *value* inside a kernel: index into the image table plus the resize
// factors of an IMAGE_RESIZE wrapper (drawable.c:211-228).
struct mm_image { int idx; int pw; int ph; float xf; float yf; int resized; };

enum { MM_IMG_DRAWABLE = 0, MM_IMG_FLOATMAP = 1, MM_IMG_NULL = 2 };

// One entry of the image table in HBM (input drawables and float maps).
struct mm_image_desc {
    const void *data;      // drawable: color_t[h][w] (0xRRGGBBAA); floatmap: float[h][w][4]
    int w, h;
    int kind;
    int num_frames;
    float scale_x, scale_y, middle_x, middle_y;   // userval.c:262-280
    float ax, bx, ay, by;                          // floatmap.c:30-46
};

union mm_userval { int i; float f; color_t c; int image; };

// Everything a kernel needs about the invocation / frame / slice
// (mathmap.h:162-226).  Passed by value => lives in SGPRs.
struct mm_args {
    int img_width, img_height;        // __canvasPixelW/H
    int render_width, render_height;  // __renderPixelW/H
    int frame_render_width, frame_render_height;
    float t;
    int frame;
    float R;
    int region_x, region_y, region_width, region_height;
    float sampling_offset_x, sampling_offset_y;
    int first_row, num_rows;          // rows of the slice rendered by this launch
    int output_bpp;
    int row_stride;                   // bytes (u8 output) between output rows
    int floatmap;                     // 1: write float4 instead of bytes
    color_t edge_color_x, edge_color_y;
    const mm_userval *uservals;
    const mm_image_desc *images;
    const void *curves;               // float[n][1024]
    const void *gradients;            // color_t[n][1024]
    void *out;
    int native_slot_base;             // image-table index of native-filter result 0
    int ppt;                          // pixels (rows, MM_TILE_H apart) each work-item renders
    // per-column x and per-row y virtual coordinates of this launch, filled by the
    // prologue kernel (the reference's y_vars / per-row x-const code,
    // new_template.c.in:245,260,357-372): the double divide is done once per row and
    // column instead of once per pixel.
    float *xtab;
    float *ytab;
    unsigned tiles_magic;             // workgroup id / tile columns by multiply-high (mm_host_abi.h), 0: divide
    unsigned num_images;              // entries of `images` (a handle beyond it -- a value no path of the frame-constant code assigned -- reads as no image)
    // values that depend on the row alone (the reference's x-const slice, new_template.c.in:251-253), computed once per
    // row of the launch by mm_rows: value k of row r at rowtab[k * num_rows + r] (ints stored as their bit pattern)
    float *rowtab;
};

// ---- op macros (opmacros.h:30-47) --------------------------------------------------
#define NOP() (0.0)
#define INT2FLOAT(x) ((float)(x))
// C's float -> int conversion is undefined outside the int range; the reference runs on x86-64,
// where cvttss2si returns INT_MIN ("integer indefinite") for NaN and out-of-range values, while
// v_cvt_i32_f32 saturates and turns NaN into 0.  A pixel whose coordinate is NaN (0/0 at a
// filter's singular point) must land outside the image like it does there, not on texel (0, 0).
MM_DEV int mm_f2i(float x) { return (x >= -2147483648.0f && x < 2147483648.0f) ? (int)x : (int)0x80000000; }
#define FLOAT2INT(x) (mm_f2i((float)(x)))
#define ADD(a, b) ((a) + (b))
#define SUB(a, b) ((a) - (b))
#define NEG(a) (-(a))
#define MUL(a, b) ((a) * (b))
#define DIV(a, b) ((float)(a) / (float)(b))
#define MOD(a, b) (mm_fmod((a), (b)))
#define EQ(a, b) ((a) == (b))
#define LESS(a, b) ((a) < (b))
#define LEQ(a, b) ((a) <= (b))
#define NOT(a) (!(a))
#define MIN(a, b) (((a) < (b)) ? (a) : (b))
#define MAX(a, b) (((a) < (b)) ? (b) : (a))
#define CLAMP01(x) (MAX(0, MIN(1, (x))))

// double libm, applied to promoted operands
MM_DEV double mm_fmod(double a, double b) { return fmod(a, b); }
MM_DEV double mm_fabs(double a) { return fabs(a); }
MM_DEV double mm_sqrt(double a) { return sqrt(a); }
MM_DEV double mm_hypot(double a, double b) { return hypot(a, b); }
// hypot of two floats widened to double: the squares are exact (48 bits), they cannot overflow
// or underflow (|x| < 2^128), so sqrt(x*x + y*y) needs none of the scaling a general double hypot
// carries; one rounding in the sum and a correctly rounded sqrt keep it within 0.75 ulp (OCML's
// hypot: 1 ulp).  Inf / NaN operands take the library function (hypot(inf, nan) = inf).
MM_DEV double mm_hypot_ff(double a, double b) {
    const double s = a * a + b * b;
    if (!(s <= 1.7976931348623157e308)) return hypot(a, b);
    return __builtin_sqrt(s);
}
MM_DEV double mm_sin(double a) { return sin(a); }
MM_DEV double mm_cos(double a) { return cos(a); }
MM_DEV double mm_tan(double a) { return tan(a); }
MM_DEV double mm_asin(double a) { return asin(a); }
MM_DEV double mm_acos(double a) { return acos(a); }
MM_DEV double mm_atan(double a) { return atan(a); }
MM_DEV double mm_atan2(double a, double b) { return atan2(a, b); }
MM_DEV double mm_pow(double a, double b) { return pow(a, b); }
MM_DEV double mm_exp(double a) { return exp(a); }
MM_DEV double mm_log(double a) { return log(a); }
MM_DEV double mm_sinh(double a) { return sinh(a); }
MM_DEV double mm_cosh(double a) { return cosh(a); }
MM_DEV double mm_tanh(double a) { return tanh(a); }
MM_DEV double mm_asinh(double a) { return asinh(a); }
MM_DEV double mm_acosh(double a) { return acosh(a); }
MM_DEV double mm_atanh(double a) { return atanh(a); }
MM_DEV double mm_floor(double a) { return floor(a); }
MM_DEV double mm_ceil(double a) { return ceil(a); }
// floor / ceil assigned to an int variable: the C assignment converts the double with cvttsd2si
MM_DEV int mm_d2i(double x) { return (x >= -2147483648.0 && x < 2147483648.0) ? (int)x : (int)0x80000000; }
MM_DEV int mm_floor_i(double a) { return mm_d2i(floor(a)); }
MM_DEV int mm_ceil_i(double a) { return mm_d2i(ceil(a)); }
// GSL / GLib operators (mm_gslmath.h, restated; parity unpinned)
MM_DEV mm_tup<2> mm_solve_linear_2(const mm_tup<4> &m, const mm_tup<2> &v) {
    double A[4], x[2];
    for (int i = 0; i < 4; ++i) A[i] = m.v[i];
    x[0] = v.v[0]; x[1] = v.v[1];
    mmg_hh_svx(2, A, x);
    mm_tup<2> r;
    r.v[0] = (float)x[0]; r.v[1] = (float)x[1];
    return r;
}
MM_DEV mm_tup<3> mm_solve_linear_3(const mm_tup<9> &m, const mm_tup<3> &v) {
    double A[9], x[3];
    for (int i = 0; i < 9; ++i) A[i] = m.v[i];
    for (int i = 0; i < 3; ++i) x[i] = v.v[i];
    mmg_hh_svx(3, A, x);
    mm_tup<3> r;
    for (int i = 0; i < 3; ++i) r.v[i] = (float)x[i];
    return r;
}
MM_DEV mm_tup<3> mm_ell_jac(double u, double m) {
    double sn, cn, dn;
    mmg_elljac(u, m, &sn, &cn, &dn);
    mm_tup<3> r;
    r.v[0] = (float)sn; r.v[1] = (float)cn; r.v[2] = (float)dn;
    return r;
}
#define SOLVE_LINEAR_2(m, v) (mm_solve_linear_2((m), (v)))
#define SOLVE_LINEAR_3(m, v) (mm_solve_linear_3((m), (v)))
#define ELL_JAC(u, m) (mm_ell_jac((u), (m)))
#define ELL_INT_K_COMP(k) (mmg_ellint_Kcomp((k)))
#define ELL_INT_E_COMP(k) (mmg_ellint_Ecomp((k)))
#define ELL_INT_F(phi, k) (mmg_ellint_F((phi), (k)))
#define ELL_INT_E(phi, k) (mmg_ellint_E((phi), (k)))
#define ELL_INT_P(phi, k, n) (mmg_ellint_P((phi), (k), (n)))
#define ELL_INT_D(phi, k, n) (mmg_ellint_D((phi), (k)))      /* GSL >= 2: no n (opmacros.h:108-112) */
#define ELL_INT_RC(x, y) (mmg_ellint_RC((x), (y)))
#define ELL_INT_RD(x, y, z) (mmg_ellint_RD((x), (y), (z)))
#define ELL_INT_RF(x, y, z) (mmg_ellint_RF((x), (y), (z)))
#define ELL_INT_RJ(x, y, z, p) (mmg_ellint_RJ((x), (y), (z), (p)))
// g_random_double_range(a, b): u * (b - a) + a in double.  `col`, `rl`, `mm_rand_ctr` are the pixel
// kernel's locals: absolute pixel position, so the value does not depend on stripes or tiles.
#define RAND(a, b) \
    (mmg_rand_unit(col + A.region_x, A.first_row + rl, A.frame, mm_rand_ctr++) * ((double)(b) - (double)(a)) + (double)(a))

MM_DEV double mm_gamma(double a) { return (a > 171.0) ? 0.0 : tgamma(a); }   // opmacros.h:43
MM_DEV double mm_beta(double a, double b) { return exp(lgamma(a) + lgamma(b) - lgamma(a + b)); }

// (float)sqrt((double)x) == correctly rounded sqrtf(x) for every float x (53 >= 2*24+2
// bits: double rounding is innocuous for sqrt), so float operands may take the f32 path.
MM_DEV float mm_sqrt_f32(float a) { return sqrtf(a); }   // correctly rounded (v_sqrt_f32 + fma fix-up); __fsqrt_rn is the 1-ulp form
// sqrt_rn(a) < K for K a power of two  <=>  0 <= a < K*K : sqrt is monotone, sqrt(K*K) = K
// exactly, and sqrt(pred(K*K)) = K*sqrt(1 - 2^-24) = K*(1 - 2^-25 - ...) lies below the
// midpoint of pred(K) and K, so it rounds to pred(K) < K.  NaN and negative a give false on
// both sides (-0 gives true on both).  Saves the sqrt in escape-time tests `abs(z) < 2`.
#define MM_SQRT_LESS_POW2(a, k2) (((a) < (k2)) && ((a) >= 0.0f))

// ---- two pixels in lockstep (hipgen.cpp pair mode) ---------------------------------------------
// Values of the pixel slice as 2-vectors; mm_vf / mm_vi make an operand a vector of the wanted type
// (broadcasting frame constants and literals, converting int to float exactly like the C cast).
typedef float mm_f2 __attribute__((ext_vector_type(2)));
typedef int mm_i2 __attribute__((ext_vector_type(2)));
#ifndef MM_PAIR_SCALAR
#define MM_PAIR_SCALAR 1
#endif
#if MM_PAIR_SCALAR
// the two components as separate scalars: adjacent independent v_*_f32 instead of one v_pk_*_f32
struct mm_pf { float x, y; };
struct mm_pi { int x, y; };
MM_DEV mm_pf operator+(mm_pf a, mm_pf b) { return mm_pf{a.x + b.x, a.y + b.y}; }
MM_DEV mm_pf operator-(mm_pf a, mm_pf b) { return mm_pf{a.x - b.x, a.y - b.y}; }
MM_DEV mm_pf operator*(mm_pf a, mm_pf b) { return mm_pf{a.x * b.x, a.y * b.y}; }
MM_DEV mm_pf operator/(mm_pf a, mm_pf b) { return mm_pf{a.x / b.x, a.y / b.y}; }
MM_DEV mm_pf operator*(mm_pf a, float b) { return mm_pf{a.x * b, a.y * b}; }
MM_DEV mm_pf operator-(mm_pf a) { return mm_pf{-a.x, -a.y}; }
MM_DEV mm_pi operator+(mm_pi a, mm_pi b) { return mm_pi{a.x + b.x, a.y + b.y}; }
MM_DEV mm_pi operator-(mm_pi a, mm_pi b) { return mm_pi{a.x - b.x, a.y - b.y}; }
MM_DEV mm_pi operator*(mm_pi a, mm_pi b) { return mm_pi{a.x * b.x, a.y * b.y}; }
MM_DEV mm_pi operator-(mm_pi a) { return mm_pi{-a.x, -a.y}; }
#else
typedef mm_f2 mm_pf;
typedef mm_i2 mm_pi;
#endif
MM_DEV mm_pf mm_vf(float a) { return mm_pf{a, a}; }
MM_DEV mm_pf mm_vf(int a) { const float f = (float)a; return mm_pf{f, f}; }
MM_DEV mm_pf mm_vf(mm_pf a) { return a; }
MM_DEV mm_pf mm_vf(mm_pi a) { return mm_pf{(float)a.x, (float)a.y}; }
MM_DEV mm_pi mm_vi(int a) { return mm_pi{a, a}; }
MM_DEV mm_pi mm_vi(mm_pi a) { return a; }
#ifndef MM_PAIR_MASKS
#define MM_PAIR_MASKS 1      // measured on Mandelbrot 8192^2: 0.300 ms with bool pairs, 0.268 ms with lane masks
#endif
#if MM_PAIR_MASKS
// Truth values of the pair as explicit lane masks: 64-bit wave-uniform scalars, one per pixel of the pair.
// Comparisons are ballots (the v_cmp's own result), logic is scalar arithmetic, selects read the mask as the
// v_cndmask's lane predicate, and a loop that runs "while either pixel of any lane is active" branches on a scalar.
// Valid because the pair body has no divergent control flow: its ifs are if-converted and its loops wave-uniform,
// so every ballot sees the same lanes (those that passed the kernel's column test).
struct mm_bb { unsigned long x, y; };
MM_DEV unsigned long mm_lanes() { return __builtin_amdgcn_ballot_w64(true); }
MM_DEV mm_bb mm_bu(bool u) { const unsigned long m = u ? mm_lanes() : 0ul; return mm_bb{m, m}; }      // wave-uniform truth value
MM_DEV mm_bb mm_bl(bool x, bool y) { return mm_bb{__builtin_amdgcn_ballot_w64(x), __builtin_amdgcn_ballot_w64(y)}; }
MM_DEV float mm_msel(unsigned long m, float a, float b) { float r; asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m)); return r; }
MM_DEV int mm_msel(unsigned long m, int a, int b) { int r; asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m)); return r; }
MM_DEV mm_pi mm_vi(mm_bb b) { return mm_pi{mm_msel(b.x, 1, 0), mm_msel(b.y, 1, 0)}; }
MM_DEV mm_pf mm_vf(mm_bb b) { return mm_pf{mm_msel(b.x, 1.0f, 0.0f), mm_msel(b.y, 1.0f, 0.0f)}; }
MM_DEV mm_bb mm_tob(mm_pi a) { return mm_bl(a.x != 0, a.y != 0); }
MM_DEV mm_bb mm_notb(mm_bb a) { const unsigned long l = mm_lanes(); return mm_bb{~a.x & l, ~a.y & l}; }
MM_DEV mm_bb mm_andb(mm_bb a, mm_bb b) { return mm_bb{a.x & b.x, a.y & b.y}; }
MM_DEV mm_bb mm_eqb(mm_bb a, mm_bb b) { const unsigned long l = mm_lanes(); return mm_bb{~(a.x ^ b.x) & l, ~(a.y ^ b.y) & l}; }
MM_DEV mm_bb mm_lt(mm_pf a, mm_pf b) { return mm_bl(a.x < b.x, a.y < b.y); }
MM_DEV mm_bb mm_le(mm_pf a, mm_pf b) { return mm_bl(a.x <= b.x, a.y <= b.y); }
MM_DEV mm_bb mm_eq(mm_pf a, mm_pf b) { return mm_bl(a.x == b.x, a.y == b.y); }
MM_DEV mm_bb mm_lt(mm_pi a, mm_pi b) { return mm_bl(a.x < b.x, a.y < b.y); }
MM_DEV mm_bb mm_le(mm_pi a, mm_pi b) { return mm_bl(a.x <= b.x, a.y <= b.y); }
MM_DEV mm_bb mm_eq(mm_pi a, mm_pi b) { return mm_bl(a.x == b.x, a.y == b.y); }
MM_DEV mm_pf mm_sel2(mm_bb c, mm_pf a, mm_pf b) { return mm_pf{mm_msel(c.x, a.x, b.x), mm_msel(c.y, a.y, b.y)}; }
MM_DEV mm_pi mm_sel2(mm_bb c, mm_pi a, mm_pi b) { return mm_pi{mm_msel(c.x, a.x, b.x), mm_msel(c.y, a.y, b.y)}; }
MM_DEV mm_bb mm_sel2(mm_bb c, mm_bb a, mm_bb b) { return mm_bb{(c.x & a.x) | (~c.x & b.x), (c.y & a.y) | (~c.y & b.y)}; }
#else
// truth values (comparison results and their logic) as a pair of bools: scalar lane masks, scalar logic
struct mm_bb { bool x, y; };
MM_DEV mm_bb mm_bu(bool u) { return mm_bb{u, u}; }
MM_DEV mm_bb mm_bl(bool x, bool y) { return mm_bb{x, y}; }
MM_DEV mm_pi mm_vi(mm_bb b) { return mm_pi{b.x ? 1 : 0, b.y ? 1 : 0}; }
MM_DEV mm_pf mm_vf(mm_bb b) { return mm_pf{b.x ? 1.0f : 0.0f, b.y ? 1.0f : 0.0f}; }
MM_DEV mm_bb mm_tob(mm_pi a) { return mm_bb{a.x != 0, a.y != 0}; }
MM_DEV mm_bb mm_notb(mm_bb a) { return mm_bb{!a.x, !a.y}; }
MM_DEV mm_bb mm_andb(mm_bb a, mm_bb b) { return mm_bb{a.x && b.x, a.y && b.y}; }
MM_DEV mm_bb mm_eqb(mm_bb a, mm_bb b) { return mm_bb{a.x == b.x, a.y == b.y}; }
MM_DEV mm_bb mm_lt(mm_pf a, mm_pf b) { return mm_bb{a.x < b.x, a.y < b.y}; }
MM_DEV mm_bb mm_le(mm_pf a, mm_pf b) { return mm_bb{a.x <= b.x, a.y <= b.y}; }
MM_DEV mm_bb mm_eq(mm_pf a, mm_pf b) { return mm_bb{a.x == b.x, a.y == b.y}; }
MM_DEV mm_bb mm_lt(mm_pi a, mm_pi b) { return mm_bb{a.x < b.x, a.y < b.y}; }
MM_DEV mm_bb mm_le(mm_pi a, mm_pi b) { return mm_bb{a.x <= b.x, a.y <= b.y}; }
MM_DEV mm_bb mm_eq(mm_pi a, mm_pi b) { return mm_bb{a.x == b.x, a.y == b.y}; }
MM_DEV mm_pf mm_sel2(mm_bb c, mm_pf a, mm_pf b) { return mm_pf{c.x ? a.x : b.x, c.y ? a.y : b.y}; }
MM_DEV mm_pi mm_sel2(mm_bb c, mm_pi a, mm_pi b) { return mm_pi{c.x ? a.x : b.x, c.y ? a.y : b.y}; }
MM_DEV mm_bb mm_sel2(mm_bb c, mm_bb a, mm_bb b) { return mm_bb{c.x ? a.x : b.x, c.y ? a.y : b.y}; }
#endif
MM_DEV mm_pf mm_sqrt2(mm_pf a) { return mm_pf{mm_sqrt_f32(a.x), mm_sqrt_f32(a.y)}; }

// ---- complex (float _Complex) -----------------------------------------------------------
// In the generated C, complex values only flow COMPLEX() -> c*f() -> crealf/cimagf
// (builtins.lisp:679-941).  All sixteen libm functions of that list run glibc's own float algorithms
// (mm_glibcf.h); cgamma restates the reference's own builtins/spec_func.c below.
MM_DEV mm_complex mm_cmake(float r, float i) { mm_complex c; c.re = r; c.im = i; return c; }
// opmacros.h:63: COMPLEX(r, i) = (r) + (i) * I.  C's I is the complex value (0, 1), so gcc evaluates the real part
// as r + i * 0.0f: NaN for an infinite or NaN imaginary part, and -0 + (+0) = +0 for r = -0 with i >= +0 -- both
// visible behind the branch cuts of clogf / csqrtf.  Same two float operations here.
MM_DEV mm_complex mm_complex_from_parts(float r, float i) { return mm_cmake(r + i * 0.0f, i); }
#define COMPLEX(r, i) mm_complex_from_parts((float)(r), (float)(i))
MM_DEV float crealf(mm_complex c) { return c.re; }
MM_DEV float cimagf(mm_complex c) { return c.im; }

struct mm_dc { double re, im; };
MM_DEV mm_dc mm_dcmake(double r, double i) { mm_dc c; c.re = r; c.im = i; return c; }
MM_DEV mm_dc mm_widen(mm_complex c) { return mm_dcmake((double)c.re, (double)c.im); }
MM_DEV mm_complex mm_narrow(mm_dc c) { return mm_cmake((float)c.re, (float)c.im); }
MM_DEV mm_dc mm_dcmul(mm_dc a, mm_dc b) { return mm_dcmake(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
MM_DEV mm_dc mm_dcadd(mm_dc a, mm_dc b) { return mm_dcmake(a.re + b.re, a.im + b.im); }
MM_DEV mm_dc mm_dcsub(mm_dc a, mm_dc b) { return mm_dcmake(a.re - b.re, a.im - b.im); }
MM_DEV mm_dc mm_dcdiv(mm_dc a, mm_dc b) {
    double d = b.re * b.re + b.im * b.im;
    return mm_dcmake((a.re * b.re + a.im * b.im) / d, (a.im * b.re - a.re * b.im) / d);
}
MM_DEV mm_dc mm_dclog(mm_dc z) { return mm_dcmake(mmf_log_any(hypot(z.re, z.im)), atan2(z.im, z.re)); }
MM_DEV mm_dc mm_dcexp(mm_dc z) {
    double e = mmf_exp_any(z.re), s, c;
    mmf_sincos_d(z.im, &s, &c);
    if (z.im == 0.0) return mm_dcmake(e, z.im);
    return mm_dcmake(e * c, e * s);
}
MM_DEV mm_dc mm_dcsqrt(mm_dc z) {
    if (z.re == 0.0 && z.im == 0.0) return mm_dcmake(0.0, z.im);
    double r = hypot(z.re, z.im);
    if (z.re > 0.0) {
        double t = sqrt(0.5 * (r + z.re));
        return mm_dcmake(t, z.im / (2.0 * t));
    }
    double t = sqrt(0.5 * (r - z.re));
    return mm_dcmake(fabs(z.im) / (2.0 * t), copysign(t, z.im));
}

// glibc's float algorithms (mm_glibcf.h, verified bit for bit against the host's libm): the reference's
// generated C calls exactly these (ops.lisp:194-213), and they differ from the correctly rounded value by
// 1-2 float ulps often enough to show in a frame.
MM_DEV mm_complex mm_from_q(mmq_cf c) { return mm_cmake(c.re, c.im); }
MM_DEV mmq_cf mm_to_q(mm_complex c) { return mmq_cmake(c.re, c.im); }
MM_DEV mm_complex csqrtf(mm_complex z) { return mm_from_q(mmq_csqrtf(mm_to_q(z))); }
MM_DEV mm_complex cexpf(mm_complex z) { return mm_from_q(mmq_cexpf(mm_to_q(z))); }
MM_DEV mm_complex clogf(mm_complex z) { return mm_from_q(mmq_clogf(mm_to_q(z))); }
MM_DEV float cargf(mm_complex z) { return mmq_cargf(mm_to_q(z)); }
MM_DEV mm_complex cpowf(mm_complex x, mm_complex c) { return mm_from_q(mmq_cpowf(mm_to_q(x), mm_to_q(c))); }
MM_DEV mm_complex csinf(mm_complex z) { return mm_from_q(mmq_csinf(mm_to_q(z))); }
MM_DEV mm_complex ccosf(mm_complex z) { return mm_from_q(mmq_ccosf(mm_to_q(z))); }
MM_DEV mm_complex ctanf(mm_complex z) { return mm_from_q(mmq_ctanf(mm_to_q(z))); }
MM_DEV mm_complex csinhf(mm_complex z) { return mm_from_q(mmq_csinhf(mm_to_q(z))); }
MM_DEV mm_complex ccoshf(mm_complex z) { return mm_from_q(mmq_ccoshf(mm_to_q(z))); }
MM_DEV mm_complex ctanhf(mm_complex z) { return mm_from_q(mmq_ctanhf(mm_to_q(z))); }
MM_DEV mm_complex casinhf(mm_complex z) { return mm_from_q(mmq_casinhf(mm_to_q(z))); }
MM_DEV mm_complex casinf(mm_complex z) { return mm_from_q(mmq_casinf(mm_to_q(z))); }
MM_DEV mm_complex cacosf(mm_complex z) { return mm_from_q(mmq_cacosf(mm_to_q(z))); }
MM_DEV mm_complex cacoshf(mm_complex z) { return mm_from_q(mmq_cacoshf(mm_to_q(z))); }
MM_DEV mm_complex catanf(mm_complex z) { return mm_from_q(mmq_catanf(mm_to_q(z))); }
MM_DEV mm_complex catanhf(mm_complex z) { return mm_from_q(mmq_catanhf(mm_to_q(z))); }
// builtins/spec_func.c:35-64 (Luke's approximation; double-complex internally)
MM_DEV mm_dc mm_dcpow(mm_dc a, mm_dc b) { return mm_dcexp(mm_dcmul(b, mm_dclog(a))); }
MM_DEV mm_complex cgamma(mm_complex zf) {
    const double coeff[7] = {41.624436916439068, -51.224241022374774, 11.338755813488977, -0.747732687772388,
                             0.008782877493061, -1.899030264e-6, 1.946335e-9};
    mm_dc z = mm_widen(zf);
    mm_dc denom = mm_dcmake(1.0, 0.0);
    bool reflected = false;
    // creal(z) < 0 (spec_func.c:44-53): denom = prod (z + n), n < flr = -floor(creal z), then cgamma(z + flr) / denom.  z is a
    // complex *float* there: z + n and z + flr are float additions (rounded to float), the factors multiply in double, and
    // the inner call's result comes back as a complex float -- rounded -- before the double division.  z + flr lies in
    // [0, 1]: one level.
    if (z.re < 0.0) {
        const int flr = (int)-floor(z.re);
        for (int n = 0; n < flr; ++n) denom = mm_dcmul(denom, mm_dcmake((double)(zf.re + (float)n), z.im));
        z = mm_widen(mm_cmake(zf.re + (float)flr, zf.im));
        reflected = true;
    }
    mm_dc w = mm_dcmake(z.re - 1.0, z.im);
    mm_dc s = mm_dcmake(coeff[0], 0.0), H = mm_dcmake(1.0, 0.0);
    for (int n = 1; n < 7; ++n) {
        H = mm_dcmul(H, mm_dcdiv(mm_dcmake(w.re + 1 - n, w.im), mm_dcmake(w.re + n, w.im)));
        s = mm_dcadd(s, mm_dcmake(coeff[n] * H.re, coeff[n] * H.im));
    }
    mm_dc e = mm_dcexp(mm_dcmake(-w.re - 5.5, -w.im));
    mm_dc p = mm_dcpow(mm_dcmake(w.re + 5.5, w.im), mm_dcmake(w.re + 0.5, w.im));
    mm_dc r = mm_dcmul(mm_dcmul(mm_dcmake(2.506628274631 * e.re, 2.506628274631 * e.im), p), s);
    if (!reflected) return mm_narrow(r);
    return mm_narrow(mm_dcdiv(mm_widen(mm_narrow(r)), denom));
}

// ---- colours (new_template.c.in:71-77, opmacros.h:147-154,176-181) --------------------------
#define MAKE_RGBA_COLOR(r, g, b, a) \
    ((((color_t)(r)) << 24) | (((color_t)(g)) << 16) | (((color_t)(b)) << 8) | ((color_t)(a)))
#define RED(c) ((c) >> 24)
#define GREEN(c) (((c) >> 16) & 0xff)
#define BLUE(c) (((c) >> 8) & 0xff)
#define ALPHA(c) ((c)&0xff)
// (float)(c / 255.0) == (float)(c * (1.0 / 255.0)) for every c in 0..255 (enumerated in
// tests/test_cpu_suite.py::test_byte_to_float_identity), so the double divide of
// opmacros.h:147-150 becomes a double multiply; the value assigned to a float is identical.
#define MM_BYTE_TO_UNIT(c) ((double)(c) * (1.0 / 255.0))
#define RED_FLOAT(c) MM_BYTE_TO_UNIT(RED(c))
#define GREEN_FLOAT(c) MM_BYTE_TO_UNIT(GREEN(c))
#define BLUE_FLOAT(c) MM_BYTE_TO_UNIT(BLUE(c))
#define ALPHA_FLOAT(c) MM_BYTE_TO_UNIT(ALPHA(c))
#define MAKE_COLOR(r, g, b, a) \
    (MAKE_RGBA_COLOR(CLAMP01((r)) * 255, CLAMP01((g)) * 255, CLAMP01((b)) * 255, CLAMP01((a)) * 255))
#define TUPLE_NTH(t, n) ((t).v[(n)])
// Tree vectors (tree_vectors.c:89-108 tree_vector_get, :110-150 tree_vector_set): indices are clamped to the vector,
// a write makes a new vector.  The length is static, so a vector is N floats in registers and a run-time index
// is a chain of selects (a dynamically indexed array would go to scratch memory).
template <int N> MM_DEV float mm_tv_nth(int i, const mm_tup<N> &tv) {
    i = i < 0 ? 0 : (i >= N ? N - 1 : i);
    float r = tv.v[0];
#pragma unroll
    for (int k = 1; k < N; ++k) r = i == k ? tv.v[k] : r;
    return r;
}
template <int N> MM_DEV mm_tup<N> mm_tv_set(int i, mm_tup<N> tv, float v) {
    i = i < 0 ? 0 : (i >= N ? N - 1 : i);
#pragma unroll
    for (int k = 0; k < N; ++k) tv.v[k] = i == k ? v : tv.v[k];
    return tv;
}

// k / 255 for an integer-valued float k in [0, 255], two channels at a time: bit-identical to
// MM_BYTE_TO_UNIT's (float)((double)k * (1.0 / 255.0)) -- both are the correctly rounded
// quotient (enumerated for all 256 values, tests/test_cpu_suite.py) -- by one Newton step on
// the f32 product: q = k*r, q' = q + (k - 255 q) r with the residual exact in an fma.
MM_DEV mm_f2 mm_bytes_to_unit(mm_f2 k) {
    const float r = 1.0f / 255.0f;
    const mm_f2 q = k * r;
    const mm_f2 e = __builtin_elementwise_fma(mm_f2{-255.0f, -255.0f}, q, k);
    return __builtin_elementwise_fma(e, mm_f2{r, r}, q);
}

// TUPLE_FROM_COLOR (opmacros.h): the four bytes of a colour as unit floats
MM_DEV mm_tup<4> mm_tuple_from_color(color_t c) {
    const mm_f2 rg = mm_bytes_to_unit(mm_f2{(float)RED(c), (float)GREEN(c)}), ba = mm_bytes_to_unit(mm_f2{(float)BLUE(c), (float)ALPHA(c)});
    mm_tup<4> t;
    t.v[0] = rg.x; t.v[1] = rg.y; t.v[2] = ba.x; t.v[3] = ba.y;
    return t;
}

// ---- user values / images ----------------------------------------------------------------------
#define USERVAL_INT_ACCESS(n) (A.uservals[(n)].i)
#define USERVAL_FLOAT_ACCESS(n) (A.uservals[(n)].f)
#define USERVAL_BOOL_ACCESS(n) (A.uservals[(n)].i)
#define USERVAL_COLOR_ACCESS(n) (A.uservals[(n)].c)
#define USERVAL_CURVE_ACCESS(n) (A.uservals[(n)].i)
#define USERVAL_GRADIENT_ACCESS(n) (A.uservals[(n)].i)
#define USERVAL_IMAGE_ACCESS(n) (mm_image_from_table(A, A.uservals[(n)].image))

MM_DEV mm_image mm_image_from_table(const mm_args &A, int idx) {
    mm_image im;
    im.idx = idx;
    im.pw = A.images[idx].w;
    im.ph = A.images[idx].h;
    im.xf = 1.0f;
    im.yf = 1.0f;
    im.resized = 0;
    return im;
}
MM_DEV mm_image mm_null_image() { mm_image im; im.idx = -1; im.pw = 0; im.ph = 0; im.xf = im.yf = 1.0f; im.resized = 0; return im; }
#define UNINITED_IMAGE (mm_null_image())
#define IMAGE_PIXEL_WIDTH(i) ((i).pw)
#define IMAGE_PIXEL_HEIGHT(i) ((i).ph)
MM_DEV mm_image mm_resize_image(mm_image i, float xf, float yf) { i.xf = xf; i.yf = yf; i.resized = 1; return i; }
MM_DEV mm_image mm_strip_resize(mm_image i) { i.xf = 1.0f; i.yf = 1.0f; i.resized = 0; return i; }
#define RESIZE_IMAGE(i, xf, yf) (mm_resize_image((i), (xf), (yf)))
#define STRIP_RESIZE(i) (mm_strip_resize((i)))

#define USER_CURVE_POINTS 1024
#define APPLY_CURVE(c, p) (((const float *)A.curves)[(c)*USER_CURVE_POINTS + (int)(CLAMP01((p)) * (USER_CURVE_POINTS - 1))])
#define APPLY_GRADIENT(g, p) \
    (mm_tuple_from_color(((const color_t *)A.gradients)[(g)*USER_CURVE_POINTS + (int)(CLAMP01((p)) * (USER_CURVE_POINTS - 1))]))

// ---- pixel fetch (builtins.c:40-265, mathmap.c:1195-1209, mathmap_cmdline.c:131-184) -------------
MM_DEV void mm_apply_edge_behaviour(int &x, int &y, int width, int height) {
#if MM_EDGE_X == 1
    if (x < 0) x = x % width + width;
    else if (x >= width) x %= width;
#elif MM_EDGE_X == 2
    if (x < 0) x = -x % width;
    else if (x >= width) x = (width - 1) - (x % width);
#elif MM_EDGE_X == 3
    if (x < 0) { x = -x % width; y = (height - 1) - y; }
    else if (x >= width) { x = (width - 1) - (x % width); y = (height - 1) - y; }
#endif
#if MM_EDGE_Y == 1
    if (y < 0) y = y % height + height;
    else if (y >= height) y %= height;
#elif MM_EDGE_Y == 2
    if (y < 0) y = -y % height;
    else if (y >= height) y = (height - 1) - (y % height);
#elif MM_EDGE_Y == 3
    if (y < 0) { x = (width - 1) - x; y = -y % height; }
    else if (y >= height) { x = (width - 1) - x; y = (height - 1) - (y % height); }
#endif
}

// Image data lives in HBM: say so, so the loads are global_load (vmcnt only) instead of flat.
typedef const __attribute__((address_space(1))) color_t *mm_gpix;
typedef const __attribute__((address_space(1))) float4 *mm_gmap;

// Texel (cx, cy) of a hot image, in range.  mm_fetch_is_hot guarantees w, h < 2^24 and
// 4*w*h < 2^32, so the byte offset is one 24-bit multiply-add in 32 bits and the load takes
// the scalar base + 32-bit vector offset form (no 64-bit address arithmetic per tap).
MM_DEV color_t mm_load_texel(const mm_image_desc &d, int cx, int cy) {
    const unsigned boff = (__umul24((unsigned)cy, (unsigned)d.w) + (unsigned)cx) << 2;
    return *(mm_gpix)((const __attribute__((address_space(1))) char *)d.data + boff);
}

// get_pixel (mathmap_cmdline.c:131-184 / mathmap.c:1195-1209).
// Hot variant: `d` is a bound drawable and `frame` is valid (both wave-uniform, tested once by
// the caller).  Written without branches: the texel is always loaded from the clamped
// coordinates (in bounds for any bound image) and the edge colours are selected afterwards,
// so the taps of a fetch -- and of the next unrolled pixel -- form one basic block and their
// loads are in flight together.
MM_DEV color_t mm_get_pixel(const mm_args &A, const mm_image_desc &d, int x, int y) {
    mm_apply_edge_behaviour(x, y, d.w, d.h);
    const bool out_x = x < 0 || x >= d.w, out_y = y < 0 || y >= d.h;
    const int cx = x < 0 ? 0 : (x >= d.w ? d.w - 1 : x), cy = y < 0 ? 0 : (y >= d.h ? d.h - 1 : y);
    color_t v = mm_load_texel(d, cx, cy);
    if (out_y) v = A.edge_color_y;       // (measured: the four taps' selects are 8 % of Ident's kernel time)
    if (out_x) v = A.edge_color_x;
    return v;
}
// General variant: the reference's order of tests, each an early exit.  Besides unbound images
// and odd frame numbers this is also what large filter bodies use (no hot variant is built for
// them): a wave whose lanes all sample outside the image -- most of Droste's outer levels --
// then issues no load at all, which beats the branch-free form there.
MM_DEV color_t mm_get_pixel_cold(const mm_args &A, const mm_image_desc &d, int x, int y, int frame) {
    if (d.kind == MM_IMG_NULL) return MAKE_RGBA_COLOR(255, 255, 255, 255);
    mm_apply_edge_behaviour(x, y, d.w, d.h);
    if (x < 0 || x >= d.w) return A.edge_color_x;
    if (y < 0 || y >= d.h) return A.edge_color_y;
    if (frame < 0 || frame >= d.num_frames) return MAKE_RGBA_COLOR(255, 255, 255, 255);
    return ((mm_gpix)d.data)[(long)y * d.w + x];
}
MM_DEV bool mm_fetch_is_hot(const mm_image_desc &d, int frame) {
    return d.kind == MM_IMG_DRAWABLE && frame >= 0 && frame < d.num_frames && d.w < (1 << 24) && d.h < (1 << 24) &&
           (long)d.w * d.h < (1L << 30);
}

// HOT: the caller has established mm_fetch_is_hot(d, frame) for the whole launch.
template <bool HOT>
MM_DEV color_t mm_orig_val_pixel(const mm_args &A, const mm_image_desc &d, float x, float y, int frame) {
    x = (x + d.middle_x) * d.scale_x;
    y = -((y - d.middle_y) * d.scale_y);
#if !MM_SUPERSAMPLING
    // reference: `x += 0.5` (double add, rounded back to float).  The double sum of a float
    // and 0.5 is exact below 2^52, so rounding it to float equals the correctly rounded
    // float add; above that both leave x unchanged.
    x = x + 0.5f;
    y = y + 0.5f;
#endif
    // floor((double)x) of a float is the float floor: identical integer
    if (HOT) return mm_get_pixel(A, d, mm_f2i(floorf(x)), mm_f2i(floorf(y)));
    return mm_get_pixel_cold(A, d, mm_f2i(floorf(x)), mm_f2i(floorf(y)), frame);
}

// (color_t)rintf(v) & 0xff as x86-64 computes it: cvttss2si to 64 bits, low byte.  For the
// [0, 255.5) a sane bilinear sum lies in this is just the value; it matters when a coordinate is
// NaN, infinite or beyond +-2^31 pixels, where the weights (x - (float)INT_MIN ...) and therefore
// the sums are garbage the reference nevertheless converts deterministically: |v| >= 2^32 (and
// NaN) gives 0 -- a float that large is a multiple of 2^9 --, a negative v the two's complement.
MM_DEV color_t mm_x86_byte(float v) {
    const float a = fabsf(v);
    const unsigned u = (a < 4294967296.0f) ? (unsigned)a : 0u;
    return ((v < 0.0f) ? (0u - u) : u) & 0xffu;
}

MM_DEV mm_tup<4> mm_floatmap_pixel(const mm_image_desc &d, float x, float y) {
    mm_tup<4> t;
    int ix = (int)lrintf(d.ax * x + d.bx);
    int iy = (int)lrintf(d.ay * y + d.by);
    const bool outside = ix < 0 || ix >= d.w || iy < 0 || iy >= d.h;
    const int cx = ix < 0 ? 0 : (ix >= d.w ? d.w - 1 : ix), cy = iy < 0 ? 0 : (iy >= d.h ? d.h - 1 : iy);
    const float4 p = ((mm_gmap)d.data)[(long)cy * d.w + cx];
    t.v[0] = outside ? 0.0f : p.x; t.v[1] = outside ? 0.0f : p.y;
    t.v[2] = outside ? 0.0f : p.z; t.v[3] = outside ? 0.0f : p.w;
    return t;
}

// The descriptor of an image value.  For frame-constant images the generated kernel calls
// this once per work-item, before the pixel loop: after the first store of the loop the
// compiler may no longer use scalar loads for it.
MM_DEV mm_image_desc mm_load_desc(const mm_args &A, mm_image img) {
    if ((unsigned)img.idx >= A.num_images) {      // (negative: the null image and closure handles)
        mm_image_desc d;
        d.data = nullptr;
        d.w = d.h = 0;
        d.kind = MM_IMG_NULL;
        d.num_frames = 0;
        d.scale_x = d.scale_y = d.middle_x = d.middle_y = 0.0f;
        d.ax = d.bx = d.ay = d.by = 0.0f;
        return d;
    }
    return A.images[img.idx];
}

MM_DEV mm_tup<4> mm_intersample_tuple_cold(const mm_args &A, const mm_image_desc &d, float x, float y, int frame);
// opmacros.h:199-216 (closure images are applied at compile time and never reach here)
MM_DEV mm_tup<4> mm_orig_val_d(const mm_args &A, float x, float y, mm_image img, float f, const mm_image_desc &d) {
    if (img.resized) { x *= img.xf; y *= img.yf; }
    if (img.idx < 0) { mm_tup<4> t; t.v[0] = t.v[1] = t.v[2] = t.v[3] = 1.0f; return t; }
    if (d.kind == MM_IMG_FLOATMAP) return mm_floatmap_pixel(d, x, y);
#if MM_INTERSAMPLE
    return mm_intersample_tuple_cold(A, d, x, y, mm_f2i(f));
#else
    return mm_tuple_from_color(mm_orig_val_pixel<false>(A, d, x, y, mm_f2i(f)));
#endif
}
// The fetch of the kernel's hot variant: the image is a bound drawable and the frame valid
// (checked once per work-item before the pixel loop), so nothing here branches and the loads
// of consecutive unrolled pixels can overlap.  x * 1.0f is x, so the resize factors are
// applied by multiplication with a selected factor instead of under `if (img.resized)`.


// get_orig_val_intersample_pixel + TUPLE_FROM_COLOR for a hot image, fused: same operations
// in the same order per channel (builtins.c:202-250), two channels per packed instruction;
// the rounded channel value stays a float ((color_t)rintf(v) & 0xff is rintf(v) for the
// [0, 255.5) a convex combination of bytes lies in).
// `bad` is raised for a coordinate that is NaN / infinite / beyond +-2^31 pixels: the caller then
// discards this result and re-evaluates the pixel through the generic path (mm_x86_byte above).
struct mm_bilinear { mm_f2 rg, ba; };      // the four rounded channel sums: integer-valued floats in [0, 255]

// All four taps of every lane inside the image -- the case of nearly every wave: no clamping, no edge colours, and
// the taps are two 8-byte loads (x1 and x1 + 1 are neighbours in a row; global loads only need dword alignment)
// whose addresses cost three instructions.  Wave-uniform.  (One ballot per comparison: each is the v_cmp's lane
// mask itself, the OR is scalar.  An INT_MIN coordinate counts as outside.)
MM_DEV bool mm_taps_all_inside(const mm_image_desc &d, int x1, int y1) {
    return (__builtin_amdgcn_ballot_w64((unsigned)x1 >= (unsigned)(d.w - 1)) |
            __builtin_amdgcn_ballot_w64((unsigned)y1 >= (unsigned)(d.h - 1))) == 0;
}
MM_DEV void mm_load_taps_inside(const mm_image_desc &d, int x1, int y1, color_t &p1, color_t &p2, color_t &p3, color_t &p4) {
    typedef unsigned mm_u2 __attribute__((ext_vector_type(2)));
    typedef const __attribute__((address_space(1))) mm_u2 *mm_gpix2;
    const __attribute__((address_space(1))) char *base = (const __attribute__((address_space(1))) char *)d.data;
    const unsigned boff = (__umul24((unsigned)y1, (unsigned)d.w) + (unsigned)x1) << 2;
    const mm_u2 top = *(mm_gpix2)(base + boff), bot = *(mm_gpix2)(base + (boff + ((unsigned)d.w << 2)));
    p1 = top.x; p3 = top.y; p2 = bot.x; p4 = bot.y;
}
// get_orig_val_intersample_pixel's weighted sums (builtins.c:228-247), same operations in the same order per
// channel, two channels per packed instruction
MM_DEV void mm_bilinear_sums(color_t p1, color_t p2, color_t p3, color_t p4, float x2fact, float y2fact, mm_f2 &rg, mm_f2 &ba) {
    // reference: 1.0 - x2fact in double, rounded to float: the double difference is exact, so this is the
    // correctly rounded float subtraction
    const float x1fact = 1.0f - x2fact, y1fact = 1.0f - y2fact;
    const float p1fact = x1fact * y1fact, p2fact = x1fact * y2fact, p3fact = x2fact * y1fact, p4fact = x2fact * y2fact;
    rg = mm_f2{(float)RED(p1), (float)GREEN(p1)} * p1fact; ba = mm_f2{(float)BLUE(p1), (float)ALPHA(p1)} * p1fact;
    rg = rg + mm_f2{(float)RED(p2), (float)GREEN(p2)} * p2fact;
    ba = ba + mm_f2{(float)BLUE(p2), (float)ALPHA(p2)} * p2fact;
    rg = rg + mm_f2{(float)RED(p3), (float)GREEN(p3)} * p3fact;
    ba = ba + mm_f2{(float)BLUE(p3), (float)ALPHA(p3)} * p3fact;
    rg = rg + mm_f2{(float)RED(p4), (float)GREEN(p4)} * p4fact;
    ba = ba + mm_f2{(float)BLUE(p4), (float)ALPHA(p4)} * p4fact;
}

MM_DEV mm_bilinear mm_intersample_sums_hot(const mm_args &A, const mm_image_desc &d, float x, float y, bool &bad) {
    x = (x + d.middle_x) * d.scale_x;
    y = -((y - d.middle_y) * d.scale_y);
    // A coordinate the reference's (int) conversion turns into INT_MIN (NaN, infinite, beyond +-2^31 px) raises
    // `bad`: the caller discards this result and re-evaluates the pixel on the generic path.  Such a lane goes on
    // with coordinate 0, so that whatever the pixel body does with the discarded value stays tame: every lane's
    // weights are in [0, 1] and its sums in [0, 255.001).
#ifdef MM_PIXEL_INC
    // The preview's strided source (builtins.c:186-216, drawable_get_pixel_inc > 1): taps MM_PIXEL_INC apart on the grid
    // of multiples of the stride, weights in units of the stride.  Coordinates beyond +-2^30 px go to the generic path.
    x = (float)((double)x - MM_PIXEL_INC / 2.0);
    y = (float)((double)y - MM_PIXEL_INC / 2.0);
    const bool ok_x = fabsf(x) < 1073741824.0f, ok_y = fabsf(y) < 1073741824.0f;
    bad = bad || !ok_x || !ok_y;
    x = ok_x ? x : 0.0f;
    y = ok_y ? y : 0.0f;
    const int x1 = (int)(floor((double)(x / (float)MM_PIXEL_INC)) * (double)MM_PIXEL_INC), x2 = x1 + MM_PIXEL_INC;
    const int y1 = (int)(floor((double)(y / (float)MM_PIXEL_INC)) * (double)MM_PIXEL_INC), y2 = y1 + MM_PIXEL_INC;
    const color_t p1 = mm_get_pixel(A, d, x1, y1), p2 = mm_get_pixel(A, d, x1, y2);
    const color_t p3 = mm_get_pixel(A, d, x2, y1), p4 = mm_get_pixel(A, d, x2, y2);
    mm_f2 rg, ba;
    mm_bilinear_sums(p1, p2, p3, p4, (x - (float)x1) / (float)MM_PIXEL_INC, (y - (float)y1) / (float)MM_PIXEL_INC, rg, ba);
#else
    const bool ok_x = fabsf(x) < 2147483648.0f, ok_y = fabsf(y) < 2147483648.0f;
    bad = bad || !ok_x || !ok_y;
    x = ok_x ? x : 0.0f;
    y = ok_y ? y : 0.0f;
    const int x1 = (int)floorf(x), x2 = x1 + 1;
    const int y1 = (int)floorf(y), y2 = y1 + 1;
    color_t p1, p2, p3, p4;
    if (__builtin_expect(mm_taps_all_inside(d, x1, y1), 1)) mm_load_taps_inside(d, x1, y1, p1, p2, p3, p4);
    else {
        p1 = mm_get_pixel(A, d, x1, y1); p2 = mm_get_pixel(A, d, x1, y2);
        p3 = mm_get_pixel(A, d, x2, y1); p4 = mm_get_pixel(A, d, x2, y2);
    }
    mm_f2 rg, ba;
    mm_bilinear_sums(p1, p2, p3, p4, x - x1, y - y1, rg, ba);
#endif
    mm_bilinear r;      // rintf of a sum in [0, 255.001) is a byte: (color_t)rintf(v) & 0xff changes nothing
    r.rg = mm_f2{rintf(rg.x), rintf(rg.y)};
    r.ba = mm_f2{rintf(ba.x), rintf(ba.y)};
#ifdef MM_PIXEL_INC
    // the float quotient x / stride may round up to an integer the exact one stays below: a weight of -1 ulp then, a sum
    // that rounds to -0.0 where the reference's (color_t) conversion yields byte 0
    r.rg = r.rg + mm_f2{0.0f, 0.0f};
    r.ba = r.ba + mm_f2{0.0f, 0.0f};
#endif
    return r;
}

// The general fetch (any image kind, any frame number; what large filter bodies use throughout): the reference's
// order of tests per tap, each an early exit -- a wave whose lanes all sample outside the image, like most of
// Droste's outer levels, issues no load at all.
MM_DEV mm_tup<4> mm_intersample_tuple_cold(const mm_args &A, const mm_image_desc &d, float x, float y, int frame) {
    x = (x + d.middle_x) * d.scale_x;
    y = -((y - d.middle_y) * d.scale_y);
#ifdef MM_PIXEL_INC
    // builtins.c:186-216 with drawable_get_pixel_inc > 1 (the preview's strided source), operation for operation: the
    // half-stride shift in double, the float division by the stride, floor and the product in double, x86's conversion
    x = (float)((double)x - MM_PIXEL_INC / 2.0);
    y = (float)((double)y - MM_PIXEL_INC / 2.0);
    const int x1 = mm_d2i(floor((double)(x / (float)MM_PIXEL_INC)) * (double)MM_PIXEL_INC), x2 = (int)((unsigned)x1 + MM_PIXEL_INC);
    const int y1 = mm_d2i(floor((double)(y / (float)MM_PIXEL_INC)) * (double)MM_PIXEL_INC), y2 = (int)((unsigned)y1 + MM_PIXEL_INC);
    const float x2fact = (x - (float)x1) / (float)MM_PIXEL_INC, y2fact = (y - (float)y1) / (float)MM_PIXEL_INC;
#else
    const int x1 = mm_f2i(floorf(x)), x2 = x1 + 1;     // == (int)floor((double)x) on x86-64
    const int y1 = mm_f2i(floorf(y)), y2 = y1 + 1;
    const float x2fact = x - x1, y2fact = y - y1;
#endif
    const bool wild = x1 == (int)0x80000000 || y1 == (int)0x80000000;
#if MM_EDGE_X == 0 && MM_EDGE_Y == 0 && !defined(MM_NO_OUTSIDE_SHORTCUT) && !defined(MM_PIXEL_INC)      // (it knows the taps as x1, x1 + 1)
    // Every lane of the wave entirely outside a bound image (most fetches of Droste's level loop): get_pixel answers
    // x outside -> edge colour x, else y outside -> edge colour y, so a lane whose two columns are both outside has
    // four taps of edge colour x, one whose columns are both inside and whose rows are both outside four of edge
    // colour y -- and four equal taps are that colour (below).  Known from x1, y1 alone: no taps at all.
    if (d.kind != MM_IMG_NULL) {
        const bool x_out = (unsigned)x1 + 1u > (unsigned)d.w, x_in = (unsigned)x1 < (unsigned)(d.w - 1);
        const bool y_out = (unsigned)y1 + 1u > (unsigned)d.h;
        if (__builtin_amdgcn_ballot_w64(wild || !(x_out || (x_in && y_out))) == 0)
            return mm_tuple_from_color(x_out ? A.edge_color_x : A.edge_color_y);
    }
#endif
    color_t p1, p2, p3, p4;
    // Two alternatives were measured on Droste 8192^2 (1.79 ms as it is): the two-load form of the hot fetch for waves
    // whose taps are all inside a bound drawable (2.05 ms), and branch-free taps with clamped addresses inside one
    // region that waves without a tap to load skip (1.87 ms).
    p1 = mm_get_pixel_cold(A, d, x1, y1, frame); p2 = mm_get_pixel_cold(A, d, x1, y2, frame);
    p3 = mm_get_pixel_cold(A, d, x2, y1, frame); p4 = mm_get_pixel_cold(A, d, x2, y2, frame);
#ifndef MM_NO_SAME_TAPS
    // Four equal taps -- every lane of a wave sampling outside the image (Droste's outer levels) gets four times
    // the edge colour -- need no interpolation: the weights are in [0, 1] and sum to 1 within 2^-22, so each
    // channel's sum is within 255 * 2^-21 of the common byte and rintf returns that byte.  Wave-uniform test.
    if (__builtin_amdgcn_ballot_w64(wild || p1 != p2 || p1 != p3 || p1 != p4) == 0) return mm_tuple_from_color(p1);
#endif
    mm_f2 rg, ba;
    mm_bilinear_sums(p1, p2, p3, p4, x2fact, y2fact, rg, ba);
    rg = mm_f2{rintf(rg.x), rintf(rg.y)};
    ba = mm_f2{rintf(ba.x), rintf(ba.y)};
    if (wild)      // invalid coordinate: garbage sums, exact conversion
        return mm_tuple_from_color(MAKE_RGBA_COLOR(mm_x86_byte(rg.x), mm_x86_byte(rg.y), mm_x86_byte(ba.x), mm_x86_byte(ba.y)));
    // a valid coordinate has weights in [0, 1]: the sums are in [0, 255.001) and (color_t)rintf(v) & 0xff is rintf(v)
    rg = mm_bytes_to_unit(rg);
    ba = mm_bytes_to_unit(ba);
    mm_tup<4> t;
    t.v[0] = rg.x; t.v[1] = rg.y; t.v[2] = ba.x; t.v[3] = ba.y;
    return t;
}

MM_DEV mm_tup<4> mm_intersample_tuple_hot(const mm_args &A, const mm_image_desc &d, float x, float y, bool &bad) {
    const mm_bilinear s = mm_intersample_sums_hot(A, d, x, y, bad);
    const mm_f2 rg = mm_bytes_to_unit(s.rg), ba = mm_bytes_to_unit(s.ba);
    mm_tup<4> t;
    t.v[0] = rg.x; t.v[1] = rg.y; t.v[2] = ba.x; t.v[3] = ba.y;
    return t;
}

#if MM_INTERSAMPLE
// the fetch of a pixel that goes to the output unchanged: the rounded sums are kept for mm_store_fetched_pixel
MM_DEV mm_bilinear mm_orig_val_sums_hot(const mm_args &A, float x, float y, mm_image img, const mm_image_desc &d, bool &bad) {
    x *= img.resized ? img.xf : 1.0f;
    y *= img.resized ? img.yf : 1.0f;
    return mm_intersample_sums_hot(A, d, x, y, bad);
}
MM_DEV mm_tup<4> mm_tuple_of_sums(const mm_bilinear &s) {
    const mm_f2 rg = mm_bytes_to_unit(s.rg), ba = mm_bytes_to_unit(s.ba);
    mm_tup<4> t;
    t.v[0] = rg.x; t.v[1] = rg.y; t.v[2] = ba.x; t.v[3] = ba.y;
    return t;
}
#endif
MM_DEV mm_tup<4> mm_orig_val_hot(const mm_args &A, float x, float y, mm_image img, const mm_image_desc &d, bool &bad) {
    x *= img.resized ? img.xf : 1.0f;
    y *= img.resized ? img.yf : 1.0f;
#if MM_INTERSAMPLE
    return mm_intersample_tuple_hot(A, d, x, y, bad);
#else
    return mm_tuple_from_color(mm_orig_val_pixel<true>(A, d, x, y, 0));
#endif
}
MM_DEV mm_tup<4> mm_orig_val(const mm_args &A, float x, float y, mm_image img, float f) {
    return mm_orig_val_d(A, x, y, img, f, mm_load_desc(A, img));
}
#define ORIG_VAL(x, y, i, f) (mm_orig_val(A, (x), (y), (i), (f)))

// ---- coordinates and output (opmacros.h:156-157, new_template.c.in:243-309) ---------------------
#define CALC_VIRTUAL_X(pxl, size, off) (((pxl) - ((size)-1) / 2.0 + (off)) / (((size)-1) / 2.0))
#define CALC_VIRTUAL_Y(pxl, size, off) ((-(pxl) + ((size)-1) / 2.0 - (off)) / (((size)-1) / 2.0))

// CLAMP01 for the pack: MAX(0, MIN(1, x)) sends NaN to 0, and so does v_med3_f32 (with a NaN
// operand it returns the minimum of the others); the sign of a zero result is not observable
// after the multiply and the conversion to a byte.
MM_DEV float mm_clamp01(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, 1.0f); }

// The output frame is written once and not read again by this launch: non-temporal stores keep it from displacing the
// input in L2 / Infinity Cache (measured at 8192^2: Ident 0.175 -> 0.144 ms, Pond 0.599 -> 0.548, Droste NoTransparency=1 0.93 -> 0.905;
// Mandelbrot, which reads nothing, unchanged).  Set per kernel by hipgen.cpp (fetching kernels only); MMHIP_NT_STORE
// overrides.
#ifndef MM_NT_STORE
#define MM_NT_STORE 0
#endif
#if MM_NT_STORE
#define MM_STORE_U32(p, v) __builtin_nontemporal_store((unsigned)(v), (p))
#else
#define MM_STORE_U32(p, v) (*(p) = (v))
#endif
// Four unit-range channels to RGBA8 bytes (R in the lowest byte = first in memory).  new_template.c.in:279-293
// computes (unsigned char)(c * 255.0) with c = CLAMP01(v): the double product of a float and 255 is exact and
// the conversion truncates, so the byte is floor(255 c).  One f32 fma under round-toward-zero gives the same:
// RTZ(255 c + 2^23) has integer spacing, i.e. it is 2^23 + floor(255 c), and that integer sits in the low
// mantissa bits -- no f64 conversions and products (half rate), no float-to-int conversion.  The mode change is
// inside one asm statement together with the instructions that need it, so nothing can move across it.
MM_DEV unsigned mm_pack_rgba8(float r, float g, float b, float a) {
    r = mm_clamp01(r); g = mm_clamp01(g); b = mm_clamp01(b); a = mm_clamp01(a);
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
// the same bytes from the rounded sums of a bilinear fetch (integer-valued floats k in [0, 255]) whose unit
// values k / 255 go to the output unchanged: floor(255 * RN(k / 255)) is k for all 256 values (enumerated in
// tests/test_cpu_suite.py), so the division, the clamp and the product cancel
MM_DEV unsigned mm_float_bits(float f) { union { float f; unsigned u; } c; c.f = f; return c.u; }
MM_DEV unsigned mm_pack_bytes(const mm_bilinear &s) {
    const mm_f2 rg = s.rg + 8388608.0f, ba = s.ba + 8388608.0f;      // the integer lands in the low mantissa bits
    // (by value through mm_float_bits: __builtin_bit_cast of a vector element reads element 0 with this compiler)
    const float r = rg.x, g = rg.y, b = ba.x, a = ba.y;
    const unsigned lo = __builtin_amdgcn_perm(mm_float_bits(g), mm_float_bits(r), 0x0c0c0400u);
    const unsigned hi = __builtin_amdgcn_perm(mm_float_bits(a), mm_float_bits(b), 0x0c0c0400u);
    return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
}

MM_DEV void mm_store_pixel(const mm_args &A, int row_in_launch, int col, const mm_tup<4> &rt) {
    if (A.floatmap) {
        float4 *o = (float4 *)A.out + (long)row_in_launch * A.frame_render_width + col;
        *o = make_float4(rt.v[0], rt.v[1], rt.v[2], rt.v[3]);
        return;
    }
    unsigned char *p = (unsigned char *)A.out + (long)row_in_launch * A.row_stride + (long)col * A.output_bpp;
    const int bpp = A.output_bpp;
    if (bpp == 4) {
        MM_STORE_U32((unsigned *)p, mm_pack_rgba8(rt.v[0], rt.v[1], rt.v[2], rt.v[3]));      // one aligned 32-bit store
        return;
    }
    // new_template.c.in:279-293: the products are double, the conversion to a byte truncates
    if (bpp == 1 || bpp == 2)
        p[0] = (mm_clamp01(rt.v[0]) * 0.299 + mm_clamp01(rt.v[1]) * 0.587 + mm_clamp01(rt.v[2]) * 0.114) * 255.0;
    else {
        p[0] = mm_clamp01(rt.v[0]) * 255.0;
        p[1] = mm_clamp01(rt.v[1]) * 255.0;
        p[2] = mm_clamp01(rt.v[2]) * 255.0;
    }
    if (bpp == 2 || bpp == 4) p[bpp - 1] = mm_clamp01(rt.v[3]) * 255.0;
}

// a pixel whose four channels are one hot bilinear fetch, unchanged (every pure distortion ends like this)
MM_DEV void mm_store_fetched_pixel(const mm_args &A, int row_in_launch, int col, const mm_bilinear &s) {
    if (__builtin_expect(!A.floatmap && A.output_bpp == 4, 1)) {
        MM_STORE_U32((unsigned *)((unsigned char *)A.out + (long)row_in_launch * A.row_stride + (long)col * 4), mm_pack_bytes(s));
        return;
    }
    const mm_f2 rg = mm_bytes_to_unit(s.rg), ba = mm_bytes_to_unit(s.ba);
    mm_tup<4> t;
    t.v[0] = rg.x; t.v[1] = rg.y; t.v[2] = ba.x; t.v[3] = ba.y;
    mm_store_pixel(A, row_in_launch, col, t);
}

#endif  // MM_DEVICE_H
