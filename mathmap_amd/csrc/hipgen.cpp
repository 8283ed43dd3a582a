// IR -> HIP C++ source.  See hipgen.h.
//
// Shape of the generated translation unit:
//
//   #define MM_INTERSAMPLE / MM_EDGE_X / MM_TILE_W / MM_UNROLL ...   (compile-time options)
//   <mm_fastmath.h + tables, mm_gslmath.h>                           (float-argument libm, GSL ops)
//   <mm_device.h>  [<mm_noise_device.h> + gradient table]            (device runtime)
//   extern "C" __global__ void mm_prologue(mm_args A, char *XY)      lane 0: frame-constant values ->
//                                                                    constant buffer, native-filter call
//                                                                    records; all lanes: x / y tables
//   extern "C" __global__ void mm_pixels(mm_args A, const char *XY)  pixel kernel, one of two shapes:
//       loop shape   : frame constants + image descriptors loaded once, then A.ppt rows per work-item,
//                      MM_UNROLL pixels evaluated back to back (stores deferred), with a branch-free
//                      "hot" copy of the loop when every fetch reads a bound drawable
//       single shape : one pixel per work-item, lazy scalar loads (large bodies: no SGPR spills)
//       pair mode    : loop shape for small arithmetic-only bodies -- two vertically adjacent pixels as
//                      pairs of values in lockstep (two interleaved instruction streams), see pair_stmts
//
// Environment hooks for experiments (never needed for correct operation): MMHIP_UNROLL, MMHIP_TILE_W,
// MMHIP_SINGLE_PIXEL, MMHIP_PAIR, MMHIP_PAIR_DEBUG, MMHIP_WAVES_PER_EU, MMHIP_NO_FETCHED_RESULT, MMHIP_NO_SAME_TAPS, MMHIP_NO_OUTSIDE_SHORTCUT, MMHIP_PAIR_MASKS, MMHIP_NT_STORE,
// MMHIP_MAX_CALL_DEPTH here; MMHIP_NO_CSE in passes.cpp; MMHIP_PPT, MMHIP_HIPRTC_FLAGS, MMHIP_NO_CACHE, MMHIP_CACHE_DIR,
// MMHIP_SOURCE_OVERRIDE in runtime.cpp.
//
// Statement printing follows the reference's backends/cc.c:192-397 (one C variable
// per SSA value, phi copies at the end of branches / loop bodies), so the arithmetic
// the GPU executes is statement-for-statement the arithmetic gcc compiled for the
// cc backend.
#include "hipgen.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <set>
#include <sstream>
#include <stdexcept>

#include "front.h"

namespace mm {

namespace {

enum Slice { PROLOGUE, PIXEL, ROWS };

struct Generator {
    FilterCode &code;
    const KernelOptions &opt;
    std::ostringstream out;
    KernelSource ks;
    std::map<Value *, int> transfer_off;       // hoisted values read by the pixel kernel
    std::map<const Stmt *, int> dual_base_off; // outermost loops of both slices with native calls: where the prologue leaves the
                                               // number of dynamic entries taken before the loop
    std::vector<Value *> transfer_order;
    std::map<const Stmt *, int> native_index;

    Generator(FilterCode &c, const KernelOptions &o) : code(c), opt(o) {}

    // filter_$name functions (FilterCode::functions): `fn_root` is the code whose functions are callable,
    // `in_function` is set while one of their bodies is printed
    FilterCode *fn_root = nullptr;
    bool in_function = false;
    int function_index(const Filter *f) const {
        const FilterCode *root = fn_root ? fn_root : &code;
        for (size_t i = 0; i < root->functions.size(); ++i)
            if (root->functions[i]->filter == f) return (int)i;
        return -1;
    }

    static std::string vname(const Value *v) {
        char buf[64];
        snprintf(buf, sizeof buf, "v%d_%d", v->var->id, v->index);
        return buf;
    }

    static std::string ctype(const CompVar *v) {
        switch (v->type) {
            case Ty::Int: return "int";
            case Ty::Float: return "float";
            case Ty::Complex: return "mm_complex";
            case Ty::Color: return "color_t";
            case Ty::Curve:
            case Ty::Gradient: return "int";
            case Ty::Image: return "mm_image";
            case Ty::Tuple: return "mm_tup<" + std::to_string(v->tuple_len > 0 ? v->tuple_len : 4) + ">";
            case Ty::TreeVector:       // a tree vector's length is static (ir.cpp propagate_types): `len' floats
                if (v->tuple_len <= 0) throw CompileError("HIP backend: tree vector of unknown length");
                return "mm_tup<" + std::to_string(v->tuple_len) + ">";
            default: throw CompileError(std::string("HIP backend: unsupported variable type ") + ty_name(v->type));
        }
    }

    static int type_size(const CompVar *v) {
        switch (v->type) {
            case Ty::Complex: return 8;
            case Ty::Image: return 24;
            case Ty::Tuple: return 4 * (v->tuple_len > 0 ? v->tuple_len : 4);
            case Ty::TreeVector: return 4 * std::max(v->tuple_len, 1);
            default: return 4;
        }
    }

    static std::string float_literal(float f) {
        if (std::isnan(f)) return "__builtin_nan(\"\")";
        if (std::isinf(f)) return f > 0 ? "(1.0/0.0)" : "(-1.0/0.0)";
        char buf[64];
        snprintf(buf, sizeof buf, "%.17g", (double)f);
        std::string s = buf;
        if (s.find_first_of(".en") == std::string::npos) s += ".0";
        return s;
    }

    std::string prim(const Primary &p, Slice) {
        switch (p.kind) {
            case Primary::Val: {
                const Value *v = p.value;
                if (v->index < 0) {
                    switch (v->var->type) {
                        case Ty::Image: return "UNINITED_IMAGE";
                        case Ty::Complex: return "mm_cmake(0.0f, 0.0f)";
                        case Ty::Tuple:
                        case Ty::TreeVector: return ctype(v->var) + "{}";
                        default: return "0";
                    }
                }
                if (pair_uniform.count(v)) return "u" + vname(v);      // pair mode: wave-uniform loop value kept as a scalar
                return vname(v);
            }
            case Primary::IntConst: return p.i < 0 ? "(" + std::to_string(p.i) + ")" : std::to_string(p.i);
            case Primary::FloatConst: {
                std::string s = float_literal(p.f);
                return s[0] == '-' ? "(" + s + ")" : s;
            }
            case Primary::ComplexConst: return "COMPLEX(" + float_literal(p.f) + "," + float_literal(p.f2) + ")";
            case Primary::ColorConst: return std::to_string(p.color) + "u";
            default: return "0";
        }
    }

    // is `v` nameable from code of slice `sl` (defined there, or transferred into it)?
    bool value_visible(const Value *v, Slice sl) const {
        if (v->index < 0) return true;
        if (sl == PROLOGUE) return v->hoisted || v->loop_const;
        if (sl == ROWS) return v->row_const || transfer_off.count(const_cast<Value *>(v)) > 0;
        return !v->hoisted || transfer_off.count(const_cast<Value *>(v)) > 0;
    }

    static const char *libm_name(const char *cname) {
        static const std::pair<const char *, const char *> table[] = {
            {"sqrt", "mm_sqrt"}, {"hypot", "mm_hypot"}, {"sin", "mm_sin"}, {"cos", "mm_cos"}, {"tan", "mm_tan"},
            {"asin", "mm_asin"}, {"acos", "mm_acos"}, {"atan", "mm_atan"}, {"atan2", "mm_atan2"}, {"pow", "mm_pow"},
            {"exp", "mm_exp"}, {"log", "mm_log"}, {"sinh", "mm_sinh"}, {"cosh", "mm_cosh"}, {"tanh", "mm_tanh"},
            {"asinh", "mm_asinh"}, {"acosh", "mm_acosh"}, {"atanh", "mm_atanh"}, {"fabs", "mm_fabs"},
            {"floor", "mm_floor"}, {"ceil", "mm_ceil"}, {"GAMMA", "mm_gamma"}, {"gsl_sf_beta", "mm_beta"}};
        for (auto &t : table)
            if (!strcmp(t.first, cname)) return t.second;
        return nullptr;
    }

    // "mm_native_call(A, <record>, <counters>, k, " for a call site outside loops, the dynamic form for one inside
    std::string native_call_head(int k) const {
        const std::string ctr = "(int *)(XY + " + std::to_string(ks.native_ctr_offset) + "), ";
        if (!ks.natives[k].in_loop)
            return "mm_native_call(A, XY + " + std::to_string(ks.natives[k].record_offset) + ", " + ctr + std::to_string(k) + ", ";
        return "mm_native_call_in_loop(A, XY + " + std::to_string(ks.natives[ks.native_sites].record_offset) + ", " + ctr +
               std::to_string(k) + ", " + std::to_string(ks.native_sites) + ", ";
    }

    // the pixel slice's form of a call the prologue made from inside a loop: the handle of the next dynamic entry
    std::string native_result_in_loop() const {
        return "mm_native_result_in_loop(A, mm_dyn_ctr, " + std::to_string(ks.native_sites) + ")";
    }

    std::string rhs(const Rhs &r, Slice sl, const Stmt *stmt, const CompVar *lhs) {
        switch (r.kind) {
            case Rhs::Prim: return prim(r.prim, sl);
            case Rhs::Internal: return r.internal;
            case Rhs::Tuple:
            case Rhs::TreeVector: {      // backends/cc.c:287-300: float tuple[n] = { args }; ALLOC_TREE_VECTOR(n, tuple)
                std::string s = "mm_tup<" + std::to_string(r.args.size()) + ">{{";
                for (size_t i = 0; i < r.args.size(); ++i) s += (i ? ", (float)(" : "(float)(") + prim(r.args[i], sl) + ")";
                return s + "}}";
            }
            case Rhs::Closure: {
                if (r.filter->kind == Filter::MathMap)      // index -2 - id: rendered for a native filter by the runtime
                    return "mm_closure_image(A, " + std::to_string(stmt ? stmt->closure_id : -1) + ")";
                auto it = native_index.find(stmt);
                if (it != native_index.end() && sl == PIXEL && stmt->hoisted && ks.natives[it->second].in_loop)
                    return native_result_in_loop();      // a loop in both slices (passes.cpp mark_dual_loops): the prologue made this call
                if (it == native_index.end() || sl != PROLOGUE)
                    throw CompileError("native filter `" + r.filter->name +
                                       "' is called with pixel-dependent arguments (or, with a filter closure among them, under pixel-dependent "
                                       "control); the HIP backend needs them frame-constant");
                int k = it->second;
                std::string s = native_call_head(k) + std::to_string(r.args.size());
                for (size_t i = 0; i < r.args.size(); ++i) s += ", mm_narg(" + prim(r.args[i], sl) + ")";
                for (size_t i = r.args.size(); i < 4; ++i) s += ", mm_narg(0)";
                return s + ")";
            }
            case Rhs::FilterCall: {
                // backends/cc.c:221-235: build the callee's argument block (output_make_mathmap_filter_closure), call
                // filter_$name(invocation, image, x, y, t, pools).  Here a statement expression around a template
                // instance: the template parameter is the call depth, so the call graph is a finite DAG (see emit_functions)
                const int k = function_index(r.filter);
                if (k < 0) throw CompileError("internal: call of filter `" + r.filter->name + "' without a function body");
                const size_t n = r.filter->uservals.size();
                if (r.args.size() != n + 3) throw CompileError("internal: malformed filter call");
                std::string e = "({ mm_uvarg mm_ca[" + std::to_string(n ? n : 1) + "]; ";
                for (size_t i = 0; i < n; ++i) {
                    const UservalInfo &u = r.filter->uservals[i];
                    const char *field = u.kind == UvKind::Float ? ".f = (float)(" : u.kind == UvKind::Color ? ".c = (color_t)(" : u.kind == UvKind::Image ? ".img = (" : ".i = (int)(";
                    e += "mm_ca[" + std::to_string(u.index) + "]" + field + prim(r.args[i], sl) + "); ";
                }
                e += "mm_filter_" + std::to_string(k) + "<" + (in_function ? "MM_D + 1" : "0") + ">(A, mm_ca, (float)(" + prim(r.args[n], sl) +
                     "), (float)(" + prim(r.args[n + 1], sl) + "), (float)(" + prim(r.args[n + 2], sl) + "), col, rl, mm_rand_ctr); })";
                return e;
            }
            case Rhs::Op: {
                const char *cn = r.op->cname;
                if (!strcmp(cn, "RENDER")) {
                    auto it = native_index.find(stmt);
                    if (it != native_index.end() && sl == PIXEL && stmt->hoisted && ks.natives[it->second].in_loop)
                        return native_result_in_loop();
                    if (it == native_index.end() || sl != PROLOGUE)
                        throw CompileError("render() needs frame-constant arguments in the HIP backend");
                    int k = it->second;
                    return native_call_head(k) + "3, mm_narg(" + prim(r.args[0], sl) + "), mm_narg(" + prim(r.args[1], sl) +
                           "), mm_narg(" + prim(r.args[2], sl) + "), mm_narg(0))";
                }
                for (const char *bad : {"SOLVE_POLY_2", "SOLVE_POLY_3",      // unimplemented stubs in the reference too (opmacros.h:97-99)
                                        "START_DEBUG_TUPLE",
                                        "SET_DEBUG_TUPLE_DATA", "OUTPUT_TUPLE"})
                    if (!strcmp(cn, bad)) throw CompileError(std::string("HIP backend: op ") + cn + " is not supported yet");
                if (!strcmp(cn, "PRINT_FLOAT") || !strcmp(cn, "NEWLINE")) return "0";
                // opmacros.h:189-190: the index is passed to a C `int' parameter (converted like FLOAT2INT), the value to a float
                if (!strcmp(cn, "TREE_VECTOR_NTH"))
                    return "mm_tv_nth(FLOAT2INT(" + prim(r.args[0], sl) + "), " + prim(r.args[1], sl) + ")";
                if (!strcmp(cn, "SET_TREE_VECTOR_NTH"))
                    return "mm_tv_set(FLOAT2INT(" + prim(r.args[0], sl) + "), " + prim(r.args[1], sl) + ", (float)(" + prim(r.args[2], sl) + "))";
                // escape-time test `sqrt(a) < 2^k`  ->  0 <= a < 4^k (exact, see mm_device.h)
                if (opt.fast_math_exact && !strcmp(cn, "LESS") && r.args[0].kind == Primary::Val && r.args[1].is_const()) {
                    const Stmt *d = r.args[0].value->def;
                    double k = r.args[1].kind == Primary::IntConst ? (double)r.args[1].i
                             : r.args[1].kind == Primary::FloatConst ? (double)r.args[1].f : -1.0;
                    int ex = 0;
                    bool pow2 = k > 0 && std::frexp(k, &ex) == 0.5 && ex > -50 && ex < 50;
                    if (pow2 && d && d->kind == Stmt::Assign && d->rhs.kind == Rhs::Op && !strcmp(d->rhs.op->cname, "sqrt") &&
                        d->lhs->var->type == Ty::Float && d->rhs.args[0].type() == Ty::Float &&
                        d->rhs.args[0].kind == Primary::Val && value_visible(d->rhs.args[0].value, sl))
                    {
                        // a sum of float squares is >= +0 or NaN, and `a < K*K` is false for NaN like
                        // sqrt(NaN) < K: the `a >= 0` half of the test is then dead
                        if (nonneg_or_nan(d->rhs.args[0].value, 0))
                            return "((" + prim(d->rhs.args[0], sl) + ") < " + float_literal((float)(k * k)) + "f)";
                        return "MM_SQRT_LESS_POW2(" + prim(d->rhs.args[0], sl) + ", " + float_literal((float)(k * k)) + "f)";
                    }
                }
                if (sl == PIXEL && hot_mode && stmt == fetched_result)
                    return "mm_tuple_of_sums(mm_rs[mm_u] = mm_orig_val_sums_hot(A, " + prim(r.args[0], sl) + ", " + prim(r.args[1], sl) + ", " +
                           prim(r.args[2], sl) + ", " + vname(r.args[2].value) + "_desc, mm_bad))";
                if (sl == PIXEL && hot_mode && hot_sites.count(stmt))
                    return "mm_orig_val_hot(A, " + prim(r.args[0], sl) + ", " + prim(r.args[1], sl) + ", " + prim(r.args[2], sl) +
                           ", " + vname(r.args[2].value) + "_desc, mm_bad)";
                if (sl == PIXEL && !strcmp(cn, "ORIG_VAL") && r.args.size() == 4 && r.args[2].kind == Primary::Val &&
                    preloaded_desc.count(r.args[2].value))
                    return "mm_orig_val_d(A, " + prim(r.args[0], sl) + ", " + prim(r.args[1], sl) + ", " + prim(r.args[2], sl) +
                           ", " + prim(r.args[3], sl) + ", " + vname(r.args[2].value) + "_desc)";
                // x / c with c = +-2^k: x * (1/c) is the same correctly rounded value (scaling by a power of two
                // is exact or rounds identically, subnormals and overflow included), one multiply instead of
                // the ~10-instruction IEEE division sequence
                if (opt.fast_math_exact && !strcmp(cn, "DIV") && r.args.size() == 2 && r.args[1].kind == Primary::FloatConst) {
                    int ex = 0;
                    const float c = r.args[1].f;
                    if (std::isfinite(c) && c != 0.0f && std::fabs(std::frexp(c, &ex)) == 0.5f && ex > -100 && ex < 100)
                        return "((float)(" + prim(r.args[0], sl) + ") * " + float_literal(1.0f / c) + "f)";
                }
                if (in_function && !strncmp(cn, "USERVAL_", 8) && r.args.size() == 1 && r.args[0].kind == Primary::IntConst) {
                    // inside filter_$name the user values are the call's arguments (new_template.c.in:375-422: the closure's args)
                    const char *field = !strcmp(cn, "USERVAL_FLOAT_ACCESS") ? "f" : !strcmp(cn, "USERVAL_COLOR_ACCESS") ? "c"
                                      : !strcmp(cn, "USERVAL_IMAGE_ACCESS") ? "img" : "i";
                    return "(UV[" + std::to_string(r.args[0].i) + "]." + field + ")";
                }
                std::string name = cn;
                if (const char *lm = libm_name(cn)) name = lm;
                // exact f32 fast path: (float)sqrt((double)f) == sqrtf(f), correctly rounded
                if (opt.fast_math_exact && !strcmp(cn, "sqrt") && lhs && lhs->type == Ty::Float &&
                    r.args[0].type() == Ty::Float)
                    name = "mm_sqrt_f32";
                // (float)sin((double)f), (float)cos((double)f): table-driven evaluation verified
                // against glibc for every float below 2^22 (mm_fastmath.h, tools/verify_fastmath.c)
                if (opt.fast_math_exact && lhs && lhs->type == Ty::Float && r.args.size() == 1 && r.args[0].type() == Ty::Float) {
                    if (!strcmp(cn, "sin")) name = "mmf_sin_f32";
                    else if (!strcmp(cn, "cos")) name = "mmf_cos_f32";
                    else if (!strcmp(cn, "exp")) name = "mmf_exp_f32";
                    else if (!strcmp(cn, "log")) name = "mmf_log_f32";
                    // the platform's double function and the list of the arguments where its float differs from glibc's
                    // (mm_libm_exceptions.h; the other one-argument functions have no such argument: tools/libm_exceptions.py)
                    else if (!strcmp(cn, "asinh")) name = "mmf_asinh_f32";
                    else if (!strcmp(cn, "acosh")) name = "mmf_acosh_f32";
                }
                if (lhs && lhs->type == Ty::Int && (!strcmp(cn, "floor") || !strcmp(cn, "ceil")))
                    name = !strcmp(cn, "floor") ? "mm_floor_i" : "mm_ceil_i";      // x86 double -> int conversion
                if (opt.fast_math_exact && !strcmp(cn, "pow") && lhs && lhs->type == Ty::Float && r.args.size() == 2 &&
                    r.args[0].type() == Ty::Float && r.args[1].type() == Ty::Float)
                    name = "mmf_pow_f32";
                if (opt.fast_math_exact && !strcmp(cn, "hypot") && r.args.size() == 2 && r.args[0].type() == Ty::Float &&
                    r.args[1].type() == Ty::Float) {
                    // glibc's own arithmetic for two floats (mm_fastmath.h); a double-typed result keeps the earlier form
                    name = lhs && lhs->type == Ty::Float ? "mmf_hypot_f32" : "mm_hypot_ff";
                }
                std::string s = name + "(";
                for (size_t i = 0; i < r.args.size(); ++i) s += (i ? "," : "") + prim(r.args[i], sl);
                return s + ")";
            }
            default: return "0";
        }
    }

    // ---- which values live where -------------------------------------------------------
    void collect_values(Block &b, Slice sl, std::vector<Value *> &defs, std::set<Value *> &uses) {
        auto use = [&](const Rhs &r) {
            if (r.kind == Rhs::Prim && r.prim.kind == Primary::Val) uses.insert(r.prim.value);
            for (const Primary &p : r.args)
                if (p.kind == Primary::Val) uses.insert(p.value);
        };
        for (Stmt *s : b) {
            bool mine = sl == PROLOGUE ? s->hoisted : sl == ROWS ? s->in_row : s->in_pixel;
            if (!mine) continue;
            switch (s->kind) {
                case Stmt::Assign: defs.push_back(s->lhs); use(s->rhs); break;
                case Stmt::Phi: defs.push_back(s->lhs); use(s->rhs); use(s->rhs2); break;
                case Stmt::If:
                    use(s->cond);
                    collect_values(s->then_, sl, defs, uses);
                    collect_values(s->else_, sl, defs, uses);
                    collect_values(s->phis, sl, defs, uses);
                    break;
                case Stmt::While:
                    collect_values(s->phis, sl, defs, uses);
                    use(s->cond);
                    collect_values(s->body, sl, defs, uses);
                    break;
                default: break;
            }
        }
    }

    static bool uses_noise(const Block &b) {
        for (const Stmt *s : b) {
            if ((s->kind == Stmt::Assign) && s->rhs.kind == Rhs::Op && !strncmp(s->rhs.op->cname, "libnoise_", 9)) return true;
            if (s->kind == Stmt::If && (uses_noise(s->then_) || uses_noise(s->else_))) return true;
            if (s->kind == Stmt::While && uses_noise(s->body)) return true;
        }
        return false;
    }

    // Is the float value provably >= +0 (or NaN)?  Squares of one value, non-negative literals,
    // and sums / copies of such.  Products and sums are f32 here (the C type of float (op) float
    // is promoted to double by the op macros only for libm calls, not for + and *).
    static bool nonneg_or_nan(const Value *v, int depth) {
        if (!v || depth > 16 || v->var->type != Ty::Float) return false;
        const Stmt *d = v->def;
        if (!d || d->kind != Stmt::Assign) return false;
        const Rhs &r = d->rhs;
        auto prim_ok = [&](const Primary &p) {
            if (p.kind == Primary::FloatConst) return p.f >= 0.0f && !std::signbit(p.f);
            if (p.kind == Primary::IntConst) return p.i >= 0;
            if (p.kind == Primary::Val) return nonneg_or_nan(p.value, depth + 1);
            return false;
        };
        if (r.kind == Rhs::Prim) return prim_ok(r.prim);
        if (r.kind != Rhs::Op) return false;
        if (!strcmp(r.op->cname, "MUL") && r.args.size() == 2 && r.args[0].kind == Primary::Val && r.args[1].kind == Primary::Val &&
            r.args[0].value == r.args[1].value && r.args[0].value->var->type == Ty::Float)
            return true;
        if (!strcmp(r.op->cname, "ADD") && r.args.size() == 2) return prim_ok(r.args[0]) && prim_ok(r.args[1]);
        return false;
    }

    std::set<const Value *> preloaded_desc;   // image values whose descriptor is loaded before the pixel loop
    std::set<const Stmt *> hot_sites;         // ORIG_VAL statements eligible for mm_orig_val_hot
    bool hot_mode = false;
    // The hot fetch whose four channels are the filter's result, unchanged (the last statement of every pure
    // distortion: `in(f(xy))`): the hot loop then keeps the fetch's rounded byte sums and stores them directly
    // (mm_store_fetched_pixel) instead of dividing by 255, clamping and multiplying by 255 again.
    const Stmt *fetched_result = nullptr;
    const Stmt *find_fetched_result() const {
        if (!opt.intersample || getenv("MMHIP_NO_FETCHED_RESULT")) return nullptr;
        const Stmt *fetch = nullptr;
        for (int i = 0; i < 4; ++i) {
            const Value *v = code.result[i];
            const Stmt *d = v ? v->def : nullptr;
            // copies between the TUPLE_NTH and the result are gone after copy propagation
            if (!d || d->kind != Stmt::Assign || d->parent || !d->in_pixel || d->rhs.kind != Rhs::Op || strcmp(d->rhs.op->cname, "TUPLE_NTH") ||
                d->rhs.args[0].kind != Primary::Val || d->rhs.args[1].kind != Primary::IntConst || d->rhs.args[1].i != i)
                return nullptr;
            const Stmt *f = d->rhs.args[0].value->def;
            if (!f || f->parent || !hot_sites.count(f) || (fetch && f != fetch)) return nullptr;
            fetch = f;
        }
        return fetch;
    }

    // ORIG_VALs of the pixel slice whose image descriptor is preloaded and whose frame
    // argument is a literal or a frame constant: their "bound drawable, valid frame" test can
    // be made once per work-item.  Appends one condition per site.
    void find_hot_fetches(const Block &b, std::vector<std::string> &conds) {
        for (const Stmt *s : b) {
            if (!s->in_pixel) continue;
            if (s->kind == Stmt::Assign && s->rhs.kind == Rhs::Op && !strcmp(s->rhs.op->cname, "ORIG_VAL") &&
                s->rhs.args.size() == 4 && s->rhs.args[2].kind == Primary::Val && preloaded_desc.count(s->rhs.args[2].value)) {
                const Primary &f = s->rhs.args[3];
                const bool frame_const = f.is_const() || (f.kind == Primary::Val && transfer_off.count(f.value));
                if (frame_const) {
                    hot_sites.insert(s);
                    conds.push_back("mm_fetch_is_hot(" + vname(s->rhs.args[2].value) + "_desc, (int)(" + prim(f, PIXEL) + "))");
                }
            }
            if (s->kind == Stmt::If) { find_hot_fetches(s->then_, conds); find_hot_fetches(s->else_, conds); }
            if (s->kind == Stmt::While) find_hot_fetches(s->body, conds);
        }
    }

    static bool hoisted_uses_time(const Block &b) {
        for (const Stmt *s : b) {
            if (s->hoisted && s->kind == Stmt::Assign && s->rhs.kind == Rhs::Internal && (s->rhs.internal == "t" || s->rhs.internal == "frame"))
                return true;
            if (s->kind == Stmt::If && (hoisted_uses_time(s->then_) || hoisted_uses_time(s->else_))) return true;
            if (s->kind == Stmt::While && hoisted_uses_time(s->body)) return true;
        }
        return false;
    }

    // pixel-slice statistics for the unroll choice
    static void pixel_stats(const Block &b, int &stmts, int &fetches) {
        for (const Stmt *s : b) {
            if (!s->in_pixel) continue;
            ++stmts;
            if (s->kind == Stmt::Assign && s->rhs.kind == Rhs::Op && !strcmp(s->rhs.op->cname, "ORIG_VAL")) ++fetches;
            if (s->kind == Stmt::If) { pixel_stats(s->then_, stmts, fetches); pixel_stats(s->else_, stmts, fetches); }
            if (s->kind == Stmt::While) pixel_stats(s->body, stmts, fetches);
        }
    }

    // Filters that fetch pixels are bound by memory latency with one pixel in flight per
    // work-item (measured: a nearest fetch cost 0.25 ms at 8192^2 against 0.06 ms for the
    // store); evaluating several pixels back to back multiplies the loads in flight.  Measured at 8192^2
    // (tools/ab_unroll.sh, profiles/r02_ab_unroll.txt): Ident 0.233 / 0.204 / 0.205 ms and Pond 0.705 / 0.673 /
    // 0.752 ms for 2 / 4 / 8 pixels.  Large bodies are left alone: more code and registers cost more
    // occupancy than the overlap gains.
    int auto_unroll() const {
        if (const char *e = getenv("MMHIP_UNROLL")) { int u = atoi(e); if (u >= 1 && u <= 8) return u; }
        int stmts = 0, fetches = 0;
        pixel_stats(code.body, stmts, fetches);
        if (fetches == 0) return 1;
        return stmts <= 64 ? 4 : stmts <= 400 ? 2 : 1;
    }
    // Columns of a workgroup's 256 work-items.  16 x 16 keeps the gathers of a distortion local in both directions;
    // a body that is little more than its fetch (a copy, a scale, a flip) streams rows, and a wave that covers
    // 64 pixels of one row reads and writes whole cache lines (same A/B: Ident 0.204 -> 0.187 ms, Pond 0.673 -> 0.684).
    int auto_tile_w() const {
        int stmts = 0, fetches = 0;
        pixel_stats(code.body, stmts, fetches);
        return fetches >= 1 && stmts <= 12 ? 64 : 16;
    }

    void find_natives(Block &b, int loop_depth = 0) {
        for (Stmt *s : b) {
            if (s->kind == Stmt::Assign) {
                bool native = (s->rhs.kind == Rhs::Closure && s->rhs.filter->kind == Filter::Native) ||
                              (s->rhs.kind == Rhs::Op && !strcmp(s->rhs.op->cname, "RENDER"));
                if (native) {
                    NativeCall nc;
                    nc.func = s->rhs.kind == Rhs::Closure ? s->rhs.filter->native_func : "RENDER";
                    for (const Primary &p : s->rhs.args) nc.arg_types.push_back(p.type());
                    nc.in_loop = loop_depth > 0;
                    native_index[s] = (int)ks.natives.size();
                    ks.natives.push_back(nc);
                }
            } else if (s->kind == Stmt::If) {
                find_natives(s->then_, loop_depth);
                find_natives(s->else_, loop_depth);
            } else if (s->kind == Stmt::While)
                find_natives(s->body, loop_depth + 1);
        }
    }

    // ---- statement printing ------------------------------------------------------------------
    void phis(Block &list, int branch, Slice sl, const std::string &ind) {
        std::vector<Stmt *> mine;
        for (Stmt *p : list) {
            if (p->kind != Stmt::Phi) continue;
            if (!(sl == PROLOGUE ? p->hoisted : p->in_pixel)) continue;
            const Rhs &r = branch == 0 ? p->rhs : p->rhs2;
            if (r.kind == Rhs::Prim && r.prim.kind == Primary::Val && r.prim.value == p->lhs) continue;
            mine.push_back(p);
        }
        // phis are parallel copies: if a source is the target of another copy in the
        // list, go through temporaries
        std::set<Value *> targets;
        for (Stmt *p : mine) targets.insert(p->lhs);
        bool hazard = false;
        for (Stmt *p : mine) {
            const Rhs &r = branch == 0 ? p->rhs : p->rhs2;
            if (r.kind == Rhs::Prim && r.prim.kind == Primary::Val && targets.count(r.prim.value)) hazard = true;
        }
        if (!hazard) {
            for (Stmt *p : mine)
                out << ind << vname(p->lhs) << " = " << rhs(branch == 0 ? p->rhs : p->rhs2, sl, p, p->lhs->var) << ";\n";
            return;
        }
        out << ind << "{\n";
        for (size_t i = 0; i < mine.size(); ++i)
            out << ind << "  " << ctype(mine[i]->lhs->var) << " pc" << i << " = "
                << rhs(branch == 0 ? mine[i]->rhs : mine[i]->rhs2, sl, mine[i], mine[i]->lhs->var) << ";\n";
        for (size_t i = 0; i < mine.size(); ++i) out << ind << "  " << vname(mine[i]->lhs) << " = pc" << i << ";\n";
        out << ind << "}\n";
    }


    // ---- pair mode: two pixels of a work-item evaluated in lockstep as 2-vectors ------------------
    // For a pixel body that is nothing but int/float arithmetic, comparisons and structured control
    // flow (Mandelbrot and its relatives), the two pixels a work-item renders per loop step are
    // evaluated together: every SSA value is a 2-vector (x component: the first pixel), `if`s are
    // if-converted (both sides evaluated -- the slice is pure -- and the exit phis select), a `while`
    // runs while either pixel is active with the loop phis frozen per pixel by a select.  Each component
    // sees exactly the scalar kernel's operations in the scalar kernel's order.  What it buys: a gfx950
    // SIMD hands a wave an issue slot every ~4 cycles, in which the wave can issue two independent vector
    // instructions (2 cycles each) -- a single pixel's dependent chain uses half of that, whatever the
    // occupancy (tools/pk_rate.hip: dependent v_mul/v_add 4.25 cycles per instruction at 8 waves per SIMD,
    // two independent chains 2.4).  The pair's components are kept as separate scalars, so the
    // instruction stream alternates between the two pixels; as v_pk_*_f32 (MM_PAIR_SCALAR=0) the same
    // work issues in one 4-cycle instruction and gains much less (Mandelbrot 8192^2: 0.372 ms one pixel
    // at a time, 0.360 packed, 0.313 interleaved).
    bool pair_mode = false;
    std::set<const Value *> pair_defs;     // values defined in the pixel slice (vectors in pair mode)
    int pair_ids = 0;

    static bool pair_scalar_ty(Ty t) { return t == Ty::Int || t == Ty::Float; }
    bool pair_prim_ok(const Primary &p) const {
        if (p.kind == Primary::IntConst || p.kind == Primary::FloatConst) return true;
        return p.kind == Primary::Val && pair_scalar_ty(p.value->var->type);
    }
    bool pair_rhs_ok(const Rhs &r) const {
        if (r.kind == Rhs::Prim) return pair_prim_ok(r.prim);
        if (r.kind == Rhs::Internal) return r.internal == "x" || r.internal == "y";
        if (r.kind != Rhs::Op) return false;
        static const char *ok[] = {"ADD", "SUB", "MUL", "NEG", "DIV", "LESS", "LEQ", "EQ", "NOT", "sqrt"};
        bool found = false;
        for (const char *o : ok) found = found || !strcmp(r.op->cname, o);
        if (!found) return false;
        for (const Primary &a : r.args) if (!pair_prim_ok(a)) return false;
        if (!strcmp(r.op->cname, "sqrt") && r.args[0].type() != Ty::Float) return false;
        if (!strcmp(r.op->cname, "NOT") && r.args[0].type() != Ty::Int) return false;
        return true;
    }
    bool pair_block_ok(const Block &b) const {
        for (const Stmt *s : b) {
            if (!s->in_pixel) continue;
            switch (s->kind) {
                case Stmt::Assign:
                    if (!s->lhs || !pair_scalar_ty(s->lhs->var->type) || !pair_rhs_ok(s->rhs)) {
                        if (getenv("MMHIP_PAIR_DEBUG"))
                            fprintf(stderr, "pair mode: statement not covered (%s)\n",
                                    s->rhs.kind == Rhs::Op ? s->rhs.op->cname : s->rhs.kind == Rhs::Internal ? s->rhs.internal.c_str() : "rhs kind");
                        return false;
                    }
                    break;
                case Stmt::If:
                case Stmt::While:
                    if (s->cond.kind != Rhs::Prim || !pair_prim_ok(s->cond.prim) || s->cond.prim.type() != Ty::Int) return false;
                    for (const Stmt *ph : s->phis) {
                        if (!ph->in_pixel) continue;
                        if (!pair_scalar_ty(ph->lhs->var->type) || ph->rhs.kind != Rhs::Prim || ph->rhs2.kind != Rhs::Prim ||
                            !pair_prim_ok(ph->rhs.prim) || !pair_prim_ok(ph->rhs2.prim))
                            return false;
                    }
                    if (s->kind == Stmt::If ? !(pair_block_ok(s->then_) && pair_block_ok(s->else_)) : !pair_block_ok(s->body)) return false;
                    break;
                default: break;
            }
        }
        return true;
    }
    bool pair_eligible() const {
        if (!opt.fast_math_exact || !ks.natives.empty()) return false;
        const char *force = getenv("MMHIP_PAIR");          // 0: never, 1: whenever the body is covered, unset: small bodies
        if (force && !atoi(force)) return false;
        int stmts = 0, fetches = 0;
        pixel_stats(code.body, stmts, fetches);
        // measured at 8192^2: Mandelbrot with its parameters baked in (24 statements) 0.372 -> 0.355 ms, the generic
        // quaternion form (57 statements, four loop-carried components to keep per pixel) 0.75 -> 0.86 ms
        if (fetches || stmts < 4 || stmts > (force ? 400 : 40)) return false;
        const bool dbg = getenv("MMHIP_PAIR_DEBUG") != nullptr;
        for (int i = 0; i < 4; ++i)
            if (!code.result[i] || !pair_scalar_ty(code.result[i]->var->type)) {
                if (dbg) fprintf(stderr, "pair mode: result %d is not an int / float value\n", i);
                return false;
            }
        const bool ok = pair_block_ok(code.body);
        if (dbg) fprintf(stderr, "pair mode: body %s (%d statements)\n", ok ? "covered" : "not covered", stmts);
        return ok;
    }
    // Int values that only ever hold a truth value (results of comparisons, NOT, the literals 0 / 1 and
    // phis / copies of such): kept as a pair of bools (mm_bb), which the compiler keeps in scalar lane masks
    // where their logic is scalar arithmetic -- as ints they would cost two vector instructions per operation.
    std::set<const Value *> pair_bools;
    // a frame constant (scalar, defined in the hoisted slice) that holds a truth value
    static bool pair_const_is_bool(const Value *v, int depth) {
        if (!v || depth > 12 || v->var->type != Ty::Int) return false;
        if (v->index < 0) return true;                      // uninitialised: reads as 0
        const Stmt *d = v->def;
        if (!d) return false;
        auto prim_ok = [&](const Primary &p) {
            if (p.kind == Primary::IntConst) return p.i == 0 || p.i == 1;
            return p.kind == Primary::Val && pair_const_is_bool(p.value, depth + 1);
        };
        if (d->kind == Stmt::Phi) return d->rhs.kind == Rhs::Prim && d->rhs2.kind == Rhs::Prim && prim_ok(d->rhs.prim) && prim_ok(d->rhs2.prim);
        if (d->kind != Stmt::Assign) return false;
        if (d->rhs.kind == Rhs::Prim) return prim_ok(d->rhs.prim);
        if (d->rhs.kind != Rhs::Op) return false;
        const char *cn = d->rhs.op->cname;
        return !strcmp(cn, "LESS") || !strcmp(cn, "LEQ") || !strcmp(cn, "EQ") || !strcmp(cn, "NOT");
    }
    bool pair_prim_bool(const Primary &p) const {
        if (p.kind == Primary::IntConst) return p.i == 0 || p.i == 1;
        if (p.kind != Primary::Val) return false;
        if (p.value->index < 0) return p.value->var->type == Ty::Int;
        if (pair_bools.count(p.value)) return true;
        return !pair_defs.count(p.value) && pair_const_is_bool(p.value, 0);
    }
    void pair_collect_int_defs(const Block &b, std::vector<const Stmt *> &defs) const {
        for (const Stmt *s : b) {
            if (!s->in_pixel) continue;
            if (s->kind == Stmt::Assign && s->lhs->var->type == Ty::Int) defs.push_back(s);
            if (s->kind == Stmt::If || s->kind == Stmt::While)
                for (const Stmt *ph : s->phis) if (ph->in_pixel && ph->lhs->var->type == Ty::Int) defs.push_back(ph);
            if (s->kind == Stmt::If) { pair_collect_int_defs(s->then_, defs); pair_collect_int_defs(s->else_, defs); }
            if (s->kind == Stmt::While) pair_collect_int_defs(s->body, defs);
        }
    }
    void pair_infer_bools() {
        std::vector<const Stmt *> defs;
        pair_collect_int_defs(code.body, defs);
        for (Value *v : pix_defs) pair_defs.insert(v);
        for (const Stmt *d : defs) pair_bools.insert(d->lhs);            // optimistic, then remove until stable
        for (bool changed = true; changed;) {
            changed = false;
            for (const Stmt *d : defs) {
                if (!pair_bools.count(d->lhs)) continue;
                bool ok;
                if (d->kind == Stmt::Phi) ok = d->rhs.kind == Rhs::Prim && d->rhs2.kind == Rhs::Prim && pair_prim_bool(d->rhs.prim) && pair_prim_bool(d->rhs2.prim);
                else if (d->rhs.kind == Rhs::Prim) ok = pair_prim_bool(d->rhs.prim);
                else if (d->rhs.kind == Rhs::Op) {
                    const char *cn = d->rhs.op->cname;
                    ok = !strcmp(cn, "LESS") || !strcmp(cn, "LEQ") || !strcmp(cn, "EQ") || !strcmp(cn, "NOT");
                } else ok = false;
                if (!ok) { pair_bools.erase(d->lhs); changed = true; }
            }
        }
    }
    // values read outside the body of loop `w` (statements of other blocks, other loops' phis, the results):
    // only those of w's phis, and its condition, have to keep their value once a pixel has left the loop
    void pair_uses(const Block &b, const Stmt *skip, std::set<const Value *> &uses) const {
        auto use = [&](const Primary &p) { if (p.kind == Primary::Val) uses.insert(p.value); };
        auto use_rhs = [&](const Rhs &r) { if (r.kind == Rhs::Prim) use(r.prim); for (const Primary &a : r.args) use(a); };
        for (const Stmt *s : b) {
            if (!s->in_pixel) continue;
            if (s->kind == Stmt::Assign) use_rhs(s->rhs);
            if (s->kind == Stmt::If) {
                use_rhs(s->cond);
                pair_uses(s->then_, skip, uses);
                pair_uses(s->else_, skip, uses);
                for (const Stmt *ph : s->phis) if (ph->in_pixel) { use_rhs(ph->rhs); use_rhs(ph->rhs2); }
            }
            if (s->kind == Stmt::While) {
                for (const Stmt *ph : s->phis) if (ph->in_pixel) { use_rhs(ph->rhs); if (s != skip) use_rhs(ph->rhs2); }
                if (s != skip) { use_rhs(s->cond); pair_uses(s->body, skip, uses); }
            }
        }
    }

    // operand as a 2-vector of the wanted type (mm_vf / mm_vi broadcast scalars and convert int -> float)
    std::string pbool(const Primary &p) {        // operand as mm_bb
        if (p.kind == Primary::IntConst) return p.i ? "mm_bu(true)" : "mm_bu(false)";
        if (p.kind == Primary::Val && p.value->index < 0) return "mm_bu(false)";
        if (pair_uniform.count(p.value)) return "mm_bu((bool)u" + vname(p.value) + ")";
        if (pair_bools.count(p.value)) return vname(p.value);
        return "mm_tob(" + pprim(p, Ty::Int) + ")";
    }
    std::string pprim(const Primary &p, Ty want) {
        const char *w = want == Ty::Float ? "mm_vf(" : "mm_vi(";
        if (p.kind == Primary::IntConst) return std::string(w) + std::to_string(p.i) + ")";
        if (p.kind == Primary::FloatConst) return std::string(w) + float_literal(p.f) + "f)";
        if (p.value->index < 0) return std::string(w) + "0)";
        if (pair_uniform.count(p.value))
            return std::string(w) + (p.value->var->type == Ty::Float ? "(float)u" : "(int)u") + vname(p.value) + ")";
        return std::string(w) + vname(p.value) + ")";
    }
    static Ty pair_arith_ty(const Rhs &r) {
        for (const Primary &a : r.args) if (a.type() == Ty::Float) return Ty::Float;
        return Ty::Int;
    }
    std::string prhs(const Rhs &r, const Value *lhs) {
        const Ty lhs_ty = lhs->var->type;
        const bool as_bool = pair_bools.count(lhs) > 0;
        if (r.kind == Rhs::Prim) return as_bool ? pbool(r.prim) : pprim(r.prim, lhs_ty);
        if (r.kind == Rhs::Internal) return r.internal == "x" ? "mm_vf(x)" : "mm_y2";
        const char *cn = r.op->cname;
        const Ty t = pair_arith_ty(r);
        // An int value next to a float literal: the scalar kernel's literal is a double (cc.c prints them so),
        // C computes (double)i op literal and the assignment rounds once -- (float)i first would round twice
        // for |i| >= 2^24.  Same arithmetic here, per component.
        if (r.args.size() == 2 && ((r.args[0].type() == Ty::Int && r.args[1].kind == Primary::FloatConst) ||
                                   (r.args[1].type() == Ty::Int && r.args[0].kind == Primary::FloatConst))) {
            const char *op = !strcmp(cn, "ADD") ? "+" : !strcmp(cn, "SUB") ? "-" : !strcmp(cn, "MUL") ? "*" : !strcmp(cn, "LESS") ? "<"
                             : !strcmp(cn, "LEQ") ? "<=" : !strcmp(cn, "EQ") ? "==" : nullptr;
            if (op) {
                auto comp = [&](const Primary &p, const char *c) {
                    if (p.kind == Primary::FloatConst) return "(double)" + float_literal(p.f) + "f";
                    return "(double)" + pprim(p, Ty::Int) + "." + c;
                };
                const std::string ex = comp(r.args[0], "x") + " " + op + " " + comp(r.args[1], "x");
                const std::string ey = comp(r.args[0], "y") + " " + op + " " + comp(r.args[1], "y");
                if (op[0] == '<' || op[0] == '=') {
                    const std::string b = "mm_bl(" + ex + ", " + ey + ")";
                    return as_bool ? b : "mm_vi(" + b + ")";
                }
                return "mm_pf{(float)(" + ex + "), (float)(" + ey + ")}";
            }
        }
        if (!strcmp(cn, "ADD")) return "(" + pprim(r.args[0], t) + " + " + pprim(r.args[1], t) + ")";
        if (!strcmp(cn, "SUB")) return "(" + pprim(r.args[0], t) + " - " + pprim(r.args[1], t) + ")";
        if (!strcmp(cn, "MUL")) return "(" + pprim(r.args[0], t) + " * " + pprim(r.args[1], t) + ")";
        if (!strcmp(cn, "NEG")) return "(-" + pprim(r.args[0], t) + ")";
        if (!strcmp(cn, "sqrt")) return "mm_sqrt2(" + pprim(r.args[0], Ty::Float) + ")";
        if (!strcmp(cn, "DIV")) {
            if (r.args[1].kind == Primary::FloatConst || r.args[1].kind == Primary::IntConst) {      // x / +-2^k = x * 2^-k, exactly
                int ex = 0;
                const float c = r.args[1].kind == Primary::FloatConst ? r.args[1].f : (float)r.args[1].i;
                if (std::isfinite(c) && c != 0.0f && std::fabs(std::frexp(c, &ex)) == 0.5f && ex > -100 && ex < 100)
                    return "(" + pprim(r.args[0], Ty::Float) + " * " + float_literal(1.0f / c) + "f)";
            }
            return "(" + pprim(r.args[0], Ty::Float) + " / " + pprim(r.args[1], Ty::Float) + ")";
        }
        // the truth-valued operators: a pair of bools; as an int (0 / 1) only if the value is used as one
        std::string b;
        if (!strcmp(cn, "NOT")) b = "mm_notb(" + pbool(r.args[0]) + ")";
        else if (!strcmp(cn, "EQ") && pair_prim_bool(r.args[0]) && pair_prim_bool(r.args[1])) {
            // b == 0 is !b, b == 1 is b, otherwise the equivalence of two truth values
            const Primary &x = r.args[0], &y = r.args[1];
            if (y.kind == Primary::IntConst) b = y.i ? pbool(x) : "mm_notb(" + pbool(x) + ")";
            else if (x.kind == Primary::IntConst) b = x.i ? pbool(y) : "mm_notb(" + pbool(y) + ")";
            else b = "mm_eqb(" + pbool(x) + ", " + pbool(y) + ")";
        } else {
            bool done = false;
            if (!strcmp(cn, "LESS") && r.args[0].kind == Primary::Val && r.args[1].is_const()) {
                // sqrt(a) < 2^k  ->  0 <= a < 4^k, like the scalar generator (rhs() above)
                const Stmt *d = r.args[0].value->def;
                double k = r.args[1].kind == Primary::IntConst ? (double)r.args[1].i : (double)r.args[1].f;
                int ex = 0;
                bool pow2 = k > 0 && std::frexp(k, &ex) == 0.5 && ex > -50 && ex < 50;
                if (pow2 && d && d->kind == Stmt::Assign && d->rhs.kind == Rhs::Op && !strcmp(d->rhs.op->cname, "sqrt") &&
                    d->lhs->var->type == Ty::Float && d->rhs.args[0].type() == Ty::Float && d->rhs.args[0].kind == Primary::Val &&
                    value_visible(d->rhs.args[0].value, PIXEL)) {
                    const std::string a = pprim(d->rhs.args[0], Ty::Float), k2 = "mm_vf(" + float_literal((float)(k * k)) + "f)";
                    b = nonneg_or_nan(d->rhs.args[0].value, 0) ? "mm_lt(" + a + ", " + k2 + ")"
                                                                 : "mm_andb(mm_lt(" + a + ", " + k2 + "), mm_le(mm_vf(0.0f), " + a + "))";
                    done = true;
                }
            }
            if (!done) {
                const char *fn = !strcmp(cn, "LESS") ? "mm_lt(" : !strcmp(cn, "LEQ") ? "mm_le(" : "mm_eq(";
                b = fn + pprim(r.args[0], t) + ", " + pprim(r.args[1], t) + ")";
            }
        }
        return as_bool ? b : "mm_vi(" + b + ")";
    }
    static const char *pair_ctype(Ty t) { return t == Ty::Float ? "mm_pf" : "mm_pi"; }
    void pair_decls(const std::vector<Value *> &defs, const std::string &ind) {
        std::set<Value *> seen;
        for (Value *v : defs) {
            if (v->index < 0 || !seen.insert(v).second) continue;
            out << ind << (pair_bools.count(v) ? "mm_bb" : pair_ctype(v->var->type)) << " " << vname(v) << ";\n";
        }
    }
    std::string pval_as(const Primary &p, const Value *lhs) {      // operand in the representation of `lhs`
        return pair_bools.count(lhs) ? pbool(p) : pprim(p, lhs->var->type);
    }
    // Wave-uniform values inside a pair-mode loop.  A loop phi that starts from a literal or a frame constant
    // and is stepped by one -- `n = n + 1`, the iteration counter of every escape-time filter -- holds the same
    // value in every lane and in both pixels for as long as they are in the loop, and so does everything
    // computed from such values and loop-invariant scalars alone (`n + 1`, `n < 31`).  Those are kept as plain
    // scalars (the compiler holds them in SGPRs and evaluates them on the scalar unit: uniform inside a loop
    // with divergent exits) instead of as per-lane pairs; the per-pixel copy a phi needs after the loop is one
    // select per iteration instead of an add, a compare and a select.  Values of pixels that have left the loop
    // are don't-cares inside it, exactly as before.
    std::set<const Value *> pair_uniform;
    bool pair_invariant_scalar(const Primary &p) const {
        if (p.kind == Primary::IntConst || p.kind == Primary::FloatConst) return true;
        if (p.kind != Primary::Val) return false;
        return p.value->index < 0 || !pair_defs.count(p.value);          // uninitialised (0) or a frame constant
    }
    bool pair_uniform_operand(const Primary &p) const {
        return pair_invariant_scalar(p) || (p.kind == Primary::Val && pair_uniform.count(p.value));
    }
    void pair_mark_uniform(const Block &b) {      // forward pass over a loop body (SSA: definitions precede uses)
        for (const Stmt *s : b) {
            if (!s->in_pixel) continue;
            if (s->kind == Stmt::Assign && s->rhs.kind != Rhs::Internal) {
                bool ok = s->rhs.kind == Rhs::Prim ? pair_uniform_operand(s->rhs.prim) : s->rhs.kind == Rhs::Op;
                if (ok && s->rhs.kind == Rhs::Op)
                    for (const Primary &a : s->rhs.args) ok = ok && pair_uniform_operand(a);
                // a plain copy of an invariant scalar stays a broadcast (nothing to gain); ops on uniform values do not
                if (ok && s->rhs.kind == Rhs::Prim && !(s->rhs.prim.kind == Primary::Val && pair_uniform.count(s->rhs.prim.value))) ok = false;
                // truth values stay pairs of bools built from the (broadcast) uniform operands: the compiler evaluates such a
                // comparison on the scalar unit anyway, and keeps the pair in lane masks only in that form (checked in the ISA)
                if (ok && pair_bools.count(s->lhs)) ok = false;
                if (ok) pair_uniform.insert(s->lhs);
            } else if (s->kind == Stmt::If) {
                pair_mark_uniform(s->then_);
                pair_mark_uniform(s->else_);
            }
        }
    }
    // the phis of `w` that are uniform induction variables; marks them and what follows from them
    std::vector<const Stmt *> pair_find_uniform_ivs(const Stmt *w) {
        std::vector<const Stmt *> ivs;
        if (getenv("MMHIP_PAIR_NO_UNIFORM")) return ivs;
        for (const Stmt *ph : w->phis) {
            if (!ph->in_pixel || ph->rhs.kind != Rhs::Prim || ph->rhs2.kind != Rhs::Prim) continue;
            if (!pair_scalar_ty(ph->lhs->var->type) || pair_bools.count(ph->lhs)) continue;
            if (!pair_invariant_scalar(ph->rhs.prim) || ph->rhs2.prim.kind != Primary::Val) continue;
            const Stmt *d = ph->rhs2.prim.value->def;
            if (!d || d->kind != Stmt::Assign || d->parent != w || d->rhs.kind != Rhs::Op || d->rhs.args.size() != 2) continue;
            const char *cn = d->rhs.op->cname;
            const Primary &a0 = d->rhs.args[0], &a1 = d->rhs.args[1];
            const bool self0 = a0.kind == Primary::Val && a0.value == ph->lhs, self1 = a1.kind == Primary::Val && a1.value == ph->lhs;
            const bool step = (!strcmp(cn, "ADD") && ((self0 && pair_invariant_scalar(a1)) || (self1 && pair_invariant_scalar(a0)))) ||
                              (!strcmp(cn, "SUB") && self0 && pair_invariant_scalar(a1));
            if (!step) continue;
            ivs.push_back(ph);
            pair_uniform.insert(ph->lhs);
        }
        if (!ivs.empty()) pair_mark_uniform(w->body);
        return ivs;
    }

    // `mask`: the expression (mm_bb) under which the block runs
    void pair_stmts(Block &b, const std::string &ind, const std::string &mask) {
        for (Stmt *s : b) {
            if (!s->in_pixel) continue;
            switch (s->kind) {
                case Stmt::Assign:
                    if (pair_uniform.count(s->lhs)) {       // scalar statement, the scalar kernel's own expression
                        const char *ty = pair_bools.count(s->lhs) ? "bool" : s->lhs->var->type == Ty::Float ? "float" : "int";
                        out << ind << "const " << ty << " u" << vname(s->lhs) << " = " << rhs(s->rhs, PIXEL, s, s->lhs->var) << ";\n";
                        break;
                    }
                    out << ind << vname(s->lhs) << " = " << prhs(s->rhs, s->lhs) << ";\n";
                    break;
                case Stmt::If: {
                    const std::string c = "mm_c" + std::to_string(pair_ids++);
                    out << ind << "const mm_bb " << c << " = " << pbool(s->cond.prim) << ";\n";
                    pair_stmts(s->then_, ind, "mm_andb(" + mask + ", " + c + ")");
                    pair_stmts(s->else_, ind, "mm_andb(" + mask + ", mm_notb(" + c + "))");
                    for (Stmt *ph : s->phis)
                        if (ph->in_pixel)
                            out << ind << vname(ph->lhs) << " = mm_sel2(" << c << ", " << pval_as(ph->rhs.prim, ph->lhs) << ", "
                                << pval_as(ph->rhs2.prim, ph->lhs) << ");\n";
                    break;
                }
                case Stmt::While: {
                    const std::string a = "mm_a" + std::to_string(pair_ids++);
                    std::set<const Value *> outside;
                    pair_uses(code.body, s, outside);
                    for (int i = 0; i < 4; ++i) outside.insert(code.result[i]);
                    if (s->cond.prim.kind == Primary::Val) outside.insert(s->cond.prim.value);
                    for (Stmt *ph : s->phis)
                        if (ph->in_pixel) out << ind << vname(ph->lhs) << " = " << pval_as(ph->rhs.prim, ph->lhs) << ";\n";
                    // uniform induction variables: a scalar twin, read inside the loop instead of the pair
                    const std::vector<const Stmt *> ivs = pair_find_uniform_ivs(s);
                    for (const Stmt *ph : ivs) {
                        pair_uniform.erase(ph->lhs);       // the initial value is printed with the ordinary names
                        const std::string init = prim(ph->rhs.prim, PIXEL);
                        pair_uniform.insert(ph->lhs);
                        out << ind << (ph->lhs->var->type == Ty::Float ? "float u" : "int u") << vname(ph->lhs) << " = " << init << ";\n";
                    }
                    out << ind << "mm_bb " << a << " = mm_andb(" << mask << ", " << pbool(s->cond.prim) << ");\n";
                    out << ind << "while (" << a << ".x | " << a << ".y) {\n";
                    pair_stmts(s->body, ind + "  ", a);
                    // back edge: a parallel copy (temporaries first).  A phi that is read after the loop, and the
                    // loop condition, keep their value once their pixel has left the loop; the others may run on.
                    int k = 0;
                    for (Stmt *ph : s->phis)
                        if (ph->in_pixel) {
                            const std::string ty = pair_bools.count(ph->lhs) ? "mm_bb" : pair_ctype(ph->lhs->var->type);
                            const std::string nv = pval_as(ph->rhs2.prim, ph->lhs);
                            out << ind << "  const " << ty << " " << a << "_n" << k++ << " = "
                                << (outside.count(ph->lhs) ? "mm_sel2(" + a + ", " + nv + ", " + vname(ph->lhs) + ")" : nv) << ";\n";
                        }
                    k = 0;
                    for (Stmt *ph : s->phis)
                        if (ph->in_pixel) out << ind << "  " << vname(ph->lhs) << " = " << a << "_n" << k++ << ";\n";
                    for (const Stmt *ph : ivs) out << ind << "  u" << vname(ph->lhs) << " = " << prim(ph->rhs2.prim, PIXEL) << ";\n";
                    out << ind << "  " << a << " = mm_andb(" << a << ", " << pbool(s->cond.prim) << ");\n";
                    out << ind << "}\n";
                    // after the loop a phi is read through its per-pixel copy (frozen at the pixel's own exit)
                    for (const Stmt *ph : ivs) pair_uniform.erase(ph->lhs);
                    break;
                }
                default: break;
            }
        }
    }

    // sin(v) and cos(v) of the same float value in one block (toXY of an `ra` filter): evaluated
    // together by mmf_sincos_f32 -- one argument reduction -- at the first of the two statements.
    static bool is_fast_sincos(const Stmt *s, bool *is_sin) {
        if (s->kind != Stmt::Assign || s->rhs.kind != Rhs::Op || s->rhs.args.size() != 1) return false;
        const char *cn = s->rhs.op->cname;
        if (strcmp(cn, "sin") && strcmp(cn, "cos")) return false;
        if (!s->lhs || s->lhs->var->type != Ty::Float || s->rhs.args[0].type() != Ty::Float || s->rhs.args[0].kind != Primary::Val)
            return false;
        *is_sin = !strcmp(cn, "sin");
        return true;
    }
    struct SinCosRole { int id; bool first; };
    std::map<const Stmt *, SinCosRole> sincos_role;
    std::set<const Block *> sincos_scanned;
    int sincos_ids = 0;
    void pair_sincos(Block &b, Slice sl) {
        for (size_t i = 0; i < b.size(); ++i) {
            bool sin_i, sin_j;
            Stmt *si = b[i];
            if (!(sl == PROLOGUE ? si->hoisted : si->in_pixel) || sincos_role.count(si) || !is_fast_sincos(si, &sin_i)) continue;
            for (size_t j = i + 1; j < b.size(); ++j) {
                Stmt *sj = b[j];
                if (!(sl == PROLOGUE ? sj->hoisted : sj->in_pixel) || sincos_role.count(sj) || !is_fast_sincos(sj, &sin_j)) continue;
                if (sin_j == sin_i || sj->rhs.args[0].value != si->rhs.args[0].value) continue;
                sincos_role[si] = SinCosRole{sincos_ids, true};
                sincos_role[sj] = SinCosRole{sincos_ids, false};
                ++sincos_ids;
                break;
            }
        }
    }

    void stmts(Block &b, Slice sl, const std::string &ind) {
        if (opt.fast_math_exact && sl == PIXEL && sincos_scanned.insert(&b).second) pair_sincos(b, sl);
        for (Stmt *s : b) {
            bool mine = sl == PROLOGUE ? s->hoisted : sl == ROWS ? s->in_row : s->in_pixel;
            if (!mine) continue;
            switch (s->kind) {
                case Stmt::Assign: {
                    auto sc = sl == PIXEL ? sincos_role.find(s) : sincos_role.end();
                    if (sc != sincos_role.end()) {
                        const std::string t = "mm_sc" + std::to_string(sc->second.id);
                        if (sc->second.first)
                            out << ind << "const mmf_sincos_t " << t << " = mmf_sincos_f32(" << prim(s->rhs.args[0], sl) << ");\n";
                        out << ind << vname(s->lhs) << " = " << t << (!strcmp(s->rhs.op->cname, "sin") ? ".s" : ".c") << ";\n";
                        break;
                    }
                    out << ind << vname(s->lhs) << " = " << rhs(s->rhs, sl, s, s->lhs->var) << ";\n";
                    break;
                }
                case Stmt::If:
                    out << ind << "if (" << rhs(s->cond, sl, s, nullptr) << ") {\n";
                    stmts(s->then_, sl, ind + "  ");
                    phis(s->phis, 0, sl, ind + "  ");
                    out << ind << "} else {\n";
                    stmts(s->else_, sl, ind + "  ");
                    phis(s->phis, 1, sl, ind + "  ");
                    out << ind << "}\n";
                    break;
                case Stmt::While: {
                    // a loop of both slices whose calls the prologue numbers: the pixel slice starts counting where the
                    // prologue stood when it entered the loop
                    auto db = dual_base_off.find(s);
                    if (db != dual_base_off.end() && sl == PROLOGUE)
                        out << ind << "*(int *)(XY + " << db->second << ") = ((const int *)(XY + " << ks.native_ctr_offset << "))[1];\n";
                    if (db != dual_base_off.end() && sl == PIXEL)
                        out << ind << "mm_dyn_ctr = *(const int *)(XY + " << db->second << ");\n";
                    phis(s->phis, 0, sl, ind);
                    out << ind << "while (" << rhs(s->cond, sl, s, nullptr) << ") {\n";
                    stmts(s->body, sl, ind + "  ");
                    phis(s->phis, 1, sl, ind + "  ");
                    out << ind << "}\n";
                    break;
                }
                default: break;
            }
        }
    }

    // `null_images`: image handles start as the null image -- in the prologue, whose every transferred value is stored at
    // the end whether or not the branch that assigns it ran (the pixel kernel loads the descriptors of transferred images
    // up front)
    void decls(const std::vector<Value *> &defs, const std::string &ind, bool null_images = false) {
        std::set<Value *> seen;
        for (Value *v : defs) {
            if (v->index < 0 || !seen.insert(v).second) continue;
            out << ind << ctype(v->var) << " " << vname(v) << (null_images && v->var->type == Ty::Image ? " = mm_null_image()" : "") << ";\n";
        }
    }

    static unsigned long long fnv(const std::string &s) {
        unsigned long long h = 1469598103934665603ull;
        for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
        return h;
    }

    void run() {
        analyze_and_layout();
        emit_source();
    }

    std::vector<Value *> pro_defs, pix_defs;
    std::set<Value *> pro_uses, pix_uses;

    // Follows plain copies (and phis-free assignments of a primary) to the defining statement.
    static const Stmt *def_through_copies(const Primary &p) {
        const Primary *q = &p;
        for (int guard = 0; guard < 64; ++guard) {
            if (q->kind != Primary::Val || !q->value->def) return nullptr;
            const Stmt *d = q->value->def;
            if (d->kind != Stmt::Assign) return nullptr;
            if (d->rhs.kind == Rhs::Prim) { q = &d->rhs.prim; continue; }
            return d;
        }
        return nullptr;
    }
    // The filter whose pixel is nothing but the result of native call k sampled at the pixel's own
    // coordinates -- `soft = gaussian_blur(in, ...); soft(xy)`, examples/Blur/Gaussian Blur.mm.  The
    // host may then let the native filter's last kernel write the output pixels itself (when the
    // sample positions are the pixel centres, which it checks) and skip the pixel kernel.
    int find_direct_native() const {
        const Stmt *fetch = nullptr;
        for (int i = 0; i < 4; ++i) {
            if (!code.result[i]) return -1;
            const Stmt *d = def_through_copies(Primary::V(code.result[i]));
            if (!d || d->rhs.kind != Rhs::Op || strcmp(d->rhs.op->cname, "TUPLE_NTH") || d->rhs.args.size() != 2 ||
                d->rhs.args[1].kind != Primary::IntConst || d->rhs.args[1].i != i)
                return -1;
            const Stmt *t = def_through_copies(d->rhs.args[0]);
            if (!t || (fetch && t != fetch)) return -1;
            fetch = t;
        }
        if (fetch->rhs.kind != Rhs::Op || strcmp(fetch->rhs.op->cname, "ORIG_VAL") || fetch->rhs.args.size() < 3) return -1;
        const Stmt *dx = def_through_copies(fetch->rhs.args[0]), *dy = def_through_copies(fetch->rhs.args[1]);
        if (!dx || dx->rhs.kind != Rhs::Internal || dx->rhs.internal != "x") return -1;
        if (!dy || dy->rhs.kind != Rhs::Internal || dy->rhs.internal != "y") return -1;
        const Stmt *img = def_through_copies(fetch->rhs.args[2]);
        auto it = img ? native_index.find(img) : native_index.end();
        return it == native_index.end() || ks.natives[it->second].in_loop ? -1 : it->second;
    }

    // ---- the per-row slice ------------------------------------------------------------------
    // The reference evaluates code that depends on y alone once per row (its "x-const" slice, new_template.c.in:251-253,
    // compiler.c:4550-4611).  Here: top-level assignments of the pixel slice whose operands are literals, frame constants,
    // the row coordinate or other such values form the row slice -- when it contains a library call (everything else is
    // cheaper to recompute per pixel than to load) -- and a kernel of its own, mm_rows, evaluates it once per row of the
    // launch; the pixel kernel reads the values it needs from mm_args.rowtab like it reads y from ytab.
    std::vector<Value *> row_transfer;        // row values the pixel slice uses, in table order
    static bool row_scalar(Ty t) { return t == Ty::Int || t == Ty::Float || t == Ty::Complex; }
    static bool row_expensive(const char *cn) {
        static const char *cheap[] = {"fabs", "floor", "ceil", "crealf", "cimagf"};
        for (const char *c : cheap) if (!strcmp(cn, c)) return false;
        return std::islower((unsigned char)cn[0]) || !strncmp(cn, "ELL_", 4) || !strcmp(cn, "GAMMA");
    }
    void uses_of_pixel_code(const Block &b, std::set<const Value *> &used) const {
        auto use = [&](const Rhs &r) {
            if (r.kind == Rhs::Prim && r.prim.kind == Primary::Val) used.insert(r.prim.value);
            for (const Primary &p : r.args) if (p.kind == Primary::Val) used.insert(p.value);
        };
        for (const Stmt *s : b) {
            if (!s->in_pixel) continue;
            switch (s->kind) {
                case Stmt::Assign: use(s->rhs); break;
                case Stmt::Phi: use(s->rhs); use(s->rhs2); break;
                case Stmt::If: use(s->cond); uses_of_pixel_code(s->then_, used); uses_of_pixel_code(s->else_, used); uses_of_pixel_code(s->phis, used); break;
                case Stmt::While: uses_of_pixel_code(s->phis, used); use(s->cond); uses_of_pixel_code(s->body, used); break;
                default: break;
            }
        }
    }
    void find_row_slice() {
        if (fn_root || in_function || getenv("MMHIP_NO_ROW_SLICE")) return;
        std::vector<Stmt *> cand;
        std::set<const Value *> rows;
        auto operand_ok = [&](const Primary &p) {
            return p.kind != Primary::Val || p.value->index < 0 || p.value->hoisted || rows.count(p.value) > 0;
        };
        for (Stmt *s : code.body) {           // top level only: no control flow in the row slice
            if (s->kind != Stmt::Assign || !s->in_pixel || s->hoisted || !row_scalar(s->lhs->var->type)) continue;
            const Rhs &r = s->rhs;
            bool ok = false;
            if (r.kind == Rhs::Internal) ok = r.internal == "y";
            else if (r.kind == Rhs::Prim) ok = operand_ok(r.prim) && r.prim.kind == Primary::Val && rows.count(r.prim.value);
            else if (r.kind == Rhs::Op && r.op->pure) {
                ok = true;
                bool any_row = false;
                for (const Primary &p : r.args) {
                    ok = ok && operand_ok(p) && (p.kind != Primary::Val || p.value->index < 0 || row_scalar(p.value->var->type));
                    any_row = any_row || (p.kind == Primary::Val && rows.count(p.value));
                }
                ok = ok && any_row && strncmp(r.op->cname, "USERVAL_", 8) != 0;
            }
            if (!ok) continue;
            cand.push_back(s);
            rows.insert(s->lhs);
        }
        // what the pixel code reads must travel as a 32-bit word: a complex row value that is used per pixel goes back to
        // the pixel slice, and with it whatever was computed from it
        for (bool changed = true; changed;) {
            changed = false;
            for (Stmt *s : cand) s->in_pixel = !rows.count(s->lhs);
            std::set<const Value *> used;
            uses_of_pixel_code(code.body, used);
            for (int i = 0; i < 4; ++i) used.insert(code.result[i]);
            for (Stmt *s : cand) {
                if (!rows.count(s->lhs)) continue;
                bool drop = used.count(s->lhs) && s->lhs->var->type == Ty::Complex;
                if (s->rhs.kind == Rhs::Prim) drop = drop || !rows.count(s->rhs.prim.value);
                for (const Primary &p : s->rhs.args)
                    drop = drop || (p.kind == Primary::Val && p.value->index >= 0 && !p.value->hoisted && !rows.count(p.value));
                if (drop) { rows.erase(s->lhs); changed = true; }
            }
        }
        bool expensive = false;
        for (Stmt *s : cand)
            expensive = expensive || (rows.count(s->lhs) && s->rhs.kind == Rhs::Op && row_expensive(s->rhs.op->cname));
        std::set<const Value *> used;
        for (Stmt *s : cand) s->in_pixel = !rows.count(s->lhs);
        uses_of_pixel_code(code.body, used);
        for (int i = 0; i < 4; ++i) used.insert(code.result[i]);
        std::vector<Value *> transfer;
        for (Stmt *s : cand) if (rows.count(s->lhs) && used.count(s->lhs)) transfer.push_back(s->lhs);
        if (!expensive || transfer.empty() || transfer.size() > 24) {
            for (Stmt *s : cand) s->in_pixel = true;
            return;
        }
        for (Stmt *s : cand)
            if (rows.count(s->lhs)) { s->in_row = true; s->in_pixel = false; s->lhs->row_const = true; }
        row_transfer = transfer;
        ks.row_values = (int)transfer.size();
        ks.rows_name = "mm_rows";
    }
    // the row values a pixel of row `row` needs, from the table mm_rows filled
    void row_loads(const std::string &ind, const char *row) {
        for (size_t k = 0; k < row_transfer.size(); ++k) {
            Value *v = row_transfer[k];
            const std::string at = "A.rowtab[" + std::to_string(k) + " * A.num_rows + " + row + "]";
            out << ind << "const " << ctype(v->var) << " " << vname(v) << " = "
                << (v->var->type == Ty::Int ? "__float_as_int(" + at + ")" : at) << ";\n";
        }
    }

    void analyze_and_layout() {
        find_natives(code.body);
        ks.native_sites = (int)ks.natives.size();
        // A call site inside a loop makes one call per iteration, each with a result of its own (the next iteration, or
        // the code behind the loop, may read it): such calls are numbered as the prologue makes them and recorded in
        // dynamic entries behind the sites'.  The host runs all recorded calls in the order they were made.
        bool any_in_loop = false;
        for (const NativeCall &nc : ks.natives) any_in_loop = any_in_loop || nc.in_loop;
        if (any_in_loop)
            for (int n = 0; n < MM_NATIVE_DYN_CALLS; ++n) {
                NativeCall nc;
                nc.dynamic = true;
                ks.natives.push_back(nc);
            }
        ks.direct_native = find_direct_native();
        find_row_slice();
        std::vector<Value *> row_defs;
        std::set<Value *> row_uses;
        collect_values(code.body, PROLOGUE, pro_defs, pro_uses);
        collect_values(code.body, PIXEL, pix_defs, pix_uses);
        collect_values(code.body, ROWS, row_defs, row_uses);
        for (Value *v : row_uses) if (v && v->hoisted) pix_uses.insert(v);      // frame constants the row slice reads travel in XY too
        for (int i = 0; i < 4; ++i) pix_uses.insert(code.result[i]);
        std::set<Value *> pix_def_set(pix_defs.begin(), pix_defs.end());
        for (Value *v : row_transfer) pix_def_set.insert(v);                    // defined (loaded) per pixel
        int off = 0;
        // native call records first (fixed layout the host can parse)
        for (NativeCall &nc : ks.natives) {
            nc.record_offset = off;
            off += MM_NATIVE_REC_BYTES;
        }
        if (!ks.natives.empty()) {
            ks.native_ctr_offset = off;
            off += 16;
        }
        for (Value *v : pro_defs) {
            if (!pix_uses.count(v) || pix_def_set.count(v) || transfer_off.count(v)) continue;
            int sz = type_size(v->var);
            int align = sz >= 8 ? 8 : 4;
            off = (off + align - 1) / align * align;
            transfer_off[v] = off;
            transfer_order.push_back(v);
            off += sz;
        }
        {
            std::function<void(Block &)> find_dual = [&](Block &b) {
                for (Stmt *s : b) {
                    if (s->kind == Stmt::If) { find_dual(s->then_); find_dual(s->else_); }
                    if (s->kind != Stmt::While) continue;
                    if (s->hoisted && s->in_pixel) {        // outermost: loops inside it go on counting
                        off = (off + 3) / 4 * 4;
                        dual_base_off[s] = off;
                        off += 4;
                    } else
                        find_dual(s->body);
                }
            };
            if (ks.native_sites < (int)ks.natives.size()) find_dual(code.body);
        }
        ks.xy_bytes = (off + 15) / 16 * 16;
        if (ks.xy_bytes == 0) ks.xy_bytes = 16;
        ks.has_prologue = !pro_defs.empty();
        // a value used by the pixel slice must be defined there or transferred
        for (Value *v : pix_uses)
            if (v && v->index >= 0 && !pix_def_set.count(v) && !transfer_off.count(v))
                throw CompileError("internal: value " + vname(v) + " used in the pixel kernel but defined nowhere");
    }

    void emit_source() {
        int tw = opt.tile_w;
        if (const char *e = getenv("MMHIP_TILE_W")) tw = atoi(e);      // experiments (tools/ab_unroll.sh)
        if (tw != 8 && tw != 16 && tw != 32 && tw != 64 && tw != 128 && tw != 256) tw = auto_tile_w();
        ks.tile_w = tw;
        ks.tile_h = 256 / tw;
        {
            // non-temporal output stores keep the frame from displacing the *input* in the caches: for kernels that fetch
            // (a kernel that reads nothing gains nothing, and its 64-byte row segments then reach memory uncombined:
            // Mandelbrot 8192^2 wrote 347 MB instead of 268 MB in the WRITE_SIZE counter, same time)
            int stmts = 0, fetches = 0;
            pixel_stats(code.body, stmts, fetches);
            int nt = fetches > 0;
            if (const char *e = getenv("MMHIP_NT_STORE")) nt = atoi(e);
            out << "#define MM_NT_STORE " << nt << "\n";
            // Workgroup -> tile order.  Workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  Giving every
            // XCD one contiguous band of tiles (1) lets neighbouring gathers share an L2 -- and makes every XCD's share of
            // the work depend on *where* in the frame the work is: the rows of a Mandelbrot frame that cross the set iterate
            // 2-3 times longer than its top and bottom rows, Droste's level loop runs for some regions only, and the XCDs
            // that own the cheap bands idle while the others finish (first seen as two processes sharing the GPU rendering
            // 20 % more frames than one: the idle XCDs took the other process's workgroups).  Tiles in dispatch order (0)
            // spread every region over all XCDs but put horizontal neighbours on different L2s (Pond fetched 2.8x its
            // input).  The default (2) deals runs of about one tile row to the XCDs in turn: balanced like (0), and a row's
            // tiles share an L2 like in (1).  A/B at 8192^2, ms for orders 0 / 1 / 2 (tools/ab_xcd_order.sh,
            // profiles/r03_ab_xcd_order.txt): Mandelbrot 0.209 / 0.263 / 0.201, Droste 0.860 / 1.217 / 0.855 (NoTransparency=1:
            // 0.853 / 0.905 / 0.852), Pond 0.569 / 0.567 / 0.556, Ident 0.140 / 0.133 / 0.133.
            int xo = 2;
            (void)stmts;
            if (const char *e = getenv("MMHIP_XCD_ORDER")) xo = atoi(e);
            out << "#define MM_XCD_ORDER " << xo << "\n";
        }
        if (const char *e = getenv("MMHIP_PAIR_MASKS")) out << "#define MM_PAIR_MASKS " << atoi(e) << "\n";
        if (getenv("MMHIP_NO_SAME_TAPS")) out << "#define MM_NO_SAME_TAPS 1\n";      // A/B switches
        if (getenv("MMHIP_NO_OUTSIDE_SHORTCUT")) out << "#define MM_NO_OUTSIDE_SHORTCUT 1\n";
        out << "#define MM_INTERSAMPLE " << opt.intersample << "\n";
        out << "#define MM_SUPERSAMPLING " << opt.supersampling << "\n";
        out << "#define MM_EDGE_X " << opt.edge_x << "\n#define MM_EDGE_Y " << opt.edge_y << "\n";
        if (opt.pixel_inc > 1) out << "#define MM_PIXEL_INC " << opt.pixel_inc << "\n";
        out << "#define MM_TILE_W " << ks.tile_w << "\n#define MM_TILE_H " << ks.tile_h << "\n";
        ks.unroll = opt.unroll > 0 ? opt.unroll : auto_unroll();
        pair_mode = opt.unroll <= 0 && !getenv("MMHIP_UNROLL") && ks.row_values == 0 && pair_eligible();   // (row values are per pixel of a pair)
        if (pair_mode) { ks.unroll = 2; pair_infer_bools(); }
        out << "#define MM_UNROLL " << ks.unroll << "\n";
        out << "#define MM_NATIVE_REC_BYTES " << (int)MM_NATIVE_REC_BYTES << "\n#define MM_NATIVE_DYN_CALLS " << (int)MM_NATIVE_DYN_CALLS << "\n";
        // float-argument sin/cos (mm_fastmath.h), the same text the host verifier compiles; it
        // precedes the device prelude, whose complex functions use mmf_sincos_d
        out << "#define MMF_FN static __device__ __forceinline__\n#define MMF_CONST_TABLE static __device__ const\n"
               "#define MMG_FN static __device__\n"
               "#define MMF_FMA(a, b, c) __builtin_fma((a), (b), (c))\n#define MMF_RINT(a) __builtin_rint((a))\n"
               "#define MMF_FABSF(a) __builtin_fabsf((a))\n#define MMF_FABS(a) __builtin_fabs((a))\n"
               "#define MMF_SIN_SLOW(a) sin((a))\n#define MMF_COS_SLOW(a) cos((a))\n"
               "#define MMF_LDEXP(a, e) __builtin_ldexp((a), (e))\n#define MMF_EXP_SLOW(a) exp((a))\n#define MMF_LOG_SLOW(a) log((a))\n"
               "#define MMF_POW_SLOW(a, b) pow((a), (b))\n"
               "#define MMF_ASINH_SLOW(a) asinh((a))\n#define MMF_ACOSH_SLOW(a) acosh((a))\n"
               "#define MMF_SQRT(a) __builtin_sqrt((a))\n#define MMF_HYPOT_SLOW(a, b) hypot((a), (b))\n"
               // mm_glibcf.h (glibc's float algorithms for the complex ops); float sqrt / division are the correctly
               // rounded device ones, double sqrt is the compiler's correctly rounded expansion
               "#define MMQ_FN static __device__ __forceinline__\n#define MMQ_TABLE static __device__ const\n"
               "#define MMQ_FMA(a, b, c) __builtin_fma((a), (b), (c))\n#define MMQ_SQRT(a) sqrt((a))\n"
               "#define MMQ_SQRTF(a) sqrtf((a))\n"
            << device_fastmath_prelude() << "\n";
        out << device_prelude() << "\n";
        bool noise = uses_noise(code.body);
        for (auto &fn : (fn_root ? fn_root : &code)->functions) noise = noise || uses_noise(fn->body);
        if (noise) {
            if (!noise_table_text())
                throw CompileError("the noise builtins need libnoise's gradient table, which was not available when this "
                                   "library was built (see tools/extract_noise_table.py)");
            out << noise_table_text() << device_noise_prelude() << "\n";
        }
        out << R"(
MM_DEV mm_image mm_closure_image(const mm_args &A, int closure_id) {
    mm_image im; im.idx = closure_id < 0 ? -1 : -2 - closure_id; im.pw = A.img_width; im.ph = A.img_height;
    im.xf = im.yf = 1.0f; im.resized = 0; return im;
}
struct mm_narg_t { int kind; int i; float f; mm_image img; };
MM_DEV mm_narg_t mm_narg(int v) { mm_narg_t a; a.kind = 0; a.i = v; a.f = (float)v; a.img = mm_null_image(); return a; }
MM_DEV mm_narg_t mm_narg(float v) { mm_narg_t a; a.kind = 1; a.i = (int)v; a.f = v; a.img = mm_null_image(); return a; }
MM_DEV mm_narg_t mm_narg(double v) { return mm_narg((float)v); }
MM_DEV mm_narg_t mm_narg(mm_image v) { mm_narg_t a; a.kind = 2; a.i = v.idx; a.f = 0.0f; a.img = v; return a; }
// Records a native-filter call for the host (which runs the filter's kernels between
// the prologue and the pixel kernel) and returns the handle of its result float map.
// ctr[0] counts the calls of this frame in the order they are made (the host runs them in that order).
MM_DEV mm_image mm_native_call(const mm_args &A, char *rec, int *ctr, int k, int nargs, mm_narg_t a0, mm_narg_t a1, mm_narg_t a2, mm_narg_t a3) {
    int *hdr = (int *)rec;
    hdr[0] = 1; hdr[1] = k; hdr[2] = nargs; hdr[3] = ctr[0]++;
    mm_narg_t *args = (mm_narg_t *)(rec + 16);
    args[0] = a0; args[1] = a1; args[2] = a2; args[3] = a3;
    mm_image im; im.idx = A.native_slot_base + k; im.pw = A.render_width; im.ph = A.render_height;
    im.xf = im.yf = 1.0f; im.resized = 0;
    return im;
}
// The same for a call site inside a loop: every call takes the next of the MM_NATIVE_DYN_CALLS dynamic entries behind
// the `sites' call sites (record, result slot); one call too many raises ctr[2] and the host refuses the frame.
MM_DEV mm_image mm_native_call_in_loop(const mm_args &A, char *dyn, int *ctr, int site, int sites, int nargs, mm_narg_t a0, mm_narg_t a1,
                                       mm_narg_t a2, mm_narg_t a3) {
    int n = ctr[1]++;
    if (n >= MM_NATIVE_DYN_CALLS) { ctr[2] = 1; n = MM_NATIVE_DYN_CALLS - 1; }
    mm_image im = mm_native_call(A, dyn + n * MM_NATIVE_REC_BYTES, ctr, site, nargs, a0, a1, a2, a3);
    im.idx = A.native_slot_base + sites + n;
    return im;
}
// (float) frame column / row of the pixel being evaluated (render_image's per-pixel loop, builtins.c:324-333)
#define __colF ((float)(col + A.region_x))
#define __rowF ((float)(rl + A.first_row))
#define MM_INTERNALS \
    const float t = A.t; const float R = A.R; const int frame = A.frame; \
    const int __canvasPixelW = A.img_width, __canvasPixelH = A.img_height; \
    const int __renderPixelW = A.render_width, __renderPixelH = A.render_height; \
    (void)t; (void)R; (void)frame; (void)__canvasPixelW; (void)__canvasPixelH; (void)__renderPixelW; (void)__renderPixelH;
)";
        if (ks.native_sites < (int)ks.natives.size())      // (only kernels with in-loop native calls: the others' text, and cache keys, stay as they were)
            out << R"(// The pixel slice's side of a loop that runs in both slices: its n-th call from an in-loop site is the prologue's n-th.
MM_DEV mm_image mm_native_result_in_loop(const mm_args &A, int &n, int sites) {
    mm_image im; im.idx = A.native_slot_base + sites + (n < MM_NATIVE_DYN_CALLS ? n : MM_NATIVE_DYN_CALLS - 1); ++n;
    im.pw = A.render_width; im.ph = A.render_height; im.xf = im.yf = 1.0f; im.resized = 0;
    return im;
}
)";
        emit_functions();
        // ---- prologue ----
        ks.prologue_uses_time = hoisted_uses_time(code.body);
        ks.prologue_name = "mm_prologue";
        ks.pixel_name = "mm_pixels";
        out << "extern \"C\" __global__ void __launch_bounds__(256) mm_prologue(mm_args A, char *XY) {\n";
        out << "  {\n    const int gid = blockIdx.x * 256 + threadIdx.x;\n"
               "    if (gid < A.region_width) A.xtab[gid] = CALC_VIRTUAL_X(gid + A.region_x, A.frame_render_width, A.sampling_offset_x);\n"
               "    if (gid < A.num_rows) A.ytab[gid] = CALC_VIRTUAL_Y(A.first_row + gid, A.frame_render_height, A.sampling_offset_y);\n"
               "    if (gid != 0) return;\n  }\n  MM_INTERNALS\n";
        if (!(fn_root ? fn_root : &code)->functions.empty()) out << "  const int col = 0, rl = 0; unsigned mm_rand_ctr = 0; (void)col; (void)rl; (void)mm_rand_ctr;\n";
        decls(pro_defs, "  ", true);
        if (!ks.natives.empty()) {      // no call recorded yet this frame
            for (const NativeCall &nc : ks.natives) out << "  *(int *)(XY + " << nc.record_offset << ") = 0;\n";
            out << "  { int *mm_ctr = (int *)(XY + " << ks.native_ctr_offset << "); mm_ctr[0] = mm_ctr[1] = mm_ctr[2] = mm_ctr[3] = 0; }\n";
        }
        stmts(code.body, PROLOGUE, "  ");
        for (Value *v : transfer_order)
            out << "  *(" << ctype(v->var) << " *)(XY + " << transfer_off[v] << ") = " << vname(v) << ";\n";
        out << "}\n\n";
        // ---- rows kernel: the per-row slice, one lane per row of the launch ----
        if (ks.row_values > 0) {
            out << "extern \"C\" __global__ void __launch_bounds__(256) mm_rows(mm_args A, const char *__restrict__ XY) {\n"
                   "  const int rl = blockIdx.x * 256 + threadIdx.x;\n"
                   "  if (rl >= A.num_rows) return;\n"
                   "  MM_INTERNALS\n"
                   "  const float y = A.ytab[rl];    // CALC_VIRTUAL_Y(first_row + rl, ...), by the prologue\n"
                   "  (void)y;\n";
            std::vector<Value *> rdefs;
            std::set<Value *> ruses;
            collect_values(code.body, ROWS, rdefs, ruses);
            for (Value *v : transfer_order)
                if (ruses.count(v))
                    out << "  const " << ctype(v->var) << " " << vname(v) << " = *(const " << ctype(v->var) << " *)(XY + " << transfer_off[v] << ");\n";
            decls(rdefs, "  ");
            stmts(code.body, ROWS, "  ");
            for (size_t k = 0; k < row_transfer.size(); ++k) {
                Value *v = row_transfer[k];
                out << "  A.rowtab[" << k << " * A.num_rows + rl] = "
                    << (v->var->type == Ty::Int ? "__int_as_float(" + vname(v) + ")" : vname(v)) << ";\n";
            }
            out << "}\n\n";
        }
        // ---- pixel kernel ----
        // experiment hook: ask the register allocator for a minimum occupancy (waves per SIMD)
        if (const char *e = getenv("MMHIP_WAVES_PER_EU")) out << "__attribute__((amdgpu_waves_per_eu(" << atoi(e) << "))) ";
        out << R"(extern "C" __global__ void __launch_bounds__(256) mm_pixels(mm_args A, const char *__restrict__ XY) {
  MM_INTERNALS
  // XCD-aware tile order (MM_XCD_ORDER, above): workgroups are dealt round-robin to the 8 XCDs; give each
  // XCD one contiguous band of tiles so neighbouring gathers share its L2 -- or, for a kernel that reads nothing,
  // take the tiles in dispatch order so that cheap and expensive regions of the frame are spread over all XCDs.
  const int tiles_x = (A.region_width + MM_TILE_W - 1) / MM_TILE_W;
  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
#if MM_XCD_ORDER == 1
  const int xcd = bid & 7, q = bid >> 3;
  const int per = nwg >> 3, rem = nwg & 7;
  const int swz = xcd * per + (xcd < rem ? xcd : rem) + q;
#elif MM_XCD_ORDER == 2
  // runs of C = 2^m consecutive tiles (m: one tile row or a little more) dealt to the XCDs in turn: neighbours in a row
  // share an L2 like in a band, and every XCD gets runs from all over the frame; the last partial round keeps dispatch order
  const int m = 32 - __builtin_clz((unsigned)(tiles_x > 1 ? tiles_x - 1 : 1));
  const int full = (nwg >> (m + 3)) << (m + 3);
  const int q = bid >> 3;
  const int swz = bid < full ? ((((q >> m) << 3) + (bid & 7)) << m) + (q & ((1 << m) - 1)) : bid;
#else
  const int swz = bid;
  (void)nwg;
#endif
  const int tile_y = A.tiles_magic ? (int)__umulhi((unsigned)swz, A.tiles_magic) : swz / tiles_x, tile_x = swz - tile_y * tiles_x;
  const int col = tile_x * MM_TILE_W + (threadIdx.x % MM_TILE_W);
  // a workgroup owns MM_TILE_W x (MM_TILE_H * A.ppt) pixels; each work-item walks A.ppt rows
  // MM_TILE_H apart, so wave start-up (kernarg / descriptor loads, tile arithmetic) and the
  // dispatcher's per-workgroup cost are paid once per A.ppt pixels
  const int row0 = tile_y * (MM_TILE_H * A.ppt) + (threadIdx.x / MM_TILE_W);   // row within this launch
  if (col >= A.region_width) return;
  const float x = A.xtab[col];   // CALC_VIRTUAL_X(col + region_x, ...), once per column
  (void)x;
)";
        for (Value *v : transfer_order)
            out << "  const " << ctype(v->var) << " " << vname(v) << " = *(const " << ctype(v->var) << " *)(XY + "
                << transfer_off[v] << ");\n";
        // Large bodies keep the one-pixel-per-work-item shape: a pixel loop makes every frame
        // constant live across it in SGPRs (after the loop's first store the compiler may not
        // re-issue scalar loads), and Droste's ~60 of them spilled to VGPR lanes -- 1.7x slower.
        // Such kernels are compute-bound; the per-workgroup dispatch cost does not show.
        {
            int stmts = 0, fetches = 0;
            pixel_stats(code.body, stmts, fetches);
            ks.single_pixel = stmts > 400 || transfer_order.size() > 24;
            if (const char *e = getenv("MMHIP_SINGLE_PIXEL")) ks.single_pixel = atoi(e) != 0;
        }
        if (ks.single_pixel) {
            ks.unroll = 1;
            out << "  const int rl = row0;   // A.ppt is 1 for this kernel (KernelSource::single_pixel)\n"
                   "  if (rl >= A.num_rows) return;\n"
                   "  const float y = A.ytab[rl];    // CALC_VIRTUAL_Y(first_row + rl, ...), once per row by the prologue\n"
                   "  unsigned mm_rand_ctr = 0;      // RAND call number within this pixel\n"
                   "  (void)y; (void)mm_rand_ctr;\n";
            if (!dual_base_off.empty())
                out << "  int mm_dyn_ctr = 0;            // in-loop native calls made so far by this pixel's copy of a loop of both slices\n";
            row_loads("  ", "rl");
            decls(pix_defs, "  ");
            stmts(code.body, PIXEL, "  ");
            out << "  mm_tup<4> rt;\n";
            for (int i = 0; i < 4; ++i) out << "  rt.v[" << i << "] = " << prim(Primary::V(code.result[i]), PIXEL) << ";\n";
            out << "  mm_store_pixel(A, rl, col, rt);\n}\n";
            finish_source();
            return;
        }
        // descriptors of frame-constant images, loaded (scalar) once before the pixel loop
        for (Value *v : transfer_order)
            if (v->var->type == Ty::Image) {
                out << "  const mm_image_desc " << vname(v) << "_desc = mm_load_desc(A, " << vname(v) << ");\n";
                preloaded_desc.insert(v);
            }
        // MM_UNROLL pixels are evaluated back to back before any of them is stored: with the
        // branch-free fetch their image loads are independent and overlap (one load per wave in
        // flight cannot cover HBM latency); rows past the end are computed on the last row and
        // simply not stored.  A.ppt is a multiple of MM_UNROLL (runtime.cpp).
        if (ks.single_pixel) pair_mode = false;
        auto emit_loop = [&](const char *ind, bool hot) {
            std::string I = ind;
            if (pair_mode) {
                // two pixels (rows mm_p and mm_p + 1 of this work-item's column) in lockstep, see pair_stmts
                out << "#pragma unroll 1\n" << I << "for (; mm_p < A.ppt; mm_p += 2) {\n"
                    << I << "  // vertically adjacent pixels: they mostly take the same path (a wave covers 16 x 8 pixels per step)\n"
                    << I << "  const int rl_a = row0 + (int)(threadIdx.x / MM_TILE_W) + mm_p * MM_TILE_H, rl_b = rl_a + 1;\n"
                    << I << "  const int row_a = rl_a < A.num_rows ? rl_a : A.num_rows - 1, row_b = rl_b < A.num_rows ? rl_b : A.num_rows - 1;\n"
                    << I << "  const mm_pf mm_y2 = {A.ytab[row_a], A.ytab[row_b]};    // CALC_VIRTUAL_Y per row, by the prologue\n";
                pair_decls(pix_defs, I + "  ");
                pair_stmts(code.body, I + "  ", "mm_bu(true)");
                out << I << "  mm_tup<4> mm_ra, mm_rb;\n";
                for (int i = 0; i < 4; ++i) {
                    const std::string v = pprim(Primary::V(code.result[i]), Ty::Float);
                    out << I << "  mm_ra.v[" << i << "] = " << v << ".x; mm_rb.v[" << i << "] = " << v << ".y;\n";
                }
                out << I << "  // a row past the end was evaluated as the last row: storing it there again writes the same bytes\n"
                    << I << "  mm_store_pixel(A, row_a, col, mm_ra);\n"
                    << I << "  mm_store_pixel(A, row_b, col, mm_rb);\n" << I << "}\n";
                return;
            }
            const bool fetched = hot && fetched_result;
            // The row coordinates of an iteration are loaded during the one before it (those of the first before the
            // loop): a work-item's iterations are a serial chain, and the table load in front of each would add one
            // memory round trip per iteration to it.  (Rows past the end read the last row's entry.)
            out << I << "float mm_y[MM_UNROLL];\n"
                << "#pragma unroll\n" << I << "for (int mm_u = 0; mm_u < MM_UNROLL; ++mm_u) {\n"
                << I << "  const int rl_raw = row0 + (mm_p + mm_u) * MM_TILE_H;\n"
                << I << "  mm_y[mm_u] = A.ytab[rl_raw < A.num_rows ? rl_raw : A.num_rows - 1];\n"
                << I << "}\n";
            out << "#pragma unroll 1\n" << I << "for (; mm_p < A.ppt; mm_p += MM_UNROLL) {\n"
                << I << (fetched ? "  mm_bilinear mm_rs[MM_UNROLL];\n" : "  mm_tup<4> mm_rt[MM_UNROLL];\n")
                << I << "  float mm_yn[MM_UNROLL];\n"
                << I << "  bool mm_bad = false;   // a hot fetch met a NaN / inf / > 2^31 px coordinate\n"
                << "#pragma unroll\n" << I << "  for (int mm_u = 0; mm_u < MM_UNROLL; ++mm_u) {\n"
                << I << "    const int rl_raw = row0 + (mm_p + MM_UNROLL + mm_u) * MM_TILE_H;\n"
                << I << "    mm_yn[mm_u] = A.ytab[rl_raw < A.num_rows ? rl_raw : A.num_rows - 1];\n"
                << I << "  }\n"
                << "#pragma unroll\n" << I << "  for (int mm_u = 0; mm_u < MM_UNROLL; ++mm_u) {\n"
                << I << "    const float y = mm_y[mm_u];    // CALC_VIRTUAL_Y(first_row + rl, ...), once per row by the prologue\n"
                << I << "    const int rl_u = row0 + (mm_p + mm_u) * MM_TILE_H;\n"
                << I << "    const int rl = rl_u < A.num_rows ? rl_u : A.num_rows - 1;\n"
                << I << "    unsigned mm_rand_ctr = 0;      // RAND call number within this pixel\n"
                << I << "    (void)y; (void)rl; (void)mm_rand_ctr;\n";
            if (!dual_base_off.empty())
                out << I << "    int mm_dyn_ctr = 0;            // in-loop native calls made so far by this pixel's copy of a loop of both slices\n";
            row_loads(I + "    ", "rl");
            decls(pix_defs, (I + "    ").c_str());
            stmts(code.body, PIXEL, (I + "    ").c_str());
            for (int i = 0; i < 4 && !fetched; ++i)
                out << I << "    mm_rt[mm_u].v[" << i << "] = " << prim(Primary::V(code.result[i]), PIXEL) << ";\n";
            out << I << "  }\n";
            if (hot)
                out << I << "  if (mm_bad) break;     // these pixels (and this work-item's remaining ones) take the generic loop below\n";
            out << "#pragma unroll\n" << I << "  for (int mm_u = 0; mm_u < MM_UNROLL; ++mm_u) {\n"
                << I << "    // a row past the end was evaluated as the last row: storing it there again writes the\n"
                << I << "    // same bytes, and keeps the pixel bodies free of a store guard the compiler would\n"
                << I << "    // otherwise sink them (and their loads) into\n"
                << I << "    const int rl_raw = row0 + (mm_p + mm_u) * MM_TILE_H;\n"
                << I << (fetched ? "    mm_store_fetched_pixel(A, rl_raw < A.num_rows ? rl_raw : A.num_rows - 1, col, mm_rs[mm_u]);\n"
                                 : "    mm_store_pixel(A, rl_raw < A.num_rows ? rl_raw : A.num_rows - 1, col, mm_rt[mm_u]);\n")
                << I << "  }\n"
                << "#pragma unroll\n" << I << "  for (int mm_u = 0; mm_u < MM_UNROLL; ++mm_u) mm_y[mm_u] = mm_yn[mm_u];\n"
                << I << "}\n";
        };
        // Hot variant: when every fetch through a preloaded descriptor reads a bound drawable
        // at a valid, frame-constant frame number (true for every ordinary render), those
        // conditions -- all wave-uniform -- are tested once here instead of inside each fetch,
        // which leaves the unrolled pixel bodies free of branches.  The generic loop follows it and
        // takes over from the same mm_p: for everything when the hot conditions do not hold, and for a
        // work-item that met an invalid coordinate (mm_bad), whose garbage the reference converts in a
        // way only the generic fetch imitates.
        std::vector<std::string> hot_conds;
        find_hot_fetches(code.body, hot_conds);
        out << "  int mm_p = 0;\n";
        if (ks.unroll > 1 && !hot_conds.empty()) {
            out << "  bool mm_hot = true;\n";
            for (const std::string &c : hot_conds) out << "  mm_hot = mm_hot && " << c << ";\n";
            out << "  if (mm_hot) {\n";
            hot_mode = true;
            fetched_result = find_fetched_result();
            emit_loop("    ", true);
            fetched_result = nullptr;
            hot_mode = false;
            out << "  }\n";
        } else {
            hot_sites.clear();
        }
        emit_loop("  ", false);
        out << "}\n";
        finish_source();
    }

    // filter_$name of every filter that is called at run time.  The reference's are ordinary recursive C functions;
    // a GPU kernel wants a stack bound it can prove, so each is a template on the call depth: depth D calls depth
    // D + 1, and depth MM_MAX_CALL_DEPTH returns the zero tuple without evaluating anything -- the call graph is
    // a finite DAG, the compiler computes the exact stack need, nothing can overflow.  (A recursion deeper than
    // that is cut off; the reference would keep going until its C stack overflows.)
    void emit_functions() {
        FilterCode &root = fn_root ? *fn_root : code;
        if (root.functions.empty()) return;
        out << "#ifndef MM_MAX_CALL_DEPTH\n#define MM_MAX_CALL_DEPTH " << (getenv("MMHIP_MAX_CALL_DEPTH") ? atoi(getenv("MMHIP_MAX_CALL_DEPTH")) : 16)
            << "\n#endif\n"
               "struct mm_uvarg { int i; float f; color_t c; mm_image img; };\n";
        for (size_t k = 0; k < root.functions.size(); ++k)
            out << "template <int MM_D> __device__ __noinline__ mm_tup<4> mm_filter_" << k
                << "(const mm_args &A, const mm_uvarg *UV, float x, float y, float t, int col, int rl, unsigned &mm_rand_ctr);\n";
        for (size_t k = 0; k < root.functions.size(); ++k) {
            FilterCode &fn = *root.functions[k];
            Generator g(fn, opt);
            g.fn_root = &root;
            g.in_function = true;
            std::vector<Value *> defs;
            std::set<Value *> uses;
            g.collect_values(fn.body, PIXEL, defs, uses);
            out << "// filter_" << (fn.filter ? fn.filter->name : "") << "\n"
                << "template <int MM_D> __device__ __noinline__ mm_tup<4> mm_filter_" << k
                << "(const mm_args &A, const mm_uvarg *UV, float x, float y, float t, int col, int rl, unsigned &mm_rand_ctr) {\n"
                   "  mm_tup<4> rt;\n  rt.v[0] = rt.v[1] = rt.v[2] = rt.v[3] = 0.0f;\n"
                   "  if constexpr (MM_D >= MM_MAX_CALL_DEPTH) { return rt; } else {\n"
                   "  const float R = A.R; const int frame = 0;      // new_template.c.in:379: filter_$name has `int frame = 0`\n"
                   "  const int __canvasPixelW = A.img_width, __canvasPixelH = A.img_height;\n"
                   "  const int __renderPixelW = A.render_width, __renderPixelH = A.render_height;\n"
                   "  (void)R; (void)frame; (void)__canvasPixelW; (void)__canvasPixelH; (void)__renderPixelW; (void)__renderPixelH;\n";
            g.decls(defs, "  ");
            g.stmts(fn.body, PIXEL, "  ");
            out << g.out.str();
            for (int i = 0; i < 4; ++i) out << "  rt.v[" << i << "] = " << g.prim(Primary::V(fn.result[i]), PIXEL) << ";\n";
            out << "  return rt;\n  }\n}\n";
        }
    }

    void finish_source() {
        ks.source = out.str();
        char buf[32];
        snprintf(buf, sizeof buf, "%016llx", fnv(ks.source));
        ks.key = buf;
    }
};

}  // namespace

KernelSource generate_hip(FilterCode &code, const KernelOptions &opt, FilterCode *functions_of) {
    Generator g(code, opt);
    g.fn_root = functions_of;
    g.run();
    return std::move(g.ks);
}

}  // namespace mm
