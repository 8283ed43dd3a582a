// C ABI + GPU runtime: hiprtc JIT, HBM-resident images and user values, launches.
//
// One mmhip_invocation corresponds to the reference's mathmap_invocation_t
// (mathmap.h:162-202): canvas/render size, user values, input drawables, edge
// behaviour.  Rendering a row band = one prologue launch (frame constants, the
// reference's init_frame, new_template.c.in:314-337) + native-filter kernels if
// the filter calls any + one pixel-kernel launch (calc_lines, :208-312).
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/mmhip.h"
#include "front.h"
#include "hipgen.h"
#include "mm_host_abi.h"
#include "native_filters.h"
#include "passes.h"
#include "runtime_internal.h"

using namespace mm;

thread_local std::string g_mmhip_err;
#define g_err g_mmhip_err

namespace {

int fail(const std::string &msg) {
    g_err = msg;
    return -1;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

std::string cache_dir() {
    const char *env = getenv("MMHIP_CACHE_DIR");
    std::string d = env ? env : "/tmp/mmhip-cache-" + std::to_string((int)getuid());
    mkdir(d.c_str(), 0700);
    return d;
}

}  // namespace

extern "C" {

const char *mmhip_last_error(void) { return g_err.c_str(); }
const char *mmhip_version(void) { return "mathmap_amd 0.1 (gfx950)"; }

void mmhip_default_options(mmhip_options *o) {
    memset(o, 0, sizeof *o);
    o->intersample = 1;
    o->tile_w = 0;
}

// kernels of the closure images this filter hands to native filters (FilterCode::closure_renders)
static void generate_closure_kernels(mmhip_filter *f, const KernelOptions &ko) {
    f->closures.clear();
    for (auto &sub : f->code->closure_renders) {
        mmhip_closure_kernel ck;
        ck.ks = generate_hip(*sub, ko, f->code.get());     // may call native filters itself: render_closure runs them first
        f->closures.push_back(std::move(ck));
    }
}

static mmhip_filter *compile_source(const char *source, const mmhip_options *opts, const std::map<int, Primary> *consts) {
    std::unique_ptr<mmhip_filter> f(new mmhip_filter());
    try {
        parse_module(f->module, source);
        f->source = source;
        if (opts) f->opts = *opts;
        f->code = lower_filter(f->module, f->module.main, consts);
        f->ir_json_raw = dump_ir(*f->code);
        if (consts) specialize_constants(*f->code);
        optimize(*f->code);
        analyze_frame_constants(*f->code);
        for (auto &sub : f->code->closure_renders) {
            if (consts) specialize_constants(*sub);
            optimize(*sub);
            eliminate_dead_cycles(*sub);
            analyze_frame_constants(*sub);
        }
        for (auto &fn : f->code->functions) optimize(*fn);     // function bodies: no frame-constant slice, no user-value literals
        KernelOptions ko;
        if (opts) {
            ko.intersample = opts->intersample;
            ko.supersampling = opts->supersampling;
            ko.edge_x = opts->edge_behaviour_x;
            ko.edge_y = opts->edge_behaviour_y;
            ko.pixel_inc = opts->pixel_inc > 1 ? opts->pixel_inc : 1;
            if (opts->tile_w) ko.tile_w = opts->tile_w;
            f->opts = *opts;
            f->specialize = opts->specialize_uservals != 0;
        }
        f->source = source;
        f->kopt = ko;
        f->ir_json = dump_ir(*f->code);
        f->ks = generate_hip(*f->code, ko);
        generate_closure_kernels(f.get(), ko);
    } catch (const CompileError &e) {
        g_err = e.what();
        if (e.pos >= 0) g_err += " (at offset " + std::to_string(e.pos) + ")";
        return nullptr;
    } catch (const std::exception &e) {
        g_err = e.what();
        return nullptr;
    }
    return f.release();
}

mmhip_filter *mmhip_compile(const char *source, const mmhip_options *opts) { return compile_source(source, opts, nullptr); }

// Compiles with the given scalar user values (index, value) baked in as literals -- the same
// variant active_filter() builds lazily; exposed so the specialised kernel can be inspected
// and tested without a GPU.  Values of int/bool user values are truncated to int.
mmhip_filter *mmhip_compile_specialized(const char *source, const mmhip_options *opts, int n, const int *indices,
                                        const double *values) {
    Module probe;
    try {
        parse_module(probe, source);
    } catch (const std::exception &e) {
        g_err = e.what();
        return nullptr;
    }
    std::map<int, Primary> consts;
    for (int i = 0; i < n; ++i) {
        if (indices[i] < 0 || indices[i] >= (int)probe.main->uservals.size()) { g_err = "user value index out of range"; return nullptr; }
        const UservalInfo &u = probe.main->uservals[indices[i]];
        if (u.kind == UvKind::Float) consts[u.index] = Primary::F((float)values[i]);
        else if (u.kind == UvKind::Int || u.kind == UvKind::Bool) consts[u.index] = Primary::I((int)values[i]);
    }
    return compile_source(source, opts, &consts);
}

}  // extern "C"

mmhip_filter *mmhip_filter_new_empty() { return new mmhip_filter(); }

// `f->module.main` and `f->code` have been filled in by the caller (the ABI importer).
bool mmhip_filter_finalize(mmhip_filter *f, const KernelOptions &ko, std::string *err) {
    try {
        f->code->filter = f->module.main;
        if (f->ir_json_raw.empty()) f->ir_json_raw = dump_ir(*f->code);
        optimize(*f->code);
        analyze_frame_constants(*f->code);
        for (auto &sub : f->code->closure_renders) {
            sub->filter = f->module.main;
            optimize(*sub);
            eliminate_dead_cycles(*sub);
            analyze_frame_constants(*sub);
        }
        for (auto &fn : f->code->functions) optimize(*fn);
        f->kopt = ko;
        f->ir_json = dump_ir(*f->code);
        f->ks = generate_hip(*f->code, ko);
        generate_closure_kernels(f, ko);
    } catch (const std::exception &e) {
        *err = e.what();
        return false;
    }
    return true;
}

extern "C" {

void mmhip_filter_free(mmhip_filter *f) {
    if (!f) return;
    for (auto &p : f->spec_cache) mmhip_filter_free(p.second);
    for (mmhip_closure_kernel &ck : f->closures) if (ck.mod) (void)hipModuleUnload(ck.mod);
    if (f->mod) (void)hipModuleUnload(f->mod);
    delete f;
}

const char *mmhip_filter_name(const mmhip_filter *f) { return f->module.main->name.c_str(); }
int mmhip_filter_num_uservals(const mmhip_filter *f) { return (int)f->module.main->uservals.size(); }

int mmhip_filter_userval_info(const mmhip_filter *f, int index, mmhip_userval_info *out) {
    const auto &uvs = f->module.main->uservals;
    if (index < 0 || index >= (int)uvs.size()) return fail("user value index out of range");
    const UservalInfo &u = uvs[index];
    memset(out, 0, sizeof *out);
    out->kind = (int)u.kind;
    out->index = u.index;
    snprintf(out->name, sizeof out->name, "%s", u.name.c_str());
    out->int_min = u.imin; out->int_max = u.imax; out->int_default = u.idef;
    out->float_min = u.fmin; out->float_max = u.fmax; out->float_default = u.fdef;
    out->bool_default = u.bdef;
    out->image_flags = u.image_flags;
    return 0;
}

const char *mmhip_filter_ir_json(mmhip_filter *f) { return f->ir_json.c_str(); }
const char *mmhip_filter_ir_json_raw(mmhip_filter *f) { return f->ir_json_raw.c_str(); }
const char *mmhip_filter_kernel_source(mmhip_filter *f) { return f->ks.source.c_str(); }
int mmhip_filter_num_native_calls(const mmhip_filter *f) { return f->ks.native_sites; }
double mmhip_filter_jit_seconds(const mmhip_filter *f) { return f->jit_seconds; }

// hiprtc-compiles one kernel source (or fetches it from the on-disk cache); 0 on success
static int jit_source(const KernelSource &ks, std::vector<char> &code_object) {
    if (!code_object.empty()) return 0;
    // extra hiprtc options for experiments (space separated); part of the cache key
    std::vector<std::string> extra;
    std::string extra_key;
    if (const char *e = getenv("MMHIP_HIPRTC_FLAGS")) {
        std::istringstream is(e);
        for (std::string w; is >> w;) { extra.push_back(w); extra_key += "_" + std::to_string(std::hash<std::string>()(w) & 0xffff); }
    }
    std::string path = cache_dir() + "/" + ks.key + "_o2" + extra_key + ".hsaco";   // _o2: option-set version
    std::ifstream in(path, std::ios::binary);
    const char *ov = getenv("MMHIP_SOURCE_OVERRIDE");
    const bool may_override = ov && !strncmp(ov, ks.key.c_str(), ks.key.size()) && ov[ks.key.size()] == ':';
    if (in && !getenv("MMHIP_NO_CACHE") && !may_override) code_object.assign(std::istreambuf_iterator<char>(in), std::istreambuf_iterator<char>());
    if (!code_object.empty()) return 0;
    // kernel experiments: MMHIP_SOURCE_OVERRIDE=<key>:<file> compiles the file's text in place of the generated kernel
    // with that key (hand-edited variants of one kernel under the unchanged launch code; never cached)
    std::string source = ks.source;
    bool overridden = false;
    if (const char *e = getenv("MMHIP_SOURCE_OVERRIDE")) {
        const std::string spec = e;
        const size_t c = spec.find(':');
        if (c != std::string::npos && spec.substr(0, c) == ks.key) {
            std::ifstream f(spec.substr(c + 1));
            if (!f) return fail("MMHIP_SOURCE_OVERRIDE: cannot read " + spec.substr(c + 1));
            source.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
            overridden = true;
        }
    }
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, source.c_str(), "mathmap_filter.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS)
        return fail("hiprtcCreateProgram failed");
    // -fno-slp-vectorize: the SLP vectoriser packs scalar f32 chains into v_pk_* at the price of
    // register shuffles (Mandelbrot's loop: 11 VALU with it, 10 without; measured +9 %); the
    // fetch path gets its packed math from explicit float2 code instead
    std::vector<const char *> opts = {"--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-fno-slp-vectorize"};
    for (const std::string &w : extra) opts.push_back(w.c_str());
    hiprtcResult r = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
    if (r != HIPRTC_SUCCESS) {
        size_t n = 0;
        hiprtcGetProgramLogSize(prog, &n);
        std::string log(n, 0);
        if (n) hiprtcGetProgramLog(prog, &log[0]);
        hiprtcDestroyProgram(&prog);
        return fail("hiprtc compile failed:\n" + log);
    }
    size_t n = 0;
    hiprtcGetCodeSize(prog, &n);
    code_object.resize(n);
    hiprtcGetCode(prog, code_object.data());
    hiprtcDestroyProgram(&prog);
    if (overridden) return 0;
    std::string tmp = path + ".tmp" + std::to_string((int)getpid());
    std::ofstream o(tmp, std::ios::binary);
    if (o) {
        o.write(code_object.data(), (std::streamsize)code_object.size());
        o.close();
        rename(tmp.c_str(), path.c_str());
    }
    return 0;
}

static int load_kernels(const KernelSource &ks, const std::vector<char> &code_object, hipModule_t *mod, hipFunction_t *f_pix,
                        hipFunction_t *f_pro) {
    hipError_t e = hipModuleLoadData(mod, code_object.data());
    if (e != hipSuccess) return fail(std::string("hipModuleLoadData: ") + hipGetErrorString(e));
    e = hipModuleGetFunction(f_pix, *mod, ks.pixel_name.c_str());
    if (e != hipSuccess) return fail(std::string("hipModuleGetFunction(pixels): ") + hipGetErrorString(e));
    e = hipModuleGetFunction(f_pro, *mod, ks.prologue_name.c_str());
    if (e != hipSuccess) return fail(std::string("hipModuleGetFunction(prologue): ") + hipGetErrorString(e));
    return 0;
}

long mmhip_filter_jit(mmhip_filter *f, int load_module) {
    auto t0 = std::chrono::steady_clock::now();
    if (jit_source(f->ks, f->code_object) != 0) return -1;
    for (mmhip_closure_kernel &ck : f->closures)
        if (jit_source(ck.ks, ck.code_object) != 0) return -1;
    if (load_module && !f->loaded) {
        if (load_kernels(f->ks, f->code_object, &f->mod, &f->f_pix, &f->f_pro) != 0) return -1;
        if (f->ks.row_values > 0) {
            hipError_t e = hipModuleGetFunction(&f->f_rows, f->mod, f->ks.rows_name.c_str());
            if (e != hipSuccess) return fail(std::string("hipModuleGetFunction(rows): ") + hipGetErrorString(e));
        }
        for (mmhip_closure_kernel &ck : f->closures)
            if (!ck.loaded) {
                if (load_kernels(ck.ks, ck.code_object, &ck.mod, &ck.f_pix, &ck.f_pro) != 0) return -1;
                ck.loaded = true;
            }
        f->loaded = true;
    }
    f->jit_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return (long)f->code_object.size();
}

// ---------------------------------------------------------------------------
// invocation
// ---------------------------------------------------------------------------
mmhip_invocation *mmhip_invoke(mmhip_filter *f, int img_width, int img_height) {
    if (img_width <= 0 || img_height <= 0) { fail("image size must be positive"); return nullptr; }
    if (mmhip_filter_jit(f, 1) < 0) return nullptr;
    std::unique_ptr<mmhip_invocation> inv(new mmhip_invocation());
    inv->f = f;
    inv->img_w = inv->render_w = img_width;
    inv->img_h = inv->render_h = img_height;
    const auto &uvs = f->module.main->uservals;
    inv->uv.resize(std::max<size_t>(uvs.size(), 1));
    inv->image_slot_of_uv.assign(uvs.size(), -1);
    for (const UservalInfo &u : uvs) {   // defaults: userval.c:361-409
        HUserval &v = inv->uv[u.index];
        v.i = 0;
        switch (u.kind) {
            case UvKind::Int: v.i = u.idef; break;
            case UvKind::Float: v.f = u.fdef; break;
            case UvKind::Bool: v.i = u.bdef ? 1 : 0; break;
            case UvKind::Color: v.c = 0x000000ffu; break;   // opaque black
            case UvKind::Curve: {   // default curve: identity ramp (userval.c:282-311)
                v.i = (int)(inv->curves.size() / 1024);
                for (int i = 0; i < 1024; ++i) inv->curves.push_back((float)i / (float)(1024 - 1));
                break;
            }
            case UvKind::Gradient: {   // default gradient: opaque grey ramp (mathmap.c:356-361)
                v.i = (int)(inv->gradients.size() / 1024);
                for (int i = 0; i < 1024; ++i) {
                    float g = (float)i / (float)(1024 - 1);
                    uint32_t q = (uint32_t)(int)(g * 255.0) & 0xff;
                    inv->gradients.push_back((q << 24) | (q << 16) | (q << 8) | 255u);
                }
                break;
            }
            case UvKind::Image: {
                int slot = (int)inv->images.size();
                inv->image_slot_of_uv[u.index] = slot;
                HImageDesc d{};
                d.kind = IMG_NULL;
                inv->images.push_back(d);
                v.image = slot;
                break;
            }
        }
    }
    inv->native_slot_base = (int)inv->images.size();
    for (size_t k = 0; k < f->ks.natives.size(); ++k) {
        HImageDesc d{};
        d.kind = IMG_NULL;
        inv->images.push_back(d);
    }
    inv->closure_state.resize(f->closures.size());
    for (size_t c = 0; c < f->closures.size(); ++c) {      // the closure kernels' own native-filter results: slots of their own
        auto &st = inv->closure_state[c];
        st.native_slot_base = (int)inv->images.size();
        st.native_maps.assign(f->closures[c].ks.natives.size(), nullptr);
        for (size_t k = 0; k < f->closures[c].ks.natives.size(); ++k) {
            HImageDesc d{};
            d.kind = IMG_NULL;
            inv->images.push_back(d);
        }
    }
    inv->native_maps.assign(f->ks.natives.size(), nullptr);
    inv->native_map_size.assign(f->ks.natives.size(), {0, 0});
    inv->native_gen.assign(f->ks.natives.size(), 0);
    inv->native_memo_deps.assign(f->ks.natives.size(), {});
    inv->native_memo.resize(f->ks.natives.size());
    inv->native_memo_gen.assign(f->ks.natives.size(), 0);
    inv->native_seen.resize(f->ks.natives.size());
    inv->native_seen_gen.assign(f->ks.natives.size(), ~0ULL);
    inv->native_rows.assign(f->ks.natives.size(), {0, 0});
    if (inv->images.empty()) { HImageDesc d{}; d.kind = IMG_NULL; inv->images.push_back(d); }
    auto bail = [&](const char *what, hipError_t e) -> mmhip_invocation * {
        fail(std::string(what) + ": " + hipGetErrorString(e));
        return nullptr;
    };
    hipError_t e;
    if ((e = hipStreamCreate(&inv->stream)) != hipSuccess) return bail("hipStreamCreate", e);
    if ((e = hipMalloc((void **)&inv->d_uv, inv->uv.size() * sizeof(HUserval))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMalloc((void **)&inv->d_images, inv->images.size() * sizeof(HImageDesc))) != hipSuccess) return bail("hipMalloc", e);
    const int xy_bytes = std::max(f->ks.xy_bytes, 256);
    if ((e = hipMalloc((void **)&inv->d_xy, xy_bytes)) != hipSuccess) return bail("hipMalloc", e);
    if (!inv->curves.empty() && (e = hipMalloc((void **)&inv->d_curves, inv->curves.size() * 4)) != hipSuccess) return bail("hipMalloc", e);
    if (!inv->gradients.empty() && (e = hipMalloc((void **)&inv->d_gradients, inv->gradients.size() * 4)) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMemset(inv->d_xy, 0, xy_bytes)) != hipSuccess) return bail("hipMemset", e);
    inv->xy_cap = xy_bytes;
    return inv.release();
}

void mmhip_invocation_free(mmhip_invocation *inv) {
    if (!inv) return;
    if (inv->stream) (void)hipStreamSynchronize(inv->stream);
    for (void *p : inv->owned) (void)hipFree(p);
    for (void *p : inv->native_maps) if (p) (void)hipFree(p);
    if (inv->ss_lines) (void)hipFree(inv->ss_lines);
    for (auto &c : inv->closure_state) {
        if (c.map) (void)hipFree(c.map);
        if (c.d_xy) (void)hipFree(c.d_xy);
        if (c.d_xtab) (void)hipFree(c.d_xtab);
        if (c.d_ytab) (void)hipFree(c.d_ytab);
        for (void *p : c.native_maps) if (p) (void)hipFree(p);
    }
    inv->ws.release();
    if (inv->d_uv) (void)hipFree(inv->d_uv);
    if (inv->d_images) (void)hipFree(inv->d_images);
    if (inv->d_xy) (void)hipFree(inv->d_xy);
    if (inv->d_xtab) (void)hipFree(inv->d_xtab);
    if (inv->d_ytab) (void)hipFree(inv->d_ytab);
    if (inv->d_rowtab) (void)hipFree(inv->d_rowtab);
    if (inv->d_curves) (void)hipFree(inv->d_curves);
    if (inv->d_gradients) (void)hipFree(inv->d_gradients);
    for (auto &p : inv->ev_pool) {
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
    }
    if (inv->stream) (void)hipStreamDestroy(inv->stream);
    delete inv;
}

static const UservalInfo *uv_info(mmhip_invocation *inv, int index, UvKind kind) {
    const auto &uvs = inv->f->module.main->uservals;
    if (index < 0 || index >= (int)uvs.size()) { fail("user value index out of range"); return nullptr; }
    if (uvs[index].kind != kind) { fail("user value `" + uvs[index].name + "' has a different type"); return nullptr; }
    return &uvs[index];
}

int mmhip_set_int(mmhip_invocation *inv, int index, int value) {
    const UservalInfo *u = uv_info(inv, index, UvKind::Int);
    if (!u) return -1;
    inv->uv[index].i = value;
    inv->tables_dirty = true;
    return 0;
}

int mmhip_set_float(mmhip_invocation *inv, int index, float value) {
    const UservalInfo *u = uv_info(inv, index, UvKind::Float);
    if (!u) return -1;
    inv->uv[index].f = value;
    inv->tables_dirty = true;
    return 0;
}

int mmhip_set_bool(mmhip_invocation *inv, int index, int value) {
    const UservalInfo *u = uv_info(inv, index, UvKind::Bool);
    if (!u) return -1;
    inv->uv[index].i = value ? 1 : 0;
    inv->tables_dirty = true;
    return 0;
}

int mmhip_set_color(mmhip_invocation *inv, int index, float r, float g, float b, float a) {
    const UservalInfo *u = uv_info(inv, index, UvKind::Color);
    if (!u) return -1;
    auto q = [](float v) { v = v < 0 ? 0 : v > 1 ? 1 : v; return (uint32_t)(v * 255.0); };
    inv->uv[index].c = (q(r) << 24) | (q(g) << 16) | (q(b) << 8) | q(a);
    inv->tables_dirty = true;
    return 0;
}

// -Dname=value (mathmap_cmdline.c:756-796; images are bound by the caller)
int mmhip_set_by_name(mmhip_invocation *inv, const char *name, const char *value) {
    for (const UservalInfo &u : inv->f->module.main->uservals) {
        if (u.name != name) continue;
        switch (u.kind) {
            case UvKind::Int: return mmhip_set_int(inv, u.index, atoi(value));
            case UvKind::Float: return mmhip_set_float(inv, u.index, (float)atof(value));
            case UvKind::Bool: return mmhip_set_bool(inv, u.index, atoi(value));
            default: return fail(std::string("user value `") + name + "' cannot be set from a string");
        }
    }
    return fail(std::string("filter has no user value `") + name + "'");
}

static void fill_drawable_desc(HImageDesc &d, const void *data, int w, int h) {
    d.data = data;
    d.w = w;
    d.h = h;
    d.kind = IMG_DRAWABLE;
    d.num_frames = 1;
    d.scale_x = (float)((w - 1) / 2.0);    // userval.c:272-276
    d.scale_y = (float)((h - 1) / 2.0);
    d.middle_x = 1.0f;
    d.middle_y = 1.0f;
    d.ax = d.bx = d.ay = d.by = 0.f;
}

int mmhip_set_image_device(mmhip_invocation *inv, int index, const void *device_rgba32, int width, int height) {
    const UservalInfo *u = uv_info(inv, index, UvKind::Image);
    if (!u) return -1;
    int slot = inv->image_slot_of_uv[index];
    fill_drawable_desc(inv->images[slot], device_rgba32, width, height);
    inv->tables_dirty = true;
    ++inv->input_generation;
    return 0;
}

int mmhip_set_image_host(mmhip_invocation *inv, int index, const uint8_t *pixels, int width, int height, int channels) {
    if (channels != 3 && channels != 4) return fail("channels must be 3 or 4");
    if (!uv_info(inv, index, UvKind::Image)) return -1;
    size_t n = (size_t)width * height;
    std::vector<uint32_t> packed(n);
    for (size_t i = 0; i < n; ++i) {
        const uint8_t *p = pixels + i * channels;
        uint32_t a = channels == 4 ? p[3] : 255u;
        packed[i] = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | a;
    }
    void *d = nullptr;
    HIP_TRY(hipMalloc(&d, n * 4));
    HIP_TRY(hipMemcpy(d, packed.data(), n * 4, hipMemcpyHostToDevice));
    // the upload this one replaces (if it was ours) is freed once nothing in flight reads it
    const void *old = inv->images[inv->image_slot_of_uv[index]].data;
    for (size_t i = 0; i < inv->owned.size(); ++i)
        if (inv->owned[i] == old) {
            (void)hipDeviceSynchronize();     // renders may have been queued on caller streams
            (void)hipFree(inv->owned[i]);
            inv->owned.erase(inv->owned.begin() + i);
            break;
        }
    inv->owned.push_back(d);
    return mmhip_set_image_device(inv, index, d, width, height);
}

int mmhip_set_curve(mmhip_invocation *inv, int index, const float *values1024) {
    if (!uv_info(inv, index, UvKind::Curve)) return -1;
    float *dst = &inv->curves[(size_t)inv->uv[index].i * 1024];
    if (memcmp(dst, values1024, 1024 * sizeof(float)) == 0) return 0;     // unchanged: no re-upload
    memcpy(dst, values1024, 1024 * sizeof(float));
    inv->tables_dirty = true;
    return 0;
}

int mmhip_set_gradient(mmhip_invocation *inv, int index, const uint32_t *rgba1024) {
    if (!uv_info(inv, index, UvKind::Gradient)) return -1;
    uint32_t *dst = &inv->gradients[(size_t)inv->uv[index].i * 1024];
    if (memcmp(dst, rgba1024, 1024 * sizeof(uint32_t)) == 0) return 0;
    memcpy(dst, rgba1024, 1024 * sizeof(uint32_t));
    inv->tables_dirty = true;
    return 0;
}

// Row-striped rendering of filters with native-filter calls (one stripe per GPU): with a margin
// >= 0 a render of rows [a, b) of the full frame lets the native filters fill only rows
// [a - margin, b + margin) of their maps (plus whatever halo the filter itself needs).  The caller
// asserts that the filter samples a native map no further than `margin` rows from the output row
// (0 for `blurred(xy)`).  -1 (default): always the whole map, like the reference.
int mmhip_set_native_row_margin(mmhip_invocation *inv, int margin) {
    inv->native_row_margin = margin;
    return 0;
}

int mmhip_set_edge_colors(mmhip_invocation *inv, uint32_t cx, uint32_t cy) {
    inv->edge_color_x = cx;
    inv->edge_color_y = cy;
    return 0;
}

int mmhip_set_render_size(mmhip_invocation *inv, int rw, int rh) {
    inv->render_w = rw;
    inv->render_h = rh;
    return 0;
}

int mmhip_set_sampling_offset(mmhip_invocation *inv, float ox, float oy) {
    inv->sampling_offset_x = ox;
    inv->sampling_offset_y = oy;
    return 0;
}

int mmhip_enable_timing(mmhip_invocation *inv, int on) {
    inv->timing = on != 0;
    inv->ws.timing = on != 0;
    return 0;
}

// Durations (ms) of the native filters' own kernels (gaussian_blur: its four scan kernels) launched since the last
// drain, in launch order; names[i * 64 ...] receives the kernel's label.  Waits for the last of them.
int mmhip_drain_native_kernel_ms(mmhip_invocation *inv, char *names, double *out_ms, int cap) {
    auto &tm = inv->ws.timed;
    if (!tm.empty() && hipEventSynchronize(tm.back().b) != hipSuccess) return fail("event sync failed");
    int n = 0;
    for (auto &t : tm) {
        float ms = 0;
        if (n < cap && hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) {
            snprintf(names + (size_t)n * 64, 64, "%s", t.name);
            out_ms[n++] = ms;
        }
        inv->ws.timed_free.push_back({t.a, t.b});
    }
    tm.clear();
    return n;
}

// The next event pair for a timed launch (grown on demand, wraps after 4096 pending launches).
static int next_event_pair(mmhip_invocation *inv) {
    if (inv->ev_used == inv->ev_pool.size()) {
        if (inv->ev_pool.size() >= 4096) inv->ev_used = 0;
        else {
            hipEvent_t a, b;
            HIP_TRY(hipEventCreate(&a));
            HIP_TRY(hipEventCreate(&b));
            inv->ev_pool.push_back({a, b});
        }
    }
    inv->ev0 = inv->ev_pool[inv->ev_used].first;
    inv->ev1 = inv->ev_pool[inv->ev_used].second;
    ++inv->ev_used;
    return 0;
}

// Durations (ms) of the pixel kernel of every timed launch since the last drain, oldest first;
// waits for the last of them.  Returns how many were written (at most `cap`).
long mmhip_direct_native_launches(mmhip_invocation *inv) { return inv->direct_native_launches; }

int mmhip_drain_kernel_ms(mmhip_invocation *inv, double *out_ms, int cap) {
    int n = 0;
    if (inv->ev_used > 0 && hipEventSynchronize(inv->ev_pool[inv->ev_used - 1].second) != hipSuccess) return fail("event sync failed");
    for (size_t i = 0; i < inv->ev_used && n < cap; ++i) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, inv->ev_pool[i].first, inv->ev_pool[i].second) != hipSuccess) return fail("event elapsed failed");
        out_ms[n++] = ms;
    }
    inv->ev_used = 0;
    inv->ev_valid = false;
    return n;
}

double mmhip_last_kernel_ms(mmhip_invocation *inv) {
    if (!inv->ev_valid) return -1.0;
    if (hipEventSynchronize(inv->ev1) != hipSuccess) return -1.0;
    float ms = 0;
    if (hipEventElapsedTime(&ms, inv->ev0, inv->ev1) != hipSuccess) return -1.0;
    return ms;
}

static int upload_tables(mmhip_invocation *inv, hipStream_t s) {
    if (!inv->tables_dirty) return 0;
    // make sure nothing in flight still reads the old tables
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipMemcpy(inv->d_uv, inv->uv.data(), inv->uv.size() * sizeof(HUserval), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(inv->d_images, inv->images.data(), inv->images.size() * sizeof(HImageDesc), hipMemcpyHostToDevice));
    if (inv->d_curves) HIP_TRY(hipMemcpy(inv->d_curves, inv->curves.data(), inv->curves.size() * 4, hipMemcpyHostToDevice));
    if (inv->d_gradients) HIP_TRY(hipMemcpy(inv->d_gradients, inv->gradients.data(), inv->gradients.size() * 4, hipMemcpyHostToDevice));
    inv->tables_dirty = false;
    ++inv->table_generation;
    return 0;
}

static int rows_per_item(const KernelSource &ks, int tiles_x, int num_rows) {
    // rows per work-item: enough workgroups must remain to fill 256 CUs several times over
    const long wg1 = (long)tiles_x * ((num_rows + ks.tile_h - 1) / ks.tile_h);
    int ppt = wg1 >= 262144 ? 16 : wg1 >= 131072 ? 8 : wg1 >= 32768 ? 4 : wg1 >= 8192 ? 2 : 1;
    if (const char *e = getenv("MMHIP_PPT")) ppt = std::max(1, atoi(e));
    if (ks.single_pixel) ppt = 1;
    const int u = std::max(1, ks.unroll);          // the kernel steps MM_UNROLL rows at a time
    return (ppt + u - 1) / u * u;
}

// render_image's closure branch (builtins.c:273-298): closure image #cid of the filter rendered over the
// whole frame into a float map -- calc_lines(slice 0,0,w,h; first_row 0, last_row h; floatmap = 1) on a
// frame made by invocation_new_frame(invocation, image, 0, 0.0): frame 0, t = 0, sampling offsets 0.
// The native-filter calls the prologue recorded in `host' (a copy of the frame-constant buffer), in the order they
// were made: entry k of ks.natives (a call site outside loops, or the n-th dynamic entry of the in-loop sites) and its
// record.  The record's `pad' is the call's number within the frame (mm_native_call in hipgen.cpp).
struct RecordedCall { size_t k; HNativeRec rec; const std::string *func; };
static int recorded_calls(const KernelSource &ks, const std::vector<char> &host, std::vector<RecordedCall> *calls) {
    calls->clear();
    if (ks.natives.empty()) return 0;
    int ctr[4];
    memcpy(ctr, host.data() + ks.native_ctr_offset, sizeof ctr);
    if (ctr[2])
        return fail("native filters are called more than " + std::to_string((int)MM_NATIVE_DYN_CALLS) +
                    " times from inside a loop of the frame-constant code: not supported");
    for (size_t k = 0; k < ks.natives.size(); ++k) {
        RecordedCall c;
        c.k = k;
        memcpy(&c.rec, host.data() + ks.natives[k].record_offset, sizeof c.rec);
        if (!c.rec.executed) continue;
        if (c.rec.index < 0 || c.rec.index >= ks.native_sites) return fail("internal: native call record names no call site");
        c.func = &ks.natives[c.rec.index].func;
        calls->push_back(c);
    }
    std::sort(calls->begin(), calls->end(), [](const RecordedCall &x, const RecordedCall &y) { return x.rec.pad < y.rec.pad; });
    return 0;
}

// Is the call `c' recorded by a closure's render kernel the call `m' the main code has already made this frame?  The render
// kernel evaluates the main filter's code once more (it computes the closure's arguments), native calls included: those are
// the same calls on the same images -- in the reference the closure's argument simply *is* the image the main code computed,
// and its cache would answer (native-filters/cache.c:110-147).  Scalars by their bits, images by what they refer to: an input
// image or closure by its handle, a native result by the main call its own producer was matched with (`alias': render
// kernel's entry -> main's image-table slot, -1 unmatched).
static bool same_native_call(const RecordedCall &c, const RecordedCall &m, const std::vector<int> &alias, int closure_slot_base) {
    if (*c.func != *m.func || c.rec.nargs != m.rec.nargs) return false;
    for (int i = 0; i < c.rec.nargs && i < 4; ++i) {
        const HNativeArg &x = c.rec.args[i], &y = m.rec.args[i];
        if (x.kind != y.kind) return false;
        if (x.kind != 2) {
            if (x.i != y.i || memcmp(&x.f, &y.f, sizeof x.f) != 0) return false;
            continue;
        }
        int idx = x.img.idx;
        const int rel = idx - closure_slot_base;
        if (rel >= 0 && rel < (int)alias.size()) {
            if (alias[rel] < 0) return false;
            idx = alias[rel];
        }
        if (idx != y.img.idx || x.img.pw != y.img.pw || x.img.ph != y.img.ph || x.img.resized != y.img.resized ||
            memcmp(&x.img.xf, &y.img.xf, sizeof x.img.xf) != 0 || memcmp(&x.img.yf, &y.img.yf, sizeof x.img.yf) != 0)
            return false;
    }
    return true;
}

// `main_done': the main code's calls run so far this frame (their maps are valid)
static int render_closure(mmhip_invocation *inv, mmhip_filter *f, int cid, const HArgs &main_args, hipStream_t s,
                          const std::vector<RecordedCall> &main_done) {
    mmhip_closure_kernel &ck = f->closures[cid];
    auto &st = inv->closure_state[cid];
    const int w = main_args.render_width, h = main_args.render_height;
    if (st.map && (st.w != w || st.h != h)) {
        HIP_TRY(hipDeviceSynchronize());
        (void)hipFree(st.map);
        st.map = nullptr;
        for (void *&p : st.native_maps) { if (p) (void)hipFree(p); p = nullptr; }
    }
    if (!st.map) { HIP_TRY(hipMalloc(&st.map, (size_t)w * h * 16)); st.w = w; st.h = h; }
    if (w > st.xtab_cap) {
        if (st.d_xtab) { HIP_TRY(hipStreamSynchronize(s)); (void)hipFree(st.d_xtab); }
        HIP_TRY(hipMalloc((void **)&st.d_xtab, (size_t)w * sizeof(float)));
        st.xtab_cap = w;
    }
    if (h > st.ytab_cap) {
        if (st.d_ytab) { HIP_TRY(hipStreamSynchronize(s)); (void)hipFree(st.d_ytab); }
        HIP_TRY(hipMalloc((void **)&st.d_ytab, (size_t)h * sizeof(float)));
        st.ytab_cap = h;
    }
    const int xy_bytes = std::max(ck.ks.xy_bytes, 256);
    if (xy_bytes > st.xy_cap) {
        if (st.d_xy) { HIP_TRY(hipStreamSynchronize(s)); (void)hipFree(st.d_xy); }
        HIP_TRY(hipMalloc((void **)&st.d_xy, xy_bytes));
        HIP_TRY(hipMemset(st.d_xy, 0, xy_bytes));
        st.xy_cap = xy_bytes;
    }
    // (t and frame stay the frame's own: the closure's *arguments* are values of the main filter's code at the current
    // time; the closure's body is lowered with t = 0.0 and frame = 0 as literals, lower.cpp native_image_argument)
    HArgs a = main_args;
    a.region_x = a.region_y = 0;
    a.region_width = w;
    a.region_height = h;
    a.sampling_offset_x = a.sampling_offset_y = 0.0f;
    a.first_row = 0;
    a.num_rows = h;
    a.output_bpp = 4;
    a.row_stride = w * 4;
    a.floatmap = 1;
    a.out = st.map;
    a.xtab = st.d_xtab;
    a.ytab = st.d_ytab;
    const int tiles_x = (w + ck.ks.tile_w - 1) / ck.ks.tile_w;
    a.ppt = rows_per_item(ck.ks, tiles_x, h);
    const int tiles_y = (h + ck.ks.tile_h * a.ppt - 1) / (ck.ks.tile_h * a.ppt);
    a.tiles_magic = tile_division_magic(tiles_x, (long)tiles_x * tiles_y);
    char *xy = st.d_xy;
    void *params[] = {&a, &xy};
    const int n = std::max(w, h);
    a.native_slot_base = st.native_slot_base;
    HIP_TRY(hipModuleLaunchKernel(ck.f_pro, (unsigned)((n + 255) / 256), 1, 1, 256, 1, 1, 0, s, params, nullptr));
    if (!ck.ks.natives.empty()) {
        // The closure's own calc_lines starts with its init_frame, and that is where *its* native filters run
        // (builtins.c:273-298 -> new_template.c.in:314-337): the records its prologue just wrote, each executed call into a
        // map of the closure's own.  Like the closure image itself these are recomputed on every render (the reference
        // gives the closure a fresh id: nothing of it is ever found in the cache).
        std::vector<char> host(ck.ks.xy_bytes);
        HIP_TRY(hipMemcpyAsync(host.data(), st.d_xy, host.size(), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        std::vector<RecordedCall> calls;
        if (recorded_calls(ck.ks, host, &calls) != 0) return -1;
        std::vector<int> alias(ck.ks.natives.size(), -1);
        for (const RecordedCall &call : calls) {
            const size_t k = call.k;
            const HNativeRec &rec = call.rec;
            // a call of the main code's copy: the main code has its result already (same_native_call)
            bool aliased = false;
            static const bool no_alias = getenv("MMHIP_NO_NATIVE_ALIAS") != nullptr;      // (experiments: run every recorded call)
            for (const RecordedCall &m : main_done) {
                if (no_alias) break;
                const int mslot = inv->native_slot_base + (int)m.k;
                if (!same_native_call(call, m, alias, st.native_slot_base) || !inv->native_maps[m.k] || inv->images[mslot].kind != IMG_FLOATMAP ||
                    inv->images[mslot].w != w || inv->images[mslot].h != h || inv->native_rows[m.k].first > 0 || inv->native_rows[m.k].second < h)
                    continue;
                inv->images[st.native_slot_base + (int)k] = inv->images[mslot];      // the same map under the render kernel's handle
                alias[k] = mslot;
                aliased = true;
                break;
            }
            if (aliased) continue;
            for (int i = 0; i < rec.nargs && i < 4; ++i)
                if (rec.args[i].kind == 2 && rec.args[i].img.idx <= -2)
                    return fail("a filter closure rendered for a native filter hands another closure to a native filter: not supported");
            if (!st.native_maps[k]) HIP_TRY(hipMalloc(&st.native_maps[k], (size_t)w * h * 16));
            inv->ws.env.supersampling = f->kopt.supersampling;
            inv->ws.env.edge_x = f->kopt.edge_x;
            inv->ws.env.edge_y = f->kopt.edge_y;
            inv->ws.env.edge_color_x = inv->edge_color_x;
            inv->ws.env.edge_color_y = inv->edge_color_y;
            std::string err;
            int lo = 0, hi = h;
            if (run_native_filter(*call.func, rec, inv->images, w, h, (float *)st.native_maps[k], inv->ws, s, &err, &lo, &hi) != 0)
                return fail(err);
            HImageDesc &d = inv->images[st.native_slot_base + (int)k];
            d.data = st.native_maps[k];
            d.w = w;
            d.h = h;
            d.kind = IMG_FLOATMAP;
            d.num_frames = 1;
            d.ax = d.bx = (float)((float)(d.w - 1) / 2.0);     // floatmap.c:39-41
            d.ay = d.by = (float)((float)(d.h - 1) / 2.0);
            d.ay *= -1.0f;
        }
        HIP_TRY(hipStreamSynchronize(s));
        HIP_TRY(hipMemcpy(inv->d_images, inv->images.data(), inv->images.size() * sizeof(HImageDesc), hipMemcpyHostToDevice));
    }
    HIP_TRY(hipModuleLaunchKernel(ck.f_pix, (unsigned)((long)tiles_x * tiles_y), 1, 1, 256, 1, 1, 0, s, params, nullptr));
    return 0;
}

static int run_natives(mmhip_invocation *inv, mmhip_filter *f, const HArgs &a, hipStream_t s, bool *direct_written) {
    std::vector<char> host(f->ks.xy_bytes);
    HIP_TRY(hipMemcpyAsync(host.data(), inv->d_xy, host.size(), hipMemcpyDeviceToHost, s));
    // Direct output (hipgen.cpp find_direct_native): the pixel is native result k sampled at (x, y).
    // If every sample position of this launch is the pixel's own centre -- get_floatmap_pixel's
    // lrintf(ax x + bx) (builtins.c:247-265) evaluated here for each column and row with the
    // coordinates the prologue just computed -- the native filter may write the RGBA8 pixels itself.
    NativeDirectOut direct;
    const bool try_direct = f->ks.direct_native >= 0 && !a.floatmap && a.output_bpp == 4 && (a.row_stride & 3) == 0 &&
                            ((uintptr_t)a.out & 3) == 0 && !getenv("MMHIP_NO_DIRECT_NATIVE");
    std::vector<float> xt, yt;
    if (try_direct) {
        xt.resize(a.region_width);
        yt.resize(a.num_rows);
        HIP_TRY(hipMemcpyAsync(xt.data(), a.xtab, xt.size() * sizeof(float), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(yt.data(), a.ytab, yt.size() * sizeof(float), hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    if (try_direct) {
        const float ax = (float)((float)(a.render_width - 1) / 2.0), bx = ax;       // floatmap.c:39-41
        const float by = (float)((float)(a.render_height - 1) / 2.0), ay = by * -1.0f;
        bool identity = true;
        for (int c = 0; c < a.region_width && identity; ++c) identity = lrintf(ax * xt[c] + bx) == (long)a.region_x + c;
        for (int r = 0; r < a.num_rows && identity; ++r) identity = lrintf(ay * yt[r] + by) == (long)a.first_row + r;
        if (identity) {
            direct.out = a.out;
            direct.row_stride = a.row_stride;
            direct.first_row = a.first_row;
            direct.num_rows = a.num_rows;
            direct.region_x = a.region_x;
            direct.region_w = a.region_width;
        }
    }
    bool table_changed = false;
    std::vector<RecordedCall> calls, done;      // done: calls whose maps stand (what a closure's render kernel may refer to)
    if (recorded_calls(f->ks, host, &calls) != 0) return -1;
    for (const RecordedCall &call : calls) {
        const size_t k = call.k;
        HNativeRec rec = call.rec;
        int slot = inv->native_slot_base + (int)k;
        // memo (native-filters/cache.c:110-147): same arguments on unchanged inputs -> keep the map
        // rows of the map this launch may read: everything, or -- opt-in, full-frame regions only --
        // the stripe being rendered (the filter samples the map within its own rows +- margin)
        int want_lo = 0, want_hi = a.render_height;
        if (inv->native_row_margin >= 0 && a.region_x == 0 && a.region_y == 0 && a.region_width == a.render_width &&
            a.region_height == a.render_height) {
            want_lo = std::max(0, a.first_row - inv->native_row_margin);
            want_hi = std::min(a.render_height, a.first_row + a.num_rows + inv->native_row_margin);
        }
        // A map belongs to the render size it was allocated for: the GIMP flow renders a small preview
        // and then the full image on one invocation (mathmap.c:2191-2223).  On a change the map is
        // reallocated and everything remembered about it dropped.
        if (inv->native_maps[k] && (inv->native_map_size[k].first != a.render_width || inv->native_map_size[k].second != a.render_height)) {
            HIP_TRY(hipDeviceSynchronize());
            (void)hipFree(inv->native_maps[k]);
            inv->native_maps[k] = nullptr;
            inv->native_memo_gen[k] = ~0ULL;
            inv->native_seen_gen[k] = ~0ULL;
            inv->native_rows[k] = {0, 0};
            inv->images[slot].kind = IMG_NULL;
            inv->images[slot].data = nullptr;
            table_changed = true;
        }
        // Closure images among the arguments (index -2 - id, mm_closure_image): rendered into float maps now and
        // handed to the native filter as such.  The reference gives every closure image a fresh id
        // (image_new_id), so a call on a closure never hits the cache: always recomputed here too.
        std::vector<HImageDesc> images_k;                 // inv->images + the rendered closures, only when needed
        bool has_closure_arg = false;
        for (int i = 0; i < rec.nargs && i < 4; ++i) {
            if (rec.args[i].kind != 2 || rec.args[i].img.idx > -2) continue;
            const int cid = -2 - rec.args[i].img.idx;
            if (cid >= (int)f->closures.size()) return fail("internal: closure image without a render kernel");
            if (render_closure(inv, f, cid, a, s, done) != 0) return -1;
            if (images_k.empty()) images_k = inv->images;
            HImageDesc d{};
            d.data = inv->closure_state[cid].map;
            d.w = a.render_width;
            d.h = a.render_height;
            d.kind = IMG_FLOATMAP;
            d.num_frames = 1;
            d.ax = d.bx = (float)((float)(d.w - 1) / 2.0);
            d.ay = d.by = (float)((float)(d.h - 1) / 2.0);
            d.ay *= -1.0f;
            rec.args[i].img.idx = (int)images_k.size();
            rec.args[i].img.pw = d.w;
            rec.args[i].img.ph = d.h;
            rec.args[i].img.xf = rec.args[i].img.yf = 1.0f;      // a plain map: the closure's render kernel has applied the
            rec.args[i].img.resized = 0;                          // wrapper's factors to its coordinates already (lower.cpp)
            images_k.push_back(d);
            has_closure_arg = true;
        }
        // generations of the native maps among this call's image arguments (cache.c keys on image ids)
        std::vector<unsigned long long> deps;
        for (int i = 0; i < rec.nargs && i < 4; ++i)
            if (rec.args[i].kind == 2 && rec.args[i].img.idx >= inv->native_slot_base &&
                rec.args[i].img.idx < inv->native_slot_base + (int)inv->native_gen.size())
                deps.push_back(inv->native_gen[rec.args[i].img.idx - inv->native_slot_base]);
        if (!has_closure_arg && inv->native_maps[k] && inv->native_memo_gen[k] == inv->input_generation &&
            memcmp(&inv->native_memo[k], &rec, sizeof rec) == 0 && inv->native_memo_deps[k] == deps &&
            inv->native_rows[k].first <= want_lo && inv->native_rows[k].second >= want_hi) {
            done.push_back(call);
            continue;
        }
        size_t bytes = (size_t)a.render_width * a.render_height * 16;
        if (!inv->native_maps[k]) {
            HIP_TRY(hipMalloc(&inv->native_maps[k], bytes));
            inv->native_map_size[k] = {a.render_width, a.render_height};
        }
        std::string err;
        int got_lo = want_lo, got_hi = want_hi;
        NativeDirectOut *dk = (direct.out && (int)k == f->ks.direct_native) ? &direct : nullptr;
        // A launch that covers the whole frame needs the map for nothing but the memo.  The first time
        // an argument set is seen it is therefore not written (16 of the second pass's 20 B/px); a
        // second request for the same set -- an animation that keeps the blur's arguments -- computes
        // it once more, with the map, and is memoised from then on.
        const bool whole = a.region_x == 0 && a.region_y == 0 && a.region_width == a.render_width &&
                           a.region_height == a.render_height && a.first_row == 0 && a.num_rows == a.render_height;
        if (dk) {
            dk->skip_map = whole && !(inv->native_seen_gen[k] == inv->input_generation && inv->native_memo_deps[k] == deps &&
                                      memcmp(&inv->native_seen[k], &rec, sizeof rec) == 0);
            inv->native_seen[k] = rec;
            inv->native_seen_gen[k] = inv->input_generation;
        }
        inv->ws.env.supersampling = f->kopt.supersampling;
        inv->ws.env.edge_x = f->kopt.edge_x;
        inv->ws.env.edge_y = f->kopt.edge_y;
        inv->ws.env.edge_color_x = inv->edge_color_x;
        inv->ws.env.edge_color_y = inv->edge_color_y;
        int rc = run_native_filter(*call.func, rec, has_closure_arg ? images_k : inv->images, a.render_width, a.render_height,
                                   (float *)inv->native_maps[k], inv->ws, s, &err, &got_lo, &got_hi, dk);
        if (rc != 0) return fail(err);
        inv->native_gen[k] = ++inv->native_gen_counter;
        inv->native_memo_deps[k] = deps;
        if (dk && dk->written) *direct_written = true;
        if (dk && dk->written && dk->skip_map) {       // nothing to memoise, no map to describe
            inv->native_memo_gen[k] = ~0ULL;
            inv->native_rows[k] = {0, 0};
            continue;
        }
        HImageDesc &d = inv->images[slot];
        d.data = inv->native_maps[k];
        d.w = a.render_width;
        d.h = a.render_height;
        d.kind = IMG_FLOATMAP;
        d.num_frames = 1;
        d.ax = d.bx = (float)((float)(d.w - 1) / 2.0);     // floatmap.c:39-41
        d.ay = d.by = (float)((float)(d.h - 1) / 2.0);
        d.ay *= -1.0f;
        inv->native_memo[k] = rec;
        inv->native_memo_gen[k] = inv->input_generation;
        inv->native_rows[k] = {got_lo, got_hi};
        table_changed = true;
        done.push_back(call);
    }
    // dynamic entries this frame did not use (the loop ran fewer times than before): their maps go back (a map is
    // 16 B per pixel of the frame; sixteen of them at 16384^2 are 69 GB)
    if (f->ks.native_sites < (int)f->ks.natives.size()) {
        std::vector<char> used(f->ks.natives.size(), 0);
        for (const RecordedCall &call : calls) used[call.k] = 1;
        for (size_t k = (size_t)f->ks.native_sites; k < f->ks.natives.size(); ++k) {
            if (used[k] || !inv->native_maps[k]) continue;
            HIP_TRY(hipStreamSynchronize(s));
            (void)hipFree(inv->native_maps[k]);
            inv->native_maps[k] = nullptr;
            inv->native_memo_gen[k] = ~0ULL;
            inv->native_seen_gen[k] = ~0ULL;
            inv->native_rows[k] = {0, 0};
            HImageDesc &d = inv->images[inv->native_slot_base + (int)k];
            d.kind = IMG_NULL;
            d.data = nullptr;
            table_changed = true;
        }
    }
    if (table_changed) {
        HIP_TRY(hipStreamSynchronize(s));
        HIP_TRY(hipMemcpy(inv->d_images, inv->images.data(), inv->images.size() * sizeof(HImageDesc), hipMemcpyHostToDevice));
    }
    return 0;
}

// Specialisation of a filter that has no source text (IR imported through the reference-ABI tier or
// mmhip_compile_ir_json): reload its own IR dump, replace the scalar USERVAL_*_ACCESS reads by the
// literals, and run the same constant propagation / folding as the source-level variant.
static void bake_uservals(Block &b, const std::map<int, Primary> &consts) {
    for (Stmt *st : b) {
        if (st->kind == Stmt::Assign && st->rhs.kind == Rhs::Op && st->rhs.args.size() == 1 &&
            st->rhs.args[0].kind == Primary::IntConst) {
            const char *n = st->rhs.op->cname;
            if (!strcmp(n, "USERVAL_INT_ACCESS") || !strcmp(n, "USERVAL_FLOAT_ACCESS") || !strcmp(n, "USERVAL_BOOL_ACCESS")) {
                auto it = consts.find(st->rhs.args[0].i);
                if (it != consts.end()) st->rhs = Rhs::P(it->second);
            }
        }
        if (st->kind == Stmt::If) { bake_uservals(st->then_, consts); bake_uservals(st->else_, consts); }
        if (st->kind == Stmt::While) bake_uservals(st->body, consts);
    }
}

static mmhip_filter *compile_ir_specialized(const mmhip_filter *f, const std::map<int, Primary> &consts) {
    mmhip_filter *sp = mmhip_filter_new_empty();
    try {
        sp->code.reset(new FilterCode());
        load_ir_json(sp->module, *sp->code, (f->ir_json_raw.empty() ? f->ir_json : f->ir_json_raw).c_str());
        bake_uservals(sp->code->body, consts);
        specialize_constants(*sp->code);
        for (auto &sub : sp->code->closure_renders) {
            bake_uservals(sub->body, consts);
            specialize_constants(*sub);
        }
        std::string err;
        if (!mmhip_filter_finalize(sp, f->kopt, &err)) throw CompileError(err);
    } catch (const std::exception &e) {
        g_err = e.what();
        mmhip_filter_free(sp);
        return nullptr;
    }
    return sp;
}

// The variant of a compiled filter with n scalar user values (index, value) baked in as literals: what
// active_filter() builds lazily for a value set, for filters of either origin (source text or IR dump).
extern "C" mmhip_filter *mmhip_filter_specialized(const mmhip_filter *f, int n, const int *indices, const double *values) {
    std::map<int, Primary> consts;
    const auto &uvs = f->module.main->uservals;
    for (int i = 0; i < n; ++i) {
        if (indices[i] < 0 || indices[i] >= (int)uvs.size()) { g_err = "user value index out of range"; return nullptr; }
        const UservalInfo &u = uvs[indices[i]];
        if (u.kind == UvKind::Float) consts[u.index] = Primary::F((float)values[i]);
        else if (u.kind == UvKind::Int || u.kind == UvKind::Bool) consts[u.index] = Primary::I((int)values[i]);
    }
    mmhip_options o = f->opts;
    o.specialize_uservals = 0;
    return f->source.empty() ? compile_ir_specialized(f, consts) : compile_source(f->source.c_str(), &o, &consts);
}

// The kernel set to launch: the generic filter, or -- with options.specialize_uservals -- a
// variant with the current scalar user values baked in as literals (built on first use per
// value set, cached on the filter and on disk through the hiprtc cache).
static mmhip_filter *active_filter(mmhip_invocation *inv, int frame = 0, float t = 0.0f) {
    mmhip_filter *f = inv->f;
    if (!f->specialize || !f->ks.natives.empty() || (f->source.empty() && f->ir_json.empty() && f->ir_json_raw.empty())) return f;
    g_err.clear();
    const auto &uvs = f->module.main->uservals;
    std::string key;
    std::map<int, Primary> consts;
    for (const UservalInfo &u : uvs) {
        if (u.kind == UvKind::Int || u.kind == UvKind::Bool) consts[u.index] = Primary::I(inv->uv[u.index].i);
        else if (u.kind == UvKind::Float) consts[u.index] = Primary::F(inv->uv[u.index].f);
        else continue;
        key.append((const char *)&inv->uv[u.index], sizeof(HUserval));
    }
    if (consts.empty()) return f;
    auto it = f->spec_cache.find(key);
    if (it != f->spec_cache.end()) return it->second ? it->second : f;
    // a host that changes values on every render (interactive sliders) should not pay a JIT each
    // time: build the variant on the spec_min_uses-th render with the same values
    // A "use" is a render of a new frame with these values: the bands of one frame count once, so a
    // host that animates a user value (new values every frame, several calc_lines bands each) never
    // triggers a JIT per frame; an animation over t with fixed values specialises at its 2nd frame.
    if (f->spec_min_uses > 1) {
        if (f->spec_uses.size() > 1024) f->spec_uses.clear();
        auto &u = f->spec_uses[key];
        if (u.count == 0 || u.frame != frame || u.t != t) {
            ++u.count;
            u.frame = frame;
            u.t = t;
        }
        if (u.count < f->spec_min_uses) return f;
    }
    f->spec_uses.erase(key);
    mmhip_options o = f->opts;
    o.specialize_uservals = 0;
    mmhip_filter *sp = f->source.empty() ? compile_ir_specialized(f, consts) : compile_source(f->source.c_str(), &o, &consts);
    if (sp && mmhip_filter_jit(sp, 1) < 0) { mmhip_filter_free(sp); sp = nullptr; }
    f->spec_cache[key] = sp;          // nullptr = fall back to the generic kernel for this value set
    return sp ? sp : f;
}

int mmhip_render(mmhip_invocation *inv, int frame, float t, int region_x, int region_y, int region_w, int region_h,
                 int first_row, int last_row, void *out_device, int row_stride, int bpp, int floatmap, void *stream) {
    mmhip_filter *f = active_filter(inv, frame, t);
    hipStream_t s = stream ? (hipStream_t)stream : inv->stream;
    if (bpp < 1 || bpp > 4) return fail("output_bpp must be 1..4");
    if (region_w <= 0 || region_h <= 0) return fail("empty region");
    // new_template.c.in:238-239
    if (first_row < 0) first_row = 0;
    if (last_row > region_y + region_h) last_row = region_y + region_h;
    if (last_row <= first_row) return 0;
    if (upload_tables(inv, s) != 0) return -1;

    HArgs a{};
    a.img_width = inv->img_w;
    a.img_height = inv->img_h;
    a.render_width = inv->render_w;
    a.render_height = inv->render_h;
    a.frame_render_width = inv->render_w;     // invocation_new_frame, mathmap_common.c:805-806
    a.frame_render_height = inv->render_h;
    a.t = t;
    a.frame = frame;
    a.R = (float)sqrt(2.0);                   // mathmap_common.c:770
    a.region_x = region_x;
    a.region_y = region_y;
    a.region_width = region_w;
    a.region_height = region_h;
    a.sampling_offset_x = inv->sampling_offset_x;
    a.sampling_offset_y = inv->sampling_offset_y;
    a.first_row = first_row;
    a.num_rows = last_row - first_row;
    a.output_bpp = bpp;
    a.row_stride = row_stride;
    a.floatmap = floatmap;
    a.edge_color_x = inv->edge_color_x;
    a.edge_color_y = inv->edge_color_y;
    a.uservals = inv->d_uv;
    a.images = inv->d_images;
    a.num_images = (uint32_t)inv->images.size();
    a.curves = inv->d_curves;
    a.gradients = inv->d_gradients;
    a.out = out_device;
    a.native_slot_base = inv->native_slot_base;

    // coordinate tables (grown on demand; stream order protects re-use)
    if (region_w > inv->xtab_cap) {
        if (inv->d_xtab) { HIP_TRY(hipStreamSynchronize(s)); (void)hipFree(inv->d_xtab); }
        HIP_TRY(hipMalloc((void **)&inv->d_xtab, (size_t)region_w * sizeof(float)));
        inv->xtab_cap = region_w;
    }
    if (a.num_rows > inv->ytab_cap) {
        if (inv->d_ytab) { HIP_TRY(hipStreamSynchronize(s)); (void)hipFree(inv->d_ytab); }
        HIP_TRY(hipMalloc((void **)&inv->d_ytab, (size_t)a.num_rows * sizeof(float)));
        inv->ytab_cap = a.num_rows;
    }
    a.xtab = inv->d_xtab;
    a.ytab = inv->d_ytab;
    if (f->ks.row_values > 0) {
        const size_t need = (size_t)f->ks.row_values * (size_t)a.num_rows;
        if (need > inv->rowtab_cap) {
            if (inv->d_rowtab) { HIP_TRY(hipStreamSynchronize(s)); (void)hipFree(inv->d_rowtab); }
            inv->d_rowtab = nullptr;
            inv->rowtab_cap = 0;
            HIP_TRY(hipMalloc((void **)&inv->d_rowtab, need * sizeof(float)));
            inv->rowtab_cap = need;
            inv->pro_filter = nullptr;           // the table is new: fill it
        }
        a.rowtab = inv->d_rowtab;
    }

    if (f->ks.xy_bytes > inv->xy_cap) {
        if (inv->d_xy) { HIP_TRY(hipStreamSynchronize(s)); (void)hipFree(inv->d_xy); }
        HIP_TRY(hipMalloc((void **)&inv->d_xy, f->ks.xy_bytes));
        HIP_TRY(hipMemset(inv->d_xy, 0, f->ks.xy_bytes));
        inv->xy_cap = f->ks.xy_bytes;
    }
    char *xy = inv->d_xy;
    void *params[] = {&a, &xy};
    bool direct_written = false;      // a native filter wrote this launch's pixels itself (run_natives)
    {
        // The prologue's outputs (frame constants, coordinate tables) depend on the filter, the
        // geometry, the user values / image table and -- only if its code reads them -- t and
        // frame: while none of those changed since it last ran on these buffers, skip the launch
        // (an animation of a filter whose constants do not involve t pays for it once).
        HArgs key = a;
        key.out = nullptr;
        key.row_stride = key.output_bpp = key.floatmap = key.ppt = 0;
        key.tiles_magic = 0;
        if (!f->ks.prologue_uses_time) { key.t = 0.0f; key.frame = 0; }
        const bool fresh = f->ks.natives.empty() && inv->pro_filter == f && inv->pro_stream == (void *)s &&
                           inv->pro_generation == inv->table_generation &&
                           memcmp(&key, &inv->pro_args, sizeof key) == 0;
        if (!fresh) {
            int n = std::max(region_w, a.num_rows);
            HIP_TRY(hipModuleLaunchKernel(f->f_pro, (unsigned)((n + 255) / 256), 1, 1, 256, 1, 1, 0, s, params, nullptr));
            // the per-row slice (x-const code): once per row of the launch, after the frame constants it reads
            if (f->ks.row_values > 0)
                HIP_TRY(hipModuleLaunchKernel(f->f_rows, (unsigned)((a.num_rows + 255) / 256), 1, 1, 256, 1, 1, 0, s, params, nullptr));
            if (!f->ks.natives.empty() && run_natives(inv, f, a, s, &direct_written) != 0) return -1;
            inv->pro_args = key;
            inv->pro_filter = f;
            inv->pro_stream = (void *)s;
            inv->pro_generation = inv->table_generation;
        }
    }
    int tiles_x = (region_w + f->ks.tile_w - 1) / f->ks.tile_w;
    a.ppt = rows_per_item(f->ks, tiles_x, a.num_rows);
    int tiles_y = (a.num_rows + f->ks.tile_h * a.ppt - 1) / (f->ks.tile_h * a.ppt);
    long nwg = (long)tiles_x * tiles_y;
    if (nwg > 0x7fffffffL) return fail("region too large for one launch");
    a.tiles_magic = tile_division_magic(tiles_x, nwg);
    if (inv->timing) {
        if (next_event_pair(inv) != 0) return -1;
        HIP_TRY(hipEventRecord(inv->ev0, s));
    }
    if (direct_written) ++inv->direct_native_launches;
    if (!direct_written) HIP_TRY(hipModuleLaunchKernel(f->f_pix, (unsigned)nwg, 1, 1, 256, 1, 1, 0, s, params, nullptr));
    if (inv->timing) {
        HIP_TRY(hipEventRecord(inv->ev1, s));
        inv->ev_valid = true;
    }
    return 0;
}

// call_invocation with supersampling (mathmap_common.c:880-927), whole region on the GPU.
// The filter must have been compiled with options.supersampling = 1 (nearest fetch without
// the +0.5, builtins.c:154-158).
int mmhip_render_supersampled(mmhip_invocation *inv, int frame, float t, int region_x, int region_y, int region_w,
                              int region_h, void *out_device, int row_stride, int bpp, void *stream) {
    hipStream_t s = stream ? (hipStream_t)stream : inv->stream;
    const size_t long_bytes = (size_t)(region_w + 1) * bpp * region_h;
    const size_t short_bytes = (size_t)region_w * bpp * region_h;
    // own allocation: the nested renders run native filters, which reallocate inv->ws
    if (long_bytes + short_bytes > inv->ss_bytes) {
        if (inv->ss_lines) { HIP_TRY(hipDeviceSynchronize()); (void)hipFree(inv->ss_lines); }
        inv->ss_lines = nullptr;
        inv->ss_bytes = 0;
        if (hipMalloc(&inv->ss_lines, long_bytes + short_bytes) != hipSuccess)
            return fail("out of device memory for the supersampling lines");
        inv->ss_bytes = long_bytes + short_bytes;
    }
    unsigned char *tmp = (unsigned char *)inv->ss_lines;
    const float ox = inv->sampling_offset_x, oy = inv->sampling_offset_y;
    // long slice: region_width + 1 columns, offsets -0.5 (invocation_init_slice, :892)
    inv->sampling_offset_x = -0.5f;
    inv->sampling_offset_y = -0.5f;
    int rc = mmhip_render(inv, frame, t, region_x, region_y, region_w + 1, region_h, region_y, region_y + region_h, tmp,
                          (region_w + 1) * bpp, bpp, 0, s);
    inv->sampling_offset_x = 0.f;
    inv->sampling_offset_y = 0.f;
    if (rc == 0)
        rc = mmhip_render(inv, frame, t, region_x, region_y, region_w, region_h, region_y, region_y + region_h,
                          tmp + long_bytes, region_w * bpp, bpp, 0, s);
    inv->sampling_offset_x = ox;
    inv->sampling_offset_y = oy;
    if (rc != 0) return rc;
    launch_supersample_combine(tmp, tmp + long_bytes, (unsigned char *)out_device, region_w, region_h, bpp, row_stride, s);
    return 0;
}

int mmhip_sync(mmhip_invocation *inv) {
    HIP_TRY(hipStreamSynchronize(inv->stream));
    return 0;
}

int mmhip_render_host(mmhip_invocation *inv, int frame, float t, uint8_t *out_rgba) {
    size_t bytes = (size_t)inv->render_w * inv->render_h * 4;
    void *d = nullptr;
    HIP_TRY(hipMalloc(&d, bytes));
    int rc = mmhip_render(inv, frame, t, 0, 0, inv->render_w, inv->render_h, 0, inv->render_h, d, inv->render_w * 4, 4, 0,
                          nullptr);
    if (rc == 0) {
        hipError_t e = hipStreamSynchronize(inv->stream);
        if (e == hipSuccess) e = hipMemcpy(out_rgba, d, bytes, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(std::string("render: ") + hipGetErrorString(e));
    }
    (void)hipFree(d);
    return rc;
}

void *mmhip_device_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) { fail("hipMalloc failed"); return nullptr; }
    return p;
}
void mmhip_device_free(void *p) { (void)hipFree(p); }
int mmhip_copy_to_host(void *dst, const void *src, size_t bytes) {
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return 0;
}
int mmhip_copy_to_device(void *dst, const void *src, size_t bytes) {
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return 0;
}
int mmhip_set_device(int ordinal) {
    HIP_TRY(hipSetDevice(ordinal));
    return 0;
}
int mmhip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

}  // extern "C"
