// Internal definitions shared by runtime.cpp and abi_backend.cpp.
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../../include/mmhip.h"
#include "front.h"
#include "hipgen.h"
#include "mm_host_abi.h"
#include "native_filters.h"

namespace mm { void load_ir_json(Module &mod, FilterCode &code, const char *json); }   // ir_json.cpp

// The kernels that render closure image #k of a filter into a float map (FilterCode::closure_renders[k]):
// same user values and images as the owning filter, own code object.
struct mmhip_closure_kernel {
    mm::KernelSource ks;
    std::vector<char> code_object;
    hipModule_t mod = nullptr;
    hipFunction_t f_pro = nullptr, f_pix = nullptr;
    bool loaded = false;
};

struct mmhip_filter {
    std::vector<mmhip_closure_kernel> closures;
    mm::Module module;
    std::unique_ptr<mm::FilterCode> code;
    mm::KernelOptions kopt;
    mm::KernelSource ks;
    std::string ir_json;        // after the passes: what the kernels were generated from
    std::string ir_json_raw;    // straight out of lowering (or the importer), before any pass: what the
                                // oracle prints, so that the passes are tested differentially
    std::vector<char> code_object;
    hipModule_t mod = nullptr;
    hipFunction_t f_pro = nullptr, f_pix = nullptr;
    hipFunction_t f_rows = nullptr;           // the per-row slice's kernel (KernelSource::row_values > 0)
    bool loaded = false;
    double jit_seconds = 0;
    // user-value specialisation (specialize.cpp): kernels with the scalar user values baked in,
    // keyed by the value bytes; owned by this filter
    std::string source;
    mmhip_options opts{};
    bool specialize = false;
    std::map<std::string, mmhip_filter *> spec_cache;
    struct SpecUse { int count = 0; int frame = 0; float t = 0.0f; };
    std::map<std::string, SpecUse> spec_uses; // frames seen per value set that has no variant yet
    int spec_min_uses = 1;                    // build the variant on this many-th render with one value set
    // a filter that only compiles with its scalar user values baked in (recursion whose depth
    // they control): no generic code/kernels, every render goes through spec_cache
};

struct mmhip_invocation {
    mmhip_filter *f = nullptr;
    int img_w = 0, img_h = 0, render_w = 0, render_h = 0;
    std::vector<mm::HUserval> uv;
    std::vector<mm::HImageDesc> images;
    std::vector<int> image_slot_of_uv;     // userval index -> image table slot (or -1)
    std::vector<float> curves;             // [n_curves][1024] (userval.h:37, userval.c:282-311)
    std::vector<uint32_t> gradients;       // [n_gradients][1024] packed 0xRRGGBBAA
    float *d_curves = nullptr;
    uint32_t *d_gradients = nullptr;
    int native_slot_base = 0;
    mm::HUserval *d_uv = nullptr;
    mm::HImageDesc *d_images = nullptr;
    bool tables_dirty = true;
    std::vector<void *> owned;             // device buffers we allocated for input images
    std::vector<void *> native_maps;       // float4 maps produced by native filters
    std::vector<std::pair<int, int>> native_map_size;   // render size native_maps[k] was allocated for
    // Every recomputation of map k gets a new generation (the reference gives every native result a
    // new image id, cache.c:65-68); a memo entry records the generations of the native maps among its
    // image arguments, so a consumer is recomputed when its producer was.
    std::vector<unsigned long long> native_gen;
    std::vector<std::vector<unsigned long long>> native_memo_deps;
    unsigned long long native_gen_counter = 0;
    // closure images rendered for native filters: float map + the sub-launch's own constant buffer / tables
    struct ClosureState { void *map = nullptr; int w = 0, h = 0; char *d_xy = nullptr; int xy_cap = 0;
                          float *d_xtab = nullptr, *d_ytab = nullptr; int xtab_cap = 0, ytab_cap = 0;
                          int native_slot_base = 0; std::vector<void *> native_maps; };   // native filters the closure's own code calls
    std::vector<ClosureState> closure_state;
    void *ss_lines = nullptr;              // the two slices of a supersampled render (own allocation:
    size_t ss_bytes = 0;                   // native filters reallocate `ws` underneath a nested render)
    std::vector<mm::HNativeRec> native_memo;   // args of the call that produced native_maps[k]
    std::vector<unsigned long long> native_memo_gen;
    std::vector<mm::HNativeRec> native_seen;   // last argument set whose map was asked for (direct output: materialised on its second use)
    std::vector<unsigned long long> native_seen_gen;
    std::vector<std::pair<int, int>> native_rows;   // rows of native_maps[k] that are valid
    int native_row_margin = -1;                     // mmhip_set_native_row_margin
    long direct_native_launches = 0;                // mmhip_direct_native_launches
    // the prologue kernel is skipped while nothing it reads has changed (mmhip_render)
    mm::HArgs pro_args{};
    const mmhip_filter *pro_filter = nullptr;
    void *pro_stream = nullptr;
    unsigned long long pro_generation = 0, table_generation = 1;
    unsigned long long input_generation = 1;
    char *d_xy = nullptr;
    int xy_cap = 0;
    float *d_xtab = nullptr, *d_ytab = nullptr;   // per-column / per-row coordinates of the current launch
    int xtab_cap = 0, ytab_cap = 0;
    float *d_rowtab = nullptr;                    // per-row values of the current launch (mm_rows): [value][row]
    size_t rowtab_cap = 0;
    hipStream_t stream = nullptr;
    uint32_t edge_color_x = 0, edge_color_y = 0;
    float sampling_offset_x = 0.f, sampling_offset_y = 0.f;
    bool timing = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;   // the pair of the most recent launch (aliases into ev_pool)
    bool ev_valid = false;
    // one event pair per timed launch, so a caller can queue many launches and read all durations
    // afterwards without a synchronisation per launch (mmhip_drain_kernel_ms)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    size_t ev_used = 0;
    mm::NativeWorkspace ws;
};


extern thread_local std::string g_mmhip_err;   // message behind mmhip_last_error()

// Two-step construction from already lowered IR (used by the reference-ABI importer):
// create, fill f->module (filters, main) and f->code, then finalize (passes + codegen).
mmhip_filter *mmhip_filter_new_empty();
bool mmhip_filter_finalize(mmhip_filter *f, const mm::KernelOptions &ko, std::string *err);
