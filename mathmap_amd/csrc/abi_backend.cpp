// Reference-ABI tier: gen_and_load_hip_code / unload_hip_code and the mathfuncs_t
// shims (init_frame, init_slice, calc_lines) a MathMap build calls instead of the cc
// backend.  Reads the reference's structures through the layout mirrors of
// include/mathmap_abi.h.
//
// Flow (reference call sites): compile_mathmap (mathmap_common.c:551-556) calls
// gen_and_load_hip_code -> we import the main filter's IR (statement_t list) into our
// own IR, keep it in a ModuleInfo and return `hip_mathmapinit`.  init_invocation
// (mathmap_common.c:714-734) calls that and gets the three shims.  Per frame the host
// calls init_frame, init_slice (no-ops here: frame constants are evaluated on the GPU
// by the prologue kernel) and calc_lines, which renders the row band on the GPU and
// copies it into the host buffer `q`.
#include <dlfcn.h>

#include <algorithm>
#include <functional>

#include <cstring>
#include <map>
#include <mutex>

#include "../../include/mathmap_hip_backend.h"
#include "gen.h"
#include "passes.h"
#include "runtime_internal.h"

using namespace mm;

namespace {

mmabi_get_pixel_func_t g_get_pixel = nullptr;

void host_error(const std::string &msg) {
    // the reference reports backend errors through its global error_string (cc.c:653-693)
    if (char *es = (char *)dlsym(RTLD_DEFAULT, "error_string")) {
        strncpy(es, msg.c_str(), 1023);
        es[1023] = 0;
    }
}

struct Variant {
    mmhip_filter *flt = nullptr;
    std::map<mmabi_invocation_t *, mmhip_invocation *> invs;
};

struct DrawableCopy { int width = 0, height = 0; void *dev = nullptr; };

struct ModuleInfo {
    std::mutex mu;
    // the imported, un-optimised IR is re-imported per variant from a serialised form:
    // we keep one pristine filter object per option set instead
    std::map<int, Variant> variants;           // key = option bits
    mmabi_mathmap_t *mathmap = nullptr;
    // everything needed to rebuild a variant
    struct SavedFilter { Filter f; };
    std::function<mmhip_filter *(const KernelOptions &, std::string *)> build;
    std::map<mmabi_input_drawable_t *, DrawableCopy> drawables;
    std::map<const float *, void *> floatmaps;
    void *staging = nullptr;
    size_t staging_bytes = 0;
};

std::mutex g_registry_mu;
std::map<mmabi_input_drawable_t *, ModuleInfo *> g_drawable_owner;

// ---------------------------------------------------------------------------
// IR import (compiler-internals.h structures -> mm::FilterCode)
// ---------------------------------------------------------------------------
struct Importer {
    Module &mod;
    FilterCode &code;
    // shared by the importers of one module: the main filter's code and the bodies of the filters it calls
    struct Shared {
        std::map<mmabi_filter_t *, Filter *> filters;
        std::vector<mmabi_filter_t *> called;      // callees of RHS_FILTER statements, in first-seen order
    };
    Shared &shared;
    std::map<mmabi_filter_t *, Filter *> &filters;
    std::map<mmabi_compvar_t *, CompVar *> vars;
    std::map<mmabi_value_t *, Value *> vals;

    Importer(Module &m, FilterCode &c, Shared &sh) : mod(m), code(c), shared(sh), filters(sh.filters) {}

    static unsigned flags_of(mmabi_filter_t *f) {   // mathmap_common.c:59-72
        bool pixel = false, stretched = false;
        if (f->v.mathmap.decl)
            for (mmabi_option_t *o = f->v.mathmap.decl->v.filter.options; o; o = o->next) {
                if (!strcmp(o->name, "pixel")) pixel = true;
                if (!strcmp(o->name, "stretched")) stretched = true;
            }
        unsigned fl = 0;
        if (!pixel) { fl |= IMAGE_FLAG_UNIT; if (!stretched) fl |= IMAGE_FLAG_SQUARE; }
        return fl;
    }

    Filter *filter(mmabi_filter_t *rf) {
        auto it = filters.find(rf);
        if (it != filters.end()) return it->second;
        Filter *f = nullptr;
        if (rf->kind == MMABI_FILTER_NATIVE) {
            for (auto &own : mod.filters)
                if (own->kind == Filter::Native && own->native_func == rf->v.native.func_name) f = own.get();
            if (!f) throw CompileError(std::string("unknown native filter ") + rf->v.native.func_name);
        } else {
            mod.filters.emplace_back(new Filter());
            f = mod.filters.back().get();
            f->kind = Filter::MathMap;
            f->name = rf->name;
            f->flags = flags_of(rf);
            for (mmabi_userval_info_t *u = rf->userval_infos; u; u = u->next) {
                UservalInfo ui;
                ui.name = u->name;
                ui.index = u->index;
                switch (u->type) {
                    case MMABI_USERVAL_INT_CONST:
                        ui.kind = UvKind::Int; ui.imin = u->v.int_const.min; ui.imax = u->v.int_const.max;
                        ui.idef = u->v.int_const.default_value; break;
                    case MMABI_USERVAL_FLOAT_CONST:
                        ui.kind = UvKind::Float; ui.fmin = u->v.float_const.min; ui.fmax = u->v.float_const.max;
                        ui.fdef = u->v.float_const.default_value; break;
                    case MMABI_USERVAL_BOOL_CONST: ui.kind = UvKind::Bool; ui.bdef = u->v.bool_const.default_value != 0; break;
                    case MMABI_USERVAL_COLOR: ui.kind = UvKind::Color; break;
                    case MMABI_USERVAL_CURVE: ui.kind = UvKind::Curve; break;
                    case MMABI_USERVAL_GRADIENT: ui.kind = UvKind::Gradient; break;
                    case MMABI_USERVAL_IMAGE: ui.kind = UvKind::Image; ui.image_flags = u->v.image.flags; break;
                    default: throw CompileError("unknown user value type in filter " + f->name);
                }
                f->uservals.push_back(ui);
            }
            // the reference keeps the list in registration order with ascending indices
            std::sort(f->uservals.begin(), f->uservals.end(),
                      [](const UservalInfo &a, const UservalInfo &b) { return a.index < b.index; });
        }
        filters[rf] = f;
        return f;
    }

    CompVar *var(mmabi_compvar_t *rv) {
        auto it = vars.find(rv);
        if (it != vars.end()) return it->second;
        std::string name = rv->var ? rv->var->name : "";
        CompVar *v = code.new_var((Ty)rv->type, name, rv->n);
        vars[rv] = v;
        return v;
    }

    Value *value(mmabi_value_t *rv) {
        auto it = vals.find(rv);
        if (it != vals.end()) return it->second;
        CompVar *cv = var(rv->compvar);
        Value *v;
        if (rv->index < 0) v = cv->current;   // the uninitialised value every compvar starts with
        else {
            v = code.new_value(cv);
            v->index = rv->index;
        }
        vals[rv] = v;
        return v;
    }

    Primary primary(const mmabi_primary_t &p) {
        if (p.kind == MMABI_PRIMARY_VALUE) return Primary::V(value(p.v.value));
        Primary q;
        switch (p.const_type) {
            case MMABI_TYPE_INT: return Primary::I(p.v.constant.int_value);
            case MMABI_TYPE_FLOAT: return Primary::F(p.v.constant.float_value);
            case MMABI_TYPE_COMPLEX:
                q.kind = Primary::ComplexConst; q.f = p.v.constant.complex_value[0]; q.f2 = p.v.constant.complex_value[1];
                return q;
            case MMABI_TYPE_COLOR: q.kind = Primary::ColorConst; q.color = p.v.constant.color_value; return q;
            default: throw CompileError("constant of a non-scalar type in the IR");
        }
    }

    Rhs rhs(mmabi_rhs_t *r) {
        Rhs out;
        switch (r->kind) {
            case MMABI_RHS_PRIMARY: return Rhs::P(primary(r->v.primary));
            case MMABI_RHS_INTERNAL: return Rhs::Int(r->v.internal->name);
            case MMABI_RHS_OP: {
                const OpInfo *op = op_by_cname(r->v.op.op->name, r->v.op.op->num_args);
                if (!op) throw CompileError(std::string("unknown IR operator ") + r->v.op.op->name);
                std::vector<Primary> args;
                for (int i = 0; i < op->nargs; ++i) args.push_back(primary(r->v.op.args[i]));
                return Rhs::O(op, args);
            }
            case MMABI_RHS_CLOSURE: {
                out.kind = Rhs::Closure;
                out.filter = filter(r->v.closure.filter);
                for (int i = 0; i < r->v.closure.filter->num_uservals; ++i) out.args.push_back(primary(r->v.closure.args[i]));
                return out;
            }
            case MMABI_RHS_TUPLE:
                out.kind = Rhs::Tuple;
                for (int i = 0; i < r->v.tuple.length; ++i) out.args.push_back(primary(r->v.tuple.args[i]));
                return out;
            case MMABI_RHS_FILTER: {
                // a call the reference's inliner left alone (can_inline, compiler.c:4219-4237: recursion): the callee's
                // user values, then x, y, t (optimize_closure_application, compiler.c:3258-3275)
                mmabi_filter_t *callee = r->v.filter.filter;
                if (callee->kind != MMABI_FILTER_MATHMAP) throw CompileError("RHS_FILTER of a native filter");
                out.kind = Rhs::FilterCall;
                out.filter = filter(callee);
                for (int i = 0; i < callee->num_uservals + 3; ++i) out.args.push_back(primary(r->v.filter.args[i]));
                if (std::find(shared.called.begin(), shared.called.end(), callee) == shared.called.end()) shared.called.push_back(callee);
                return out;
            }
            case MMABI_RHS_TREE_VECTOR:      // make_tree_vector_rhs (compiler.c:553-563) shares RHS_TUPLE's fields
                out.kind = Rhs::TreeVector;
                for (int i = 0; i < r->v.tuple.length; ++i) out.args.push_back(primary(r->v.tuple.args[i]));
                return out;
            default: throw CompileError("unknown rhs kind in the IR");
        }
    }

    void phis(mmabi_statement_t *s, Block &out, Stmt *parent) {
        for (; s; s = s->next) {
            if (s->kind != MMABI_STMT_PHI_ASSIGN) continue;
            Stmt *p = code.new_stmt(Stmt::Phi);
            p->lhs = value(s->v.assign.lhs);
            p->lhs->def = p;
            p->rhs = rhs(s->v.assign.rhs);
            p->rhs2 = rhs(s->v.assign.rhs2);
            p->parent = parent;
            out.push_back(p);
        }
    }

    void block(mmabi_statement_t *s, Block &out, Stmt *parent) {
        for (; s; s = s->next) {
            switch (s->kind) {
                case MMABI_STMT_NIL: break;
                case MMABI_STMT_ASSIGN: {
                    Stmt *a = code.new_stmt(Stmt::Assign);
                    a->lhs = value(s->v.assign.lhs);
                    a->lhs->def = a;
                    a->rhs = rhs(s->v.assign.rhs);
                    a->parent = parent;
                    out.push_back(a);
                    break;
                }
                case MMABI_STMT_IF_COND: {
                    Stmt *i = code.new_stmt(Stmt::If);
                    i->cond = rhs(s->v.if_cond.condition);
                    i->parent = parent;
                    block(s->v.if_cond.consequent, i->then_, i);
                    block(s->v.if_cond.alternative, i->else_, i);
                    phis(s->v.if_cond.exit, i->phis, i);
                    out.push_back(i);
                    break;
                }
                case MMABI_STMT_WHILE_LOOP: {
                    Stmt *w = code.new_stmt(Stmt::While);
                    w->parent = parent;
                    phis(s->v.while_loop.entry, w->phis, w);
                    w->cond = rhs(s->v.while_loop.invariant);
                    block(s->v.while_loop.body, w->body, w);
                    out.push_back(w);
                    break;
                }
                default: throw CompileError("phi statement outside a phi list");
            }
        }
    }

    // compiler.c:4692-4697: the filter result is `dummy = OUTPUT_TUPLE(tuple)` where
    // `tuple` was assigned RHS_TUPLE(r,g,b,a)
    void find_result() {
        Stmt *outp = nullptr;
        for (Stmt *s : code.body)
            if (s->kind == Stmt::Assign && s->rhs.kind == Rhs::Op && !strcmp(s->rhs.op->cname, "OUTPUT_TUPLE")) outp = s;
        if (!outp || outp->rhs.args[0].kind != Primary::Val) throw CompileError("IR has no OUTPUT_TUPLE statement");
        Stmt *def = outp->rhs.args[0].value->def;
        if (!def || def->rhs.kind != Rhs::Tuple || def->rhs.args.size() != 4)
            throw CompileError("OUTPUT_TUPLE argument is not a 4-tuple");
        Block extra;
        for (int i = 0; i < 4; ++i) {
            const Primary &p = def->rhs.args[i];
            if (p.kind == Primary::Val) code.result[i] = p.value;
            else {
                CompVar *t = code.new_var(p.type());
                Stmt *a = code.new_stmt(Stmt::Assign);
                a->lhs = code.new_value(t);
                a->lhs->def = a;
                a->rhs = Rhs::P(p);
                extra.push_back(a);
                code.result[i] = a->lhs;
            }
        }
        // drop the OUTPUT_TUPLE pseudo statement; the backends pack the result themselves
        Block kept;
        for (Stmt *s : code.body)
            if (s != outp) kept.push_back(s);
        for (Stmt *s : extra) kept.push_back(s);
        code.body.swap(kept);
    }
};

// Closure images that reach a native filter's image argument or render(): render_image's closure branch
// (builtins.c:267-302).  Numbered in statement order; `out` gets each closure's defining statement and the chain of
// resize wrappers between it and its first use.
void number_native_closures(Block &b, std::vector<std::pair<Stmt *, ImageChain>> &out) {
    for (Stmt *s : b) {
        if (s->kind == Stmt::Assign) {
            const Rhs &r = s->rhs;
            const bool native = (r.kind == Rhs::Closure && r.filter->kind == Filter::Native) ||
                                (r.kind == Rhs::Op && !strcmp(r.op->cname, "RENDER"));
            if (!native) continue;
            for (const Primary &p : r.args) {
                if (p.kind != Primary::Val || p.value->var->type != Ty::Image) continue;
                ImageChain ch = resolve_image_chain(p.value);
                if (ch.base != ImageChain::MathMapClosure || ch.closure_def->closure_id >= 0) continue;
                ch.closure_def->closure_id = (int)out.size();
                out.push_back({ch.closure_def, ch});
            }
        } else if (s->kind == Stmt::If) {
            number_native_closures(s->then_, out);
            number_native_closures(s->else_, out);
        } else if (s->kind == Stmt::While) {
            number_native_closures(s->body, out);
        }
    }
}

int option_key(const KernelOptions &k) {
    return (k.intersample & 1) | ((k.supersampling & 1) << 1) | ((k.edge_x & 3) << 2) | ((k.edge_y & 3) << 4) | (k.pixel_inc << 6);
}

// The stride of the image sources right now: the host's drawable_get_pixel_inc (mathmap.h:314; mathmap.c:1320-1327
// answers fast_image_source_scale while the GIMP dialog is previewing, cocoa.c:58-62 always 1), which the reference's
// compiled filters call per bilinear fetch (builtins.c:182-184).  Found in the host program by name, like the compiled
// module's own references to host functions; a host without it has full-resolution sources.
int host_pixel_inc(mmabi_invocation_t *inv) {
    typedef void (*get_inc_t)(void *, void *, int *, int *);
    static get_inc_t fn = (get_inc_t)dlsym(RTLD_DEFAULT, "drawable_get_pixel_inc");
    if (!fn) return 1;
    int ix = 1, iy = 1;
    fn(inv, nullptr, &ix, &iy);
    if (ix != iy) return -1;
    return ix > 1 ? ix : 1;
}

Variant *get_variant(ModuleInfo *mi, mmabi_invocation_t *inv) {
    KernelOptions ko;
    ko.intersample = inv->antialiasing ? 1 : 0;       // invocation_set_antialiasing, mathmap_common.c:736-743
    ko.supersampling = inv->supersampling ? 1 : 0;
    ko.edge_x = inv->edge_behaviour_x;                // EDGE_BEHAVIOUR_* (mathmap.h:134-138) = our numbering + 1
    ko.edge_y = inv->edge_behaviour_y;
    // reference: COLOR=1 WRAP=2 REFLECT=3 ROTATE=4
    ko.edge_x = ko.edge_x >= 1 && ko.edge_x <= 4 ? ko.edge_x - 1 : 0;
    ko.edge_y = ko.edge_y >= 1 && ko.edge_y <= 4 ? ko.edge_y - 1 : 0;
    ko.pixel_inc = host_pixel_inc(inv);
    if (ko.pixel_inc < 0 || ko.pixel_inc > 4096) { host_error("HIP backend: drawable_get_pixel_inc answered strides the backend does not take"); return nullptr; }
    Variant &v = mi->variants[option_key(ko)];
    if (!v.flt) {
        std::string err;
        v.flt = mi->build(ko, &err);
        if (!v.flt) { host_error("HIP backend: " + err); return nullptr; }
    }
    return &v;
}

// ---------------------------------------------------------------------------
// mathfuncs shims
// ---------------------------------------------------------------------------
void hip_init_frame(mmabi_frame_t *, mmabi_image_t *) {}
void hip_init_slice(mmabi_slice_t *, mmabi_image_t *) {}

bool upload_drawable(ModuleInfo *mi, mmabi_invocation_t *inv, mmabi_input_drawable_t *d, DrawableCopy *out) {
    auto it = mi->drawables.find(d);
    if (it != mi->drawables.end() && it->second.width == d->image.pixel_width && it->second.height == d->image.pixel_height) {
        *out = it->second;
        return true;
    }
    mmabi_get_pixel_func_t gp = g_get_pixel;
    if (!gp) gp = (mmabi_get_pixel_func_t)dlsym(RTLD_DEFAULT, "mathmap_get_pixel");
    if (!gp) { host_error("HIP backend: host symbol mathmap_get_pixel not found"); return false; }
    int w = d->image.pixel_width, h = d->image.pixel_height;
    std::vector<uint32_t> px((size_t)w * h);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) px[(size_t)y * w + x] = gp(inv, d, 0, x, y);
    DrawableCopy c;
    c.width = w;
    c.height = h;
    if (hipMalloc(&c.dev, px.size() * 4) != hipSuccess) { host_error("HIP backend: out of device memory"); return false; }
    (void)hipMemcpy(c.dev, px.data(), px.size() * 4, hipMemcpyHostToDevice);
    if (it != mi->drawables.end() && it->second.dev) (void)hipFree(it->second.dev);
    mi->drawables[d] = c;
    {
        std::lock_guard<std::mutex> g(g_registry_mu);
        g_drawable_owner[d] = mi;
    }
    *out = c;
    return true;
}

void hip_calc_lines(mmabi_slice_t *slice, mmabi_image_t *closure, int first_row, int last_row, void *q, int floatmap) {
    mmabi_frame_t *frame = slice->frame;
    mmabi_invocation_t *inv = frame->invocation;
    ModuleInfo *mi = (ModuleInfo *)inv->mathmap->module_info;
    std::lock_guard<std::mutex> guard(mi->mu);   // calc_lines may be entered from several row threads
    Variant *v = get_variant(mi, inv);
    if (!v) return;
    mmhip_invocation *&hi = v->invs[inv];
    // a host invocation allocated at a recycled address after the old one was freed: the canvas size
    // tells them apart (everything else is re-sent below with every call)
    if (hi && (hi->img_w != inv->img_width || hi->img_h != inv->img_height)) {
        mmhip_invocation_free(hi);
        hi = nullptr;
    }
    if (!hi) {
        // bound the cache: hosts that never call mathmap_hip_release_invocation (the reference has no
        // hook in free_invocation) would otherwise leak a stream and buffers per invocation.  `hi`
        // refers to the entry of `inv`, which stays (std::map references survive other erasures).
        if (v->invs.size() > 8)
            for (auto it = v->invs.begin(); it != v->invs.end();) {
                if (it->first != inv) { if (it->second) mmhip_invocation_free(it->second); it = v->invs.erase(it); }
                else ++it;
            }
        hi = mmhip_invoke(v->flt, inv->img_width, inv->img_height);
        if (!hi) { host_error(std::string("HIP backend: ") + mmhip_last_error()); return; }
    }
    mmhip_set_render_size(hi, frame->frame_render_width, frame->frame_render_height);
    mmhip_set_sampling_offset(hi, slice->sampling_offset_x, slice->sampling_offset_y);
    mmhip_set_edge_colors(hi, inv->edge_color_x, inv->edge_color_y);
    // user values come from the closure's argument block (new_template.c.in:234)
    const auto &uvs = v->flt->module.main->uservals;
    for (const UservalInfo &u : uvs) {
        const mmabi_userval_t &a = closure->v.closure.args[u.index];
        switch (u.kind) {
            case UvKind::Int: mmhip_set_int(hi, u.index, a.v.int_const); break;
            case UvKind::Float: mmhip_set_float(hi, u.index, a.v.float_const); break;
            case UvKind::Bool: mmhip_set_bool(hi, u.index, a.v.bool_const); break;
            case UvKind::Color:
                hi->uv[u.index].c = a.v.color.value;
                hi->tables_dirty = true;
                break;
            case UvKind::Image: {
                mmabi_image_t *img = a.v.image;
                while (img && img->type == MMABI_IMAGE_RESIZE) img = img->v.resize.original;
                if (img && img->type == MMABI_IMAGE_DRAWABLE && img->v.drawable) {
                    DrawableCopy c;
                    if (!upload_drawable(mi, inv, img->v.drawable, &c)) return;
                    int slot = hi->image_slot_of_uv[u.index];
                    if (hi->images[slot].data != c.dev) mmhip_set_image_device(hi, u.index, c.dev, c.width, c.height);
                    // the host may have set its own scale/middle (userval.c:262-280); keep them
                    hi->images[slot].scale_x = img->v.drawable->scale_x;
                    hi->images[slot].scale_y = img->v.drawable->scale_y;
                    hi->images[slot].middle_x = img->v.drawable->middle_x;
                    hi->images[slot].middle_y = img->v.drawable->middle_y;
                } else if (img && img->type == MMABI_IMAGE_CLOSURE) {
                    host_error("HIP backend: closure images as top-level arguments are not supported yet");
                    return;
                }
                break;
            }
            case UvKind::Curve:      // curve_t.values: USER_CURVE_POINTS floats (userval.h:38,89-96)
                if (a.v.curve && a.v.curve->values && mmhip_set_curve(hi, u.index, a.v.curve->values) != 0) {
                    host_error(std::string("HIP backend: ") + mmhip_last_error());
                    return;
                }
                break;
            case UvKind::Gradient:   // gradient_t.values: USER_GRADIENT_POINTS color_t (userval.h:39,98-101)
                if (a.v.gradient && a.v.gradient->values && mmhip_set_gradient(hi, u.index, a.v.gradient->values) != 0) {
                    host_error(std::string("HIP backend: ") + mmhip_last_error());
                    return;
                }
                break;
            default: break;
        }
    }
    // new_template.c.in:238-239
    if (first_row < 0) first_row = 0;
    if (last_row > slice->region_y + slice->region_height) last_row = slice->region_y + slice->region_height;
    int rows = last_row - first_row;
    if (rows <= 0) return;
    int bpp = inv->output_bpp;
    size_t dev_stride = floatmap ? (size_t)frame->frame_render_width * 16 : (size_t)slice->region_width * bpp;
    size_t need = dev_stride * rows;
    if (need > mi->staging_bytes) {
        if (mi->staging) (void)hipFree(mi->staging);
        mi->staging = nullptr;
        mi->staging_bytes = 0;
        if (hipMalloc(&mi->staging, need) != hipSuccess) { host_error("HIP backend: out of device memory"); return; }
        mi->staging_bytes = need;
    }
    int rc = mmhip_render(hi, frame->current_frame, frame->current_t, slice->region_x, slice->region_y, slice->region_width,
                          slice->region_height, first_row, last_row, mi->staging, (int)dev_stride, bpp, floatmap, nullptr);
    if (rc != 0) { host_error(std::string("HIP backend: ") + mmhip_last_error()); return; }
    mmhip_sync(hi);
    if (floatmap) {
        // rows are frame_render_width float4s apart on both sides (new_template.c.in:297); only the
        // region's columns were rendered
        (void)hipMemcpy2D(q, dev_stride, mi->staging, dev_stride, (size_t)slice->region_width * 16, (size_t)rows, hipMemcpyDeviceToHost);
    } else {
        (void)hipMemcpy2D(q, (size_t)inv->row_stride, mi->staging, dev_stride, dev_stride, (size_t)rows, hipMemcpyDeviceToHost);
    }
    // new_template.c.in:307-308
    if (!inv->supersampling && inv->rows_finished)
        for (int r = first_row - slice->region_y; r < last_row - slice->region_y; ++r) inv->rows_finished[r] = 1;
}

mmabi_mathfuncs_t hip_mathmapinit(mmabi_invocation_t *) {
    mmabi_mathfuncs_t f;
    memset(&f, 0, sizeof f);
    f.init_frame = hip_init_frame;
    f.init_slice = hip_init_slice;
    f.calc_lines = hip_calc_lines;
    return f;
}

}  // namespace

extern "C" {

mmabi_initfunc_t gen_and_load_hip_code(mmabi_mathmap_t *mathmap, void **module_info, char *, char *,
                                       mmabi_filter_code_t **filter_codes) {
    int index = 0;
    mmabi_filter_code_t *main_code = nullptr;
    for (mmabi_filter_t *f = mathmap->filters; f; f = f->next, ++index)
        if (f == mathmap->main_filter) main_code = filter_codes[index];
    if (!main_code) { host_error("HIP backend: main filter has no code"); return nullptr; }

    // Import once into a pristine filter object; per option set we re-import from the
    // reference structures only now (they are freed right after this call,
    // mathmap_common.c:558), so keep a serialised pristine copy: the IR JSON is not
    // re-parsable here, therefore all option variants are built eagerly on demand from
    // a deep copy made by importing again into separate objects up front.
    std::unique_ptr<ModuleInfo> mi(new ModuleInfo());
    mi->mathmap = mathmap;
    // Pre-build the 2 sampling variants x default edge mode now; other edge modes are
    // rare (GIMP dialog only) and are imported eagerly too to stay independent of the
    // freed IR.
    std::vector<KernelOptions> wanted;
    for (int inter = 0; inter < 2; ++inter)
        for (int ss = 0; ss < 2; ++ss)
            for (int ex = 0; ex < 4; ++ex)
                for (int ey = 0; ey < 4; ++ey) {
                    KernelOptions k;
                    k.intersample = inter; k.supersampling = ss; k.edge_x = ex; k.edge_y = ey;
                    wanted.push_back(k);
                }
    // importing is cheap; hiprtc compilation happens lazily on first use of a variant
    std::map<int, mmhip_filter *> prebuilt;
    std::string first_err;
    for (const KernelOptions &k : wanted) {
        mmhip_filter *f = mmhip_filter_new_empty();
        try {
            f->code.reset(new FilterCode());
            Importer::Shared shared;
            Importer imp(f->module, *f->code, shared);
            f->module.main = imp.filter(mathmap->main_filter);
            f->code->filter = f->module.main;
            imp.block(main_code->first_stmt, f->code->body, nullptr);
            imp.find_result();
            propagate_types(*f->code);      // the types are the reference's; this fills in the tuple / tree-vector lengths
            // The render code of every closure image a native filter (or render()) is given: the main code once more
            // -- it computes the closure's arguments, at the frame's own time -- followed by a call of the closure's
            // filter function at the pixel's raw coordinates with t = 0.0 (the function's own `frame' is 0), its value
            // as the result.  What the call does not need is dead code for the ordinary passes.  (The standalone
            // tier inlines the closure's body instead, lower.cpp native_image_argument; same values.)
            std::vector<std::pair<Stmt *, ImageChain>> closures;
            number_native_closures(f->code->body, closures);
            for (size_t k = 0; k < closures.size(); ++k) {
                std::unique_ptr<FilterCode> sub(new FilterCode());
                sub->filter = f->module.main;
                Importer imps(f->module, *sub, shared);
                imps.block(main_code->first_stmt, sub->body, nullptr);
                imps.find_result();
                propagate_types(*sub);
                std::vector<std::pair<Stmt *, ImageChain>> again;
                number_native_closures(sub->body, again);
                if (again.size() != closures.size()) throw CompileError("internal: closure numbering differs between imports");
                Stmt *def = again[k].first;
                // the conditionals around the closure, outermost first (a loop would make it one closure per iteration)
                std::vector<std::pair<Stmt *, int>> nest;
                for (const Stmt *c = def; c->parent; c = c->parent) {
                    Stmt *p = c->parent;
                    if (p->kind != Stmt::If) throw CompileError("a filter closure passed to a native filter inside a loop is not supported");
                    const bool in_then = std::find(p->then_.begin(), p->then_.end(), c) != p->then_.end();
                    nest.insert(nest.begin(), {p, in_then ? 0 : 1});
                }
                Gen g(*sub);
                // The result: four variables, 0 at the top of the body, assigned next to the closure; each enclosing `if'
                // gets exit phis that carry the value -- or, where the branch does not run and the native filter is not
                // called either, the 0 -- to the top level (lower.cpp native_image_argument does the same)
                CompVar *rv[4];
                std::vector<Stmt *> inits;
                for (int i = 0; i < 4; ++i) {
                    rv[i] = g.temp(Ty::Float);
                    inits.push_back(g.assign(rv[i], Rhs::F(0.0f))->def);
                }
                for (Stmt *st : inits) sub->body.erase(std::find(sub->body.begin(), sub->body.end(), st));
                sub->body.insert(sub->body.begin(), inits.begin(), inits.end());
                for (auto &lv : nest) g.reenter_if(lv.first, lv.second);
                Primary rx, ry;
                emit_closure_render_coordinates(g, again[k].second, again[k].second.factors.empty(), &rx, &ry);
                // The closure's own code: normally a call of its filter function.  When that code calls native filters
                // itself (the closure's calc_lines would run them from its init_frame, new_template.c.in:314-337) it is
                // inlined instead -- user-value reads become the closure's arguments, x / y / t the render coordinates and
                // 0.0, frame 0 -- so that its native calls sit in this kernel's frame-constant slice like any other and
                // runtime.cpp render_closure runs them (a native filter inside a *called* function would run per pixel).
                mmabi_filter_code_t *callee_code = nullptr;
                {
                    int idx = 0;
                    for (mmabi_filter_t *gf = mathmap->filters; gf; gf = gf->next, ++idx)
                        if (shared.filters.count(gf) && shared.filters[gf] == def->rhs.filter) callee_code = filter_codes[idx];
                }
                Block inl;
                bool inline_body = false;
                Importer impc(f->module, *sub, shared);
                if (callee_code) {
                    impc.block(callee_code->first_stmt, inl, nullptr);
                    std::vector<std::pair<Stmt *, ImageChain>> inner;
                    std::function<bool(const Block &)> has_native = [&](const Block &b) {
                        for (const Stmt *st : b) {
                            if (st->kind == Stmt::Assign && ((st->rhs.kind == Rhs::Closure && st->rhs.filter->kind == Filter::Native) ||
                                                             (st->rhs.kind == Rhs::Op && !strcmp(st->rhs.op->cname, "RENDER")))) return true;
                            if (st->kind == Stmt::If && (has_native(st->then_) || has_native(st->else_))) return true;
                            if (st->kind == Stmt::While && has_native(st->body)) return true;
                        }
                        return false;
                    };
                    inline_body = has_native(inl);
                }
                if (inline_body) {
                    Stmt *outp = nullptr;
                    for (Stmt *st : inl)
                        if (st->kind == Stmt::Assign && st->rhs.kind == Rhs::Op && !strcmp(st->rhs.op->cname, "OUTPUT_TUPLE")) outp = st;
                    if (!outp || outp->rhs.args[0].kind != Primary::Val || !outp->rhs.args[0].value->def ||
                        outp->rhs.args[0].value->def->rhs.kind != Rhs::Tuple || outp->rhs.args[0].value->def->rhs.args.size() != 4)
                        throw CompileError("closure filter `" + def->rhs.filter->name + "' has no OUTPUT_TUPLE of a 4-tuple");
                    const std::vector<Primary> result = outp->rhs.args[0].value->def->rhs.args;
                    const std::vector<Primary> cargs = def->rhs.args;
                    std::function<void(Rhs &)> subst_rhs = [&](Rhs &r) {
                        if (r.kind == Rhs::Internal) {
                            if (r.internal == "x") r = Rhs::P(rx);
                            else if (r.internal == "y") r = Rhs::P(ry);
                            else if (r.internal == "t") r = Rhs::F(0.0f);
                            else if (r.internal == "frame") r = Rhs::I(0);
                        } else if (r.kind == Rhs::Op && !strncmp(r.op->cname, "USERVAL_", 8) && r.args.size() == 1 &&
                                   r.args[0].kind == Primary::IntConst) {
                            const int ui = r.args[0].i;
                            if (ui < 0 || ui >= (int)cargs.size()) throw CompileError("closure argument index out of range");
                            r = Rhs::P(cargs[ui]);
                        }
                    };
                    std::function<void(Block &)> subst = [&](Block &b) {
                        for (Stmt *st : b) {
                            if (st->kind == Stmt::Assign) subst_rhs(st->rhs);
                            if (st->kind == Stmt::Phi) { subst_rhs(st->rhs); subst_rhs(st->rhs2); }
                            if (st->kind == Stmt::If) { subst_rhs(st->cond); subst(st->then_); subst(st->else_); subst(st->phis); }
                            if (st->kind == Stmt::While) { subst(st->phis); subst_rhs(st->cond); subst(st->body); }
                        }
                    };
                    subst(inl);
                    for (Stmt *st : inl)
                        if (st != outp) g.append(st);
                    for (int i = 0; i < 4; ++i) g.assign(rv[i], Rhs::P(result[i]));
                } else {
                    Rhs call;
                    call.kind = Rhs::FilterCall;
                    call.filter = def->rhs.filter;
                    call.args = def->rhs.args;
                    call.args.push_back(rx);
                    call.args.push_back(ry);
                    call.args.push_back(Primary::F(0.0f));
                    CompVar *tv = g.temp(Ty::Tuple);
                    tv->tuple_len = 4;
                    g.assign(tv, call);
                    for (int i = 0; i < 4; ++i) g.assign_op(rv[i], "TUPLE_NTH", {g.P(tv), Primary::I(i)});
                }
                for (size_t lv = 0; lv < nest.size(); ++lv) g.end_if();
                for (int i = 0; i < 4; ++i) sub->result[i] = rv[i]->current;
                propagate_types(*sub);          // (tuple lengths of what was added)
                if (!inline_body)
                    for (auto &kv : shared.filters)
                        if (kv.second == def->rhs.filter && std::find(shared.called.begin(), shared.called.end(), kv.first) == shared.called.end())
                            shared.called.push_back(kv.first);
                f->code->closure_renders.push_back(std::move(sub));
            }
            // filter_$name bodies of the filters called at run time, and of those they call (backends/cc.c:189-196
            // prints every filter's code as a function; only the called ones are needed here)
            for (size_t ci = 0; ci < shared.called.size(); ++ci) {
                mmabi_filter_code_t *fc = nullptr;
                int idx = 0;
                for (mmabi_filter_t *g = mathmap->filters; g; g = g->next, ++idx)
                    if (g == shared.called[ci]) fc = filter_codes[idx];
                if (!fc) throw CompileError(std::string("filter `") + shared.called[ci]->name + "' is called but has no code");
                std::unique_ptr<FilterCode> fn(new FilterCode());
                fn->filter = imp.filter(shared.called[ci]);
                Importer impf(f->module, *fn, shared);
                impf.block(fc->first_stmt, fn->body, nullptr);
                impf.find_result();
                propagate_types(*fn);
                f->code->functions.push_back(std::move(fn));
            }
            std::string err;
            if (!mmhip_filter_finalize(f, k, &err)) throw CompileError(err);
            // user values arrive with every calc_lines call: specialise a kernel for a value set once it
            // is rendered for a second frame (an animation over t), never for bands of one frame or
            // for values that change with every render (slider moves, animated user values);
            // MATHMAP_HIP_SPECIALIZE=0 turns it off
            const char *e = getenv("MATHMAP_HIP_SPECIALIZE");
            f->specialize = !(e && atoi(e) == 0);
            f->spec_min_uses = 2;
        } catch (const std::exception &e) {
            first_err = e.what();
            mmhip_filter_free(f);
            f = nullptr;
        }
        if (!f) break;
        prebuilt[option_key(k)] = f;
    }
    if (!first_err.empty()) {
        for (auto &p : prebuilt) mmhip_filter_free(p.second);
        host_error("HIP backend: " + first_err);
        return nullptr;
    }
    for (auto &p : prebuilt) mi->variants[p.first].flt = p.second;
    // Variants beyond the prepared ones differ in the source stride only (drawable_get_pixel_inc > 1: the dialog's
    // preview): built when first asked for, from the IR dump of the prepared variant with the same other options.
    ModuleInfo *mip = mi.get();
    mi->build = [mip](const KernelOptions &ko, std::string *err) -> mmhip_filter * {
        KernelOptions base = ko;
        base.pixel_inc = 1;
        auto it = mip->variants.find(option_key(base));
        if (ko.pixel_inc <= 1 || it == mip->variants.end() || !it->second.flt) {
            *err = "kernel variant was not prepared";
            return nullptr;
        }
        const mmhip_filter *b = it->second.flt;
        mmhip_filter *f = mmhip_filter_new_empty();
        try {
            f->code.reset(new FilterCode());
            load_ir_json(f->module, *f->code, (b->ir_json_raw.empty() ? b->ir_json : b->ir_json_raw).c_str());
            if (!mmhip_filter_finalize(f, ko, err)) throw CompileError(*err);
            f->specialize = b->specialize;
            f->spec_min_uses = b->spec_min_uses;
        } catch (const std::exception &e) {
            *err = e.what();
            mmhip_filter_free(f);
            return nullptr;
        }
        return f;
    };
    *module_info = mi.release();
    return hip_mathmapinit;
}

void unload_hip_code(void *module_info) {
    ModuleInfo *mi = (ModuleInfo *)module_info;
    if (!mi) return;
    {
        std::lock_guard<std::mutex> g(g_registry_mu);
        for (auto it = g_drawable_owner.begin(); it != g_drawable_owner.end();)
            it = it->second == mi ? g_drawable_owner.erase(it) : std::next(it);
    }
    for (auto &v : mi->variants) {
        for (auto &i : v.second.invs) mmhip_invocation_free(i.second);
        mmhip_filter_free(v.second.flt);
    }
    for (auto &d : mi->drawables) if (d.second.dev) (void)hipFree(d.second.dev);
    if (mi->staging) (void)hipFree(mi->staging);
    delete mi;
}

void mathmap_hip_set_get_pixel(mmabi_get_pixel_func_t fn) { g_get_pixel = fn; }

// Lock order: mi->mu before g_registry_mu (hip_calc_lines registers a drawable while it holds the
// module's lock), so the owner is looked up and unregistered first and the registry lock released
// before the module is locked.  The host must not unload the module concurrently (it never does:
// both are main-thread operations).
void mathmap_hip_invalidate_drawable(mmabi_input_drawable_t *drawable) {
    ModuleInfo *mi = nullptr;
    {
        std::lock_guard<std::mutex> g(g_registry_mu);
        auto it = g_drawable_owner.find(drawable);
        if (it == g_drawable_owner.end()) return;
        mi = it->second;
        g_drawable_owner.erase(it);
    }
    std::lock_guard<std::mutex> g2(mi->mu);     // waits for a render in flight
    auto d = mi->drawables.find(drawable);
    if (d != mi->drawables.end()) {
        for (auto &v : mi->variants)            // no invocation may keep the freed pointer bound
            for (auto &i : v.second.invs)
                if (i.second)
                    for (auto &img : i.second->images)
                        if (img.data == d->second.dev) { img.kind = IMG_NULL; img.data = nullptr; i.second->tables_dirty = true; ++i.second->input_generation; }
        (void)hipDeviceSynchronize();
        if (d->second.dev) (void)hipFree(d->second.dev);
        mi->drawables.erase(d);
    }
}

// Drops the device-side state (stream, buffers, native-filter memo) kept for a host invocation; call
// it from free_invocation (mathmap_common.c:303-319).  Optional: the cache is bounded without it.
void mathmap_hip_release_invocation(mmabi_invocation_t *inv) {
    if (!inv || !inv->mathmap || !inv->mathmap->module_info) return;
    ModuleInfo *mi = (ModuleInfo *)inv->mathmap->module_info;
    std::lock_guard<std::mutex> g(mi->mu);
    for (auto &v : mi->variants) {
        auto it = v.second.invs.find(inv);
        if (it == v.second.invs.end()) continue;
        if (it->second) mmhip_invocation_free(it->second);
        v.second.invs.erase(it);
    }
}

}  // extern "C"
