// Self-test of the reference-ABI tier.  The reference itself cannot be built in this
// environment (needs clisp/bison/glib/GIMP headers), so the importer is exercised by
// exporting our own IR into the reference's structure layouts (mathmap_abi.h), then
// driving gen_and_load_hip_code() and the returned mathfuncs exactly as
// mathmap_common.c does (init_invocation :714-734, invocation_new_frame :797-816,
// invocation_init_slice :848-865, calc_lines :837-846).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>

#include "../../include/mathmap_hip_backend.h"
#include "../../include/mmhip.h"
#include "front.h"

namespace mm { void load_ir_json(Module &mod, FilterCode &code, const char *json); }   // ir_json.cpp

using namespace mm;

namespace {

struct Exporter {
    std::deque<mmabi_statement_t> stmts;
    std::deque<mmabi_rhs_t> rhss;
    std::deque<mmabi_value_t> values;
    std::deque<mmabi_compvar_t> compvars;
    std::deque<mmabi_temporary_t> temps;
    std::deque<mmabi_variable_t> variables;
    std::deque<mmabi_operation_t> ops;
    std::deque<mmabi_internal_t> internals;
    std::deque<mmabi_filter_t> filters;
    std::deque<mmabi_userval_info_t> uvinfos;
    std::deque<mmabi_option_t> options;
    std::deque<mmabi_top_level_decl_t> decls;
    std::deque<std::vector<mmabi_primary_t>> primary_arrays;
    std::deque<std::string> strings;
    std::map<const CompVar *, mmabi_compvar_t *> cv;
    std::map<const Value *, mmabi_value_t *> vv;
    std::map<const OpInfo *, mmabi_operation_t *> opmap;
    std::map<std::string, mmabi_internal_t *> intmap;
    std::map<const Filter *, mmabi_filter_t *> fmap;

    char *str(const std::string &s) {
        strings.push_back(s);
        return const_cast<char *>(strings.back().c_str());
    }

    mmabi_compvar_t *compvar(const CompVar *c) {
        auto it = cv.find(c);
        if (it != cv.end()) return it->second;
        compvars.emplace_back();
        mmabi_compvar_t *r = &compvars.back();
        memset(r, 0, sizeof *r);
        r->index = c->id;
        r->type = (int)c->type;
        r->n = c->elem;
        if (c->is_temp) {
            temps.emplace_back();
            temps.back().number = c->id;
            r->temp = &temps.back();
        } else {
            variables.emplace_back();
            memset(&variables.back(), 0, sizeof(mmabi_variable_t));
            variables.back().name = str(c->name);
            r->var = &variables.back();
        }
        cv[c] = r;
        return r;
    }

    mmabi_value_t *value(const Value *v) {
        auto it = vv.find(v);
        if (it != vv.end()) return it->second;
        values.emplace_back();
        mmabi_value_t *r = &values.back();
        memset(r, 0, sizeof *r);
        r->compvar = compvar(v->var);
        r->index = v->index;
        r->global_index = v->gid;
        vv[v] = r;
        return r;
    }

    mmabi_primary_t primary(const Primary &p) {
        mmabi_primary_t r;
        memset(&r, 0, sizeof r);
        switch (p.kind) {
            case Primary::Val: r.kind = MMABI_PRIMARY_VALUE; r.v.value = value(p.value); break;
            case Primary::IntConst: r.kind = MMABI_PRIMARY_CONST; r.const_type = MMABI_TYPE_INT; r.v.constant.int_value = p.i; break;
            case Primary::FloatConst: r.kind = MMABI_PRIMARY_CONST; r.const_type = MMABI_TYPE_FLOAT; r.v.constant.float_value = p.f; break;
            default: r.kind = MMABI_PRIMARY_CONST; r.const_type = MMABI_TYPE_INT; break;
        }
        return r;
    }

    mmabi_filter_t *filter(const Filter *f) {
        auto it = fmap.find(f);
        if (it != fmap.end()) return it->second;
        filters.emplace_back();
        mmabi_filter_t *r = &filters.back();
        memset(r, 0, sizeof *r);
        r->name = str(f->name);
        r->num_uservals = (int)f->uservals.size();
        mmabi_userval_info_t *prev = nullptr;
        for (const UservalInfo &u : f->uservals) {
            uvinfos.emplace_back();
            mmabi_userval_info_t *ui = &uvinfos.back();
            memset(ui, 0, sizeof *ui);
            ui->name = str(u.name);
            ui->index = u.index;
            switch (u.kind) {
                case UvKind::Int: ui->type = MMABI_USERVAL_INT_CONST; ui->v.int_const.min = u.imin; ui->v.int_const.max = u.imax; ui->v.int_const.default_value = u.idef; break;
                case UvKind::Float: ui->type = MMABI_USERVAL_FLOAT_CONST; ui->v.float_const.min = u.fmin; ui->v.float_const.max = u.fmax; ui->v.float_const.default_value = u.fdef; break;
                case UvKind::Bool: ui->type = MMABI_USERVAL_BOOL_CONST; ui->v.bool_const.default_value = u.bdef; break;
                case UvKind::Color: ui->type = MMABI_USERVAL_COLOR; break;
                case UvKind::Curve: ui->type = MMABI_USERVAL_CURVE; break;
                case UvKind::Gradient: ui->type = MMABI_USERVAL_GRADIENT; break;
                case UvKind::Image: ui->type = MMABI_USERVAL_IMAGE; ui->v.image.flags = u.image_flags; break;
            }
            if (prev) prev->next = ui; else r->userval_infos = ui;
            prev = ui;
        }
        if (f->kind == Filter::Native) {
            r->kind = MMABI_FILTER_NATIVE;
            r->v.native.func_name = str(f->native_func);
            r->v.native.is_pure = 1;
        } else {
            r->kind = MMABI_FILTER_MATHMAP;
            decls.emplace_back();
            mmabi_top_level_decl_t *d = &decls.back();
            memset(d, 0, sizeof *d);
            d->name = r->name;
            auto add_opt = [&](const char *n) {
                options.emplace_back();
                options.back().name = str(n);
                options.back().suboptions = nullptr;
                options.back().next = d->v.filter.options;
                d->v.filter.options = &options.back();
            };
            if (!(f->flags & IMAGE_FLAG_UNIT)) add_opt("pixel");
            else if (!(f->flags & IMAGE_FLAG_SQUARE)) add_opt("stretched");
            r->v.mathmap.decl = d;
        }
        fmap[f] = r;
        return r;
    }

    mmabi_rhs_t *rhs(const Rhs &s) {
        rhss.emplace_back();
        mmabi_rhs_t *r = &rhss.back();
        memset(r, 0, sizeof *r);
        switch (s.kind) {
            case Rhs::Prim: r->kind = MMABI_RHS_PRIMARY; r->v.primary = primary(s.prim); break;
            case Rhs::Internal: {
                r->kind = MMABI_RHS_INTERNAL;
                auto it = intmap.find(s.internal);
                if (it == intmap.end()) {
                    internals.emplace_back();
                    memset(&internals.back(), 0, sizeof(mmabi_internal_t));
                    strncpy(internals.back().name, s.internal.c_str(), 63);
                    it = intmap.emplace(s.internal, &internals.back()).first;
                }
                r->v.internal = it->second;
                break;
            }
            case Rhs::Op: {
                r->kind = MMABI_RHS_OP;
                auto it = opmap.find(s.op);
                if (it == opmap.end()) {
                    ops.emplace_back();
                    mmabi_operation_t *o = &ops.back();
                    memset(o, 0, sizeof *o);
                    o->index = s.op->index;
                    o->name = str(s.op->cname);
                    o->num_args = s.op->nargs;
                    it = opmap.emplace(s.op, o).first;
                }
                r->v.op.op = it->second;
                for (size_t i = 0; i < s.args.size(); ++i) r->v.op.args[i] = primary(s.args[i]);
                break;
            }
            case Rhs::Closure: {
                r->kind = MMABI_RHS_CLOSURE;
                r->v.closure.filter = filter(s.filter);
                primary_arrays.emplace_back();
                for (const Primary &p : s.args) primary_arrays.back().push_back(primary(p));
                r->v.closure.args = primary_arrays.back().data();
                break;
            }
            case Rhs::Tuple:
            case Rhs::TreeVector: {
                r->kind = s.kind == Rhs::Tuple ? MMABI_RHS_TUPLE : MMABI_RHS_TREE_VECTOR;
                r->v.tuple.length = (int)s.args.size();
                primary_arrays.emplace_back();
                for (const Primary &p : s.args) primary_arrays.back().push_back(primary(p));
                r->v.tuple.args = primary_arrays.back().data();
                break;
            }
            case Rhs::None: break;
            case Rhs::FilterCall: {
                r->kind = MMABI_RHS_FILTER;
                r->v.filter.filter = filter(s.filter);
                primary_arrays.emplace_back();
                for (const Primary &p : s.args) primary_arrays.back().push_back(primary(p));
                r->v.filter.args = primary_arrays.back().data();
                break;
            }
        }
        return r;
    }

    mmabi_statement_t *new_stmt(int kind) {
        stmts.emplace_back();
        mmabi_statement_t *s = &stmts.back();
        memset(s, 0, sizeof *s);
        s->kind = kind;
        return s;
    }

    mmabi_statement_t *phis(const Block &b) {
        mmabi_statement_t *first = nullptr, **tail = &first;
        for (const Stmt *p : b) {
            if (p->kind != Stmt::Phi) continue;
            mmabi_statement_t *s = new_stmt(MMABI_STMT_PHI_ASSIGN);
            s->v.assign.lhs = value(p->lhs);
            s->v.assign.rhs = rhs(p->rhs);
            s->v.assign.rhs2 = rhs(p->rhs2);
            *tail = s;
            tail = &s->next;
        }
        if (!first) first = new_stmt(MMABI_STMT_NIL);   // the reference terminates empty lists with a NIL statement
        return first;
    }

    mmabi_statement_t *block(const Block &b) {
        mmabi_statement_t *first = nullptr, **tail = &first;
        for (const Stmt *st : b) {
            mmabi_statement_t *s = nullptr;
            switch (st->kind) {
                case Stmt::Assign:
                    s = new_stmt(MMABI_STMT_ASSIGN);
                    s->v.assign.lhs = value(st->lhs);
                    s->v.assign.rhs = rhs(st->rhs);
                    break;
                case Stmt::If:
                    s = new_stmt(MMABI_STMT_IF_COND);
                    s->v.if_cond.condition = rhs(st->cond);
                    s->v.if_cond.consequent = block(st->then_);
                    s->v.if_cond.alternative = block(st->else_);
                    s->v.if_cond.exit = phis(st->phis);
                    break;
                case Stmt::While:
                    s = new_stmt(MMABI_STMT_WHILE_LOOP);
                    s->v.while_loop.entry = phis(st->phis);
                    s->v.while_loop.invariant = rhs(st->cond);
                    s->v.while_loop.body = block(st->body);
                    break;
                default: continue;
            }
            *tail = s;
            tail = &s->next;
        }
        if (!first) first = new_stmt(MMABI_STMT_NIL);
        return first;
    }
};

const uint8_t *g_img = nullptr;
int g_iw = 0, g_ih = 0, g_ic = 3;

mmabi_color_t selftest_get_pixel(mmabi_invocation_t *inv, mmabi_input_drawable_t *d, int frame, int x, int y) {
    // mathmap.c:1195-1209 + mathmap_cmdline.c:131-184
    if (x < 0 || x >= d->image.pixel_width) return inv->edge_color_x;
    if (y < 0 || y >= d->image.pixel_height) return inv->edge_color_y;
    (void)frame;
    const uint8_t *p = g_img + (size_t)g_ic * ((size_t)g_iw * y + x);
    return ((mmabi_color_t)p[0] << 24) | ((mmabi_color_t)p[1] << 16) | ((mmabi_color_t)p[2] << 8) | (g_ic == 4 ? p[3] : 255u);
}

thread_local std::string g_selftest_err;

}  // namespace

extern "C" {

const char *mmhip_selftest_error(void) { return g_selftest_err.c_str(); }

// the host's drawable_get_pixel_inc (mathmap.h:314; mathmap.c:1320-1327): the stride of the image sources, > 1 while
// the GIMP dialog previews.  The backend finds it by name, like a compiled module would.
static int g_pixel_inc = 1;
void drawable_get_pixel_inc(void *, void *, int *inc_x, int *inc_y) { *inc_x = *inc_y = g_pixel_inc; }
void mmhip_selftest_set_pixel_inc(int inc) { g_pixel_inc = inc; }

// the host's error buffer: the backend reports through it like the reference's backends do (exprtree.c:40, cc.c:653-693)
char error_string[1024];

// Compiles `source` with our front-end, exports the (un-optimised) IR into reference-layout
// structures, pushes it through gen_and_load_hip_code and renders a w x h frame through the
// returned mathfuncs into `out_rgba` (w*h*4 bytes).  Returns 0 on success.
int mmhip_selftest_abi_roundtrip(const char *source, int intersample, const uint8_t *image, int iw, int ih, int ichannels,
                                 int w, int h, float t, int num_bands, uint8_t *out_rgba) {
    try {
        // `source`: .mm text, or -- first character '{' -- an IR dump (mmhip_filter_ir_json_raw's form): the
        // reference's own filters are kept as IR fixtures only (tests/filters.py)
        Module m;
        std::unique_ptr<FilterCode> code;
        const char *first = source;
        while (*first == ' ' || *first == '\n' || *first == '\t' || *first == '\r') ++first;
        const bool from_ir = *first == '{';
        if (from_ir) {
            code.reset(new FilterCode());
            load_ir_json(m, *code, source);
        } else {
            parse_module(m, source);
            code = lower_filter(m, m.main);
        }

        Exporter ex;
        // filter list in module order (native filters first, like register_native_filters
        // runs before parsing, mathmap_common.c:411-421)
        std::vector<mmabi_filter_t *> flist;
        for (auto &f : m.filters) flist.push_back(ex.filter(f.get()));
        for (size_t i = 0; i + 1 < flist.size(); ++i) flist[i]->next = flist[i + 1];

        // body + `tuple = (r,g,b,a); dummy = OUTPUT_TUPLE(tuple)` (compiler.c:4692-4697), for the main filter and
        // for every filter it calls at run time (compiler_compile_filters hands the backend one code per filter)
        auto export_code = [&](FilterCode &c, mmabi_filter_code_t &out) {
            CompVar *tuple = c.new_var(Ty::Tuple);
            tuple->tuple_len = 4;
            Stmt *ta = c.new_stmt(Stmt::Assign);
            ta->lhs = c.new_value(tuple);
            ta->rhs.kind = Rhs::Tuple;
            for (int i = 0; i < 4; ++i) ta->rhs.args.push_back(Primary::V(c.result[i]));
            c.body.push_back(ta);
            CompVar *dummy = c.new_var(Ty::Int);
            Stmt *oa = c.new_stmt(Stmt::Assign);
            oa->lhs = c.new_value(dummy);
            oa->rhs = Rhs::O(op_by_cname("OUTPUT_TUPLE", 1), {Primary::V(ta->lhs)});
            c.body.push_back(oa);
            out.first_stmt = ex.block(c.body);
        };
        mmabi_filter_code_t fc;
        fc.filter = ex.filter(m.main);
        export_code(*code, fc);
        std::deque<mmabi_filter_code_t> fn_codes;
        std::vector<const Filter *> have;
        for (auto &fn : code->functions) {
            fn_codes.emplace_back();
            fn_codes.back().filter = ex.filter(fn->filter);
            export_code(*fn, fn_codes.back());
            have.push_back(fn->filter);
        }
        // the reference hands the backend a code for *every* filter of the module (compiler_compile_filters); the ones a
        // backend can need besides the called ones are the filters whose closures are made in the main code (a closure
        // image given to a native filter is rendered by its own filter's code)
        std::vector<std::unique_ptr<FilterCode>> closure_codes;
        std::function<void(const Block &)> find_closures = [&](const Block &b) {
            for (const Stmt *st : b) {
                if (st->kind == Stmt::Assign && st->rhs.kind == Rhs::Closure && st->rhs.filter->kind == Filter::MathMap &&
                    st->rhs.filter != m.main && std::find(have.begin(), have.end(), st->rhs.filter) == have.end()) {
                    have.push_back(st->rhs.filter);
                    // an IR dump carries no bodies of the filters whose closures it applies inline: no code to hand over (the
                    // backend asks for one only when the closure image reaches a native filter, and reports that itself)
                    if (from_ir) continue;
                    closure_codes.push_back(lower_function(m, const_cast<Filter *>(st->rhs.filter)));
                    fn_codes.emplace_back();
                    fn_codes.back().filter = ex.filter(st->rhs.filter);
                    export_code(*closure_codes.back(), fn_codes.back());
                }
                if (st->kind == Stmt::If) { find_closures(st->then_); find_closures(st->else_); }
                if (st->kind == Stmt::While) find_closures(st->body);
            }
        };
        find_closures(code->body);

        std::vector<mmabi_filter_code_t *> codes(flist.size(), nullptr);
        for (size_t i = 0; i < flist.size(); ++i) {
            if (flist[i] == fc.filter) codes[i] = &fc;
            for (mmabi_filter_code_t &c : fn_codes)
                if (flist[i] == c.filter && !(c.filter == fc.filter)) codes[i] = &c;
        }

        mmabi_mathmap_t mathmap;
        memset(&mathmap, 0, sizeof mathmap);
        mathmap.filters = flist.empty() ? nullptr : flist[0];
        mathmap.main_filter = fc.filter;

        g_img = image; g_iw = iw; g_ih = ih; g_ic = ichannels;
        mathmap_hip_set_get_pixel(selftest_get_pixel);
        mathmap.initfunc = gen_and_load_hip_code(&mathmap, &mathmap.module_info, nullptr, nullptr, codes.data());
        if (!mathmap.initfunc) { g_selftest_err = std::string("gen_and_load_hip_code failed: ") + error_string + mmhip_last_error(); return -1; }

        // invoke_mathmap defaults (mathmap_common.c:746-795)
        mmabi_invocation_t inv;
        memset(&inv, 0, sizeof inv);
        inv.mathmap = &mathmap;
        inv.antialiasing = intersample;
        inv.output_bpp = 4;
        inv.edge_behaviour_x = inv.edge_behaviour_y = 1;
        inv.img_width = inv.render_width = w;
        inv.img_height = inv.render_height = h;
        inv.image_R = (float)sqrt(2.0);
        inv.row_stride = w * 4;
        std::vector<unsigned char> rows_finished(h, 0);
        inv.rows_finished = rows_finished.data();
        inv.mathfuncs = mathmap.initfunc(&inv);

        // closure image with the user values (closure_image_alloc, drawable.c:230-249)
        size_t nuv = m.main->uservals.size();
        std::vector<char> closure_mem(sizeof(mmabi_image_t) + (nuv + 1) * sizeof(mmabi_userval_t), 0);
        mmabi_image_t *closure = (mmabi_image_t *)closure_mem.data();
        closure->type = MMABI_IMAGE_CLOSURE;
        closure->pixel_width = w;
        closure->pixel_height = h;
        closure->v.closure.funcs = &inv.mathfuncs;
        closure->v.closure.num_args = (int)nuv;
        mmabi_input_drawable_t drawable;
        memset(&drawable, 0, sizeof drawable);
        drawable.used = 1;
        drawable.kind = 2;
        drawable.image.type = MMABI_IMAGE_DRAWABLE;
        drawable.image.pixel_width = iw;
        drawable.image.pixel_height = ih;
        drawable.image.v.drawable = &drawable;
        drawable.scale_x = (float)((iw - 1) / 2.0);
        drawable.scale_y = (float)((ih - 1) / 2.0);
        drawable.middle_x = drawable.middle_y = 1.0f;
        std::deque<std::vector<float>> curve_tabs;
        std::deque<std::vector<mmabi_color_t>> grad_tabs;
        std::deque<mmabi_curve_t> curves;
        std::deque<mmabi_gradient_t> grads;
        for (const UservalInfo &u : m.main->uservals) {
            mmabi_userval_t &a = closure->v.closure.args[u.index];
            switch (u.kind) {
                case UvKind::Int: a.v.int_const = u.idef; break;
                case UvKind::Float: a.v.float_const = u.fdef; break;
                case UvKind::Bool: a.v.bool_const = u.bdef; break;
                case UvKind::Color: a.v.color.value = 0x000000ffu; break;
                case UvKind::Image: a.v.image = image ? &drawable.image : nullptr; break;
                // the deterministic tables of mathmap_amd/workloads.py test_curve / test_gradient
                case UvKind::Curve:
                    curve_tabs.emplace_back(1024);
                    for (int i = 0; i < 1024; ++i) curve_tabs.back()[i] = ((float)i * (float)i) / (float)(1023.0 * 1023.0);
                    curves.emplace_back();
                    memset(&curves.back(), 0, sizeof(mmabi_curve_t));
                    curves.back().values = curve_tabs.back().data();
                    a.v.curve = &curves.back();
                    break;
                case UvKind::Gradient:
                    grad_tabs.emplace_back(1024);
                    for (unsigned i = 0; i < 1024; ++i)
                        grad_tabs.back()[i] = ((i >> 2) << 24) | (((1023 - i) >> 2) << 16) | (0x40u << 8) | 0xFFu;
                    grads.emplace_back();
                    grads.back().values = grad_tabs.back().data();
                    a.v.gradient = &grads.back();
                    break;
                default: break;
            }
        }
        inv.uservals = closure->v.closure.args;

        mmabi_frame_t frame;
        memset(&frame, 0, sizeof frame);
        frame.invocation = &inv;
        frame.frame_render_width = w;
        frame.frame_render_height = h;
        frame.current_frame = 0;
        // MMHIP_SELFTEST_WARM_FRAMES=n: render n earlier frames (t - n/16 ... t - 1/16) first, like an
        // animation; from its 2nd frame on the backend runs the user-value-specialised variant
        int warm = 0;
        if (const char *e = getenv("MMHIP_SELFTEST_WARM_FRAMES")) warm = atoi(e);
        if (num_bands < 1) num_bands = 1;
        for (int wf = warm; wf >= 0; --wf) {
        std::fill(rows_finished.begin(), rows_finished.end(), 0);
        frame.current_t = t - (float)wf / 16.0f;
        inv.mathfuncs.init_frame(&frame, closure);
        // call_invocation_parallel (mathmap_common.c:972-1006): contiguous row bands
        for (int b = 0; b < num_bands; ++b) {
            int lo = h * b / num_bands, hi = h * (b + 1) / num_bands;
            mmabi_slice_t slice;
            memset(&slice, 0, sizeof slice);
            slice.frame = &frame;
            slice.region_x = 0;
            slice.region_y = lo;
            slice.region_width = w;
            slice.region_height = hi - lo;
            inv.mathfuncs.init_slice(&slice, closure);
            inv.mathfuncs.calc_lines(&slice, closure, lo, hi, out_rgba + (size_t)lo * inv.row_stride, 0);
        }
        }
        int finished = 0;
        for (int r = 0; r < h; ++r) finished += rows_finished[r];
        unload_hip_code(mathmap.module_info);
        mathmap_hip_set_get_pixel(nullptr);
        // the template indexes rows_finished with the slice-relative row (new_template.c.in:307-308),
        // so with several bands only the first max-band-height entries are set
        int expect = 0;
        for (int b = 0; b < num_bands; ++b) expect = std::max(expect, h * (b + 1) / num_bands - h * b / num_bands);
        if (finished != expect) {
            g_selftest_err = "rows_finished bookkeeping differs from the template's";
            if (error_string[0]) g_selftest_err += std::string(" (the backend reported: ") + error_string + ")";
            return -2;
        }
        return 0;
    } catch (const std::exception &e) {
        g_selftest_err = e.what();
        return -3;
    }
}

}  // extern "C"
