// Typed expression tree -> structured SSA IR, with every non-recursive filter call
// inlined and MathMap closures applied at compile time.
//
// Behavioural spec: compiler.c:1715-1773 (image resize factors), :1875-2227
// (gen_code), :2338-2511 (coordinate / limit / r,a bindings), :2610-2664
// (gen_filter_code).  The reference reaches the same code shape through its
// closure-application + inlining + copy-propagation passes (compiler.c:4702-4764);
// here application happens directly while lowering, because the image operand's
// definition chain is already known in SSA form.
#include <cassert>
#include <cmath>
#include <map>
#include <set>

#include "front.h"
#include "gen.h"
#include "passes.h"

namespace mm {

namespace {

struct Env {
    Filter *filter = nullptr;
    FilterVars *vars = nullptr;
    std::map<std::string, Value *> internals;
    std::map<int, Value *> uservals;
    Env *parent = nullptr;
};

class Lowerer {
   public:
    Lowerer(Module &m, FilterCode &code) : m_(m), code_(code), g_(code) {}
    const std::map<int, Primary> *uv_consts_ = nullptr;
    void run(Filter *f);
    void run_function(Filter *f);
    bool unroll_recursion_ = false;     // user values are literals: unroll recursive applications instead of calling
    std::vector<Filter *> called_;      // filters reached through Rhs::FilterCall (each needs a function body)
    // closure images handed to native filters (render_image's closure branch)
    int render_target_ = -1;        // >= 0: lower "closure #render_target_ applied at (x, y), t = 0" as the result
    int closure_counter_ = 0;       // closures numbered in the order their first native use is lowered
    bool target_done_ = false, in_target_body_ = false;
    // the target closure's value: four variables set to 0 at the top of the body and assigned where the closure meets its
    // native filter -- inside a conditional the construct's exit phis carry the value (or the 0) out to the top level
    CompVar *target_var_[4] = {nullptr, nullptr, nullptr, nullptr};
    void native_image_argument(CompVar *image, bool stripped);

   private:
    Module &m_;
    FilterCode &code_;
    Gen g_;
    Env *env_ = nullptr;
    std::vector<Filter *> inlining_;
    int while_depth_ = 0;
    enum { MAX_RECURSION = 64 };   // activations of one filter on the inlining stack

    bool needs_xy_scaling(unsigned flags) const {
        return (flags & (IMAGE_FLAG_UNIT | IMAGE_FLAG_SQUARE)) != IMAGE_FLAG_UNIT;
    }

    Value *internal_value(const std::string &name, bool allow_bindings);
    void bind_internal(const std::string &name, Rhs rhs, Ty ty = Ty::Int);
    Value *resize_image_if_necessary(Primary image, unsigned flags);
    void gen_limit_bindings();
    void gen_xy_bindings(Value *x, Value *y);
    void gen_ra_bindings();
    void gen_filter(Filter *f, const std::vector<Primary> *args, CompVar *result[4]);

    void gen(AstNode *n, CompVar **dest, bool alloced);
    std::vector<CompVar *> gen_new(AstNode *n) {
        std::vector<CompVar *> d(n->result.len, nullptr);
        gen(n, d.data(), false);
        return d;
    }
    void gen_func(AstNode *n, CompVar **dest, bool alloced);
    void gen_closure(AstNode *n, CompVar **dest, bool alloced);
    bool single_const(AstNode *n, int *iv);
    bool const_value(const Value *v, Primary *out, int depth = 0);
    // tree vectors (dynamic tuple subscripts), lowered to element variables + select chains
    std::set<const AstNode *> vector_selects_;
    void find_vector_variables(AstNode *n);
    CompVar *gen_tree_vector(AstNode *tree, CompVar **dest, bool alloced);
    void to_float(CompVar *dst, CompVar *src) { g_.assign_op(dst, "INT2FLOAT", {g_.P(src)}); }
    void alloc_var(Variable *v);
    void reset_vars(FilterVars *fv);
};

Value *Lowerer::internal_value(const std::string &name, bool allow_bindings) {
    if (allow_bindings) {
        auto it = env_->internals.find(name);
        if (it != env_->internals.end()) return it->second;
    }
    CompVar *t = g_.temp();
    return g_.assign(t, Rhs::Int(name));
}

void Lowerer::bind_internal(const std::string &name, Rhs rhs, Ty ty) {
    CompVar *t = g_.temp(ty);
    env_->internals[name] = g_.assign(t, std::move(rhs));
}

// compiler.c:1715-1773
Value *Lowerer::resize_image_if_necessary(Primary image, unsigned flags) {
    CompVar *resized = g_.temp(Ty::Image);
    if (!needs_xy_scaling(flags)) return g_.assign(resized, Rhs::P(image));
    CompVar *pw = g_.temp(), *ph = g_.temp(), *xf = g_.temp(), *yf = g_.temp();
    g_.assign_op(pw, "IMAGE_PIXEL_WIDTH", {image});
    g_.assign_op(ph, "IMAGE_PIXEL_HEIGHT", {image});
    if (flags == 0) {
        g_.assign_op(xf, "DIV", {Primary::I(2), g_.P(pw)});
        g_.assign_op(yf, "DIV", {Primary::I(2), g_.P(ph)});
    } else {
        CompVar *mx = g_.temp();
        g_.assign_op(mx, "MAX", {g_.P(pw), g_.P(ph)});
        g_.assign_op(xf, "DIV", {g_.P(mx), g_.P(pw)});
        g_.assign_op(yf, "DIV", {g_.P(mx), g_.P(ph)});
    }
    g_.assign_op(resized, "STRIP_RESIZE", {image});
    return g_.assign_op(resized, "RESIZE_IMAGE", {g_.P(resized), g_.P(xf), g_.P(yf)});
}

// compiler.c:2338-2403
void Lowerer::gen_limit_bindings() {
    unsigned fl = env_->filter->flags & (IMAGE_FLAG_UNIT | IMAGE_FLAG_SQUARE);
    if (fl == 0) {
        bind_internal("W", Rhs::V(internal_value("__canvasPixelW", false)));
        bind_internal("H", Rhs::V(internal_value("__canvasPixelH", false)));
        for (auto p : {std::pair<const char *, const char *>{"X", "__canvasPixelW"}, {"Y", "__canvasPixelH"}}) {
            CompVar *m1 = g_.temp();
            g_.assign_op(m1, "SUB", {Primary::V(internal_value(p.second, true)), Primary::I(1)});
            bind_internal(p.first, Rhs::O(g_.op("DIV", 2), {g_.P(m1), Primary::I(2)}));
        }
    } else if (fl == IMAGE_FLAG_UNIT) {
        bind_internal("W", Rhs::I(2));
        bind_internal("H", Rhs::I(2));
        bind_internal("X", Rhs::I(1));
        bind_internal("Y", Rhs::I(1));
    } else {
        CompVar *mx = g_.temp();
        g_.assign_op(mx, "MAX", {Primary::V(internal_value("__canvasPixelW", false)),
                                 Primary::V(internal_value("__canvasPixelH", false))});
        Value *mxv = mx->current;
        bind_internal("X", Rhs::O(g_.op("DIV", 2), {Primary::V(internal_value("__canvasPixelW", false)), Primary::V(mxv)}));
        Value *xv = env_->internals["X"];
        bind_internal("Y", Rhs::O(g_.op("DIV", 2), {Primary::V(internal_value("__canvasPixelH", false)), Primary::V(mxv)}));
        Value *yv = env_->internals["Y"];
        bind_internal("W", Rhs::O(g_.op("MUL", 2), {Primary::V(xv), Primary::I(2)}));
        bind_internal("H", Rhs::O(g_.op("MUL", 2), {Primary::V(yv), Primary::I(2)}));
    }
}

// compiler.c:2405-2420
void Lowerer::gen_xy_bindings(Value *x, Value *y) {
    bind_internal("x", Rhs::O(g_.op("MUL", 2), {Primary::V(x), Primary::V(internal_value("X", true))}));
    bind_internal("y", Rhs::O(g_.op("MUL", 2), {Primary::V(y), Primary::V(internal_value("Y", true))}));
}

// compiler.c:2467-2511
void Lowerer::gen_ra_bindings() {
    CompVar *r = g_.temp(Ty::Float), *a = g_.temp(Ty::Float), *xr = g_.temp(Ty::Float);
    Value *x = internal_value("x", true);
    Value *y = internal_value("y", true);
    g_.assign_op(r, "hypot", {Primary::V(x), Primary::V(y)});
    g_.start_if(Rhs::O(g_.op("EQ", 2), {g_.P(r), Primary::F(0.0f)}));
    g_.assign(a, Rhs::F(0.0f));
    g_.switch_branch();
    g_.assign_op(xr, "DIV", {Primary::V(x), g_.P(r)});
    g_.assign_op(a, "acos", {g_.P(xr)});
    g_.end_if();
    g_.start_if(Rhs::O(g_.op("LESS", 2), {Primary::V(y), Primary::F(0.0f)}));
    g_.assign_op(a, "SUB", {Primary::F((float)(2 * M_PI)), g_.P(a)});
    g_.switch_branch();
    g_.end_if();
    bind_internal("r", Rhs::V(r->current), Ty::Float);
    bind_internal("a", Rhs::V(a->current), Ty::Float);
}

void Lowerer::reset_vars(FilterVars *fv) {
    for (auto &v : fv->vars) v->compvar.assign(v->type.len, nullptr);
}

// compiler.c:1775-1790 alloc_var_compvars_if_needed
void Lowerer::alloc_var(Variable *v) {
    if (v->is_vector) {       // one compvar holding the whole vector (make_tree_vector_variable, compiler.c:303-316)
        if (!v->compvar[0]) {
            v->compvar[0] = code_.new_var(Ty::TreeVector, v->name, 0);
            v->compvar[0]->tuple_len = v->type.len;
        }
        return;
    }
    for (int i = 0; i < v->type.len; ++i)
        if (!v->compvar[i]) {
            bool is_image = v->type.tag == m_.tags.image && v->type.len == 1;
            v->compvar[i] = code_.new_var(is_image ? Ty::Image : Ty::Int, v->name, i);
        }
}

// compiler.c:2610-2664.  `args` = closure arguments followed by x, y, t for an
// inlined call; nullptr for the main filter.
void Lowerer::gen_filter(Filter *f, const std::vector<Primary> *args, CompVar *result[4]) {
    // A recursive application stays a run-time call of filter_$name (gen_func; compiler.c:2165-2222, backends/cc.c:189)
    // unless the user values are baked in: then it is unrolled here -- an `if' whose condition folds to a literal
    // (const_value) only lowers the taken branch -- which gives a kernel without calls.  An unrolling that does
    // not end is reported to lower_filter, which lowers again with calls.
    int active = 0;
    for (Filter *h : inlining_) active += h == f;
    if (active >= MAX_RECURSION) {
        CompileError e("recursive filter `" + f->name + "': the recursion does not end for compile-time-constant arguments");
        e.recursion_limit = true;
        throw e;
    }
    inlining_.push_back(f);
    Env env;
    env.filter = f;
    env.vars = m_.vars[f].get();
    env.parent = env_;
    // variables are per-activation: save the callee's compvars (it may be active
    // further up the inlining stack under a different name) and start fresh
    std::vector<std::vector<CompVar *>> saved;
    for (auto &v : env.vars->vars) saved.push_back(v->compvar);
    reset_vars(env.vars);
    env_ = &env;

    find_vector_variables(f->body);
    gen_limit_bindings();
    if (args) {
        int n = (int)args->size();
        int nuv = (int)f->uservals.size();
        assert(n == nuv + 3);
        for (int i = 0; i < nuv; ++i) {
            const UservalInfo &u = f->uservals[i];
            Ty ty = u.kind == UvKind::Float ? Ty::Float : u.kind == UvKind::Color ? Ty::Color
                  : u.kind == UvKind::Curve ? Ty::Curve : u.kind == UvKind::Gradient ? Ty::Gradient
                  : u.kind == UvKind::Image ? Ty::Image : Ty::Int;
            CompVar *bv = g_.temp(ty);
            if (u.kind == UvKind::Image)
                env.uservals[i] = g_.assign(bv, Rhs::V(resize_image_if_necessary((*args)[i], u.image_flags)));
            else
                env.uservals[i] = g_.assign(bv, Rhs::P((*args)[i]));
        }
        CompVar *xt = g_.temp(), *yt = g_.temp();
        g_.assign(xt, Rhs::P((*args)[n - 3]));
        g_.assign(yt, Rhs::P((*args)[n - 2]));
        gen_xy_bindings(xt->current, yt->current);
        bind_internal("t", Rhs::P((*args)[n - 1]));
    } else {
        static const char *getters[] = {"USERVAL_INT_ACCESS", "USERVAL_FLOAT_ACCESS", "USERVAL_BOOL_ACCESS",
                                        "USERVAL_COLOR_ACCESS", "USERVAL_CURVE_ACCESS", "USERVAL_GRADIENT_ACCESS",
                                        "USERVAL_IMAGE_ACCESS"};
        for (const UservalInfo &u : f->uservals) {
            Ty ty = u.kind == UvKind::Float ? Ty::Float : u.kind == UvKind::Color ? Ty::Color
                  : u.kind == UvKind::Curve ? Ty::Curve : u.kind == UvKind::Gradient ? Ty::Gradient
                  : u.kind == UvKind::Image ? Ty::Image : Ty::Int;
            if (u.kind == UvKind::Image) {
                CompVar *img = g_.temp(Ty::Image);
                g_.assign_op(img, "USERVAL_IMAGE_ACCESS", {Primary::I(u.index)});
                Value *rz = resize_image_if_necessary(g_.P(img), u.image_flags);
                CompVar *bv = g_.temp(ty);
                env.uservals[u.index] = g_.assign(bv, Rhs::V(rz));
            } else {
                CompVar *bv = g_.temp(ty);
                auto cit = uv_consts_ ? uv_consts_->find(u.index) : std::map<int, Primary>::const_iterator();
                if (uv_consts_ && cit != uv_consts_->end() && (u.kind == UvKind::Int || u.kind == UvKind::Float || u.kind == UvKind::Bool))
                    env.uservals[u.index] = g_.assign(bv, Rhs::P(cit->second));
                else
                    env.uservals[u.index] = g_.assign_op(bv, getters[(int)u.kind], {Primary::I(u.index)});
            }
        }
        if (needs_xy_scaling(f->flags)) gen_xy_bindings(internal_value("x", false), internal_value("y", false));
    }
    if (f->uses_ra) gen_ra_bindings();

    gen(f->body, result, false);

    env_ = env.parent;
    size_t k = 0;
    for (auto &v : env.vars->vars) {
        if (k < saved.size()) v->compvar = saved[k];
        ++k;
    }
    inlining_.pop_back();
}

// exprtree.c:1424-1469
bool Lowerer::single_const(AstNode *n, int *iv) {
    switch (n->kind) {
        case AstNode::IntConst: *iv = n->ival; return true;
        case AstNode::FloatConst: *iv = (int)n->fval; return true;
        case AstNode::Tuple:
            if (n->result.len == 1) return single_const(n->kids[0], iv);
            return false;
        case AstNode::Func:
            if (n->entry->name == "__neg" && single_const(n->kids[0], iv)) { *iv = -*iv; return true; }
            return false;
        default: return false;
    }
}

// compiler.c:2521-2600 (find_all_vector_variables): a variable that is subscripted with a
// non-constant index anywhere in the filter is a "tree vector" everywhere in it: one TYPE_TREE_VECTOR
// compvar, read with TREE_VECTOR_NTH, written with SET_TREE_VECTOR_NTH (a new vector value per write),
// built from element values by RHS_TREE_VECTOR.  The reference stores it as a persistent tree of floats
// (tree_vectors.c); its length is static, so the kernels keep it as `len` floats in registers.
void Lowerer::find_vector_variables(AstNode *n) {
    if (!n) return;
    for (AstNode *k : n->kids) find_vector_variables(k);
    for (AstNode *k : n->subs) find_vector_variables(k);
    int dummy;
    if (n->kind == AstNode::Select) {
        for (AstNode *sub : n->subs)
            if (!single_const(sub, &dummy)) {
                vector_selects_.insert(n);
                if (n->kids[0]->kind == AstNode::Var) n->kids[0]->var->is_vector = true;
            }
    } else if (n->kind == AstNode::SubAssign) {
        for (AstNode *sub : n->subs)
            if (!single_const(sub, &dummy)) n->var->is_vector = true;
    }
}

// compiler.c:1839-1873 gen_tree_vector: the vector holding the value of `tree`, whose elements go to `dest`
CompVar *Lowerer::gen_tree_vector(AstNode *tree, CompVar **dest, bool alloced) {
    const int len = tree->result.len;
    if (tree->kind == AstNode::Var && tree->var->is_vector) {
        alloc_var(tree->var);
        CompVar *tv = tree->var->compvar[0];
        for (int i = 0; i < len; ++i) {
            if (!alloced) dest[i] = g_.temp(Ty::Float);
            g_.assign_op(dest[i], "TREE_VECTOR_NTH", {Primary::I(i), g_.P(tv)});
        }
        return tv;
    }
    gen(tree, dest, alloced);
    Rhs r;
    r.kind = Rhs::TreeVector;
    for (int i = 0; i < len; ++i) r.args.push_back(g_.P(dest[i]));
    CompVar *tv = g_.temp(Ty::TreeVector);
    tv->tuple_len = len;
    g_.assign(tv, r);
    return tv;
}

// The literal an SSA value is known to hold while lowering (copies and foldable ops of
// literals), used to decide `if's at compile time.  Only asked outside loops: inside one, a
// value defined before the loop may still be rewritten into a loop phi (gen.cpp commit()).
bool Lowerer::const_value(const Value *v, Primary *out, int depth) {
    if (!v || v->index < 0 || depth > 64) return false;
    const Stmt *d = v->def;
    if (!d || d->kind != Stmt::Assign) return false;
    const Rhs &r = d->rhs;
    auto prim = [&](const Primary &p, Primary *o) {
        if (p.kind == Primary::IntConst || p.kind == Primary::FloatConst) { *o = p; return true; }
        if (p.kind == Primary::Val) return const_value(p.value, o, depth + 1);
        return false;
    };
    if (r.kind == Rhs::Prim) return prim(r.prim, out);
    if (r.kind != Rhs::Op) return false;
    std::vector<Primary> cs(r.args.size());
    for (size_t i = 0; i < r.args.size(); ++i)
        if (!prim(r.args[i], &cs[i])) return false;
    return fold_constant_op(r.op, cs, *out);
}

// An image argument of a native filter / of render(): when it is a MathMap closure, the native filter
// renders it first -- render_image's closure branch (builtins.c:267-302) launches the closure's own
// calc_lines over the whole frame with floatmap = 1 at frame 0, t = 0.0.  The closure gets a number;
// lowering the filter again with render_target_ set to that number produces the code of that launch:
// the closure's body inlined at the pixel's raw coordinates, its values as the filter result.
//
// `stripped`: the native filter sees the image without its resize wrapper (native filters proper: their image
// arguments go through STRIP_RESIZE).  render() does not strip: what it gets for a closure made in a filter
// with the default (unit, square) coordinates is an IMAGE_RESIZE of the closure, which render_image treats
// like a drawable -- its else branch (builtins.c:303-343) walks the pixels and evaluates
// ORIG_VAL(fx, fy, image, 0.0) with fx = ((float)x - bx) / ax in float arithmetic, the resize factors
// applied by the macro (opmacros.h:203-207).  Same result type, slightly different coordinates.
void Lowerer::native_image_argument(CompVar *image, bool stripped) {
    ImageChain ch = resolve_image_chain(image->current);
    if (ch.base != ImageChain::MathMapClosure) return;
    Stmt *def = ch.closure_def;
    if (in_target_body_) return;     // (a closure rendered for a native filter may not feed native filters itself)
    if (def->closure_id < 0) def->closure_id = closure_counter_++;
    if (def->closure_id != render_target_ || target_done_) return;
    if (while_depth_ > 0) throw CompileError("a filter closure passed to a native filter inside a loop is not supported");
    std::vector<Primary> cargs = def->rhs.args;
    Primary rx, ry;
    emit_closure_render_coordinates(g_, ch, stripped, &rx, &ry);
    cargs.push_back(rx);
    cargs.push_back(ry);
    cargs.push_back(Primary::F(0.0f));                     // frame 0 / t = 0.0 in both branches
    CompVar *res[4];
    in_target_body_ = true;          // numbering must not depend on which closure is the target
    gen_filter(def->rhs.filter, &cargs, res);
    in_target_body_ = false;
    // Inside a conditional (render_image runs when the native call does: only in the branch taken) the assignment below
    // gets an exit phi per enclosing `if': the closure's value where the branch runs, the initial 0 where it does not --
    // and there the native filter is not called either (its record stays unexecuted), so the map is not even rendered.
    for (int i = 0; i < 4; ++i) g_.assign(target_var_[i], Rhs::V(res[i]->current));
    target_done_ = true;
}

// EXPR_FILTER_CLOSURE, compiler.c:2165-2222
void Lowerer::gen_closure(AstNode *n, CompVar **dest, bool alloced) {
    Filter *callee = n->filter;
    std::vector<Primary> prims;
    for (size_t i = 0; i < n->kids.size(); ++i) {
        std::vector<CompVar *> a = gen_new(n->kids[i]);
        const UservalInfo &u = callee->uservals[i];
        if (u.kind == UvKind::Color) {
            CompVar *c = g_.temp(Ty::Color);
            g_.assign_op(c, "MAKE_COLOR", {g_.P(a[0]), g_.P(a[1]), g_.P(a[2]), g_.P(a[3])});
            prims.push_back(g_.P(c));
        } else if (u.kind == UvKind::Image) {
            if (callee->kind == Filter::Native) native_image_argument(a[0], true);
            CompVar *c = g_.temp(Ty::Image);
            g_.assign_op(c, "STRIP_RESIZE", {g_.P(a[0])});
            prims.push_back(g_.P(c));
        } else
            prims.push_back(g_.P(a[0]));
    }
    CompVar *image = g_.temp(Ty::Image);
    Rhs r;
    r.kind = Rhs::Closure;
    r.filter = callee;
    r.args = prims;
    g_.assign(image, r);
    Value *rz = resize_image_if_necessary(g_.P(image), env_->filter->flags);
    if (!alloced) dest[0] = g_.temp(Ty::Image);
    g_.assign(dest[0], Rhs::V(rz));
}

void Lowerer::gen_func(AstNode *n, CompVar **dest, bool alloced) {
    std::vector<std::vector<CompVar *>> args;
    std::vector<TInfo> types;
    for (AstNode *k : n->kids) {
        args.push_back(gen_new(k));
        types.push_back(k->result);
    }
    std::vector<CompVar *> result(n->result.len);
    for (int i = 0; i < n->result.len; ++i) {
        if (!alloced) {
            bool is_image = n->result.tag == m_.tags.image && n->result.len == 1;
            dest[i] = g_.temp(is_image ? Ty::Image : Ty::Int);
        }
        result[i] = dest[i];
    }
    // Application of a MathMap closure: inline the callee at the sampled position.
    if (n->entry->id == "origValXY") {
        ImageChain chain = resolve_image_chain(args[2][0]->current);
        if (chain.base == ImageChain::MathMapClosure) {
            CompVar *x = args[0][0], *y = args[0][1];
            for (auto &fac : chain.factors) {   // compopt/resize.c:29-98
                CompVar *nx = g_.temp(), *ny = g_.temp();
                g_.assign_op(nx, "MUL", {g_.P(x), fac.first});
                g_.assign_op(ny, "MUL", {g_.P(y), fac.second});
                x = nx;
                y = ny;
            }
            std::vector<Primary> cargs = chain.closure_def->rhs.args;
            cargs.push_back(g_.P(x));
            cargs.push_back(g_.P(y));
            cargs.push_back(g_.P(args[1][0]));
            Filter *callee = chain.closure_def->rhs.filter;
            bool recursive = false;
            for (Filter *h : inlining_) recursive = recursive || h == callee;
            if (recursive && !unroll_recursion_) {
                // compiler.c:4219-4237 can_inline: a filter that is already being inlined is not inlined again --
                // the application stays a call of filter_$name (RHS_FILTER), evaluated at run time
                Rhs call;
                call.kind = Rhs::FilterCall;
                call.filter = callee;
                call.args = cargs;
                CompVar *tv = g_.temp(Ty::Tuple);
                tv->tuple_len = 4;
                g_.assign(tv, call);
                for (int i = 0; i < 4; ++i) g_.assign_op(result[i], "TUPLE_NTH", {g_.P(tv), Primary::I(i)});
                bool known = false;
                for (Filter *c : called_) known = known || c == callee;
                if (!known) called_.push_back(callee);
                return;
            }
            CompVar *res[4];
            gen_filter(callee, &cargs, res);
            for (int i = 0; i < 4; ++i) g_.copy(result[i], res[i]);
            return;
        }
    }
    if (n->entry->id == "render") native_image_argument(args[0][0], false);
    GenScope scope(g_);
    n->entry->gen(g_, args, types, result);
}

void Lowerer::gen(AstNode *n, CompVar **dest, bool alloced) {
    switch (n->kind) {
        case AstNode::IntConst:
            if (!alloced) dest[0] = g_.temp();
            g_.assign(dest[0], Rhs::I(n->ival));
            break;
        case AstNode::FloatConst:
            if (!alloced) dest[0] = g_.temp(Ty::Float);
            g_.assign(dest[0], Rhs::F(n->fval));
            break;
        case AstNode::Tuple:
            for (size_t i = 0; i < n->kids.size(); ++i) gen(n->kids[i], dest + i, alloced);
            break;
        case AstNode::Select: {      // compiler.c:1911-1958
            int len = n->kids[0]->result.len;
            std::vector<CompVar *> temps(len, nullptr);
            CompVar *tv = nullptr;
            if (vector_selects_.count(n)) tv = gen_tree_vector(n->kids[0], temps.data(), false);
            else temps = gen_new(n->kids[0]);
            for (size_t i = 0; i < n->subs.size(); ++i) {
                int sub;
                if (!single_const(n->subs[i], &sub)) {
                    std::vector<CompVar *> sv = gen_new(n->subs[i]);
                    if (!alloced) dest[i] = g_.temp();
                    g_.assign_op(dest[i], "TREE_VECTOR_NTH", {g_.P(sv[0]), g_.P(tv)});
                    continue;
                }
                if (sub < 0) sub = 0;
                if (sub >= len) sub = len - 1;
                if (!alloced) dest[i] = temps[sub];
                else g_.copy(dest[i], temps[sub]);
            }
            break;
        }
        case AstNode::Var:
            alloc_var(n->var);
            if (n->var->is_vector) {      // compiler.c:1962-1970
                for (int i = 0; i < n->var->type.len; ++i) {
                    if (!alloced) dest[i] = g_.temp();
                    g_.assign_op(dest[i], "TREE_VECTOR_NTH", {Primary::I(i), g_.P(n->var->compvar[0])});
                }
                break;
            }
            for (int i = 0; i < n->var->type.len; ++i) {
                if (!alloced) dest[i] = n->var->compvar[i];
                else g_.copy(dest[i], n->var->compvar[i]);
            }
            break;
        case AstNode::Internal: {
            if (!alloced) dest[0] = g_.temp();
            if (in_target_body_ && n->name == "frame") {      // calc_lines_$name of a closure image runs on a frame made by
                g_.assign(dest[0], Rhs::I(0));                // invocation_new_frame(invocation, image, 0, 0.0): frame = 0
                break;
            }
            auto it = env_->internals.find(n->name);
            if (it != env_->internals.end()) g_.assign(dest[0], Rhs::V(it->second));
            else g_.assign(dest[0], Rhs::Int(n->name));
            break;
        }
        case AstNode::Assign:
            alloc_var(n->var);
            if (n->var->is_vector) {      // compiler.c:1995-1999
                CompVar *tv = gen_tree_vector(n->kids[0], dest, alloced);
                g_.copy(n->var->compvar[0], tv);
                break;
            }
            gen(n->kids[0], n->var->compvar.data(), true);
            for (int i = 0; i < n->result.len; ++i) {
                if (alloced) g_.copy(dest[i], n->var->compvar[i]);
                else dest[i] = n->var->compvar[i];
            }
            break;
        case AstNode::SubAssign: {
            alloc_var(n->var);
            std::vector<CompVar *> temps = gen_new(n->kids[0]);
            int len = n->var->type.len;
            for (size_t i = 0; i < n->subs.size(); ++i) {
                int sub;
                if (n->var->is_vector) {      // SET_TREE_VECTOR_NTH, also for constant subscripts (compiler.c:2027-2038)
                    std::vector<CompVar *> sv = gen_new(n->subs[i]);
                    CompVar *tv = n->var->compvar[0];
                    g_.assign_op(tv, "SET_TREE_VECTOR_NTH", {g_.P(sv[0]), g_.P(tv), g_.P(temps[i])});
                    if (alloced) g_.copy(dest[i], temps[i]);
                    else dest[i] = temps[i];
                    continue;
                }
                if (!single_const(n->subs[i], &sub))
                    throw CompileError("internal: dynamic subscript on a variable that was not marked as a vector", n->pos);
                if (sub < 0) sub = 0;
                if (sub >= len) sub = len - 1;
                g_.copy(n->var->compvar[sub], temps[i]);
                if (alloced) g_.copy(dest[i], temps[i]);
                else dest[i] = temps[i];
            }
            break;
        }
        case AstNode::Cast: gen(n->kids[0], dest, alloced); break;
        case AstNode::Func: gen_func(n, dest, alloced); break;
        case AstNode::Seq: {
            gen_new(n->kids[0]);
            gen(n->kids[1], dest, alloced);
            break;
        }
        case AstNode::IfThen:
        case AstNode::IfThenElse: {
            bool is_image = n->result.tag == m_.tags.image && n->result.len == 1;
            std::vector<CompVar *> result(n->result.len);
            for (auto &r : result) r = g_.temp(is_image ? Ty::Image : Ty::Int);
            std::vector<CompVar *> cond = gen_new(n->kids[0]);
            Primary known;
            if (while_depth_ == 0 && const_value(cond[0]->current, &known)) {
                // decided at compile time: lower the taken branch only (this is what lets a
                // recursive filter whose depth is a literal / baked-in user value terminate)
                const bool truth = known.kind == Primary::IntConst ? known.i != 0 : known.f != 0.0f;
                if (truth) gen(n->kids[1], result.data(), true);
                else if (n->kind == AstNode::IfThenElse) gen(n->kids[2], result.data(), true);
            } else {
                g_.start_if(Rhs::V(cond[0]->current));
                gen(n->kids[1], result.data(), true);
                g_.switch_branch();
                if (n->kind == AstNode::IfThenElse) gen(n->kids[2], result.data(), true);
                g_.end_if();
            }
            for (int i = 0; i < n->result.len; ++i) {
                if (alloced) g_.copy(dest[i], result[i]);
                else dest[i] = result[i];
            }
            break;
        }
        case AstNode::While:
        case AstNode::DoWhile: {
            CompVar *inv = g_.temp();
            if (n->kind == AstNode::DoWhile) gen_new(n->kids[1]);
            gen(n->kids[0], &inv, true);
            g_.start_while(inv);
            ++while_depth_;
            gen_new(n->kids[1]);
            gen(n->kids[0], &inv, true);
            --while_depth_;
            g_.end_while();
            if (!alloced) dest[0] = g_.temp();
            g_.assign(dest[0], Rhs::I(0));
            break;
        }
        case AstNode::Userval: {
            if (n->uv->kind == UvKind::Color) {
                // colours are deconstructed into four floats (compiler.c:1824-1837, :2141-2150)
                auto it = env_->uservals.find(n->uv->index);
                CompVar *c = g_.temp(Ty::Color);
                if (it != env_->uservals.end()) g_.assign(c, Rhs::V(it->second));
                else g_.assign_op(c, "USERVAL_COLOR_ACCESS", {Primary::I(n->uv->index)});
                static const char *parts[] = {"RED_FLOAT", "GREEN_FLOAT", "BLUE_FLOAT", "ALPHA_FLOAT"};
                for (int i = 0; i < 4; ++i) {
                    if (!alloced) dest[i] = g_.temp(Ty::Float);
                    g_.assign_op(dest[i], parts[i], {g_.P(c)});
                }
            } else {
                auto it = env_->uservals.find(n->uv->index);
                if (it == env_->uservals.end()) throw CompileError("internal: unbound user value " + n->uv->name, n->pos);
                if (!alloced) dest[0] = g_.temp(it->second->var->type);
                g_.assign(dest[0], Rhs::V(it->second));
            }
            break;
        }
        case AstNode::Closure: gen_closure(n, dest, alloced); break;
    }
}

// filter_$name (new_template.c.in:375-422): the filter as a function of its closure arguments and (x, y, t).  The
// arguments are read with the USERVAL_*_ACCESS operators -- inside a function body they index the call's
// argument block instead of the invocation's user values -- and x, y, t are the internals of that name, which are
// the function's parameters there.  Same code path as an inlined application (gen_filter with arguments).
void Lowerer::run_function(Filter *f) {
    static const char *getters[] = {"USERVAL_INT_ACCESS", "USERVAL_FLOAT_ACCESS", "USERVAL_BOOL_ACCESS", "USERVAL_COLOR_ACCESS",
                                    "USERVAL_CURVE_ACCESS", "USERVAL_GRADIENT_ACCESS", "USERVAL_IMAGE_ACCESS"};
    Env env;                       // internal_value needs an environment before gen_filter makes the callee's
    env.filter = f;
    env_ = &env;
    std::vector<Primary> cargs;
    for (const UservalInfo &u : f->uservals) {
        Ty ty = u.kind == UvKind::Float ? Ty::Float : u.kind == UvKind::Color ? Ty::Color
              : u.kind == UvKind::Curve ? Ty::Curve : u.kind == UvKind::Gradient ? Ty::Gradient
              : u.kind == UvKind::Image ? Ty::Image : Ty::Int;
        CompVar *a = g_.temp(ty);
        g_.assign_op(a, getters[(int)u.kind], {Primary::I(u.index)});
        cargs.push_back(g_.P(a));
    }
    cargs.push_back(Primary::V(internal_value("x", false)));
    cargs.push_back(Primary::V(internal_value("y", false)));
    cargs.push_back(Primary::V(internal_value("t", false)));
    env_ = nullptr;
    CompVar *res[4];
    gen_filter(f, &cargs, res);
    for (int i = 0; i < 4; ++i) code_.result[i] = res[i]->current;
}

void Lowerer::run(Filter *f) {
    CompVar *res[4];
    if (render_target_ >= 0)
        for (int i = 0; i < 4; ++i) {
            target_var_[i] = g_.temp(Ty::Float);
            g_.assign(target_var_[i], Rhs::F(0.0f));
        }
    gen_filter(f, nullptr, res);
    for (int i = 0; i < 4; ++i) code_.result[i] = res[i]->current;
    if (render_target_ >= 0) {
        if (!target_done_) throw CompileError("internal: closure render target not reached");
        for (int i = 0; i < 4; ++i) code_.result[i] = target_var_[i]->current;
    }
}

}  // namespace

static std::unique_ptr<FilterCode> lower_filter_impl(Module &m, Filter *f, const std::map<int, Primary> *uv_consts, bool unroll) {
    if (f->kind != Filter::MathMap) throw CompileError("cannot lower a native filter");
    std::unique_ptr<FilterCode> code(new FilterCode());
    code->filter = f;
    Lowerer l(m, *code);
    l.uv_consts_ = uv_consts;
    l.unroll_recursion_ = unroll;
    l.run(f);
    propagate_types(*code);
    for (int k = 0; k < l.closure_counter_; ++k) {
        std::unique_ptr<FilterCode> sub(new FilterCode());
        sub->filter = f;
        Lowerer ls(m, *sub);
        ls.uv_consts_ = uv_consts;
        ls.unroll_recursion_ = unroll;
        ls.render_target_ = k;
        ls.run(f);
        propagate_types(*sub);
        code->closure_renders.push_back(std::move(sub));
        for (Filter *c : ls.called_) {
            bool known = false;
            for (Filter *k2 : l.called_) known = known || k2 == c;
            if (!known) l.called_.push_back(c);
        }
    }
    // function bodies of the filters called at run time, and of those they call
    std::vector<Filter *> work = l.called_;
    for (size_t i = 0; i < work.size(); ++i) {
        std::unique_ptr<FilterCode> fn(new FilterCode());
        fn->filter = work[i];
        Lowerer lf(m, *fn);
        lf.run_function(work[i]);
        propagate_types(*fn);
        for (Filter *c : lf.called_) {
            bool known = false;
            for (Filter *k2 : work) known = known || k2 == c;
            if (!known) work.push_back(c);
        }
        code->functions.push_back(std::move(fn));
    }
    return code;
}

std::unique_ptr<FilterCode> lower_function(Module &m, Filter *f) {
    if (f->kind != Filter::MathMap) throw CompileError("cannot lower a native filter");
    std::unique_ptr<FilterCode> fn(new FilterCode());
    fn->filter = f;
    Lowerer lf(m, *fn);
    lf.run_function(f);
    propagate_types(*fn);
    return fn;
}

std::unique_ptr<FilterCode> lower_filter(Module &m, Filter *f, const std::map<int, Primary> *uv_consts) {
    if (uv_consts) {
        try {
            return lower_filter_impl(m, f, uv_consts, true);
        } catch (const CompileError &e) {
            if (!e.recursion_limit) throw;
        }
    }
    return lower_filter_impl(m, f, uv_consts, false);
}

}  // namespace mm
