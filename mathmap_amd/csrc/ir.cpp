// IR support: operator table, type propagation, JSON dump.
#include "ir.h"

#include <cassert>
#include <cstring>
#include <sstream>

namespace mm {

const char *ty_name(Ty t) {
    switch (t) {
        case Ty::Nil: return "nil";
        case Ty::Int: return "int";
        case Ty::Float: return "float";
        case Ty::Complex: return "complex";
        case Ty::Color: return "color";
        case Ty::Curve: return "curve";
        case Ty::Gradient: return "gradient";
        case Ty::Image: return "image";
        case Ty::Tuple: return "tuple";
        case Ty::TreeVector: return "tree_vector";
    }
    return "?";
}

// ---------------------------------------------------------------------------
// Operator table.  Restates ops.lisp:112-253 (the clisp-generated opdefs.h is
// not available anywhere, so the table is rebuilt here by hand).
// ---------------------------------------------------------------------------
namespace {

std::vector<OpInfo> build_ops() {
    std::vector<OpInfo> v;
    auto add = [&](const char *key, const char *cname, int nargs, TypeProp prop, Ty type,
                   std::vector<Ty> args, bool pure = true, bool foldable = true, int tuple_len = 0) {
        OpInfo o{};
        o.index = (int)v.size();
        o.key = key;
        o.cname = cname;
        o.nargs = nargs;
        o.prop = prop;
        o.type = type;
        for (int i = 0; i < 9; ++i) o.arg_types[i] = Ty::Float;
        assert((int)args.size() == nargs || args.size() == 1 || args.empty());
        for (int i = 0; i < nargs; ++i)
            o.arg_types[i] = args.empty() ? Ty::Float : (args.size() == 1 ? args[0] : args[i]);
        o.pure = pure;
        o.foldable = pure && foldable;
        o.tuple_len = tuple_len;
        v.push_back(o);
    };
    const Ty I = Ty::Int, F = Ty::Float, C = Ty::Complex, K = Ty::Color, IM = Ty::Image, T = Ty::Tuple;
    const TypeProp CO = TypeProp::Const, MX = TypeProp::Max, MF = TypeProp::MaxFloat;

    add("nop", "NOP", 0, CO, I, {});
    add("int-to-float", "INT2FLOAT", 1, CO, F, {I});
    add("float-to-int", "FLOAT2INT", 1, CO, I, {F});
    add("int-to-complex", "INT2COMPLEX", 1, CO, C, {I});
    add("float-to-complex", "FLOAT2COMPLEX", 1, CO, C, {F});

    add("+", "ADD", 2, MX, Ty::Nil, {});
    add("-", "SUB", 2, MX, Ty::Nil, {});
    add("neg", "NEG", 1, MX, Ty::Nil, {});
    add("*", "MUL", 2, MX, Ty::Nil, {});
    add("/", "DIV", 2, CO, F, {});
    add("%", "MOD", 2, CO, F, {});

    add("abs", "fabs", 1, MF, Ty::Nil, {});
    add("min", "MIN", 2, MF, Ty::Nil, {});
    add("max", "MAX", 2, MF, Ty::Nil, {});

    for (const char *n : {"sqrt", "sin", "cos", "tan", "asin", "acos", "atan", "exp", "log", "sinh", "cosh",
                          "tanh", "asinh", "acosh", "atanh"})
        add(n, n, 1, CO, F, {});
    add("hypot", "hypot", 2, CO, F, {});
    add("atan2", "atan2", 2, CO, F, {});
    add("pow", "pow", 2, CO, F, {});
    add("gamma", "GAMMA", 1, CO, F, {});
    add("beta", "gsl_sf_beta", 2, CO, F, {});

    add("floor", "floor", 1, CO, I, {});
    add("ceil", "ceil", 1, CO, I, {});
    add("=", "EQ", 2, CO, I, {});
    add("<", "LESS", 2, CO, I, {});
    add("<=", "LEQ", 2, CO, I, {});
    add("not", "NOT", 1, CO, I, {I});

    add("print", "PRINT_FLOAT", 1, CO, I, {}, false);
    add("newline", "NEWLINE", 0, CO, I, {}, false);
    add("start-debug-tuple", "START_DEBUG_TUPLE", 1, CO, I, {I}, false);
    add("set-debug-tuple-data", "SET_DEBUG_TUPLE_DATA", 2, CO, I, {I, F}, false);

    add("apply-curve", "APPLY_CURVE", 2, CO, F, {Ty::Curve, F}, true, false);
    add("apply-gradient", "APPLY_GRADIENT", 2, CO, T, {Ty::Gradient, F}, true, false, 4);
    add("orig-val", "ORIG_VAL", 4, CO, T, {F, F, IM, F}, true, false, 4);
    add("resize-image", "RESIZE_IMAGE", 3, CO, IM, {IM, F, F}, true, false);
    add("strip-resize", "STRIP_RESIZE", 1, CO, IM, {IM}, true, false);
    add("render", "RENDER", 3, CO, IM, {IM, I, I}, true, false);
    add("image-pixel-width", "IMAGE_PIXEL_WIDTH", 1, CO, I, {IM}, true, false);
    add("image-pixel-height", "IMAGE_PIXEL_HEIGHT", 1, CO, I, {IM}, true, false);

    add("make-rgba-color", "MAKE_COLOR", 4, CO, K, {F, F, F, F}, true, false);
    add("red", "RED_FLOAT", 1, CO, F, {K}, true, false);
    add("green", "GREEN_FLOAT", 1, CO, F, {K}, true, false);
    add("blue", "BLUE_FLOAT", 1, CO, F, {K}, true, false);
    add("alpha", "ALPHA_FLOAT", 1, CO, F, {K}, true, false);

    add("tuple-nth", "TUPLE_NTH", 2, CO, F, {T, I}, true, false);
    add("tree-vector-nth", "TREE_VECTOR_NTH", 2, CO, F, {I, Ty::TreeVector}, true, false);
    add("set-tree-vector-nth", "SET_TREE_VECTOR_NTH", 3, CO, Ty::TreeVector, {I, Ty::TreeVector, F}, true, false);

    add("complex", "COMPLEX", 2, CO, C, {});
    add("c-real", "crealf", 1, CO, F, {C});
    add("c-imag", "cimagf", 1, CO, F, {C});
    for (auto p : {std::pair<const char *, const char *>{"c-sqrt", "csqrtf"}, {"c-sin", "csinf"}, {"c-cos", "ccosf"},
                   {"c-tan", "ctanf"}, {"c-asin", "casinf"}, {"c-acos", "cacosf"}, {"c-atan", "catanf"}})
        add(p.first, p.second, 1, CO, C, {C});
    add("c-pow", "cpowf", 2, CO, C, {C});
    add("c-exp", "cexpf", 1, CO, C, {C});
    add("c-log", "clogf", 1, CO, C, {C});
    add("c-arg", "cargf", 1, CO, F, {C});
    for (auto p : {std::pair<const char *, const char *>{"c-sinh", "csinhf"}, {"c-cosh", "ccoshf"},
                   {"c-tanh", "ctanhf"}, {"c-asinh", "casinhf"}, {"c-acosh", "cacoshf"}, {"c-atanh", "catanhf"},
                   {"c-gamma", "cgamma"}})
        add(p.first, p.second, 1, CO, C, {C});

    add("ell-int-k-comp", "ELL_INT_K_COMP", 1, CO, F, {});
    add("ell-int-e-comp", "ELL_INT_E_COMP", 1, CO, F, {});
    add("ell-int-f", "ELL_INT_F", 2, CO, F, {});
    add("ell-int-e", "ELL_INT_E", 2, CO, F, {});
    add("ell-int-p", "ELL_INT_P", 3, CO, F, {});
    add("ell-int-d", "ELL_INT_D", 3, CO, F, {});
    add("ell-int-rc", "ELL_INT_RC", 2, CO, F, {});
    add("ell-int-rd", "ELL_INT_RD", 3, CO, F, {});
    add("ell-int-rf", "ELL_INT_RF", 3, CO, F, {});
    add("ell-int-rj", "ELL_INT_RJ", 4, CO, F, {});
    add("ell-jac", "ELL_JAC", 2, CO, T, {}, true, false, 3);

    add("solve-linear-2", "SOLVE_LINEAR_2", 2, CO, T, {T, T}, true, false, 2);
    add("solve-linear-3", "SOLVE_LINEAR_3", 2, CO, T, {T, T}, true, false, 3);
    add("solve-poly-2", "SOLVE_POLY_2", 3, CO, T, {}, true, false, 2);
    add("solve-poly-3", "SOLVE_POLY_3", 4, CO, T, {}, true, false, 3);

    add("rand", "RAND", 2, CO, F, {}, false);

    add("libnoise-perlin", "libnoise_perlin", 6, CO, F, {I, F, F, F, F, F});
    add("libnoise-billow", "libnoise_billow", 6, CO, F, {I, F, F, F, F, F});
    add("libnoise-ridged-multi", "libnoise_ridged_multi", 5, CO, F, {I, F, F, F, F});
    add("libnoise-voronoi", "libnoise_voronoi", 4, CO, F, {F, F, F, F});

    add("userval-int", "USERVAL_INT_ACCESS", 1, CO, I, {I}, true, false);
    add("userval-float", "USERVAL_FLOAT_ACCESS", 1, CO, F, {I}, true, false);
    add("userval-bool", "USERVAL_BOOL_ACCESS", 1, CO, I, {I}, true, false);
    add("userval-color", "USERVAL_COLOR_ACCESS", 1, CO, K, {I}, true, false);
    add("userval-curve", "USERVAL_CURVE_ACCESS", 1, CO, Ty::Curve, {I}, true, false);
    add("userval-gradient", "USERVAL_GRADIENT_ACCESS", 1, CO, Ty::Gradient, {I}, true, false);
    add("userval-image", "USERVAL_IMAGE_ACCESS", 1, CO, IM, {I}, true, false);

    add("output-tuple", "OUTPUT_TUPLE", 1, CO, I, {T}, false);
    return v;
}

const std::vector<OpInfo> &ops() {
    static const std::vector<OpInfo> table = build_ops();
    return table;
}

}  // namespace

const std::vector<OpInfo> &all_ops() { return ops(); }

const OpInfo *op_by_cname(const char *cname, int nargs) {
    for (const OpInfo &o : ops())
        if (o.nargs == nargs && std::strcmp(o.cname, cname) == 0) return &o;
    return nullptr;
}

const OpInfo *op_by_key(const char *key, int nargs) {
    for (const OpInfo &o : ops())
        if (o.nargs == nargs && std::strcmp(o.key, key) == 0) return &o;
    return nullptr;
}

// ---------------------------------------------------------------------------

Ty Primary::type() const {
    switch (kind) {
        case Val: return value->var->type;
        case IntConst: return Ty::Int;
        case FloatConst: return Ty::Float;
        case ComplexConst: return Ty::Complex;
        case ColorConst: return Ty::Color;
        default: return Ty::Nil;
    }
}

// compiler.c:2752-2801
Ty Rhs::type() const {
    switch (kind) {
        case Prim: return prim.type();
        case Internal: return Ty::Float;
        case Op: {
            if (op->prop == TypeProp::Const) return op->type;
            int mx = (int)Ty::Int;
            for (const Primary &p : args) mx = std::max(mx, (int)p.type());
            return (Ty)mx;
        }
        case Tuple:
        case FilterCall: return Ty::Tuple;
        case Closure: return Ty::Image;
        case TreeVector: return Ty::TreeVector;
        default: return Ty::Nil;
    }
}

CompVar *FilterCode::new_var(Ty t, const std::string &name, int elem) {
    vars.emplace_back();
    CompVar *v = &vars.back();
    v->id = next_var++;
    v->type = t;
    v->is_temp = name.empty();
    v->name = name;
    v->elem = elem;
    // every compvar starts with an "uninitialised" value (index -1)
    values.emplace_back();
    Value *u = &values.back();
    u->var = v;
    u->index = -1;
    u->gid = next_val++;
    v->current = u;
    return v;
}

Value *FilterCode::new_value(CompVar *v) {
    values.emplace_back();
    Value *val = &values.back();
    val->var = v;
    val->index = (int)v->values.size();
    val->gid = next_val++;
    v->values.push_back(val);
    return val;
}

Stmt *FilterCode::new_stmt(Stmt::Kind k) {
    stmts.emplace_back();
    Stmt *s = &stmts.back();
    s->kind = k;
    return s;
}

// Type propagation: a compvar's type is the maximum over everything assigned to
// any of its values (compiler.c:2811-2865).  Iterate to a fixpoint.
static bool propagate_block(Block &b) {
    bool changed = false;
    for (Stmt *s : b) {
        switch (s->kind) {
            case Stmt::Assign:
            case Stmt::Phi: {
                Ty t = s->rhs.type();
                if (s->kind == Stmt::Phi) {
                    Ty t2 = s->rhs2.type();
                    if ((int)t2 > (int)t) t = t2;
                }
                if ((int)t > (int)s->lhs->var->type) {
                    s->lhs->var->type = t;
                    changed = true;
                }
                if (t == Ty::TreeVector && s->lhs->var->tuple_len == 0) {
                    // the length of a tree vector is static: it comes from the RHS_TREE_VECTOR that built it and
                    // travels through copies, phis and SET_TREE_VECTOR_NTH
                    auto len_of = [](const Rhs &r) {
                        if (r.kind == Rhs::TreeVector) return (int)r.args.size();
                        if (r.kind == Rhs::Prim && r.prim.kind == Primary::Val) return r.prim.value->var->tuple_len;
                        if (r.kind == Rhs::Op && r.args.size() == 3 && r.args[1].kind == Primary::Val) return r.args[1].value->var->tuple_len;
                        return 0;
                    };
                    int len = len_of(s->rhs);
                    if (!len && s->kind == Stmt::Phi) len = len_of(s->rhs2);
                    if (len) { s->lhs->var->tuple_len = len; changed = true; }
                }
                if (t == Ty::Tuple && s->lhs->var->tuple_len == 0) {
                    int len = 0;
                    if (s->rhs.kind == Rhs::Op) len = s->rhs.op->tuple_len;
                    else if (s->rhs.kind == Rhs::Tuple) len = (int)s->rhs.args.size();
                    else if (s->rhs.kind == Rhs::Prim && s->rhs.prim.kind == Primary::Val)
                        len = s->rhs.prim.value->var->tuple_len;
                    else if (s->rhs.kind == Rhs::FilterCall) len = 4;
                    if (len) { s->lhs->var->tuple_len = len; changed = true; }
                }
                break;
            }
            case Stmt::If:
                changed |= propagate_block(s->then_);
                changed |= propagate_block(s->else_);
                changed |= propagate_block(s->phis);
                break;
            case Stmt::While:
                changed |= propagate_block(s->phis);
                changed |= propagate_block(s->body);
                break;
            default: break;
        }
    }
    return changed;
}

void propagate_types(FilterCode &code) {
    while (propagate_block(code.body)) {
    }
}

// ---------------------------------------------------------------------------
// JSON dump (consumed by oracle/ccgen.py and by tests).
// ---------------------------------------------------------------------------
namespace {

void js_str(std::ostringstream &o, const std::string &s) {
    o << '"';
    for (char c : s) {
        if (c == '"' || c == '\\') o << '\\' << c;
        else if (c == '\n') o << "\\n";
        else o << c;
    }
    o << '"';
}

void js_prim(std::ostringstream &o, const Primary &p) {
    switch (p.kind) {
        case Primary::Val: o << "[\"v\"," << p.value->var->id << "," << p.value->index << "]"; break;
        case Primary::IntConst: o << "[\"i\"," << p.i << "]"; break;
        case Primary::FloatConst: {
            uint32_t bits;
            std::memcpy(&bits, &p.f, 4);
            o << "[\"f\"," << bits << "]";
            break;
        }
        case Primary::ComplexConst: {
            uint32_t b1, b2;
            std::memcpy(&b1, &p.f, 4);
            std::memcpy(&b2, &p.f2, 4);
            o << "[\"c\"," << b1 << "," << b2 << "]";
            break;
        }
        case Primary::ColorConst: o << "[\"k\"," << p.color << "]"; break;
        default: o << "null";
    }
}

void js_rhs(std::ostringstream &o, const Rhs &r) {
    switch (r.kind) {
        case Rhs::Prim: o << "{\"k\":\"prim\",\"p\":"; js_prim(o, r.prim); o << "}"; break;
        case Rhs::Internal: o << "{\"k\":\"internal\",\"name\":"; js_str(o, r.internal); o << "}"; break;
        case Rhs::Op:
            o << "{\"k\":\"op\",\"op\":"; js_str(o, r.op->cname); o << ",\"args\":[";
            for (size_t i = 0; i < r.args.size(); ++i) { if (i) o << ","; js_prim(o, r.args[i]); }
            o << "]}";
            break;
        case Rhs::Tuple:
        case Rhs::TreeVector:
            o << "{\"k\":\"" << (r.kind == Rhs::Tuple ? "tuple" : "treevector") << "\",\"args\":[";
            for (size_t i = 0; i < r.args.size(); ++i) { if (i) o << ","; js_prim(o, r.args[i]); }
            o << "]}";
            break;
        case Rhs::Closure:
        case Rhs::FilterCall:
            o << "{\"k\":\"" << (r.kind == Rhs::Closure ? "closure" : "filtercall") << "\",\"filter\":";
            js_str(o, r.filter->name);
            o << ",\"native\":"; js_str(o, r.filter->kind == Filter::Native ? r.filter->native_func : "");
            o << ",\"args\":[";
            for (size_t i = 0; i < r.args.size(); ++i) { if (i) o << ","; js_prim(o, r.args[i]); }
            o << "]}";
            break;
        default: o << "null";
    }
}

void js_block(std::ostringstream &o, const Block &b);

void js_phis(std::ostringstream &o, const Block &b) {
    o << "[";
    bool first = true;
    for (const Stmt *s : b) {
        if (s->kind != Stmt::Phi) continue;
        if (!first) o << ",";
        first = false;
        o << "{\"lhs\":[" << s->lhs->var->id << "," << s->lhs->index << "],\"rhs\":";
        js_rhs(o, s->rhs);
        o << ",\"rhs2\":";
        js_rhs(o, s->rhs2);
        o << ",\"hoisted\":" << (s->hoisted ? 1 : 0) << ",\"pixel\":" << (s->in_pixel ? 1 : 0) << "}";
    }
    o << "]";
}

void js_block(std::ostringstream &o, const Block &b) {
    o << "[";
    bool first = true;
    for (const Stmt *s : b) {
        if (s->kind == Stmt::Nil) continue;
        if (!first) o << ",";
        first = false;
        switch (s->kind) {
            case Stmt::Assign:
                o << "{\"k\":\"assign\",\"lhs\":[" << s->lhs->var->id << "," << s->lhs->index << "],\"rhs\":";
                js_rhs(o, s->rhs);
                if (s->closure_id >= 0) o << ",\"cid\":" << s->closure_id;
                break;
            case Stmt::If:
                o << "{\"k\":\"if\",\"cond\":"; js_rhs(o, s->cond);
                o << ",\"then\":"; js_block(o, s->then_);
                o << ",\"else\":"; js_block(o, s->else_);
                o << ",\"phis\":"; js_phis(o, s->phis);
                break;
            case Stmt::While:
                o << "{\"k\":\"while\",\"phis\":"; js_phis(o, s->phis);
                o << ",\"cond\":"; js_rhs(o, s->cond);
                o << ",\"body\":"; js_block(o, s->body);
                break;
            default: o << "{\"k\":\"nil\"";
        }
        o << ",\"hoisted\":" << (s->hoisted ? 1 : 0) << ",\"pixel\":" << (s->in_pixel ? 1 : 0) << "}";
    }
    o << "]";
}

}  // namespace

std::string dump_ir(const FilterCode &code) {
    std::ostringstream o;
    o.precision(9);
    o << "{\"filter\":"; js_str(o, code.filter ? code.filter->name : "");
    o << ",\"flags\":" << (code.filter ? code.filter->flags : 0);
    o << ",\"uservals\":[";
    if (code.filter)
        for (size_t i = 0; i < code.filter->uservals.size(); ++i) {
            const UservalInfo &u = code.filter->uservals[i];
            if (i) o << ",";
            o << "{\"index\":" << u.index << ",\"kind\":" << (int)u.kind << ",\"name\":"; js_str(o, u.name);
            o << ",\"imin\":" << u.imin << ",\"imax\":" << u.imax << ",\"idef\":" << u.idef;
            o << ",\"fmin\":" << u.fmin << ",\"fmax\":" << u.fmax << ",\"fdef\":" << u.fdef;
            o << ",\"bdef\":" << (u.bdef ? 1 : 0) << ",\"image_flags\":" << u.image_flags << "}";
        }
    o << "],\"vars\":[";
    bool first = true;
    for (const CompVar &v : code.vars) {
        if (!first) o << ",";
        first = false;
        o << "{\"id\":" << v.id << ",\"type\":\"" << ty_name(v.type) << "\",\"tuple_len\":" << v.tuple_len
          << ",\"name\":"; js_str(o, v.name); o << ",\"elem\":" << v.elem << "}";
    }
    o << "],\"body\":";
    js_block(o, code.body);
    o << ",\"result\":[";
    for (int i = 0; i < 4; ++i) {
        if (i) o << ",";
        if (code.result[i]) o << "[" << code.result[i]->var->id << "," << code.result[i]->index << "]";
        else o << "null";
    }
    o << "]";
    if (!code.functions.empty()) {
        o << ",\"functions\":[";
        for (size_t i = 0; i < code.functions.size(); ++i) { if (i) o << ","; o << dump_ir(*code.functions[i]); }
        o << "]";
    }
    if (!code.closure_renders.empty()) {
        o << ",\"closure_renders\":[";
        for (size_t i = 0; i < code.closure_renders.size(); ++i) { if (i) o << ","; o << dump_ir(*code.closure_renders[i]); }
        o << "]";
    }
    o << "}";
    return o.str();
}

}  // namespace mm
