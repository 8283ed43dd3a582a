// User-value specialisation: constant propagation over the structured SSA.
//
// When a filter is invoked, its scalar user values are known.  The specialising JIT lowers
// the filter with those values as literals and then applies, to what are now literal
// constants, exactly the folds the reference's own optimiser applies to literals
// (constant folding through the op macros, compiler.c:3383-3458, and simplify_ops,
// compiler.c:3460-3535: x+0, 0+x, x-0 -> x; x*1, 1*x, x/1 -> x; x*0, 0*x -> 0), followed by
// dead-branch removal (compiler.c:3537-3700).  Constants are propagated *optimistically*
// through loop phis (sparse-conditional-constant style lattice: Top > Const > Bottom), so a
// component that starts at 0 and is only ever multiplied/added with zeros (e.g. the j,k parts
// of the quaternion Mandelbrot for cj = ck = pj = pk = 0) disappears from the kernel.
//
// Like the reference's literal folds, x*0 -> 0 ignores NaN/Inf in x; the generic
// (unspecialised) kernel remains the default and the reference for parity tests.
#include <cmath>
#include <cstring>
#include <functional>
#include <map>

#include "passes.h"

namespace mm {

namespace {

struct Lat {
    enum K { Top, Const, Bottom } k = Top;
    Primary c;
};

bool same_const(const Primary &a, const Primary &b) {
    if (a.kind != b.kind) return false;
    if (a.kind == Primary::IntConst) return a.i == b.i;
    if (a.kind == Primary::FloatConst) return std::memcmp(&a.f, &b.f, 4) == 0;
    return false;
}

bool is_zero(const Primary &p) {
    return (p.kind == Primary::IntConst && p.i == 0) || (p.kind == Primary::FloatConst && p.f == 0.0f);
}
bool is_one(const Primary &p) {
    return (p.kind == Primary::IntConst && p.i == 1) || (p.kind == Primary::FloatConst && p.f == 1.0f);
}
double as_double(const Primary &p) { return p.kind == Primary::IntConst ? (double)p.i : (double)p.f; }

// C assignment conversion of a constant to the type of the variable it is stored in
bool convert_to(Ty t, Primary &p) {
    if (t == Ty::Int) {
        if (p.kind == Primary::FloatConst) {
            if (!(std::fabs(p.f) < 2e9f)) return false;
            p = Primary::I((int)p.f);
        }
        return p.kind == Primary::IntConst;
    }
    if (t == Ty::Float) {
        if (p.kind == Primary::IntConst) p = Primary::F((float)p.i);
        return p.kind == Primary::FloatConst;
    }
    return false;
}

bool both_int(const Primary &a, const Primary &b) { return a.kind == Primary::IntConst && b.kind == Primary::IntConst; }

// folds an operator on constants with the C semantics of its macro (opmacros.h)
bool fold_impl(const OpInfo *op, const std::vector<Primary> &a, Primary &out) {
    const char *n = op->cname;
    auto bin = [&](auto fi, auto fd) {
        if (both_int(a[0], a[1])) out = Primary::I(fi(a[0].i, a[1].i));
        else out = Primary::F((float)fd(as_double(a[0]), as_double(a[1])));   // double op, one rounding (innocuous for + - *)
        return true;
    };
    if (!strcmp(n, "ADD")) return bin([](int x, int y) { return (int)((unsigned)x + (unsigned)y); }, [](double x, double y) { return x + y; });
    if (!strcmp(n, "SUB")) return bin([](int x, int y) { return (int)((unsigned)x - (unsigned)y); }, [](double x, double y) { return x - y; });
    if (!strcmp(n, "MUL")) return bin([](int x, int y) { return (int)((unsigned)x * (unsigned)y); }, [](double x, double y) { return x * y; });
    if (!strcmp(n, "NEG")) {
        out = a[0].kind == Primary::IntConst ? Primary::I((int)(0u - (unsigned)a[0].i)) : Primary::F(-a[0].f);
        return true;
    }
    if (!strcmp(n, "DIV")) { out = Primary::F((float)as_double(a[0]) / (float)as_double(a[1])); return true; }
    auto cmp = [&](auto f) { out = Primary::I(both_int(a[0], a[1]) ? f((double)a[0].i, (double)a[1].i) : f(as_double(a[0]), as_double(a[1]))); return true; };
    if (!strcmp(n, "EQ")) return cmp([](double x, double y) { return x == y ? 1 : 0; });
    if (!strcmp(n, "LESS")) return cmp([](double x, double y) { return x < y ? 1 : 0; });
    if (!strcmp(n, "LEQ")) return cmp([](double x, double y) { return x <= y ? 1 : 0; });
    if (!strcmp(n, "NOT") && a[0].kind == Primary::IntConst) { out = Primary::I(!a[0].i); return true; }
    if (!strcmp(n, "MIN") || !strcmp(n, "MAX")) {
        bool pick_first = !strcmp(n, "MIN") ? as_double(a[0]) < as_double(a[1]) : !(as_double(a[0]) < as_double(a[1]));
        const Primary &p = pick_first ? a[0] : a[1];
        out = both_int(a[0], a[1]) ? p : Primary::F((float)as_double(p));
        return true;
    }
    if (!strcmp(n, "INT2FLOAT") && a[0].kind == Primary::IntConst) { out = Primary::F((float)a[0].i); return true; }
    return false;   // libm and everything else stays in the kernel / prologue
}

struct Sccp {
    static bool fold(const OpInfo *op, const std::vector<Primary> &a, Primary &out) { return fold_impl(op, a, out); }

    FilterCode &code;
    std::map<const Value *, Lat> lat;
    bool changed = false;

    explicit Sccp(FilterCode &c) : code(c) {}

    Lat of(const Primary &p) {
        Lat l;
        if (p.kind == Primary::IntConst || p.kind == Primary::FloatConst) { l.k = Lat::Const; l.c = p; return l; }
        if (p.kind != Primary::Val) { l.k = Lat::Bottom; return l; }
        const Value *v = p.value;
        if (v->index < 0) {   // uninitialised reads as 0
            if (v->var->type == Ty::Int) { l.k = Lat::Const; l.c = Primary::I(0); return l; }
            if (v->var->type == Ty::Float) { l.k = Lat::Const; l.c = Primary::F(0.0f); return l; }
            l.k = Lat::Bottom;
            return l;
        }
        auto it = lat.find(v);
        return it == lat.end() ? Lat() : it->second;
    }

    Lat eval(const Rhs &r, const CompVar *lhs) {
        Lat l;
        switch (r.kind) {
            case Rhs::Prim: l = of(r.prim); break;
            case Rhs::Op: {
                std::vector<Lat> as;
                bool any_top = false, all_const = true;
                for (const Primary &p : r.args) {
                    as.push_back(of(p));
                    any_top |= as.back().k == Lat::Top;
                    all_const &= as.back().k == Lat::Const;
                }
                const char *n = r.op->cname;
                if (!strcmp(n, "MUL") && ((as[0].k == Lat::Const && is_zero(as[0].c)) || (as[1].k == Lat::Const && is_zero(as[1].c)))) {
                    l.k = Lat::Const;
                    l.c = Primary::I(0);
                    break;
                }
                if (any_top) { l.k = Lat::Top; return l; }
                if (all_const && r.op->pure) {
                    std::vector<Primary> cs;
                    for (Lat &x : as) cs.push_back(x.c);
                    Primary out;
                    if (fold(r.op, cs, out)) { l.k = Lat::Const; l.c = out; break; }
                }
                l.k = Lat::Bottom;
                break;
            }
            default: l.k = Lat::Bottom;
        }
        if (l.k == Lat::Const && !convert_to(lhs->type, l.c)) l.k = Lat::Bottom;
        return l;
    }

    static Lat meet(const Lat &a, const Lat &b) {
        if (a.k == Lat::Top) return b;
        if (b.k == Lat::Top) return a;
        if (a.k == Lat::Const && b.k == Lat::Const && same_const(a.c, b.c)) return a;
        Lat l;
        l.k = Lat::Bottom;
        return l;
    }

    void set(const Value *v, const Lat &n) {
        Lat &cur = lat[v];
        Lat m = meet(cur, n);          // monotone descent
        if (cur.k == Lat::Top && n.k == Lat::Top) return;
        if (m.k != cur.k || (m.k == Lat::Const && !same_const(m.c, cur.c))) { cur = m; changed = true; }
    }

    void walk(Block &b) {
        for (Stmt *s : b) {
            switch (s->kind) {
                case Stmt::Assign: set(s->lhs, eval(s->rhs, s->lhs->var)); break;
                case Stmt::If:
                    walk(s->then_);
                    walk(s->else_);
                    for (Stmt *p : s->phis) set(p->lhs, meet(eval(p->rhs, p->lhs->var), eval(p->rhs2, p->lhs->var)));
                    break;
                case Stmt::While:
                    for (Stmt *p : s->phis) set(p->lhs, meet(eval(p->rhs, p->lhs->var), eval(p->rhs2, p->lhs->var)));
                    walk(s->body);
                    for (Stmt *p : s->phis) set(p->lhs, meet(eval(p->rhs, p->lhs->var), eval(p->rhs2, p->lhs->var)));
                    break;
                default: break;
            }
        }
    }

    void run() {
        do {
            changed = false;
            walk(code.body);
        } while (changed);
    }

    // ---- rewriting ------------------------------------------------------------------------
    void subst(Primary &p) {
        if (p.kind != Primary::Val || p.value->index < 0) return;
        auto it = lat.find(p.value);
        if (it != lat.end() && it->second.k == Lat::Const) p = it->second.c;
    }
    void subst(Rhs &r) {
        if (r.kind == Rhs::Prim) subst(r.prim);
        for (Primary &p : r.args) subst(p);
    }

    // simplify_ops identities on the remaining operators
    static void simplify(Rhs &r) {
        if (r.kind != Rhs::Op) return;
        const char *n = r.op->cname;
        auto is_c = [&](int i) { return r.args[i].is_const(); };
        if (!strcmp(n, "ADD")) {
            if (is_c(0) && is_zero(r.args[0])) r = Rhs::P(r.args[1]);
            else if (is_c(1) && is_zero(r.args[1])) r = Rhs::P(r.args[0]);
        } else if (!strcmp(n, "SUB")) {
            if (is_c(1) && is_zero(r.args[1])) r = Rhs::P(r.args[0]);
        } else if (!strcmp(n, "MUL")) {
            if (is_c(0) && is_one(r.args[0])) r = Rhs::P(r.args[1]);
            else if (is_c(1) && is_one(r.args[1])) r = Rhs::P(r.args[0]);
        } else if (!strcmp(n, "DIV")) {
            if (is_c(1) && is_one(r.args[1]) && r.args[0].type() == Ty::Float) r = Rhs::P(r.args[0]);
        }
    }

    static bool truthy(const Primary &p, bool *known) {
        *known = p.kind == Primary::IntConst || p.kind == Primary::FloatConst;
        return p.kind == Primary::IntConst ? p.i != 0 : p.f != 0.0f;
    }

    void rewrite(Block &b) {
        Block out;
        for (Stmt *s : b) {
            switch (s->kind) {
                case Stmt::Assign: {
                    auto it = lat.find(s->lhs);
                    if (it != lat.end() && it->second.k == Lat::Const) s->rhs = Rhs::P(it->second.c);
                    else { subst(s->rhs); simplify(s->rhs); }
                    out.push_back(s);
                    break;
                }
                case Stmt::If: {
                    subst(s->cond);
                    rewrite(s->then_);
                    rewrite(s->else_);
                    for (Stmt *p : s->phis) { subst(p->rhs); subst(p->rhs2); }
                    bool known = false;
                    bool t = s->cond.kind == Rhs::Prim ? truthy(s->cond.prim, &known) : false;
                    if (known) {   // dead branch: splice the live one, phis become plain copies
                        Block &live = t ? s->then_ : s->else_;
                        for (Stmt *x : live) { x->parent = s->parent; out.push_back(x); }
                        for (Stmt *p : s->phis) {
                            Stmt *a = code.new_stmt(Stmt::Assign);
                            a->lhs = p->lhs;
                            a->lhs->def = a;
                            a->rhs = t ? p->rhs : p->rhs2;
                            a->parent = s->parent;
                            out.push_back(a);
                        }
                    } else
                        out.push_back(s);
                    break;
                }
                case Stmt::While: {
                    for (Stmt *p : s->phis) { subst(p->rhs); subst(p->rhs2); }
                    // a loop whose condition is the constant 0 never runs
                    bool known = false, t = true;
                    if (s->cond.kind == Rhs::Prim && s->cond.prim.kind == Primary::Val) {
                        auto it = lat.find(s->cond.prim.value);
                        if (it != lat.end() && it->second.k == Lat::Const) t = truthy(it->second.c, &known);
                    }
                    if (known && !t) {
                        for (Stmt *p : s->phis) {
                            Stmt *a = code.new_stmt(Stmt::Assign);
                            a->lhs = p->lhs;
                            a->lhs->def = a;
                            a->rhs = p->rhs;
                            a->parent = s->parent;
                            out.push_back(a);
                        }
                        break;
                    }
                    subst(s->cond);
                    rewrite(s->body);
                    // phis proven constant turn into assignments in front of the loop
                    Block keep;
                    for (Stmt *p : s->phis) {
                        auto it = lat.find(p->lhs);
                        if (it != lat.end() && it->second.k == Lat::Const) {
                            Stmt *a = code.new_stmt(Stmt::Assign);
                            a->lhs = p->lhs;
                            a->lhs->def = a;
                            a->rhs = Rhs::P(it->second.c);
                            a->parent = s->parent;
                            out.push_back(a);
                        } else
                            keep.push_back(p);
                    }
                    s->phis.swap(keep);
                    out.push_back(s);
                    break;
                }
                default: break;
            }
        }
        b.swap(out);
    }
};

}  // namespace

// Runs constant propagation + simplification + dead-branch removal to a fixpoint.
void specialize_constants(FilterCode &code) {
    propagate_types(code);
    for (int round = 0; round < 8; ++round) {
        Sccp s(code);
        s.run();
        s.rewrite(code.body);
        for (int i = 0; i < 4; ++i) {
            // results must stay values: materialise a constant result
            auto it = s.lat.find(code.result[i]);
            (void)it;
        }
        bool c = copy_propagate(code);
        c |= eliminate_dead_code(code);
        if (!c && round > 0) break;
    }
}

bool fold_constant_op(const OpInfo *op, const std::vector<Primary> &args, Primary &out) { return fold_impl(op, args, out); }

}  // namespace mm
