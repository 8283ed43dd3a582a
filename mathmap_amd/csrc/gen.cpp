// SSA builder: assignment commit, if/while phi bookkeeping.  See gen.h.
#include "gen.h"

#include <cassert>

namespace mm {

const OpInfo *Gen::op(const char *cname, int nargs) const {
    const OpInfo *o = op_by_cname(cname, nargs);
    if (!o) throw CompileError(std::string("internal: unknown op ") + cname);
    return o;
}

void Gen::emit(Stmt *s) {
    s->parent = parents_.empty() ? nullptr : parents_.back();
    blocks_.back()->push_back(s);
}

void Gen::rewrite_rhs(Rhs &r, Value *from, Value *to) {
    if (r.kind == Rhs::Prim && r.prim.kind == Primary::Val && r.prim.value == from) r.prim.value = to;
    for (Primary &p : r.args)
        if (p.kind == Primary::Val && p.value == from) p.value = to;
}

void Gen::rewrite_block(Block &b, Value *from, Value *to) {
    for (Stmt *s : b) {
        switch (s->kind) {
            case Stmt::Assign: rewrite_rhs(s->rhs, from, to); break;
            case Stmt::Phi: rewrite_rhs(s->rhs, from, to); rewrite_rhs(s->rhs2, from, to); break;
            case Stmt::If:
                rewrite_rhs(s->cond, from, to);
                rewrite_block(s->then_, from, to);
                rewrite_block(s->else_, from, to);
                rewrite_block(s->phis, from, to);
                break;
            case Stmt::While:
                // the entry operand (rhs) of a nested loop's phi is evaluated in *our* scope
                rewrite_block(s->phis, from, to);
                rewrite_rhs(s->cond, from, to);
                rewrite_block(s->body, from, to);
                break;
            default: break;
        }
    }
}

// Registers the new value `nv` of compvar `c` with the innermost open construct
// and makes it current.
void Gen::commit(CompVar *c, Value *nv) {
    Value *old = c->current;
    if (!frames_.empty()) {
        Frame &f = frames_.back();
        auto it = f.phi_of.find(c);
        Stmt *phi = it == f.phi_of.end() ? nullptr : it->second;
        if (!f.is_while) {
            if (!phi) {
                phi = code.new_stmt(Stmt::Phi);
                phi->lhs = code.new_value(c);
                phi->lhs->def = phi;
                phi->rhs = Rhs::V(old);
                phi->rhs2 = Rhs::V(old);
                phi->old_value = old;
                phi->parent = f.stmt;
                f.phis.push_back(phi);
                f.phi_of[c] = phi;
            }
            (f.branch == 0 ? phi->rhs : phi->rhs2) = Rhs::V(nv);
        } else {
            if (!phi) {
                phi = code.new_stmt(Stmt::Phi);
                phi->lhs = code.new_value(c);
                phi->lhs->def = phi;
                phi->rhs = Rhs::V(old);
                phi->old_value = old;
                phi->parent = f.stmt;
                f.phis.push_back(phi);
                f.phi_of[c] = phi;
                f.stmt->phis.push_back(phi);
                // every use of the pre-loop value inside the loop now reads the phi
                rewrite_rhs(f.stmt->cond, old, phi->lhs);
                rewrite_block(f.stmt->body, old, phi->lhs);
                for (Stmt *other : f.stmt->phis)
                    if (other != phi) rewrite_rhs(other->rhs2, old, phi->lhs);
            }
            phi->rhs2 = Rhs::V(nv);
        }
    }
    c->current = nv;
}

Value *Gen::assign(CompVar *dst, Rhs rhs) {
    Stmt *s = code.new_stmt(Stmt::Assign);
    s->rhs = std::move(rhs);
    Value *nv = code.new_value(dst);
    s->lhs = nv;
    nv->def = s;
    emit(s);
    commit(dst, nv);
    return nv;
}

Value *Gen::assign_op(CompVar *dst, const char *cname, std::vector<Primary> args) {
    const OpInfo *o = op(cname, (int)args.size());
    return assign(dst, Rhs::O(o, std::move(args)));
}

void Gen::start_if(Rhs cond) {
    Stmt *s = code.new_stmt(Stmt::If);
    s->cond = std::move(cond);
    emit(s);
    frames_.push_back(Frame{s, false, 0, {}, {}});
    parents_.push_back(s);
    blocks_.push_back(&s->then_);
}

void Gen::reenter_if(Stmt *s, int branch) {
    assert(s->kind == Stmt::If);
    frames_.push_back(Frame{s, false, branch, {}, {}});
    parents_.push_back(s);
    blocks_.push_back(branch == 0 ? &s->then_ : &s->else_);
}

void Gen::switch_branch() {
    Frame &f = frames_.back();
    assert(!f.is_while && f.branch == 0);
    for (Stmt *phi : f.phis) phi->lhs->var->current = phi->old_value;
    f.branch = 1;
    blocks_.pop_back();
    blocks_.push_back(&f.stmt->else_);
}

void Gen::end_if() {
    Frame f = std::move(frames_.back());
    frames_.pop_back();
    parents_.pop_back();
    blocks_.pop_back();
    for (Stmt *phi : f.phis) phi->lhs->var->current = phi->old_value;
    for (Stmt *phi : f.phis) {
        f.stmt->phis.push_back(phi);
        commit(phi->lhs->var, phi->lhs);
    }
}

void Gen::start_while(CompVar *invariant) {
    Stmt *s = code.new_stmt(Stmt::While);
    s->cond = Rhs::V(invariant->current);
    emit(s);
    frames_.push_back(Frame{s, true, 0, {}, {}});
    parents_.push_back(s);
    blocks_.push_back(&s->body);
}

void Gen::end_while() {
    Frame f = std::move(frames_.back());
    frames_.pop_back();
    parents_.pop_back();
    blocks_.pop_back();
    for (Stmt *phi : f.phis) phi->lhs->var->current = phi->old_value;
    for (Stmt *phi : f.phis) commit(phi->lhs->var, phi->lhs);
}

CompVar *emit_cond(Gen &g, const Cond &c) {
    switch (c.kind) {
        case Cond::Cmp: {
            CompVar *t = g.temp();
            g.assign_op(t, c.cname, {g.P(c.a.v), g.P(c.b.v)});
            return t;
        }
        case Cond::And: {   // c = A; if (c) c = B;
            CompVar *t = emit_cond(g, c.sub[0]);
            g.start_if(Rhs::V(t->current));
            CompVar *b = emit_cond(g, c.sub[1]);
            g.copy(t, b);
            g.switch_branch();
            g.end_if();
            return t;
        }
        case Cond::Or: {    // c = A; if (c) {} else c = B;
            CompVar *t = emit_cond(g, c.sub[0]);
            g.start_if(Rhs::V(t->current));
            g.switch_branch();
            CompVar *b = emit_cond(g, c.sub[1]);
            g.copy(t, b);
            g.end_if();
            return t;
        }
        case Cond::Not: {
            CompVar *t = emit_cond(g, c.sub[0]);
            g.assign_op(t, "NOT", {g.P(t)});
            return t;
        }
    }
    return nullptr;
}

void gen_if(const Cond &c, const std::function<void()> &then_, const std::function<void()> &else_) {
    Gen &g = *GenScope::cur();
    CompVar *t = emit_cond(g, c);
    g.start_if(Rhs::V(t->current));
    if (then_) then_();
    g.switch_branch();
    if (else_) else_();
    g.end_if();
}

ImageChain resolve_image_chain(Value *v) {
    ImageChain c;
    bool stripped = false;
    for (int guard = 0; guard < 1000 && v; ++guard) {
        Stmt *d = v->def;
        if (!d || d->kind != Stmt::Assign) { c.base = v->index < 0 ? ImageChain::Unknown : ImageChain::Runtime; return c; }
        const Rhs &r = d->rhs;
        if (r.kind == Rhs::Prim && r.prim.kind == Primary::Val) { v = r.prim.value; continue; }
        if (r.kind == Rhs::Op && std::string(r.op->cname) == "STRIP_RESIZE" && r.args[0].kind == Primary::Val) {
            stripped = true;
            v = r.args[0].value;
            continue;
        }
        if (r.kind == Rhs::Op && std::string(r.op->cname) == "RESIZE_IMAGE" && r.args[0].kind == Primary::Val) {
            if (!stripped) c.factors.push_back({r.args[1], r.args[2]});
            stripped = false;   // a strip only removes the one resize directly below it
            v = r.args[0].value;
            continue;
        }
        if (r.kind == Rhs::Closure && r.filter->kind == Filter::MathMap) {
            c.base = ImageChain::MathMapClosure;
            c.closure_def = d;
            return c;
        }
        c.base = ImageChain::Runtime;
        return c;
    }
    c.base = ImageChain::Runtime;
    return c;
}

void emit_closure_render_coordinates(Gen &g, const ImageChain &ch, bool raw, Primary *xo, Primary *yo) {
    if (raw || ch.factors.empty()) {
        CompVar *x = g.temp(), *y = g.temp();
        g.assign(x, Rhs::Int("x"));
        g.assign(y, Rhs::Int("y"));
        *xo = g.P(x);
        *yo = g.P(y);
        return;
    }
    // floatmap.c:39-41: ax = bx = (float)(w - 1) / 2.0, by = (float)(h - 1) / 2.0, ay = by * -1.0
    CompVar *w = g.temp(), *h = g.temp(), *w1 = g.temp(), *h1 = g.temp(), *ax = g.temp(), *by = g.temp(), *ay = g.temp();
    CompVar *cf = g.temp(), *rf = g.temp(), *dx = g.temp(), *dy = g.temp(), *fx = g.temp(), *fy = g.temp();
    g.assign(w, Rhs::Int("__renderPixelW"));
    g.assign(h, Rhs::Int("__renderPixelH"));
    g.assign_op(w1, "SUB", {g.P(w), Primary::I(1)});
    g.assign_op(h1, "SUB", {g.P(h), Primary::I(1)});
    g.assign_op(ax, "DIV", {g.P(w1), Primary::I(2)});
    g.assign_op(by, "DIV", {g.P(h1), Primary::I(2)});
    g.assign_op(ay, "NEG", {g.P(by)});
    g.assign(cf, Rhs::Int("__colF"));                // (float)column of the pixel in the frame
    g.assign(rf, Rhs::Int("__rowF"));
    g.assign_op(dx, "SUB", {g.P(cf), g.P(ax)});
    g.assign_op(dy, "SUB", {g.P(rf), g.P(by)});
    g.assign_op(fx, "DIV", {g.P(dx), g.P(ax)});
    g.assign_op(fy, "DIV", {g.P(dy), g.P(ay)});
    CompVar *x = fx, *y = fy;
    for (auto &fac : ch.factors) {                    // opmacros.h:203-207, one wrapper per resize
        CompVar *nx = g.temp(), *ny = g.temp();
        g.assign_op(nx, "MUL", {g.P(x), fac.first});
        g.assign_op(ny, "MUL", {g.P(y), fac.second});
        x = nx;
        y = ny;
    }
    *xo = g.P(x);
    *yo = g.P(y);
}

}  // namespace mm
