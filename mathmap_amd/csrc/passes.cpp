// Clean-up and analysis passes over the structured SSA IR.
//
//  * copy propagation between compvars of equal type, dead-code elimination
//    (the reference runs the same two among its passes, compiler.c:4702-4764;
//    they never change a computed value, only shrink the kernel text)
//  * constness analysis + slicing into a frame-constant prologue and the per-pixel
//    body (reference: analyze_constants compiler.c:3208-3234, slices
//    compiler.c:4550-4642).  On the GPU only one split matters: values that do not
//    depend on the pixel position are computed once per frame by a one-lane
//    prologue kernel and reach the pixel kernel through a small constant buffer
//    (scalar loads), everything else runs per work-item.
#include <cctype>
#include <cstring>
#include <functional>
#include <map>
#include <set>
#include <stdexcept>
#include <algorithm>

#include "passes.h"

namespace mm {

namespace {

void for_each_prim(Rhs &r, const std::function<void(Primary &)> &fn) {
    if (r.kind == Rhs::Prim) fn(r.prim);
    for (Primary &p : r.args) fn(p);
}

void for_each_stmt(Block &b, const std::function<void(Stmt *)> &fn) {
    for (Stmt *s : b) {
        fn(s);
        if (s->kind == Stmt::If) {
            for_each_stmt(s->then_, fn);
            for_each_stmt(s->else_, fn);
            for_each_stmt(s->phis, fn);
        } else if (s->kind == Stmt::While) {
            for_each_stmt(s->phis, fn);
            for_each_stmt(s->body, fn);
        }
    }
}

bool rhs_pure(const Rhs &r) {
    if (r.kind == Rhs::Op) return r.op->pure;
    return true;   // native closures are memoised per frame -> treated as pure (mathmap_common.c:321-377 is_pure)
}

}  // namespace

// v = w (same compvar type)  ==>  uses of v read w
bool copy_propagate(FilterCode &code) {
    std::map<Value *, Primary> repl;
    for_each_stmt(code.body, [&](Stmt *s) {
        if (s->kind != Stmt::Assign || s->rhs.kind != Rhs::Prim) return;
        const Primary &p = s->rhs.prim;
        if (p.kind == Primary::Val) {
            if (p.value->index < 0) return;
            if (p.value->var->type != s->lhs->var->type) return;
            if (p.value->var->type == Ty::Tuple && p.value->var->tuple_len != s->lhs->var->tuple_len) return;
            repl[s->lhs] = p;
        } else if ((p.kind == Primary::IntConst && s->lhs->var->type == Ty::Int) ||
                   (p.kind == Primary::FloatConst && s->lhs->var->type == Ty::Float)) {
            // a constant of exactly the variable's type can stand in for it: the C
            // expression sees the same operand type either way (floats print as the
            // double literal of the float value, cc.c / ops.lisp:343-358)
            repl[s->lhs] = p;
        }
    });
    if (repl.empty()) return false;
    auto resolve = [&](Primary p) {
        for (int guard = 0; guard < 10000 && p.kind == Primary::Val; ++guard) {
            auto it = repl.find(p.value);
            if (it == repl.end()) break;
            p = it->second;
        }
        return p;
    };
    bool changed = false;
    auto fix = [&](Primary &p) {
        if (p.kind != Primary::Val) return;
        Primary q = resolve(p);
        if (q.kind != p.kind || q.value != p.value) { p = q; changed = true; }
    };
    for_each_stmt(code.body, [&](Stmt *s) {
        switch (s->kind) {
            case Stmt::Assign: for_each_prim(s->rhs, fix); break;
            case Stmt::Phi: for_each_prim(s->rhs, fix); for_each_prim(s->rhs2, fix); break;
            case Stmt::If:
            case Stmt::While: for_each_prim(s->cond, fix); break;
            default: break;
        }
    });
    for (int i = 0; i < 4; ++i)
        if (code.result[i]) {
            Primary p = resolve(Primary::V(code.result[i]));
            // results must stay values; only follow value->value copies
            if (p.kind == Primary::Val) code.result[i] = p.value;
        }
    return changed;
}

// Liveness from the roots -- the results, the conditions of the control statements, impure statements -- instead of use
// counts: removes values that only feed each other around a loop (`while .. do img = render(img) end` with img unused behind
// it), which eliminate_dead_code's counts keep.  For the render code of a closure image: it is the main filter's code once
// more with another result, most of it dead, and what stays of the main code's native calls is *run* again (runtime.cpp
// render_closure) -- a consumer of the closure itself among them would be a closure inside a closure's render.
bool eliminate_dead_cycles(FilterCode &code) {
    std::set<const Value *> live;
    std::vector<const Value *> work;
    auto mark = [&](const Primary &p) { if (p.kind == Primary::Val && p.value->index >= 0 && live.insert(p.value).second) work.push_back(p.value); };
    auto mark_rhs = [&](const Rhs &r) { if (r.kind == Rhs::Prim) mark(r.prim); for (const Primary &p : r.args) mark(p); };
    for_each_stmt(code.body, [&](Stmt *s) {
        if (s->kind == Stmt::If || s->kind == Stmt::While) mark_rhs(s->cond);
        if (s->kind == Stmt::Assign && !rhs_pure(s->rhs)) { mark(Primary::V(s->lhs)); }
    });
    for (int i = 0; i < 4; ++i)
        if (code.result[i]) mark(Primary::V(code.result[i]));
    while (!work.empty()) {
        const Value *v = work.back();
        work.pop_back();
        const Stmt *d = v->def;
        if (!d) continue;
        mark_rhs(d->rhs);
        if (d->kind == Stmt::Phi) mark_rhs(d->rhs2);
    }
    bool changed = false;
    std::function<void(Block &)> sweep = [&](Block &b) {
        Block out;
        for (Stmt *s : b) {
            bool dead = (s->kind == Stmt::Assign || s->kind == Stmt::Phi) && !live.count(s->lhs) && (s->kind == Stmt::Phi || rhs_pure(s->rhs));
            if (s->kind == Stmt::If) { sweep(s->then_); sweep(s->else_); sweep(s->phis); }
            if (s->kind == Stmt::While) { sweep(s->phis); sweep(s->body); }
            if (dead) changed = true;
            else out.push_back(s);
        }
        b.swap(out);
    };
    sweep(code.body);
    if (changed) eliminate_dead_code(code);      // (conditionals emptied by the sweep)
    return changed;
}

bool eliminate_dead_code(FilterCode &code) {
    bool any = false;
    for (;;) {
        std::map<Value *, int> uses;
        auto count = [&](Primary &p) { if (p.kind == Primary::Val) ++uses[p.value]; };
        for_each_stmt(code.body, [&](Stmt *s) {
            switch (s->kind) {
                case Stmt::Assign: for_each_prim(s->rhs, count); break;
                case Stmt::Phi: {
                    // a loop phi feeding only itself is dead: do not count self-uses
                    auto cnt = [&](Primary &p) { if (p.kind == Primary::Val && p.value != s->lhs) ++uses[p.value]; };
                    for_each_prim(s->rhs, cnt);
                    for_each_prim(s->rhs2, cnt);
                    break;
                }
                case Stmt::If:
                case Stmt::While: for_each_prim(s->cond, count); break;
                default: break;
            }
        });
        for (int i = 0; i < 4; ++i)
            if (code.result[i]) ++uses[code.result[i]];
        bool changed = false;
        std::function<void(Block &)> sweep = [&](Block &b) {
            Block out;
            for (Stmt *s : b) {
                bool dead = false;
                if (s->kind == Stmt::Assign && uses[s->lhs] == 0 && rhs_pure(s->rhs)) dead = true;
                if (s->kind == Stmt::Phi && uses[s->lhs] == 0) dead = true;
                if (s->kind == Stmt::If) {
                    sweep(s->then_);
                    sweep(s->else_);
                    sweep(s->phis);
                    if (s->then_.empty() && s->else_.empty() && s->phis.empty()) dead = true;
                }
                if (s->kind == Stmt::While) {
                    sweep(s->phis);
                    sweep(s->body);
                }
                if (s->kind == Stmt::Nil) dead = true;
                if (dead) changed = true;
                else out.push_back(s);
            }
            b.swap(out);
        };
        sweep(code.body);
        if (!changed) break;
        any = true;
    }
    return any;
}

// ---------------------------------------------------------------------------
// Loop-carried common subexpressions
// ---------------------------------------------------------------------------
// A `while` body that starts by computing op(phi values) and ends (before the back edge) by
// computing the same op on the values those phis receive computes every such value twice: at
// the end of iteration i and again at the start of iteration i+1.  Mandelbrot's escape test
// |z'|^2 = a'^2 + b'^2 and the next z'^2 = (a'^2 - b'^2, 2a'b') share both squares this way.
// The op at the top is pure, so it is replaced by a new entry phi
//     s = phi(op(entry values) [materialised before the loop], t)   with t the bottom computation.
// Value-preserving by construction: the same operator on the same operand values.
namespace {

bool same_primary(const Primary &a, const Primary &b) {
    if (a.kind != b.kind) return false;
    switch (a.kind) {
        case Primary::Val: return a.value == b.value;
        case Primary::IntConst: return a.i == b.i;
        case Primary::FloatConst: return std::memcmp(&a.f, &b.f, sizeof a.f) == 0;
        default: return false;
    }
}

bool loop_cse_block(FilterCode &code, Block &blk) {
    bool changed = false;
    for (size_t wi = 0; wi < blk.size(); ++wi) {
        Stmt *w = blk[wi];
        if (w->kind == Stmt::If) { changed |= loop_cse_block(code, w->then_); changed |= loop_cse_block(code, w->else_); }
        if (w->kind != Stmt::While) continue;
        changed |= loop_cse_block(code, w->body);
        std::map<const Value *, Stmt *> phi_of;
        for (Stmt *ph : w->phis) phi_of[ph->lhs] = ph;
        std::set<const Value *> defined_in_body;
        for_each_stmt(w->body, [&](Stmt *s) { if (s->lhs) defined_in_body.insert(s->lhs); });
        for (size_t si = 0; si < w->body.size(); ++si) {
            Stmt *s = w->body[si];
            if (s->kind != Stmt::Assign || s->rhs.kind != Rhs::Op || !s->rhs.op->pure) continue;
            const Ty ty = s->lhs->var->type;
            if (ty != Ty::Int && ty != Ty::Float) continue;
            // operands: loop phis (at least one), constants, or values defined outside the loop
            bool ok = true, any_phi = false;
            std::vector<Primary> back = s->rhs.args, entry = s->rhs.args;
            for (size_t a = 0; a < s->rhs.args.size() && ok; ++a) {
                const Primary &p = s->rhs.args[a];
                if (p.kind != Primary::Val) { ok = p.kind == Primary::IntConst || p.kind == Primary::FloatConst; continue; }
                auto it = phi_of.find(p.value);
                if (it != phi_of.end()) {
                    if (it->second->rhs.kind != Rhs::Prim || it->second->rhs2.kind != Rhs::Prim) { ok = false; break; }
                    entry[a] = it->second->rhs.prim;
                    back[a] = it->second->rhs2.prim;
                    any_phi = true;
                } else if (defined_in_body.count(p.value))
                    ok = false;
            }
            if (!ok || !any_phi) continue;
            // the same op on the back-edge values, at the top level of the body (runs every iteration)
            Stmt *t = nullptr;
            for (size_t ti = 0; ti < w->body.size() && !t; ++ti) {
                Stmt *c = w->body[ti];
                if (c == s || c->kind != Stmt::Assign || c->rhs.kind != Rhs::Op || c->rhs.op != s->rhs.op) continue;
                if (c->lhs->var->type != ty || c->rhs.args.size() != back.size()) continue;
                bool eq = true;
                for (size_t a = 0; a < back.size() && eq; ++a) eq = same_primary(c->rhs.args[a], back[a]);
                if (eq) t = c;
            }
            if (!t) continue;
            // u = op(entry values), once before the loop
            Stmt *u = code.new_stmt(Stmt::Assign);
            u->lhs = code.new_value(code.new_var(ty));
            u->lhs->def = u;
            u->rhs = Rhs::O(s->rhs.op, entry);
            u->parent = w->parent;
            blk.insert(blk.begin() + wi, u);
            ++wi;
            // s becomes an entry phi of the loop
            s->kind = Stmt::Phi;
            s->rhs = Rhs::V(u->lhs);
            s->rhs2 = Rhs::V(t->lhs);
            s->old_value = u->lhs;
            s->parent = w;
            w->body.erase(w->body.begin() + si);
            --si;
            w->phis.push_back(s);
            phi_of[s->lhs] = s;
            changed = true;
        }
    }
    return changed;
}

}  // namespace

bool loop_carried_cse(FilterCode &code) { return loop_cse_block(code, code.body); }

// ---------------------------------------------------------------------------
// Common subexpressions
// ---------------------------------------------------------------------------
// `v = op(args)` where a dominating statement already computed the same pure operator on the same operand values
// into a compvar of the same type becomes the copy `v = w` (copy propagation then removes it).  ADD and MUL match
// with their operands in either order: the complex product the front-end emits for z * z computes a*b and b*a.
// SSA values never change, so whatever an enclosing block has computed is still valid inside its ifs and loops.
namespace {

std::string prim_key(const Primary &p) {
    char buf[64];
    switch (p.kind) {
        case Primary::Val: snprintf(buf, sizeof buf, "v%p", (const void *)p.value); break;
        case Primary::IntConst: snprintf(buf, sizeof buf, "i%d", p.i); break;
        case Primary::FloatConst: { unsigned u; memcpy(&u, &p.f, 4); snprintf(buf, sizeof buf, "f%08x", u); break; }
        case Primary::ComplexConst: { unsigned u, w; memcpy(&u, &p.f, 4); memcpy(&w, &p.f2, 4); snprintf(buf, sizeof buf, "c%08x_%08x", u, w); break; }
        case Primary::ColorConst: snprintf(buf, sizeof buf, "k%08x", p.color); break;
        default: snprintf(buf, sizeof buf, "?"); break;
    }
    return buf;
}

bool cse_block(Block &blk, std::vector<std::map<std::string, Value *>> &scopes) {
    bool changed = false;
    scopes.emplace_back();
    for (Stmt *s : blk) {
        if (s->kind == Stmt::Assign && s->rhs.kind == Rhs::Op && s->rhs.op->pure && s->lhs->index >= 0) {
            std::vector<std::string> keys;
            bool ok = true;
            for (const Primary &p : s->rhs.args) {
                if (p.kind == Primary::Val && p.value->index < 0) ok = false;      // uninitialised reads: leave alone
                keys.push_back(prim_key(p));
            }
            if (ok) {
                const char *cn = s->rhs.op->cname;
                if (keys.size() == 2 && (!strcmp(cn, "ADD") || !strcmp(cn, "MUL")) && keys[1] < keys[0]) std::swap(keys[0], keys[1]);
                std::string key = std::string(cn) + "/" + std::to_string((int)s->lhs->var->type) + "/" + std::to_string(s->lhs->var->tuple_len);
                for (const std::string &k : keys) key += "," + k;
                Value *have = nullptr;
                for (auto it = scopes.rbegin(); it != scopes.rend() && !have; ++it) {
                    auto f = it->find(key);
                    if (f != it->end()) have = f->second;
                }
                if (have) {
                    s->rhs = Rhs::V(have);
                    changed = true;
                } else {
                    scopes.back()[key] = s->lhs;
                }
            }
        } else if (s->kind == Stmt::If) {
            changed |= cse_block(s->then_, scopes);
            changed |= cse_block(s->else_, scopes);
        } else if (s->kind == Stmt::While) {
            changed |= cse_block(s->body, scopes);
        }
    }
    scopes.pop_back();
    return changed;
}

}  // namespace

bool common_subexpressions(FilterCode &code) {
    std::vector<std::map<std::string, Value *>> scopes;
    return cse_block(code.body, scopes);
}

void optimize(FilterCode &code) {
    propagate_types(code);
    for (int i = 0; i < 20; ++i) {
        bool c = copy_propagate(code);
        c |= eliminate_dead_code(code);
        if (!c) break;
    }
    bool cse = !getenv("MMHIP_NO_CSE") && common_subexpressions(code);
    if (loop_carried_cse(code) || cse) {
        propagate_types(code);
        for (int i = 0; i < 20; ++i) {
            bool c = copy_propagate(code);
            c |= eliminate_dead_code(code);
            if (!c) break;
        }
    }
}

// ---------------------------------------------------------------------------
// Frame-constant analysis
// ---------------------------------------------------------------------------
namespace {

bool internal_is_frame_const(const std::string &n) { return n != "x" && n != "y" && n != "__colF" && n != "__rowF"; }

}  // namespace

// Marks every value that depends on the pixel position (directly, through data
// flow, or through control flow) as not hoistable; the rest is `hoisted`.  Then
// statements are assigned to the prologue slice, the pixel slice, or both
// (control statements whose bodies contain work of both kinds).
static void analyze_frame_constants_once(FilterCode &code);
static bool is_native_call(const Stmt *s);

// A loop that holds per-pixel code runs per pixel as a whole (below) -- but a native filter called in it has to be run by
// the host, from the frame-constant slice:
//     while i < n do img = gaussian_blur(img, s, s); acc = acc + img(xy) * 0.3; i = i + 1 end
// Such a loop is emitted in both slices.  Its *loop-constant* part -- statements resting on literals, frame constants from
// outside, and each other, under control that does likewise; the native calls among them -- runs in the prologue too,
// where the calls are numbered and recorded as they are made (hipgen.cpp mm_native_call_in_loop); the pixel slice runs the
// whole loop and, making the same calls in the same order (the control around them is loop-constant), takes the n-th
// call's result for its n-th call.  Marks: Stmt::hoisted and Stmt::in_pixel both set, Value::loop_const.
static void mark_dual_loops(FilterCode &code) {
    std::set<const Value *> was;
    for (Value &v : code.values) { if (v.loop_const) was.insert(&v); v.loop_const = false; }
    auto outside_ok = [&](const Primary &p) { return p.kind != Primary::Val || p.value->index < 0 || p.value->hoisted || was.count(p.value); };
    std::function<void(Block &, bool)> visit = [&](Block &b, bool ctx_const) {
        for (Stmt *top : b) {
            if (top->kind == Stmt::If) {
                bool c = ctx_const;
                if (top->cond.kind == Rhs::Prim) c = c && outside_ok(top->cond.prim);
                else if (top->cond.kind == Rhs::Internal) c = c && internal_is_frame_const(top->cond.internal);
                else for (const Primary &p : top->cond.args) c = c && outside_ok(p);
                visit(top->then_, c);
                visit(top->else_, c);
                continue;
            }
            if (top->kind != Stmt::While || !ctx_const || top->hoisted || !top->in_pixel) continue;      // outermost per-pixel loops under frame-constant control
            // candidates: every assignment and phi of the loop, with the control statements around each (inside the loop)
            std::map<Stmt *, std::vector<Stmt *>> control;      // statement -> enclosing control statements, `top' first
            std::vector<Stmt *> order;
            bool has_native = false;
            std::vector<Stmt *> stack{top};
            std::function<void(Block &)> gather = [&](Block &bb) {
                for (Stmt *s : bb) {
                    if (s->kind == Stmt::Assign || s->kind == Stmt::Phi) { control[s] = stack; order.push_back(s); has_native = has_native || is_native_call(s); }
                    if (s->kind == Stmt::If) { stack.push_back(s); gather(s->then_); gather(s->else_); gather(s->phis); stack.pop_back(); }
                    if (s->kind == Stmt::While) { stack.push_back(s); gather(s->phis); gather(s->body); stack.pop_back(); }
                }
            };
            gather(top->phis);
            gather(top->body);
            if (!has_native) continue;
            std::set<const Stmt *> in;
            for (Stmt *s : order) in.insert(s);
            auto operand_ok = [&](const Primary &p) {
                if (p.kind != Primary::Val || p.value->index < 0) return true;
                if (p.value->def && control.count(p.value->def)) return in.count(p.value->def) != 0;      // defined in this loop
                return p.value->hoisted || p.value->loop_const || was.count(p.value) != 0;      // (an earlier loop's, this run or the last)
            };
            auto rhs_ok = [&](const Rhs &r) {
                if (r.kind == Rhs::Prim) return operand_ok(r.prim);
                if (r.kind == Rhs::Internal) return internal_is_frame_const(r.internal);
                for (const Primary &p : r.args)
                    if (!operand_ok(p)) return false;
                return true;
            };
            auto stmt_ok = [&](Stmt *s) {
                if (s->kind == Stmt::Phi) { if (!rhs_ok(s->rhs) || !rhs_ok(s->rhs2)) return false; }
                else {
                    const Rhs &r = s->rhs;
                    const bool kind_ok = r.kind == Rhs::Prim || r.kind == Rhs::Tuple || r.kind == Rhs::Internal || is_native_call(s) ||
                                         (r.kind == Rhs::Op && r.op->pure && strcmp(r.op->cname, "ORIG_VAL"));
                    if (!kind_ok || !rhs_ok(r)) return false;
                }
                for (Stmt *c : control[s])
                    if (!rhs_ok(c->cond) || (c->cond.kind == Rhs::Op && !c->cond.op->pure)) return false;
                return true;
            };
            for (bool again = true; again;) {
                again = false;
                for (Stmt *s : order)
                    if (in.count(s) && !stmt_ok(s)) { in.erase(s); again = true; }
            }
            bool keeps_native = false;
            for (Stmt *s : order) keeps_native = keeps_native || (in.count(s) && is_native_call(s));
            if (!keeps_native) continue;
            for (Stmt *s : order) {
                if (!in.count(s)) continue;
                s->hoisted = true;
                s->lhs->loop_const = true;
                for (Stmt *c : control[s]) c->hoisted = true;
            }
            for (Stmt *a = top->parent; a; a = a->parent) a->hoisted = true;
        }
    };
    visit(code.body, true);
}

// Speculative hoisting: a pure library call (cexpf, sin, pow ...) whose operands are literals or
// frame constants is itself a frame constant even when it sits inside pixel-dependent control
// flow -- Droste evaluates cexpf(0 + 2 pi i) in a branch most pixels take.  Pure and total, so
// executing it unconditionally is safe; moved in front of its outermost enclosing statement it
// lands in the prologue slice (once per frame).  Only calls (and the COMPLEX(...) constructors
// feeding them) are moved: hoisting plain arithmetic would just turn registers into transfers.
static bool hoist_speculative(FilterCode &code) {
    bool moved_any = false;
    for (size_t ti = 0; ti < code.body.size(); ++ti) {
        Stmt *top = code.body[ti];
        if (top->kind != Stmt::If && top->kind != Stmt::While) continue;
        std::set<const Stmt *> inside;
        std::function<void(Block &)> collect = [&](Block &b) {
            for (Stmt *s : b) {
                inside.insert(s);
                if (s->kind == Stmt::If) { collect(s->then_); collect(s->else_); collect(s->phis); }
                if (s->kind == Stmt::While) { collect(s->phis); collect(s->body); }
            }
        };
        collect(top->then_); collect(top->else_); collect(top->phis); collect(top->body);
        std::vector<Stmt *> moved;
        std::set<const Value *> moved_vals;
        std::function<void(Block &)> scan = [&](Block &b) {
            for (size_t i = 0; i < b.size(); ++i) {
                Stmt *s = b[i];
                if (s->kind == Stmt::If) { scan(s->then_); scan(s->else_); continue; }
                if (s->kind == Stmt::While) { scan(s->body); continue; }
                if (s->kind != Stmt::Assign || s->rhs.kind != Rhs::Op || !s->rhs.op->pure || s->lhs->hoisted) continue;
                const char *n = s->rhs.op->cname;
                if (!(std::islower((unsigned char)n[0]) || !strcmp(n, "COMPLEX"))) continue;
                bool ok = true;
                for (const Primary &p : s->rhs.args) {
                    if (p.kind != Primary::Val) continue;
                    if (moved_vals.count(p.value)) continue;
                    if (p.value->index < 0 || !p.value->hoisted || inside.count(p.value->def)) { ok = false; break; }
                }
                if (!ok) continue;
                b.erase(b.begin() + i);
                --i;
                moved.push_back(s);
                moved_vals.insert(s->lhs);
            }
        };
        scan(top->then_); scan(top->else_); scan(top->body);
        // a COMPLEX(...) that no moved call uses goes back?  It is harmless at top level: it is a
        // frame constant there and dead-code elimination removes it if unused
        for (Stmt *s : moved) {
            s->parent = top->parent;
            code.body.insert(code.body.begin() + ti, s);
            ++ti;
            moved_any = true;
        }
    }
    return moved_any;
}

// A native-filter call (or render()) whose arguments are frame constants, sitting under pixel-dependent control --
// `if inside then b = gaussian_blur(in, s, s); b(xy) else in(xy) end`.  The reference runs the filter when the first
// pixel reaches the call and finds it in the cache from then on (native-filters/cache.c:110-147); the result does not
// depend on which pixel that was.  Here the call moves, with the pure statements that prepare its arguments, in front
// of the outermost pixel-dependent statement around it -- staying inside any frame-constant conditional further out --
// and so into the frame-constant slice: run once per frame, before the pixel kernel.  (Speculative if no pixel takes
// the branch: a blur too many, never a different pixel.)  Calls whose arguments depend on the pixel stay where they
// are and are refused by the code generator.
static bool is_native_call(const Stmt *s) {
    if (s->kind != Stmt::Assign) return false;
    if (s->rhs.kind == Rhs::Closure) return s->rhs.filter->kind == Filter::Native;
    return s->rhs.kind == Rhs::Op && !strcmp(s->rhs.op->cname, "RENDER");
}

static bool hoist_native_calls(FilterCode &code) {
    auto prim_const = [](const Primary &p) { return p.kind != Primary::Val || p.value->index < 0 || p.value->hoisted || p.value->loop_const; };
    auto cond_const = [&](const Stmt *c) {
        if (c->cond.kind == Rhs::Prim) return prim_const(c->cond.prim);
        if (c->cond.kind == Rhs::Internal) return internal_is_frame_const(c->cond.internal);
        for (const Primary &p : c->cond.args)
            if (!prim_const(p)) return false;
        return c->cond.kind != Rhs::Op || c->cond.op->pure;
    };
    // candidates first (the tree is edited afterwards)
    std::vector<std::pair<Stmt *, Stmt *>> work;      // (call, outermost pixel-dependent statement around it)
    std::vector<Stmt *> stack;
    std::function<void(Block &)> find = [&](Block &b) {
        for (Stmt *s : b) {
            if (is_native_call(s) && !s->lhs->hoisted) {
                for (Stmt *anc : stack)
                    if (!cond_const(anc)) { work.push_back({s, anc}); break; }
            }
            if (s->kind == Stmt::If) { stack.push_back(s); find(s->then_); find(s->else_); stack.pop_back(); }
            if (s->kind == Stmt::While) { stack.push_back(s); find(s->body); stack.pop_back(); }
        }
    };
    find(code.body);
    bool moved_any = false;
    for (auto &w : work) {
        Stmt *call = w.first, *outer = w.second;
        std::set<const Stmt *> inside;
        std::function<void(Block &)> collect = [&](Block &b) {
            for (Stmt *s : b) {
                inside.insert(s);
                if (s->kind == Stmt::If) { collect(s->then_); collect(s->else_); collect(s->phis); }
                if (s->kind == Stmt::While) { collect(s->phis); collect(s->body); }
            }
        };
        collect(outer->then_); collect(outer->else_); collect(outer->phis); collect(outer->body);
        if (!inside.count(call)) continue;      // moved along with an earlier call
        // the call and what prepares its arguments inside `outer', definitions before uses
        std::vector<Stmt *> order;
        std::set<const Stmt *> seen;
        bool ok = true;
        std::function<void(Stmt *)> need = [&](Stmt *s) {
            if (!ok || seen.count(s)) return;
            seen.insert(s);
            const bool fine = s->kind == Stmt::Assign &&
                              (s == call || s->rhs.kind == Rhs::Prim || s->rhs.kind == Rhs::Tuple || is_native_call(s) ||
                               (s->rhs.kind == Rhs::Internal && internal_is_frame_const(s->rhs.internal)) ||
                               (s->rhs.kind == Rhs::Op && s->rhs.op->pure && strcmp(s->rhs.op->cname, "ORIG_VAL")));
            if (!fine) { ok = false; return; }
            std::vector<Primary> ps = s->rhs.args;
            if (s->rhs.kind == Rhs::Prim) ps.push_back(s->rhs.prim);
            for (const Primary &p : ps) {
                if (p.kind != Primary::Val || p.value->index < 0) continue;
                if (p.value->def && inside.count(p.value->def)) need(p.value->def);
                else if (!p.value->hoisted && !p.value->loop_const) ok = false;
                if (!ok) return;
            }
            order.push_back(s);
        };
        need(call);
        if (!ok) continue;
        // out of their blocks ...
        std::function<void(Block &)> strip = [&](Block &b) {
            for (size_t i = 0; i < b.size(); ++i) {
                Stmt *s = b[i];
                if (seen.count(s)) { b.erase(b.begin() + i); --i; continue; }
                if (s->kind == Stmt::If) { strip(s->then_); strip(s->else_); }
                if (s->kind == Stmt::While) strip(s->body);
            }
        };
        strip(outer->then_); strip(outer->else_); strip(outer->body);
        // ... and in front of `outer'
        Block *home = &code.body;
        if (Stmt *p = outer->parent) {
            if (std::find(p->then_.begin(), p->then_.end(), outer) != p->then_.end()) home = &p->then_;
            else if (std::find(p->else_.begin(), p->else_.end(), outer) != p->else_.end()) home = &p->else_;
            else home = &p->body;
        }
        auto at = std::find(home->begin(), home->end(), outer);
        if (at == home->end()) throw std::runtime_error("internal: statement not found in its parent's block");
        for (Stmt *s : order) {
            s->parent = outer->parent;
            s->lhs->hoisted = true;      // frame-constant operands under frame-constant control now: what the next call's arguments may rest on
        }
        home->insert(at, order.begin(), order.end());
        moved_any = true;
    }
    return moved_any;
}

// analyze_frame_constants_once until the marks stand still: what a loop of both slices leaves behind (loop_const) is a
// frame constant to the code behind it, which the next run then sees
static void analyze_to_fixpoint(FilterCode &code) {
    for (Value &v : code.values) v.loop_const = false;
    size_t before = ~(size_t)0;
    for (int iter = 0; iter < 6; ++iter) {
        analyze_frame_constants_once(code);
        size_t now = 0;
        for (const Value &v : code.values) now += (v.hoisted ? 1 : 0) + (v.loop_const ? 0x10000 : 0);
        if (now == before) break;
        before = now;
    }
}

void analyze_frame_constants(FilterCode &code) {
    analyze_to_fixpoint(code);
    if (hoist_speculative(code)) analyze_to_fixpoint(code);
    if (hoist_native_calls(code)) analyze_to_fixpoint(code);
}

static void analyze_frame_constants_once(FilterCode &code) {
    for (Value &v : code.values) v.hoisted = true;
    bool changed = true;
    // Images that are (or may be, through copies, phis and resize wrappers) results of native filters / render(): their
    // pixels exist only after the host has run the filter's kernels, *behind* the frame-constant code -- a fetch from one
    // is per-pixel code even at frame-constant coordinates (the reference's init_frame computes the map on the spot).
    std::set<const Value *> native_img;
    {
        bool grew = true;
        std::function<void(Block &)> scan = [&](Block &b) {
            for (Stmt *s : b) {
                if (s->kind == Stmt::Assign || s->kind == Stmt::Phi) {
                    bool d = is_native_call(s);
                    if (!d && s->lhs->var && s->lhs->var->type == Ty::Image) {
                        auto from = [&](const Rhs &r) {
                            if (r.kind == Rhs::Prim && r.prim.kind == Primary::Val && native_img.count(r.prim.value)) return true;
                            for (const Primary &p : r.args)
                                if (p.kind == Primary::Val && native_img.count(p.value)) return true;
                            return false;
                        };
                        d = from(s->rhs) || (s->kind == Stmt::Phi && from(s->rhs2));
                    }
                    if (d && native_img.insert(s->lhs).second) grew = true;
                }
                if (s->kind == Stmt::If) { scan(s->then_); scan(s->else_); scan(s->phis); }
                if (s->kind == Stmt::While) { scan(s->phis); scan(s->body); }
            }
        };
        while (grew) { grew = false; scan(code.body); }
    }
    // (loop_const, from the previous run: the value a loop of both slices leaves behind is a frame constant to the code
    // behind the loop -- the prologue ran its copy of the loop; inside the loop everything is demoted below anyway)
    auto prim_hoisted = [](const Primary &p) { return p.kind != Primary::Val || p.value->index < 0 || p.value->hoisted || p.value->loop_const; };
    auto rhs_hoisted = [&](const Rhs &r) {
        if (r.kind == Rhs::Internal) return internal_is_frame_const(r.internal);
        if (r.kind == Rhs::Op && !r.op->pure) return false;
        if (r.kind == Rhs::Op && !strcmp(r.op->cname, "ORIG_VAL") && r.args.size() >= 3 && r.args[2].kind == Primary::Val &&
            native_img.count(r.args[2].value))
            return false;
        if (r.kind == Rhs::Prim) return prim_hoisted(r.prim);
        for (const Primary &p : r.args)
            if (!prim_hoisted(p)) return false;
        return true;
    };
    std::function<void(Block &, bool)> walk = [&](Block &b, bool ctx) {
        for (Stmt *s : b) {
            switch (s->kind) {
                case Stmt::Assign: {
                    bool h = ctx && rhs_hoisted(s->rhs);
                    if (!h && s->lhs->hoisted) { s->lhs->hoisted = false; changed = true; }
                    break;
                }
                case Stmt::If: {
                    bool c = ctx && rhs_hoisted(s->cond);
                    walk(s->then_, c);
                    walk(s->else_, c);
                    for (Stmt *p : s->phis) {
                        bool h = c && rhs_hoisted(p->rhs) && rhs_hoisted(p->rhs2);
                        if (!h && p->lhs->hoisted) { p->lhs->hoisted = false; changed = true; }
                    }
                    break;
                }
                case Stmt::While: {
                    // the loop condition reads entry phis, so iterate locally
                    bool c = ctx && rhs_hoisted(s->cond);
                    for (Stmt *p : s->phis) {
                        bool h = c && rhs_hoisted(p->rhs) && rhs_hoisted(p->rhs2);
                        if (!h && p->lhs->hoisted) { p->lhs->hoisted = false; changed = true; }
                    }
                    c = ctx && rhs_hoisted(s->cond);
                    walk(s->body, c);
                    break;
                }
                default: break;
            }
        }
    };
    while (changed) {
        changed = false;
        walk(code.body, true);
    }
    // statement slices
    std::function<void(Block &, bool &, bool &)> mark = [&](Block &b, bool &any_h, bool &any_p) {
        for (Stmt *s : b) {
            switch (s->kind) {
                case Stmt::Assign:
                case Stmt::Phi:
                    s->hoisted = s->lhs->hoisted;
                    s->in_pixel = !s->lhs->hoisted;
                    break;
                case Stmt::If: {
                    bool h = false, p = false;
                    mark(s->then_, h, p);
                    mark(s->else_, h, p);
                    mark(s->phis, h, p);
                    s->hoisted = h;
                    s->in_pixel = p;
                    break;
                }
                case Stmt::While: {
                    bool h = false, p = false;
                    mark(s->phis, h, p);
                    mark(s->body, h, p);
                    s->hoisted = h;
                    s->in_pixel = p;
                    // a loop cannot be split: if any of it is per-pixel the whole loop
                    // (including its frame-constant parts) runs per pixel
                    if (p && h) {
                        std::function<void(Block &)> demote = [&](Block &bb) {
                            for (Stmt *t : bb) {
                                if (t->kind == Stmt::Assign || t->kind == Stmt::Phi) t->lhs->hoisted = false;
                                if (t->kind == Stmt::If) { demote(t->then_); demote(t->else_); demote(t->phis); }
                                if (t->kind == Stmt::While) { demote(t->phis); demote(t->body); }
                            }
                        };
                        demote(s->phis);
                        demote(s->body);
                    }
                    break;
                }
                default: break;
            }
            any_h |= s->hoisted;
            any_p |= s->in_pixel;
        }
    };
    // demotion inside mixed loops can invalidate earlier decisions -> iterate
    for (int iter = 0; iter < 10; ++iter) {
        bool h = false, p = false;
        size_t before = 0, after = 0;
        for (Value &v : code.values) before += v.hoisted;
        mark(code.body, h, p);
        changed = true;
        while (changed) { changed = false; walk(code.body, true); }
        for (Value &v : code.values) after += v.hoisted;
        if (before == after) { mark(code.body, h, p); break; }
    }
    mark_dual_loops(code);
}

}  // namespace mm
