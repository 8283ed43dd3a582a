// SSA emission context shared by the tree lowering (lower.cpp) and the builtin
// generators (builtins.cpp).
//
// Structured SSA is built on the fly: every assignment inside an open if/while
// registers a phi in the innermost construct; when the construct closes, its phis
// are committed into the next enclosing one.  This gives the same IR shape the
// reference builds in compiler.c:1009-1378 (exit phis on ifs, entry phis on
// loops, uses inside a loop rewritten to the loop phi).
#pragma once
#include <map>
#include <vector>

#include "front.h"
#include "ir.h"

namespace mm {

class Gen {
   public:
    explicit Gen(FilterCode &c) : code(c) { blocks_.push_back(&c.body); }
    FilterCode &code;

    CompVar *temp(Ty t = Ty::Int) { return code.new_var(t); }
    Primary P(CompVar *v) const { return Primary::V(v->current); }
    const OpInfo *op(const char *cname, int nargs) const;

    Value *assign(CompVar *dst, Rhs rhs);
    Value *assign_op(CompVar *dst, const char *cname, std::vector<Primary> args);
    void copy(CompVar *dst, CompVar *src) { assign(dst, Rhs::V(src->current)); }

    void start_if(Rhs cond);
    void switch_branch();
    void end_if();
    // continue emitting at the end of a branch (0 = then, 1 = else) of an `if' that already exists; closed by end_if(),
    // which adds exit phis for what was assigned meanwhile
    void reenter_if(Stmt *s, int branch);
    void start_while(CompVar *invariant);
    void end_while();
    // appends an already built statement (and what it contains) at the current position
    void append(Stmt *s) { emit(s); }

   private:
    struct Frame {
        Stmt *stmt;
        bool is_while;
        int branch;
        std::vector<Stmt *> phis;
        std::map<CompVar *, Stmt *> phi_of;
    };
    std::vector<Block *> blocks_;
    std::vector<Frame> frames_;
    std::vector<Stmt *> parents_;

    void emit(Stmt *s);
    void commit(CompVar *c, Value *nv);
    static void rewrite_rhs(Rhs &r, Value *from, Value *to);
    static void rewrite_block(Block &b, Value *from, Value *to);
};

// An image value followed through copies, STRIP_RESIZE and RESIZE_IMAGE down to what it is made of
// (lower.cpp, abi_backend.cpp: closure images handed to native filters).
struct ImageChain {
    enum Base { Unknown, MathMapClosure, Runtime } base = Unknown;
    Stmt *closure_def = nullptr;
    std::vector<std::pair<Primary, Primary>> factors;   // resize factors applied on top of the base, outermost first
};
ImageChain resolve_image_chain(Value *v);
// The coordinates at which render_image evaluates a closure for the pixel being computed: the pixel's own raw x, y
// (its closure branch, builtins.c:273-298) or, for a closure that still wears resize wrappers, the float arithmetic
// of the drawable branch with the wrappers' factors applied (builtins.c:303-343, opmacros.h:203-207).
void emit_closure_render_coordinates(Gen &g, const ImageChain &ch, bool raw, Primary *x, Primary *y);

// ---------------------------------------------------------------------------
// Tiny expression DSL for builtin generators: an `E` is a scalar compvar; the
// overloaded operators emit the corresponding IR op into a fresh temporary.
// Constants are emitted as `tmp = const` assignments exactly like the reference
// generators do (builtins.lisp:245-253), so C promotion rules see the same
// operand types.
// ---------------------------------------------------------------------------
struct E {
    Gen *g = nullptr;
    CompVar *v = nullptr;
    E() {}
    E(Gen &gen, CompVar *cv) : g(&gen), v(cv) {}
};

struct GenScope {
    static Gen *&cur() { static thread_local Gen *g = nullptr; return g; }
    Gen *saved;
    explicit GenScope(Gen &g) : saved(cur()) { cur() = &g; }
    ~GenScope() { cur() = saved; }
};

inline E lit(int i) { Gen &g = *GenScope::cur(); CompVar *t = g.temp(); g.assign(t, Rhs::I(i)); return E(g, t); }
inline E lit(float f) { Gen &g = *GenScope::cur(); CompVar *t = g.temp(); g.assign(t, Rhs::F(f)); return E(g, t); }
inline E lit(double f) { return lit((float)f); }

inline E opcall(const char *cname, std::vector<E> args) {
    Gen &g = *GenScope::cur();
    std::vector<Primary> ps;
    for (E &e : args) ps.push_back(g.P(e.v));
    CompVar *t = g.temp();
    g.assign_op(t, cname, ps);
    return E(g, t);
}

inline E operator+(E a, E b) { return opcall("ADD", {a, b}); }
inline E operator-(E a, E b) { return opcall("SUB", {a, b}); }
inline E operator*(E a, E b) { return opcall("MUL", {a, b}); }
inline E operator/(E a, E b) { return opcall("DIV", {a, b}); }
inline E operator%(E a, E b) { return opcall("MOD", {a, b}); }
inline E operator-(E a) { return opcall("NEG", {a}); }
inline E operator+(E a, int b) { return a + lit(b); }
inline E operator-(E a, int b) { return a - lit(b); }
inline E operator*(E a, int b) { return a * lit(b); }
inline E operator+(int a, E b) { return lit(a) + b; }
inline E operator-(int a, E b) { return lit(a) - b; }
inline E operator*(int a, E b) { return lit(a) * b; }
inline E operator/(E a, int b) { return a / lit(b); }

// Conditions (builtins.lisp:313-337): comparison ops yield 0/1 ints; and/or/not
// are lowered to nested ifs writing one int temporary.
struct Cond {
    enum Kind { Cmp, And, Or, Not } kind = Cmp;
    const char *cname = nullptr;
    E a, b;
    std::vector<Cond> sub;
};
inline Cond c_eq(E a, E b) { Cond c; c.cname = "EQ"; c.a = a; c.b = b; return c; }
inline Cond c_less(E a, E b) { Cond c; c.cname = "LESS"; c.a = a; c.b = b; return c; }
inline Cond c_leq(E a, E b) { Cond c; c.cname = "LEQ"; c.a = a; c.b = b; return c; }
inline Cond c_eq(E a, int b) { return c_eq(a, lit(b)); }
inline Cond c_less(E a, int b) { return c_less(a, lit(b)); }
inline Cond c_less(int a, E b) { return c_less(lit(a), b); }
inline Cond c_leq(E a, int b) { return c_leq(a, lit(b)); }
inline Cond c_leq(int a, E b) { return c_leq(lit(a), b); }
inline Cond c_and(Cond x, Cond y) { Cond c; c.kind = Cond::And; c.sub = {x, y}; return c; }
inline Cond c_or(Cond x, Cond y) { Cond c; c.kind = Cond::Or; c.sub = {x, y}; return c; }
inline Cond c_not(Cond x) { Cond c; c.kind = Cond::Not; c.sub = {x}; return c; }

// Emits the condition into a fresh int temporary and returns it.
CompVar *emit_cond(Gen &g, const Cond &c);
// if (cond) then_() else else_()
void gen_if(const Cond &c, const std::function<void()> &then_, const std::function<void()> &else_);

}  // namespace mm
