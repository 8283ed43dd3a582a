// The builtin function library of the .mm language as IR generators.
//
// Every entry restates the *arithmetic* of one definition in the reference's
// builtins.lisp:432-1395 (operation order, guards, constants) using the C++
// expression DSL of gen.h; registration order equals the reference's, because
// overload resolution picks the first match (overload.c:226-275).  Generators
// write into fresh result temporaries which are copied to the destination at the
// end (as the reference's gen_builtin does, builtins.lisp:417-421), so a
// destination that aliases an argument is safe.
#include <cmath>

#include "front.h"
#include "gen.h"

namespace mm {

class Parser;
AstNode *macro_origval(Parser &p, std::vector<AstNode *> &args, int pos, bool with_frame);

namespace {

struct B {
    Gen &g;
    const std::vector<std::vector<CompVar *>> &a;
    const std::vector<TInfo> &t;
    std::vector<CompVar *> &r;
    E A(int i, int j) const { return E(g, a[i][j]); }
    int len(int i) const { return (int)a[i].size(); }
    int rlen() const { return (int)r.size(); }
    void set(int i, E e) { g.copy(r[i], e.v); }
    void seti(int i, int c) { g.assign(r[i], Rhs::I(c)); }
    void setf(int i, float c) { g.assign(r[i], Rhs::F(c)); }
    void setall(std::initializer_list<E> es) { int i = 0; for (E e : es) set(i++, e); }
};

using Body = std::function<void(B &)>;

Pat parse_pat(Module &m, const std::string &s, bool is_tag) {
    Pat p;
    if (s == "_") { p.kind = Pat::Wild; return p; }
    if (isdigit((unsigned char)s[0])) { p.kind = Pat::Const; p.value = atoi(s.c_str()); return p; }
    if (isupper((unsigned char)s[0]) && s.size() == 1) { p.kind = Pat::Named; p.name = s[0]; return p; }
    p.kind = Pat::Const;
    p.value = is_tag ? m.tags.number(s) : atoi(s.c_str());
    return p;
}

ArgPat parse_argpat(Module &m, const std::string &s) {
    size_t c = s.find(':');
    ArgPat a;
    a.tag = parse_pat(m, s.substr(0, c), true);
    a.len = parse_pat(m, s.substr(c + 1), false);
    return a;
}

struct Reg {
    Module &m;
    void def(const char *name, const char *id, const char *res, std::vector<const char *> args, Body body) {
        BuiltinEntry e;
        e.name = name;
        e.id = id;
        e.result = parse_argpat(m, res);
        for (const char *a : args) e.args.push_back(parse_argpat(m, a));
        e.gen = [body](Gen &g, const std::vector<std::vector<CompVar *>> &a, const std::vector<TInfo> &t,
                       std::vector<CompVar *> &result) {
            std::vector<CompVar *> tmps(result.size());
            for (size_t i = 0; i < result.size(); ++i) tmps[i] = g.temp(result[i]->type);
            B b{g, a, t, tmps};
            body(b);
            for (size_t i = 0; i < result.size(); ++i) g.copy(result[i], tmps[i]);
        };
        m.builtins.push_back(std::move(e));
    }
    void macro(const char *name, const char *res, std::vector<const char *> args, MacroFn fn) {
        BuiltinEntry e;
        e.name = name;
        e.id = std::string("macro_") + name;
        e.result = parse_argpat(m, res);
        for (const char *a : args) e.args.push_back(parse_argpat(m, a));
        e.macro = std::move(fn);
        m.builtins.push_back(std::move(e));
    }
};

E fn1(const char *c, E x) { return opcall(c, {x}); }
E fn2(const char *c, E x, E y) { return opcall(c, {x, y}); }
E emin(E a, E b) { return opcall("MIN", {a, b}); }
E emax(E a, E b) { return opcall("MAX", {a, b}); }
E clamp01(E x) { return emax(lit(0), emin(lit(1), x)); }

// left fold of + over the elements of a tuple-valued expression (the `sum` form)
E sum_of(int n, const std::function<E(int)> &elem) {
    if (n == 1) return elem(0);
    E acc = elem(0) + elem(1);
    for (int i = 2; i < n; ++i) acc = acc + elem(i);
    return acc;
}

// complex function of one complex argument
void complex1(B &b, const char *cfn) {
    E c = fn1(cfn, fn2("COMPLEX", b.A(0, 0), b.A(0, 1)));
    b.setall({fn1("crealf", c), fn1("cimagf", c)});
}

}  // namespace

void register_builtins(Module &m) {
    Reg R{m};

    R.def("print", "print", "nil:1", {"_:_"}, [](B &b) {
        for (int i = 0; i < b.len(0); ++i) fn1("PRINT_FLOAT", b.A(0, i));
        opcall("NEWLINE", {});
        b.seti(0, 0);
    });

    // ---- addition / subtraction / negation -------------------------------------
    auto elementwise = [](const char *op) {
        return [op](B &b) { for (int i = 0; i < b.rlen(); ++i) b.set(i, opcall(op, {b.A(0, i), b.A(1, i)})); };
    };
    R.def("__add", "add_ri", "ri:2", {"ri:2", "ri:2"}, elementwise("ADD"));
    R.def("__add", "add_ri_1", "ri:2", {"ri:2", "_:1"},
          [](B &b) { b.setall({b.A(0, 0) + b.A(1, 0), b.A(0, 1) + lit(0)}); });
    R.def("__add", "add_1_ri", "ri:2", {"_:1", "ri:2"},
          [](B &b) { b.setall({b.A(1, 0) + b.A(0, 0), b.A(1, 1) + lit(0)}); });
    R.def("__add", "add_1", "T:1", {"T:1", "T:1"}, elementwise("ADD"));
    R.def("__add", "add_s", "T:L", {"T:L", "_:1"},
          [](B &b) { for (int i = 0; i < b.rlen(); ++i) b.set(i, b.A(0, i) + b.A(1, 0)); });
    R.def("__add", "add_n", "T:L", {"T:L", "T:L"}, elementwise("ADD"));

    R.def("__sub", "sub_ri", "ri:2", {"ri:2", "ri:2"}, elementwise("SUB"));
    R.def("__sub", "sub_ri_1", "ri:2", {"ri:2", "_:1"},
          [](B &b) { b.setall({b.A(0, 0) - b.A(1, 0), b.A(0, 1) - lit(0)}); });
    R.def("__sub", "sub_1_ri", "ri:2", {"_:1", "ri:2"},
          [](B &b) { b.setall({b.A(0, 0) - b.A(1, 0), lit(0) - b.A(1, 1)}); });
    R.def("__sub", "sub_1", "T:1", {"T:1", "T:1"}, elementwise("SUB"));
    R.def("__sub", "sub_s", "T:L", {"T:L", "_:1"},
          [](B &b) { for (int i = 0; i < b.rlen(); ++i) b.set(i, b.A(0, i) - b.A(1, 0)); });
    R.def("__sub", "sub_n", "T:L", {"T:L", "T:L"}, elementwise("SUB"));

    R.def("__neg", "neg", "T:L", {"T:L"}, [](B &b) { for (int i = 0; i < b.rlen(); ++i) b.set(i, -b.A(0, i)); });

    // ---- multiplication ------------------------------------------------------------
    R.def("__mul", "mul_ri", "ri:2", {"ri:2", "ri:2"}, [](B &b) {
        b.setall({b.A(0, 0) * b.A(1, 0) - b.A(0, 1) * b.A(1, 1), b.A(0, 0) * b.A(1, 1) + b.A(1, 0) * b.A(0, 1)});
    });
    R.def("__mul", "mul_1_ri", "ri:2", {"_:1", "ri:2"},
          [](B &b) { b.setall({b.A(0, 0) * b.A(1, 0), b.A(0, 0) * b.A(1, 1)}); });
    auto matmul = [](int n) {
        return [n](B &b) {
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j)
                    b.set(i * n + j, sum_of(n, [&](int k) { return b.A(0, i * n + k) * b.A(1, k * n + j); }));
        };
    };
    R.def("__mul", "mul_m2x2", "m2x2:4", {"m2x2:4", "m2x2:4"}, matmul(2));
    R.def("__mul", "mul_m3x3", "m3x3:9", {"m3x3:9", "m3x3:9"}, matmul(3));
    auto vecmat = [](int n) {
        return [n](B &b) {
            for (int i = 0; i < n; ++i) b.set(i, sum_of(n, [&](int j) { return b.A(0, j) * b.A(1, i + n * j); }));
        };
    };
    R.def("__mul", "mul_v2m2x2", "v2:2", {"v2:2", "m2x2:4"}, vecmat(2));
    R.def("__mul", "mul_v3m3x3", "v3:3", {"v3:3", "m3x3:9"}, vecmat(3));
    auto matvec = [](int n) {
        return [n](B &b) {
            for (int i = 0; i < n; ++i) b.set(i, sum_of(n, [&](int j) { return b.A(0, j + n * i) * b.A(1, j); }));
        };
    };
    R.def("__mul", "mul_m2x2v2", "v2:2", {"m2x2:4", "v2:2"}, matvec(2));
    R.def("__mul", "mul_m3x3v3", "v3:3", {"m3x3:9", "v3:3"}, matvec(3));

    // 4-component algebras: result[k] = sum over four signed products, in the
    // association order ((p0 + p1) + p2) + p3 with negated terms emitted as NEG.
    struct Term { int i, j, sign; };
    auto algebra = [](std::vector<std::vector<Term>> rows) {
        return [rows](B &b) {
            for (int k = 0; k < 4; ++k) {
                E acc;
                for (size_t n = 0; n < rows[k].size(); ++n) {
                    const Term &t = rows[k][n];
                    E p = b.A(0, t.i) * b.A(1, t.j);
                    if (t.sign < 0) p = -p;
                    acc = n == 0 ? p : acc + p;
                }
                b.set(k, acc);
            }
        };
    };
    R.def("__mul", "mul_quat", "quat:4", {"quat:4", "quat:4"},
          algebra({{{0, 0, 1}, {1, 1, -1}, {2, 2, -1}, {3, 3, -1}},
                   {{0, 1, 1}, {1, 0, 1}, {2, 3, 1}, {3, 2, -1}},
                   {{0, 2, 1}, {2, 0, 1}, {1, 3, -1}, {3, 1, 1}},
                   {{0, 3, 1}, {3, 0, 1}, {1, 2, 1}, {2, 1, -1}}}));
    R.def("__mul", "mul_cquat", "cquat:4", {"cquat:4", "cquat:4"},
          algebra({{{0, 0, 1}, {1, 1, -1}, {2, 2, 1}, {3, 3, 1}},
                   {{0, 1, 1}, {1, 0, 1}, {2, 3, 1}, {3, 2, 1}},
                   {{0, 2, 1}, {2, 0, 1}, {1, 3, -1}, {3, 1, -1}},
                   {{0, 3, 1}, {3, 0, 1}, {1, 2, -1}, {2, 1, -1}}}));
    R.def("__mul", "mul_hyper", "hyper:4", {"hyper:4", "hyper:4"},
          algebra({{{0, 0, 1}, {1, 1, -1}, {2, 2, -1}, {3, 3, 1}},
                   {{0, 1, 1}, {1, 0, 1}, {2, 3, -1}, {3, 2, -1}},
                   {{0, 2, 1}, {2, 0, 1}, {1, 3, -1}, {3, 1, -1}},
                   {{0, 3, 1}, {3, 0, 1}, {1, 2, 1}, {2, 1, 1}}}));
    R.def("__mul", "mul_1", "T:1", {"T:1", "T:1"}, [](B &b) { b.set(0, b.A(0, 0) * b.A(1, 0)); });
    R.def("__mul", "mul_s", "T:L", {"T:L", "_:1"},
          [](B &b) { for (int i = 0; i < b.rlen(); ++i) b.set(i, b.A(0, i) * b.A(1, 0)); });
    R.def("__mul", "mul_n", "T:L", {"T:L", "T:L"}, elementwise("MUL"));

    // ---- division / modulo -----------------------------------------------------------
    R.def("__div", "div_ri", "ri:2", {"ri:2", "ri:2"}, [](B &b) {
        gen_if(c_and(c_eq(b.A(1, 0), 0), c_eq(b.A(1, 1), 0)),
               [&] { b.seti(0, 0); b.seti(1, 0); },
               [&] {
                   E c = b.A(1, 0) * b.A(1, 0) + b.A(1, 1) * b.A(1, 1);
                   b.set(0, (b.A(0, 0) * b.A(1, 0) + b.A(0, 1) * b.A(1, 1)) / c);
                   b.set(1, ((-b.A(0, 0)) * b.A(1, 1) + b.A(1, 0) * b.A(0, 1)) / c);
               });
    });
    R.def("__div", "div_1_ri", "ri:2", {"T:1", "ri:2"}, [](B &b) {
        E tmp = b.A(1, 0) * b.A(1, 0) + b.A(1, 1) * b.A(1, 1);
        gen_if(c_eq(tmp, 0), [&] { b.seti(0, 0); b.seti(1, 0); },
               [&] {
                   b.set(0, (b.A(0, 0) * b.A(1, 0)) / tmp);
                   b.set(1, -((b.A(0, 0) * b.A(1, 1)) / tmp));
               });
    });
    auto div_vm = [](int n) {
        return [n](B &b) {
            Gen &g = b.g;
            std::vector<Primary> mp, vp;
            for (int i = 0; i < n * n; ++i) mp.push_back(g.P(b.a[1][i]));
            for (int i = 0; i < n; ++i) vp.push_back(g.P(b.a[0][i]));
            CompVar *mt = g.temp(Ty::Tuple), *vt = g.temp(Ty::Tuple), *rt = g.temp(Ty::Tuple);
            Rhs rm; rm.kind = Rhs::Tuple; rm.args = mp; g.assign(mt, rm);
            Rhs rv; rv.kind = Rhs::Tuple; rv.args = vp; g.assign(vt, rv);
            g.assign_op(rt, n == 2 ? "SOLVE_LINEAR_2" : "SOLVE_LINEAR_3", {g.P(mt), g.P(vt)});
            for (int i = 0; i < n; ++i) {
                CompVar *e = g.temp();
                g.assign_op(e, "TUPLE_NTH", {g.P(rt), Primary::I(i)});
                b.set(i, E(g, e));
            }
        };
    };
    R.def("__div", "div_v2m2x2", "v2:2", {"_:2", "m2x2:4"}, div_vm(2));
    R.def("__div", "div_v3m3x3", "v3:3", {"_:3", "m3x3:9"}, div_vm(3));
    auto guarded = [](const char *op, int divisor_mode) {
        // mode 0: scalar/scalar, 1: tuple/scalar, 2: tuple/tuple element-wise
        return [op, divisor_mode](B &b) {
            if (divisor_mode == 2) {
                for (int i = 0; i < b.len(1); ++i)
                    gen_if(c_eq(b.A(1, i), 0), [&] { b.seti(i, 0); },
                           [&] { b.set(i, opcall(op, {b.A(0, i), b.A(1, i)})); });
            } else {
                gen_if(c_eq(b.A(1, 0), 0), [&] { for (int i = 0; i < b.rlen(); ++i) b.seti(i, 0); },
                       [&] { for (int i = 0; i < b.rlen(); ++i) b.set(i, opcall(op, {b.A(0, i), b.A(1, 0)})); });
            }
        };
    };
    R.def("__div", "div_1", "T:1", {"T:1", "T:1"}, guarded("DIV", 0));
    R.def("__div", "div_s", "T:L", {"T:L", "_:1"}, guarded("DIV", 1));
    R.def("__div", "div_n", "T:L", {"T:L", "T:L"}, guarded("DIV", 2));
    R.def("__mod", "mod_1", "T:1", {"T:1", "T:1"}, guarded("MOD", 0));
    R.def("__mod", "mod_s", "T:L", {"T:L", "_:1"}, guarded("MOD", 1));
    R.def("__mod", "mod_n", "T:L", {"T:L", "T:L"}, guarded("MOD", 2));
    R.def("pmod", "pmod", "T:1", {"T:1", "T:1"}, [](B &b) {
        E mod = b.A(0, 0) % b.A(1, 0);
        gen_if(c_less(b.A(0, 0), 0), [&] { b.set(0, mod + b.A(1, 0)); }, [&] { b.set(0, mod); });
    });

    R.def("sqrt", "sqrt_ri", "ri:2", {"ri:2"}, [](B &b) { complex1(b, "csqrtf"); });
    R.def("sqrt", "sqrt_1", "T:1", {"T:1"}, [](B &b) { b.set(0, fn1("sqrt", b.A(0, 0))); });
    R.def("sum", "sum", "nil:1", {"T:L"}, [](B &b) { b.set(0, sum_of(b.len(0), [&](int i) { return b.A(0, i); })); });

    // ---- vectors -----------------------------------------------------------------------
    R.def("dotp", "dotp", "nil:1", {"T:L", "T:L"},
          [](B &b) { b.set(0, sum_of(b.len(0), [&](int i) { return b.A(0, i) * b.A(1, i); })); });
    R.def("crossp", "crossp", "T:3", {"T:3", "T:3"}, [](B &b) {
        b.setall({b.A(0, 1) * b.A(1, 2) - b.A(0, 2) * b.A(1, 1), b.A(0, 2) * b.A(1, 0) - b.A(0, 0) * b.A(1, 2),
                  b.A(0, 0) * b.A(1, 1) - b.A(0, 1) * b.A(1, 0)});
    });
    R.def("det", "det_m2x2", "nil:1", {"m2x2:4"},
          [](B &b) { b.set(0, b.A(0, 0) * b.A(0, 3) - b.A(0, 1) * b.A(0, 2)); });
    R.def("det", "det_m3x3", "nil:1", {"m3x3:9"}, [](B &b) {
        auto a = [&](int i) { return b.A(0, i); };
        E pos = a(0) * a(4) * a(8) + a(1) * a(5) * a(6) + a(2) * a(3) * a(7);
        E neg = a(2) * a(4) * a(6) + a(0) * a(5) * a(7) + a(1) * a(3) * a(8);
        b.set(0, pos - neg);
    });
    R.def("normalize", "normalize", "T:L", {"T:L"}, [](B &b) {
        E l = sum_of(b.len(0), [&](int i) { return b.A(0, i) * b.A(0, i); });
        gen_if(c_eq(l, 0), [&] { for (int i = 0; i < b.rlen(); ++i) b.seti(i, 0); },
               [&] { for (int i = 0; i < b.rlen(); ++i) b.set(i, b.A(0, i) / fn1("sqrt", l)); });
    });

    R.def("abs", "abs_ri", "nil:1", {"ri:2"}, [](B &b) { b.set(0, fn2("hypot", b.A(0, 0), b.A(0, 1))); });
    auto norm = [](int n) {
        return [n](B &b) { b.set(0, fn1("sqrt", sum_of(n, [&](int i) { return b.A(0, i) * b.A(0, i); }))); };
    };
    R.def("abs", "abs_quat", "nil:1", {"quat:4"}, norm(4));
    R.def("abs", "abs_cquat", "nil:1", {"cquat:4"}, norm(4));
    R.def("abs", "abs_hyper", "nil:1", {"hyper:4"}, norm(4));
    R.def("abs", "abs_v2", "nil:1", {"v2:2"}, norm(2));
    R.def("abs", "abs_v3", "nil:1", {"v3:3"}, norm(3));
    auto absn = [](B &b) { for (int i = 0; i < b.rlen(); ++i) b.set(i, fn1("fabs", b.A(0, i))); };
    R.def("abs", "abs_1", "T:1", {"T:1"}, absn);
    R.def("abs", "abs_n", "T:L", {"T:L"}, absn);

    // ---- trigonometry --------------------------------------------------------------------
    R.def("deg2rad", "deg2rad", "nil:1", {"_:1"}, [](B &b) { b.set(0, b.A(0, 0) * lit(0.017453292519943295722)); });
    R.def("rad2deg", "rad2deg", "deg:1", {"_:1"}, [](B &b) { b.set(0, b.A(0, 0) * lit(57.2957795130823208768)); });
    auto real1 = [](const char *fn) { return [fn](B &b) { b.set(0, fn1(fn, b.A(0, 0))); }; };
    auto cplx1 = [](const char *fn) { return [fn](B &b) { complex1(b, fn); }; };
    R.def("sin", "sin_ri", "ri:2", {"ri:2"}, cplx1("csinf"));
    R.def("sin", "sin", "T:1", {"T:1"}, real1("sin"));
    R.def("cos", "cos_ri", "ri:2", {"ri:2"}, cplx1("ccosf"));
    R.def("cos", "cos", "T:1", {"T:1"}, real1("cos"));
    R.def("tan", "tan_ri", "ri:2", {"ri:2"}, cplx1("ctanf"));
    R.def("tan", "tan", "T:1", {"T:1"}, real1("tan"));
    auto guarded_unit = [](const char *fn) {
        return [fn](B &b) {
            gen_if(c_or(c_less(b.A(0, 0), -1), c_less(1, b.A(0, 0))), [&] { b.seti(0, 0); },
                   [&] { b.set(0, fn1(fn, b.A(0, 0))); });
        };
    };
    R.def("asin", "asin_ri", "ri:2", {"ri:2"}, cplx1("casinf"));
    R.def("asin", "asin", "T:1", {"T:1"}, guarded_unit("asin"));
    R.def("acos", "acos_ri", "ri:2", {"ri:2"}, cplx1("cacosf"));
    R.def("acos", "acos", "T:1", {"T:1"}, guarded_unit("acos"));
    R.def("atan", "atan_ri", "ri:2", {"ri:2"}, cplx1("catanf"));
    R.def("atan", "atan", "T:1", {"T:1"}, real1("atan"));
    R.def("atan", "atan2", "T:1", {"T:1", "T:1"}, [](B &b) { b.set(0, fn2("atan2", b.A(0, 0), b.A(1, 0))); });

    // ---- exponentials ----------------------------------------------------------------------
    auto cpow = [](int mode) {
        return [mode](B &b) {
            E base = mode == 2 ? fn2("COMPLEX", b.A(0, 0), lit(0.0)) : fn2("COMPLEX", b.A(0, 0), b.A(0, 1));
            E ex = mode == 0 ? fn2("COMPLEX", b.A(1, 0), lit(0.0)) : fn2("COMPLEX", b.A(1, 0), b.A(1, 1));
            E c = fn2("cpowf", base, ex);
            b.setall({fn1("crealf", c), fn1("cimagf", c)});
        };
    };
    R.def("__pow", "pow_ri_1", "ri:2", {"ri:2", "T:1"}, cpow(0));
    R.def("__pow", "pow_ri", "ri:2", {"ri:2", "ri:2"}, cpow(1));
    R.def("__pow", "pow_1_ri", "ri:2", {"T:1", "ri:2"}, cpow(2));
    R.def("__pow", "pow_1", "T:1", {"T:1", "T:1"}, [](B &b) {
        gen_if(c_and(c_leq(b.A(1, 0), 0), c_eq(b.A(0, 0), 0)), [&] { b.seti(0, 0); },
               [&] { b.set(0, fn2("pow", b.A(0, 0), b.A(1, 0))); });
    });
    R.def("__pow", "pow_s", "T:L", {"T:L", "_:1"}, [](B &b) {
        for (int i = 0; i < b.len(0); ++i)
            gen_if(c_and(c_leq(b.A(1, 0), 0), c_eq(b.A(0, i), 0)), [&] { b.seti(i, 0); },
                   [&] { b.set(i, fn2("pow", b.A(0, i), b.A(1, 0))); });
    });
    R.def("exp", "exp_ri", "ri:2", {"ri:2"}, cplx1("cexpf"));
    R.def("exp", "exp_1", "T:1", {"T:1"}, real1("exp"));
    R.def("log", "log_ri", "ri:2", {"ri:2"}, cplx1("clogf"));
    R.def("log", "log_1", "T:1", {"T:1"}, [](B &b) {
        gen_if(c_leq(b.A(0, 0), 0), [&] { b.seti(0, 0); }, [&] { b.set(0, fn1("log", b.A(0, 0))); });
    });

    R.def("arg", "arg_ri", "nil:1", {"ri:2"},
          [](B &b) { b.set(0, fn1("cargf", fn2("COMPLEX", b.A(0, 0), b.A(0, 1)))); });
    R.def("conj", "conj_ri", "ri:2", {"ri:2"}, [](B &b) { b.setall({b.A(0, 0), -b.A(0, 1)}); });

    R.def("sinh", "sinh_ri", "ri:2", {"ri:2"}, cplx1("csinhf"));
    R.def("sinh", "sinh_1", "T:1", {"T:1"}, real1("sinh"));
    R.def("cosh", "cosh_ri", "ri:2", {"ri:2"}, cplx1("ccoshf"));
    R.def("cosh", "cosh_1", "T:1", {"T:1"}, real1("cosh"));
    R.def("tanh", "tanh_ri", "ri:2", {"ri:2"}, cplx1("ctanhf"));
    R.def("tanh", "tanh_1", "T:1", {"T:1"}, real1("tanh"));
    R.def("asinh", "asinh_ri", "ri:2", {"ri:2"}, cplx1("casinhf"));
    R.def("asinh", "asinh_1", "T:1", {"T:1"}, real1("asinh"));
    R.def("acosh", "acosh_ri", "ri:2", {"ri:2"}, cplx1("cacoshf"));
    R.def("acosh", "acosh_1", "T:1", {"T:1"}, real1("acosh"));
    R.def("atanh", "atanh_ri", "ri:2", {"ri:2"}, cplx1("catanhf"));
    R.def("atanh", "atanh_1", "T:1", {"T:1"}, real1("atanh"));
    R.def("gamma", "gamma_ri", "ri:2", {"ri:2"}, cplx1("cgamma"));
    R.def("gamma", "gamma_1", "T:1", {"T:1"}, [](B &b) {
        gen_if(c_less(b.A(0, 0), 0), [&] { b.seti(0, 0); }, [&] { b.set(0, fn1("GAMMA", b.A(0, 0))); });
    });
    R.def("beta", "beta_1", "T:1", {"T:1", "T:1"}, [](B &b) {
        gen_if(c_or(c_less(b.A(0, 0), 0), c_less(b.A(1, 0), 0)), [&] { b.seti(0, 0); },
               [&] { b.set(0, fn2("gsl_sf_beta", b.A(0, 0), b.A(1, 0))); });
    });

    // ---- elliptic -------------------------------------------------------------------------
    auto simple = [](const char *cname, int n) {
        return [cname, n](B &b) {
            std::vector<E> args;
            for (int i = 0; i < n; ++i) args.push_back(b.A(i, 0));
            b.set(0, opcall(cname, args));
        };
    };
    R.def("ell_int_Kcomp", "ell_int_Kcomp", "T:1", {"T:1"}, simple("ELL_INT_K_COMP", 1));
    R.def("ell_int_Ecomp", "ell_int_Ecomp", "T:1", {"T:1"}, simple("ELL_INT_E_COMP", 1));
    R.def("ell_int_F", "ell_int_F", "T:1", {"T:1", "T:1"}, simple("ELL_INT_F", 2));
    R.def("ell_int_E", "ell_int_E", "T:1", {"T:1", "T:1"}, simple("ELL_INT_E", 2));
    R.def("ell_int_P", "ell_int_P", "T:1", {"T:1", "T:1", "T:1"}, simple("ELL_INT_P", 3));
    R.def("ell_int_D", "ell_int_D", "T:1", {"T:1", "T:1", "T:1"}, simple("ELL_INT_D", 3));
    R.def("ell_int_RC", "ell_int_RC", "T:1", {"T:1", "T:1"}, simple("ELL_INT_RC", 2));
    R.def("ell_int_RD", "ell_int_RD", "T:1", {"T:1", "T:1", "T:1"}, simple("ELL_INT_RD", 3));
    R.def("ell_int_RF", "ell_int_RF", "T:1", {"T:1", "T:1", "T:1"}, simple("ELL_INT_RF", 3));
    R.def("ell_int_RJ", "ell_int_RJ", "T:1", {"T:1", "T:1", "T:1", "T:1"}, simple("ELL_INT_RJ", 4));
    auto tnth = [](E tup, int i) { return opcall("TUPLE_NTH", {tup, lit(i)}); };
    auto jac1 = [tnth](int which) {
        return [tnth, which](B &b) { b.set(0, tnth(fn2("ELL_JAC", b.A(0, 0), b.A(1, 0)), which)); };
    };
    R.def("ell_jac_sn", "ell_jac_sn_1", "T:1", {"T:1", "T:1"}, jac1(0));
    R.def("ell_jac_cn", "ell_jac_cn_1", "T:1", {"T:1", "T:1"}, jac1(1));
    R.def("ell_jac_dn", "ell_jac_dn_1", "T:1", {"T:1", "T:1"}, jac1(2));
    auto jacri = [tnth](int which) {
        return [tnth, which](B &b) {
            E m = b.A(1, 0);
            E v = fn2("ELL_JAC", b.A(0, 0), m);
            E v1 = fn2("ELL_JAC", b.A(0, 1), 1 - m);
            E s = tnth(v, 0), c = tnth(v, 1), d = tnth(v, 2);
            E s1 = tnth(v1, 0), c1 = tnth(v1, 1), d1 = tnth(v1, 2);
            E denom = c1 * c1 + m * ((s * s) * (s1 * s1));
            E rn, in;
            if (which == 0) { rn = s * d1; in = (c * d) * (s1 * c1); }
            else if (which == 1) { rn = c * c1; in = -((s * d) * (s1 * d1)); }
            else { rn = c1 * (d * d1); in = (s * s1) - (m * c); }
            b.set(0, rn / denom);
            b.set(1, in / denom);
        };
    };
    R.def("ell_jac_sn", "ell_jac_sn_ri", "ri:2", {"ri:2", "_:1"}, jacri(0));
    R.def("ell_jac_cn", "ell_jac_cn_ri", "ri:2", {"ri:2", "_:1"}, jacri(1));
    R.def("ell_jac_dn", "ell_jac_dn_ri", "ri:2", {"ri:2", "_:1"}, jacri(2));

    // ---- floor and friends -----------------------------------------------------------------
    R.def("floor", "floor", "T:1", {"T:1"}, real1("floor"));
    R.def("ceil", "ceil", "T:1", {"T:1"}, real1("ceil"));
    R.def("sign", "sign_n", "T:L", {"T:L"}, [](B &b) {
        for (int i = 0; i < b.len(0); ++i)
            gen_if(c_less(b.A(0, i), 0), [&] { b.seti(i, -1); },
                   [&] { gen_if(c_less(0, b.A(0, i)), [&] { b.seti(i, 1); }, [&] { b.seti(i, 0); }); });
    });
    R.def("min", "min_n", "T:L", {"T:L", "T:L"},
          [](B &b) { for (int i = 0; i < b.len(0); ++i) b.set(i, emin(b.A(0, i), b.A(1, i))); });
    R.def("max", "max_n", "T:L", {"T:L", "T:L"},
          [](B &b) { for (int i = 0; i < b.len(0); ++i) b.set(i, emax(b.A(0, i), b.A(1, i))); });
    R.def("clamp", "clamp", "T:L", {"T:L", "T:L", "T:L"}, [](B &b) {
        for (int i = 0; i < b.len(0); ++i)
            gen_if(c_less(b.A(0, i), b.A(1, i)), [&] { b.set(i, b.A(1, i)); },
                   [&] { gen_if(c_less(b.A(2, i), b.A(0, i)), [&] { b.set(i, b.A(2, i)); }, [&] { b.set(i, b.A(0, i)); }); });
    });
    R.def("lerp", "lerp_1", "T:L", {"_:1", "T:L", "T:L"}, [](B &b) {
        E l = 1 - b.A(0, 0);
        for (int i = 0; i < b.len(1); ++i) b.set(i, l * b.A(1, i) + b.A(0, 0) * b.A(2, i));
    });
    R.def("lerp", "lerp_n", "T:L", {"T:L", "T:L", "T:L"}, [](B &b) {
        for (int i = 0; i < b.len(1); ++i) b.set(i, (1 - b.A(0, i)) * b.A(1, i) + b.A(0, i) * b.A(2, i));
    });
    R.def("scale", "scale", "T:L", {"T:L", "T:L", "T:L", "T:L", "T:L"}, [](B &b) {
        for (int i = 0; i < b.len(0); ++i) {
            E div = b.A(2, i) - b.A(1, i);
            gen_if(c_eq(div, 0), [&] { b.seti(i, 0); },
                   [&] { b.set(i, ((b.A(0, i) - b.A(1, i)) / div) * (b.A(4, i) - b.A(3, i)) + b.A(3, i)); });
        }
    });

    // ---- logic --------------------------------------------------------------------------------
    R.def("__not", "not", "T:1", {"T:1"},
          [](B &b) { gen_if(c_eq(b.A(0, 0), 0), [&] { b.seti(0, 1); }, [&] { b.seti(0, 0); }); });
    R.def("__or", "or", "T:1", {"T:1", "T:1"}, [](B &b) {
        gen_if(c_and(c_eq(b.A(0, 0), 0), c_eq(b.A(1, 0), 0)), [&] { b.seti(0, 0); }, [&] { b.seti(0, 1); });
    });
    R.def("__and", "and", "T:1", {"T:1", "T:1"}, [](B &b) {
        gen_if(c_or(c_eq(b.A(0, 0), 0), c_eq(b.A(1, 0), 0)), [&] { b.seti(0, 0); }, [&] { b.seti(0, 1); });
    });
    R.def("__xor", "xor", "T:1", {"T:1", "T:1"}, [](B &b) {
        gen_if(c_or(c_and(c_not(c_eq(b.A(0, 0), 0)), c_eq(b.A(1, 0), 0)),
                    c_and(c_not(c_eq(b.A(1, 0), 0)), c_eq(b.A(0, 0), 0))),
               [&] { b.seti(0, 1); }, [&] { b.seti(0, 0); });
    });

    // ---- comparison -----------------------------------------------------------------------------
    R.def("__equal", "equal_ri", "nil:1", {"ri:2", "ri:2"}, [](B &b) {
        gen_if(c_and(c_eq(b.A(0, 0), b.A(1, 0)), c_eq(b.A(0, 1), b.A(1, 1))), [&] { b.seti(0, 1); }, [&] { b.seti(0, 0); });
    });
    R.def("__equal", "equal_ri_1", "nil:1", {"ri:2", "_:1"}, [](B &b) {
        gen_if(c_and(c_eq(b.A(0, 0), b.A(1, 0)), c_eq(b.A(0, 1), 0)), [&] { b.seti(0, 1); }, [&] { b.seti(0, 0); });
    });
    R.def("__equal", "equal_1_ri", "nil:1", {"_:1", "ri:2"}, [](B &b) {
        gen_if(c_and(c_eq(b.A(1, 0), b.A(0, 0)), c_eq(b.A(1, 1), 0)), [&] { b.seti(0, 1); }, [&] { b.seti(0, 0); });
    });
    auto cmp = [](const char *op, bool swap, bool negate) {
        return [op, swap, negate](B &b) {
            E r = swap ? opcall(op, {b.A(1, 0), b.A(0, 0)}) : opcall(op, {b.A(0, 0), b.A(1, 0)});
            if (negate) r = fn1("NOT", r);
            b.set(0, r);
        };
    };
    R.def("__equal", "equal", "nil:1", {"T:1", "T:1"}, cmp("EQ", false, false));
    R.def("__less", "less", "nil:1", {"T:1", "T:1"}, cmp("LESS", false, false));
    R.def("__greater", "greater", "nil:1", {"T:1", "T:1"}, cmp("LESS", true, false));
    R.def("__lessequal", "lessequal", "nil:1", {"T:1", "T:1"}, cmp("LEQ", false, false));
    R.def("__greaterequal", "greaterequal", "nil:1", {"T:1", "T:1"}, cmp("LEQ", true, false));
    R.def("__notequal", "notequal", "nil:1", {"T:1", "T:1"}, cmp("EQ", false, true));
    R.def("inintv", "inintv", "nil:1", {"T:1", "T:1", "T:1"}, [](B &b) {
        gen_if(c_and(c_leq(b.A(1, 0), b.A(0, 0)), c_leq(b.A(0, 0), b.A(2, 0))), [&] { b.seti(0, 1); }, [&] { b.seti(0, 0); });
    });

    // ---- application ------------------------------------------------------------------------------
    R.def("__applyCurve", "apply_curve", "nil:1", {"curve:1", "_:1"},
          [](B &b) { b.set(0, fn2("APPLY_CURVE", b.A(0, 0), b.A(1, 0))); });
    R.def("__applyGradient", "apply_gradient", "rgba:4", {"gradient:1", "_:1"}, [tnth](B &b) {
        E t = fn2("APPLY_GRADIENT", b.A(0, 0), b.A(1, 0));
        for (int i = 0; i < 4; ++i) b.set(i, tnth(t, i));
    });
    R.def("__origVal", "origValXY", "rgba:4", {"xy:2", "nil:1", "image:1"}, [tnth](B &b) {
        E t = opcall("ORIG_VAL", {b.A(0, 0), b.A(0, 1), b.A(2, 0), b.A(1, 0)});
        for (int i = 0; i < 4; ++i) b.set(i, tnth(t, i));
    });
    R.def("render", "render", "image:1", {"image:1"}, [](B &b) {
        Gen &g = b.g;
        CompVar *w = g.temp(), *h = g.temp();
        g.assign(w, Rhs::Int("__renderPixelW"));
        g.assign(h, Rhs::Int("__renderPixelH"));
        b.set(0, opcall("RENDER", {b.A(0, 0), E(g, w), E(g, h)}));
    });
    R.def("pixelSize", "pixelSize", "xy:2", {"image:1"}, [](B &b) {
        b.setall({fn1("IMAGE_PIXEL_WIDTH", b.A(0, 0)), fn1("IMAGE_PIXEL_HEIGHT", b.A(0, 0))});
    });

    // ---- colours -------------------------------------------------------------------------------------
    auto component = [](int i) { return [i](B &b) { b.set(0, b.A(0, i)); }; };
    R.def("red", "red", "nil:1", {"rgba:4"}, component(0));
    R.def("green", "green", "nil:1", {"rgba:4"}, component(1));
    R.def("blue", "blue", "nil:1", {"rgba:4"}, component(2));
    R.def("alpha", "alpha", "nil:1", {"rgba:4"}, component(3));
    R.def("gray", "gray", "nil:1", {"rgba:4"}, [](B &b) {
        b.set(0, lit(0.299) * b.A(0, 0) + lit(0.587) * b.A(0, 1) + lit(0.114) * b.A(0, 2));
    });
    R.def("rgbColor", "rgbColor", "rgba:4", {"T:1", "T:1", "T:1"},
          [](B &b) { b.set(0, b.A(0, 0)); b.set(1, b.A(1, 0)); b.set(2, b.A(2, 0)); b.seti(3, 1); });
    R.def("rgbaColor", "rgbaColor", "rgba:4", {"T:1", "T:1", "T:1", "T:1"},
          [](B &b) { for (int i = 0; i < 4; ++i) b.set(i, b.A(i, 0)); });
    R.def("grayColor", "grayColor", "rgba:4", {"T:1"},
          [](B &b) { for (int i = 0; i < 3; ++i) b.set(i, b.A(0, 0)); b.seti(3, 1); });
    R.def("grayaColor", "grayaColor", "rgba:4", {"T:1", "T:1"},
          [](B &b) { for (int i = 0; i < 3; ++i) b.set(i, b.A(0, 0)); b.set(3, b.A(1, 0)); });

    R.def("toHSVA", "toHSVA", "hsva:4", {"rgba:4"}, [](B &b) {
        Gen &g = b.g;
        E r = clamp01(b.A(0, 0)), gr = clamp01(b.A(0, 1)), bl = clamp01(b.A(0, 2));
        b.set(3, clamp01(b.A(0, 3)));
        E mx = emax(r, emax(gr, bl));
        E mn = emin(r, emin(gr, bl));
        b.set(2, mx);
        gen_if(c_eq(mx, 0), [&] { b.seti(0, 0); b.seti(1, 0); },
               [&] {
                   E delta = mx - mn;
                   E h = lit(0);
                   b.set(1, delta / mx);
                   gen_if(c_eq(r, mx), [&] { g.copy(h.v, ((gr - bl) / delta).v); },
                          [&] {
                              gen_if(c_eq(gr, mx), [&] { g.copy(h.v, (2 + (bl - r) / delta).v); },
                                     [&] { g.copy(h.v, (4 + (r - gr) / delta).v); });
                          });
                   g.copy(h.v, (h / lit(6.0)).v);
                   gen_if(c_less(h, 0), [&] { b.set(0, h + 1); }, [&] { b.set(0, h); });
               });
    });
    R.def("toRGBA", "toRGBA", "rgba:4", {"hsva:4"}, [](B &b) {
        Gen &g = b.g;
        E s = clamp01(b.A(0, 1)), v = clamp01(b.A(0, 2));
        b.set(3, clamp01(b.A(0, 3)));
        gen_if(c_eq(s, 0), [&] { b.set(0, v); b.set(1, v); b.set(2, v); },
               [&] {
                   E h = emax(lit(0), b.A(0, 0));
                   gen_if(c_leq(1, h), [&] { g.assign(h.v, Rhs::I(0)); }, [&] { g.copy(h.v, (h * 6).v); });
                   E i = fn1("floor", h);
                   E f = h - i;
                   E p = v * (1 - s);
                   E q = v * (1 - s * f);
                   E t = v * (1 - s * (1 - f));
                   auto rgb = [&](E x, E y, E z) { return [&b, x, y, z] { b.set(0, x); b.set(1, y); b.set(2, z); }; };
                   gen_if(c_eq(i, 0), rgb(v, t, p), [&] {
                       gen_if(c_eq(i, 1), rgb(q, v, p), [&] {
                           gen_if(c_eq(i, 2), rgb(p, v, t), [&] {
                               gen_if(c_eq(i, 3), rgb(p, q, v), [&] { gen_if(c_eq(i, 4), rgb(t, p, v), rgb(v, p, q)); });
                           });
                       });
                   });
               });
    });

    // ---- coordinates ------------------------------------------------------------------------------------
    R.def("toXY", "toXY", "xy:2", {"ra:2"}, [](B &b) {
        b.setall({fn1("cos", b.A(0, 1)) * b.A(0, 0), fn1("sin", b.A(0, 1)) * b.A(0, 0)});
    });
    R.def("toXY", "toXY_trivial", "xy:2", {"xy:2"}, [](B &b) { b.setall({b.A(0, 0), b.A(0, 1)}); });
    R.def("toRA", "toRA", "ra:2", {"xy:2"}, [](B &b) {
        E r = fn2("hypot", b.A(0, 0), b.A(0, 1));
        gen_if(c_eq(r, 0), [&] { b.seti(0, 0); b.seti(1, 0); },
               [&] {
                   E a = fn1("acos", b.A(0, 0) / r);
                   b.set(0, r);
                   gen_if(c_less(b.A(0, 1), 0), [&] { b.set(1, (2 * lit((float)M_PI)) - a); }, [&] { b.set(1, a); });
               });
    });
    R.def("toRA", "toRA_trivial", "ra:2", {"ra:2"}, [](B &b) { b.setall({b.A(0, 0), b.A(0, 1)}); });

    R.def("rand", "rand", "T:1", {"T:1", "T:1"}, [](B &b) { b.set(0, fn2("RAND", b.A(0, 0), b.A(1, 0))); });

    // ---- libnoise ---------------------------------------------------------------------------------------
    R.def("noise", "noise_perlin_simple", "nil:1", {"_:3"}, [](B &b) {
        b.set(0, opcall("libnoise_perlin", {lit(1), lit(0), lit(0), b.A(0, 0), b.A(0, 1), b.A(0, 2)}));
    });
    R.def("noise", "noise_perlin_full", "nil:1", {"_:1", "_:1", "_:1", "_:3"}, [](B &b) {
        b.set(0, opcall("libnoise_perlin", {b.A(0, 0), b.A(1, 0), b.A(2, 0), b.A(3, 0), b.A(3, 1), b.A(3, 2)}));
    });
    R.def("noiseBillow", "noise_billow", "nil:1", {"_:1", "_:1", "_:1", "_:3"}, [](B &b) {
        b.set(0, opcall("libnoise_billow", {b.A(0, 0), b.A(1, 0), b.A(2, 0), b.A(3, 0), b.A(3, 1), b.A(3, 2)}));
    });
    R.def("noiseRidgedMulti", "noise_ridged_multi", "nil:1", {"_:1", "_:1", "_:3"}, [](B &b) {
        b.set(0, opcall("libnoise_ridged_multi", {b.A(0, 0), b.A(1, 0), b.A(2, 0), b.A(2, 1), b.A(2, 2)}));
    });
    R.def("voronoiCells", "noise_voronoi", "nil:1", {"_:3"}, [](B &b) {
        b.set(0, opcall("libnoise_voronoi", {lit(1), b.A(0, 0), b.A(0, 1), b.A(0, 2)}));
    });

    // ---- macros (registered after all builtins, macros.c:192-194) ------------------------------------------
    R.macro("__origVal", "rgba:4", {"xy:2", "image:1"},
            [](Parser &p, std::vector<AstNode *> &a, int pos) { return macro_origval(p, a, pos, false); });
    R.macro("__origVal", "rgba:4", {"ra:2", "image:1"},
            [](Parser &p, std::vector<AstNode *> &a, int pos) { return macro_origval(p, a, pos, false); });
    R.macro("__origVal", "rgba:4", {"ra:2", "nil:1", "image:1"},
            [](Parser &p, std::vector<AstNode *> &a, int pos) { return macro_origval(p, a, pos, true); });
}

}  // namespace mm
