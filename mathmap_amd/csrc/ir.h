// Structured-SSA intermediate representation of a MathMap filter.
//
// This is the IR the HIP backend lowers to a kernel string.  Its shape follows the
// contract of the reference compiler (compiler-internals.h:56-235): typed compvars
// with SSA values, assignments whose right-hand side is a primary / internal /
// operator application / closure / tuple, and *structured* control flow
// (if with exit phis, while with entry phis).  It is an independent C++ design
// (arena-owned nodes, std::vector blocks); `abi_import.cpp` converts the
// reference's C structs into it for the drop-in boundary.
#pragma once
#include <cstdint>
#include <deque>
#include <memory>
#include <string>
#include <vector>

namespace mm {

// Runtime types, numbered like ops.lisp:37-66 so that "promote to the larger
// type" (compiler.c:2811-2859) is a plain integer max.
enum class Ty : int {
    Nil = 0, Int = 1, Float = 2, Complex = 3, Color = 4, Curve = 5,
    Gradient = 6, Image = 7, Tuple = 8, TreeVector = 9
};
const char *ty_name(Ty t);

enum class TypeProp { Const, Max, MaxFloat };

// One entry per IR operator (ops.lisp:112-253).  `cname` is the C-level macro or
// function name the reference prints (backends/cc.c:210); it is the key the ABI
// importer matches on.
struct OpInfo {
    int index;
    const char *key;     // lisp-ish name, used in dumps
    const char *cname;   // reference C name
    int nargs;
    TypeProp prop;
    Ty type;             // result type if prop == Const
    Ty arg_types[9];
    bool pure;
    bool foldable;
    int tuple_len;       // for ops returning Ty::Tuple: its length (0 = n/a)
};
const OpInfo *op_by_cname(const char *cname, int nargs);
const OpInfo *op_by_key(const char *key, int nargs);
const std::vector<OpInfo> &all_ops();

struct Stmt;
struct Value;
struct Filter;

struct CompVar {
    int id = 0;
    Ty type = Ty::Int;
    bool is_temp = true;
    std::string name;      // user variable name (empty for temporaries)
    int elem = 0;          // tuple element index of a user variable
    int tuple_len = 0;     // only for Ty::Tuple and Ty::TreeVector (its static length)
    Value *current = nullptr;
    std::vector<Value *> values;
};

// Constness lattice (internals.h of the reference): bit set = "does not depend on".
enum : unsigned { CONST_NONE = 0, CONST_X = 1, CONST_Y = 2, CONST_T = 4, CONST_MAX = 7 };

struct Value {
    CompVar *var = nullptr;
    int index = -1;        // SSA version; -1 = uninitialised (reads as 0)
    int gid = 0;           // module-wide id
    Stmt *def = nullptr;
    unsigned constness = CONST_MAX;
    bool hoisted = false;  // lives in the frame-constant block
    bool loop_const = false;  // frame constant per iteration of a loop that also holds per-pixel code: defined in both slices
                              // (passes.cpp mark_dual_loops)
    bool row_const = false;   // depends on the row alone: computed once per row by the rows kernel (hipgen.cpp find_row_slice)
};

struct Primary {
    enum Kind { None, Val, IntConst, FloatConst, ComplexConst, ColorConst } kind = None;
    Value *value = nullptr;
    int i = 0;
    float f = 0.f, f2 = 0.f;
    unsigned color = 0;
    static Primary V(Value *v) { Primary p; p.kind = Val; p.value = v; return p; }
    static Primary I(int v) { Primary p; p.kind = IntConst; p.i = v; return p; }
    static Primary F(float v) { Primary p; p.kind = FloatConst; p.f = v; return p; }
    bool is_const() const { return kind != Val && kind != None; }
    Ty type() const;
};

struct Rhs {
    enum Kind { None, Prim, Internal, Op, Closure, Tuple, FilterCall, TreeVector } kind = None;
    Primary prim;                 // Prim
    std::string internal;         // Internal
    const OpInfo *op = nullptr;   // Op
    std::vector<Primary> args;    // Op / Closure / Tuple / FilterCall / TreeVector
    Filter *filter = nullptr;     // Closure / FilterCall
    static Rhs P(Primary p) { Rhs r; r.kind = Prim; r.prim = p; return r; }
    static Rhs V(Value *v) { return P(Primary::V(v)); }
    static Rhs I(int v) { return P(Primary::I(v)); }
    static Rhs F(float v) { return P(Primary::F(v)); }
    static Rhs Int(const std::string &n) { Rhs r; r.kind = Internal; r.internal = n; return r; }
    static Rhs O(const OpInfo *op, std::vector<Primary> a) { Rhs r; r.kind = Op; r.op = op; r.args = std::move(a); return r; }
    Ty type() const;
};

using Block = std::vector<Stmt *>;

struct Stmt {
    enum Kind { Nil, Assign, Phi, If, While } kind = Nil;
    // Assign / Phi
    Value *lhs = nullptr;
    Rhs rhs, rhs2;            // rhs2 only for Phi (branch 1 / loop back edge)
    Value *old_value = nullptr;
    // If: cond, then_, else_, phis (exit).  While: phis (entry), cond (invariant), body.
    Rhs cond;
    Block then_, else_, phis, body;
    Stmt *parent = nullptr;
    bool hoisted = false;     // belongs (also) to the frame-constant slice
    bool in_pixel = true;     // belongs (also) to the per-pixel slice
    bool in_row = false;      // belongs to the per-row slice (the reference's x-const code, new_template.c.in:251-253)
    // Assign of a MathMap closure that a native filter (or render()) takes as an image: index into
    // FilterCode::closure_renders, the code that renders it into a float map (render_image's closure
    // branch, builtins.c:267-302); -1 otherwise
    int closure_id = -1;
};

// User-value (filter argument) kinds; numbering is ours, the ABI importer maps.
enum class UvKind { Int, Float, Bool, Color, Curve, Gradient, Image };

struct UservalInfo {
    UvKind kind = UvKind::Int;
    std::string name;
    int index = 0;
    int imin = 0, imax = 0, idef = 0;
    float fmin = 0, fmax = 0, fdef = 0;
    bool bdef = false;
    unsigned image_flags = 0;   // IMAGE_FLAG_UNIT|SQUARE of the *argument*
};

enum : unsigned { IMAGE_FLAG_UNIT = 1, IMAGE_FLAG_SQUARE = 2 };

struct AstNode;

struct Filter {
    enum Kind { MathMap, Native } kind = MathMap;
    std::string name;
    std::string native_func;       // e.g. "native_filter_gaussian_blur"
    unsigned flags = IMAGE_FLAG_UNIT | IMAGE_FLAG_SQUARE;
    std::vector<UservalInfo> uservals;
    AstNode *body = nullptr;
    bool uses_ra = false;
    bool uses_t = false;
    int index = 0;
};

// A fully lowered filter: the statement tree of its main function plus the tables
// the backends need.
struct FilterCode {
    Filter *filter = nullptr;
    Block body;
    Value *result[4] = {nullptr, nullptr, nullptr, nullptr};
    std::deque<CompVar> vars;
    std::deque<Value> values;
    std::deque<Stmt> stmts;
    int next_var = 0, next_val = 0;
    // One entry per closure image handed to a native filter: the same filter lowered again with its
    // result replaced by that closure applied at the pixel's own coordinates at t = 0 (what
    // render_image's calc_lines(..., floatmap = 1) launch computes).  The runtime renders it into the
    // native filter's input map before the native filter runs.
    std::vector<std::unique_ptr<FilterCode>> closure_renders;
    // filter_$name of the filters that are called at run time (new_template.c.in:375-422): a call the
    // compiler does not inline -- recursion: the callee is already on the inlining stack, compiler.c:4219-4237 --
    // stays an Rhs::FilterCall (the reference's RHS_FILTER: closure arguments, then x, y, t), and its
    // callee is compiled once as a function of (arguments, x, y, t).  functions[i]->filter is the callee.
    std::vector<std::unique_ptr<FilterCode>> functions;

    CompVar *new_var(Ty t, const std::string &name = "", int elem = 0);
    Value *new_value(CompVar *v);
    Stmt *new_stmt(Stmt::Kind k);
};

void propagate_types(FilterCode &code);
std::string dump_ir(const FilterCode &code);

}  // namespace mm
