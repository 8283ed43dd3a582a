// Front-end of the .mm filter language: tokens, typed expression tree, module.
//
// Behavioural spec (not code) taken from the reference: token rules scanner.c:219-387,
// grammar/precedence parser.y:52-271, tree construction + name resolution
// exprtree.c:674-1385, overload unification overload.c:211-279, variable macros
// macros.c:95-195.  Implemented as a hand-written recursive-descent parser that
// types the tree while it is built (variables are typed by their first assignment,
// exactly like the reference's single bison pass).
#pragma once
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "ir.h"

namespace mm {

struct CompileError : std::runtime_error {
    int pos;
    // lowering with baked-in user values unrolls recursion; this is set when that unrolling did not end
    // (lower_filter then lowers again with run-time filter_$name calls)
    bool recursion_limit = false;
    CompileError(const std::string &m, int p = -1) : std::runtime_error(m), pos(p) {}
};

struct TInfo {
    int tag = 0;
    int len = 1;
    bool operator==(const TInfo &o) const { return tag == o.tag && len == o.len; }
    bool operator!=(const TInfo &o) const { return !(*this == o); }
};

// Interned tuple tags (tags.c:52-63): nil xy ra rgba ri image curve gradient first.
class Tags {
   public:
    Tags();
    int number(const std::string &name);
    const std::string &name(int n) const { return names_[n]; }
    int nil, xy, ra, rgba, ri, image, curve, gradient;

   private:
    std::vector<std::string> names_;
};

struct Variable {
    std::string name;
    TInfo type;
    std::vector<CompVar *> compvar;   // one per element (reset on every inlining)
    bool is_vector = false;           // dynamically subscripted -> tree vector
};

struct BuiltinEntry;
class Gen;

struct AstNode {
    enum Kind {
        IntConst, FloatConst, Tuple, Select, Var, Internal, Assign, SubAssign, Cast, Func,
        Seq, IfThen, IfThenElse, While, DoWhile, Userval, Closure
    } kind = IntConst;
    TInfo result;
    int pos = 0;
    int ival = 0;
    float fval = 0.f;
    std::string name;                  // Internal
    std::vector<AstNode *> kids;       // operands / args / elements
    std::vector<AstNode *> subs;       // subscripts (Select / SubAssign)
    Variable *var = nullptr;
    const BuiltinEntry *entry = nullptr;
    const UservalInfo *uv = nullptr;
    Filter *filter = nullptr;          // Closure
    Filter *owner = nullptr;           // filter this node belongs to
};

// Pattern element of a builtin signature: constant, named variable or wildcard.
struct Pat {
    enum Kind { Const, Named, Wild } kind = Wild;
    int value = 0;     // tag number or length
    char name = 0;     // 'T', 'L', ...
};
struct ArgPat { Pat tag, len; };

using BuiltinGen = std::function<void(Gen &g, const std::vector<std::vector<CompVar *>> &args,
                                      const std::vector<TInfo> &arg_types, std::vector<CompVar *> &result)>;
using MacroFn = std::function<AstNode *(class Parser &, std::vector<AstNode *> &args, int pos)>;

struct BuiltinEntry {
    std::string name;       // overloaded name ("__add", "sin", ...)
    std::string id;         // unique name ("add_ri", ...)
    ArgPat result;
    std::vector<ArgPat> args;
    BuiltinGen gen;         // empty for macros
    MacroFn macro;
};

struct FilterVars {
    std::vector<std::unique_ptr<Variable>> vars;
    int tmp_counter = 0;
    Variable *lookup(const std::string &name);
    Variable *add(const std::string &name, TInfo t);
};

struct Module {
    Tags tags;
    std::vector<std::unique_ptr<Filter>> filters;
    std::map<Filter *, std::unique_ptr<FilterVars>> vars;
    std::vector<std::unique_ptr<AstNode>> nodes;
    std::vector<BuiltinEntry> builtins;
    Filter *main = nullptr;

    Module();
    Filter *lookup_filter(const std::string &name);
    AstNode *node(AstNode::Kind k, TInfo t, int pos);
    const BuiltinEntry *resolve(const std::string &name, const std::vector<TInfo> &args, TInfo *result) const;
    bool has_overload(const std::string &name) const;
    void register_native_filters();
};

void register_builtins(Module &m);   // builtins.cpp

// Parses `source` (one or more filters) into the module; the last filter becomes
// `main` (mathmap_common.c:474).  Throws CompileError.
void parse_module(Module &m, const std::string &source);

// Lowers filter `f` (with all callees inlined) to IR.  lower.cpp
// `uv_consts` (optional): user-value index -> literal to bake in instead of the run-time
// USERVAL_*_ACCESS read (user-value specialisation, specialize.cpp).
std::unique_ptr<FilterCode> lower_filter(Module &m, Filter *f, const std::map<int, Primary> *uv_consts = nullptr);
// The body of filter_$name (new_template.c.in:375-422): user values read from the call's argument block, x y t as
// internals.  What lower_filter puts into FilterCode::functions; also what the reference hands a backend as the
// filter_code of a filter other than the main one (tests/libmathmap_hip_selftest exports it in that role).
std::unique_ptr<FilterCode> lower_function(Module &m, Filter *f);

}  // namespace mm
