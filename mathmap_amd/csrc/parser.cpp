// Recursive-descent parser for the .mm language.  See front.h for the spec sources.
#include <cctype>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "front.h"

namespace mm {

// ---------------------------------------------------------------------------
Tags::Tags() {
    nil = number("nil");
    xy = number("xy");
    ra = number("ra");
    rgba = number("rgba");
    ri = number("ri");
    image = number("image");
    curve = number("curve");
    gradient = number("gradient");
}

int Tags::number(const std::string &name) {
    for (size_t i = 0; i < names_.size(); ++i)
        if (names_[i] == name) return (int)i;
    names_.push_back(name);
    return (int)names_.size() - 1;
}

Variable *FilterVars::lookup(const std::string &name) {
    for (auto &v : vars)
        if (v->name == name) return v.get();
    return nullptr;
}

Variable *FilterVars::add(const std::string &name, TInfo t) {
    vars.emplace_back(new Variable());
    Variable *v = vars.back().get();
    v->name = name;
    v->type = t;
    v->compvar.assign(t.len, nullptr);
    return v;
}

Module::Module() {
    register_builtins(*this);
    register_native_filters();
}

Filter *Module::lookup_filter(const std::string &name) {
    for (auto &f : filters)
        if (f->name == name) return f.get();
    return nullptr;
}

AstNode *Module::node(AstNode::Kind k, TInfo t, int pos) {
    nodes.emplace_back(new AstNode());
    AstNode *n = nodes.back().get();
    n->kind = k;
    n->result = t;
    n->pos = pos;
    return n;
}

bool Module::has_overload(const std::string &name) const {
    for (const BuiltinEntry &e : builtins)
        if (e.name == name) return true;
    return false;
}

// First entry (registration order) whose patterns unify with the argument types
// (overload.c:211-279).  Named pattern variables are shared between the arguments
// of one entry; wildcards match anything.
const BuiltinEntry *Module::resolve(const std::string &name, const std::vector<TInfo> &args, TInfo *result) const {
    for (const BuiltinEntry &e : builtins) {
        if (e.name != name || e.args.size() != args.size()) continue;
        std::map<char, int> bind;
        bool match = true;
        auto unify = [&](const Pat &p, int value) {
            switch (p.kind) {
                case Pat::Const: return p.value == value;
                case Pat::Wild: return true;
                case Pat::Named: {
                    auto it = bind.find(p.name);
                    if (it == bind.end()) { bind[p.name] = value; return true; }
                    return it->second == value;
                }
            }
            return false;
        };
        for (size_t i = 0; i < args.size() && match; ++i) {
            if (!unify(e.args[i].tag, args[i].tag)) match = false;
            else if (!unify(e.args[i].len, args[i].len)) match = false;
        }
        if (!match) continue;
        auto value_of = [&](const Pat &p) {
            if (p.kind == Pat::Const) return p.value;
            auto it = bind.find(p.name);
            if (p.kind == Pat::Named && it != bind.end()) return it->second;
            throw CompileError("internal: unbound result pattern in builtin " + e.id);
        };
        result->tag = value_of(e.result.tag);
        result->len = value_of(e.result.len);
        return &e;
    }
    return nullptr;
}

// Native filters callable from .mm (mathmap_common.c:345-377).
void Module::register_native_filters() {
    auto mk = [&](const char *name, const char *func, std::vector<std::pair<UvKind, const char *>> args) {
        filters.emplace_back(new Filter());
        Filter *f = filters.back().get();
        f->kind = Filter::Native;
        f->name = name;
        f->native_func = func;
        f->index = (int)filters.size() - 1;
        int i = 0;
        for (auto &a : args) {
            UservalInfo u;
            u.kind = a.first;
            u.name = a.second;
            u.index = i++;
            if (u.kind == UvKind::Image) u.image_flags = IMAGE_FLAG_UNIT | IMAGE_FLAG_SQUARE;
            f->uservals.push_back(u);
        }
    };
    mk("gaussian_blur", "native_filter_gaussian_blur",
       {{UvKind::Image, "in"}, {UvKind::Float, "horizontal_std_dev"}, {UvKind::Float, "vertical_std_dev"}});
    mk("convolve", "native_filter_convolve",
       {{UvKind::Image, "in"}, {UvKind::Image, "kernel"}, {UvKind::Bool, "normalize"}, {UvKind::Bool, "copy_alpha"}});
    mk("half_convolve", "native_filter_half_convolve",
       {{UvKind::Image, "in"}, {UvKind::Image, "mask"}, {UvKind::Bool, "copy_alpha"}});
    mk("visualize_fft", "native_filter_visualize_fft", {{UvKind::Image, "in"}, {UvKind::Bool, "ignore_alpha"}});
}

// ---------------------------------------------------------------------------
// Lexer
// ---------------------------------------------------------------------------
enum Tok {
    T_EOF = 0, T_IDENT = 256, T_STRING, T_INT, T_FLOAT, T_RANGE, T_FILTER, T_FLOAT_TYPE, T_INT_TYPE, T_BOOL_TYPE,
    T_COLOR_TYPE, T_GRADIENT_TYPE, T_CURVE_TYPE, T_IMAGE_TYPE, T_IF, T_THEN, T_ELSE, T_END, T_WHILE, T_DO, T_FOR,
    T_XOR, T_EQUAL, T_LESSEQUAL, T_GREATEREQUAL, T_NOTEQUAL, T_OR, T_AND, T_CONVERT
};

struct Token {
    int tok = T_EOF;
    std::string text;
    int ival = 0;
    float fval = 0;
    int pos = 0;
};

class Lexer {
   public:
    explicit Lexer(const std::string &s) : sp_(&s) {}
    Token next();

   private:
    const std::string *sp_;   // pointer (not reference) so the lexer state can be saved/restored
    size_t p_ = 0;
};

Token Lexer::next() {
    const std::string &s_ = *sp_;
    Token t;
restart:
    while (p_ < s_.size() && isspace((unsigned char)s_[p_])) ++p_;
    t.pos = (int)p_;
    if (p_ >= s_.size()) return t;
    char c = s_[p_];
    if (c == '#') {
        while (p_ < s_.size() && s_[p_] != '\n') ++p_;
        goto restart;
    }
    if (c == '_' || isalpha((unsigned char)c)) {
        size_t b = p_;
        while (p_ < s_.size() && (s_[p_] == '_' || isalnum((unsigned char)s_[p_]))) ++p_;
        t.text = s_.substr(b, p_ - b);
        static const std::pair<const char *, int> kw[] = {
            {"filter", T_FILTER}, {"if", T_IF}, {"then", T_THEN}, {"else", T_ELSE}, {"end", T_END},
            {"while", T_WHILE}, {"do", T_DO}, {"for", T_FOR}, {"xor", T_XOR}, {"int", T_INT_TYPE},
            {"float", T_FLOAT_TYPE}, {"bool", T_BOOL_TYPE}, {"color", T_COLOR_TYPE}, {"curve", T_CURVE_TYPE},
            {"gradient", T_GRADIENT_TYPE}, {"image", T_IMAGE_TYPE}};
        t.tok = T_IDENT;
        for (auto &k : kw)
            if (t.text == k.first) t.tok = k.second;
        if (t.text == "function" || t.text == "lambda")
            throw CompileError("`" + t.text + "' is a reserved keyword", t.pos);
        return t;
    }
    if (c == '"') {
        size_t b = ++p_;
        while (p_ < s_.size() && s_[p_] != '"') ++p_;
        if (p_ >= s_.size()) throw CompileError("String not terminated", t.pos);
        t.tok = T_STRING;
        t.text = s_.substr(b, p_ - b);
        ++p_;
        return t;
    }
    if (c == '.' || isdigit((unsigned char)c)) {
        // ".." is the range token; "1..3" lexes as INT RANGE INT (scanner.c:286-336)
        if (c == '.' && p_ + 1 < s_.size() && s_[p_ + 1] == '.') {
            p_ += 2;
            t.tok = T_RANGE;
            return t;
        }
        size_t b = p_;
        bool dot = false, digits = false;
        while (p_ < s_.size()) {
            char d = s_[p_];
            if (isdigit((unsigned char)d)) { digits = true; ++p_; }
            else if (d == '.') {
                if (p_ + 1 < s_.size() && s_[p_ + 1] == '.' ) break;   // start of a range
                if (dot) break;
                dot = true;
                ++p_;
            } else break;
        }
        if (!digits) throw CompileError("Misplaced decimal point", t.pos);
        t.text = s_.substr(b, p_ - b);
        if (dot) { t.tok = T_FLOAT; t.fval = (float)strtod(t.text.c_str(), nullptr); }
        else { t.tok = T_INT; t.ival = atoi(t.text.c_str()); }
        return t;
    }
    static const char *singles = "-<>!,()+*/%=;^:[]";
    char d = p_ + 1 < s_.size() ? s_[p_ + 1] : 0;
    static const std::pair<const char *, int> two[] = {{"==", T_EQUAL}, {"<=", T_LESSEQUAL}, {">=", T_GREATEREQUAL},
                                                        {"!=", T_NOTEQUAL}, {"||", T_OR}, {"&&", T_AND},
                                                        {"::", T_CONVERT}};
    for (auto &k : two)
        if (c == k.first[0] && d == k.first[1]) {
            t.tok = k.second;
            t.text = k.first;
            p_ += 2;
            return t;
        }
    if (strchr(singles, c)) {
        t.tok = c;
        t.text = std::string(1, c);
        ++p_;
        return t;
    }
    throw CompileError("Illegal character", t.pos);
}

// ---------------------------------------------------------------------------
// Parser
// ---------------------------------------------------------------------------
class Parser {
   public:
    Parser(Module &m, const std::string &src) : m_(m), lex_(src) { advance(); }
    void parse_filters();

    // tree constructors (also used by macros)
    AstNode *make_int(int v, int pos);
    AstNode *make_float(float v, int pos);
    AstNode *make_var(const std::string &name, int pos);
    AstNode *make_tuple(std::vector<AstNode *> elems, int pos);
    AstNode *make_cast(const std::string &tag, AstNode *e, int pos);
    AstNode *make_function(const std::string &name, std::vector<AstNode *> args, int pos);
    AstNode *make_assignment(const std::string &name, AstNode *value, int pos);
    AstNode *make_sequence(AstNode *l, AstNode *r);
    Module &module() { return m_; }
    Filter *filter() { return cur_; }
    FilterVars &fvars() { return *m_.vars[cur_]; }

   private:
    Module &m_;
    Lexer lex_;
    Token tok_;
    Filter *cur_ = nullptr;

    void advance() { tok_ = lex_.next(); }
    bool is(int t) const { return tok_.tok == t; }
    void expect(int t, const char *what) {
        if (!is(t)) throw CompileError(std::string("Parse error: expected ") + what, tok_.pos);
        advance();
    }
    [[noreturn]] void fail(const std::string &msg, int pos) { throw CompileError(msg, pos); }

    void parse_filter();
    std::vector<std::string> parse_options();
    void parse_arg_decl(Filter *f);
    bool parse_number(bool *is_float, int *iv, float *fv);

    // precedence levels, low -> high (parser.y:52-61)
    AstNode *parse_seq();       // ';'
    AstNode *parse_assignlevel();
    AstNode *parse_logic();     // || && xor
    AstNode *parse_compare();
    AstNode *parse_additive();
    AstNode *parse_mult();
    AstNode *parse_pow();
    AstNode *parse_unary();
    AstNode *parse_primary();
    std::vector<AstNode *> parse_subscripts();
    void skip_optional_semicolon_before(int t1, int t2 = -1);

    AstNode *make_userval(const UservalInfo *info, std::vector<AstNode *> args, int pos);
    AstNode *make_filter_call(Filter *f, std::vector<AstNode *> args, int pos);
    AstNode *make_image_call(AstNode *image, std::vector<AstNode *> args, int pos);
    AstNode *make_select(AstNode *tuple, std::vector<AstNode *> subs, int pos);
    AstNode *make_sub_assignment(const std::string &name, std::vector<AstNode *> subs, AstNode *value, int pos);
    AstNode *make_binop(const char *fn, AstNode *l, AstNode *r, int pos) { return make_function(fn, {l, r}, pos); }
    const UservalInfo *lookup_userval(const std::string &name);
};

static bool is_internal_name(const std::string &n) {
    static const char *names[] = {"x", "y", "r", "a", "t", "R", "frame", "X", "Y", "W", "H", "__canvasPixelW",
                                  "__canvasPixelH", "__renderPixelW", "__renderPixelH"};
    for (const char *s : names)
        if (n == s) return true;
    return false;
}

const UservalInfo *Parser::lookup_userval(const std::string &name) {
    for (const UservalInfo &u : cur_->uservals)
        if (u.name == name) return &u;
    return nullptr;
}

AstNode *Parser::make_int(int v, int pos) {
    AstNode *n = m_.node(AstNode::IntConst, {m_.tags.nil, 1}, pos);
    n->ival = v;
    return n;
}

AstNode *Parser::make_float(float v, int pos) {
    AstNode *n = m_.node(AstNode::FloatConst, {m_.tags.nil, 1}, pos);
    n->fval = v;
    return n;
}

AstNode *Parser::make_tuple(std::vector<AstNode *> elems, int pos) {
    for (AstNode *e : elems)
        if (e->result.len != 1) fail("Tuples cannot contain tuples of length other than 1.", e->pos);
    AstNode *n = m_.node(AstNode::Tuple, {m_.tags.nil, (int)elems.size()}, pos);
    n->kids = std::move(elems);
    return n;
}

AstNode *Parser::make_cast(const std::string &tag, AstNode *e, int pos) {
    AstNode *n = m_.node(AstNode::Cast, {m_.tags.number(tag), e->result.len}, pos);
    n->kids = {e};
    return n;
}

// Identifier resolution order: internal, variable macro, user value, variable
// (exprtree.c:674-718).
AstNode *Parser::make_var(const std::string &name, int pos) {
    if (is_internal_name(name)) {
        AstNode *n = m_.node(AstNode::Internal, {m_.tags.nil, 1}, pos);
        n->name = name;
        n->owner = cur_;
        if (name == "r" || name == "a") cur_->uses_ra = true;
        if (name == "t") cur_->uses_t = true;
        return n;
    }
    if (name == "xy") return make_cast("xy", make_tuple({make_var("x", pos), make_var("y", pos)}, pos), pos);
    if (name == "ra") return make_cast("ra", make_tuple({make_var("r", pos), make_var("a", pos)}, pos), pos);
    if (name == "XY") return make_cast("xy", make_tuple({make_var("X", pos), make_var("Y", pos)}, pos), pos);
    if (name == "WH") return make_cast("xy", make_tuple({make_var("W", pos), make_var("H", pos)}, pos), pos);
    if (name == "I") return make_cast("ri", make_tuple({make_int(0, pos), make_int(1, pos)}, pos), pos);
    if (name == "pi") return make_float((float)M_PI, pos);
    if (name == "e") return make_float((float)M_E, pos);
    if (const UservalInfo *u = lookup_userval(name)) return make_userval(u, {}, pos);
    if (Variable *v = fvars().lookup(name)) {
        AstNode *n = m_.node(AstNode::Var, v->type, pos);
        n->var = v;
        return n;
    }
    fail("Undefined variable " + name + ".", pos);
}

// exprtree.c:562-659
AstNode *Parser::make_userval(const UservalInfo *info, std::vector<AstNode *> args, int pos) {
    TInfo t{m_.tags.nil, 1};
    switch (info->kind) {
        case UvKind::Int:
        case UvKind::Float:
        case UvKind::Bool:
            if (!args.empty()) fail("Number, bool and color inputs take no arguments.", pos);
            break;
        case UvKind::Color:
            if (!args.empty()) fail("Number, bool and color inputs take no arguments.", pos);
            t = {m_.tags.rgba, 4};
            break;
        case UvKind::Curve: t = {m_.tags.curve, 1}; break;
        case UvKind::Gradient: t = {m_.tags.gradient, 1}; break;
        case UvKind::Image: t = {m_.tags.image, 1}; break;
    }
    AstNode *n = m_.node(AstNode::Userval, t, pos);
    n->uv = info;
    n->owner = cur_;
    switch (info->kind) {
        case UvKind::Curve:
            if (args.size() == 1) return make_function("__applyCurve", {n, args[0]}, pos);
            if (!args.empty()) fail("A curve takes one argument.", pos);
            break;
        case UvKind::Gradient:
            if (args.size() == 1) return make_function("__applyGradient", {n, args[0]}, pos);
            if (!args.empty()) fail("A gradient takes one argument.", pos);
            break;
        case UvKind::Image:
            if (args.size() == 1 || args.size() == 2) {
                args.push_back(n);
                return make_function("__origVal", args, pos);
            }
            if (!args.empty()) fail("An image takes one or two arguments.", pos);
            break;
        default: break;
    }
    return n;
}

// exprtree.c:819-935
AstNode *Parser::make_filter_call(Filter *f, std::vector<AstNode *> args, int pos) {
    int nuv = (int)f->uservals.size();
    int nargs = (int)args.size();
    if (nargs < nuv || nargs >= nuv + 3)
        fail("Filter " + f->name + " takes " + std::to_string(nuv) + " to " + std::to_string(nuv + 2) +
                 " arguments but is called with " + std::to_string(nargs) + ".", pos);
    for (int i = 0; i < nuv; ++i) {
        if (f->uservals[i].kind == UvKind::Color) {
            if (args[i]->result.tag != m_.tags.rgba || args[i]->result.len != 4)
                fail("Can only pass tuples of type rgba:4 as colors.", args[i]->pos);
        } else if (args[i]->result.len != 1)
            fail("Can only pass tuples of length 1 as numbers, booleans, curves, gradients, or images.", args[i]->pos);
    }
    AstNode *closure = m_.node(AstNode::Closure, {m_.tags.image, 1}, pos);
    closure->filter = f;
    closure->owner = cur_;
    closure->kids.assign(args.begin(), args.begin() + nuv);
    if (nargs == nuv) return closure;

    std::vector<AstNode *> call(args.begin() + nuv, args.end());
    if (call[0]->result.len != 2 || (call[0]->result.tag != m_.tags.xy && call[0]->result.tag != m_.tags.ra))
        fail("The coordinate argument to a filter must be a tuple of type xy:2 or ra:2.", call[0]->pos);
    if (call[0]->result.tag == m_.tags.ra) call[0] = make_function("toXY", {call[0]}, call[0]->pos);
    if (call.size() == 1) call.push_back(make_var("t", pos));
    else if (call[1]->result.len != 1) fail("The time argument to a filter must be a tuple of length 1.", call[1]->pos);
    call.push_back(closure);
    return make_function("__origVal", call, pos);
}

// exprtree.c:937-976
AstNode *Parser::make_image_call(AstNode *image, std::vector<AstNode *> args, int pos) {
    if (args.size() != 1 && args.size() != 2) fail("An image must be invoked with one or two arguments.", pos);
    if (args[0]->result.len != 2 || (args[0]->result.tag != m_.tags.xy && args[0]->result.tag != m_.tags.ra))
        fail("The coordinate argument to an image must be of type xy:2 or ra:2.", pos);
    if (args[0]->result.tag == m_.tags.ra) args[0] = make_function("toXY", {args[0]}, args[0]->pos);
    if (args.size() == 2 && args[1]->result.len != 1) fail("The time argument to an image have length 1.", pos);
    args.push_back(image);
    return make_function("__origVal", args, pos);
}

// exprtree.c:1038-1127
AstNode *Parser::make_function(const std::string &name, std::vector<AstNode *> args, int pos) {
    if (const UservalInfo *u = lookup_userval(name)) return make_userval(u, args, pos);
    if (Filter *f = m_.lookup_filter(name)) return make_filter_call(f, args, pos);

    std::vector<TInfo> types;
    for (AstNode *a : args) types.push_back(a->result);
    TInfo res;
    if (!args.empty()) {
        if (const BuiltinEntry *e = m_.resolve(name, types, &res)) {
            if (e->macro) return e->macro(*this, args, pos);
            AstNode *n = m_.node(AstNode::Func, res, pos);
            n->entry = e;
            n->kids = std::move(args);
            return n;
        }
    }
    if (Variable *v = fvars().lookup(name)) {
        if (v->type.tag != m_.tags.image || v->type.len != 1)
            fail("Variable " + name + " is not an image and cannot be invoked.", pos);
        AstNode *n = m_.node(AstNode::Var, v->type, pos);
        n->var = v;
        return make_image_call(n, args, pos);
    }
    static const std::pair<const char *, const char *> ops[] = {
        {"__add", "+"}, {"__sub", "-"}, {"__mul", "*"}, {"__div", "/"}, {"__mod", "%"}, {"__pow", "^"},
        {"__equal", "=="}, {"__less", "<"}, {"__greater", ">"}, {"__lessequal", "<="}, {"__greaterequal", ">="},
        {"__notequal", "!="}, {"__or", "||"}, {"__and", "&&"}, {"__xor", "xor"}, {"__neg", "-"}, {"__not", "!"}};
    for (auto &o : ops)
        if (name == o.first) fail(std::string("Unable to resolve invocation of operator `") + o.second + "'.", pos);
    fail("Unable to resolve invocation of function `" + name + "'.", pos);
}

AstNode *Parser::make_sequence(AstNode *l, AstNode *r) {
    AstNode *n = m_.node(AstNode::Seq, r->result, l->pos);
    n->kids = {l, r};
    return n;
}

// exprtree.c:1175-1216
AstNode *Parser::make_assignment(const std::string &name, AstNode *value, int pos) {
    Variable *v = fvars().lookup(name);
    if (!v) {
        if (is_internal_name(name) || name == "xy" || name == "ra" || name == "XY" || name == "WH" || name == "I" ||
            name == "pi" || name == "e")
            fail("Cannot assign to internal variable `" + name + "'.", pos);
        if (lookup_userval(name)) fail("Cannot assign to filter argument `" + name + "'.", pos);
        v = fvars().add(name, value->result);
    }
    if (v->type != value->result) fail("Variable " + name + " is being assigned two different types.", pos);
    AstNode *n = m_.node(AstNode::Assign, v->type, pos);
    n->var = v;
    n->kids = {value};
    return n;
}

AstNode *Parser::make_sub_assignment(const std::string &name, std::vector<AstNode *> subs, AstNode *value, int pos) {
    Variable *v = fvars().lookup(name);
    if (!v) fail("Undefined variable " + name + ".", pos);
    if ((int)subs.size() != value->result.len) fail("Lhs does not match rhs in sub assignment.", pos);
    AstNode *n = m_.node(AstNode::SubAssign, value->result, pos);
    n->var = v;
    n->subs = std::move(subs);
    n->kids = {value};
    return n;
}

AstNode *Parser::make_select(AstNode *tuple, std::vector<AstNode *> subs, int pos) {
    TInfo t = subs.size() == 1 ? TInfo{m_.tags.nil, 1} : TInfo{tuple->result.tag, (int)subs.size()};
    for (AstNode *s : subs)
        if (s->result.len != 1) fail("Tuples cannot contain tuples of length other than 1.", s->pos);
    AstNode *n = m_.node(AstNode::Select, t, pos);
    n->kids = {tuple};
    n->subs = std::move(subs);
    return n;
}

// ---------------------------------------------------------------------------
std::vector<std::string> Parser::parse_options() {
    // options are plain identifiers, optionally with parenthesised sub-options
    std::vector<std::string> opts;
    while (is(T_IDENT)) {
        opts.push_back(tok_.text);
        advance();
        if (is('(')) {
            advance();
            parse_options();
            expect(')', "`)'");
        }
    }
    return opts;
}

static unsigned flags_from_options(const std::vector<std::string> &opts) {
    bool pixel = false, stretched = false;
    for (auto &o : opts) {
        if (o == "pixel") pixel = true;
        if (o == "stretched") stretched = true;
    }
    unsigned f = 0;
    if (!pixel) {
        f |= IMAGE_FLAG_UNIT;
        if (!stretched) f |= IMAGE_FLAG_SQUARE;
    }
    return f;
}

bool Parser::parse_number(bool *is_float, int *iv, float *fv) {
    bool neg = false;
    if (is('-')) { neg = true; advance(); }
    if (is(T_INT)) {
        *is_float = false;
        *iv = neg ? -tok_.ival : tok_.ival;
        *fv = (float)*iv;
        advance();
        return true;
    }
    if (is(T_FLOAT)) {
        *is_float = true;
        *fv = neg ? -tok_.fval : tok_.fval;
        *iv = (int)*fv;
        advance();
        return true;
    }
    fail("Parse error.", tok_.pos);
}

void Parser::parse_arg_decl(Filter *f) {
    std::vector<std::string> opts = parse_options();
    UservalInfo u;
    int ty = tok_.tok;
    int pos = tok_.pos;
    switch (ty) {
        case T_INT_TYPE: u.kind = UvKind::Int; u.imin = -100000; u.imax = 100000; u.idef = 0; break;
        case T_FLOAT_TYPE: u.kind = UvKind::Float; u.fmin = -1.f; u.fmax = 1.f; u.fdef = 0.f; break;
        case T_BOOL_TYPE: u.kind = UvKind::Bool; break;
        case T_COLOR_TYPE: u.kind = UvKind::Color; break;
        case T_GRADIENT_TYPE: u.kind = UvKind::Gradient; break;
        case T_CURVE_TYPE: u.kind = UvKind::Curve; break;
        case T_IMAGE_TYPE: u.kind = UvKind::Image; u.image_flags = flags_from_options(opts); break;
        case T_FILTER: fail("filter-typed arguments are not supported", pos);
        default: fail("Parse error.", pos);
    }
    advance();
    if (!is(T_IDENT)) fail("Parse error.", tok_.pos);
    u.name = tok_.text;
    for (const UservalInfo &o : f->uservals)
        if (o.name == u.name) fail("The argument `" + u.name + "' is declared more than once.", tok_.pos);
    advance();
    bool have_limits = false;
    if (is(':')) {   // limits_opt
        advance();
        bool f1, f2;
        int i1, i2;
        float v1, v2;
        int lpos = tok_.pos;
        parse_number(&f1, &i1, &v1);
        expect('-', "`-'");
        parse_number(&f2, &i2, &v2);
        have_limits = true;
        if (u.kind == UvKind::Int) {
            if (f1 || f2) fail("Only integers can be limits for an int argument", lpos);
            if (i1 >= i2) fail("Lower limit must be less than upper limit", lpos);
            u.imin = i1; u.imax = i2; u.idef = i1;
        } else if (u.kind == UvKind::Float) {
            if (v1 >= v2) fail("Lower limit must be less than upper limit", lpos);
            u.fmin = v1; u.fmax = v2; u.fdef = v1;
        } else
            fail("Limits applied to wrongly typed argument", lpos);
    }
    (void)have_limits;
    if (is('(')) {   // default_opt
        advance();
        bool fl;
        int iv;
        float fv;
        int dpos = tok_.pos;
        parse_number(&fl, &iv, &fv);
        expect(')', "`)'");
        if (u.kind == UvKind::Int) {
            if (fl) fail("Only integers can be defaults for an int argument", dpos);
            if (iv < u.imin || iv > u.imax) fail("Default value outside of bounds", dpos);
            u.idef = iv;
        } else if (u.kind == UvKind::Float) {
            if (fv < u.fmin || fv > u.fmax) fail("Default value outside of bounds", dpos);
            u.fdef = fv;
        } else if (u.kind == UvKind::Bool) {
            if (fl) fail("Only integers can be defaults for a bool argument", dpos);
            u.bdef = iv != 0;
        } else
            fail("Default applied to wrongly typed argument", dpos);
    }
    if (is(T_STRING)) advance();   // docstring
    u.index = (int)f->uservals.size();
    f->uservals.push_back(u);
}

void Parser::parse_filter() {
    std::vector<std::string> opts = parse_options();
    expect(T_FILTER, "`filter'");
    if (!is(T_IDENT)) fail("Parse error.", tok_.pos);
    m_.filters.emplace_back(new Filter());
    Filter *f = m_.filters.back().get();
    f->name = tok_.text;
    f->index = (int)m_.filters.size() - 1;
    f->flags = flags_from_options(opts);
    m_.vars[f].reset(new FilterVars());
    advance();
    expect('(', "`('");
    if (!is(')')) {
        parse_arg_decl(f);
        while (is(',')) {
            advance();
            parse_arg_decl(f);
        }
    }
    expect(')', "`)'");
    if (is(T_STRING)) advance();
    cur_ = f;
    AstNode *body = parse_seq();
    if (is(';')) advance();
    expect(T_END, "`end'");
    f->body = body;
    if (body->result.tag != m_.tags.rgba || body->result.len != 4)
        fail("The filter `" + f->name + "' must have the result type rgba:4.", body->pos);
    cur_ = nullptr;
    m_.main = f;
}

void Parser::parse_filters() {
    while (!is(T_EOF)) parse_filter();
}

// `;` may precede `end` / `else` (parser.y:265-271)
void Parser::skip_optional_semicolon_before(int, int) {}

AstNode *Parser::parse_seq() {
    AstNode *l = parse_assignlevel();
    while (is(';')) {
        // a ';' directly before end/else belongs to the enclosing construct
        Lexer save = lex_;
        Token tsave = tok_;
        advance();
        if (is(T_END) || is(T_ELSE)) { lex_ = save; tok_ = tsave; break; }
        AstNode *r = parse_assignlevel();
        l = make_sequence(l, r);
    }
    return l;
}

AstNode *Parser::parse_assignlevel() { return parse_logic(); }

AstNode *Parser::parse_logic() {
    AstNode *l = parse_compare();
    for (;;) {
        const char *fn = is(T_OR) ? "__or" : is(T_AND) ? "__and" : is(T_XOR) ? "__xor" : nullptr;
        if (!fn) return l;
        int pos = tok_.pos;
        advance();
        l = make_binop(fn, l, parse_compare(), pos);
    }
}

AstNode *Parser::parse_compare() {
    AstNode *l = parse_additive();
    for (;;) {
        const char *fn = is(T_EQUAL) ? "__equal" : is('<') ? "__less" : is('>') ? "__greater"
                       : is(T_LESSEQUAL) ? "__lessequal" : is(T_GREATEREQUAL) ? "__greaterequal"
                       : is(T_NOTEQUAL) ? "__notequal" : nullptr;
        if (!fn) return l;
        int pos = tok_.pos;
        advance();
        l = make_binop(fn, l, parse_additive(), pos);
    }
}

AstNode *Parser::parse_additive() {
    AstNode *l = parse_mult();
    for (;;) {
        const char *fn = is('+') ? "__add" : is('-') ? "__sub" : nullptr;
        if (!fn) return l;
        int pos = tok_.pos;
        advance();
        l = make_binop(fn, l, parse_mult(), pos);
    }
}

AstNode *Parser::parse_mult() {
    AstNode *l = parse_pow();
    for (;;) {
        const char *fn = is('*') ? "__mul" : is('/') ? "__div" : is('%') ? "__mod" : nullptr;
        if (!fn) return l;
        int pos = tok_.pos;
        advance();
        l = make_binop(fn, l, parse_pow(), pos);
    }
}

AstNode *Parser::parse_pow() {   // right associative, binds weaker than unary
    AstNode *l = parse_unary();
    if (is('^')) {
        int pos = tok_.pos;
        advance();
        return make_binop("__pow", l, parse_pow(), pos);
    }
    return l;
}

AstNode *Parser::parse_unary() {
    if (is('-') || is('!')) {
        const char *fn = is('-') ? "__neg" : "__not";
        int pos = tok_.pos;
        advance();
        return make_function(fn, {parse_unary()}, pos);
    }
    return parse_primary();
}

std::vector<AstNode *> Parser::parse_subscripts() {
    std::vector<AstNode *> subs;
    for (;;) {
        if (is(T_INT)) {
            // INT RANGE INT ?
            Lexer save = lex_;
            Token tsave = tok_;
            int first = tok_.ival, pos = tok_.pos;
            advance();
            if (is(T_RANGE)) {
                advance();
                if (!is(T_INT)) fail("Parse error.", tok_.pos);
                int last = tok_.ival;
                advance();
                if (first > last) fail("Invalid range " + std::to_string(first) + ".." + std::to_string(last) + ".", pos);
                for (int i = first; i <= last; ++i) subs.push_back(make_int(i, pos));
                goto next;
            }
            lex_ = save;
            tok_ = tsave;
        }
        subs.push_back(parse_seq());
    next:
        if (is(',')) { advance(); continue; }
        break;
    }
    return subs;
}

AstNode *Parser::parse_primary() {
    int pos = tok_.pos;
    switch (tok_.tok) {
        case T_INT: { AstNode *n = make_int(tok_.ival, pos); advance(); return n; }
        case T_FLOAT: { AstNode *n = make_float(tok_.fval, pos); advance(); return n; }
        case '[': {
            advance();
            std::vector<AstNode *> elems;
            elems.push_back(parse_seq());
            while (is(',')) { advance(); elems.push_back(parse_seq()); }
            expect(']', "`]'");
            return make_tuple(elems, pos);
        }
        case '(': {
            advance();
            AstNode *e = parse_seq();
            expect(')', "`)'");
            if (is('[')) {
                advance();
                std::vector<AstNode *> subs = parse_subscripts();
                expect(']', "`]'");
                return make_select(e, subs, pos);
            }
            return e;
        }
        case T_IF: {
            advance();
            AstNode *c = parse_seq();
            expect(T_THEN, "`then'");
            AstNode *a = parse_seq();
            if (is(';')) advance();
            if (c->result.len != 1) fail("Condition to if statement must have length 1.", c->pos);
            if (is(T_ELSE)) {
                advance();
                AstNode *b = parse_seq();
                if (is(';')) advance();
                expect(T_END, "`end'");
                if (a->result != b->result)
                    fail("Consequent and alternative must have the same type in if statement.", a->pos);
                AstNode *n = m_.node(AstNode::IfThenElse, a->result, pos);
                n->kids = {c, a, b};
                return n;
            }
            expect(T_END, "`end'");
            AstNode *n = m_.node(AstNode::IfThen, a->result, pos);
            n->kids = {c, a};
            return n;
        }
        case T_WHILE: {
            advance();
            AstNode *c = parse_seq();
            expect(T_DO, "`do'");
            AstNode *b = parse_seq();
            if (is(';')) advance();
            expect(T_END, "`end'");
            if (c->result.len != 1) fail("Invariant of while loop must have length 1.", c->pos);
            AstNode *n = m_.node(AstNode::While, {m_.tags.nil, 1}, pos);
            n->kids = {c, b};
            return n;
        }
        case T_DO: {
            advance();
            AstNode *b = parse_seq();
            expect(T_WHILE, "`while'");
            AstNode *c = parse_seq();
            if (is(';')) advance();
            expect(T_END, "`end'");
            if (c->result.len != 1) fail("Invariant of do-while loop must have length 1.", c->pos);
            AstNode *n = m_.node(AstNode::DoWhile, {m_.tags.nil, 1}, pos);
            n->kids = {c, b};
            return n;
        }
        case T_FOR: {   // for v = a .. b do body end  (exprtree.c:1354-1385)
            advance();
            if (!is(T_IDENT)) fail("Parse error.", tok_.pos);
            std::string counter = tok_.text;
            int cpos = tok_.pos;
            advance();
            expect('=', "`='");
            AstNode *start = parse_logic();
            if (start->result.len != 1)
                fail("The start and end of a for loop interval must be tuples of length 1.", start->pos);
            AstNode *counter_init = make_assignment(counter, start, cpos);
            expect(T_RANGE, "`..'");
            AstNode *end = parse_logic();
            expect(T_DO, "`do'");
            AstNode *body = parse_seq();
            if (is(';')) advance();
            expect(T_END, "`end'");
            if (end->result.len != 1 || start->result.tag != end->result.tag)
                fail("The start and end of a for loop interval must be tuples of the same tag and length 1.", cpos);
            std::string end_name = "___tmp___" + std::to_string(fvars().tmp_counter++) + "___";
            AstNode *end_init = make_assignment(end_name, end, cpos);
            AstNode *init = make_sequence(counter_init, end_init);
            AstNode *inc = make_assignment(counter, make_function("__add", {make_var(counter, cpos), make_int(1, cpos)}, cpos), cpos);
            AstNode *inv = make_function("__lessequal", {make_var(counter, cpos), make_var(end_name, cpos)}, cpos);
            AstNode *loop = m_.node(AstNode::While, {m_.tags.nil, 1}, pos);
            loop->kids = {inv, make_sequence(body, inc)};
            return make_sequence(init, loop);
        }
        case T_IDENT: {
            std::string name = tok_.text;
            advance();
            if (is(':')) {   // tag cast, binds tighter than unary operators
                advance();
                AstNode *e = parse_unary();
                return make_cast(name, e, pos);
            }
            if (is(T_CONVERT)) fail("tag conversion (::) is not implemented", pos);
            if (is('(')) {
                advance();
                std::vector<AstNode *> args;
                if (!is(')')) {
                    args.push_back(parse_seq());
                    while (is(',')) { advance(); args.push_back(parse_seq()); }
                }
                expect(')', "`)'");
                if (args.empty() && !lookup_userval(name) && !m_.lookup_filter(name))
                    fail("Unable to resolve invocation of function `" + name + "'.", pos);
                return make_function(name, args, pos);
            }
            if (is('=')) {
                advance();
                AstNode *v = parse_logic();   // '=' is right associative and binds tighter than ';'
                return make_assignment(name, v, pos);
            }
            if (is('[')) {
                advance();
                std::vector<AstNode *> subs = parse_subscripts();
                expect(']', "`]'");
                if (is('=')) {
                    advance();
                    AstNode *v = parse_logic();
                    return make_sub_assignment(name, subs, v, pos);
                }
                return make_select(make_var(name, pos), subs, pos);
            }
            return make_var(name, pos);
        }
        default: fail("Parse error.", pos);
    }
}

void parse_module(Module &m, const std::string &source) {
    Parser p(m, source);
    p.parse_filters();
    if (!m.main) throw CompileError("No filter defined.", 0);
}

// Macro bodies for __origVal(xy|ra, image) and __origVal(ra, t, image)  (macros.c:138-178)
AstNode *macro_origval(Parser &p, std::vector<AstNode *> &args, int pos, bool with_frame) {
    Module &m = p.module();
    std::string tmp = "___tmp___" + std::to_string(p.fvars().tmp_counter++) + "___";
    AstNode *assign = p.make_assignment(tmp, args[0], pos);
    AstNode *xy = p.make_function("toXY", {p.make_var(tmp, pos)}, pos);
    std::vector<AstNode *> call;
    call.push_back(xy);
    if (with_frame) { call.push_back(args[1]); call.push_back(args[2]); }
    else { call.push_back(p.make_var("t", pos)); call.push_back(args[1]); }
    (void)m;
    return p.make_sequence(assign, p.make_function("__origVal", call, pos));
}

}  // namespace mm
