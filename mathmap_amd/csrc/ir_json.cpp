// IR (de)serialisation: rebuilds a FilterCode from the JSON produced by dump_ir().
// Lets a compiled filter travel as data (tests/golden/ir/*.json are IR dumps of the
// reference's test-suite filters, produced by tests/make_ir_fixtures.py where the reference
// tree is available) and gives tools an IR-level entry point next to the source-level one.
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "front.h"
#include "runtime_internal.h"

namespace mm {

namespace {

struct J {
    enum K { Null, Num, Str, Arr, Obj } k = Null;
    double num = 0;
    std::string str;
    std::vector<J> arr;
    std::vector<std::pair<std::string, J>> obj;
    const J &operator[](const char *key) const {
        static const J null;
        for (auto &p : obj)
            if (p.first == key) return p.second;
        return null;
    }
    const J &operator[](size_t i) const { return arr.at(i); }
    long i() const { return (long)num; }
};

struct JParser {
    const char *p;
    explicit JParser(const char *s) : p(s) {}
    void ws() { while (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r') ++p; }
    [[noreturn]] void fail(const char *m) { throw CompileError(std::string("IR JSON: ") + m); }
    J parse() {
        ws();
        J j;
        if (*p == '{') {
            j.k = J::Obj;
            ++p;
            ws();
            if (*p == '}') { ++p; return j; }
            for (;;) {
                ws();
                J key = parse();
                if (key.k != J::Str) fail("object key must be a string");
                ws();
                if (*p++ != ':') fail("expected ':'");
                j.obj.emplace_back(key.str, parse());
                ws();
                if (*p == ',') { ++p; continue; }
                if (*p == '}') { ++p; break; }
                fail("expected ',' or '}'");
            }
        } else if (*p == '[') {
            j.k = J::Arr;
            ++p;
            ws();
            if (*p == ']') { ++p; return j; }
            for (;;) {
                j.arr.push_back(parse());
                ws();
                if (*p == ',') { ++p; continue; }
                if (*p == ']') { ++p; break; }
                fail("expected ',' or ']'");
            }
        } else if (*p == '"') {
            j.k = J::Str;
            ++p;
            while (*p && *p != '"') {
                if (*p == '\\' && p[1]) { ++p; j.str += (*p == 'n' ? '\n' : *p); }
                else j.str += *p;
                ++p;
            }
            if (*p != '"') fail("unterminated string");
            ++p;
        } else if (!strncmp(p, "null", 4)) {
            p += 4;
        } else {
            char *end;
            j.k = J::Num;
            j.num = strtod(p, &end);
            if (end == p) fail("unexpected character");
            p = end;
        }
        return j;
    }
};

struct Loader {
    Module &mod;
    FilterCode &code;
    std::map<int, CompVar *> vars;
    std::map<std::pair<int, int>, Value *> vals;
    Loader(Module &m, FilterCode &c) : mod(m), code(c) {}

    static Ty type_of(const std::string &s) {
        static const std::pair<const char *, Ty> t[] = {{"int", Ty::Int}, {"float", Ty::Float}, {"complex", Ty::Complex},
            {"color", Ty::Color}, {"curve", Ty::Curve}, {"gradient", Ty::Gradient}, {"image", Ty::Image},
            {"tuple", Ty::Tuple}, {"tree_vector", Ty::TreeVector}, {"nil", Ty::Nil}};
        for (auto &p : t) if (s == p.first) return p.second;
        throw CompileError("IR JSON: unknown type " + s);
    }

    Value *value(long vid, long idx) {
        auto key = std::make_pair((int)vid, (int)idx);
        auto it = vals.find(key);
        if (it != vals.end()) return it->second;
        CompVar *cv = vars.at((int)vid);
        Value *v;
        if (idx < 0) v = cv->current;
        else { v = code.new_value(cv); v->index = (int)idx; }
        vals[key] = v;
        return v;
    }

    Primary prim(const J &j) {
        const std::string &k = j[(size_t)0].str;
        Primary p;
        if (k == "v") return Primary::V(value(j[1].i(), j[2].i()));
        if (k == "i") return Primary::I((int)j[1].i());
        if (k == "f") { uint32_t b = (uint32_t)j[1].num; float f; memcpy(&f, &b, 4); return Primary::F(f); }
        if (k == "c") { uint32_t a = (uint32_t)j[1].num, b = (uint32_t)j[2].num; p.kind = Primary::ComplexConst; memcpy(&p.f, &a, 4); memcpy(&p.f2, &b, 4); return p; }
        if (k == "k") { p.kind = Primary::ColorConst; p.color = (unsigned)j[1].num; return p; }
        throw CompileError("IR JSON: bad primary");
    }

    Rhs rhs(const J &j) {
        const std::string &k = j["k"].str;
        Rhs r;
        if (k == "prim") return Rhs::P(prim(j["p"]));
        if (k == "internal") return Rhs::Int(j["name"].str);
        auto args = [&](Rhs &out) { for (const J &a : j["args"].arr) out.args.push_back(prim(a)); };
        if (k == "op") {
            const J &a = j["args"];
            const OpInfo *op = op_by_cname(j["op"].str.c_str(), (int)a.arr.size());
            if (!op) throw CompileError("IR JSON: unknown op " + j["op"].str);
            r.kind = Rhs::Op;
            r.op = op;
            args(r);
            return r;
        }
        if (k == "tuple") { r.kind = Rhs::Tuple; args(r); return r; }
        if (k == "treevector") { r.kind = Rhs::TreeVector; args(r); return r; }
        if (k == "closure") {
            r.kind = Rhs::Closure;
            args(r);
            const std::string &native = j["native"].str;
            for (auto &f : mod.filters)
                if ((native.empty() && f->kind == Filter::MathMap && f->name == j["filter"].str) ||
                    (!native.empty() && f->kind == Filter::Native && f->native_func == native))
                    r.filter = f.get();
            if (!r.filter) {   // a MathMap closure only needs a name and its identity
                mod.filters.emplace_back(new Filter());
                r.filter = mod.filters.back().get();
                r.filter->name = j["filter"].str;
            }
            return r;
        }
        if (k == "filtercall") {
            r.kind = Rhs::FilterCall;
            args(r);
            for (auto &f : mod.filters)
                if (f->kind == Filter::MathMap && f->name == j["filter"].str) r.filter = f.get();
            if (!r.filter) throw CompileError("IR JSON: call of unknown filter " + j["filter"].str);
            return r;
        }
        throw CompileError("IR JSON: unsupported rhs kind " + k);
    }

    void phis(const J &list, Block &out, Stmt *parent) {
        for (const J &p : list.arr) {
            Stmt *s = code.new_stmt(Stmt::Phi);
            s->lhs = value(p["lhs"][(size_t)0].i(), p["lhs"][1].i());
            s->lhs->def = s;
            s->rhs = rhs(p["rhs"]);
            s->rhs2 = rhs(p["rhs2"]);
            s->parent = parent;
            out.push_back(s);
        }
    }

    void block(const J &list, Block &out, Stmt *parent) {
        for (const J &j : list.arr) {
            const std::string &k = j["k"].str;
            if (k == "assign") {
                Stmt *s = code.new_stmt(Stmt::Assign);
                s->lhs = value(j["lhs"][(size_t)0].i(), j["lhs"][1].i());
                s->lhs->def = s;
                s->rhs = rhs(j["rhs"]);
                if (j["cid"].k == J::Num) s->closure_id = (int)j["cid"].i();
                s->parent = parent;
                out.push_back(s);
            } else if (k == "if") {
                Stmt *s = code.new_stmt(Stmt::If);
                s->parent = parent;
                s->cond = rhs(j["cond"]);
                block(j["then"], s->then_, s);
                block(j["else"], s->else_, s);
                phis(j["phis"], s->phis, s);
                out.push_back(s);
            } else if (k == "while") {
                Stmt *s = code.new_stmt(Stmt::While);
                s->parent = parent;
                phis(j["phis"], s->phis, s);
                s->cond = rhs(j["cond"]);
                block(j["body"], s->body, s);
                out.push_back(s);
            }
        }
    }

    // body, variables and result of a closure-render dump: same filter (user values) as the main code
    void load_sub(const J &root, Filter *main_filter) {
        code.filter = main_filter;
        for (const J &v : root["vars"].arr) {
            CompVar *cv = code.new_var(type_of(v["type"].str), v["name"].str, (int)v["elem"].i());
            cv->tuple_len = (int)v["tuple_len"].i();
            vars[(int)v["id"].i()] = cv;
        }
        block(root["body"], code.body, nullptr);
        for (int i = 0; i < 4; ++i) {
            const J &r = root["result"][(size_t)i];
            code.result[i] = value(r[(size_t)0].i(), r[1].i());
        }
    }

    void load(const J &root) {
        mod.filters.emplace_back(new Filter());
        Filter *f = mod.filters.back().get();
        f->name = root["filter"].str;
        f->flags = (unsigned)root["flags"].i();
        for (const J &u : root["uservals"].arr) {
            UservalInfo ui;
            ui.index = (int)u["index"].i();
            ui.kind = (UvKind)u["kind"].i();
            ui.name = u["name"].str;
            ui.imin = (int)u["imin"].i(); ui.imax = (int)u["imax"].i(); ui.idef = (int)u["idef"].i();
            ui.fmin = (float)u["fmin"].num; ui.fmax = (float)u["fmax"].num; ui.fdef = (float)u["fdef"].num;
            ui.bdef = u["bdef"].i() != 0;
            ui.image_flags = (unsigned)u["image_flags"].i();
            f->uservals.push_back(ui);
        }
        mod.main = f;
        code.filter = f;
        // filters called at run time: their Filter objects first (calls refer to them by name), bodies below
        std::vector<Filter *> fn_filters;
        for (const J &fn : root["functions"].arr) {
            Filter *ff = nullptr;
            if (fn["filter"].str == f->name) ff = f;            // the main filter calling itself
            else {
                mod.filters.emplace_back(new Filter());
                ff = mod.filters.back().get();
                ff->name = fn["filter"].str;
                ff->flags = (unsigned)fn["flags"].i();
                for (const J &u : fn["uservals"].arr) {
                    UservalInfo ui;
                    ui.index = (int)u["index"].i();
                    ui.kind = (UvKind)u["kind"].i();
                    ui.name = u["name"].str;
                    ui.imin = (int)u["imin"].i(); ui.imax = (int)u["imax"].i(); ui.idef = (int)u["idef"].i();
                    ui.fmin = (float)u["fmin"].num; ui.fmax = (float)u["fmax"].num; ui.fdef = (float)u["fdef"].num;
                    ui.bdef = u["bdef"].i() != 0;
                    ui.image_flags = (unsigned)u["image_flags"].i();
                    ff->uservals.push_back(ui);
                }
            }
            fn_filters.push_back(ff);
        }
        for (size_t i = 0; i < fn_filters.size(); ++i) {
            code.functions.emplace_back(new FilterCode());
            Loader lf(mod, *code.functions.back());
            lf.load_sub(root["functions"][i], fn_filters[i]);
        }
        for (const J &v : root["vars"].arr) {
            CompVar *cv = code.new_var(type_of(v["type"].str), v["name"].str, (int)v["elem"].i());
            cv->tuple_len = (int)v["tuple_len"].i();
            vars[(int)v["id"].i()] = cv;
        }
        block(root["body"], code.body, nullptr);
        for (int i = 0; i < 4; ++i) {
            const J &r = root["result"][(size_t)i];
            code.result[i] = value(r[(size_t)0].i(), r[1].i());
        }
        for (const J &sub : root["closure_renders"].arr) {
            code.closure_renders.emplace_back(new FilterCode());
            Loader ls(mod, *code.closure_renders.back());
            ls.load_sub(sub, f);
        }
    }
};

}  // namespace

void load_ir_json(Module &mod, FilterCode &code, const char *json) {
    JParser jp(json);
    J root = jp.parse();
    Loader l(mod, code);
    l.load(root);
}

}  // namespace mm

extern "C" mmhip_filter *mmhip_compile_ir_json(const char *json, const mmhip_options *opts) {
    mmhip_filter *f = mmhip_filter_new_empty();
    try {
        f->code.reset(new mm::FilterCode());
        mm::load_ir_json(f->module, *f->code, json);
        mm::KernelOptions ko;
        if (opts) {
            ko.intersample = opts->intersample;
            ko.supersampling = opts->supersampling;
            ko.edge_x = opts->edge_behaviour_x;
            ko.edge_y = opts->edge_behaviour_y;
            ko.pixel_inc = opts->pixel_inc > 1 ? opts->pixel_inc : 1;
            if (opts->tile_w) ko.tile_w = opts->tile_w;
        }
        std::string err;
        if (!mmhip_filter_finalize(f, ko, &err)) throw mm::CompileError(err);
        if (opts) {
            f->opts = *opts;
            f->specialize = opts->specialize_uservals != 0;     // per-value-set variants from the IR itself
        }
    } catch (const std::exception &e) {
        g_mmhip_err = e.what();
        mmhip_filter_free(f);
        return nullptr;
    }
    return f;
}
