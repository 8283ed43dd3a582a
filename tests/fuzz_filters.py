"""Seeded random .mm filters for differential testing (HIP kernel generator vs oracle C printer,
specialised vs generic kernels).  Not a copy of any reference script: expressions are built from
the language's operators and builtins at random, with bounded loops and guarded domains."""
import random

SCALAR_FUNCS_1 = ["sin", "cos", "abs", "floor", "ceil", "sqrt_abs", "exp_c", "log_abs", "atan", "tanh", "sign"]
SCALAR_FUNCS_2 = ["min", "max", "atan2", "hyp"]


class Gen:
    def __init__(self, seed):
        self.r = random.Random(seed)
        self.vars = []           # scalar variable names in scope
        self.lines = []
        self.nvar = 0
        self.uses_image = False

    def lit(self):
        r = self.r
        return r.choice(["0.5", "2", "3", "0.25", "1.5", "7", "0.1", "pi", "%d" % r.randint(1, 9), "%.3f" % r.uniform(0.01, 4)])

    def atom(self):
        r = self.r
        c = r.random()
        if c < 0.45 or not self.vars:
            return r.choice(["x", "y", "r", "a", "t", "x", "y", "k", "m"])
        if c < 0.7:
            return r.choice(self.vars)
        return self.lit()

    def scalar(self, depth):
        r = self.r
        if depth <= 0 or r.random() < 0.2:
            return self.atom()
        c = r.random()
        if c < 0.45:
            op = r.choice(["+", "-", "*", "*", "+", "/", "%"])
            a, b = self.scalar(depth - 1), self.scalar(depth - 1)
            if op in "/%":
                b = "(abs(%s) + 0.5)" % b
            return "(%s %s %s)" % (a, op, b)
        if c < 0.7:
            f = r.choice(SCALAR_FUNCS_1)
            a = self.scalar(depth - 1)
            if f == "sqrt_abs":
                return "sqrt(abs(%s))" % a
            if f == "exp_c":
                return "exp(min(%s, 3))" % a
            if f == "log_abs":
                return "log(abs(%s) + 0.01)" % a
            return "%s(%s)" % (f, a)
        if c < 0.82:
            f = r.choice(SCALAR_FUNCS_2)
            a, b = self.scalar(depth - 1), self.scalar(depth - 1)
            if f == "hyp":
                return "abs(ri:[%s, %s])" % (a, b)
            if f == "atan2":
                return "atan(%s, %s)" % (a, b)
            return "%s(%s, %s)" % (f, a, b)
        if c < 0.93:
            return "(if %s then %s else %s end)" % (self.cond(depth - 1), self.scalar(depth - 1), self.scalar(depth - 1))
        if self.uses_image_ok and c < 1.0:
            self.uses_image = True
            ch = r.choice(["red", "green", "blue", "gray"])
            return "%s(in(xy:[%s, %s]))" % (ch, self.scalar(depth - 2), self.scalar(depth - 2))
        return self.atom()

    def cond(self, depth):
        r = self.r
        a, b = self.scalar(depth), self.scalar(depth)
        c = "%s %s %s" % (a, r.choice(["<", ">", "<=", ">=", "=="]), b)
        if r.random() < 0.3:
            c = "(%s) %s (%s %s %s)" % (c, r.choice(["&&", "||"]), self.scalar(depth), r.choice(["<", ">"]), self.scalar(depth))
        return c

    def new_var(self, expr):
        name = "v%d" % self.nvar
        self.nvar += 1
        self.lines.append("%s = %s;" % (name, expr))
        self.vars.append(name)
        return name

    def statement(self):
        r = self.r
        c = r.random()
        if c < 0.6:
            self.new_var(self.scalar(3))
        elif c < 0.8:
            # bounded loop: accumulate
            acc = self.new_var(self.scalar(1))
            n = "n%d" % self.nvar
            self.nvar += 1
            lim = r.choice(["k", "3", "5", "(k % 4 + 1)"])
            self.lines.append("%s = 0; while %s < %s do %s = %s * 0.5 + %s; %s = %s + 1 end;" % (
                n, n, lim, acc, acc, self.scalar(2), n, n))
        else:
            v = r.choice(self.vars) if self.vars else self.new_var(self.atom())
            self.lines.append("if %s then %s = %s else %s = %s end;" % (self.cond(2), v, self.scalar(2), v, self.scalar(2)))


def make_filter_ex(seed):
    """Richer variant: filter flags (pixel / stretched), a second image of another size, complex
    arithmetic, a helper filter applied as a closure, tuple arithmetic on fetched colours.  Returns
    (source, image names, render options)."""
    g = Gen(seed ^ 0x5eed)
    r = g.r
    g.uses_image_ok = True
    flag = r.choice(["", "", "pixel ", "stretched "])
    for _ in range(r.randint(1, 4)):
        g.statement()
    scale = "W" if flag == "pixel " else "1"
    kind = r.random()
    images = ["in"]
    helper = ""
    if kind < 0.3:
        images.append("in2")
        result = "in(xy) * %s + in2(xy * %s + xy:[%s, %s] * %s * 0.05) * (1 - %s)" % (
            "0.5", r.choice(["0.5", "1", "1.5"]), g.scalar(2), g.scalar(2), scale, "0.5")
    elif kind < 0.55:
        z = "ri:[%s, %s]" % (g.scalar(2), g.scalar(2))
        op = r.choice(["%s * %s", "%s + %s", "exp(%s * 0.3) + %s", "sin(%s) * %s", "sqrt(%s) - %s", "%s / (%s + ri:[2, 0.5])"])
        zz = op % (z, "ri:[%s, 0.5]" % g.scalar(1))
        g.lines.append("zc = %s;" % zz)
        result = "in(xy + xy:[zc[0], zc[1]] * %s * 0.02)" % scale
    elif kind < 0.8:
        helper = "filter hlp%d (image im, float s: 0-2 (1))\n  im(xy * s) * 0.5 + grayColor(%s) * 0.5\nend\n\n" % (seed, "abs(sin(x * 3 + s))")
        result = "hlp%d(in, %s, xy + xy:[%s, 0] * %s * 0.03)" % (seed, r.choice(["0.7", "1", "m"]), g.scalar(2), scale)
    else:
        result = "lerp(%s, in(xy), rgba:[%s, %s, 0.5, 1])" % ("0.5", g.scalar(2), g.scalar(2))
    params = ["image in"] + (["image in2"] if "in2" in images else []) + ["int k: 0-8 (3)", "float m: 0-2 (0.7)"]
    src = "%s%sfilter fx%d (%s)\n  %s\n  %s\nend\n" % (helper, flag, seed, ", ".join(params), "\n  ".join(g.lines), result)
    src = src.replace("\\n", "\n")
    opts = dict(intersample=r.random() < 0.7, edge_x=r.choice([0, 0, 1, 2, 3]), edge_y=r.choice([0, 0, 1, 2, 3]))
    return src, images, opts


def make_filter(seed, with_image=True):
    g = Gen(seed)
    g.uses_image_ok = with_image
    for _ in range(g.r.randint(2, 6)):
        g.statement()
    kind = g.r.random()
    if with_image and kind < 0.45:
        g.uses_image = True
        result = "in(xy + xy:[%s, %s] * 0.1) * %s" % (g.scalar(2), g.scalar(2), g.r.choice(["1", "0.8", "(0.5 + 0.5 * sin(t))"]))
    elif kind < 0.75:
        result = "rgba:[%s, %s, %s, 1]" % (g.scalar(3), g.scalar(3), g.scalar(2))
    else:
        result = "grayColor(%s)" % g.scalar(3)
    params = ["int k: 0-8 (3)", "float m: 0-2 (0.7)"]
    if g.uses_image:
        params.insert(0, "image in")
    src = "filter fz%d (%s)\n  %s\n  %s\nend\n" % (seed, ", ".join(params), "\n  ".join(g.lines), result)
    return src, g.uses_image


def make_filter_arith(seed):
    """Arithmetic-only filters (+ - * /, comparisons, if, bounded and data-dependent while loops, int
    counters): the class the generator evaluates two pixels at a time in lockstep (hipgen.cpp pair mode)."""
    r = random.Random(seed ^ 0xa217)
    lines = []
    names = ["x", "y"]

    def atom():
        c = r.random()
        if c < 0.55:
            return r.choice(names)
        return r.choice(["0.5", "2", "3", "0.25", "1.5", "0.1", "%.3f" % r.uniform(0.05, 3), "k", "m"])

    def expr(d):
        if d <= 0 or r.random() < 0.25:
            return atom()
        op = r.choice(["+", "-", "*", "*", "+", "/"])
        a, b = expr(d - 1), expr(d - 1)
        if op == "/":
            b = "(%s * %s + 0.5)" % (b, b)
        return "(%s %s %s)" % (a, op, b)

    def cond(d):
        c = "%s %s %s" % (expr(d), r.choice(["<", ">", "<=", ">=", "=="]), expr(d))
        if r.random() < 0.4:
            c = "(%s) %s (%s %s %s)" % (c, r.choice(["&&", "||"]), expr(d), r.choice(["<", ">"]), expr(d))
        return c

    for i in range(r.randint(2, 5)):
        v = "v%d" % i
        c = r.random()
        if c < 0.4:
            lines.append("%s = %s;" % (v, expr(3)))
        elif c < 0.65:
            lines.append("%s = if %s then %s else %s end;" % (v, cond(2), expr(2), expr(2)))
        else:
            # escape-time style loop: data-dependent exit, bounded by a counter
            n = "n%d" % i
            lines.append("%s = %s; %s = 0; while (%s * %s < %s) && (%s < %s) do %s = %s * %s * 0.5 + %s; %s = %s + 1 end;" % (
                v, expr(1), n, v, v, r.choice(["4", "9", "2.5"]), n, r.choice(["k + 2", "7", "12"]), v, v, r.choice([v, expr(1)]),
                expr(1), n, n))
            lines.append("%s = %s + %s * 0.1;" % (v, v, n))
        names.append(v)
    result = "rgba:[%s, %s, %s, 1]" % (expr(2), expr(2), names[-1])
    return "filter fa%d (int k: 0-8 (3), float m: 0-2 (0.7))\n  %s\n  %s\nend\n" % (seed, "\n  ".join(lines), result)
