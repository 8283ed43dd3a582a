"""CPU ORACLE (test infrastructure, NOT product code): IR -> C, like the reference's cc backend.

Takes the IR dump of a compiled filter (``mmhip_filter_ir_json``) and prints it as C in
the shape the reference's ``backends/cc.c`` + ``new_template.c.in`` produce:

* one C variable per SSA value, phi copies at the end of branches / loop bodies
  (cc.c:308-397);
* ``init_frame`` evaluating the frame-constant statements once into an ``xy_vars``
  struct (new_template.c.in:314-337) and ``calc_lines`` with the row/column double loop,
  ``CALC_VIRTUAL_X/Y`` in double and the byte packing of new_template.c.in:243-309.

The text is compiled exactly as the reference compiles its generated code --
``gcc -O2 -c -fPIC`` + ``gcc -shared`` (reference Makefile:58-60, cc.c:673-681) -- and
linked with oracle/mm_oracle_rt.c.  libm / complex libm calls go to the host's glibc
just as in the reference.  This is the "port" CPU baseline of bench.py and the checker
of the parity tests; it is an independent second implementation of statement printing
(the product's is mathmap_amd/csrc/hipgen.cpp).
"""
import ctypes as C
import hashlib
import json
import os
import struct
import subprocess
import threading

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# MM_ORACLE_SANITIZE=1: the oracle's own C (runtime + generated filters) under AddressSanitizer + UBSan, in a build
# directory of its own; run python with LD_PRELOAD=$(gcc -print-file-name=libasan.so) (tools/asan_oracle.sh)
SANITIZE = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-g"] if os.environ.get("MM_ORACLE_SANITIZE") else []
BUILD = os.path.join(HERE, "_build_san" if SANITIZE else "_build")

CTYPES = {"int": "int", "float": "float", "complex": "float _Complex", "color": "color_t",
          "curve": "int", "gradient": "int", "image": "mmo_image"}

UNSUPPORTED = {"SOLVE_POLY_2", "SOLVE_POLY_3"}
NOISE_OPS = {"libnoise_perlin", "libnoise_billow", "libnoise_ridged_multi", "libnoise_voronoi"}
NOISE_LIB = os.path.join(HERE, "_ref", "libmmnoise.so")

# The reference folds every pure op whose arguments are all literals itself, at MathMap compile
# time, by running the op's macro on the host -- i.e. with glibc (compiler.c:3383-3458
# constant_folding / the generated fold_rhs; all libm and complex ops are foldable, ops.lisp:88-108).
# gcc therefore never sees `cexpf(literal)` in the reference's generated C.  This printer does not
# fold, so gcc would -- with MPFR/MPC, correctly rounded, 1-2 float ulps away from glibc's float
# complex functions.  Declaring the libm functions non-builtin makes gcc call glibc for them, which
# is the value the reference's own folding produced.  (sqrt / floor / ceil / fabs / fmod stay builtin:
# their results are exact or correctly rounded either way.)
LIBM_NO_BUILTIN = ["-fno-builtin-" + f for f in (
    "sin cos tan asin acos atan atan2 pow exp log sinh cosh tanh asinh acosh atanh hypot "
    "csqrtf csinf ccosf ctanf casinf cacosf catanf cexpf clogf cpowf csinhf ccoshf ctanhf casinhf cacoshf "
    "catanhf cargf cabsf sincos").split()]


class OracleUnsupported(Exception):
    pass


def _float_literal(bits):
    f = struct.unpack("<f", struct.pack("<I", bits))[0]
    if f != f:
        return "(0.0/0.0)"
    if f in (float("inf"), float("-inf")):
        return "(1.0/0.0)" if f > 0 else "(-1.0/0.0)"
    s = repr(float(f))
    if "." not in s and "e" not in s and "n" not in s:
        s += ".0"
    return "(%s)" % s if s.startswith("-") else s


class Gen:
    def __init__(self, ir, functions=None, in_function=False):
        self.ir = ir
        self.vars = {v["id"]: v for v in ir["vars"]}
        self.uses_noise = False
        # filter_$name bodies this code may call ("functions" of the root dump), by filter name
        self.functions = functions if functions is not None else ir.get("functions", [])
        self.in_function = in_function
        self.natives = {}      # id(stmt) -> slot
        self.nnative = 0
        self._find_natives(ir["body"])

    # same traversal order as hipgen.cpp:find_natives so slot numbers agree
    def _find_natives(self, block):
        for s in block:
            if s["k"] == "assign":
                r = s["rhs"]
                if (r["k"] == "closure" and r["native"]) or (r["k"] == "op" and r["op"] == "RENDER"):
                    self.natives[id(s)] = self.nnative
                    self.nnative += 1
            elif s["k"] == "if":
                self._find_natives(s["then"])
                self._find_natives(s["else"])
            elif s["k"] == "while":
                self._find_natives(s["body"])

    def ctype(self, vid):
        v = self.vars[vid]
        if v["type"] == "tuple":
            return "mmo_tup%d" % (v["tuple_len"] or 4)
        if v["type"] == "tree_vector":
            return "mmo_tup%d" % v["tuple_len"]
        return CTYPES[v["type"]]

    def vname(self, vid, idx):
        return "v%d_%d" % (vid, idx)

    def prim(self, p):
        k = p[0]
        if k == "v":
            vid, idx = p[1], p[2]
            if idx < 0:
                t = self.vars[vid]["type"]
                if t == "image":
                    return "UNINITED_IMAGE"
                if t in ("tuple", "tree_vector"):
                    return "(%s){{0}}" % self.ctype(vid)
                return "0"
            return self.vname(vid, idx)
        if k == "i":
            return "(%d)" % p[1] if p[1] < 0 else str(p[1])
        if k == "f":
            return _float_literal(p[1])
        if k == "c":
            return "COMPLEX(%s,%s)" % (_float_literal(p[1]), _float_literal(p[2]))
        if k == "k":
            return "%du" % p[1]
        raise ValueError(p)

    def rhs(self, r, stmt=None):
        k = r["k"]
        if k == "prim":
            return self.prim(r["p"])
        if k == "internal":
            return r["name"]
        if k in ("tuple", "treevector"):      # backends/cc.c:268-300: float tuple[n] = { args }
            return "(mmo_tup%d){{%s}}" % (len(r["args"]), ", ".join(self.prim(a) for a in r["args"]))
        if k == "closure":
            if not r["native"]:
                return "mmo_closure_image(A, %d)" % (stmt.get("cid", -1) if stmt else -1)
            slot = self.natives[id(stmt)]
            args = ", ".join(self.prim(a) for a in r["args"])
            fn = {"native_filter_gaussian_blur": "mmo_native_gaussian_blur",
                  "native_filter_convolve": "mmo_native_convolve",
                  "native_filter_half_convolve": "mmo_native_half_convolve",
                  "native_filter_visualize_fft": "mmo_native_visualize_fft"}.get(r["native"])
            if fn is None:
                raise OracleUnsupported("native filter %s" % r["native"])
            return "%s(A, %d, %s)" % (fn, slot, args)
        if k == "filtercall":
            # backends/cc.c:221-235: the callee's arguments become its user values, then filter_$name(x, y, t)
            names = [fn["filter"] for fn in self.functions]
            fk = names.index(r["filter"])
            uvs = self.functions[fk]["uservals"]
            args = r["args"]
            field = {1: "f", 3: "c", 6: "img"}
            fill = " ".join("mm_ca[%d].%s = %s;" % (u["index"], field.get(u["kind"], "i"), self.prim(a)) for u, a in zip(uvs, args))
            x, y, t = (self.prim(a) for a in args[len(uvs):])
            depth = "mm_depth + 1" if self.in_function else "0"
            return "({ mmo_uvarg mm_ca[%d]; %s mmo_filter_%d(A, mm_ca, %s, %s, %s, col, row, &mm_rand_ctr, %s); })" % (
                max(1, len(uvs)), fill, fk, x, y, t, depth)
        if k == "op":
            op = r["op"]
            if self.in_function and op.startswith("USERVAL_") and op.endswith("_ACCESS"):
                f = {"USERVAL_FLOAT_ACCESS": "f", "USERVAL_COLOR_ACCESS": "c", "USERVAL_IMAGE_ACCESS": "img"}.get(op, "i")
                return "(UV[%d].%s)" % (r["args"][0][1], f)
            if op in UNSUPPORTED:
                raise OracleUnsupported("op %s" % op)
            if op in NOISE_OPS:
                if not os.path.exists(NOISE_LIB):
                    raise OracleUnsupported("op %s (oracle/_ref/libmmnoise.so not built: no reference tree)" % op)
                self.uses_noise = True
            if op in ("TREE_VECTOR_NTH", "SET_TREE_VECTOR_NTH"):
                # opmacros.h:189-190 (the index converts to the C int parameter; the vector's length is static)
                a = r["args"]
                n = self.vars[a[1][1]]["tuple_len"]
                if op == "TREE_VECTOR_NTH":
                    return "({ mmo_tup%d tv_ = %s; mmo_tv_nth((int)(%s), tv_.v, %d); })" % (n, self.prim(a[1]), self.prim(a[0]), n)
                return "({ mmo_tup%d tv_ = %s; mmo_tv_set((int)(%s), tv_.v, %d, %s); tv_; })" % (
                    n, self.prim(a[1]), self.prim(a[0]), n, self.prim(a[2]))
            if op == "RENDER":
                return "mmo_render(A, %d, %s)" % (self.natives[id(stmt)], ", ".join(self.prim(a) for a in r["args"]))
            return "%s(%s)" % (op, ",".join(self.prim(a) for a in r["args"]))
        raise OracleUnsupported("rhs kind %s" % k)

    # ---- slices ------------------------------------------------------------------
    def mine(self, s, hoisted):
        return s["hoisted"] if hoisted else s["pixel"]

    def collect(self, block, hoisted, defs, uses):
        def use(r):
            if r is None:
                return
            if r["k"] == "prim" and r["p"][0] == "v":
                uses.add((r["p"][1], r["p"][2]))
            for a in r.get("args", []):
                if a[0] == "v":
                    uses.add((a[1], a[2]))
        for s in block:
            if not self.mine(s, hoisted):
                continue
            if s["k"] == "assign":
                defs.append(tuple(s["lhs"]))
                use(s["rhs"])
            elif s["k"] == "if":
                use(s["cond"])
                self.collect(s["then"], hoisted, defs, uses)
                self.collect(s["else"], hoisted, defs, uses)
                self.collect_phis(s["phis"], hoisted, defs, uses, use)
            elif s["k"] == "while":
                self.collect_phis(s["phis"], hoisted, defs, uses, use)
                use(s["cond"])
                self.collect(s["body"], hoisted, defs, uses)

    def collect_phis(self, phis, hoisted, defs, uses, use):
        for p in phis:
            if self.mine(p, hoisted):
                defs.append(tuple(p["lhs"]))
                use(p["rhs"])
                use(p["rhs2"])

    def phis(self, phis, branch, hoisted, ind, out):
        mine = []
        for p in phis:
            if not self.mine(p, hoisted):
                continue
            r = p["rhs"] if branch == 0 else p["rhs2"]
            if r["k"] == "prim" and r["p"][0] == "v" and [r["p"][1], r["p"][2]] == p["lhs"]:
                continue
            mine.append((p, r))
        targets = {tuple(p["lhs"]) for p, _ in mine}
        hazard = any(r["k"] == "prim" and r["p"][0] == "v" and (r["p"][1], r["p"][2]) in targets for _, r in mine)
        if not hazard:
            for p, r in mine:
                out.append("%s%s = %s;" % (ind, self.vname(*p["lhs"]), self.rhs(r)))
            return
        out.append(ind + "{")
        for i, (p, r) in enumerate(mine):
            out.append("%s  %s pc%d = %s;" % (ind, self.ctype(p["lhs"][0]), i, self.rhs(r)))
        for i, (p, r) in enumerate(mine):
            out.append("%s  %s = pc%d;" % (ind, self.vname(*p["lhs"]), i))
        out.append(ind + "}")

    def stmts(self, block, hoisted, ind, out):
        for s in block:
            if not self.mine(s, hoisted):
                continue
            if s["k"] == "assign":
                out.append("%s%s = %s;" % (ind, self.vname(*s["lhs"]), self.rhs(s["rhs"], s)))
            elif s["k"] == "if":
                out.append("%sif (%s) {" % (ind, self.rhs(s["cond"])))
                self.stmts(s["then"], hoisted, ind + "  ", out)
                self.phis(s["phis"], 0, hoisted, ind + "  ", out)
                out.append(ind + "} else {")
                self.stmts(s["else"], hoisted, ind + "  ", out)
                self.phis(s["phis"], 1, hoisted, ind + "  ", out)
                out.append(ind + "}")
            elif s["k"] == "while":
                self.phis(s["phis"], 0, hoisted, ind, out)
                out.append("%swhile (%s) {" % (ind, self.rhs(s["cond"])))
                self.stmts(s["body"], hoisted, ind + "  ", out)
                self.phis(s["phis"], 1, hoisted, ind + "  ", out)
                out.append(ind + "}")

    def decls(self, defs, ind, out, skip=()):
        seen = set()
        for d in defs:
            if d in seen or d[1] < 0 or d in skip:
                continue
            seen.add(d)
            t = self.ctype(d[0])
            init = " = 0" if t in ("int", "float", "color_t") else ""
            out.append("%s%s %s%s;" % (ind, t, self.vname(*d), init))

    def function_source(self, out):
        """filter_$name of every filter called at run time (new_template.c.in:375-422: a C function per filter,
        called through the closure).  Ordinary recursive C, as in the reference, except that depth
        MM_MAX_CALL_DEPTH returns zeros -- the HIP path's stack bound, restated so both cut off alike."""
        if not self.functions:
            return
        out.append("typedef struct { int i; float f; color_t c; mmo_image img; } mmo_uvarg;")
        out.append("#ifndef MM_MAX_CALL_DEPTH\n#define MM_MAX_CALL_DEPTH %d\n#endif" % int(os.environ.get("MMHIP_MAX_CALL_DEPTH", "16")))
        sig = "static mmo_tup4 mmo_filter_%d(const mmo_args *A, const mmo_uvarg *UV, float x, float y, float t, int col, int row, unsigned *mm_rand_p, int mm_depth)"
        for k in range(len(self.functions)):
            out.append(sig % k + ";")
        out.append("#define mm_rand_ctr (*mm_rand_p)")
        for k, fn in enumerate(self.functions):
            g = Gen(fn, functions=self.functions, in_function=True)
            defs, uses = [], set()
            g.collect(fn["body"], False, defs, uses)
            out.append("/* filter_%s */" % fn["filter"])
            out.append(sig % k + " {")
            out.append("  mmo_tup4 rt = {{0, 0, 0, 0}};")
            out.append("  if (mm_depth >= MM_MAX_CALL_DEPTH) return rt;")
            out.append("  const float R = A->R; const int frame = 0;      /* new_template.c.in:379 */")
            out.append("  const int __canvasPixelW = A->img_width, __canvasPixelH = A->img_height;")
            out.append("  const int __renderPixelW = A->render_width, __renderPixelH = A->render_height;")
            out.append("  (void)R; (void)frame; (void)__canvasPixelW; (void)__canvasPixelH; (void)__renderPixelW; (void)__renderPixelH;")
            g.decls(defs, "  ", out)
            g.stmts(fn["body"], False, "  ", out)
            for i, r in enumerate(fn["result"]):
                out.append("  rt.v[%d] = %s;" % (i, g.prim(["v", r[0], r[1]])))
            out.append("  return rt;\n}")
            self.uses_noise = self.uses_noise or g.uses_noise
        out.append("#undef mm_rand_ctr\n")

    def source(self):
        ir = self.ir
        pro_defs, pro_uses, pix_defs, pix_uses = [], set(), [], set()
        self.collect(ir["body"], True, pro_defs, pro_uses)
        self.collect(ir["body"], False, pix_defs, pix_uses)
        for r in ir["result"]:
            pix_uses.add(tuple(r))
        pix_def_set = set(pix_defs)
        transfers = []
        for d in pro_defs:
            if d in pix_uses and d not in pix_def_set and d not in transfers:
                transfers.append(d)
        out = ['#include "mm_oracle.h"',
               "float libnoise_perlin(int, float, float, float, float, float);",
               "float libnoise_billow(int, float, float, float, float, float);",
               "float libnoise_ridged_multi(int, float, float, float, float);",
               "float libnoise_voronoi(float, float, float, float);", ""]
        lens = set()
        for code in [ir] + list(self.functions):
            for v in code["vars"]:
                if v["type"] in ("tuple", "tree_vector") and v["tuple_len"] not in (0, 2, 3, 4, 9):
                    lens.add(v["tuple_len"])
        for n in sorted(lens):      # mm_oracle.h has the lengths the builtins use
            out.append("typedef struct { float v[%d]; } mmo_tup%d;" % (n, n))
        out.append("typedef struct {")
        for d in transfers:
            out.append("  %s %s;" % (self.ctype(d[0]), self.vname(*d)))
        out.append("  int unused_;")
        out.append("} xy_vars_t;")
        out.append("""
#define MMO_INTERNALS \\
    const float t = A->t; const float R = A->R; const int frame = A->frame; \\
    const int __canvasPixelW = A->img_width, __canvasPixelH = A->img_height; \\
    const int __renderPixelW = A->render_width, __renderPixelH = A->render_height; \\
    (void)t; (void)R; (void)frame; (void)__canvasPixelW; (void)__canvasPixelH; (void)__renderPixelW; (void)__renderPixelH;

int mmo_xy_size(void) { return (int)sizeof(xy_vars_t); }
int mmo_num_natives(void) { return %d; }
""" % self.nnative)
        self.function_source(out)
        out.append("""void mmo_init_frame(const mmo_args *A, void *xyv) {
  xy_vars_t *xy_vars = (xy_vars_t *)xyv;
  MMO_INTERNALS""")
        if self.functions:
            out.append("  const int col = 0, row = 0; unsigned mm_rand_ctr = 0; (void)col; (void)row; (void)mm_rand_ctr;")
        self.decls(pro_defs, "  ", out)
        self.stmts(ir["body"], True, "  ", out)
        for d in transfers:
            out.append("  xy_vars->%s = %s;" % (self.vname(*d), self.vname(*d)))
        out.append("  (void)xy_vars;\n}\n")
        out.append("""void mmo_calc_lines(const mmo_args *A, const void *xyv, int first_row, int last_row, void *q) {
  const xy_vars_t *xy_vars = (const xy_vars_t *)xyv;
  MMO_INTERNALS
  int row, col;
  first_row = MAX(0, first_row);
  last_row = MIN(last_row, A->region_y + A->region_height);""")
        for d in transfers:
            out.append("  const %s %s = xy_vars->%s;" % (self.ctype(d[0]), self.vname(*d), self.vname(*d)))
        out.append("""  (void)xy_vars;
  for (row = first_row - A->region_y; row < last_row - A->region_y; ++row) {
    float y = CALC_VIRTUAL_Y(row + A->region_y, A->frame_render_height, A->sampling_offset_y);
    unsigned char *p = q;
    float *fp = q;
    for (col = 0; col < A->region_width; ++col) {
      float x = CALC_VIRTUAL_X(col + A->region_x, A->frame_render_width, A->sampling_offset_x);
      float rt[4];
      const float __colF = (float)(col + A->region_x), __rowF = (float)(row + A->region_y);
      unsigned mm_rand_ctr = 0;
      (void)__colF; (void)__rowF;
      (void)x; (void)y; (void)mm_rand_ctr;""")
        self.decls(pix_defs, "      ", out)
        self.stmts(ir["body"], False, "      ", out)
        for i, r in enumerate(ir["result"]):
            out.append("      rt[%d] = %s;" % (i, self.prim(["v", r[0], r[1]])))
        out.append("""      mmo_store_pixel(A, p, fp, rt);
      p += A->output_bpp;
      fp += 4;
    }
    if (A->floatmap) q = (float *)q + A->frame_render_width * 4;
    else q = (unsigned char *)q + A->row_stride;
  }
}""")
        return "\n".join(out) + "\n"


# ---------------------------------------------------------------------------------
# build + run
# ---------------------------------------------------------------------------------
class _ImageDesc(C.Structure):
    _fields_ = [("data", C.c_void_p), ("w", C.c_int), ("h", C.c_int), ("kind", C.c_int), ("num_frames", C.c_int),
                ("channels", C.c_int), ("scale_x", C.c_float), ("scale_y", C.c_float), ("middle_x", C.c_float),
                ("middle_y", C.c_float), ("ax", C.c_float), ("bx", C.c_float), ("ay", C.c_float), ("by", C.c_float)]


MEMO_EXTRA = 32


class _Memo(C.Structure):
    _fields_ = [("valid", C.c_int), ("func", C.c_int), ("in_idx", C.c_int), ("in2_idx", C.c_int), ("a1", C.c_float), ("a2", C.c_float),
                ("map", C.c_void_p), ("w", C.c_int), ("h", C.c_int)]


class _Userval(C.Union):
    _fields_ = [("i", C.c_int), ("f", C.c_float), ("c", C.c_uint), ("image", C.c_int)]


class _Args(C.Structure):
    _fields_ = [("img_width", C.c_int), ("img_height", C.c_int), ("render_width", C.c_int), ("render_height", C.c_int),
                ("frame_render_width", C.c_int), ("frame_render_height", C.c_int), ("t", C.c_float),
                ("frame", C.c_int), ("R", C.c_float), ("region_x", C.c_int), ("region_y", C.c_int),
                ("region_width", C.c_int), ("region_height", C.c_int), ("sampling_offset_x", C.c_float),
                ("sampling_offset_y", C.c_float), ("output_bpp", C.c_int), ("row_stride", C.c_int),
                ("floatmap", C.c_int), ("intersample", C.c_int), ("supersampling", C.c_int),
                ("edge_behaviour_x", C.c_int), ("edge_behaviour_y", C.c_int), ("edge_color_x", C.c_uint),
                ("edge_color_y", C.c_uint), ("uservals", C.POINTER(_Userval)), ("images", C.POINTER(_ImageDesc)),
                ("num_images", C.c_int), ("native_slot_base", C.c_int), ("memo", C.POINTER(_Memo)),
                ("curves", C.c_void_p), ("gradients", C.c_void_p), ("closure_base", C.c_int), ("pixel_inc", C.c_int),
                ("memo_sites", C.c_int), ("memo_cap", C.c_int)]


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("%s failed:\n%s" % (" ".join(cmd), r.stdout))


def build_runtime():
    """Compiles the oracle's C runtime once (gcc -O2 -fPIC, like the reference's own objects)
    and returns the object files to link into every generated filter."""
    os.makedirs(BUILD, exist_ok=True)
    hdr = os.path.join(HERE, "mm_oracle.h")
    objs = []
    for name in ("mm_oracle_rt", "mm_oracle_fft"):
        obj = os.path.join(BUILD, name + ".o")
        src = os.path.join(HERE, name + ".c")
        if not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
            tmp = "%s.%d.tmp" % (obj, os.getpid())
            _run(["gcc", "-O2", "-c", "-fPIC", "-Wno-comment"] + SANITIZE + ["-o", tmp, src])
            os.replace(tmp, obj)
        objs.append(obj)
    return objs


def runtime_library():
    """The oracle's C runtime as a shared object of its own (for the entry points that are called
    without a generated filter: gauss_rows)."""
    rt = build_runtime()
    so = os.path.join(BUILD, "libmm_oracle_rt.so")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(o) for o in rt):
        tmp = "%s.%d.tmp" % (so, os.getpid())
        _run(["gcc", "-shared"] + SANITIZE + ["-o", tmp] + rt + ["-lm"])
        os.replace(tmp, so)
    return C.CDLL(so)


def gauss_rows(image, hdev, vdev, rows, threads=1):
    """Rows `rows` of gaussian_blur(image, hdev, vdev) as the reference computes it for a
    `stretched image` of the canvas size (render_image + gauss_iir, native-filters/gauss.c), as
    float32 [len(rows), W, 4] -- for frames too large to blur whole in test time.  The vertical
    pass still runs over every column (spread over `threads`), only the sampled rows are kept."""
    lib = runtime_library()
    a = np.ascontiguousarray(image, dtype=np.uint8)
    h, w, c = a.shape
    rows_a = np.ascontiguousarray(rows, dtype=np.int32)
    out = np.zeros((len(rows_a), w, 4), np.float32)
    lib.mmo_gauss_rows_vertical.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p,
                                            C.c_int, C.c_void_p]
    lib.mmo_gauss_rows_horizontal.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float]

    def part(lo, hi):
        lib.mmo_gauss_rows_vertical(a.ctypes.data, c, w, h, vdev, lo, hi, rows_a.ctypes.data, len(rows_a), out.ctypes.data)
    n = max(1, int(threads))
    ths = [threading.Thread(target=part, args=(w * k // n, w * (k + 1) // n)) for k in range(n)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    lib.mmo_gauss_rows_horizontal(out.ctypes.data, w, h, len(rows_a), hdev)
    return out


class CpuFilter:
    """A filter compiled by the oracle's cc-equivalent path."""

    def __init__(self, ir_json, extra_cflags=()):
        """`extra_cflags`: added to the reference's `gcc -O2 -c -fPIC`.  The one use is `-fno-builtin`
        in tests that compare two printings of one filter where only one of them has literal
        arguments in libm calls: gcc evaluates `cexpf(literal)` at compile time with MPFR/MPC
        (correctly rounded), glibc's run-time cexpf is 1-2 ulps off that, so the two would differ
        by gcc's doing, not the IR's.  (The reference compiles without the flag, and so does every
        parity check.)"""
        self.ir = json.loads(ir_json) if isinstance(ir_json, str) else ir_json
        gen = Gen(self.ir)
        self.source = gen.source()
        self.nnative = gen.nnative
        noise = [NOISE_LIB, "-Wl,-rpath," + os.path.dirname(NOISE_LIB)] if gen.uses_noise else []
        key = hashlib.sha1((self.source + " ".join(LIBM_NO_BUILTIN + list(extra_cflags))).encode()).hexdigest()[:16]
        rt = build_runtime()
        so = os.path.join(BUILD, "f_%s.so" % key)
        if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(o) for o in rt):
            # per-process intermediate names + atomic rename: several ranks may build the same filter at once
            cfile = os.path.join(BUILD, "f_%s.%d.c" % (key, os.getpid()))
            ofile = os.path.join(BUILD, "f_%s.%d.o" % (key, os.getpid()))
            with open(cfile, "w") as f:
                f.write(self.source)
            # the reference's CGEN_CC / CGEN_LD (Makefile:58-60)
            _run(["gcc", "-O2", "-c", "-fPIC", "-Wno-comment", "-I", HERE] + SANITIZE + LIBM_NO_BUILTIN + list(extra_cflags) +
                 ["-o", ofile, cfile])
            tmp_so = "%s.%d.tmp" % (so, os.getpid())
            _run(["gcc", "-shared"] + SANITIZE + ["-o", tmp_so, ofile] + rt + noise + ["-lm"])
            os.replace(tmp_so, so)
            for f in (cfile, ofile):
                try:
                    os.remove(f)
                except OSError:
                    pass
        # closure images handed to native filters: each is rendered by its own code (render_image's closure
        # branch launches the closure's calc_lines), the IR dump carries that code under "closure_renders"
        # The render code of closure k evaluates the main filter's code once more (it computes the closure's arguments), and
        # that code may hold native calls on the closures before it (`b = gaussian_blur(inner(in, ..)); c = gaussian_blur(inner(b, ..))`):
        # closure k's own renders are the main filter's renders 0 .. k-1 (same numbering; an argument can only rest on what came before)
        renders = self.ir.get("closure_renders", [])
        self.subs = [CpuFilter(dict(sub, functions=self.ir.get("functions", []), closure_renders=sub.get("closure_renders") or renders[:k]),
                               extra_cflags)
                     for k, sub in enumerate(renders)]
        self.lib = C.CDLL(so)
        self.lib.mmo_xy_size.restype = C.c_int
        self.lib.mmo_init_frame.argtypes = [C.POINTER(_Args), C.c_void_p]
        self.lib.mmo_calc_lines.argtypes = [C.POINTER(_Args), C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        self.lib.mmo_free_memo.argtypes = [C.POINTER(_Args), C.c_int]

    def render(self, width, height, uservals=None, images=None, t=0.0, frame=0, intersample=True, threads=1,
               bpp=4, floatmap=False, edge=(0, 0), edge_colors=(0, 0), rows=None, timing=None,
               region_width=None, sampling_offset=(0.0, 0.0), supersampling=False, render_size=None, pixel_inc=1):
        """Renders on the CPU.  `uservals`: {name: value}; `images`: {name: uint8 [H,W,3|4]}.
        `threads` > 1 splits the rows into contiguous bands like call_invocation_parallel
        (mathmap_common.c:972-1006).  Returns uint8 [H,W,bpp] or float32 [H,W,4]."""
        import time
        uservals = uservals or {}
        images = images or {}
        a_img_w, a_img_h = width, height
        infos = self.ir["uservals"]
        n_uv = max(len(infos), 1)
        uv = (_Userval * n_uv)()
        descs = []
        keep = []
        curves, grads = [], []
        for u in infos:
            k, i, name = u["kind"], u["index"], u["name"]
            if k == 0:
                uv[i].i = int(uservals.get(name, u["idef"]))
            elif k == 1:
                uv[i].f = float(uservals.get(name, np.float32(u["fdef"])))
            elif k == 2:
                uv[i].i = int(bool(uservals.get(name, u["bdef"])))
            elif k == 3:
                col = uservals.get(name, (0.0, 0.0, 0.0, 1.0))
                q = [int(min(max(c, 0.0), 1.0) * 255.0) for c in col]
                uv[i].c = (q[0] << 24) | (q[1] << 16) | (q[2] << 8) | q[3]
            elif k == 6:
                d = _ImageDesc()
                if name in images:
                    a = np.ascontiguousarray(images[name], dtype=np.uint8)
                    keep.append(a)
                    h, w, c = a.shape
                    d.data, d.w, d.h, d.kind, d.num_frames, d.channels = a.ctypes.data, w, h, 0, 1, c
                    d.scale_x = np.float32((w - 1) / 2.0)
                    d.scale_y = np.float32((h - 1) / 2.0)
                    d.middle_x = d.middle_y = 1.0
                else:
                    d.kind = 2
                uv[i].image = len(descs)
                descs.append(d)
            elif k == 4:   # curve: default identity ramp (userval.c:282-311)
                uv[i].i = len(curves)
                curves.append(uservals.get(name, np.arange(1024, dtype=np.float32) / np.float32(1023)))
            elif k == 5:   # gradient: default opaque grey ramp (mathmap.c:356-361)
                uv[i].i = len(grads)
                if name in uservals:
                    grads.append(np.asarray(uservals[name], dtype=np.uint32))
                else:
                    v = np.arange(1024, dtype=np.float32) / np.float32(1023)
                    q = (v.astype(np.float64) * 255.0).astype(np.uint32) & 0xff
                    grads.append((q << 24) | (q << 16) | (q << 8) | np.uint32(255))
            else:
                uv[i].i = 0
        nbase = len(descs)
        # one result slot per native call site, and room for the further argument sets of sites that run more than once
        # per frame (mm_oracle_rt.c mmo_memo_slot)
        nmemo = self.nnative + MEMO_EXTRA if self.nnative else 0
        for _ in range(nmemo):
            d = _ImageDesc()
            d.kind = 2
            descs.append(d)
        closure_base = -1
        if self.subs:
            # builtins.c:273-298: the whole frame, sampling offsets 0, float map output; the closure's body has t = 0.0 and
            # frame = 0 as literals (the frame render_image makes), its arguments are the main code's values at this t
            closure_base = len(descs)
            for sub in self.subs:
                m = sub.render(a_img_w, a_img_h, uservals=uservals, images=images, t=t, frame=frame, intersample=intersample,
                               floatmap=True, edge=edge, edge_colors=edge_colors, supersampling=supersampling,
                               render_size=render_size, pixel_inc=pixel_inc)
                keep.append(m)
                d = _ImageDesc()
                d.data, d.w, d.h, d.kind, d.num_frames, d.channels = m.ctypes.data, m.shape[1], m.shape[0], 1, 1, 4
                d.ax = d.bx = np.float32(np.float32(d.w - 1) / 2.0)          # floatmap.c:39-41
                d.by = np.float32(np.float32(d.h - 1) / 2.0)
                d.ay = np.float32(np.float64(d.by) * -1.0)
                descs.append(d)
        if not descs:
            d = _ImageDesc()
            d.kind = 2
            descs.append(d)
        dtab = (_ImageDesc * len(descs))(*descs)
        memo = (_Memo * max(nmemo, 1))()
        a = _Args()
        # `render_size`: a render of the (width x height) canvas at another pixel size -- the GIMP
        # preview (mathmap.c:2191-2223): img_width/height stay, render_width/height change
        a.img_width, a.img_height = width, height
        if render_size is not None:
            width, height = render_size
        a.render_width = a.frame_render_width = width
        a.render_height = a.frame_render_height = height
        a.t, a.frame, a.R = t, frame, np.float32(np.sqrt(2.0))
        a.region_x = a.region_y = 0
        rw = region_width or width
        a.region_width, a.region_height = rw, height
        a.sampling_offset_x, a.sampling_offset_y = sampling_offset
        a.output_bpp, a.row_stride, a.floatmap = bpp, rw * bpp, 1 if floatmap else 0
        a.intersample, a.supersampling = 1 if intersample else 0, 1 if supersampling else 0
        a.edge_behaviour_x, a.edge_behaviour_y = edge
        a.edge_color_x, a.edge_color_y = edge_colors
        a.uservals, a.images, a.num_images, a.native_slot_base, a.memo = uv, dtab, len(descs), nbase, memo
        ctab = np.ascontiguousarray(np.concatenate(curves).astype(np.float32)) if curves else np.zeros(1, np.float32)
        gtab = np.ascontiguousarray(np.concatenate(grads).astype(np.uint32)) if grads else np.zeros(1, np.uint32)
        a.curves, a.gradients = ctab.ctypes.data, gtab.ctypes.data
        a.closure_base = closure_base
        a.pixel_inc = pixel_inc
        a.memo_sites, a.memo_cap = self.nnative, nmemo
        out = np.zeros((height, rw, 4), np.float32) if floatmap else np.zeros((height, rw, bpp), np.uint8)
        xy = C.create_string_buffer(max(self.lib.mmo_xy_size(), 16))
        r0, r1 = rows if rows is not None else (0, height)
        t0 = time.perf_counter()
        self.lib.mmo_init_frame(C.byref(a), xy)
        stride = out.strides[0]

        def band(lo, hi):
            self.lib.mmo_calc_lines(C.byref(a), xy, lo, hi, C.c_void_p(out.ctypes.data + lo * stride))
        if threads <= 1:
            band(r0, r1)
        else:
            ths = []
            n = r1 - r0
            for k in range(threads):
                lo, hi = r0 + n * k // threads, r0 + n * (k + 1) // threads
                th = threading.Thread(target=band, args=(lo, hi))
                th.start()
                ths.append(th)
            for th in ths:
                th.join()
        if timing is not None:
            timing.append(time.perf_counter() - t0)
        self.lib.mmo_free_memo(C.byref(a), nmemo)
        del keep
        return out


def render_supersampled(cf, width, height, **kw):
    """call_invocation's supersampling branch (mathmap_common.c:880-927) on top of CpuFilter."""
    longs = cf.render(width, height, region_width=width + 1, sampling_offset=(-0.5, -0.5), supersampling=True, **kw).astype(np.int32)
    shorts = cf.render(width, height, supersampling=True, **kw).astype(np.int32)
    l1 = longs
    l3 = np.concatenate([longs[1:], longs[-1:]], axis=0)   # the last line3 is never re-rendered
    out = (l1[:, :-1] + l1[:, 1:] + 2 * shorts + l3[:, :-1] + l3[:, 1:]) // 6
    return out.astype(np.uint8)
