"""CPU-only tests (`-m "not gpu"`): the oracle against the reference's golden PNGs, the
host logic (front-end, IR, code generator, offline hiprtc compile) and the C-ABI
library's exported surface.  No compute call needs a GPU."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

import mathmap_amd as mm
from mathmap_amd._lib import BACKEND_SYMBOLS, LIB_PATH, SYMBOLS, lib
from oracle.ccgen import CpuFilter
from tests import filters as F
from tests.conftest import REFERENCE, ROOT, load_png_rgb


def oracle_render(src, uv=None, image=None, w=256, h=256, t=0.0, intersample=True):
    """`src`: .mm text, or a compiled Filter (the reference's filters come from IR fixtures, tests/filters.py)."""
    flt = src if isinstance(src, mm.Filter) else mm.Filter(src)
    images = {"in": image} if image is not None else {}
    return CpuFilter(flt.ir_json_raw).render(w, h, uservals=uv, images=images, t=t, intersample=intersample)


# ---- the oracle is pinned by the reference's own golden vectors --------------------------
GOLDEN = [
    ("mandelbrot", "render_mandelbrot.png", {}, False, 0),
    ("ident", "utilities_ident.png", {}, True, 0),
    ("pond", "distorts_pond.png", {}, True, 0),
    ("gaussian_blur", "blur_gaussian_blur.png", {"dev": 0.1}, True, 0),
    # float-complex glibc functions: the golden was made with an older glibc; the survey
    # measured the same 24 values off by one with a hand-written C restatement
    ("droste", "map_droste.png", {}, True, 1),
    ("closure_value", "apply.png", {}, False, 0),
    ("closure_call", "circle.png", {}, True, 0),
    ("closure_arg", "closure.png", {}, True, 0),
    ("nested_calls", "twice.png", {}, True, 0),
    # native FFT filter: the oracle's direct DFT against the reference's FFTW-made golden
    ("visualize_fft", "utilities_visualize_fft.png", {}, True, 0),
]


@pytest.mark.parametrize("name,golden,uv,needs,tol", GOLDEN)
def test_oracle_matches_reference_golden(name, golden, uv, needs, tol, marlene):
    got = oracle_render(F.load(name), uv, marlene if needs else None)
    want = load_png_rgb(golden)
    d = np.abs(got[:, :, :3].astype(int) - want.astype(int))
    assert d.max() <= tol
    if name == "droste":
        assert (d > 0).sum() <= 40


def test_ident_golden_equals_input(marlene):
    assert np.array_equal(load_png_rgb("utilities_ident.png"), marlene)


# ---- golden sweep over the reference's test-suite where the filters are supported --------
def _run_tests_cases():
    """Parses tests/run_tests.sh of the reference: (script, golden, {-D values}, needs image)."""
    path = os.path.join(REFERENCE, "tests", "run_tests.sh")
    if not os.path.exists(path):
        return []
    cases = []
    for line in open(path):
        m = re.match(r'\s*run_(modify|render)_test\s+("[^"]+"|\S+)\s+(\S+)\s*(.*)$', line)
        if not m:
            continue
        kind, script, golden, rest = m.groups()
        script = script.strip('"')
        if script == "()":      # the shell function definitions themselves
            continue
        uv = {k: float(v) for k, v in re.findall(r"-D(\w+)=([-\d.]+)", rest)}
        cases.append((script, golden, uv, kind == "modify"))
    return cases


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="reference tree not present")
def test_oracle_sweep_over_reference_suite(marlene):
    """Every case of the reference's run_tests.sh: the oracle (printing the IR as lowered, before any
    pass) against the reference's golden PNG, held to the per-case record in
    tests/golden/expected_oracle_vs_golden.json -- bit-exact unless listed there."""
    from tests.expectations import Expectations
    exp = Expectations("oracle_vs_golden")
    cases = _run_tests_cases()
    assert len(cases) == 80
    results = {}
    for script, golden, uv, needs in cases:
        src = open(os.path.join(REFERENCE, "tests", script)).read()
        got = oracle_render(src, uv, marlene if needs else None)
        want = load_png_rgb(golden)
        d = np.abs(got[:, :, :3].astype(int) - want.astype(int))
        results[golden] = (int(d.max()), int((d > 0).sum()), int((d > 1).sum()))
    failures = []
    for golden, (mx, nd, n1) in sorted(results.items()):
        try:
            exp.check(golden, mx, nd, n1, 256 * 256 * 3, default=(0, 0))
        except AssertionError as e:
            failures.append(str(e))
    report = os.path.join(ROOT, "tests", "golden_sweep_report.json")
    json.dump({g: {"max": r[0], "n_diff": r[1], "n_gt1": r[2]} for g, r in sorted(results.items())}, open(report, "w"), indent=1)
    assert not failures, failures
    assert sum(1 for r in results.values() if r[0] == 0) >= 72


def _fixture_sets():
    import glob
    out = []
    for sub in ("ir", "ir_examples"):
        out += sorted(glob.glob(os.path.join(ROOT, "tests", "golden", sub, "*.json.gz")))
    return out


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="reference tree not present")
def test_ir_fixtures_equal_a_fresh_compile():
    """tests/golden/ir*/*.json.gz (what the GPU box replays, it has no reference tree) are the IR
    of the reference's scripts as lowered by today's front-end, before any pass: regenerate them
    with tests/make_ir_fixtures.py whenever the front-end changes."""
    import glob
    import gzip
    stale = []
    n = 0
    man = json.load(open(os.path.join(ROOT, "tests", "golden", "ir", "manifest.json")))
    by_golden = {c[1]: c for c in _run_tests_cases()}
    for m in man:
        script, golden, uv, needs = by_golden[m["golden"]]
        flt = mm.Filter(open(os.path.join(REFERENCE, "tests", script)).read())
        have = gzip.open(os.path.join(ROOT, "tests", "golden", "ir", m["ir"]), "rt").read()
        n += 1
        if have != flt.ir_json_raw:
            stale.append(m["ir"])
    for path in sorted(glob.glob(os.path.join(REFERENCE, "examples", "**", "*.mm"), recursive=True)):
        rel = os.path.relpath(path, os.path.join(REFERENCE, "examples"))
        stem = rel[:-3].replace("/", "__").replace(" ", "_")
        flt = mm.Filter(open(path, errors="replace").read())
        have = gzip.open(os.path.join(ROOT, "tests", "golden", "ir_examples", stem + ".json.gz"), "rt").read()
        n += 1
        if have != flt.ir_json_raw:
            stale.append(stem)
    assert n == 80 + 189
    assert not stale, stale


def _every(items, step, phase=0):
    return [x for i, x in enumerate(items) if i % step == phase]


@pytest.mark.parametrize("path", _every(_fixture_sets(), 4), ids=lambda p: os.path.basename(p)[:-8])
def test_passes_preserve_results_on_cpu(path, marlene):
    """The product's passes (copy propagation / DCE, loop-carried CSE, frame-constant hoisting,
    speculative hoisting of pure calls) are exact by contract.  CPU differential check: the oracle
    prints the fixture IR as lowered and the IR the product generates kernels from (after the
    passes, replayed through mmhip_compile_ir_json); same gcc, same glibc => identical bytes.
    Every 4th fixture here; the GPU suite does all of them against the pre-pass IR."""
    import gzip
    raw = gzip.open(path, "rt").read()
    flt = mm.Filter("", ir_json=raw)
    w, h = 96, 64
    img = np.ascontiguousarray(marlene[:h, :w])
    images = {u["name"]: img for u in flt.uservals if u["kind"] == 6}
    try:
        a = CpuFilter(raw).render(w, h, images=images, t=0.3)
    except Exception as e:
        if "Unsupported" in type(e).__name__:
            pytest.skip(str(e))
        raise
    b = CpuFilter(flt.ir_json).render(w, h, images=images, t=0.3)
    assert np.array_equal(a, b), np.abs(a.astype(int) - b.astype(int)).max()


@pytest.mark.parametrize("name,uv", [
    ("mandelbrot", {}), ("mandelbrot", {"num_iterations": 50, "pj": 0.3}), ("droste", {}),
    ("droste", {"NoTransparency": 1, "Zoom": 3}), ("pond", {"height": 0.0}), ("closure_call", {"radius": 0.0}),
])
def test_userval_specialisation_preserves_results_on_cpu(name, uv):
    """specialize_constants (SCCP with the user values as literals + the reference's literal folds)
    against the generic IR as lowered, both printed by the oracle."""
    w, h = 120, 80
    img = F.synthetic_image(w, h, seed=6)
    generic = F.load(name)
    images = {"in": img} if F.image_names(generic) else {}
    special = generic.specialized(uv)
    # -fno-builtin on both sides: with the user values as literals gcc would evaluate the frame-constant
    # libm calls itself (MPFR/MPC, correctly rounded) instead of calling glibc (see CpuFilter)
    a = CpuFilter(generic.ir_json_raw, extra_cflags=("-fno-builtin",)).render(w, h, uservals=uv, images=images, t=0.4)
    b = CpuFilter(special.ir_json, extra_cflags=("-fno-builtin",)).render(w, h, uservals=uv, images=images, t=0.4)
    assert np.array_equal(a, b), np.abs(a.astype(int) - b.astype(int)).max()


def test_tile_division_by_multiply_high_is_exact_where_it_is_used():
    """mm_host_abi.h tile_division_magic: workgroup id / tile columns as umulhi(id, ceil(2^32 / d)), used only when
    nwg * d < 2^32.  Restated here and compared with integer division for every id of small launches and for the ids
    around every multiple of d of launches at the bound."""
    def magic(d, nwg):
        return 0 if d < 2 or nwg * d >= 1 << 32 else ((1 << 32) + d - 1) // d
    rng = np.random.default_rng(7)
    for d in [2, 3, 5, 7, 16, 17, 127, 128, 129, 511, 512, 513, 1000, 1024, 4095, 4096, 65535] + list(rng.integers(2, 70000, 40)):
        d = int(d)
        nwg = min((1 << 32) // d - 1, 1 << 31)
        m = magic(d, nwg)
        assert m and magic(d, (1 << 32) // d + 1) == 0
        if nwg <= 1 << 20:
            ids = np.arange(nwg, dtype=np.uint64)
        else:
            k = np.unique(np.concatenate([rng.integers(0, nwg // d + 1, 4000), [0, 1, nwg // d - 1, nwg // d]])).astype(np.uint64)
            ids = np.unique(np.concatenate([k * d + off for off in (0, 1, d - 1, d // 2)] + [np.array([nwg - 1], np.uint64)]))
            ids = ids[ids < nwg]
        assert np.array_equal((ids * np.uint64(m)) >> np.uint64(32), ids // np.uint64(d)), d


def test_closure_render_takes_arguments_at_the_current_time_and_runs_its_body_at_t_zero():
    """render_image's closure branch (builtins.c:273-298) on the oracle: the closure's calc_lines runs on a frame with
    t = 0.0 and frame = 0, its arguments are what the main filter's code computed at the current t."""
    TIMED_ARG = F.CLOSURE_TIMED_ARG
    img = F.synthetic_image(96, 64, seed=3)
    a = CpuFilter(mm.Filter(TIMED_ARG).ir_json_raw).render(96, 64, images={"in": img}, t=0.5, frame=7)
    fb = mm.Filter(TIMED_ARG.replace("k * (1 + t)", "k"))
    for t, frame in ((0.0, 0), (0.9, 3)):
        b = CpuFilter(fb.ir_json_raw).render(96, 64, uservals={"k": 0.75}, images={"in": img}, t=t, frame=frame)
        assert np.array_equal(a, b), (t, frame)


def test_recursive_filter_calls_at_run_time_and_unrolls_with_literals():
    """A recursive application is a run-time call of filter_$name (compiler.c:2165-2222 RHS_FILTER,
    backends/cc.c:221-235): the IR carries the callee's body under "functions" and one generic kernel serves
    every depth.  With the user values baked in the recursion is unrolled while lowering instead; the two
    forms must agree byte for byte.  depth = 1 equals the plain fetch; a recursion that never ends is cut
    off at MM_MAX_CALL_DEPTH (zeros) in the oracle exactly as in the kernel."""
    flt = F.load("recursive")
    assert [u["name"] for u in flt.uservals] == ["in", "depth", "s"]
    raw = json.loads(flt.ir_json_raw)
    assert [fn["filter"] for fn in raw["functions"]] == ["tree"] and '"filtercall"' in flt.ir_json_raw
    assert "mm_filter_0<0>" in flt.kernel_source and "mm_filter_0<MM_D + 1>" in flt.kernel_source
    img = F.synthetic_image(64, 48, seed=2)
    generic = CpuFilter(flt.ir_json_raw)
    ident = CpuFilter(F.load("ident").ir_json_raw).render(64, 48, images={"in": img})
    assert np.array_equal(generic.render(64, 48, uservals={"depth": 1}, images={"in": img}), ident)
    sizes = []
    for d in (1, 2, 4, 7):
        uv = {"depth": d, "s": 0.7}
        sp = flt.specialized(uv)
        assert "functions" not in json.loads(sp.ir_json_raw) and "mm_filter_" not in sp.kernel_source
        sizes.append(len(sp.ir_json))
        a = generic.render(64, 48, uservals=uv, images={"in": img})
        b = CpuFilter(sp.ir_json_raw).render(64, 48, uservals=uv, images={"in": img})
        assert np.array_equal(a, b), d
    assert sizes[0] < sizes[1] < sizes[2] < sizes[3]
    # no literal ends this recursion: even with d baked in it stays a call, and the call depth bound ends it
    endless = mm.Filter("filter f (image in, int d: 1-9 (3)) f(in, d, xy) end", constants={"d": 3})
    assert '"filtercall"' in endless.ir_json_raw
    out = CpuFilter(endless.ir_json_raw).render(16, 8, images={"in": img})
    assert not out[..., :3].any()


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="reference tree not present")
def test_every_reference_example_compiles_for_gfx950():
    """All 189 filters under the reference's examples/ go through the front-end, the lowering, the
    kernel generator and hiprtc (--offload-arch=gfx950).  (Code objects are cached on disk, so only the first run pays the ~90 s.)"""
    import glob
    files = sorted(glob.glob(os.path.join(REFERENCE, "examples", "**", "*.mm"), recursive=True))
    assert len(files) == 189
    bad = {}
    for f in files:
        try:
            flt = mm.Filter(open(f, errors="replace").read())
            flt.jit()
        except mm.MathMapError as e:
            bad[os.path.relpath(f, REFERENCE)] = str(e).splitlines()[0][:120]
    assert not bad, bad


def test_dynamic_subscripts_follow_tree_vector_semantics():
    """v[i] / v[i] = x with run-time i (the reference's tree vectors): indices are truncated to int and
    clamped to the tuple, elements are floats, a write replaces one element.  Expected values
    computed independently with numpy."""
    w, h = 64, 8
    out = CpuFilter(F.load("tree_vector").ir_json_raw).render(w, h, floatmap=True)
    col = np.arange(w, dtype=np.float64)
    x = ((col - (w - 1) / 2.0) / ((w - 1) / 2.0)).astype(np.float32)                        # X = 1
    for row in range(h):
        y = np.float32(np.float32((-(row) + (h - 1) / 2.0) / ((h - 1) / 2.0)) * np.float32(h / w))   # Y = h / max(w, h)
        v = np.stack([x, np.full(w, y, np.float32), x * y, np.ones(w, np.float32)], axis=1)
        i = np.floor((x + np.float32(1)) * np.float32(2.5)).astype(int) - 1
        ic = np.clip(i, 0, 3)
        wv = v[np.arange(w), ic]
        v[np.arange(w), np.clip(i + 1, 0, 3)] = 0.25
        u = np.array([0.1, 0.5, 0.9], np.float32)[min(int(np.floor(abs(y) * np.float32(40))), 2)]
        want = np.stack([wv, v[:, 2], np.full(w, u, np.float32), v[:, 2]], axis=1)
        assert np.array_equal(out[row], want), row


# ---- host logic ----------------------------------------------------------------------------
def test_parse_errors_are_reported():
    for bad, msg in [("filter f () [1,2] end", "rgba:4"), ("filter f () q end", "Undefined variable"),
                     ("filter f (int a, int a) grayColor(1) end", "declared more than once"),
                     ("filter f () x = 1; grayColor(1) end", "Cannot assign to internal"),
                     ("filter f () v = 1; v = [1,2]; grayColor(v) end", "two different types"),
                     ("filter f () grayColor(1", "Parse error"),
                     ("filter f () grayColor([1,2] + [1,2,3]) end", "Unable to resolve")]:
        with pytest.raises(mm.MathMapError) as e:
            mm.Filter(bad)
        assert msg in str(e.value), (bad, str(e.value))


def test_overload_resolution_and_types():
    f = mm.Filter("filter f (float a: 0-1 (0.5)) z = ri:[a, 2]; w = z * z + 1; grayColor(abs(w)) end")
    ops = []

    def walk(b):
        for s in b:
            if s["k"] == "assign" and s["rhs"]["k"] == "op":
                ops.append(s["rhs"]["op"])
            for k in ("then", "else", "body"):
                if k in s:
                    walk(s[k])
    walk(f.ir["body"])
    assert "hypot" in ops and "MUL" in ops
    types = {v["id"]: v["type"] for v in f.ir["vars"]}
    assert "float" in types.values() and "int" in types.values()


def test_frame_constant_code_is_hoisted():
    """Droste's user-value-only set-up must land in the prologue slice, the per-pixel slice
    must still contain the complex log/exp chain."""
    ir = F.load("droste").ir

    def count(b, key, pred):
        n = 0
        for s in b:
            if s["k"] == "assign" and s[key] and pred(s):
                n += 1
            for k in ("then", "else", "body"):
                if k in s:
                    n += count(s[k], key, pred)
        return n
    is_op = lambda name: (lambda s: s["rhs"]["k"] == "op" and s["rhs"]["op"] == name)
    assert count(ir["body"], "hoisted", is_op("atan")) >= 1
    assert count(ir["body"], "pixel", is_op("clogf")) >= 1
    assert count(ir["body"], "hoisted", is_op("ORIG_VAL")) == 0


def test_all_workloads_compile_for_gfx950_offline():
    for name in F.NAMES:
        f = F.load(name)
        assert "mm_pixels(mm_args A" in f.kernel_source
        assert f.jit(load=False) > 1000, name


def test_generated_c_of_oracle_has_reference_shape():
    src = CpuFilter(F.load("mandelbrot").ir_json).source
    assert "CALC_VIRTUAL_Y(row + A->region_y" in src and "while (" in src and "mmo_store_pixel" in src


# ---- C-ABI surface ------------------------------------------------------------------------------
def test_library_exports_every_declared_symbol():
    l = C.CDLL(LIB_PATH)
    declared = set()
    for hdr in ("mmhip.h", "mathmap_hip_backend.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        declared |= set(re.findall(r"\b((?:mmhip|mathmap_hip|gen_and_load_hip|unload_hip)_?\w*)\s*\(", text))
    declared -= {"mmabi_get_pixel_func_t"}
    assert len(declared) >= 35
    for name in declared:
        assert hasattr(l, name), "symbol %s declared in include/ but not exported" % name
    for name in list(SYMBOLS) + list(BACKEND_SYMBOLS):
        assert hasattr(l, name)


def test_abi_struct_layouts_match_lp64_expectations():
    """sizeof/offsetof of the reference-layout mirrors (include/mathmap_abi.h) as the host C
    compiler sees them; the expected numbers are derived by hand from the reference headers."""
    import subprocess
    import tempfile
    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "mathmap_abi.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(mmabi_userval_t), sizeof(mmabi_mathmap_pools_t),
         offsetof(mmabi_invocation_t, img_width), offsetof(mmabi_invocation_t, row_stride),
         offsetof(mmabi_image_t, v), sizeof(mmabi_primary_t), sizeof(mmabi_rhs_t),
         offsetof(mmabi_statement_t, parent), offsetof(mmabi_slice_t, region_x), sizeof(mmabi_value_t));
  return 0; }'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(prog)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", os.path.join(d, "t"), os.path.join(d, "t.c")], check=True)
        out = subprocess.run([os.path.join(d, "t")], stdout=subprocess.PIPE, text=True, check=True).stdout.split()
    got = [int(x) for x in out]
    # userval_t: union{.., struct{GimpRGB(32) + color_t}=40} + GtkObject* = 48
    # mathmap_pools_t: int + pad + pools_t(4+pad+8+20*8=176) + ptr = 192
    # invocation: mathmap*(8) uservals*(8) antialiasing(4)+pad(4) orig_val_func(8) supersampling(4) output_bpp(4)
    #             edge_x,y(8) edge colours(8) -> img_width at 56; ... image_R at 72, row_stride at 76
    assert got[0] == 48 and got[1] == 192
    assert got[2] == 56 and got[3] == 76
    assert got[4] == 16            # image_t: 4 ints then the union
    assert got[5] == 16            # primary_t: kind, const_type, 8-byte union
    assert got[6] == 8 + 8 + 9 * 16  # rhs_t: kind(+pad), op*, 9 primaries
    assert got[7] == 8 + 32        # statement_t: kind(+pad), 32-byte union, then parent
    assert got[8] == 16            # slice: frame*, 2 floats, region_x
    assert got[9] == 48


# ---- multi-process striping (gloo, world_size 2) ------------------------------------------------
def test_row_stripe_partition_is_a_partition():
    from mathmap_amd.striping import stripe_rows
    for h in (1, 7, 256, 8192, 16384):
        for g in (1, 2, 3, 8):
            rows = [stripe_rows(h, r, g) for r in range(g)]
            assert rows[0][0] == 0 and rows[-1][1] == h
            assert all(rows[i][1] == rows[i + 1][0] for i in range(g - 1))


# ---- exactness claims behind the device-side strength reductions (mm_device.h) -----------------
def test_byte_to_float_identity():
    c = np.arange(256, dtype=np.float64)
    assert np.array_equal((c / 255.0).astype(np.float32), (c * (1.0 / 255.0)).astype(np.float32))


def test_unit_of_a_byte_packs_back_to_that_byte():
    """mm_store_fetched_pixel (mm_device.h): a channel that holds RN(k / 255) for a byte k is packed by the template's
    (unsigned char)(c * 255.0) -- a double product, truncated -- back to k, for all 256 bytes; so a fetched pixel that
    goes to the output unchanged can be stored from the fetch's rounded byte sums."""
    k = np.arange(256)
    q = (k.astype(np.float64) * (1.0 / 255.0)).astype(np.float32)
    assert np.array_equal(np.trunc(q.astype(np.float64) * 255.0).astype(int), k)


def test_round_toward_zero_fma_is_the_templates_pack():
    """mm_pack_rgba8 (mm_device.h): floor(255 c) for a float c in [0, 1] -- what (unsigned char)(c * 255.0) is, the
    double product of a float and 255 being exact -- equals the low mantissa byte of the f32 value RTZ(255 c + 2^23).
    The RTZ fma is emulated exactly (255 c + 2^23 fits a double: 24 + 8 + 24 bits at most ... it does, being below
    2^24 in magnitude with an ulp of c * 2^-16 >= 2^-165; rounding toward zero to a multiple of 1 is floor)."""
    rng = np.random.default_rng(3)
    c = np.concatenate([rng.random(200000, dtype=np.float32), np.float32([0, 1, 0.5, 2 ** -149, 2 ** -126, 1 - 2 ** -24]),
                        (np.arange(256, dtype=np.float64) / 255.0).astype(np.float32),
                        np.nextafter((np.arange(1, 256, dtype=np.float64) / 255.0).astype(np.float32), np.float32(0)),
                        np.nextafter((np.arange(0, 255, dtype=np.float64) / 255.0).astype(np.float32), np.float32(1))])
    exact = c.astype(np.float64) * 255.0 + 8388608.0           # exact in float64
    rtz = np.floor(exact)                                     # f32 spacing at 2^23 is 1: round toward zero = floor
    low_byte = rtz.astype(np.int64) & 0xff
    assert np.array_equal(low_byte, np.trunc(c.astype(np.float64) * 255.0).astype(np.int64))


def test_four_equal_taps_interpolate_to_that_byte():
    """mm_intersample_tuple_cold's shortcuts (mm_device.h): get_orig_val_intersample_pixel's float sums
    c*p1 + c*p2 + c*p3 + c*p4 (builtins.c:228-247, this order, every operation rounded to float) round to c when all
    four taps are the byte c, whatever the fractional position -- the weights sum to 1 within 2^-22."""
    rng = np.random.default_rng(11)
    n = 20000
    fx = np.concatenate([rng.random(n, dtype=np.float32), np.float32([0, 0.5, 1 - 2 ** -24, 2 ** -24, 2 ** -30])])
    fy = np.concatenate([rng.random(n, dtype=np.float32), np.float32([0.5, 0, 1 - 2 ** -24, 1 - 2 ** -24, 2 ** -30])])
    one = np.float32(1.0)
    x1, y1 = one - fx, one - fy
    p1, p2, p3, p4 = x1 * y1, x1 * fy, fx * y1, fx * fy
    for c in range(256):
        cf = np.float32(c)
        sm = cf * p1
        sm = sm + cf * p2
        sm = sm + cf * p3
        sm = sm + cf * p4
        assert sm.dtype == np.float32 and np.array_equal(np.rint(sm), np.full_like(sm, cf)), c


def test_byte_to_unit_newton_step_identity():
    """mm_bytes_to_unit (mm_device.h): q = k*r, e = fma(-255, q, k), q' = fma(e, r, q) in f32 equals
    (float)((double)k * (1.0/255.0)) for every byte k.  The fmas are emulated exactly: 255*q has at
    most 32 significant bits and the residual 8, so float64 arithmetic reproduces them."""
    k = np.arange(256, dtype=np.float32)
    r = np.float32(1.0) / np.float32(255.0)
    q = (k * r).astype(np.float32)
    e = (k.astype(np.float64) - 255.0 * q.astype(np.float64))          # exact
    assert np.array_equal(e.astype(np.float32).astype(np.float64), e)   # representable: the fma is exact
    q2 = (q.astype(np.float64) + e * np.float64(r)).astype(np.float32)  # one rounding, like the fma
    want = (k.astype(np.float64) * (1.0 / 255.0)).astype(np.float32)
    assert np.array_equal(q2, want)
    assert (q != want).any()       # the plain f32 product alone is NOT enough


def test_fastmath_sin_cos_equal_glibc_sampled():
    """mm_fastmath.h (the text the JIT prelude embeds) against glibc's (float)sin((double)x) /
    (float)cos((double)x): every 97th float below 2^22, both signs -- 26 M arguments; likewise exp
    (|x| <= 700) and log (all positive floats).  The
    exhaustive run (stride 1, 2.5e9 arguments, 0 mismatches) is recorded in
    profiles/r01_verify_fastmath.json; tools/verify_fastmath.c is the checker."""
    import subprocess
    exe = os.path.join(ROOT, "oracle", "_build", "verify_fastmath")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.run(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-pthread", os.path.join(ROOT, "tools", "verify_fastmath.c"),
                    "-o", exe, "-lm"], check=True)
    r = subprocess.run([exe, "97", "2000000"], stdout=subprocess.PIPE, text=True)
    rep = json.loads(r.stdout)
    assert r.returncode == 0 and rep["sin_mismatches"] == 0 and rep["cos_mismatches"] == 0, rep
    assert rep["exp_mismatches"] == 0 and rep["log_mismatches"] == 0, rep
    assert rep["pow_mismatches"] == 0 and rep["pow_random_pairs_checked"] > 15_000_000, rep     # pow: sampled, not enumerable
    assert rep["checked"] > 25_000_000 and rep["exp_checked"] > 20_000_000 and rep["log_checked"] > 20_000_000


def test_glibc_float_functions_equal_host_libm_sampled():
    """mm_glibcf.h (glibc 2.35's float libm and float-complex functions restated; the text the JIT prelude
    embeds for the complex ops) against the host's libm, bit for bit: every 1021st float for the
    one-argument functions, the grid of special values and 400 000 random pairs per thread for atan2f,
    hypotf and the complex functions.  The exhaustive run (every float; 480 M pairs per function; 0
    mismatches) is recorded in profiles/r02_verify_glibcf.json; tools/verify_glibcf.c is the checker."""
    import subprocess
    exe = os.path.join(ROOT, "oracle", "_build", "verify_glibcf")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.run(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-pthread", os.path.join(ROOT, "tools", "verify_glibcf.c"),
                    "-o", exe, "-lm"], check=True)
    r = subprocess.run([exe, "1021", "400000"], stdout=subprocess.PIPE, text=True)
    rep = json.loads(r.stdout)
    assert r.returncode == 0 and rep["total_mismatches"] == 0, {k: v for k, v in rep["functions"].items() if v["mismatches"]}
    for name in ("expf", "logf", "sinf(sincosf)", "cosf(sincosf)", "atanf", "log1pf", "expm1f", "sinhf", "coshf"):
        assert rep["functions"][name]["checked"] > 4_000_000
    for name in ("atan2f", "hypotf", "cexpf", "clogf", "cpowf(c, z)", "csqrtf", "csinf", "ccosf", "ctanf", "csinhf", "ccoshf", "ctanhf",
                 "casinf", "cacosf", "catanf", "casinhf", "cacoshf", "catanhf"):
        assert rep["functions"][name]["checked"] > 3_000_000


def test_sqrt_less_power_of_two_identity():
    """sqrt_rn(a) < K  <=>  0 <= a < K*K for K a power of two, checked on every float in a
    window of +-2^16 ulps around K*K and on random floats."""
    rng = np.random.default_rng(0)
    for k in (0.25, 0.5, 1.0, 2.0, 4.0, 1024.0):
        k2 = np.float32(k * k)
        bits = np.array([k2], np.float32).view(np.uint32)[0]
        window = (np.arange(-65536, 65536, dtype=np.int64) + int(bits)).astype(np.uint32).view(np.float32)
        rnd = rng.uniform(0, 4 * k * k, 200000).astype(np.float32)
        special = np.array([0.0, -0.0, -1.0, np.nan, np.inf, -np.inf], np.float32)
        with np.errstate(invalid="ignore"):
            for a in (window, rnd, special):
                lhs = np.sqrt(a) < np.float32(k)          # numpy's float32 sqrt is correctly rounded
                rhs = (a < k2) & (a >= 0)
                assert np.array_equal(lhs, rhs), k


def test_float_add_half_and_floor_identities():
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-1e6, 1e6, 500000), rng.uniform(-4, 4, 500000)]).astype(np.float32)
    assert np.array_equal((x.astype(np.float64) + 0.5).astype(np.float32), x + np.float32(0.5))
    assert np.array_equal(np.floor(x.astype(np.float64)).astype(np.int64), np.floor(x).astype(np.int64))
    f = rng.uniform(0, 1, 500000).astype(np.float32)
    assert np.array_equal((1.0 - f.astype(np.float64)).astype(np.float32), np.float32(1.0) - f)


def test_oracle_dft_against_numpy_fft():
    """The oracle's direct long-double DFT (oracle/mm_oracle_fft.c) against numpy's pocketfft
    for the two transforms the reference asks of FFTW (r2c_2d, c2r_2d), even and odd sizes.
    Tolerance: 1e-12 of the largest output magnitude (double FFT round-off is ~1e-15)."""
    import ctypes as C
    import subprocess
    from oracle import ccgen
    rt = ccgen.build_runtime()
    so = os.path.join(ccgen.BUILD, "rt_only.so")
    subprocess.run(["gcc", "-shared", "-o", so] + rt + ["-lm"], check=True)
    lib = C.CDLL(so)
    rng = np.random.default_rng(5)
    for w, h in [(16, 8), (15, 9), (32, 31), (1, 7), (6, 1)]:
        x = rng.uniform(-1, 1, (h, w))
        cw = w // 2 + 1
        out = np.zeros((h, cw), np.complex128)
        lib.mmo_dft_r2c_2d(x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), w, h)
        want = np.fft.rfft2(x)
        assert np.abs(out - want).max() <= 1e-12 * np.abs(want).max(), (w, h)
        back = np.zeros((h, w))
        lib.mmo_dft_c2r_2d(want.copy().ctypes.data_as(C.c_void_p), back.ctypes.data_as(C.c_void_p), w, h)
        assert np.abs(back / (w * h) - x).max() <= 1e-12, (w, h)
        # a spectrum that is not Hermitian in its kx = 0 column (what half_convolve produces):
        # column inverse first, then the real row inverse -- numpy's irfft2 does the same
        spec = want * rng.uniform(0, 1, want.shape)
        lib.mmo_dft_c2r_2d(spec.copy().ctypes.data_as(C.c_void_p), back.ctypes.data_as(C.c_void_p), w, h)
        ref = np.fft.irfft(np.fft.ifft(spec, axis=0), n=w, axis=1) * (w * h)
        assert np.abs(back - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max()), (w, h)


def test_oracle_convolve_identities(marlene):
    """convolve with a centred unit impulse is the identity; with normalize the kernel's
    scale drops out; half_convolve with an all-ones mask is the identity (<= 1 LSB: the
    result goes double -> float -> byte)."""
    h, w = 48, 64
    img = np.ascontiguousarray(marlene[:h, :w])
    kern = np.zeros((h, w, 3), np.uint8)
    kern[h // 2 - 1, w // 2] = 255      # index n - nhalf of the flat map lands on [0][0] (convolve.c:119-122)
    src = "filter c (image in, image kernel, bool norm (1)) cv = convolve(in, kernel, norm, 1); cv(xy) end"
    flt = mm.Filter(src)
    cpu = CpuFilter(flt.ir_json_raw)
    ident = CpuFilter(mm.Filter("filter i (image in) in(xy) end").ir_json_raw).render(w, h, images={"in": img})
    for norm in (1, 0):
        got = cpu.render(w, h, uservals={"norm": norm}, images={"in": img, "kernel": kern})
        assert np.abs(got.astype(int) - ident.astype(int)).max() <= 1, norm
    grey = np.full((h, w, 3), 93, np.uint8)
    dim = (kern // 3).astype(np.uint8)
    a = cpu.render(w, h, uservals={"norm": 1}, images={"in": img, "kernel": kern})
    b = cpu.render(w, h, uservals={"norm": 1}, images={"in": img, "kernel": dim})
    assert np.abs(a.astype(int) - b.astype(int)).max() <= 1
    ones = np.full((h, w, 3), 255, np.uint8)
    hc = CpuFilter(mm.Filter("filter hc (image in, image mask) c = half_convolve(in, mask, 1); c(xy) end").ir_json_raw)
    got = hc.render(w, h, images={"in": img, "mask": ones})
    assert np.abs(got.astype(int) - ident.astype(int)).max() <= 1
    del grey


def test_oracle_cgamma_matches_reference_build():
    """oracle cgamma vs the reference's own builtins/spec_func.c compiled into oracle/_ref
    (bit-identical), and the known answer of its TEST_CGAMMA main: cgamma(2.5+0.5i)."""
    import subprocess
    from oracle import ccgen
    ref = os.path.join(ROOT, "oracle", "_ref", "libspec_func.so")
    rt = ccgen.build_runtime()
    so = os.path.join(ccgen.BUILD, "rt_only.so")
    subprocess.run(["gcc", "-shared", "-o", so] + rt + ["-lm"], check=True)
    prog = r'''
#include <complex.h>
#include <stdio.h>
#include <dlfcn.h>
typedef float _Complex (*fn)(float _Complex);
int main(int argc, char **argv) {
  fn mine = (fn)dlsym(dlopen(argv[1], RTLD_NOW), "cgamma");
  fn ref = argc > 2 ? (fn)dlsym(dlopen(argv[2], RTLD_NOW), "cgamma") : 0;
  float _Complex k = mine(2.5f + 0.5f * I);
  printf("%f %f\n", crealf(k), cimagf(k));
  int bad = 0;
  if (ref) for (float x = -3.7f; x < 6.0f; x += 0.31f) for (float y = -4.0f; y < 4.0f; y += 0.37f) {
    float _Complex a = mine(x + y * I), b = ref(x + y * I);
    if (!(crealf(a) == crealf(b) && cimagf(a) == cimagf(b)) && !(crealf(a) != crealf(a) && crealf(b) != crealf(b))) ++bad;
  }
  printf("%d\n", bad);
  return 0; }'''
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(prog)
        subprocess.run(["gcc", "-O0", "-o", os.path.join(d, "t"), os.path.join(d, "t.c"), "-ldl", "-lm"], check=True)
        args = [os.path.join(d, "t"), so] + ([ref] if os.path.exists(ref) else [])
        out = subprocess.run(args, stdout=subprocess.PIPE, text=True, check=True).stdout.split()
    assert abs(float(out[0]) - 1.172396) < 2e-6 and abs(float(out[1]) - 0.436507) < 2e-6
    assert int(out[2]) == 0


def test_elliptic_integrals_against_scipy():
    """ELL_INT_* (mm_gslmath.h, restated gsl_sf_ellint_* with GSL_PREC_SINGLE): the oracle build against
    scipy's Legendre and Carlson integrals on a square frame (x, y in [-1, 1]); P and D, which scipy
    lacks in this form, against quadrature.  GSL itself is absent: this pins the mathematics, not GSL's
    last bits."""
    import scipy.special as sp
    from scipy.integrate import quad
    n = 48

    def run(expr):
        flt = mm.Filter("filter e () v = %s; rgba:[v, v * 0.5, 0, 1] end" % expr)
        return CpuFilter(flt.ir_json_raw).render(n, n, floatmap=True)[..., 0].astype(np.float64)

    c = ((np.arange(n) - (n - 1) / 2.0) / ((n - 1) / 2.0)).astype(np.float32)
    X, Y = np.meshgrid(c.astype(np.float64), (-c).astype(np.float64))
    k = (X.astype(np.float32) * np.float32(0.9)).astype(np.float64)
    phi = (Y.astype(np.float32) * np.float32(4)).astype(np.float64)
    cases = [("ell_int_Kcomp(x * 0.9)", sp.ellipk(k * k)), ("ell_int_Ecomp(x * 0.9)", sp.ellipe(k * k)),
             ("ell_int_F(y * 4, x * 0.9)", sp.ellipkinc(phi, k * k)), ("ell_int_E(y * 4, x * 0.9)", sp.ellipeinc(phi, k * k)),
             ("ell_int_RC(x + 1.5, y + 1.2)", sp.elliprc(X + 1.5, Y + 1.2)),
             ("ell_int_RD(x + 1.5, y + 1.2, 0.7)", sp.elliprd(X + 1.5, Y + 1.2, 0.7)),
             ("ell_int_RF(x + 1.5, y + 1.2, 0.7)", sp.elliprf(X + 1.5, Y + 1.2, 0.7)),
             ("ell_int_RJ(x + 1.5, y + 1.2, 0.7, 2.5)", sp.elliprj(X + 1.5, Y + 1.2, 0.7, 2.5))]
    for expr, want in cases:
        got = run(expr)
        assert np.max(np.abs(got - want) / np.abs(want)) < 5e-7, expr
    gp, gd = run("ell_int_P(y * 4, x * 0.9, 0.3)"), run("ell_int_D(y * 4, x * 0.9, 0)")
    for i in range(0, n, 7):
        for j in range(0, n, 5):
            p, kk = phi[i, j], k[i, j]
            wp = quad(lambda t: 1 / ((1 + 0.3 * np.sin(t) ** 2) * np.sqrt(1 - kk * kk * np.sin(t) ** 2)), 0, p)[0]
            wd = quad(lambda t: np.sin(t) ** 2 / np.sqrt(1 - kk * kk * np.sin(t) ** 2), 0, p)[0]
            assert abs(gp[i, j] - wp) <= 5e-7 * abs(wp) + 1e-9 and abs(gd[i, j] - wd) <= 5e-7 * abs(wd) + 1e-9, (i, j)
    assert np.isnan(run("ell_int_Kcomp(1.5 + x * 0)")).all()           # k^2 >= 1: domain error


def test_pair_mode_engages_for_arithmetic_filters(monkeypatch):
    """hipgen.cpp pair mode (two pixels per work-item step in lockstep): chosen for the
    headline Mandelbrot filter with its parameters baked in (small bodies by default; MMHIP_PAIR=1 takes
    every covered body, =0 none) and for the arithmetic-only fuzz filters; not for filters that fetch
    pixels or call libm."""
    from tests.fuzz_filters import make_filter_arith
    monkeypatch.delenv("MMHIP_PAIR", raising=False)
    marker = "mm_p += 2)"
    flt = F.load("mandelbrot", specialize=True)
    assert marker in flt.specialized({}).kernel_source and marker not in flt.kernel_source
    monkeypatch.setenv("MMHIP_PAIR", "1")
    assert marker in F.load("mandelbrot").kernel_source
    assert sum(marker in mm.Filter(make_filter_arith(s)).kernel_source for s in range(40)) >= 35
    assert marker not in F.load("pond").kernel_source and marker not in F.load("ident").kernel_source
    monkeypatch.setenv("MMHIP_PAIR", "0")
    assert marker not in F.load("mandelbrot").kernel_source


def test_reference_abi_tier_says_what_it_does_not_take():
    """The backend reports through the host's error_string like the reference's backends (cc.c:653-693): a closure image
    that meets its native filter inside a loop would be one image per iteration."""
    import ctypes as C
    from mathmap_amd._lib import selftest_lib
    src = """
filter inner (image in, float k: 0-2 (1.0))
  in(xy * k)
end
filter outer (image in, float s: 0-1 (0.03), int n: 1-4 (2))
  i = 0; acc = rgba:[0, 0, 0, 0];
  while i < n do
    b = gaussian_blur(inner(in, 0.5 + i * 0.1), s, s);
    acc = acc + b(xy) * 0.5;
    i = i + 1
  end;
  acc
end
"""
    img = np.ascontiguousarray(F.synthetic_image(64, 48, seed=3))
    got = np.zeros((48, 64, 4), np.uint8)
    rc = selftest_lib().mmhip_selftest_abi_roundtrip(src.encode(), 1, img.ctypes.data_as(C.c_void_p), 64, 48, 3, 64, 48, 0.25, 2,
                                                     got.ctypes.data_as(C.c_void_p))
    msg = selftest_lib().mmhip_selftest_error().decode()
    assert rc != 0 and ("loop" in msg or "frame-constant" in msg), msg


@pytest.mark.parametrize("inc", [1, 2, 3, 5])
def test_oracle_strided_bilinear_fetch_against_a_numpy_restatement(inc):
    """get_orig_val_intersample_pixel with drawable_get_pixel_inc = inc (builtins.c:182-245; the GIMP preview's strided
    source): the oracle's C function called directly on random pixel coordinates (identity pixel mapping: scale 1,
    middle 0), against the same arithmetic restated in numpy with every rounding spelled out.  No reference vector
    exists for this branch (only the GIMP dialog reaches it)."""
    import ctypes as C
    from oracle import ccgen
    lib = ccgen.runtime_library()
    fn = lib.mmo_get_orig_val_intersample_pixel
    fn.restype = C.c_uint
    fn.argtypes = [C.POINTER(ccgen._Args), C.c_float, C.c_float, C.POINTER(ccgen._ImageDesc), C.c_int]
    rng = np.random.default_rng(40 + inc)
    w, h = 37, 23
    img = np.ascontiguousarray(rng.integers(0, 256, (h, w, 4), dtype=np.uint8))
    d = ccgen._ImageDesc()
    d.data, d.w, d.h, d.kind, d.num_frames, d.channels = img.ctypes.data, w, h, 0, 1, 4
    d.scale_x = d.scale_y = 1.0
    d.middle_x = d.middle_y = 0.0
    a = ccgen._Args()
    a.pixel_inc = inc
    a.edge_color_x, a.edge_color_y = 0x11223344, 0x55667788
    f32 = np.float32

    def tap(x, y):
        if x < 0 or x >= w:
            return [(a.edge_color_x >> s) & 0xff for s in (24, 16, 8, 0)]
        if y < 0 or y >= h:
            return [(a.edge_color_y >> s) & 0xff for s in (24, 16, 8, 0)]
        return [int(v) for v in img[y, x]]

    def axis(v):
        if inc > 1:
            v = f32(np.float64(v) - inc / 2.0)
            v1 = int(np.floor(np.float64(f32(v / f32(inc)))) * inc)
            return v1, v1 + inc, f32(f32(v - f32(v1)) / f32(inc))
        v1 = int(np.floor(np.float64(v)))
        return v1, v1 + 1, f32(v - f32(v1))

    for _ in range(4000):
        x, y = f32(rng.uniform(-6, w + 6)), f32(rng.uniform(-6, h + 6))
        if rng.integers(0, 4) == 0:
            x = f32(np.round(x))                 # on the grid
        x1, x2, fx = axis(x)
        y1, y2, fy = axis(f32(-f32(-y)))         # the pixel mapping negates y: pass -y, get y
        gx, gy = f32(f32(1.0) - fx), f32(f32(1.0) - fy)
        weights = [f32(gx * gy), f32(gx * fy), f32(fx * gy), f32(fx * fy)]
        taps = [tap(x1, y1), tap(x1, y2), tap(x2, y1), tap(x2, y2)]
        want = 0
        for c in range(4):
            acc = f32(f32(taps[0][c]) * weights[0])
            for k in (1, 2, 3):
                acc = f32(acc + f32(f32(taps[k][c]) * weights[k]))
            want = (want << 8) | (int(np.rint(acc)) & 0xff)
        got = fn(C.byref(a), float(x), float(-y), C.byref(d), 0)
        assert got == want, (inc, float(x), float(y), hex(got), hex(want))


def test_oracle_native_results_are_cached_per_argument_set():
    """A native call site inside a loop runs once per iteration -- and, in the oracle, once per iteration *per pixel*.
    The reference caches results per invocation under (filter, arguments) (native-filters/cache.c:110-156); the oracle's
    memo must do the same: the loop equals the chain written out, results of earlier iterations stay readable, and the
    frame takes milliseconds (one entry per call site recomputed every blur for every pixel: minutes)."""
    import time
    import mathmap_amd as mm
    from oracle.ccgen import CpuFilter
    loop = """filter keep (image in, float s: 0-1 (0.02))
      img = in; prev = in; i = 0;
      while i < 3 do prev = img; img = gaussian_blur(img, s * (i + 1), s); i = i + 1 end;
      prev(xy) * 0.5 + img(xy) * 0.5
    end"""
    flat = ("filter u (image in, float s: 0-1 (0.02)) p = gaussian_blur(in, s, s); q = gaussian_blur(p, s * 2, s); "
            "w = gaussian_blur(q, s * 3, s); q(xy) * 0.5 + w(xy) * 0.5 end")
    w, h = 96, 64
    img = F.synthetic_image(w, h, seed=11)
    t0 = time.time()
    a = CpuFilter(mm.Filter(loop).ir_json_raw).render(w, h, images={"in": img})
    assert time.time() - t0 < 20
    b = CpuFilter(mm.Filter(flat).ir_json_raw).render(w, h, images={"in": img})
    assert np.array_equal(a, b)
    assert not np.array_equal(a, CpuFilter(mm.Filter("filter i (image in) in(xy) end").ir_json_raw).render(w, h, images={"in": img}))


def test_oracle_render_samples_a_float_map_behind_its_resize_wrapper():
    """render_image returns a *plain* float map as it is (builtins.c:273-274); the image filter code hands to render()
    is behind a resize wrapper (IMAGE_RESIZE) and is sampled into a new map (builtins.c:303-343): a copy on a square
    frame; on 96 x 64 every new row r holds source row lrintf(1.5 (r - 31.5) + 31.5) -- or zeros where that is outside."""
    import mathmap_amd as mm
    from oracle.ccgen import CpuFilter
    blur = "filter f (image in) b = gaussian_blur(in, 0.06, 0.03); b(xy) end"
    both = "filter f (image in) b = gaussian_blur(in, 0.06, 0.03); rr = render(b); rr(xy) end"
    for w, h in ((64, 64), (96, 64)):
        img = F.synthetic_image(w, h, seed=15)
        a = CpuFilter(mm.Filter(blur).ir_json_raw).render(w, h, images={"in": img}, floatmap=True)
        b = CpuFilter(mm.Filter(both).ir_json_raw).render(w, h, images={"in": img}, floatmap=True)
        if w == h:
            assert np.array_equal(a, b)
            continue
        # the frame's rows sample the new map's rows 11..52 (its unit coordinates +-0.663); those hold the blur's rows
        # lrintf(1.5 (R - 31.5) + 31.5); the blur itself is sampled at rows lrintf(0.663 (r - 31.5) + 31.5) * 1.5 ...: compare
        # through the row maps
        f32 = np.float32
        def row_of_unit(y, n):          # get_floatmap_pixel: lrintf(ay y + by), ay = -(n - 1) / 2
            by = f32(f32(n - 1) / 2.0)
            ay = f32(by * f32(-1.0))
            return int(np.rint(f32(f32(ay * y) + by)))
        rows_differ = 0
        for r in range(h):
            y = f32(f32((-(r) + (h - 1) / 2.0) / ((h - 1) / 2.0)) * f32(h / w))      # CALC_VIRTUAL_Y times H / max(W, H)
            new_row = row_of_unit(y, h)                                              # rr is a plain map: no factor
            fy = f32(f32(f32(new_row) - f32((h - 1) / 2.0)) / f32(-(h - 1) / 2.0))   # render_image's own coordinate of that row
            src_row = row_of_unit(f32(fy * f32(1.5)), h)
            direct_row = row_of_unit(f32(y * f32(1.5)), h)                           # b(xy): through the wrapper
            if 0 <= src_row < h:
                # rr's row `new_row' is the blur's row `src_row'; the frame's row r of `blur' is the blur's row `direct_row'
                if src_row == direct_row:
                    assert np.array_equal(b[r], a[r]), r
                else:
                    rows_differ += 1
            else:
                assert not b[r].any(), r
                rows_differ += 1
        assert rows_differ > 0


def test_oracle_closure_on_the_result_of_a_native_filter_on_a_closure():
    """`c = render(inner(b, 1.2))` with `b = gaussian_blur(inner(in, 0.7), ..)`: the render code of the second closure
    evaluates the main filter's code once more, and that code holds the blur of the *first* closure -- whose render it
    needs (closure k's renders are the main filter's renders 0 .. k-1).  Pinned against the flattened equivalent: on a
    square frame render(closure)(xy) is the closure's body at the pixel, at t = 0."""
    import mathmap_amd as mm
    from oracle.ccgen import CpuFilter
    inner = "filter inner (image in, float k: 0-2 (1.0))\n  in(xy * k) * 0.8 + rgba:[t * 0.3, 0, 0.1, 0]\nend\n\n"
    head = "filter f (image in, float s: 0-1 (0.02))\n  b = gaussian_blur(inner(in, 0.7), s, s * 2);\n"
    nested = inner + head + "  c = render(inner(b, 1.2));\n  c(xy)\nend\n"
    flat = inner + head + "  b(xy * 1.2) * 0.8 + rgba:[0, 0, 0.1, 0]\nend\n"
    w = h = 96
    img = F.synthetic_image(w, h, seed=17)
    a = CpuFilter(mm.Filter(nested).ir_json_raw).render(w, h, images={"in": img}, t=0.25)
    b = CpuFilter(mm.Filter(flat).ir_json_raw).render(w, h, images={"in": img}, t=0.25)
    assert np.array_equal(a, b)
