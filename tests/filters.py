"""Filters used by the tests, `bench.py` and `smoke()`.

Two kinds, kept apart on purpose:

* **The reference's own filters** -- the programs BASELINE.json names (examples/Utilities/Ident,
  Render/Mandelbrot, Distorts/Pond, Blur/Gaussian Blur, Map/Droste, the native FFT filters) and the
  closure cases of the reference's test-suite (tests/{Apply,Circle,Closure,Twice}.mm) -- are **not kept as
  text**.  They are loaded from the committed IR fixtures (`tests/golden/ir_examples/`, `tests/golden/ir/`:
  compiler output of this project's front-end for the reference's scripts, generated in the build container
  by tests/make_ir_fixtures.py and pinned by test_ir_fixtures_equal_a_fresh_compile) through the IR-level
  entry point `mmhip_compile_ir_json`.
* **Project-written probes** (GAUSS_DIRECT, CONVOLVE, RECURSIVE*, CLOSURE_TIMED_ARG, CURVE_GRADIENT,
  TREE_VECTOR): small .mm texts written for this project to exercise one feature each.

`load(name, **options)` returns a compiled `mathmap_amd.Filter` for either kind.
"""
import gzip
import os

_GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# name -> IR fixture (relative to tests/golden/, without .json.gz)
REFERENCE_IR = {
    "ident": "ir_examples/Utilities__Ident",                  # BASELINE config 0
    "mandelbrot": "ir_examples/Render__Mandelbrot",           # config 1
    "droste": "ir_examples/Map__Droste",                      # config 2
    "gaussian_blur": "ir_examples/Blur__Gaussian_Blur",       # config 3 (the example's own parameterisation)
    "pond": "ir_examples/Distorts__Pond",                     # config 4
    "visualize_fft": "ir_examples/Utilities__Visualize_FFT",
    "half_convolve": "ir_examples/Combine__Half_Convolve",
    "closure_value": "ir/apply",                              # tests/Apply.mm: a closure as a value, applied
    "closure_call": "ir/circle",                              # tests/Circle.mm: a filter called with xy, t
    "closure_arg": "ir/closure",                              # tests/Closure.mm: a closure as an image argument
    "nested_calls": "ir/twice",                               # tests/Twice.mm: nested filter calls
}

# native gaussian_blur with the two deviations passed straight through (bench: sigma in
# pixels = dev * (W-1)/2, native-filters/gauss.c:659-660)
GAUSS_DIRECT = """
stretched filter gauss_direct (stretched image in, float hdev: 0-1 (0.01), float vdev: 0-1 (0.01))
  soft = gaussian_blur(in, hdev, vdev);
  soft(xy)
end
"""


CONVOLVE = """
filter combine_convolve (image in, image kernel, bool normalize (1), bool copy_alpha (1))
  convolved = convolve(in, kernel, normalize, copy_alpha);
  convolved(xy)
end
"""


# A recursive filter (the shape of examples/Map/IFS Functional.mm: the recursion depth is a
# user value).  The recursive application is a run-time call of filter_tree; with the user values
# baked in the lowering unrolls it instead.
RECURSIVE = """
filter shrink (image in, float s)
  in(xy / s)
end

filter tree (image in, int depth: 1-16 (4), float s: 0-1 (0.6))
  if depth < 2 then
    in(xy)
  else
    in(xy) * 0.5 + shrink(tree(in, depth - 1, s), s, xy) * 0.5
  end
end
"""


# recursion whose depth differs per pixel: it follows the image content along the walk
RECURSIVE_DATA = """
filter walk (image in, float budget: 0-20 (9))
  p = in(xy);
  if budget < 1 || p[0] < 0.25 then
    p
  else
    walk(in, budget - 1 - 3 * p[1], xy * 0.9 + xy:[0.07, -0.03]) * 0.7 + p * 0.3
  end
end
"""


# a recursive filter that is not the main one (inlined at its two call sites, calling itself at run time),
# with rand() inside the callee; n above MM_MAX_CALL_DEPTH exercises the cut-off
RECURSIVE_MUTUAL = """
filter fade (image in, int n: 0-64 (6))
  if n < 1 then
    in(xy)
  else
    fade(in, n - 1, xy * 0.95) * 0.9 + grayColor(rand(0, 0.1))
  end
end

filter echoes (image in, int n: 0-64 (6))
  fade(in, n, xy) + fade(in, n / 2, xy:[-x, y]) * 0.25
end
"""


# a closure image for a native filter whose argument depends on t, and whose body reads t and frame
CLOSURE_TIMED_ARG = """
filter inner2 (image in, float k: 0-2 (1.0))
  in(xy * k) * 0.8 + rgba:[t * 0.5, frame * 0.01, 0.1, 0]
end

filter outer (image in, float s: 0-1 (0.03), float k: 0-2 (0.5))
  b = gaussian_blur(inner2(in, k * (1 + t)), s, s);
  b(xy)
end
"""


# curve and gradient user values (the shape of examples/Colors/Colorify.mm)
CURVE_GRADIENT = """
filter cg (image in, curve tone, gradient colors)
  p = in(xy);
  c = colors(tone(gray(p)));
  rgba:[c[0], c[1], c[2], c[3] * p[3]]
end
"""


# dynamic tuple subscripts ("tree vectors", compiler.c:1840-2040, tree_vectors.c): reads and writes
# with run-time indices (clamped), on a variable and on an expression value
TREE_VECTOR = """
filter tv (int k: 0-8 (2))
  v = rgba:[x, y, x*y, 1];
  i = floor((x + 1) * 2.5) - 1;
  w = v[i];
  v[i + 1] = 0.25;
  u = ([0.1, 0.5, 0.9])[floor(abs(y) * 40)];
  rgba:[w, v[2], u, v[k]]
end
"""


def test_curve():
    """Deterministic non-default curve (squares) shared by tests and the ABI self-test."""
    import numpy as np
    i = np.arange(1024, dtype=np.float32)
    return (i * i) / np.float32(1023.0 * 1023.0)


def test_gradient():
    import numpy as np
    i = np.arange(1024, dtype=np.uint32)
    return ((i >> 2) << 24) | (((1023 - i) >> 2) << 16) | np.uint32(0x40 << 8) | np.uint32(0xFF)


SOURCES = {
    "gauss_direct": GAUSS_DIRECT,
    "convolve": CONVOLVE,
    "curve_gradient": CURVE_GRADIENT,
    "tree_vector": TREE_VECTOR,
    "recursive": RECURSIVE,
    "recursive_data": RECURSIVE_DATA,
    "recursive_mutual": RECURSIVE_MUTUAL,
    "closure_timed_arg": CLOSURE_TIMED_ARG,
}
NAMES = sorted(list(REFERENCE_IR) + list(SOURCES))


def ir_text(name):
    """The IR dump (JSON text) of one of the reference's filters, from the committed fixture."""
    with gzip.open(os.path.join(_GOLDEN, REFERENCE_IR[name] + ".json.gz"), "rt") as f:
        return f.read()


def load(name, **options):
    """A compiled Filter: the reference's filters from their IR fixtures, the project's probes from text.
    `options` are those of mathmap_amd.Filter (intersample, specialize, tile_w, constants ...)."""
    import mathmap_amd as mm
    if name in REFERENCE_IR:
        return mm.Filter(ir_json=ir_text(name), **options)
    return mm.Filter(SOURCES[name], **options)


def image_names(flt):
    """Names of the filter's image user values."""
    from mathmap_amd.api import UV_IMAGE
    return [u["name"] for u in flt.uservals if u["kind"] == UV_IMAGE]


def synthetic_image(width, height, seed=1):
    """Deterministic RGB8 test image: smooth gradients plus hashed texture, so gathers
    see non-trivial data.  Returns uint8 [H, W, 3]."""
    import numpy as np
    x = np.arange(width, dtype=np.uint32)[None, :]
    out = np.empty((height, width, 3), np.uint8)
    gx = x * np.uint32(255) // np.uint32(max(width - 1, 1))
    for r0 in range(0, height, 1024):          # row chunks: bounded temporaries at 16384^2
        y = np.arange(r0, min(r0 + 1024, height), dtype=np.uint32)[:, None]
        gy = y * np.uint32(255) // np.uint32(max(height - 1, 1))
        g = ((gx + gy) // np.uint32(2) * np.uint32(3)) // np.uint32(4)
        xy = (x * y) >> np.uint32(3)
        for c in range(3):
            v = (x * np.uint32(131 + 17 * c) + y * np.uint32(71 + 29 * c) + np.uint32(seed * 977 + c * 17)) ^ xy
            out[r0:r0 + y.shape[0], :, c] = ((v & np.uint32(63)) + g).astype(np.uint8)
    return out
