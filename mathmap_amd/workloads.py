"""The benchmark / parity workloads, in the .mm filter language.

Provenance, plainly: these are the reference's own example scripts -- the input programs that
BASELINE.json names (examples/Utilities/Ident.mm, examples/Render/Mandelbrot.mm,
examples/Distorts/Pond.mm, examples/Blur/Gaussian Blur.mm, examples/Map/Droste.mm, and
tests/{Apply,Circle,Closure,Twice}.mm) -- retyped with local variables renamed and comments
reworded.  They are not independent work: the algorithms, the user-value names (the filters'
public interface, ``-Dname=value``) and the statement structure are the reference's.  They are
kept in the tree only because the GPU box has no reference tree to read them from; where the
reference is present, tests/test_cpu_suite.py::test_workload_text_equals_reference_script renders
each text and the reference script it restates and requires identical output, and the GPU suite
additionally runs every reference script from its compiled IR (tests/golden/ir*/, made by
tests/make_ir_fixtures.py).  GAUSS_DIRECT, CLOSURE_*, CURVE_GRADIENT, TREE_VECTOR, RECURSIVE,
CONVOLVE, HALF_CONVOLVE and VISUALIZE_FFT are small test filters written for this project.
"""

# config 0: examples/Utilities/Ident -- plumbing: output = input sampled at xy
IDENT = """
filter ident (image in)
  in(xy)
end
"""

# config 1: examples/Render/Mandelbrot -- quaternion z -> z*z + p escape-time
# iteration, grey level = iterations / num_iterations
MANDELBROT = """
filter mandelbrot (float pj: -2-2 (0), float pk: -2-2 (0),
                   float c1: -2-2 (0), float ci: -2-2 (0), float cj: -2-2 (0), float ck: -2-2 (0),
                   int num_iterations: 2-256 (32))
  pos = ri:xy;
  offset = quat:[pos[0], pos[1], pj, pk];
  z = quat:[c1, ci, cj, ck];
  n = 0;
  while abs(z) < 2 && n < (num_iterations - 1) do
    z = z * z + offset;
    n = n + 1
  end;
  grayColor(n / num_iterations)
end
"""

# config 4: examples/Distorts/Pond -- radial sine displacement animated by t
POND = """
filter pond (image in, float height: 0-1 (0.05), float wavelength: 0-1 (0.04))
  in(ra + ra:[sin(r / wavelength + t * 2 * pi) * height, 0])
end
"""

# config 3: examples/Blur/Gaussian Blur -- native gaussian_blur with the deviation given
# as a fraction of the larger image dimension
GAUSSIAN_BLUR = """
stretched filter gaussian (stretched image in, float dev: 0 - 0.5)
  px = pixelSize(in);
  longest = max(px[0], px[1]);
  soft = gaussian_blur(in, dev * longest / px[0], dev * longest / px[1]);
  soft(xy)
end
"""

# native gaussian_blur with the two deviations passed straight through (bench: sigma in
# pixels = dev * (W-1)/2, native-filters/gauss.c:659-660)
GAUSS_DIRECT = """
stretched filter gauss_direct (stretched image in, float hdev: 0-1 (0.01), float vdev: 0-1 (0.01))
  soft = gaussian_blur(in, hdev, vdev);
  soft(xy)
end
"""

# config 2: examples/Map/Droste.mm of the reference (Escher's Droste effect after Leys / Breic), restated
# statement by statement with locals renamed (r1 -> rin, retwist -> twist, colorSoFar -> acc ...).
DROSTE = """
pixel
filter droste (pixel image in,
    float InnerRadius: 1 - 100 (25), int OuterRadius: 1 - 100 (100),
    float Periodicity: -6 - 6 (1), int Strands: -6 - 6 (1),
    int Zoom: 1-100 (1), int Rotate: -360-360 (0),
    int XShift: -100 - 100 (0), int YShift: -100 - 100 (0),
    int XCenterShift: -100 - 100 (0), int YCenterShift: -100 - 100 (0),
    int StartingLevel: 1-20 (1), int NumberOfLevels: 1-20 (10), int LevelFrequency: 1-10 (1),
    bool ShowBothPoles, int PoleRotation: -180-180 (90), int PoleLong: -100-100 (0), int PoleLat: -100-100 (0),
    bool TilePoles, bool HyperDroste, int FractalPoints: 1-10 (1),
    bool AutoSetPeriodicity, bool NoTransparency, bool ExternalTransparency, bool MirrorEffect,
    bool Untwist, bool DoNotFlattenTransparency, bool ShowGrid, bool ShowFrame)

  # user parameters -> working variables
  rin = InnerRadius/100;
  rout = OuterRadius/100;
  per = Periodicity;
  strands = Strands;
  cshiftx = XCenterShift/100;
  cshifty = YCenterShift/100;
  shiftx = (XShift*W/X)/100;
  shifty = (YShift*H/Y)/100;
  byAlpha = !(NoTransparency);
  alphaInside = !(ExternalTransparency);
  lookOut = StartingLevel;
  showEvery = LevelFrequency;
  twist = !(Untwist);

  if (AutoSetPeriodicity) then
    per = strands/2 * (1+sqrt(1-(log(rout/rin)/pi)^2));
  end;

  if per > 0 then
    rot = -(pi/180) * Rotate;
  else
    rot = (pi/180) * Rotate;
  end;

  zm = ((Zoom+InnerRadius-1)/100);
  eps = .01;

  # viewport
  if (twist) then
    bx = [-rout,rout];
    by = [-rout,rout];
  else
    by = [0,2.1*pi];
    bx = [-log(rout/rin), log(rout/rin)];
  end;

  small = min(W, H);
  mid = ri:[0.5*(bx[0]+bx[1]),0.5*(by[0]+by[1])];
  span = xy:[bx[1]-bx[0], by[1]-by[0]];
  aspect = W/H;
  span[0] = span[1]*aspect;
  bx = [mid[0]-0.5*span[0],mid[0]+0.5*span[0]];
  z = ri:[bx[0]+(bx[1]-bx[0])*(x+W/2)/W,by[0]+(by[1]-by[0])*(y+H/2)/H];

  if (twist) then
    z0 = z;
    z = z - ri:[shiftx,shifty];
    z = mid+(z-mid)/zm*exp(-I*rot);
  else
    z0 = rin*exp(z);
    z0 = z0*Zoom*exp(I*rot);
  end;

  if ShowBothPoles then
    th = (pi/180)*PoleRotation;
    u = z[0];
    v = z[1];
    dv = .5 * (1+ u^2 + v^2 + ((1-u^2-v^2) * cos(th)) - (2 * u * sin(th)));
    u = u * cos(th) + (0.5*(1 - u^2 - v^2) * sin(th));
    z = ri:[u,v];
    z = z/dv;
  else
    if HyperDroste then
      z = sin(z);
    end;
    if TilePoles then
      z = z^FractalPoints;
      z = tan(2*z);
    end
  end;

  plat = (PoleLat*W/X)/100;
  plon = (PoleLong*W/X)/100;
  z = z + ri:[plat,plon];

  if (twist) then
    lz = log(z/rin);
  else
    lz = z;
  end;

  # the twist itself
  alpha = atan(strands/per*log(rout/rin)/(2*pi));
  cosa = cos(alpha);
  beta = cosa*exp(I*alpha);

  if (strands > 0) then
    ang = 2*pi*per;
  else
    ang = -2*pi*per;
  end;

  if MirrorEffect then
    ang = ang/Strands;
  end;

  z = per*lz/beta;
  gridlz = z;
  plainlz = lz;
  z = rin*exp(z);

  if (byAlpha && lookOut > 0) then
    if (!alphaInside) then ratio = rout/rin*exp( I*ang); end;
    if ( alphaInside) then ratio = rin/rout*exp(-I*ang); end;
    z = z * (ratio^lookOut)/1;
  end;

  acc = rgba:[0,0,0,0];
  left = 1;
  sx = small/2*(z[0]+cshiftx);
  sy = small/2*(z[1]+cshifty);
  sp = xy:[sx,sy];

  tap = in(sp);
  acc = acc = acc + (tap*(alpha(tap)*left));
  left = left*(1-alpha(tap));
  dir = 0;

  if (byAlpha) then
    if ( alphaInside && left > eps) then
      dir = -1;
    end;
    if (!alphaInside && left > eps) then
      dir = 1;
    end;
  else
    rad = sqrt(z[0]*z[0]+z[1]*z[1]);
    if (rad < rin) then dir = -1; end;
    if (rad > rout) then dir = 1; end;
  end;

  if (dir < 0) then
    ratio = rout/rin*exp( I*ang);
  end;

  if (dir > 0) then
    ratio = rin/rout*exp(-I*ang);
  end;

  if (showEvery > 1) then
    ratio = exp(log(ratio)*showEvery);
  end;

  level = StartingLevel;
  lastLevel = NumberOfLevels+StartingLevel-1;

  while (dir != 0 && level < lastLevel) do
    lz = z*ratio;
    z = lz;
    gridlz = gridlz+ri:[0,-dir*ang];
    sx = small/2*(z[0]+cshiftx);
    sy = small/2*(z[1]+cshifty);
    sp = xy:[sx,sy];
    dir = 0;
    if (byAlpha) then
      tap = in(sp);
      acc = acc + (tap*(alpha(tap)*left));
      left = left*(1-alpha(tap));
      if ( alphaInside && left > eps) then dir = -1; end;
      if (!alphaInside && left > eps) then dir = 1; end;
    else
      rad = sqrt(z[0]*z[0]+z[1]*z[1]);
      acc = in(sp);
      if (rad < rin) then dir = -1; end;
      if (rad > rout) then dir = 1; end;
    end;
    level = level+1;
  end;

  tap = acc;

  if (ShowGrid) then
    g = xy:[(plainlz[0]+10*log(rout/rin))%log(rout/rin), (plainlz[1]+10*2*pi)%(2*pi)];
    if (g[0] < eps || g[0] > (log(rout/rin)-eps) || g[1] < eps || g[1] > (2*pi-eps)) then
      tap = rgba:[0,1,0,1];
    end;
    g = xy:[(gridlz[0]+10*log(rout/rin))%log(rout/rin), (gridlz[1]+10*2*pi)%(2*pi)];
    if (g[0] < eps || g[0] > (log(rout/rin)-eps) || g[1] < eps || g[1] > (2*pi-eps)) then
      tap = rgba:[0,0,1,1];
    end;
  end;

  if (ShowFrame) then
    g = xy:[z0[0],z0[1]];
    if (g[0] < (aspect*rout) && g[0] > -(aspect*rout) && g[1] < rout && g[1] > -rout) then
      ex = min((aspect*rout)-g[0], g[0]+(aspect*rout));
      ey = min(rout-g[1], g[1]+rout);
      if (ex < (4*eps) || ey < (4*eps)) then
        tap = rgba:[1,1,1,1];
      end;
      if (ex < (2*eps) || ey < (2*eps)) then
        tap = rgba:[0,0,0,1];
      end;
    else
      tap = rgba:[0.75*red(tap),0.75*green(tap),0.75*blue(tap),1];
    end;
  end;

  if !(DoNotFlattenTransparency) then
    tap = rgba:[tap[0], tap[1], tap[2], 1];
  end;

  tap
end
"""

# small filters exercising closures / filter calls (cf. the reference's tests/*.mm)
CLOSURE_CALL = """
filter disc (float radius)
  grayColor(if r < radius then t else 0 end)
end

filter main (image in, float radius: 0-1.5 (1))
  in(xy) * disc(radius, xy, 1-t)
end
"""

CLOSURE_VALUE = """
filter disc (float radius)
  grayColor(if r < radius then 1 else 0 end)
end

filter main (float radius: 0-1.5 (1))
  c = disc(radius);
  c(xy)
end
"""

CLOSURE_ARG = """
filter disc (float radius)
  grayColor(if r < radius then 0.5-t else 0 end)
end

filter both (image i1, image i2)
  i1(xy) * i2(xy)
end

filter main (image in, float radius: 0-1.5 (1))
  both(in, disc(radius), xy)
end
"""

NESTED_CALLS = """
filter dbl (image in)
  in(xy) * 2
end

filter hlf (image in)
  in(xy) / 2
end

filter main (image in)
  hlf(dbl(in), xy)
end
"""

# examples/Utilities/Visualize FFT.mm, examples/Combine/Convolve.mm, Half Convolve.mm
# (the native FFT filters, native-filters/convolve.c)
VISUALIZE_FFT = """
stretched filter util_visualize_fft (stretched image in, bool ignore_alpha (1))
  visualize_fft(in, ignore_alpha, xy)
end
"""

CONVOLVE = """
filter combine_convolve (image in, image kernel, bool normalize (1), bool copy_alpha (1))
  convolved = convolve(in, kernel, normalize, copy_alpha);
  convolved(xy)
end
"""

HALF_CONVOLVE = """
filter combine_half_convolve (image in, image mask, bool copy_alpha (1))
  convolved = half_convolve(in, mask, copy_alpha);
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


def test_curve():
    """Deterministic non-default curve (squares) shared by tests and the ABI self-test."""
    import numpy as np
    i = np.arange(1024, dtype=np.float32)
    return (i * i) / np.float32(1023.0 * 1023.0)


def test_gradient():
    import numpy as np
    i = np.arange(1024, dtype=np.uint32)
    return ((i >> 2) << 24) | (((1023 - i) >> 2) << 16) | np.uint32(0x40 << 8) | np.uint32(0xFF)


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

ALL = {
    "ident": IDENT,
    "mandelbrot": MANDELBROT,
    "pond": POND,
    "gaussian_blur": GAUSSIAN_BLUR,
    "gauss_direct": GAUSS_DIRECT,
    "droste": DROSTE,
    "closure_call": CLOSURE_CALL,
    "closure_value": CLOSURE_VALUE,
    "closure_arg": CLOSURE_ARG,
    "nested_calls": NESTED_CALLS,
    "visualize_fft": VISUALIZE_FFT,
    "convolve": CONVOLVE,
    "half_convolve": HALF_CONVOLVE,
    "curve_gradient": CURVE_GRADIENT,
    "tree_vector": TREE_VECTOR,
    "recursive": RECURSIVE,
    "recursive_data": RECURSIVE_DATA,
    "recursive_mutual": RECURSIVE_MUTUAL,
}


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
