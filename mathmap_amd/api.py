"""Python mirror of the reference's session interface for the per-pixel path.

Names follow the reference: a *filter* is compiled (``compile_mathmap``,
mathmap_common.c:503), *invoked* on a canvas size (``invoke_mathmap``, :746), user
values are set like the CLI's ``-Dname=value`` (mathmap_cmdline.c:756-796) and frames
are rendered (``call_invocation_parallel_and_join``, :1008).  All compute happens in
libmathmap_hip.so on the GPU.
"""
import ctypes as C
import json

import numpy as np

from ._lib import Options, UservalInfo, lib

UV_INT, UV_FLOAT, UV_BOOL, UV_COLOR, UV_CURVE, UV_GRADIENT, UV_IMAGE = range(7)
EDGE_COLOR, EDGE_WRAP, EDGE_REFLECT, EDGE_ROTATE = range(4)


class MathMapError(RuntimeError):
    pass


def _err():
    return lib().mmhip_last_error().decode("utf-8", "replace")


class Filter:
    """A compiled .mm filter (front-end + IR + generated HIP kernel string)."""

    def __init__(self, source="", intersample=True, supersampling=False, edge_x=EDGE_COLOR, edge_y=EDGE_COLOR,
                 tile_w=0, specialize=False, constants=None, ir_json=None, _handle=None, pixel_inc=1):
        """`source`: .mm text; or `ir_json`: an IR dump (mmhip_filter_ir_json_raw / the reference-ABI importer's
        form) -- the IR-level entry point.  `constants` (name -> number) bakes scalar user values in as literals.
        `pixel_inc` > 1: the bilinear fetch interpolates over a source sampled at that stride (the GIMP preview's
        fast image source, builtins.c:186-216)."""
        self._source = source
        self._kwargs = dict(intersample=intersample, supersampling=supersampling, edge_x=edge_x, edge_y=edge_y,
                            tile_w=tile_w, pixel_inc=pixel_inc)
        if _handle is not None:
            self._h = _handle
            return
        o = Options()
        lib().mmhip_default_options(C.byref(o))
        o.intersample = 1 if intersample else 0
        o.supersampling = 1 if supersampling else 0
        o.edge_behaviour_x, o.edge_behaviour_y = edge_x, edge_y
        o.tile_w = tile_w
        o.pixel_inc = pixel_inc
        o.specialize_uservals = 1 if specialize else 0
        if ir_json is not None:
            self._h = lib().mmhip_compile_ir_json(ir_json.encode(), C.byref(o))
        else:
            self._h = lib().mmhip_compile(source.encode(), C.byref(o))
        if not self._h:
            raise MathMapError(_err())
        if constants:
            # bake scalar user values in as literals (the variant the specialising JIT builds lazily)
            base, self._h = self._h, None
            try:
                self._h = self._specialized_handle(base, constants)
            finally:
                lib().mmhip_filter_free(base)

    @staticmethod
    def _specialized_handle(handle, constants):
        names = {}
        for i in range(lib().mmhip_filter_num_uservals(handle)):
            info = UservalInfo()
            lib().mmhip_filter_userval_info(handle, i, C.byref(info))
            names[info.name.decode()] = i
        for k in constants:
            if k not in names:
                raise MathMapError("filter has no user value `%s'" % k)
        idx = (C.c_int * len(constants))(*[names[k] for k in constants])
        val = (C.c_double * len(constants))(*[float(v) for v in constants.values()])
        h = lib().mmhip_filter_specialized(handle, len(constants), idx, val)
        if not h:
            raise MathMapError(_err())
        return h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib().mmhip_filter_free(h)

    @property
    def name(self):
        return lib().mmhip_filter_name(self._h).decode()

    @property
    def uservals(self):
        out = []
        for i in range(lib().mmhip_filter_num_uservals(self._h)):
            info = UservalInfo()
            lib().mmhip_filter_userval_info(self._h, i, C.byref(info))
            out.append(dict(kind=info.kind, index=info.index, name=info.name.decode(),
                            int_min=info.int_min, int_max=info.int_max, int_default=info.int_default,
                            float_min=info.float_min, float_max=info.float_max, float_default=info.float_default,
                            bool_default=info.bool_default, image_flags=info.image_flags))
        return out

    @property
    def ir(self):
        return json.loads(lib().mmhip_filter_ir_json(self._h).decode())

    @property
    def ir_json(self):
        return lib().mmhip_filter_ir_json(self._h).decode()

    @property
    def ir_json_raw(self):
        """The IR before any optimisation pass (what mmhip_compile_ir_json takes and the test
        oracle prints): lowering output only."""
        return lib().mmhip_filter_ir_json_raw(self._h).decode()

    def specialized(self, values=None):
        """The variant with every int/float/bool user value baked in: the declared defaults,
        overridden by `values` (name -> number).  What a render with those values runs."""
        consts = {}
        for u in self.uservals:
            if u["kind"] == UV_INT:
                consts[u["name"]] = u["int_default"]
            elif u["kind"] == UV_FLOAT:
                consts[u["name"]] = u["float_default"]
            elif u["kind"] == UV_BOOL:
                consts[u["name"]] = u["bool_default"]
        for k, v in (values or {}).items():
            if k in consts:
                consts[k] = v
        return Filter(self._source, _handle=self._specialized_handle(self._h, consts), **self._kwargs)

    @property
    def kernel_source(self):
        return lib().mmhip_filter_kernel_source(self._h).decode()

    @property
    def num_native_calls(self):
        return lib().mmhip_filter_num_native_calls(self._h)

    def jit(self, load=False):
        """hiprtc-compiles the kernel string for gfx950; returns the code-object size."""
        n = lib().mmhip_filter_jit(self._h, 1 if load else 0)
        if n < 0:
            raise MathMapError(_err())
        return n

    @property
    def jit_seconds(self):
        return lib().mmhip_filter_jit_seconds(self._h)

    def invoke(self, width, height):
        return Invocation(self, width, height)


class Invocation:
    """A filter bound to a canvas size, user values and input images (all in HBM)."""

    def __init__(self, flt, width, height):
        self.filter = flt
        self.width, self.height = width, height
        self.render_width, self.render_height = width, height
        self._h = lib().mmhip_invoke(flt._h, width, height)
        if not self._h:
            raise MathMapError(_err())
        self._keep = []

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib().mmhip_invocation_free(h)

    def _check(self, rc):
        if rc != 0:
            raise MathMapError(_err())

    def _index(self, name):
        for u in self.filter.uservals:
            if u["name"] == name:
                return u
        raise MathMapError("filter has no user value `%s'" % name)

    def set(self, name, value):
        """Sets a user value by name (the CLI's -Dname=value)."""
        u = self._index(name)
        k, i = u["kind"], u["index"]
        if k == UV_INT:
            self._check(lib().mmhip_set_int(self._h, i, int(value)))
        elif k == UV_FLOAT:
            self._check(lib().mmhip_set_float(self._h, i, float(value)))
        elif k == UV_BOOL:
            self._check(lib().mmhip_set_bool(self._h, i, int(bool(value))))
        elif k == UV_COLOR:
            r, g, b, a = value
            self._check(lib().mmhip_set_color(self._h, i, r, g, b, a))
        elif k == UV_IMAGE:
            self.set_image(name, value)
        else:
            raise MathMapError("cannot set user value `%s'" % name)

    def set_curve(self, name, values):
        """Sets a curve user value: 1024 floats, the sampled curve (userval.h:38,89-96)."""
        u = self._index(name)
        a = np.ascontiguousarray(values, dtype=np.float32)
        if a.shape != (1024,):
            raise MathMapError("a curve has 1024 samples")
        self._check(lib().mmhip_set_curve(self._h, u["index"], a.ctypes.data_as(C.c_void_p)))

    def set_gradient(self, name, rgba):
        """Sets a gradient user value: 1024 packed 0xRRGGBBAA colours (userval.h:39,98-101)."""
        u = self._index(name)
        a = np.ascontiguousarray(rgba, dtype=np.uint32)
        if a.shape != (1024,):
            raise MathMapError("a gradient has 1024 samples")
        self._check(lib().mmhip_set_gradient(self._h, u["index"], a.ctypes.data_as(C.c_void_p)))

    def set_image(self, name, array):
        """Binds a host uint8 array [H,W,3|4] as input drawable (uploaded once to HBM)."""
        u = self._index(name)
        a = np.ascontiguousarray(array, dtype=np.uint8)
        h, w, c = a.shape
        self._check(lib().mmhip_set_image_host(self._h, u["index"], a.ctypes.data_as(C.c_void_p), w, h, c))

    def set_image_device(self, name, device_ptr, width, height, keepalive=None):
        """Binds a packed 0xRRGGBBAA uint32 image already resident in HBM."""
        u = self._index(name)
        if keepalive is not None:
            self._keep.append(keepalive)
        self._check(lib().mmhip_set_image_device(self._h, u["index"], C.c_void_p(device_ptr), width, height))

    def set_native_row_margin(self, margin):
        """Striped frames: let native filters (gaussian_blur) fill only the rows a stripe render
        reads, +- `margin` rows, plus their own halo.  -1 restores whole maps."""
        self._check(lib().mmhip_set_native_row_margin(self._h, margin))

    def set_render_size(self, render_width, render_height):
        """Renders the canvas at another pixel size (the GIMP preview, mathmap.c:2191-2223);
        `render()` then returns an array of that size."""
        self._check(lib().mmhip_set_render_size(self._h, render_width, render_height))
        self.render_width, self.render_height = render_width, render_height

    def set_edge_colors(self, cx, cy):
        self._check(lib().mmhip_set_edge_colors(self._h, cx, cy))

    def enable_timing(self, on=True):
        self._check(lib().mmhip_enable_timing(self._h, 1 if on else 0))

    def last_kernel_ms(self):
        return lib().mmhip_last_kernel_ms(self._h)

    def drain_kernel_ms(self, cap=4096):
        """Pixel-kernel durations (ms) of every timed launch since the last drain, oldest first."""
        buf = (C.c_double * cap)()
        n = lib().mmhip_drain_kernel_ms(self._h, buf, cap)
        if n < 0:
            raise MathMapError(_err())
        return [buf[i] for i in range(n)]

    def drain_native_kernel_ms(self, cap=4096):
        """(label, ms) of every kernel native filters launched themselves since the last drain (gaussian_blur's
        scan kernels), in launch order."""
        names = C.create_string_buffer(cap * 64)
        buf = (C.c_double * cap)()
        n = lib().mmhip_drain_native_kernel_ms(self._h, names, buf, cap)
        if n < 0:
            raise MathMapError(_err())
        return [(names.raw[i * 64:(i + 1) * 64].split(b"\0", 1)[0].decode(), buf[i]) for i in range(n)]

    def direct_native_launches(self):
        """Launches whose pixels a native filter wrote itself (pixel kernel skipped)."""
        return lib().mmhip_direct_native_launches(self._h)

    def render(self, t=0.0, frame=0):
        """Renders the whole frame and returns it as a uint8 [H,W,4] array (RGBA)."""
        out = np.empty((self.render_height, self.render_width, 4), dtype=np.uint8)
        self._check(lib().mmhip_render_host(self._h, frame, t, out.ctypes.data_as(C.c_void_p)))
        return out

    def render_rows(self, out_ptr, first_row, last_row, t=0.0, frame=0, row_stride=None, bpp=4, floatmap=False,
                    stream=0, region=None):
        """Asynchronously renders rows [first_row,last_row) into device memory at out_ptr
        (the reference's calc_lines band, mathmap_common.c:837-846).  With `floatmap` the rows of the output are
        the frame's render width apart (16 * render_width bytes, new_template.c.in:297), whatever the region and row_stride."""
        rx, ry, rw, rh = region if region is not None else (0, 0, self.render_width, self.render_height)
        if row_stride is None:
            row_stride = rw * bpp
        self._check(lib().mmhip_render(self._h, frame, t, rx, ry, rw, rh, first_row, last_row, C.c_void_p(out_ptr),
                                       row_stride, bpp, 1 if floatmap else 0, C.c_void_p(stream)))

    def render_supersampled(self, out_ptr, t=0.0, frame=0, bpp=4, stream=0):
        """The CLI's -o (supersampling) for the whole frame, into device memory at out_ptr."""
        self._check(lib().mmhip_render_supersampled(self._h, frame, t, 0, 0, self.render_width, self.render_height,
                                                    C.c_void_p(out_ptr), self.render_width * bpp, bpp, C.c_void_p(stream)))

    def sync(self):
        self._check(lib().mmhip_sync(self._h))


def device_count():
    return lib().mmhip_device_count()


def set_device(ordinal):
    """One process per GPU: the device later invocations of this thread are created on."""
    if lib().mmhip_set_device(int(ordinal)) != 0:
        raise MathMapError(_err())
