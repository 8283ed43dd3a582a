"""ctypes binding of libmathmap_hip.so (the C ABI declared in include/mmhip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C mathmap_amd/csrc``.
There is deliberately no Python or CPU fallback: if the shared object is missing the
import fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmathmap_hip.so")


class Options(C.Structure):
    _fields_ = [("intersample", C.c_int), ("supersampling", C.c_int),
                ("edge_behaviour_x", C.c_int), ("edge_behaviour_y", C.c_int),
                ("tile_w", C.c_int), ("specialize_uservals", C.c_int), ("pixel_inc", C.c_int), ("reserved", C.c_int * 6)]


class UservalInfo(C.Structure):
    _fields_ = [("kind", C.c_int), ("index", C.c_int), ("name", C.c_char * 64),
                ("int_min", C.c_int), ("int_max", C.c_int), ("int_default", C.c_int),
                ("float_min", C.c_float), ("float_max", C.c_float), ("float_default", C.c_float),
                ("bool_default", C.c_int), ("image_flags", C.c_uint)]


# every symbol include/mmhip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "mmhip_last_error": (C.c_char_p, []),
    "mmhip_version": (C.c_char_p, []),
    "mmhip_default_options": (None, [C.POINTER(Options)]),
    "mmhip_compile": (C.c_void_p, [C.c_char_p, C.POINTER(Options)]),
    "mmhip_compile_specialized": (C.c_void_p, [C.c_char_p, C.POINTER(Options), C.c_int, C.POINTER(C.c_int),
                                               C.POINTER(C.c_double)]),
    "mmhip_compile_ir_json": (C.c_void_p, [C.c_char_p, C.POINTER(Options)]),
    "mmhip_filter_specialized": (C.c_void_p, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "mmhip_filter_free": (None, [C.c_void_p]),
    "mmhip_filter_name": (C.c_char_p, [C.c_void_p]),
    "mmhip_filter_num_uservals": (C.c_int, [C.c_void_p]),
    "mmhip_filter_userval_info": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(UservalInfo)]),
    "mmhip_filter_ir_json": (C.c_char_p, [C.c_void_p]),
    "mmhip_filter_ir_json_raw": (C.c_char_p, [C.c_void_p]),
    "mmhip_filter_kernel_source": (C.c_char_p, [C.c_void_p]),
    "mmhip_filter_num_native_calls": (C.c_int, [C.c_void_p]),
    "mmhip_filter_jit": (C.c_long, [C.c_void_p, C.c_int]),
    "mmhip_filter_jit_seconds": (C.c_double, [C.c_void_p]),
    "mmhip_invoke": (C.c_void_p, [C.c_void_p, C.c_int, C.c_int]),
    "mmhip_invocation_free": (None, [C.c_void_p]),
    "mmhip_set_int": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "mmhip_set_float": (C.c_int, [C.c_void_p, C.c_int, C.c_float]),
    "mmhip_set_bool": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "mmhip_set_color": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float]),
    "mmhip_set_native_row_margin": (C.c_int, [C.c_void_p, C.c_int]),
    "mmhip_drain_kernel_ms": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "mmhip_direct_native_launches": (C.c_long, [C.c_void_p]),
    "mmhip_drain_native_kernel_ms": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int]),
    "mmhip_set_curve": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "mmhip_set_gradient": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "mmhip_set_by_name": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p]),
    "mmhip_set_image_host": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "mmhip_set_image_device": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]),
    "mmhip_set_edge_colors": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "mmhip_set_render_size": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "mmhip_set_sampling_offset": (C.c_int, [C.c_void_p, C.c_float, C.c_float]),
    "mmhip_render": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int,
                               C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mmhip_render_supersampled": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "mmhip_render_host": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_void_p]),
    "mmhip_sync": (C.c_int, [C.c_void_p]),
    "mmhip_enable_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "mmhip_last_kernel_ms": (C.c_double, [C.c_void_p]),
    "mmhip_device_alloc": (C.c_void_p, [C.c_size_t]),
    "mmhip_device_free": (None, [C.c_void_p]),
    "mmhip_copy_to_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "mmhip_copy_to_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "mmhip_device_count": (C.c_int, []),
    "mmhip_set_device": (C.c_int, [C.c_int]),
}

# reference-ABI tier (include/mathmap_hip_backend.h) and its self-test driver
BACKEND_SYMBOLS = {
    "gen_and_load_hip_code": (C.c_void_p, [C.c_void_p, C.POINTER(C.c_void_p), C.c_char_p, C.c_char_p, C.c_void_p]),
    "unload_hip_code": (None, [C.c_void_p]),
    "mathmap_hip_set_get_pixel": (None, [C.c_void_p]),
    "mathmap_hip_invalidate_drawable": (None, [C.c_void_p]),
    "mathmap_hip_release_invocation": (None, [C.c_void_p]),
}

# test scaffolding of the reference-ABI tier: tests/libmathmap_hip_selftest.so (built from csrc/abi_selftest.cpp,
# linked against the product library; not part of it)
SELFTEST_SYMBOLS = {
    "mmhip_selftest_abi_roundtrip": (C.c_int, [C.c_char_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                               C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "mmhip_selftest_error": (C.c_char_p, []),
    "mmhip_selftest_set_pixel_inc": (None, [C.c_int]),
    "mmhip_selftest_eval_unary": (C.c_int, [C.c_int, C.c_uint, C.c_ulonglong, C.c_void_p]),
    "mmhip_selftest_eval_binary": (C.c_int, [C.c_int, C.c_ulonglong, C.c_ulonglong, C.c_void_p]),
}
SELFTEST_PATH = os.path.join(os.path.dirname(_HERE), "tests", "libmathmap_hip_selftest.so")
_selftest = None


def selftest_lib():
    global _selftest
    if _selftest is None:
        lib()
        l = C.CDLL(SELFTEST_PATH, mode=C.RTLD_GLOBAL)      # it plays the host program: the backend looks host functions up by name
        for name, (res, args) in SELFTEST_SYMBOLS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _selftest = l
    return _selftest

_lib = None


def lib():
    """Loads the shared library (once) and declares the prototypes."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C mathmap_amd/csrc` (there is no CPU fallback)" % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in list(SYMBOLS.items()) + list(BACKEND_SYMBOLS.items()):
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib
