"""ctypes binding of libraytrace_gpu.so (the C ABI declared in include/raytrace_gpu.h).

The product has no CPU fallback: if the shared library is missing this module raises at import of
the symbols, and rendering without a HIP device returns RT_ERR_HIP (raised as RuntimeError).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RT_LIB selects another build of the same library (A/B runs of kernel variants); never a different backend
LIB_PATH = os.environ.get("RT_LIB") or os.path.join(_HERE, "lib", "libraytrace_gpu.so")


class rt_material(C.Structure):
    _fields_ = [("diffuse", C.c_uint8 * 3), ("specular", C.c_uint8 * 3), ("_pad", C.c_uint8 * 2),
                ("specularity", C.c_double), ("reflectivity", C.c_double)]


class rt_stats(C.Structure):
    _fields_ = [("primary_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("tri_pairs", C.c_uint64),
                ("rect_pairs", C.c_uint64), ("exact_evals", C.c_uint64), ("ms_primary", C.c_double),
                ("ms_setup", C.c_double), ("ms_shadow", C.c_double), ("ms_final", C.c_double), ("ms_total", C.c_double),
                ("mfma_blocks_primary", C.c_uint64), ("mfma_blocks_shadow", C.c_uint64),
                ("record_tests", C.c_uint64), ("bounce_rays", C.c_uint64), ("bounce_blocks", C.c_uint64), ("prepare_ms", C.c_double), ("upload_ms", C.c_double),
                ("multi_bands_direct", C.c_uint64), ("multi_bands_staged", C.c_uint64)]


# an `fnorm` closure as a C callback: void fn(int n, const double *xy, double *normals, void *user)
NORMAL_FN = C.CFUNCTYPE(None, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)
RT_OPT_CULL, RT_OPT_FILTER, RT_OPT_HOST_BANDS, RT_OPT_MULTI_STAGE, RT_OPT_LANE, RT_OPT_CONTINUE, RT_OPT_HISTORY, RT_OPT_RGB24 = 0, 1, 2, 3, 4, 5, 6, 7


# every symbol include/raytrace_gpu.h declares: name -> (restype, argtypes)
_D3 = C.POINTER(C.c_double)
_MAT = C.POINTER(rt_material)
SYMBOLS = {
    "rt_init": (C.c_int, [C.c_int]),
    "rt_last_error": (C.c_char_p, []),
    "rt_version": (C.c_char_p, []),
    "rt_host_alloc": (C.c_void_p, [C.c_size_t]),
    "rt_host_free": (None, [C.c_void_p]),
    "rt_scene_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "rt_scene_destroy": (None, [C.c_void_p]),
    "rt_scene_set_dialect": (C.c_int, [C.c_void_p, C.c_int]),
    "rt_scene_set_option": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "rt_set_normal_callback": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _D3, NORMAL_FN, C.c_void_p]),
    "rt_scene_prepare_device": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "rt_scene_row_costs": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _D3]),
    "rt_split_rows": (C.c_int, [_D3, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "rt_split_rows_weighted": (C.c_int, [_D3, C.c_int, C.c_int, C.c_int, C.c_int, _D3, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "rt_render_multi": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, C.POINTER(C.c_int),
                                  C.c_void_p, C.c_void_p]),
    "rt_add_rectangle": (C.c_int, [C.c_void_p, _MAT, _D3, _D3, _D3]),
    "rt_add_triangle": (C.c_int, [C.c_void_p, _MAT, _D3, _D3, _D3]),
    "rt_add_sphere": (C.c_int, [C.c_void_p, _MAT, _D3, C.c_double]),
    "rt_add_cuboid": (C.c_int, [C.c_void_p, _MAT, C.c_int, _D3, C.c_double, C.c_double, C.c_double]),
    "rt_add_cylinder": (C.c_int, [C.c_void_p, _MAT, _MAT, _MAT, _D3, C.c_int, C.c_double, C.c_double]),
    "rt_begin_mconcat": (C.c_int, [C.c_void_p]),
    "rt_end_mconcat": (C.c_int, [C.c_void_p]),
    "rt_begin_mappend": (C.c_int, [C.c_void_p]),
    "rt_end_mappend": (C.c_int, [C.c_void_p]),
    "rt_set_normal_mode": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, _D3, C.c_int]),
    "rt_add_point_light": (C.c_int, [C.c_void_p, C.c_double, C.c_double, _D3]),
    "rt_add_ambient": (C.c_int, [C.c_void_p, C.c_double]),
    "rt_begin_shadow_world": (C.c_int, [C.c_void_p]),
    "rt_end_shadow_world": (C.c_int, [C.c_void_p]),
    "rt_add_point_light_world": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_double)]),
    "rt_scene_num_leaves": (C.c_int, [C.c_void_p]),
    "rt_scene_num_lights": (C.c_int, [C.c_void_p]),
    "rt_scene_get_leaf": (C.c_int, [C.c_void_p, C.c_int, _D3]),
    "rt_scene_prepare_host": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "rt_scene_get_filter_record": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "rt_scene_get_mfma_row": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_uint16)]),
    "rt_scene_get_block_cone": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "rt_scene_get_general_block": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
    "rt_scene_get_general_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_uint16)]),
    "rt_scene_get_guards": (C.c_int, [C.c_void_p, C.c_int, _D3, C.c_int]),
    "rt_camera_params": (C.c_int, [C.c_int, C.c_int, _D3]),
    "rt_render": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int,
                            C.c_void_p, C.c_void_p]),
    "rt_render_device": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int,
                                   C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rt_render_device_noalloc": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int,
                                           C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rt_synchronize": (C.c_int, []),
    "rt_get_stats": (C.c_int, [C.c_void_p, C.POINTER(rt_stats)]),
    "rt_write_bmp": (C.c_int, [C.c_char_p, C.c_void_p, C.c_int, C.c_int]),
}

VALU_LIB_PATH = os.path.join(_HERE, "lib", "libraytrace_gpu_valu.so")
_lib = None
_valu_lib = None


def _load(path):
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`"
                           " (there is no CPU fallback)")
    L = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    return L


def lib():
    """Load libraytrace_gpu.so (built by __graft_entry__.build() / raytrace_amd/csrc/Makefile)."""
    global _lib
    if _lib is None:
        _lib = _load(LIB_PATH)
    return _lib


def valu_lib():
    """The cross-check build (same sources, -DRT_VALU_FILTER): it also holds the FP32 vector-ALU kernel family, an
    independent implementation of the filters.  Tests and tools compare whole frames of the two; never the product."""
    global _valu_lib
    if _valu_lib is None:
        _valu_lib = _load(VALU_LIB_PATH)
    return _valu_lib


def check(rc, what="raytrace_gpu", L=None):
    if rc < 0:
        msg = (L or lib()).rt_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (rt_status {rc}): {msg}")
    return rc


def vec3(p):
    return (C.c_double * 3)(float(p[0]), float(p[1]), float(p[2]))
