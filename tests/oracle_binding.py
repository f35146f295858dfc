"""ctypes binding of oracle/liboracle.so and a builder that maps a raytrace_amd.api scene tree onto
the oracle's closure-tree constructors.  TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg import this."""
import ctypes as C
import os

import numpy as np

from raytrace_amd import api

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_ROOT, "oracle", "liboracle.so")

FLAG_TIE, FLAG_BBOX, FLAG_EDGE, FLAG_BBOX_EDGE = 1, 2, 4, 8


class orc_material(C.Structure):
    _fields_ = [("diffuse", C.c_uint8 * 3), ("specular", C.c_uint8 * 3), ("_pad", C.c_uint8 * 2),
                ("specularity", C.c_double), ("reflectivity", C.c_double)]


_D3 = C.POINTER(C.c_double)
ORC_FNORM = C.CFUNCTYPE(None, C.c_double, C.c_double, _D3, C.c_void_p)
_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(LIB_PATH)
        L.orc_scene_new.restype = C.c_void_p
        L.orc_scene_free.argtypes = [C.c_void_p]
        L.orc_scene_set_talk.argtypes = [C.c_void_p, C.c_int]
        L.orc_scene_set_literal_mconcat.argtypes = [C.c_void_p, C.c_int]
        L.orc_mempty.argtypes = [C.c_void_p]
        L.orc_mappend.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_mconcat.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int]
        L.orc_rectangle.argtypes = [C.c_void_p, C.POINTER(orc_material), _D3, _D3, _D3]
        L.orc_triangle.argtypes = [C.c_void_p, C.POINTER(orc_material), _D3, _D3, _D3]
        L.orc_sphere.argtypes = [C.c_void_p, C.POINTER(orc_material), _D3, C.c_double]
        L.orc_colorcuboid.argtypes = [C.c_void_p, C.POINTER(orc_material), C.c_int, _D3, C.c_double, C.c_double, C.c_double]
        L.orc_cylinder.argtypes = [C.c_void_p, C.POINTER(orc_material), C.POINTER(orc_material), C.POINTER(orc_material),
                                   _D3, C.c_int, C.c_double, C.c_double]
        L.orc_normals_gaussian.argtypes = [C.c_void_p, C.c_int, C.c_double, _D3]
        L.orc_normals_callback.argtypes = [C.c_void_p, C.c_int, C.c_double, _D3, ORC_FNORM, C.c_void_p]
        L.orc_linear_interpolation_gaussian.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double,
                                                        C.c_double, _D3, C.c_double]
        L.orc_point_light.argtypes = [C.c_void_p, C.c_double, C.c_double, _D3]
        L.orc_point_light_world.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, _D3]
        L.orc_ambient.argtypes = [C.c_void_p, C.c_double]
        L.orc_num_leaves.argtypes = [C.c_void_p, C.c_int]
        L.orc_render.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int,
                                 C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_eps.restype = C.c_double
        L.orc_camera_params.argtypes = [C.c_int, C.c_int, _D3]
        L.orc_camera_ray.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _D3]
        L.orc_grid_steps.argtypes = [C.c_double, C.c_double, C.c_double, _D3, C.c_int]
        _lib = L
    return _lib


def _v(p):
    return (C.c_double * 3)(float(p[0]), float(p[1]), float(p[2]))


def _m(m):
    cm = orc_material()
    for i in range(3):
        cm.diffuse[i] = m.diffuse[i]
        cm.specular[i] = m.specular[i]
    cm.specularity = float(m.specularity)
    cm.reflectivity = float(m.reflectivity)
    return cm


class OracleScene:
    def __init__(self, world, lights, literal_mconcat=False):
        self.L = lib()
        self.h = C.c_void_p(self.L.orc_scene_new())
        self._callbacks = []
        if literal_mconcat:      # every mconcat as the spelled-out chain of mappend nodes (Main.hs:196-217)
            assert self.L.orc_scene_set_literal_mconcat(self.h, 1) == 0
        if isinstance(world, api.TalkDialect):        # src/Talk.hs rules
            assert self.L.orc_scene_set_talk(self.h, 1) == 0
            world = world.child
        self.root = self._node(world)
        own = {}                                       # id(Shape) -> node: a light's own world (pointLight world ..., Main.hs:467)
        for l in lights:
            if isinstance(l, api.PointLight) and l.world is not None and l.world is not world:
                if id(l.world) not in own:
                    own[id(l.world)] = self._node(l.world)
                self.L.orc_point_light_world(self.h, own[id(l.world)], l.diffuse, l.specular, _v(l.pos))
            elif isinstance(l, api.PointLight):
                self.L.orc_point_light(self.h, l.diffuse, l.specular, _v(l.pos))
            else:
                self.L.orc_ambient(self.h, l.f)

    def _node(self, s):
        L, h = self.L, self.h
        if isinstance(s, api.Empty):
            return L.orc_mempty(h)
        if isinstance(s, api.Rectangle):
            return L.orc_rectangle(h, C.byref(_m(s.material)), _v(s.point), _v(s.width), _v(s.height))
        if isinstance(s, api.Triangle):
            return L.orc_triangle(h, C.byref(_m(s.material)), _v(s.pa), _v(s.pb), _v(s.pc))
        if isinstance(s, api.Sphere):
            return L.orc_sphere(h, C.byref(_m(s.material)), _v(s.center), s.radius)
        if isinstance(s, api.Cuboid):
            arr = (orc_material * len(s.materials))(*[_m(m) for m in s.materials])
            return L.orc_colorcuboid(h, arr, len(s.materials), _v(s.point), s.l, s.h, s.d)
        if isinstance(s, api.Cylinder):
            return L.orc_cylinder(h, C.byref(_m(s.top)), C.byref(_m(s.bot)), C.byref(_m(s.mantle)), _v(s.point), s.n, s.h, s.r)
        if isinstance(s, api.MAppend):
            return L.orc_mappend(h, self._node(s.left), self._node(s.right))
        if isinstance(s, api.MConcat):
            kids = [self._node(c) for c in s.children]
            return L.orc_mconcat(h, (C.c_int * max(1, len(kids)))(*kids), len(kids))
        if isinstance(s, api.GaussianNormals):
            return L.orc_normals_gaussian(h, self._node(s.child), s.scale_inv, _v(s.offset))
        if isinstance(s, api.CallbackNormals):
            fnorm = s.fnorm

            def thunk(x, y, out, _user, fnorm=fnorm):
                v = fnorm(x, y)
                out[0], out[1], out[2] = float(v[0]), float(v[1]), float(v[2])

            cb = ORC_FNORM(thunk)
            self._callbacks.append(cb)
            return L.orc_normals_callback(h, self._node(s.child), s.scale_inv, _v(s.offset), cb, None)
        raise TypeError(s)

    @property
    def num_leaves(self):
        return self.L.orc_num_leaves(self.h, self.root)

    def render(self, w, h, depth=4, spp=1, seed=0, row0=0, nrows=None, nthreads=8, use_bbox=1, flags=False, want_t=False):
        nrows = h - row0 if nrows is None else nrows
        rgb = np.zeros((nrows, w, 3), np.uint8)
        hit = np.zeros((nrows, w), np.int32)
        t = np.zeros((nrows, w), np.float64) if want_t else None
        fl = np.zeros((nrows, w), np.uint8) if flags else None
        rc = self.L.orc_render(self.h, self.root, w, h, depth, spp, seed, row0, nrows, nthreads, use_bbox,
                               rgb.ctypes.data, hit.ctypes.data, t.ctypes.data if want_t else None,
                               fl.ctypes.data if flags else None)
        if rc != 0:
            raise RuntimeError(f"orc_render failed: {rc}")
        return {"rgb": rgb, "hit_id": hit, "t": t, "flags": fl}

    def close(self):
        if self.h:
            self.L.orc_scene_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
