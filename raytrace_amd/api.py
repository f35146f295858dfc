"""Host-side mirror of the reference's scene API (reference src/Main.hs), over the C ABI.

The reference builds `world :: Shape` and `lights :: Light` from closures; here the same
constructor names build plain data (a tree of shapes in the reference's mconcat/mappend order),
and `parallelTrace` flattens it into the leaf list of libraytrace_gpu.so and renders on the GPU.
Names, argument order and meaning follow the reference:

    rectangle material point width height            Main.hs:226
    triangle material pa pb pc                       Main.hs:363
    sphere material center radius                    Main.hs:392
    cube / colorcube / cuboid / colorcuboid          Main.hs:286-320
    cylinder topM botM mantleM point n h r           Main.hs:725
    pointLight diffuse specular lpos, ambient f      Main.hs:467, 490   (world is bound at render time)
    linearInterpolation f fnorm (x1,y1) (x2,y2) step origin scale        Main.hs:937
    fixedCamera w h, trace depth lights world, parallelTrace w h camera tracer      Main.hs:506, 541, 623
    saveBmpImage path image                          Main.hs:621

The smaller program src/Talk.hs (its own normalize, rectangle test, monoid and point light) is mirrored by
talk_rectangle / talk_cube / talk_pointLight / traceAndShade / parallel (Talk.hs:112, 197, 237, 255, 171); a world
wrapped in TalkDialect is rendered with Talk.hs's rules (rt_scene_set_dialect).
"""
import ctypes as C
import math
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence, Tuple

import weakref

import numpy as np

from . import _lib

Vec = Tuple[float, float, float]


# ---------------------------------------------------------------------------------------------
# Materials and colours — Main.hs:178-185, 326-355

@dataclass(frozen=True)
class Material:
    diffuse: Tuple[int, int, int]
    specular: Tuple[int, int, int]
    specularity: float = 1.0
    reflectivity: float = 0.0

    def replace(self, **kw):
        d = dict(diffuse=self.diffuse, specular=self.specular, specularity=self.specularity,
                 reflectivity=self.reflectivity)
        d.update(kw)
        return Material(**d)


def _colour(rgb):
    return Material(rgb, rgb, 1.0, 0.0)


black = _colour((0, 0, 0))
white = _colour((255, 255, 255))
red = _colour((255, 0, 0))
green = _colour((0, 255, 0))
blue = _colour((0, 0, 255))
magenta = _colour((255, 0, 255))
cyan = _colour((0, 255, 255))
yellow = _colour((255, 255, 0))
orange = _colour((255, 134, 0))
orchid = _colour((153, 50, 204))
aquamarine = _colour((69, 139, 116))
COLOURS = [black, white, red, green, blue, magenta, cyan, yellow, orange, orchid, aquamarine]


# ---------------------------------------------------------------------------------------------
# Shapes: a tree mirroring the reference's Monoid Shape (Main.hs:196-217)

class Shape:
    """Base class.  `a + b` is `a <> b` (mappend)."""

    def __add__(self, other):
        return MAppend(self, other)


@dataclass
class Empty(Shape):          # mempty, Main.hs:204
    pass


@dataclass
class Rectangle(Shape):
    material: Material
    point: Vec
    width: Vec
    height: Vec


@dataclass
class Triangle(Shape):
    material: Material
    pa: Vec
    pb: Vec
    pc: Vec


@dataclass
class Sphere(Shape):
    material: Material
    center: Vec
    radius: float


@dataclass
class Cuboid(Shape):
    materials: Sequence[Material]
    point: Vec
    l: float
    h: float
    d: float


@dataclass
class Cylinder(Shape):
    top: Material
    bot: Material
    mantle: Material
    point: Vec
    n: int
    h: float
    r: float


@dataclass
class MAppend(Shape):
    left: Shape
    right: Shape


@dataclass
class MConcat(Shape):
    children: List[Shape]


@dataclass
class GaussianNormals(Shape):
    """linearInterpolation's `normals` wrapper with the gaussian fnorm of main (Main.hs:956-965, 580-584)."""
    child: Shape
    scale_inv: float
    offset: Vec


@dataclass
class CallbackNormals(Shape):
    """linearInterpolation's `normals` wrapper with ANY fnorm (Main.hs:956-965): a Python callable
    fnorm(x, y) -> (nx, ny, nz), evaluated on the host for the hit points the GPU finds (RT_NORMAL_CALLBACK)."""
    child: Shape
    scale_inv: float
    offset: Vec
    fnorm: Callable[[float, float], Vec]


@dataclass
class TalkDialect(Shape):
    """Marks a world as a src/Talk.hs world: rendered with Talk.hs's rules (rectangles only)."""
    child: Shape


mempty = Empty()


def mappend(a: Shape, b: Shape) -> Shape:
    return MAppend(a, b)


def mconcat(xs: Sequence[Shape]) -> Shape:
    return MConcat(list(xs))


def rectangle(material, point, width, height) -> Shape:
    return Rectangle(material, tuple(point), tuple(width), tuple(height))


def triangle(material, pa, pb, pc) -> Shape:
    return Triangle(material, tuple(pa), tuple(pb), tuple(pc))


def sphere(material, center, radius) -> Shape:          # Main.hs:392
    return Sphere(material, tuple(center), radius)


def colorcuboid(materials, point, l, h, d) -> Shape:
    if len(materials) == 0:
        raise ValueError("colorcuboid: list of materials must not be empty.")   # Main.hs:302
    return Cuboid(list(materials), tuple(point), l, h, d)


def cuboid(material, point, l, h, d) -> Shape:
    return colorcuboid([material], point, l, h, d)


def colorcube(materials, point, l) -> Shape:
    return colorcuboid(materials, point, l, l, l)


def cube(material, point, l) -> Shape:
    return colorcube([material], point, l)


def cylinder(topM, botM, mantleM, point, n, h, r) -> Shape:
    return Cylinder(topM, botM, mantleM, tuple(point), int(n), h, r)


# ---- src/Talk.hs constructors -----------------------------------------------------------------

def talk_rectangle(colour, point, width, height) -> Shape:          # Talk.hs:112
    m = colour if isinstance(colour, Material) else _colour(tuple(colour))
    return Rectangle(m, tuple(point), tuple(width), tuple(height))


def talk_cube(colours, s, p) -> Shape:                               # Talk.hs:197-218 (`<>` is infixr: a <> (b <> ...))
    s2 = s / 2
    c = [colours[i % len(colours)] for i in range(6)]                # take 6 $ cycle colours
    px, py, pz = p
    faces = [
        talk_rectangle(c[0], (px, py + s2, pz), (s, 0, 0), (0, 0, -s)),      # top
        talk_rectangle(c[1], (px, py - s2, pz), (0, 0, s), (s, 0, 0)),       # bottom
        talk_rectangle(c[2], (px, py, pz + s2), (s, 0, 0), (0, s, 0)),       # front
        talk_rectangle(c[3], (px, py, pz - s2), (s, 0, 0), (0, -s, 0)),      # back
        talk_rectangle(c[4], (px + s2, py, pz), (0, s, 0), (0, 0, s)),       # left
        talk_rectangle(c[5], (px - s2, py, pz), (0, s, 0), (0, 0, -s)),      # right
    ]
    out = faces[-1]
    for f in reversed(faces[:-1]):
        out = MAppend(f, out)
    return out


def talk_pointLight(intensity, lp):                                  # Talk.hs:237 (the world is bound at render time)
    return [PointLight(intensity, 0.0, tuple(lp))]


# ---------------------------------------------------------------------------------------------
# Lights — Main.hs:423-492.  A Light value is a list of sources (`mconcat`); the world a point
# light casts its shadow rays against is the Shape it closed over (`pointLight world ...`, Main.hs:467): None = the world
# being traced, which is what every scene of the reference passes.

@dataclass
class PointLight:
    diffuse: float
    specular: float
    pos: Vec
    world: Optional["Shape"] = None


@dataclass
class Ambient:
    f: float


def pointLight(*args, world=None):
    """pointLight diffuse specular lpos, or — the reference's own argument order, Main.hs:467 — pointLight world diffuse
    specular lpos: the light then tests its shadow rays against `world` whatever is traced."""
    if len(args) == 4:
        world, args = args[0], args[1:]
    diffuse, specular, lpos = args
    return [PointLight(diffuse, specular, tuple(lpos), world)]


def ambient(f):
    return [Ambient(f)]


def mconcat_lights(ls):
    out = []
    for l in ls:
        out.extend(l)
    return out


# ---------------------------------------------------------------------------------------------
# linearInterpolation — Main.hs:937-988

def gaussian(x, y):                      # Main.hs:580
    return math.exp(-(x * x + y * y) * 4)


GAUSSIAN_FNORM = "gaussian"              # Main.hs:581-584 (evaluated on the GPU)


def _steps(a, b, step):                  # Main.hs:948
    out = []
    v = a
    while v < b:
        out.append(v)
        v = v + step
    return out


def linearInterpolation(f: Callable[[float, float], float], fnorm, p1, p2, step, origin, scale):
    """Returns (world, lights).  fnorm is None (geometric normals, `Nothing`), GAUSSIAN_FNORM (main's gaussian,
    evaluated on the GPU) or any callable (x, y) -> (nx, ny, nz) (`Just fnorm`, evaluated on the host)."""
    (x1, y1), (x2, y2) = p1, p2
    x = (x2 + x1) / 2
    y = (y2 + y1) / 2
    w = x2 - x1
    h = y2 - y1
    scale_inv = 1.0 / scale
    offset = tuple(o - scale * c for o, c in zip(origin, (x, 0, y)))           # :982
    mat = white.replace(specularity=1)

    def fp(a, b):                                                               # :981
        fa = f(a, b)
        return (offset[0] + scale * a, offset[1] + scale * fa, offset[2] + scale * b)

    cells = []
    for gx in _steps(x1, x2, step):
        for gy in _steps(y1, y2, step):
            xp, yp = gx + step, gy + step
            two = mconcat([triangle(mat, fp(gx, yp), fp(xp, yp), fp(gx, gy)),
                           triangle(mat, fp(gx, gy), fp(xp, yp), fp(xp, gy))])
            if fnorm is None:
                cells.append(two)
            elif fnorm == GAUSSIAN_FNORM:
                cells.append(GaussianNormals(two, scale_inv, offset))
            elif callable(fnorm):                                                  # any Double -> Double -> Vector3D
                cells.append(CallbackNormals(two, scale_inv, offset, fnorm))
            else:
                raise TypeError("fnorm must be None, GAUSSIAN_FNORM or a callable (x, y) -> (nx, ny, nz)")
    shape = mconcat([mconcat(cells),
                     rectangle(aquamarine, origin, (1.5 * scale * w, 0, 0), (0, 0, -(1.5 * scale * h)))])
    light = mconcat_lights([
        pointLight(0.3, 0.6, (origin[0] + scale * w / 4, origin[1] + scale * 10, origin[2] + scale * w / 4)),
        ambient(0.3)])
    return shape, light


# ---------------------------------------------------------------------------------------------
# Scene: the flat leaf list behind the C ABI

def _cmat(m: Material):
    cm = _lib.rt_material()
    for i in range(3):
        cm.diffuse[i] = m.diffuse[i]
        cm.specular[i] = m.specular[i]
    cm.specularity = float(m.specularity)
    cm.reflectivity = float(m.reflectivity)
    return cm


class Scene:
    """Owns an rt_scene.  Built from (world, lights) in the reference's depth-first leaf order."""

    def __init__(self, world: Shape, lights, lib=None):
        self.bytes_per_pixel = 4               # of the colour outputs (3 after set_option(RT_OPT_RGB24, 1))
        self._L = lib or _lib.lib()
        self._callbacks = {}                      # id(fnorm) -> ctypes callback object (kept alive; one thunk per distinct fnorm)
        self._callback_error = None               # first exception an fnorm raised during a render (re-raised after it)
        h = C.c_void_p()
        self._check(self._L.rt_scene_create(C.byref(h)), "rt_scene_create")
        self._h = h
        if isinstance(world, TalkDialect):        # src/Talk.hs rules; must be set before the first shape
            self._check(self._L.rt_scene_set_dialect(self._h, 1), "rt_scene_set_dialect")
            world = world.child
        self._add(world)
        own = {}                                  # id(Shape) -> shadow world: a light that closed over another Shape (Main.hs:467)
        for l in lights:
            if isinstance(l, PointLight) and l.world is not None and l.world is not world and id(l.world) not in own:
                own[id(l.world)] = self._check(self._L.rt_begin_shadow_world(self._h), "rt_begin_shadow_world")
                self._add(l.world)
                self._check(self._L.rt_end_shadow_world(self._h), "rt_end_shadow_world")
        for l in lights:
            if isinstance(l, PointLight) and l.world is not None and l.world is not world:
                self._check(self._L.rt_add_point_light_world(self._h, own[id(l.world)], l.diffuse, l.specular, _lib.vec3(l.pos)))
            elif isinstance(l, PointLight):
                self._check(self._L.rt_add_point_light(self._h, l.diffuse, l.specular, _lib.vec3(l.pos)))
            elif isinstance(l, Ambient):
                self._check(self._L.rt_add_ambient(self._h, l.f))
            else:
                raise TypeError(f"not a light: {l!r}")

    def _check(self, rc, what=""):
        return _lib.check(rc, what, self._L)

    def _raise_callback_error(self):
        err, self._callback_error = self._callback_error, None
        if err is not None:
            raise RuntimeError("an fnorm callback raised during the render; its normals were left (0, 0, 0)") from err

    def _add(self, s: Shape):
        L, h = self._L, self._h
        if isinstance(s, Empty):                 # mempty = mconcat []
            self._check(L.rt_begin_mconcat(h))
            self._check(L.rt_end_mconcat(h))
            return
        if isinstance(s, Rectangle):
            self._check(L.rt_add_rectangle(h, C.byref(_cmat(s.material)), _lib.vec3(s.point), _lib.vec3(s.width), _lib.vec3(s.height)))
        elif isinstance(s, Triangle):
            self._check(L.rt_add_triangle(h, C.byref(_cmat(s.material)), _lib.vec3(s.pa), _lib.vec3(s.pb), _lib.vec3(s.pc)))
        elif isinstance(s, Sphere):
            self._check(L.rt_add_sphere(h, C.byref(_cmat(s.material)), _lib.vec3(s.center), s.radius))
        elif isinstance(s, Cuboid):
            arr = (_lib.rt_material * len(s.materials))(*[_cmat(m) for m in s.materials])
            self._check(L.rt_add_cuboid(h, arr, len(s.materials), _lib.vec3(s.point), s.l, s.h, s.d))
        elif isinstance(s, Cylinder):
            self._check(L.rt_add_cylinder(h, C.byref(_cmat(s.top)), C.byref(_cmat(s.bot)), C.byref(_cmat(s.mantle)),
                                         _lib.vec3(s.point), s.n, s.h, s.r))
        elif isinstance(s, MAppend):
            self._check(L.rt_begin_mappend(h))
            self._add(s.left)
            self._add(s.right)
            self._check(L.rt_end_mappend(h))
        elif isinstance(s, MConcat):
            self._check(L.rt_begin_mconcat(h))
            for c in s.children:
                self._add(c)
            self._check(L.rt_end_mconcat(h))
        elif isinstance(s, GaussianNormals):
            first = L.rt_scene_num_leaves(h)
            self._add(s.child)
            n = L.rt_scene_num_leaves(h) - first
            params = (C.c_double * 4)(s.scale_inv, *s.offset)
            self._check(L.rt_set_normal_mode(h, first, n, 2, params, 4))
        elif isinstance(s, CallbackNormals):
            first = L.rt_scene_num_leaves(h)
            self._add(s.child)
            n = L.rt_scene_num_leaves(h) - first
            params = (C.c_double * 4)(s.scale_inv, *s.offset)
            fnorm = s.fnorm

            cb = self._callbacks.get(id(fnorm))
            if cb is None:
                def thunk(count, xy, normals, _user, fnorm=fnorm):
                    # ctypes would print an exception and carry on with (0, 0, 0) normals: keep it and re-raise after the render
                    try:
                        for i in range(count):
                            v = fnorm(xy[2 * i], xy[2 * i + 1])
                            normals[3 * i], normals[3 * i + 1], normals[3 * i + 2] = float(v[0]), float(v[1]), float(v[2])
                    except Exception as e:          # noqa: BLE001
                        if self._callback_error is None:
                            self._callback_error = e

                cb = _lib.NORMAL_FN(thunk)
                self._callbacks[id(fnorm)] = cb
                self._callback_fns = getattr(self, "_callback_fns", []) + [fnorm]      # keeps id(fnorm) unique while the scene lives
            self._check(L.rt_set_normal_callback(h, first, n, params, cb, None))
        else:
            raise TypeError(f"not a shape: {s!r}")

    @property
    def handle(self):
        return self._h

    @property
    def num_leaves(self):
        return self._L.rt_scene_num_leaves(self._h)

    def leaf(self, i):
        out = (C.c_double * 20)()
        self._check(self._L.rt_scene_get_leaf(self._h, i, out))
        return list(out)

    def render(self, w, h, depth=4, spp=1, seed=0, row0=0, nrows=None, want_rgba=True, want_hit=True,
               out_rgba=None, out_hit=None):
        """Host-buffer render (rt_render).  Returns (rgba[nrows,w,4] uint8, hit_id[nrows,w] int32); with
        set_option(RT_OPT_RGB24, 1) the colours are [nrows,w,3].

        out_rgba / out_hit: caller-owned arrays of exactly that shape and dtype (e.g. from host_buffer(),
        which makes the copy back DMA); by default fresh pageable numpy arrays are returned."""
        nrows = h - row0 if nrows is None else nrows
        bpp = self.bytes_per_pixel
        for out, shape, dt in ((out_rgba, (nrows, w, bpp), np.uint8), (out_hit, (nrows, w), np.int32)):
            if out is not None and (out.shape != shape or out.dtype != dt or not out.flags.c_contiguous):
                raise ValueError(f"render: output array must be C-contiguous {shape} {np.dtype(dt).name}")
        want_rgba, want_hit = want_rgba or out_rgba is not None, want_hit or out_hit is not None
        rgba = (out_rgba if out_rgba is not None else np.empty((nrows, w, bpp), np.uint8)) if want_rgba else None
        hit = (out_hit if out_hit is not None else np.empty((nrows, w), np.int32)) if want_hit else None
        self._check(self._L.rt_render(self._h, w, h, depth, spp, seed, row0, nrows,
                                     rgba.ctypes.data if want_rgba else None,
                                     hit.ctypes.data if want_hit else None), "rt_render")
        self._raise_callback_error()
        return rgba, hit

    def set_option(self, option, value):
        """rt_scene_set_option: _lib.RT_OPT_CULL / RT_OPT_FILTER / RT_OPT_HOST_BANDS / RT_OPT_MULTI_STAGE / RT_OPT_LANE /
        RT_OPT_CONTINUE."""
        self._check(self._L.rt_scene_set_option(self._h, option, value), "rt_scene_set_option")
        if option == _lib.RT_OPT_RGB24:
            self.bytes_per_pixel = 3 if value else 4
        return self

    def prepare(self, w, h):
        """rt_scene_prepare_device: build the records and upload the scene now (keeps it out of a timed region)."""
        self._check(self._L.rt_scene_prepare_device(self._h, w, h), "rt_scene_prepare_device")

    def row_costs(self, w, h, ncost=32, depth=4, spp=1):
        """rt_scene_row_costs: measured relative cost of ncost equal strips of the frame."""
        out = (C.c_double * ncost)()
        self._check(self._L.rt_scene_row_costs(self._h, w, h, depth, spp, ncost, out), "rt_scene_row_costs")
        return [float(v) for v in out]

    def render_multi(self, w, h, devices, depth=4, spp=1, seed=0, want_rgba=True, want_hit=True):
        """rt_render_multi: the whole image on several devices of this process (bands of equal measured cost,
        gathered on devices[0] over xGMI, then copied to host memory)."""
        devs = (C.c_int * len(devices))(*devices)
        rgba = np.empty((h, w, self.bytes_per_pixel), np.uint8) if want_rgba else None
        hit = np.empty((h, w), np.int32) if want_hit else None
        self._check(self._L.rt_render_multi(self._h, w, h, depth, spp, seed, len(devices), devs,
                                           rgba.ctypes.data if want_rgba else None,
                                           hit.ctypes.data if want_hit else None), "rt_render_multi")
        self._raise_callback_error()
        return rgba, hit

    def render_device(self, w, h, d_rgba, d_hit, depth=4, spp=1, seed=0, row0=0, nrows=None, stream=None, lane=None, cont=False):
        """Device-buffer render (rt_render_device); d_rgba/d_hit are device addresses (int) or None.  lane = 0 / 1: the
        call uses that lane of the scene (RT_OPT_LANE; cont: it adds to the frame in progress, RT_OPT_CONTINUE) and the
        options are left that way; None: whatever the options say (default: the library splits big bands itself)."""
        nrows = h - row0 if nrows is None else nrows
        if lane is not None:
            self.set_option(_lib.RT_OPT_LANE, lane).set_option(_lib.RT_OPT_CONTINUE, int(bool(cont)))
        self._check(self._L.rt_render_device(self._h, w, h, depth, spp, seed, row0, nrows, d_rgba, d_hit, stream), "rt_render_device")

    def render_device_noalloc(self, w, h, d_rgba, d_hit, depth=4, spp=1, seed=0, row0=0, nrows=None, stream=None):
        """rt_render_device_noalloc: as render_device, but allocates nothing (static device scratch); the scene must
        have been made resident with prepare(w, h)."""
        nrows = h - row0 if nrows is None else nrows
        self._check(self._L.rt_render_device_noalloc(self._h, w, h, depth, spp, seed, row0, nrows, d_rgba, d_hit, stream), "rt_render_device_noalloc")

    def stats(self):
        st = _lib.rt_stats()
        self._check(self._L.rt_get_stats(self._h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in st._fields_}

    def close(self):
        if self._h:
            self._L.rt_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def host_buffer(shape, dtype):
    """A numpy array in page-locked host memory (rt_host_alloc), freed with the array: pass it to
    Scene.render(out_rgba= / out_hit=) and the copy back from the GPU is DMA hidden behind the render."""
    dtype = np.dtype(dtype)
    nbytes = int(np.prod(shape)) * dtype.itemsize
    L = _lib.lib()
    p = L.rt_host_alloc(max(nbytes, 1))
    if not p:
        raise RuntimeError("rt_host_alloc: " + (L.rt_last_error() or b"").decode())
    raw = (C.c_uint8 * max(nbytes, 1)).from_address(p)
    arr = np.frombuffer(raw, dtype=np.uint8, count=nbytes).view(dtype).reshape(shape)
    weakref.finalize(raw, L.rt_host_free, p)       # numpy keeps `raw` alive as the array's base
    return arr


# ---------------------------------------------------------------------------------------------
# Camera / trace / parallelTrace / saveBmpImage

@dataclass
class Camera:
    width: int
    height: int


def fixedCamera(width, height):                 # Main.hs:506
    return Camera(width, height)


@dataclass
class Tracer:
    """`trace depth lights world` partially applied (Main.hs:541, 571)."""
    depth: int
    lights: list
    world: Shape
    _scene: Optional[Scene] = field(default=None, repr=False)

    def scene(self):
        if self._scene is None:
            self._scene = Scene(self.world, self.lights)
        return self._scene


def trace(depth, lights, world):
    return Tracer(depth, lights, world)


def parallelTrace(w, h, camera: Camera, tracer: Tracer):
    """Main.hs:623-637: the whole h x w image at once, on the GPU.  Returns uint8 [h, w, 3]."""
    if (camera.width, camera.height) != (w, h):
        raise ValueError("parallelTrace: camera and image size differ")
    rgba, _ = tracer.scene().render(w, h, depth=tracer.depth, want_hit=False)
    return rgba[:, :, :3].copy()


def traceAndShade(world, camera: Camera, lights):        # Talk.hs:255-258
    return Tracer(1, lights, world if isinstance(world, TalkDialect) else TalkDialect(world))


def parallel(w, h, tracer: Tracer):             # Talk.hs:171-182
    rgba, _ = tracer.scene().render(w, h, depth=1, want_hit=False)
    return rgba[:, :, :3].copy()


def saveBmpImage(path, image):                  # Main.hs:621
    img = np.ascontiguousarray(image, dtype=np.uint8)
    h, w = img.shape[:2]
    if img.shape[2] == 3:
        img = np.concatenate([img, np.full((h, w, 1), 255, np.uint8)], axis=2)
    _lib.check(_lib.lib().rt_write_bmp(str(path).encode(), img.ctypes.data, w, h), "rt_write_bmp")
