"""Scenes: the reference's demo scenes re-expressed through the mirrored API (scene DATA only),
and the synthetic benchmark scenes of BASELINE.json / SURVEY.md §8(d).

Every function returns (world, lights).  Scenes for which the reference's `main` gives no lights use
the commented-out light set of Main.hs:607-609.
"""
import math

import numpy as np

from .api import (GAUSSIAN_FNORM, COLOURS, ambient, aquamarine, black, blue, colorcube, colorcuboid, cube, cyan,
                  cylinder, gaussian, green, linearInterpolation, magenta, mconcat, mconcat_lights, orange, orchid,
                  pointLight, rectangle, red, sphere, triangle, white, yellow)


def main_lights():                                   # Main.hs:607-609
    return mconcat_lights([pointLight(0.03, 0.2, (2, 0, 0)), pointLight(0.3, 1.0, (0, 4, -10)), ambient(0.2)])


def default_scene():                                 # Main.hs:578-585: gaussian height field
    return linearInterpolation(gaussian, GAUSSIAN_FNORM, (-1.0, -1.0), (1.0, 1.0), 0.2, (0.01, -7, -25), 10)


def planes():                                        # Main.hs:639-642
    return mconcat([
        rectangle(red, (-0.5, -0.5, -2), (1, 0, 0), (0, 1, 0)),
        rectangle(blue, (-1, -0.5, -1.5), (0, 1, 0), (0, 0, 1)),
        rectangle(green, (-0.5, -1, -1.5), (1, 0, 0), (0, 0, -1))]), main_lights()


def axes_shape():                                    # Main.hs:644-658
    c = (-0.5, -0.5, -1)

    def at(dx, dy, dz):
        return (c[0] + dx, c[1] + dy, c[2] + dz)
    return mconcat([
        rectangle(red, c, (1, 0, 0), (0, 0.1, 0)),
        rectangle(blue, c, (0, 1, 0), (0, 0, 0.1)),
        rectangle(green, c, (0.1, 0, 0), (0, 0, -1)),
        triangle(red, at(0.6, 0, 0), at(0.5, 0.05, 0), at(0.5, -0.05, 0)),
        triangle(blue, at(0, 0.6, 0), at(0, 0.5, 0.05), at(0, 0.5, -0.05)),
        triangle(green, at(0, 0, 0.6), at(0.05, 0, 0.5), at(-0.05, 0, 0.5))])


def axes():
    return axes_shape(), main_lights()


def cubes():                                         # Main.hs:661-668
    colors = [red, green, blue, magenta, cyan, aquamarine, yellow, orange, orchid]
    return mconcat([colorcube(colors, p, 1) for p in
                    [(-2, 0, -6), (0, -2, -6), (2, 0, -6), (0, 2, -6), (0, 0, -6)]]), main_lights()


def stacked_cubes():                                 # Main.hs:692-709
    colors = [red, green, yellow]
    return mconcat([
        rectangle(blue, (0, -2, 0), (20, 0, 0), (0, 0, -40)),
        rectangle(blue, (0, 4.5, 0), (20, 0, 0), (0, 0, 40)),
        colorcube(colors, (-2, -1.5, -6), 1),
        cube(white, (0, 4.2, -10), 0.1),
        cube(white, (0, 3.8, -10), 0.1),
        rectangle(green, (-0.2, 4, -10), (0, -0.5, 0), (0, 0, -0.3)),
        rectangle(green, (0.2, 4, -10), (0, -0.5, 0), (0, 0, 0.3)),
        rectangle(green, (0, 4, -9.8), (0, 0.5, 0), (0.3, 0, 0)),
        rectangle(green, (0, 4, -10.2), (0, -0.5, 0), (0.3, 0, 0)),
        rectangle(orange, (2.2, -0.5, -10), (0, 12, 0), (-5, 0, -10))]), main_lights()


def triangle_example():                              # Main.hs:711-723
    return mconcat([
        rectangle(cyan, (0, 0, -10), (4.0, 0, 0), (0, 4.0, 0)),
        rectangle(cyan, (-3, 0, -9), (2.0, 0, -2.0), (0, 4.0, 0)),
        rectangle(cyan, (3, 0, -9), (2.0, 0, 2.0), (0, 4.0, 0)),
        cube(yellow, (0, -1.5, -5), 1.0),
        triangle(orange, (0, 1, -4), (-1, 0, -4), (1, 0, -4)),
        rectangle(green, (0, 0, -3), (1, 0, 0), (0, 1, 0)),
        rectangle(blue, (0, -2, 0), (20, 0, 0), (0, 0, -40))]), main_lights()


def spec_test():                                     # Main.hs:776-788
    spec_white = white.replace(specularity=400)
    world = mconcat([
        rectangle(blue, (0, -2, 0), (20, 0, 0), (0, 0, -40)),
        rectangle(blue, (0, 10, 0), (20, 0, 0), (0, 0, -40)),
        rectangle(white, (-2, 0, -4), (0, 6, 0), (0, 0, 6)),
        rectangle(spec_white, (2, 0, -4), (-0.5, 0, 6), (0, 6, 0))])
    return world, pointLight(0.3, 0.6, (0, 0, -4)) + pointLight(0.0, 1.0, (-3, 0, -10))


def intersection():                                  # Main.hs:790-797
    world = mconcat([
        cylinder(red, red, red, (0, -1, -3), 20, 2, 0.02),
        rectangle(orange, (0, -1, -3), (1, 0, 1), (2, 0, -2))])
    return world, ambient(0.5) + pointLight(0.5, 0.2, (1, 1, -3))


def cylinder_axes():                                 # Main.hs:572 (commented world)
    return cylinder(blue, blue, green, (0, -2, -10), 20, 1, 5) + axes_shape(), main_lights()


def talk_main2():                                    # Talk.hs:263-273 geometry, lit with Main.hs's Phong lights
    world = colorcube([red, green, blue, white, black], (1, -2, -6), 2.0) + \
        rectangle(blue, (0, -5, 0), (20, 0, 0), (0, 0, -50))
    return world, mconcat_lights([ambient(0.1), pointLight(0.3, 0.0, (0, 0, 0)), pointLight(0.6, 0.0, (-1, 3, -3))])


def talkhs_main2():                                  # Talk.hs:261-273 with Talk.hs's own rules (TalkDialect)
    from .api import TalkDialect, talk_cube, talk_pointLight, talk_rectangle
    world = talk_cube([red, green, blue, white, black], 2.0, (1, -2, -6)) + \
        talk_rectangle(blue, (0, -5, 0), (20, 0, 0), (0, 0, -50))
    lights = mconcat_lights([ambient(0.1), talk_pointLight(0.3, (0, 0, 0)), talk_pointLight(0.6, (-1, 3, -3))])
    return TalkDialect(world), lights


def spheres():                                       # Main.hs:670-690
    mirror = black.replace(reflectivity=0.9, specular=(255, 255, 255), specularity=100)
    world = mconcat([
        sphere(mirror, (-2, 1, -4), 1),
        sphere(red, (0, 1, -7), 1),
        sphere(blue, (2, 1, -5), 1),
        rectangle(orange, (0, -5, -10), (0, 0, 20), (20, 0, 0)),
        rectangle(green, (0, 5, -15), (20, 0, 0), (0, 20, 0)),
        rectangle(mirror, (0, 4, -10), (20, 0, 0), (0, 0, 20)),
        rectangle(white, (10, 4, -10), (0, 0, 20), (0, 1, 0))])
    lights = mconcat_lights([pointLight(0.8, 0.1, (0, 3, -10)), pointLight(0.8, 0.8, (0, 3, 0)), ambient(0.2)])
    return world, lights


def mirror_room():
    """The planar part of `spheres` (Main.hs:670-690): mirror floor strip and walls, lights of that scene.
    Exercises trace's reflection recursion, Main.hs:546-556, without the spheres."""
    mirror = black.replace(reflectivity=0.9, specular=(255, 255, 255), specularity=100)
    world = mconcat([
        colorcube([red, green, blue], (-2, -3.5, -7), 2.0),
        triangle(yellow.replace(reflectivity=0.4), (2, -4.5, -6), (4, -4.5, -8), (3, -1.5, -7)),
        rectangle(orange, (0, -5, -10), (0, 0, 20), (20, 0, 0)),
        rectangle(green, (0, 5, -15), (20, 0, 0), (0, 20, 0)),
        rectangle(mirror, (0, 4, -10), (20, 0, 0), (0, 0, 20)),
        rectangle(white, (10, 4, -10), (0, 0, 20), (0, 1, 0)),
        rectangle(mirror, (-9, 0, -10), (0, 0, 20), (0, 10, 0))])
    lights = mconcat_lights([pointLight(0.8, 0.1, (0, 3, -10)), pointLight(0.8, 0.8, (0, 3, 0)), ambient(0.2)])
    return world, lights


def _normalize(v):
    n = math.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])
    return (v[0] / n, v[1] / n, v[2] / n)


def _cross(a, b):                                    # Main.hs:68-69
    return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - b[0] * a[1])


def sine_cosine(with_normals=False):                 # Main.hs:598-605 (commented world): 3D sine-cosine waveform
    """`linearInterpolation f Nothing (-2*pi,-pi/2) (2*pi,pi/2) (pi/10) origin 7`; with_normals=True passes the fnorm the
    reference defines beside it (`Just fnorm`) instead of Nothing: an arbitrary closure, evaluated on the host."""
    def f(x, y):
        return math.sin(x) * math.sin(y) + 2

    def fnorm(x, y):
        dfdx = math.cos(x) * math.sin(y)
        dfdy = math.sin(x) * math.cos(y)
        return _normalize(_cross((1.0, dfdx, 0.0), (0.0, dfdy, 1.0)))
    return linearInterpolation(f, fnorm if with_normals else None, (-2 * math.pi, -math.pi / 2), (2 * math.pi, math.pi / 2),
                               math.pi / 10, (0.0, -30.0, -70.0), 7.0)


def sinc(with_normals=True):                         # Main.hs:586-597 (commented world): 3D sinc function
    def f(x, y):
        if x == 0 and y == 0:
            return 4.0
        if x == 0:
            return 1 + math.sin(y) / y + 2
        if y == 0:
            return 1 + math.sin(x) / x + 2
        return math.sin(x) / x + math.sin(y) / y + 2

    def fnorm(x, y):
        if x == 0 and y == 0:
            return (0.0, 1.0, 0.0)
        # (the reference's fnorm uses dfdy for both tangent vectors, Main.hs:593, and divides by y*y: a NaN normal on
        # the line y = 0, reproduced as written)
        dfdy = (y * math.cos(y) - math.sin(y)) / (y * y) if y != 0 else float("nan")
        v = _cross((1.0, dfdy, 0.0), (0.0, dfdy, 1.0))
        n = math.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])
        return (v[0] / n, v[1] / n, v[2] / n)
    return linearInterpolation(f, fnorm if with_normals else None, (-4 * math.pi, -2 * math.pi), (4 * math.pi, 2 * math.pi),
                               math.pi / 4, (0.01, -10.0, -20.0), 1.0)


def viewport_plane():
    """A rectangle IN the viewport plane z = 0, where every camera ray starts (Main.hs:528-531): it is hit at
    t = (0 - 0) * recip = -0.0, which passes `t >= 0` (Main.hs:257) and must win against everything behind it
    (`-0.0 <= t2`, Main.hs:213), also when it is the right operand of the monoid."""
    return mconcat([
        rectangle(red, (0, 0, -3), (4, 0, 0), (0, 4, 0)),
        rectangle(green, (0.1, 0.02, 0), (0.3, 0, 0), (0, 0.1, 0)),
        rectangle(blue, (-0.2, -0.05, 0), (0, 0.1, 0), (0.2, 0, 0)),
        cube(yellow, (0, 0, -2), 0.5)]), main_lights()


REFERENCE_SCENES = {
    "default": default_scene, "planes": planes, "axes": axes, "cubes": cubes, "stacked_cubes": stacked_cubes,
    "triangle_example": triangle_example, "spec_test": spec_test, "intersection": intersection,
    "cylinder_axes": cylinder_axes, "talk_main2": talk_main2, "talkhs_main2": talkhs_main2, "mirror_room": mirror_room, "spheres": spheres,
    "sine_cosine": sine_cosine, "viewport_plane": viewport_plane,
}


# ---------------------------------------------------------------------------------------------
# Synthetic benchmark scenes (SURVEY.md §8 d)

SYNTH_SEED = 20260355


def _unit(rng):
    v = rng.normal(size=3)
    return v / np.linalg.norm(v)


def _centre(rng):
    return np.array([rng.uniform(-8, 8), rng.uniform(-6, 6), rng.uniform(-30, -6)])


def synthetic(ntri=10000, ncuboid=1000, seed=SYNTH_SEED, specularities=(1, 20, 100)):
    """configs[2]/[3]: ntri triangles (edge U[0.05,0.3]) + ncuboid axis-aligned cuboids (sides U[0.1,0.6]),
    centres U[-8,8]x[-6,6]x[-30,-6], 4 point lights (kd 0.25, ks 0.3) above the box, ambient 0.1.
    Leaf order: triangles, then cuboids (6 rectangles each)."""
    rng = np.random.default_rng(seed)
    shapes = []

    def mat():
        return COLOURS[int(rng.integers(len(COLOURS)))].replace(specularity=float(specularities[int(rng.integers(len(specularities)))]))

    for _ in range(ntri):
        c = _centre(rng)
        a, b = _unit(rng), _unit(rng)
        while abs(float(a @ b)) > 0.95:
            b = _unit(rng)
        pa, pb, pc = c, c + rng.uniform(0.05, 0.3) * a, c + rng.uniform(0.05, 0.3) * b
        shapes.append(triangle(mat(), tuple(map(float, pa)), tuple(map(float, pb)), tuple(map(float, pc))))
    for _ in range(ncuboid):
        c = _centre(rng)
        l, h, d = (float(rng.uniform(0.1, 0.6)) for _ in range(3))
        shapes.append(colorcuboid([mat()], tuple(map(float, c)), l, h, d))
    lights = mconcat_lights([pointLight(0.25, 0.3, p) for p in
                             [(-6.0, 10.0, -10.0), (6.0, 10.0, -10.0), (-6.0, 10.0, -26.0), (6.0, 10.0, -26.0)]]
                            + [ambient(0.1)])
    return mconcat(shapes), lights


def mirror_bench(ntri=4000, ncuboid=400):
    """`bench.py --workload mirror`: the mirrors of the reference's `spheres` scene (Main.hs:670-690: the reflective
    sphere, the mirror strip under the ceiling) and a mirror wall, around the synthetic triangles and cuboids, lit by
    that scene's two point lights.  Reflected rays (trace depth 4, Main.hs:546-556) see thousands of leaves."""
    mirror = black.replace(reflectivity=0.9, specular=(255, 255, 255), specularity=100)
    clutter, _ = synthetic(ntri=ntri, ncuboid=ncuboid)
    world = mconcat([
        clutter,
        sphere(mirror, (-2, 1, -4), 1),
        rectangle(orange, (0, -7, -18), (0, 0, 40), (40, 0, 0)),
        rectangle(green, (0, 0, -32), (40, 0, 0), (0, 40, 0)),
        rectangle(mirror, (0, 6.5, -18), (40, 0, 0), (0, 0, 40)),
        rectangle(mirror, (-9, 0, -18), (0, 20, 0), (0, 0, 40)),                          # normal +x: faces the room
        rectangle(mirror.replace(reflectivity=0.5), (9, 0, -18), (0, 0, 40), (0, 20, 0))])   # normal -x
    lights = mconcat_lights([pointLight(0.8, 0.1, (0, 3, -10)), pointLight(0.8, 0.8, (0, 3, 0)), ambient(0.2)])
    return world, lights


def synthetic_cylinders(ncyl=200, ncuboid=500, seed=SYNTH_SEED + 1, specularities=(100, 400)):
    """configs[4]: upright cylinders (n=16) + cuboids, specular-heavy materials."""
    rng = np.random.default_rng(seed)
    shapes = []

    def mat():
        return COLOURS[int(rng.integers(len(COLOURS)))].replace(specularity=float(specularities[int(rng.integers(len(specularities)))]))

    for _ in range(ncyl):
        c = _centre(rng)
        shapes.append(cylinder(mat(), mat(), mat(), tuple(map(float, c)), 16, float(rng.uniform(0.5, 2.0)), float(rng.uniform(0.1, 0.5))))
    for _ in range(ncuboid):
        c = _centre(rng)
        l, h, d = (float(rng.uniform(0.1, 0.6)) for _ in range(3))
        shapes.append(colorcuboid([mat()], tuple(map(float, c)), l, h, d))
    lights = mconcat_lights([pointLight(0.25, 0.3, p) for p in
                             [(-6.0, 10.0, -10.0), (6.0, 10.0, -10.0), (-6.0, 10.0, -26.0), (6.0, 10.0, -26.0)]]
                            + [ambient(0.1)])
    return mconcat(shapes), lights


def adversarial(seed=7):
    """Not a reference scene: leaves and lights placed where a conservative filter is most likely to be wrong —
    a huge floor under the eye, a parallelogram whose edge vectors are not orthogonal (Main.hs:238-241 accepts the
    dual parallelogram), slivers, tiny far triangles, leaves that cross the viewport plane z = 0 and the eye plane
    z = 0.5, a zero-area triangle and a zero-width rectangle (NaN normals: never hit), lights just above a leaf,
    in the plane of a leaf, at a vertex of a leaf and inside a cuboid, a mirror and a sphere.  GPU parity tests
    compare it with the oracle and the two first-stage filters with each other."""
    rng = np.random.default_rng(seed)
    mirror = black.replace(reflectivity=0.7, specular=(255, 255, 255), specularity=50)
    shapes = [
        rectangle(orange, (0, -4, -500), (1000, 0, 0), (0, 0, -1000)),                  # huge floor, passes under the eye
        rectangle(green, (-3, 0, -12), (4, 0, 1), (1.5, 3, 0.5)),                       # non-orthogonal edges
        rectangle(cyan.replace(specularity=40), (3, 1, -9), (0.02, 0, 0), (0, 4, 0.3)), # sliver rectangle
        triangle(red, (-6, -3, -20), (6, -2.999, -20), (0, -3.0005, -19.99)),           # sliver triangle
        triangle(yellow, (-1, -1, 0.3), (1, -1, -3), (0, 2, -1)),                       # crosses z = 0 (the ray origins)
        triangle(magenta, (-2, 0.5, 2), (2, 0.5, 2), (0, 1.5, -6)),                     # crosses the eye plane z = 0.5
        triangle(white, (1, 1, -5), (1, 1, -5), (2, 2, -5)),                            # zero area: NaN normal
        rectangle(white, (0, 0, -6), (0, 0, 0), (0, 1, 0)),                             # zero width: NaN normal
        rectangle(mirror, (5, 0, -14), (0, 0, 6), (0, 6, 0)),                           # a mirror facing -x ... +x
        sphere(blue.replace(specularity=80), (-4, -2.5, -11), 1.5),
        colorcuboid([red, green, blue], (0, -2, -10), 2.0, 2.0, 2.0),                   # holds a light
        colorcuboid([aquamarine], (2.5, -3.5, -6), 1.0, 1.0, 1.0),
        triangle(orchid, (-5, 3, -15), (-4, 3, -15), (-4.5, 4, -15)),                   # a light sits at its first vertex
    ]
    for _ in range(40):                                                                 # tiny far triangles
        c = np.array([rng.uniform(-20, 20), rng.uniform(-3, 12), rng.uniform(-90, -40)])
        a, b = _unit(rng) * 0.02, _unit(rng) * 0.02
        shapes.append(triangle(COLOURS[int(rng.integers(1, len(COLOURS)))], tuple(map(float, c)), tuple(map(float, c + a)), tuple(map(float, c + b))))
    lights = mconcat_lights([
        pointLight(0.4, 0.3, (0.0, -3.999, -8.0)),         # 1e-3 above the floor
        pointLight(0.3, 0.2, (3.0, -4.0, -5.0)),           # in the plane of the floor
        pointLight(0.3, 0.2, (0.0, -2.0, -10.0)),          # inside the cuboid
        pointLight(0.3, 0.4, (-5.0, 3.0, -15.0)),          # at a vertex of the orchid triangle
        pointLight(0.5, 0.5, (4.0, 9.0, 1.0)),             # behind the viewport plane
        ambient(0.15)])
    return mconcat(shapes), lights
