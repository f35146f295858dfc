#!/usr/bin/env python3
"""Differential fuzzing of the GPU path against the CPU oracle: random small scenes (triangles, rectangles with arbitrary edge
vectors, cuboids, cylinders, spheres, mirrors; lights anywhere, also inside and on geometry), random image sizes, depths
and sample counts.  usage: tools/fuzz_parity.py [nscenes] [seed] [mode]   (GPU box; one line per scene, fails on a mismatch)
mode: "main" (default), "valu" (the vector-ALU kernels of the cross-check build libraytrace_gpu_valu.so), "talk" (Talk.hs dialect: rectangles only),
"extreme" (scales from 1e-3 to 1e3, geometry around the eye and the viewport plane, degenerate leaves, lights on vertices),
"big" (90 to 3000 leaves in nested lists, small images), "height" (linearInterpolation height fields with the gaussian
normals, up to 32 point lights), "worlds" (point lights that close over a Shape other than the traced world).  Every mode also
renders random row bands."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding as ob  # noqa: E402
from raytrace_amd import _lib, api  # noqa: E402


def talk_scene(rng):
    def pt():
        return tuple(map(float, rng.uniform((-6, -4, -25), (6, 4, 1))))
    shapes = []
    for _ in range(int(rng.integers(1, 25))):
        s = float(10 ** rng.uniform(-1, 1.2))
        col = api.COLOURS[int(rng.integers(len(api.COLOURS)))]
        if rng.random() < 0.3:
            shapes.append(api.talk_cube([api.COLOURS[int(rng.integers(len(api.COLOURS)))] for _ in range(int(rng.integers(1, 7)))], s, pt()))
        else:
            shapes.append(api.talk_rectangle(col, pt(), tuple(map(float, rng.normal(size=3) * s)), tuple(map(float, rng.normal(size=3) * s))))
    world = shapes[-1]
    for sh in reversed(shapes[:-1]):
        world = api.mappend(sh, world)                       # a <> (b <> ...), Data.Monoid's infixr
    lights = [api.ambient(float(rng.uniform(0, 0.4)))]
    for _ in range(int(rng.integers(0, 4))):
        lights.append(api.talk_pointLight(float(rng.uniform(0.05, 1.5)), pt()))
    return api.TalkDialect(world), api.mconcat_lights(lights)


def extreme_scene(rng):
    scale = float(10 ** rng.uniform(-3, 3))
    def mat():
        return api.COLOURS[int(rng.integers(len(api.COLOURS)))].replace(specularity=float(rng.choice([1, 30, 400])), reflectivity=float(rng.choice([0, 0, 0.5])))
    def pt():
        return rng.uniform((-3, -2, -6), (3, 2, 1.5)) * scale
    shapes, verts = [], []
    for _ in range(int(rng.integers(1, 30))):
        k = int(rng.integers(5))
        s = scale * float(10 ** rng.uniform(-3, 1))
        p = pt()
        if k == 0:
            a, b = p + rng.normal(size=3) * s, p + rng.normal(size=3) * s
            if rng.random() < 0.1:
                b = a.copy()                                                             # zero area
            verts += [p, a, b]
            shapes.append(api.triangle(mat(), tuple(p), tuple(a), tuple(b)))
        elif k == 1:
            w, h = rng.normal(size=3) * s, rng.normal(size=3) * s
            if rng.random() < 0.1:
                h = w * 2.0                                                              # parallel edges: NaN normal
            verts.append(p)
            shapes.append(api.rectangle(mat(), tuple(p), tuple(w), tuple(h)))
        elif k == 2:
            shapes.append(api.colorcuboid([mat()], tuple(p), *(float(v) for v in rng.uniform(0.01, 3, 3) * scale)))
        elif k == 3:
            shapes.append(api.sphere(mat(), tuple(p), float(rng.uniform(0.01, 2) * scale)))
        else:
            shapes.append(api.cylinder(mat(), mat(), mat(), tuple(p), int(rng.integers(1, 9)), float(rng.uniform(0.05, 3) * scale), float(rng.uniform(0.05, 1.5) * scale)))
    lights = [api.ambient(float(rng.uniform(0, 0.3)))]
    for _ in range(int(rng.integers(0, 5))):
        lp = verts[int(rng.integers(len(verts)))] if verts and rng.random() < 0.3 else pt() * 1.5
        lights.append(api.pointLight(float(rng.uniform(0.05, 0.9)), float(rng.uniform(0, 0.9)), tuple(map(float, lp))))
    return api.mconcat(shapes), api.mconcat_lights(lights)


def worlds_scene(rng):
    """`pointLight world ...` (Main.hs:467): most point lights close over a Shape of their own — another random scene, or
    mempty — instead of the traced world."""
    world, lights = random_scene(rng)
    others = [random_scene(rng)[0] for _ in range(int(rng.integers(1, 3)))] + [api.mconcat([])]
    out = []
    for l in lights:
        if isinstance(l, api.PointLight) and rng.random() < 0.6:
            out.append(api.PointLight(l.diffuse, l.specular, l.pos, others[int(rng.integers(len(others)))]))
        else:
            out.append(l)
    if not any(isinstance(l, api.PointLight) for l in out):
        out.append(api.PointLight(0.4, 0.4, (0.0, 3.0, -2.0), others[0]))
    return world, out


def big_scene(rng):
    """Hundreds to thousands of leaves: exercises the block loop of the leaf stream (32 leaves per block, 3 per turn)."""
    n = int(rng.integers(90, 3000))
    def mat():
        return api.COLOURS[int(rng.integers(len(api.COLOURS)))].replace(specularity=float(rng.choice([1, 20, 100])))
    shapes = []
    while sum(6 if isinstance(s_, api.Cuboid) else 1 for s_ in shapes) < n:
        c = rng.uniform((-8, -5, -28), (8, 5, -3))
        if rng.random() < 0.6:
            shapes.append(api.triangle(mat(), tuple(c), tuple(c + rng.normal(size=3) * 0.4), tuple(c + rng.normal(size=3) * 0.4)))
        elif rng.random() < 0.5:
            shapes.append(api.colorcuboid([mat()], tuple(c), *(float(v) for v in rng.uniform(0.1, 1.0, 3))))
        else:
            shapes.append(api.rectangle(mat(), tuple(c), tuple(rng.normal(size=3) * 0.5), tuple(rng.normal(size=3) * 0.5)))
    k = int(rng.integers(1, 6))                           # a few nested lists
    cuts = sorted(set(int(v) for v in rng.integers(1, len(shapes), k)))
    parts = [api.mconcat(shapes[a:b]) for a, b in zip([0] + cuts, cuts + [len(shapes)])]
    lights = [api.ambient(0.1)] + [api.pointLight(0.3, 0.3, tuple(map(float, rng.uniform((-8, 4, -28), (8, 12, 0))))) for _ in range(int(rng.integers(1, 4)))]
    return api.mconcat(parts), api.mconcat_lights(lights)


def height_scene(rng):
    """linearInterpolation (Main.hs:937-988) with random grid, origin and scale, with and without the gaussian fnorm,
    plus a few more shapes and up to 32 point lights."""
    a = float(rng.uniform(0.3, 1.5))
    step = float(rng.choice([0.1, 0.2, 0.25, 0.5])) * a
    origin = tuple(map(float, rng.uniform((-3, -8, -30), (3, -2, -8))))
    world, lights = api.linearInterpolation(api.gaussian, api.GAUSSIAN_FNORM if rng.random() < 0.7 else None,
                                            (-a, -a), (a, a), step, origin, float(rng.uniform(2, 12)))
    extra = [api.colorcuboid([api.red, api.green], tuple(map(float, rng.uniform((-6, -6, -28), (6, 2, -6)))), 1.0, 2.0, 1.5)
             for _ in range(int(rng.integers(0, 4)))]
    if extra:
        world = api.mconcat([world] + extra) if rng.random() < 0.5 else api.mappend(world, api.mconcat(extra))
    more = [api.pointLight(float(rng.uniform(0.02, 0.3)), float(rng.uniform(0, 0.5)), tuple(map(float, rng.uniform((-20, -5, -40), (20, 60, 5)))))
            for _ in range(int(rng.choice([0, 1, 3, 31])))]
    return world, api.mconcat_lights([lights] + more)


def random_scene(rng):
    def mat():
        m = api.COLOURS[int(rng.integers(len(api.COLOURS)))]
        return m.replace(specularity=float(rng.choice([1, 5, 40, 200])),
                         reflectivity=float(rng.choice([0, 0, 0, 0.3, 0.8])))

    def pt(scale=1.0):
        return tuple(map(float, rng.uniform((-6 * scale, -4 * scale, -25), (6 * scale, 4 * scale, 1))))

    def vec(s):
        return tuple(map(float, rng.normal(size=3) * s))

    shapes = []
    for _ in range(int(rng.integers(1, 40))):
        k = int(rng.integers(6))
        s = float(10 ** rng.uniform(-1.5, 1.0))
        if k == 0:
            p = np.array(pt())
            shapes.append(api.triangle(mat(), tuple(p), tuple(p + np.array(vec(s))), tuple(p + np.array(vec(s)))))
        elif k == 1:
            shapes.append(api.rectangle(mat(), pt(), vec(s), vec(s)))                    # edge vectors need not be orthogonal
        elif k == 2:
            shapes.append(api.colorcuboid([mat() for _ in range(int(rng.integers(1, 4)))], pt(), *(float(v) for v in rng.uniform(0.1, 3, 3))))
        elif k == 3:
            shapes.append(api.cylinder(mat(), mat(), mat(), pt(), int(rng.integers(3, 12)), float(rng.uniform(0.2, 3)), float(rng.uniform(0.1, 1.5))))
        elif k == 4:
            shapes.append(api.sphere(mat(), pt(), float(rng.uniform(0.2, 2))))
        else:                                                                            # axis-aligned wall or floor
            ax = int(rng.integers(3))
            w = [0.0, 0.0, 0.0]; h = [0.0, 0.0, 0.0]
            w[(ax + 1) % 3] = float(rng.uniform(2, 40)); h[(ax + 2) % 3] = float(rng.uniform(2, 40)) * float(rng.choice([-1, 1]))
            shapes.append(api.rectangle(mat(), pt(), tuple(w), tuple(h)))
    # random tree shape: nested mconcat / mappend
    while len(shapes) > 1 and rng.random() < 0.7:
        i = int(rng.integers(len(shapes) - 1))
        if rng.random() < 0.5:
            shapes[i:i + 2] = [api.mappend(shapes[i], shapes[i + 1])]
        else:
            j = min(len(shapes), i + int(rng.integers(2, 6)))
            shapes[i:j] = [api.mconcat(shapes[i:j])]
    world = api.mconcat(shapes) if rng.random() < 0.8 else shapes[0]
    lights = [api.ambient(float(rng.uniform(0, 0.4)))]
    for _ in range(int(rng.integers(0, 5))):
        lights.append(api.pointLight(float(rng.uniform(0.05, 0.9)), float(rng.uniform(0, 0.9)), pt(1.5)))
    return world, api.mconcat_lights(lights)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    mode = sys.argv[3] if len(sys.argv) > 3 else "main"
    L = _lib.valu_lib() if mode == "valu" else _lib.lib()       # "valu": the cross-check build, vector-ALU first stage
    gen = {"main": random_scene, "valu": random_scene, "talk": talk_scene, "extreme": extreme_scene, "big": big_scene, "height": height_scene,
           "worlds": worlds_scene}[mode]
    _lib.check(L.rt_init(0))
    rng = np.random.default_rng(seed)
    bad = 0
    for i in range(n):
        world, lights = gen(rng)
        w, h = int(rng.integers(17, 200)), int(rng.integers(9, 160))
        if mode == "big":
            w, h = int(rng.integers(17, 80)), int(rng.integers(9, 60))
        depth = int(rng.integers(1, 5))
        spp = int(rng.choice([1, 1, 1, 4]))
        row0 = int(rng.integers(0, h)) if rng.random() < 0.3 else 0          # sometimes a row band, as a rank renders it
        nrows = int(rng.integers(1, h - row0 + 1)) if row0 or rng.random() < 0.2 else h
        ref = ob.OracleScene(world, lights).render(w, h, depth=depth, spp=spp, seed=i, row0=row0, nrows=nrows, flags=(spp == 1))
        sc = api.Scene(world, lights, lib=L)
        if mode == "valu":
            sc.set_option(_lib.RT_OPT_FILTER, 1)
        rgba, hit = sc.render(w, h, depth=depth, spp=spp, seed=i, row0=row0, nrows=nrows)
        # ... again: the second render of a band dispatches its waves by the first one's durations (RT_OPT_HISTORY), and every
        # other scene also as 3 bytes per pixel (RT_OPT_RGB24); both must give the same bytes
        if i & 1:
            sc.set_option(_lib.RT_OPT_RGB24, 1)
        rgba2, hit2 = sc.render(w, h, depth=depth, spp=spp, seed=i, row0=row0, nrows=nrows)
        again_ok = bool((rgba2[:, :, :3] == rgba[:, :, :3]).all() and (hit2 == hit).all())
        leaves = sc.num_leaves
        sc.close()
        d_id = hit != ref["hit_id"]
        if ref["flags"] is not None:
            d_id &= (ref["flags"] & ob.FLAG_TIE) == 0
        n_id = int(d_id.sum())
        n_rgb = int((rgba[:, :, :3] != ref["rgb"]).any(axis=2).sum())
        print(f"scene {i}: {leaves} leaves, {len(lights)} lights, {w}x{h} depth {depth} spp {spp}: hit-ID mismatches {n_id}, colour mismatches {n_rgb}"
              f"{'' if again_ok else '; SECOND RENDER DIFFERS'}", flush=True)
        if n_id or n_rgb or not again_ok:
            bad += 1
    print(f"FUZZ: {n} scenes (seed {seed}{'' if mode == 'main' else ', ' + mode}), {bad} with a mismatch")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
