#!/usr/bin/env python3
"""Differential fuzzing of the GPU path against the CPU oracle: random small scenes (triangles, rectangles with arbitrary edge
vectors, cuboids, cylinders, spheres, mirrors; lights anywhere, also inside and on geometry), random image sizes, depths
and sample counts.  usage: tools/fuzz_parity.py [nscenes] [seed]   (GPU box; prints one line per scene, fails on a mismatch)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding as ob  # noqa: E402
from raytrace_amd import _lib, api  # noqa: E402


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
    _lib.check(_lib.lib().rt_init(0))
    rng = np.random.default_rng(seed)
    bad = 0
    for i in range(n):
        world, lights = random_scene(rng)
        w, h = int(rng.integers(17, 200)), int(rng.integers(9, 160))
        depth = int(rng.integers(1, 5))
        spp = int(rng.choice([1, 1, 1, 4]))
        ref = ob.OracleScene(world, lights).render(w, h, depth=depth, spp=spp, seed=i, flags=(spp == 1))
        sc = api.Scene(world, lights)
        rgba, hit = sc.render(w, h, depth=depth, spp=spp, seed=i)
        leaves = sc.num_leaves
        sc.close()
        d_id = hit != ref["hit_id"]
        if ref["flags"] is not None:
            d_id &= (ref["flags"] & ob.FLAG_TIE) == 0
        n_id = int(d_id.sum())
        n_rgb = int((rgba[:, :, :3] != ref["rgb"]).any(axis=2).sum())
        print(f"scene {i}: {leaves} leaves, {len(lights)} lights, {w}x{h} depth {depth} spp {spp}: hit-ID mismatches {n_id}, colour mismatches {n_rgb}", flush=True)
        if n_id or n_rgb:
            bad += 1
    print(f"FUZZ: {n} scenes (seed {seed}), {bad} with a mismatch")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
