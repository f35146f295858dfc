#!/usr/bin/env python3
"""Generates tests/golden/*.npz: small oracle renders of every reference scene.

The reference (Haskell) cannot run in this image and ships no golden images, so these
vectors come from the CPU oracle (oracle/oracle.c): PARITY UNPINNED with respect to the real program.
They pin the oracle against regressions and give the GPU tests a fixture that does not need the
oracle library.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_binding as ob  # noqa: E402
from raytrace_amd import scenes  # noqa: E402

SIZES = {"default": (96, 96), "planes": (80, 60), "axes": (64, 64), "cubes": (96, 54), "stacked_cubes": (96, 96),
         "triangle_example": (90, 50), "spec_test": (64, 64), "intersection": (96, 96), "cylinder_axes": (80, 64),
         "talk_main2": (64, 64), "talkhs_main2": (96, 96), "mirror_room": (96, 72), "spheres": (96, 96)}


def main():
    for name, (w, h) in SIZES.items():
        world, lights = scenes.REFERENCE_SCENES[name]()
        o = ob.OracleScene(world, lights).render(w, h, depth=4, want_t=True)
        np.savez_compressed(os.path.join(HERE, f"{name}_{w}x{h}.npz"), rgb=o["rgb"], hit_id=o["hit_id"], t=o["t"])
        print(name, w, h, "hit fraction", float((o["hit_id"] >= 0).mean()))
    world, lights = scenes.synthetic(ntri=300, ncuboid=30)
    o = ob.OracleScene(world, lights).render(96, 54, depth=4, want_t=True)
    np.savez_compressed(os.path.join(HERE, "synthetic300+30_96x54.npz"), rgb=o["rgb"], hit_id=o["hit_id"], t=o["t"])
    world, lights = scenes.synthetic_cylinders(ncyl=6, ncuboid=15)
    o = ob.OracleScene(world, lights).render(96, 54, depth=4, spp=4, seed=7, want_t=True)
    np.savez_compressed(os.path.join(HERE, "cylinders6+15_spp4_96x54.npz"), rgb=o["rgb"], hit_id=o["hit_id"], t=o["t"])


if __name__ == "__main__":
    main()
