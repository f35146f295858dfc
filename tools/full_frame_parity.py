#!/usr/bin/env python3
"""Whole benchmark frame (configs[2]: 16 000 leaves, 4 lights, 3840 x 2160) on the GPU against the CPU oracle, every pixel.
Takes ~8 minutes of 16 host threads; prints a line per band so that the run is seen to be alive.  Not a test: the test
suite checks sampled rows of this frame; this is the full comparison quoted in DESIGN.md."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding as ob  # noqa: E402
from raytrace_amd import _lib, api, scenes  # noqa: E402

_lib.check(_lib.lib().rt_init(0))
# usage: tools/full_frame_parity.py [mirror [w h]]   (mirror: the reflective workload of bench.py --workload mirror, trace depth 4;
#        from 3 Mi pixels on the library renders it as two halves on its two lanes)
mirror = len(sys.argv) > 1 and sys.argv[1] == "mirror"
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
world, lights = scenes.mirror_bench() if mirror else scenes.synthetic()
sc = api.Scene(world, lights)
for _ in range(3):                 # (the third render dispatches its waves longest first, by the history of the second)
    rgba, hit = sc.render(w, h)
sc.close()
osc = ob.OracleScene(world, lights)
bad_id = bad_rgb = 0
t0 = time.time()
step = 54
for r0 in range(0, h, step):
    nr = min(step, h - r0)
    ref = osc.render(w, h, row0=r0, nrows=nr, nthreads=16)
    bad_id += int((hit[r0:r0 + nr] != ref["hit_id"]).sum())
    bad_rgb += int((rgba[r0:r0 + nr, :, :3] != ref["rgb"]).any(axis=2).sum())
    print(f"rows {r0}..{r0 + nr - 1}: hit-ID mismatches so far {bad_id}, colour mismatches so far {bad_rgb}, {time.time() - t0:.0f} s", flush=True)
print(f"FULL FRAME {'(mirror workload) ' if mirror else ''}{w}x{h}: {w * h} pixels, hit-ID mismatches {bad_id}, colour mismatches {bad_rgb}")
