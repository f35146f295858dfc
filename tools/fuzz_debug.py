#!/usr/bin/env python3
"""Replays scene number K of tools/fuzz_parity.py (same seed) and prints where GPU (both filters) and oracle differ."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import oracle_binding as ob
from raytrace_amd import _lib, api
from fuzz_parity import random_scene, talk_scene, extreme_scene

K = int(sys.argv[1]); seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
mode = sys.argv[3] if len(sys.argv) > 3 else 'main'
gen = {'main': random_scene, 'talk': talk_scene, 'extreme': extreme_scene}[mode]
rng = np.random.default_rng(seed)
for i in range(K + 1):
    world, lights = gen(rng)
    w, h = int(rng.integers(17, 200)), int(rng.integers(9, 160))
    depth = int(rng.integers(1, 5))
    spp = int(rng.choice([1, 1, 1, 4]))
print("scene", K, w, h, depth, spp)
print(world if len(str(world)) < 3000 else type(world))
print(lights)
osc = ob.OracleScene(world, lights)
ref = osc.render(w, h, depth=depth, spp=spp, seed=K, flags=(spp == 1), want_t=True)
refnb = osc.render(w, h, depth=depth, spp=spp, seed=K, use_bbox=0)
out = {}
for mode in ("mfma", "valu"):
    L = _lib.valu_lib() if mode == "valu" else _lib.lib()
    _lib.check(L.rt_init(0))
    sc = api.Scene(world, lights, lib=L)
    if mode == "valu":
        sc.set_option(_lib.RT_OPT_FILTER, 1)
    out[mode] = sc.render(w, h, depth=depth, spp=spp, seed=K)
    if mode == "mfma":
        for li in range(min(sc.num_leaves, 4)):
            print("leaf", li, sc.leaf(li))
    sc.close()
for mode in out:
    rgba, hit = out[mode]
    print(mode, "vs oracle: id", int((hit != ref["hit_id"]).sum()), "rgb", int((rgba[:, :, :3] != ref["rgb"]).any(axis=2).sum()),
          "| vs no-bbox oracle: id", int((hit != refnb["hit_id"]).sum()), "rgb", int((rgba[:, :, :3] != refnb["rgb"]).any(axis=2).sum()))
print("mfma vs valu: id", int((out["mfma"][1] != out["valu"][1]).sum()), "rgb", int((out["mfma"][0] != out["valu"][0]).any(axis=2).sum()))
hit = out["mfma"][1]
ys, xs = np.nonzero(hit != ref["hit_id"])
for y, x in list(zip(ys, xs))[:8]:
    print("pixel", x, y, "gpu", hit[y, x], "oracle", ref["hit_id"][y, x], "nobbox", refnb["hit_id"][y, x], "t", ref["t"][y, x], "flags", None if ref["flags"] is None else ref["flags"][y, x])
