#!/usr/bin/env python3
"""PCIe-inclusive timing of rt_render (host buffers) on the benchmark frame, for DESIGN.md §8 (never bench.py's value)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytrace_amd import api, scenes, _lib
_lib.check(_lib.lib().rt_init(0))
world, lights = scenes.synthetic()
sc = api.Scene(world, lights)
w, h = 3840, 2160
sc.render(w, h)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); sc.render(w, h); ts.append(time.perf_counter() - t0)
st = sc.stats()
print(f"rt_render host buffers {w}x{h}: min {min(ts)*1e3:.2f} ms, median {sorted(ts)[2]*1e3:.2f} ms per frame "
      f"({w*h/min(ts)/1e6:.0f} Mrays/s); kernels {st['ms_total']:.2f} ms")
