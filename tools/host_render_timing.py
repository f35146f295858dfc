#!/usr/bin/env python3
"""PCIe-inclusive timing of rt_render (host buffers) on the benchmark frame, for DESIGN.md §8 (never bench.py's value).

Usage: host_render_timing.py [bands ...]   -- RT_OPT_HOST_BANDS values to try (default: the library's own choice)
Reports pageable (numpy) and page-locked (rt_host_alloc) destinations, buffers allocated once and reused."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytrace_amd import api, scenes, _lib
_lib.check(_lib.lib().rt_init(0))
world, lights = scenes.synthetic()
sc = api.Scene(world, lights)
w, h = 3840, 2160
ref_rgba, ref_hit = sc.render(w, h)
dests = {"pageable": (np.zeros((h, w, 4), np.uint8), np.zeros((h, w), np.int32)),
         "page-locked": (api.host_buffer((h, w, 4), np.uint8), api.host_buffer((h, w), np.int32))}
rgb24 = api.host_buffer((h, w, 3), np.uint8)
for bands in (sys.argv[1:] or [""]):
    if bands:
        sc.set_option(_lib.RT_OPT_HOST_BANDS, int(bands))
    for name, (rgba, hit) in dests.items():
        rgba[:] = 0; hit[:] = 0
        ts = []
        for _ in range(6):
            t0 = time.perf_counter(); sc.render(w, h, out_rgba=rgba, out_hit=hit); ts.append(time.perf_counter() - t0)
        st = sc.stats()
        same = bool((rgba == ref_rgba).all() and (hit == ref_hit).all())
        if name == "page-locked":                       # colours only (hit_id_out = NULL): what a caller that wants the image asks for
            tc = []
            for _ in range(6):
                t0 = time.perf_counter(); sc.render(w, h, out_rgba=rgba, want_hit=False); tc.append(time.perf_counter() - t0)
            print(f"rt_render {w}x{h} -> {name:11s} bands={bands or 'auto'}, colours only: min {min(tc)*1e3:.2f} ms, median {sorted(tc)[3]*1e3:.2f} ms; "
                  f"kernels {sc.stats()['ms_total']:.2f} ms; same: {bool((rgba == ref_rgba).all())}", flush=True)
            sc.set_option(_lib.RT_OPT_RGB24, 1)             # ... as 3 bytes per pixel (the reference's PixelRGB8)
            rgb = rgb24
            tc = []
            for _ in range(6):
                t0 = time.perf_counter(); sc.render(w, h, out_rgba=rgb, want_hit=False); tc.append(time.perf_counter() - t0)
            print(f"rt_render {w}x{h} -> {name:11s} bands={bands or 'auto'}, colours only, RGB24: min {min(tc)*1e3:.2f} ms, median {sorted(tc)[3]*1e3:.2f} ms; "
                  f"kernels {sc.stats()['ms_total']:.2f} ms; same: {bool((rgb == ref_rgba[:, :, :3]).all())}", flush=True)
            sc.set_option(_lib.RT_OPT_RGB24, 0)
        print(f"rt_render {w}x{h} -> {name:11s} bands={bands or 'auto'}: min {min(ts)*1e3:.2f} ms, median {sorted(ts)[3]*1e3:.2f} ms "
              f"({w*h/min(ts)/1e6:.0f} Mrays/s); kernels {st['ms_total']:.2f} ms; primary rays {st['primary_rays']}; same as first render: {same}", flush=True)
