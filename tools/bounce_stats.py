#!/usr/bin/env python3
"""Cost of the bounce levels (k_general) on the mirror workload: per trace depth, the nearest-hit time, reflected rays,
blocks of leaves looked at and exact evaluations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytrace_amd import api, scenes, _lib
_lib.check(_lib.lib().rt_init(0))
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
world, lights = scenes.mirror_bench()
sc = api.Scene(world, lights)
if len(sys.argv) > 3 and sys.argv[3] == "nocull":
    sc.set_option(_lib.RT_OPT_CULL, 0)
rgba = torch.empty((h, w, 4), dtype=torch.uint8, device="cuda")
hit = torch.empty((h, w), dtype=torch.int32, device="cuda")
prev = None
first = None
for depth in (1, 2, 3, 4):
    # per-kernel times: the frame as ONE chain of launches (lane 0); the library's own schedule (reflective frames of
    # 3 Mi pixels or more: two halves on two lanes, whose kernels overlap) gives the frame time printed beside it
    sc.set_option(_lib.RT_OPT_LANE, -1)
    for _ in range(2):
        sc.render_device(w, h, rgba.data_ptr(), hit.data_ptr(), depth=depth)
        auto = sc.stats()
    sc.set_option(_lib.RT_OPT_LANE, 0)
    for _ in range(2):
        sc.render_device(w, h, rgba.data_ptr(), hit.data_ptr(), depth=depth)
        st = sc.stats()
    line = f"depth {depth}: nearest-hit kernels {st['ms_primary']:.3f} ms, setup {st['ms_setup']:.3f}, shadow {st['ms_shadow']:.3f}, final {st['ms_final']:.3f}, total {st['ms_total']:.3f} (frame as the library schedules it: {auto['ms_total']:.3f}); bounce rays {st['bounce_rays']}, bounce blocks {st['bounce_blocks']}, exact {st['exact_evals']}, records {st['record_tests']}"
    if prev is not None and st["bounce_rays"] > prev["bounce_rays"]:
        dr = st["bounce_rays"] - prev["bounce_rays"]
        dms = st["ms_primary"] - prev["ms_primary"]
        nblk = -(-4000 // 32) + -(-(sc.num_leaves - 4000 - 1) // 32)
        line += f"\n   this level: {dr} rays in {dms:.3f} ms = {dr / dms / 1e3:.0f} Mrays/s; blocks looked at per wave {(st['bounce_blocks'] - prev['bounce_blocks']) / (dr / 64):.1f} of ~{nblk}; exact per ray {(st['exact_evals'] - prev['exact_evals']) / dr:.2f} (incl. the level's shadow rays)"
    if first is None:
        first = st
        line += f"\n   primary level: {w * h} rays in {st['ms_primary']:.3f} ms = {w * h / st['ms_primary'] / 1e3:.0f} Mrays/s"
    print(line)
    prev = st
print(f"records = FP32 record tests so far (all kernels)")
