#!/usr/bin/env python3
"""Throughput of a SEQUENCE of independent frames when consecutive frames alternate between the scene's two lanes
(RT_OPT_LANE: each lane has its own work buffers, wave history and stream), against the same frames one after the other
on one lane (what bench.py times as `value`).  Two frames are then in flight: the kernels of frame k + 1 fill the tails
of frame k's.  usage: tools/frames_in_flight.py [frames=60] [workload=synthetic|mirror|default]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytrace_amd import _lib, api, scenes

_lib.check(_lib.lib().rt_init(0))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
workload = sys.argv[2] if len(sys.argv) > 2 else "synthetic"
w, h = 3840, 2160
world, lights = {"synthetic": scenes.synthetic, "mirror": scenes.mirror_bench, "default": scenes.default_scene}[workload]()
sc = api.Scene(world, lights)
sc.set_option(_lib.RT_OPT_LANE, 0)          # (a reflective frame would otherwise be split over both lanes by the library itself)
bufs = [(torch.empty((h, w, 4), dtype=torch.uint8, device="cuda"), torch.empty((h, w), dtype=torch.int32, device="cuda")) for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def run(lanes):
    for warm in (True, False):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(20 if warm else n):
            ln = k % lanes
            rgba, hit = bufs[ln]
            sc.render_device(w, h, rgba.data_ptr(), hit.data_ptr(), stream=streams[ln].cuda_stream, lane=ln)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return dt / n * 1e3


one = run(1)
ref = [b.clone() for b in bufs[0]]
two = run(2)
same = all(torch.equal(a, b) for pair in bufs for a, b in zip(pair, ref))
print(f"{workload} {w}x{h}, {n} frames: one after the other on one lane {one:.3f} ms per frame ({w * h / one / 1e3:.0f} Mrays/s); "
      f"alternating between the two lanes {two:.3f} ms per frame ({w * h / two / 1e3:.0f} Mrays/s); frames identical: {same}")
