#!/usr/bin/env python3
"""Experiment: does rendering two row bands of the benchmark frame side by side on two streams (two scene handles = two
workspaces) fill the tails of the kernels?  Prints ms per frame for the whole frame on one stream and for several splits,
with the streams joined after every frame."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytrace_amd import api, scenes, _lib
_lib.check(_lib.lib().rt_init(0))
w, h = 3840, 2160
world, lights = scenes.synthetic()
sc = [api.Scene(world, lights) for _ in range(2)]
rgba = torch.empty((h, w, 4), dtype=torch.uint8, device="cuda")
hit = torch.empty((h, w), dtype=torch.int32, device="cuda")
ref_rgba = torch.empty_like(rgba); ref_hit = torch.empty_like(hit)
st = [torch.cuda.Stream(), torch.cuda.Stream()]
sc[0].render_device(w, h, ref_rgba.data_ptr(), ref_hit.data_ptr(), stream=st[0].cuda_stream)
torch.cuda.synchronize()

def frame(bands):
    """bands: list of (row0, nrows, which)"""
    for row0, nrows, k in bands:
        sc[k].render_device(w, h, rgba.data_ptr() + row0 * w * 4, hit.data_ptr() + row0 * w * 4, row0=row0, nrows=nrows, stream=st[k].cuda_stream)

def timed(bands, n=60, join=True):
    for _ in range(25):
        frame(bands)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st[0]); st[1].wait_event(e0)
    for _ in range(n):
        frame(bands)
        if join:
            st[0].wait_stream(st[1]); st[1].wait_stream(st[0])
    st[0].wait_stream(st[1])
    e1.record(st[0])
    torch.cuda.synchronize()
    ok = bool((rgba == ref_rgba).all()) and bool((hit == ref_hit).all())
    return e0.elapsed_time(e1) / n, ok

def split(fracs):
    rows = [int(round(f * h / 8)) * 8 for f in fracs]
    rows[-1] = h - sum(rows[:-1])
    out, r = [], 0
    for i, n in enumerate(rows):
        out.append((r, n, i % 2)); r += n
    return out

print("whole frame, one stream: %.3f ms (same: %s)" % timed([(0, h, 0)]))
for fr in ([0.5, 0.5], [0.4, 0.6], [0.6, 0.4], [0.3, 0.7], [0.25, 0.25, 0.25, 0.25], [0.2, 0.3, 0.3, 0.2]):
    rgba.zero_(); hit.zero_()
    ms, ok = timed(split(fr))
    ms2, ok2 = timed(split(fr), join=False)
    print("split %s: joined %.3f ms, free-running %.3f ms (same: %s %s)" % (fr, ms, ms2, ok, ok2))
