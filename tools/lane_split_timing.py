#!/usr/bin/env python3
"""One frame as ONE rt_render_device call, against the same frame as pieces on the scene's two lanes (RT_OPT_LANE): (a) two
independent streams, (b) forked from and joined to one stream with events, the way a library-internal split would have to.
usage: tools/lane_split_timing.py [w h [lanes]]   (lanes > 2 needs a library built with -DRT_LANES_N=lanes)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytrace_amd import api, scenes, _lib

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
L = _lib.lib()
_lib.check(L.rt_init(0))
sc = api.Scene(*scenes.synthetic())
sc.prepare(w, h)
dev = torch.device("cuda", 0)
frame = torch.empty((h, w, 4), dtype=torch.uint8, device=dev)
hit = torch.empty((h, w), dtype=torch.int32, device=dev)
NL = int(sys.argv[3]) if len(sys.argv) > 3 else 2
ss = [torch.cuda.Stream(dev) for _ in range(NL)]
s0 = ss[0]
fork, joins = torch.cuda.Event(), [torch.cuda.Event() for _ in range(NL)]


def piece(lane, r0, nr, st, cont):
    sc.set_option(_lib.RT_OPT_LANE, lane).set_option(_lib.RT_OPT_CONTINUE, int(cont))
    sc.render_device(w, h, frame[r0:].data_ptr(), hit[r0:].data_ptr(), row0=r0, nrows=nr, stream=st.cuda_stream)


def one():
    piece(0, 0, h, s0, False)


def split(cuts, forkjoin):
    def f():
        if forkjoin:
            fork.record(s0)
            for x in ss[1:]:
                x.wait_event(fork)
        for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
            piece(i % NL, a, b - a, ss[i % NL], i > 0)
        if forkjoin:
            for k in range(1, NL):
                joins[k].record(ss[k]); s0.wait_event(joins[k])
    return f


def timed(f, frames=40):
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(frames):
            f()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / frames * 1e3)
    return best


one(); torch.cuda.synchronize()
ref = frame.clone()
print(f"{w}x{h}, {NL} lanes: one call {timed(one):.3f} ms")
r8 = lambda x: int(x) // 8 * 8
for name, cuts in (("2 halves", [0, r8(h / 2), h]), ("0.45/0.55", [0, r8(h * 0.45), h]), ("4 quarters", [0, r8(h / 4), r8(h / 2), r8(3 * h / 4), h]),
                   ("1:2:2:1", [0, r8(h / 6), r8(h / 2), r8(5 * h / 6), h]), ("3 thirds", [0, r8(h / 3), r8(2 * h / 3), h]),
                   ("8 eighths", [r8(h * k / 8) for k in range(8)] + [h])):
    for fj in (False, True):
        frame.zero_()
        t = timed(split(cuts, fj))
        print(f"   {name:10s} {'fork/join on one stream' if fj else 'two independent streams'}: {t:.3f} ms; same frame: {bool(torch.equal(frame, ref))}", flush=True)
sc.set_option(_lib.RT_OPT_LANE, -1).set_option(_lib.RT_OPT_CONTINUE, 0)     # the library's own choice (splits bands of 3 Mi pixels or more)
frame.zero_()
t = timed(lambda: sc.render_device(w, h, frame.data_ptr(), hit.data_ptr(), stream=s0.cuda_stream))
print(f"   rt_render_device left to itself (RT_OPT_LANE -1): {t:.3f} ms; same frame: {bool(torch.equal(frame, ref))}; kernels per stats {sc.stats()['ms_total']:.3f} ms")
t = timed(lambda: sc.render_device(w, h, frame.data_ptr(), hit.data_ptr()))
print(f"   the same on the library's own streams (stream = NULL): {t:.3f} ms; kernels per stats {sc.stats()['ms_total']:.3f} ms")
def lib_streams():
    sc.set_option(_lib.RT_OPT_LANE, 0).set_option(_lib.RT_OPT_CONTINUE, 0)
    sc.render_device(w, h, frame.data_ptr(), hit.data_ptr(), row0=0, nrows=r8(h / 2))
    sc.set_option(_lib.RT_OPT_LANE, 1)
    sc.render_device(w, h, frame[r8(h / 2):].data_ptr(), hit[r8(h / 2):].data_ptr(), row0=r8(h / 2), nrows=h - r8(h / 2))
t = timed(lib_streams)
print(f"   2 halves, explicit lanes on the library's own streams, no fork/join: {t:.3f} ms")
sc.close()
