#!/usr/bin/env python3
"""What does cutting a rank's band into pieces cost?  Renders one eighth of the 16384 x 16384 frame (the band of one of 8
ranks) on this GPU as 1..8 pieces (sizes 1 : 2 : ... : 2 : 1, raytrace_amd/distributed.py) and reports the time per band
and the overhead per extra piece: the `piece_overhead_ms` the N-GPU pipeline model (plan_frame) rests on.
usage: tools/piece_overhead.py [n_ranks=8]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytrace_amd import api, scenes, _lib
from raytrace_amd.distributed import piece_fractions, sub_bands, plan_frame

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
L = _lib.lib()
_lib.check(L.rt_init(0))
w = h = 16384
sc = api.Scene(*scenes.synthetic())
sc.prepare(w, h)
dev = torch.device("cuda", 0)
rows = h // n
sts = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
res = {}
for lanes in (1, 2):              # 1: every piece on lane 0; 2: pieces alternate between the scene's two lanes (RT_OPT_LANE)
    for which, r0 in (("middle band", (n // 2) * rows), ("top band", 0)):
        frame = torch.empty((rows, w, 4), dtype=torch.uint8, device=dev)
        hit = torch.empty((rows, w), dtype=torch.int32, device=dev)
        for parts in (1, 2, 3, 4, 6, 8):
            subs = [sb for sb in sub_bands((r0, rows), parts, fractions=piece_fractions(parts)) if sb[1] > 0]
            ts = []
            for rep in range(6):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i, (p0, pn) in enumerate(subs):
                    lane = i % lanes
                    sc.set_option(_lib.RT_OPT_LANE, lane)
                    sc.render_device(w, h, frame[p0 - r0:].data_ptr(), hit[p0 - r0:].data_ptr(), row0=p0, nrows=pn, stream=sts[lane].cuda_stream)
                sts[0].synchronize(); sts[1].synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            res[(lanes, which, parts)] = min(ts[1:])
        sc.set_option(_lib.RT_OPT_LANE, -1)
        base = res[(lanes, which, 1)]
        print(f"{lanes} lane(s), {which} (rows {r0}..{r0 + rows - 1} of the {w}x{h} frame): " +
              ", ".join(f"{p} piece(s) {res[(lanes, which, p)]:.3f} ms" for p in (1, 2, 3, 4, 6, 8)))
        print("   overhead per extra piece: " + ", ".join(f"{p}: {(res[(lanes, which, p)] - base) / (p - 1):.3f} ms" for p in (2, 3, 4, 6, 8)), flush=True)
full = sum(v for k, v in res.items() if k[0] == 1 and k[2] == 1) / 2 * n
for o in (0.03, 0.05, 0.1, 0.15):
    for bw in (48.0, 64.0, 76.8):
        parts, fr, wts, t, d = plan_frame(full, n, w * h * 4 / n, bw, o)
        print(f"model: frame on one GPU ~{full:.1f} ms, overhead {o} ms/piece, link {bw} GB/s -> {parts} pieces, rank-0 weight {wts[0]:.3f}, step {t:.2f} ms = {full / t:.2f}x")
sc.close()
