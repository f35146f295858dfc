#!/usr/bin/env python3
"""bench.py — primary Mrays/s of the MI355X render path (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

A "step" renders one frame of the synthetic scene of SURVEY.md §8(d) (10 000 triangles + 1 000
cuboids = 16 000 leaves, 4 point lights with shadow rays, brute force over all leaves):
  N = 1 : 3840 x 2160 on one GPU                                   (BASELINE.json configs[2])
  N > 1 : 16384 x 16384 split into N contiguous row bands, one per rank, cost-balanced from a
          1/16-scale pre-render; the bands are sent to rank 0 over RCCL (xGMI) inside the timed
          region                                                   (BASELINE.json configs[3])
The scene and its FP32 filter records are resident in HBM before the timed region; the frame
buffer stays in HBM.  value = primary rays of the whole frame / max-over-ranks wall time.

Also on the JSON line:
  roofline      — the dominant kernel (k_shadow: the shadow-ray loops) against the FP32 vector
                  peak, in ALGORITHMIC flops (52 / 44 flops per ray x triangle / rectangle test + 1
                  for the nearest select, SURVEY.md §8 d), timed with HIP events on the render stream.
  cpu_baseline  — the CPU oracle (a C restatement of the reference, NOT the Haskell program) on the
                  host cores, on a bounded band of the same frame (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from raytrace_amd import api, scenes, _lib  # noqa: E402
from raytrace_amd.distributed import gather_bands, split_rows  # noqa: E402

FLOPS_TRI, FLOPS_RECT = 52 + 1, 44 + 1
PEAK_FP32_TFLOPS = 157.3                        # MI355X_MICROARCH.md: peak FP32 (vector), packed FMA
# HBM bytes of one k_shadow launch on the default N = 1 workload, from the PMC passes committed under
# profiles/ (r01_pmc_rd / r01_pmc_wr: FETCH_SIZE and WRITE_SIZE are in KiB and are summed as reported;
# the guide's x2 correction applies to 16-B-per-lane streaming reads only, which this kernel has none of).
# The counters cannot be read from inside this process; for any other workload traffic is null.
K_SHADOW_TRAFFIC_BYTES_DEFAULT = (329533 + 24400) * 1024


def balanced_bands(scene, w, h, n, depth):
    """Contiguous row bands of about equal cost, from a 1/16-scale render (rows of a miss-only
    region cost the primary loop only; hit rows add up to 4 shadow loops)."""
    if n == 1:
        return [(0, h)]
    sw, sh = max(64, w // 16), max(64, h // 16)
    _, hit = scene.render(sw, sh, depth=depth, want_rgba=True)
    st = scene.stats()
    shadow_per_hit = st["shadow_rays"] / max(1, int((hit >= 0).sum()))
    cost = 1.0 + (hit >= 0).mean(axis=1) * shadow_per_hit          # per low-res row, in units of one primary loop
    return split_rows(cost, h, n)


def cpu_baseline(world, lights, w, h, depth, seconds=18.0):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    cores = min(16, os.cpu_count() or 1)
    osc = ob.OracleScene(world, lights)
    row0 = h // 2 - 1
    osc.render(w, h, depth=depth, row0=row0, nrows=1, nthreads=1)       # first call numbers the leaves
    t0 = time.perf_counter()
    osc.render(w, h, depth=depth, row0=row0, nrows=2, nthreads=2)       # calibration: 2 rows on 2 threads
    dt = time.perf_counter() - t0
    nrows = int(max(cores, min(h // 2, cores * seconds / max(dt, 1e-6))))
    nrows -= nrows % 2
    row0 = h // 2 - nrows // 2
    t0 = time.perf_counter()
    osc.render(w, h, depth=depth, row0=row0, nrows=nrows, nthreads=cores)
    dt = time.perf_counter() - t0
    osc.close()
    return {"value": w * nrows / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"rows {row0}..{row0 + nrows - 1} ({nrows} of {h}) of the same {w}x{h} frame, "
                      f"{dt:.1f} s; C restatement of the reference (oracle/), not the Haskell program"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--ntri", type=int, default=10000)
    ap.add_argument("--ncuboid", type=int, default=1000)
    ap.add_argument("--depth", type=int, default=4)
    ap.add_argument("--workload", default="synthetic", choices=["synthetic", "cylinders", "default"],
                    help="synthetic = configs[2]/[3] (the metric's workload); cylinders = configs[4] (specular cylinders + cuboids, "
                         "use with --spp 16 --width 7680 --height 4320); default = the Main.hs scene (configs[0]/[1])")
    ap.add_argument("--spp", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, the measured path) or gloo (rehearsal on one GPU)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank renders on GPU 0 (gloo only)")
    ap.add_argument("--verify", action="store_true", help="rank 0 re-renders the whole frame alone and compares it with the gathered one")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    n = args.gpus
    if world_size != n:
        raise SystemExit(f"--gpus {n} but WORLD_SIZE={world_size}: launch with torch.distributed.run --nproc-per-node {n}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the render path has no CPU fallback")
    if args.same_device:
        if args.dist_backend != "gloo":
            raise SystemExit("--same-device is a rehearsal mode and needs --dist-backend gloo (RCCL refuses two ranks on one GPU)")
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if n > 1:
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)       # RCCL over xGMI
        else:
            dist.init_process_group("gloo")
    staged = n > 1 and args.dist_backend != "nccl"               # gloo rehearsal: bands travel through host memory
    _lib.check(_lib.lib().rt_init(local_rank), "rt_init")

    w = args.width or (3840 if n == 1 else 16384)
    h = args.height or (2160 if n == 1 else 16384)
    if args.workload == "synthetic":
        world, lights = scenes.synthetic(ntri=args.ntri, ncuboid=args.ncuboid)
        ntri, nrect, nlt = args.ntri, 6 * args.ncuboid, 4
        wname = f"synthetic {ntri} triangles + {args.ncuboid} cuboids"
    elif args.workload == "cylinders":
        world, lights = scenes.synthetic_cylinders()
        ntri, nrect, nlt = 200 * 32, 200 * 16 + 500 * 6, 4
        wname = "200 upright cylinders (n=16) + 500 cuboids, specularity 100/400"
    else:
        world, lights = scenes.default_scene()
        ntri, nrect, nlt = 200, 1, 1
        wname = "Main.hs default scene (gaussian height field, linearInterpolation)"
    scene = api.Scene(world, lights)
    assert scene.num_leaves == ntri + nrect

    bands = balanced_bands(scene, w, h, n, args.depth)
    row0, nrows = bands[rank]
    # frame buffer: rank 0 owns the whole image (bands of the other ranks land in it), others own a band
    frame = torch.empty((h if rank == 0 else nrows, w, 4), dtype=torch.uint8, device=dev)
    mine = frame[row0:row0 + nrows] if rank == 0 else frame
    hit = torch.empty((nrows, w), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        scene.render_device(w, h, mine.data_ptr(), hit.data_ptr(), depth=args.depth, spp=args.spp, seed=20260355,
                            row0=row0, nrows=nrows, stream=stream)
        if n > 1 and not staged:   # row bands -> GPU 0 (ncclSend/ncclRecv in one group: every peer has a direct xGMI link)
            gather_bands(dist, frame, mine, bands, rank)
        elif staged:
            torch.cuda.synchronize()
            host_band = mine.cpu()
            host_frame = torch.empty((h, w, 4), dtype=torch.uint8) if rank == 0 else None
            gather_bands(dist, host_frame, host_band, bands, rank)
            if rank == 0:
                host_frame[row0:row0 + nrows] = host_band
                frame.copy_(host_frame)

    def sync():
        torch.cuda.synchronize()
        if n > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    shade_ms, prim_ms, stats = [], [], None
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        if n == 1:      # per-kernel HIP-event times (events are recorded on the render stream by the library)
            stats = scene.stats()
            shade_ms.append(stats["ms_shadow"]); prim_ms.append(stats["ms_primary"])
    sync()
    dt = time.perf_counter() - t0
    if n > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        if staged:
            tmax = tmax.cpu()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    stats = scene.stats()
    counts = torch.tensor([stats["primary_rays"], stats["shadow_rays"], stats["exact_evals"]], dtype=torch.float64, device=dev)
    shade_t = torch.tensor([stats["ms_shadow"], stats["ms_primary"]], dtype=torch.float64, device=dev)
    if n > 1:
        if staged:
            counts, shade_t = counts.cpu(), shade_t.cpu()
        dist.all_reduce(counts)
        dist.all_reduce(shade_t, op=dist.ReduceOp.MAX)
    verified = None
    if args.verify and rank == 0:
        alone = torch.empty((h, w, 4), dtype=torch.uint8, device=dev)
        scene.render_device(w, h, alone.data_ptr(), None, depth=args.depth, spp=args.spp, seed=20260355, stream=stream)
        torch.cuda.synchronize()
        verified = bool(torch.equal(alone, frame))
    if rank == 0:
        primary, shadow, exact = (float(c) for c in counts.tolist())
        ms_step = dt / args.steps * 1e3
        per_test = ntri * FLOPS_TRI + nrect * FLOPS_RECT
        ms_shade = float(np.mean(shade_ms)) if shade_ms else float(shade_t[0])
        ms_prim = float(np.mean(prim_ms)) if prim_ms else float(shade_t[1])
        # per-launch algorithmic flops of the dominant kernel (k_shadow: one filter test per shadow ray x leaf)
        shade_flops = shadow / n * per_test
        achieved = shade_flops / (ms_shade * 1e-3) / 1e12
        out = {
            "metric": "primary Mrays/s (whole node)", "value": primary / (dt / args.steps) / 1e6, "unit": "Mrays/s",
            "n_gpus": n, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "strong" if n > 1 else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{wname} ({ntri + nrect} leaves), {nlt} point light(s) + ambient, "
                                   f"{w}x{h}, depth {args.depth}, spp {args.spp}, brute force"
                                   + (f", {n} cost-balanced row bands gathered to GPU 0 over " + ("RCCL" if not staged else "gloo via host memory (rehearsal)") if n > 1 else ""),
                       "width": w, "height": h, "leaves": ntri + nrect, "lights": nlt, "spp": args.spp,
                       "bands": bands if n > 1 else None, "gathered_frame_equals_single_render": verified,
                       "shadow_rays_per_frame": shadow, "exact_fp64_evals_per_frame": exact,
                       "whole_frame_algorithmic_tflops": (primary + shadow) * per_test / (dt / args.steps) / 1e12,
                       "kernel_ms": {"k_primary": ms_prim, "k_setup": stats["ms_setup"], "k_shadow": ms_shade, "k_final": stats["ms_final"]}},
            "roofline": {"bound": "valu_fp32", "kernel": "k_shadow", "achieved": achieved, "peak": PEAK_FP32_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_TFLOPS,
                         "traffic": K_SHADOW_TRAFFIC_BYTES_DEFAULT if (n == 1 and args.spp == 1 and args.workload == "synthetic" and (w, h, ntri, nrect) == (3840, 2160, 10000, 6000)) else None,
                         "traffic_unit": "bytes per launch (rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE, profiles/r01_pmc_*)",
                         "note": "algorithmic flops (53/45 per ray x triangle/rectangle test) x shadow rays cast x leaves, per launch"},
        }
        if n == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(world, lights, w, h, args.depth)
        print(json.dumps(out), flush=True)
    scene.close()
    if n > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
