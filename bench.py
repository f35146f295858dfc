#!/usr/bin/env python3
"""bench.py — primary Mrays/s of the MI355X render path (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

A "step" renders one frame of the synthetic scene of SURVEY.md §8(d) (10 000 triangles + 1 000
cuboids = 16 000 leaves, 4 point lights with shadow rays, every ray x leaf pair decided, no hierarchy):
  N = 1 : 3840 x 2160 on one GPU                                   (BASELINE.json configs[2])
  N > 1 : 16384 x 16384 split into N contiguous row bands, one per rank, cost-balanced from a
          1/16-scale pre-render; the bands are sent to rank 0 over RCCL (xGMI) inside the timed
          region                                                   (BASELINE.json configs[3])
The scene and its FP32 filter records are resident in HBM before the timed region; the frame
buffer stays in HBM.  value = primary rays of the whole frame / max-over-ranks wall time.

Also on the JSON line:
  roofline      — the dominant kernel (the longer of k_primary_mfma: every primary ray against every
                  leaf, and k_shadow_mfma: every shadow ray against every leaf; the other one is listed
                  under "other_kernel") in ALGORITHMIC flops (52 / 44 flops per ray x triangle /
                  rectangle test + 1 for the nearest select, SURVEY.md §8 d) over its launch time (HIP
                  events on the render stream).  The first stage of these kernels runs on the matrix
                  cores, so the bound is "mfma" and peak is the dense FP16 MFMA peak; the same rate
                  against the FP32 vector peak of BASELINE.json's metric, and the EXECUTED matrix-core
                  rate (one v_mfma_f32_32x32x16_f16 = 32768 flops per 32 rays x 32 leaves), beside it.
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
from raytrace_amd.distributed import gather_bands, split_rows, sub_bands  # noqa: E402

FLOPS_TRI, FLOPS_RECT = 52 + 1, 44 + 1
PEAK_FP32_TFLOPS = 157.3                        # MI355X_MICROARCH.md: peak FP32 (vector), packed FMA
PEAK_MFMA_F16_TFLOPS = 2500.0                   # MI355X_MICROARCH.md: peak BF16/FP16 MFMA, dense
MFMA_FLOPS = 2 * 32 * 32 * 16                   # one v_mfma_f32_32x32x16_f16
# HBM bytes of one launch on the default N = 1 workload, from the PMC passes committed under profiles/
# (r01_cull_pmc_rd / _wr, separate passes), KiB per dispatch: k_primary_mfma FETCH_SIZE 9 265, WRITE_SIZE 34 145 (the
# hit-ID buffer alone is 4 B x 8 294 400 px = 32 400); k_shadow_mfma FETCH_SIZE 253 938, WRITE_SIZE 24 051 (the hit
# records of 5.6 M shadow rays at 48 B are 263 000).  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE tallies 16-B-per-lane
# reads at half their bytes; the kernels' reads (hit records, block cones, leaf rows, FP32 records) are mostly of that
# kind, so the read figure is doubled (an upper bound); WRITE_SIZE is taken as reported.
# The counters cannot be read from inside this process; for any other workload traffic is null.
TRAFFIC_BYTES_DEFAULT = {"k_primary_mfma": (2 * 9265 + 34145) * 1024, "k_shadow_mfma": (2 * 253938 + 24051) * 1024}


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
    ap.add_argument("--filter", default="mfma", choices=["mfma", "valu"],
                    help="first stage: bounding cones on the matrix cores (default) or the FP32 vector-ALU kernels")
    ap.add_argument("--no-brute-force-leg", action="store_true", help="skip the extra frames rendered with the block test off (profiling runs)")
    ap.add_argument("--no-cull", action="store_true", help="every wave scans every block of the leaf stream (RT_CULL=0): the plain brute-force scan")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, the measured path) or gloo (rehearsal on one GPU)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank renders on GPU 0 (gloo only)")
    ap.add_argument("--pieces", type=int, default=4, help="N > 1: pieces per band; piece k's gather overlaps piece k+1's render")
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
    os.environ["RT_FILTER"] = args.filter
    os.environ["RT_CULL"] = "0" if args.no_cull else "1"
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

    # every band is rendered in `pieces` consecutive pieces; piece k of all peers travels to GPU 0 in one group
    # of send/recv while piece k+1 renders (the library returns once the piece is enqueued on `stream`; RCCL's
    # own stream waits for that piece only), so only the last piece's exchange is exposed
    pieces = max(1, args.pieces) if n > 1 else 1
    subs = [sub_bands(b, pieces) for b in bands]

    def step():
        pending, host = [], []
        host_frame = torch.empty((h, w, 4), dtype=torch.uint8) if staged and rank == 0 else None
        for k in range(pieces):
            r0, nr = subs[rank][k]
            piece = mine[r0 - row0:r0 - row0 + nr]
            if nr > 0:
                scene.render_device(w, h, piece.data_ptr(), hit[r0 - row0:].data_ptr(), depth=args.depth, spp=args.spp,
                                    seed=20260355, row0=r0, nrows=nr, stream=stream)
            if n == 1:
                continue
            group = [s[k] for s in subs]
            if not staged:     # ncclSend/ncclRecv in one group: every peer has a direct xGMI link to GPU 0
                pending += gather_bands(dist, frame, piece, group, rank, wait=False)
            else:              # gloo rehearsal: the same groups, through host memory
                torch.cuda.synchronize()
                host_piece = piece.cpu()
                host.append((r0, nr, host_piece))
                pending += gather_bands(dist, host_frame, host_piece, group, rank, wait=False)
        for req in pending:
            req.wait()
        if staged and rank == 0:
            for r0, nr, host_piece in host:
                host_frame[r0:r0 + nr] = host_piece
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
    if n > 1:            # untimed pass: ray counts and per-kernel times of this rank's band, summed over its pieces
        keys = ("primary_rays", "shadow_rays", "exact_evals", "ms_shadow", "ms_primary", "ms_setup", "ms_final",
                "mfma_blocks_primary", "mfma_blocks_shadow")
        total = dict.fromkeys(keys, 0.0)
        for r0, nr in subs[rank]:
            if nr > 0:
                scene.render_device(w, h, mine[r0 - row0:].data_ptr(), hit[r0 - row0:].data_ptr(), depth=args.depth, spp=args.spp,
                                    seed=20260355, row0=r0, nrows=nr, stream=stream)
                st = scene.stats()
                for key in keys:
                    total[key] += st[key]
        stats = total
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
        assert primary == float(w) * h * args.spp, (primary, w, h, args.spp)   # every pixel of the frame was traced once per sample
        ms_step = dt / args.steps * 1e3
        per_test = ntri * FLOPS_TRI + nrect * FLOPS_RECT
        ms_shade = float(np.mean(shade_ms)) if shade_ms else float(shade_t[0])
        ms_prim = float(np.mean(prim_ms)) if prim_ms else float(shade_t[1])
        # per-launch algorithmic flops of the dominant kernel (k_primary*: one test per primary ray x leaf)
        prim_flops = primary / n * per_test
        achieved = prim_flops / (ms_prim * 1e-3) / 1e12
        mfma = args.filter == "mfma"
        # executed matrix-core work of that launch: 8 MFMAs (one per 32 rays of the wave's 256) per block of 32 leaves
        # the wave scanned (the library counts them); the brute-force scan would visit every block for every wave
        tiles = -(-w // 32) * -(-nrows // 8)
        blocks = -(-(ntri + nrect) // 32)
        peak = PEAK_MFMA_F16_TFLOPS if mfma else PEAK_FP32_TFLOPS

        def kernel_line(name, rays, ms, scanned, waves):
            return {"kernel": name + ("_mfma" if mfma else ""), "ms": ms,
                    "achieved": rays / n * per_test / (max(ms, 1e-9) * 1e-3) / 1e12,
                    "executed": float(scanned) * 8 * MFMA_FLOPS / (max(ms, 1e-9) * 1e-3) / 1e12,
                    "scanned_frac": float(scanned) / max(1.0, float(waves * blocks)) if n == 1 else None}
        kp = kernel_line("k_primary", primary, ms_prim, stats["mfma_blocks_primary"], tiles * args.spp)
        ks = kernel_line("k_shadow", shadow, ms_shade, stats["mfma_blocks_shadow"], shadow / 256.0)
        dom, oth = (kp, ks) if ms_prim >= ms_shade else (ks, kp)
        out = {
            "metric": "primary Mrays/s (whole node)", "value": primary / (dt / args.steps) / 1e6, "unit": "Mrays/s",
            "n_gpus": n, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "strong" if n > 1 else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{wname} ({ntri + nrect} leaves), {nlt} point light(s) + ambient, "
                                   f"{w}x{h}, depth {args.depth}, spp {args.spp}, every ray x leaf pair decided (no hierarchy)"
                                   + (f", {n} cost-balanced row bands gathered to GPU 0 over " + ("RCCL" if not staged else "gloo via host memory (rehearsal)") if n > 1 else ""),
                       "arithmetic": "hits and colours decided in f64 (the reference's Double); conservative filters in front: "
                                     + (("blocks of 32 leaves against the bound of a wave's 256 ray directions, " if not args.no_cull else "")
                                        + "f16 bounding cones on the matrix cores, then f32 records" if args.filter == "mfma" else "f32 records on the vector ALU"),
                       "width": w, "height": h, "leaves": ntri + nrect, "lights": nlt, "spp": args.spp,
                       "bands": bands if n > 1 else None, "pieces_per_band": pieces if n > 1 else None, "gathered_frame_equals_single_render": verified,
                       "shadow_rays_per_frame": shadow, "exact_fp64_evals_per_frame": exact,
                       "whole_frame_algorithmic_tflops": (primary + shadow) * per_test / (dt / args.steps) / 1e12,
                       "kernel_ms": {"k_primary": ms_prim, "k_setup": stats["ms_setup"], "k_shadow": ms_shade, "k_final": stats["ms_final"]}},
            "roofline": {"bound": "mfma" if mfma else "valu_fp32", "kernel": dom["kernel"],
                         "achieved": dom["achieved"], "peak": peak, "unit": "TFLOP/s", "frac": dom["achieved"] / peak,
                         "traffic": TRAFFIC_BYTES_DEFAULT.get(dom["kernel"]) if (n == 1 and args.spp == 1 and not args.no_cull and args.workload == "synthetic" and (w, h, ntri, nrect) == (3840, 2160, 10000, 6000)) else None,
                         "traffic_unit": "bytes per launch (rocprofv3 --pmc: 2 x FETCH_SIZE (gfx950 correction for 16-B-per-lane reads) + WRITE_SIZE, profiles/r01_cull_pmc_*)",
                         "launch_ms": dom["ms"],
                         "frac_of_fp32_valu_peak": dom["achieved"] / PEAK_FP32_TFLOPS,
                         "executed_mfma_tflops": dom["executed"] if mfma else None,
                         "executed_mfma_frac": dom["executed"] / PEAK_MFMA_F16_TFLOPS if mfma else None,
                         "leaf_blocks_scanned_frac": dom["scanned_frac"] if mfma else None,
                         "other_kernel": {"kernel": oth["kernel"], "launch_ms": oth["ms"], "achieved": oth["achieved"], "frac": oth["achieved"] / peak,
                                          "executed_mfma_tflops": oth["executed"] if mfma else None,
                                          "leaf_blocks_scanned_frac": oth["scanned_frac"] if mfma else None},
                         "note": "achieved = algorithmic flops (53/45 per ray x triangle/rectangle test, SURVEY.md 8d) x rays x leaves "
                                 "of one launch (primary rays for k_primary_mfma, the shadow rays cast for k_shadow_mfma) / its HIP-event time. The kernel decides every pair but executes far fewer flops: a wave "
                                 "skips the blocks of 32 leaves whose bounding cone misses the bound of its 256 rays "
                                 "(leaf_blocks_scanned_frac of the blocks are scanned), a scanned pair costs one FP16 MFMA form, and only "
                                 "candidates reach FP32/FP64 -- so frac and frac_of_fp32_valu_peak exceed 1 and say how much cheaper than "
                                 "the reference formulation the decision is, not how busy the matrix pipe is; executed_mfma_* is the "
                                 "matrix-core work actually issued against the dense FP16 peak, and brute_force_scan is the same frame "
                                 "with the block test off (every wave scans every block)"},
        }
        if n == 1 and mfma and not args.no_cull and not args.no_brute_force_leg:
            # the same frame with the block test off (every wave scans every block), timed the same way
            os.environ["RT_CULL"] = "0"
            _lib.check(_lib.lib().rt_init(local_rank), "rt_init")
            step(); sync()
            t0 = time.perf_counter()
            for _ in range(2):
                step()
            sync()
            bf = (time.perf_counter() - t0) / 2
            bst = scene.stats()
            out["brute_force_scan"] = {"value": primary / bf / 1e6, "unit": "Mrays/s", "ms_per_step": bf * 1e3,
                                       "k_primary_ms": bst["ms_primary"], "k_shadow_ms": bst["ms_shadow"],
                                       "executed_mfma_tflops": float(bst["mfma_blocks_primary"]) * 8 * MFMA_FLOPS / (bst["ms_primary"] * 1e-3) / 1e12}
            os.environ["RT_CULL"] = "1"
            _lib.check(_lib.lib().rt_init(local_rank), "rt_init")
        if n == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(world, lights, w, h, args.depth)
        print(json.dumps(out), flush=True)
    scene.close()
    if n > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
