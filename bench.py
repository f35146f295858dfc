#!/usr/bin/env python3
"""bench.py — primary Mrays/s of the MI355X render path (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

A "step" renders one frame of the synthetic scene of SURVEY.md §8(d) (10 000 triangles + 1 000
cuboids = 16 000 leaves, 4 point lights with shadow rays, every ray x leaf pair decided, no hierarchy):
  N = 1 : 3840 x 2160 on one GPU                                   (BASELINE.json configs[2])
  N > 1 : 16384 x 16384 split into N contiguous row bands of equal MEASURED cost, one per rank; the
          bands are sent to rank 0 over RCCL (xGMI) inside the timed region   (BASELINE.json configs[3])
The scene and its filter records are resident in HBM before the timed region (`scene_prepare_ms` says what that
cost); the frame buffer stays in HBM.  value = primary rays of the whole frame / max-over-ranks wall time.

Because N = 1 and N > 1 render different frames, a scaling curve must not be read off `value` alone: the N = 1 line
carries `same_frame_as_multi_gpu` (the 16384 x 16384 frame on this one GPU, same code path) and every N > 1 line carries
`n1_same_frame_ms`, measured in the same process on rank 0, and `speedup_vs_n1_same_frame`.

Also on the JSON line:
  roofline      — the dominant kernel (the longer of k_primary_mfma / k_shadow_mfma) as a UTILISATION (<= 1): the larger
                  of its vector-issue fraction (SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x kernel cycles)) and its
                  matrix-pipe fraction (MFMAs x 32768 flop / time / 2.5 PFLOP/s).  Instruction counts per launch, the
                  clock and the HBM traffic are PMC readings: they come from profiles/r03_pmc.json (written by
                  tools/profile_round.sh from separate rocprofv3 --pmc passes of this very command) and are combined
                  with the launch time measured live (HIP events on the render stream).  The MFMA count is also
                  counted live by the kernels.  The ALGORITHMIC rate (52/44 flops per ray x triangle/rectangle test,
                  SURVEY.md §8 d) is reported beside it as `algorithmic_tflops` / `speedup_vs_all_pairs`.
  rt_render     — the same frame through rt_render (host buffers, the call a Haskell caller makes, SURVEY.md §8 d):
                  into page-locked and into pageable memory, PCIe included.  Never `value`.
  two_frames_in_flight — the same frames with consecutive ones on alternating lanes of the scene (two frame buffers): what a
                  caller who renders a sequence gets per frame.  Throughput, not the time of a frame: never `value`.
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
from raytrace_amd.distributed import gather_bands, piece_fractions, plan_frame, split_rows, sub_bands  # noqa: E402

FLOPS_TRI, FLOPS_RECT = 52 + 1, 44 + 1
PEAK_FP32_TFLOPS = 157.3                        # MI355X_MICROARCH.md: peak FP32 (vector), packed FMA
PEAK_MFMA_F16_TFLOPS = 2500.0                   # MI355X_MICROARCH.md: peak BF16/FP16 MFMA, dense
PEAK_CLOCK_GHZ = 2.4
SIMDS = 256 * 4                                 # 256 CUs x 4 SIMDs; one wave64 VALU instruction holds a SIMD's issue port 4 cycles
MFMA_FLOPS = 2 * 32 * 32 * 16                   # one v_mfma_f32_32x32x16_f16
PMC_JSON = os.path.join(ROOT, "profiles", "r03_pmc.json")
# Vector-issue cycles one wave64 instruction holds a SIMD's issue port, by the classes SQ_INSTS_VALU_* counts, measured on this
# chip at 4-8 waves per SIMD (profiles/r03_issue_costs.txt, tools/microbench/issue_costs.hip).  Plain FP32 add / mul / fma are
# the only instructions that reach the 2-cycle rate of MI355X_MICROARCH.md:485 (v_pk_fma_f32 costs 4.3 but these kernels' FP32
# work is unpacked v_fma_f32); an MFMA holds the port for 8 of its 32 cycles.
ISSUE_COST = {"fp32": 2.4, "fp64": 4.25, "trans_f32": 8.2, "trans_f64": 16.3, "int": 4.25, "cvt": 4.25, "rest": 4.25, "mfma_hold": 8.0}
CLASS_COUNTERS = {"fp32": ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32"),
                  "fp64": ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64"),
                  "trans_f32": ("SQ_INSTS_VALU_TRANS_F32",), "trans_f64": ("SQ_INSTS_VALU_TRANS_F64",),
                  "int": ("SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64"), "cvt": ("SQ_INSTS_VALU_CVT",)}
PEAK_FP64_TFLOPS = 78.6                         # MI355X_MICROARCH.md: peak FP64 (vector)
MULTI_W = MULTI_H = 16384                       # configs[3]


def host_threads():
    """Threads the CPU leg may use: the cores this process may run on (its affinity mask), capped by the box's own
    OMP_NUM_THREADS share when one is set (a GPU box hands a 1-GPU job 16 of the machine's cores and asks that worker pools
    stay within them); os.cpu_count() is reported beside it as cores_available."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    share = os.environ.get("OMP_NUM_THREADS", "")
    if share.isdigit() and int(share) > 0:
        n = min(n, int(share))
    return max(1, n)


def cpu_baseline(world, lights, w, h, depth, seconds=13.0, seconds_n4=6.0):
    """The CPU oracle on a bounded band of the same frame: on all the host cores this job may use (north_star: `+RTS
    -N<cores>`) and on 4 threads (README:36 suggests `+RTS -N4`).  Contiguous row chunks per thread, as Repa's computeP."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    cores = host_threads()
    osc = ob.OracleScene(world, lights)
    row0 = h // 2 - 1
    osc.render(w, h, depth=depth, row0=row0, nrows=1, nthreads=1)       # first call numbers the leaves
    t0 = time.perf_counter()
    osc.render(w, h, depth=depth, row0=row0, nrows=2, nthreads=2)       # calibration: 2 rows on 2 threads
    per_row_thread = (time.perf_counter() - t0)                          # seconds per row per thread

    def leg(threads, budget):
        nrows = int(max(threads, min(h // 2, threads * budget / max(per_row_thread, 1e-6))))
        nrows -= nrows % 2
        r0 = h // 2 - nrows // 2
        t0 = time.perf_counter()
        osc.render(w, h, depth=depth, row0=r0, nrows=nrows, nthreads=threads)
        dt = time.perf_counter() - t0
        return w * nrows / dt / 1e6, r0, nrows, dt
    v, r0, nrows, dt = leg(cores, seconds)
    v4, r04, nrows4, dt4 = leg(min(4, cores), seconds_n4)
    osc.close()
    return {"value": v, "unit": "Mrays/s", "cores": cores, "cores_available": os.cpu_count(), "kind": "port",
            "sample": f"rows {r0}..{r0 + nrows - 1} ({nrows} of {h}) of the same {w}x{h} frame, "
                      f"{dt:.1f} s; C restatement of the reference (oracle/), not the Haskell program",
            "n4": {"value": v4, "unit": "Mrays/s", "cores": min(4, cores),
                   "sample": f"rows {r04}..{r04 + nrows4 - 1} ({nrows4} of {h}), {dt4:.1f} s; the `+RTS -N4` of the reference's README:36"}}


def cold_start_leg(workload, w, h, depth):
    """examples/cold_start (C++, the C ABI only): every constructor call of the scene, rt_scene_prepare_device and the FIRST
    rt_render into page-locked host memory, in a process of its own — what a one-shot caller like the reference's main
    (Main.hs:569-621) pays.  Best of two runs; rt_init (the HIP runtime's own start-up) is reported beside it."""
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "examples", "cold_start")
    if not os.path.exists(exe) or workload not in ("synthetic", "cylinders"):
        return None
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import dump_scene
    world, lights = (scenes.synthetic if workload == "synthetic" else scenes.synthetic_cylinders)()
    runs = []
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "scene.txt")
        dump_scene.dump(world, lights, path)
        for _ in range(2):
            try:
                r = subprocess.run([exe, path, str(w), str(h), str(depth)], capture_output=True, text=True, timeout=120)
                if r.returncode == 0:
                    runs.append(json.loads(r.stdout.strip().splitlines()[-1]))
            except (subprocess.SubprocessError, ValueError, OSError):
                pass
    if not runs:
        return None
    best = min(runs, key=lambda d: d["cold_start_ms"])
    best["note"] = ("cold_start_ms = rt_scene_create + every rt_add_* call (scene_build_ms) + rt_scene_prepare_device (host index build + one "
                    "upload) + the first rt_render into page-locked host buffers, wall clock of examples/cold_start in a process of its own, "
                    "best of 2; rt_init_ms is the HIP runtime's start-up, not counted; the very first process on a fresh machine pays "
                    "~150 ms more inside the runtime's first stream creation (profiles/r03_cold_start.txt)")
    return best


def load_pmc(workload_key):
    """PMC readings per launch of the two big kernels for the default workload (profiles/r03_pmc.json)."""
    try:
        with open(PMC_JSON) as f:
            pmc = json.load(f)
    except (OSError, ValueError):
        return None
    return pmc if pmc.get("workload_key") == workload_key else None


def utilisation(kernel, ms, mfmas_live, pmc):
    """Roofline of one launch as a utilisation: vector-issue fraction and matrix-pipe fraction (both <= 1)."""
    k = (pmc or {}).get("kernels", {}).get(kernel, {})
    clock = k.get("clock_ghz") or PEAK_CLOCK_GHZ
    cycles = ms * 1e-3 * clock * 1e9
    out = {"kernel": kernel, "launch_ms": ms, "clock_ghz": clock,
           "clock_source": "GRBM_GUI_ACTIVE / 8 / duration of the profiled launch" if k.get("clock_ghz") else "peak clock (no PMC file for this workload)"}
    mfmas = k.get("SQ_INSTS_MFMA") or mfmas_live
    out["mfma_per_launch"] = mfmas
    out["mfma_tflops"] = mfmas * MFMA_FLOPS / (ms * 1e-3) / 1e12
    out["mfma_frac"] = out["mfma_tflops"] / PEAK_MFMA_F16_TFLOPS
    if k.get("SQ_INSTS_VALU"):
        out["valu_insts_per_launch"] = k["SQ_INSTS_VALU"]
        out["valu_issue_frac_flat4"] = k["SQ_INSTS_VALU"] * 4 / (SIMDS * cycles)        # round 2's figure: every instruction at 4 cycles
        if all(c in k for cs in CLASS_COUNTERS.values() for c in cs):
            # class-weighted: every class at its measured issue cost, what no class counter covers (moves, compares, ballots,
            # mbcnt, readlane, ...) at 4.25, plus the 8 cycles an MFMA holds the vector issue port
            counts = {cls: sum(k[c] for c in cs) for cls, cs in CLASS_COUNTERS.items()}
            counts["rest"] = max(0.0, k["SQ_INSTS_VALU"] - sum(counts.values()))
            issue = sum(counts[cls] * ISSUE_COST[cls] for cls in counts) + mfmas * ISSUE_COST["mfma_hold"]
            out["valu_issue_frac"] = issue / (SIMDS * cycles)
            out["issue_cycles_by_class"] = {cls: counts[cls] * ISSUE_COST[cls] / (SIMDS * cycles) for cls in counts}
            out["issue_cycles_by_class"]["mfma_hold"] = mfmas * ISSUE_COST["mfma_hold"] / (SIMDS * cycles)
            out["insts_by_class"] = counts
            sec = ms * 1e-3
            fp32 = (k["SQ_INSTS_VALU_ADD_F32"] + k["SQ_INSTS_VALU_MUL_F32"] + 2 * k["SQ_INSTS_VALU_FMA_F32"]) * 64
            fp64 = (k["SQ_INSTS_VALU_ADD_F64"] + k["SQ_INSTS_VALU_MUL_F64"] + 2 * k["SQ_INSTS_VALU_FMA_F64"]) * 64
            out["executed_fp32_tflops"] = fp32 / sec / 1e12
            out["executed_fp64_tflops"] = fp64 / sec / 1e12
            out["fp32_peak_frac"] = out["executed_fp32_tflops"] / PEAK_FP32_TFLOPS
            out["fp64_peak_frac"] = out["executed_fp64_tflops"] / PEAK_FP64_TFLOPS
            out["int_share"] = counts["int"] / k["SQ_INSTS_VALU"]
        else:
            out["valu_issue_frac"] = out["valu_issue_frac_flat4"]
        if k.get("SQ_WAVE_CYCLES") and k.get("SQ_ACTIVE_INST_VALU") is not None and k.get("SQ_WAIT_INST_ANY") is not None:
            # a wave's life (quad-cycles, summed over waves): issuing vector instructions / stalled on a dependency / parked at a wait
            out["wave_life"] = {"issuing_valu": k["SQ_ACTIVE_INST_VALU"] / k["SQ_WAVE_CYCLES"], "stalled_on_dependency": k["SQ_WAIT_INST_ANY"] / k["SQ_WAVE_CYCLES"],
                                "parked_at_a_wait_or_not_picked": 1.0 - (k["SQ_ACTIVE_INST_VALU"] + k["SQ_WAIT_INST_ANY"]) / k["SQ_WAVE_CYCLES"]}
    else:
        out["valu_insts_per_launch"] = None
        out["valu_issue_frac"] = None
    if k.get("FETCH_SIZE_KiB") is not None and k.get("WRITE_SIZE_KiB") is not None:
        # MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE tallies wide (16 B per lane) reads at half their bytes -> x 2
        # (an upper bound here: the kernels' reads are 16-byte job / record / leaf-row loads and 8-byte FP64 loads);
        # WRITE_SIZE as reported.  Separate --pmc passes.
        out["traffic"] = int((2 * k["FETCH_SIZE_KiB"] + k["WRITE_SIZE_KiB"]) * 1024)
    else:
        out["traffic"] = None
    return out


PREWARM = 20
STATS_EVERY = 10   # N = 1: per-kernel times are sampled on every 10th timed step (reading them synchronises)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--ntri", type=int, default=10000)
    ap.add_argument("--ncuboid", type=int, default=1000)
    ap.add_argument("--depth", type=int, default=4)
    ap.add_argument("--workload", default="synthetic", choices=["synthetic", "cylinders", "default", "mirror"],
                    help="synthetic = configs[2]/[3] (the metric's workload); cylinders = configs[4] (specular cylinders + cuboids, "
                         "use with --spp 16 --width 7680 --height 4320); default = the Main.hs scene (configs[0]/[1]); "
                         "mirror = the `spheres` scene of Main.hs:670-690 (mirror materials: reflected rays, trace depth 4)")
    ap.add_argument("--spp", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--filter", default="mfma", choices=["mfma", "valu"],
                    help="first stage: bounding cones on the matrix cores (the product) or the FP32 vector-ALU kernels of the cross-check build")
    ap.add_argument("--no-extra-legs", action="store_true", help="profiling runs: only the timed steps (no brute-force, rt_render or 16K legs)")
    ap.add_argument("--no-cull", action="store_true", help="every wave scans every block of the leaf stream: the plain all-pairs scan")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, the measured path) or gloo (rehearsal on one GPU)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank renders on GPU 0 (gloo only)")
    ap.add_argument("--pieces", type=int, default=0, help="N > 1: pieces per band (0 = what the pipeline model picks: raytrace_amd/distributed.py plan_frame)")
    ap.add_argument("--link-gbps", type=float, default=48.0, help="N > 1: what the model assumes one xGMI link carries in one direction under ncclSend/ncclRecv (GB/s)")
    ap.add_argument("--piece-overhead-ms", type=float, default=0.02, help="N > 1: what the model charges per piece for its launches and kernel tails "
                    "(measured on one MI355X for a 2048-row band of the 16384-wide frame, profiles/r03_piece_overhead.txt: 0.06-0.14 ms per extra piece on one "
                    "lane; -0.08 to +0.01 ms with the pieces alternating between the scene's two lanes, which is how they are rendered)")
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
    L = _lib.valu_lib() if args.filter == "valu" else _lib.lib()
    _lib.check(L.rt_init(local_rank), "rt_init", L)

    w = args.width or (3840 if n == 1 else MULTI_W)
    h = args.height or (2160 if n == 1 else MULTI_H)
    if args.workload == "synthetic":
        world, lights = scenes.synthetic(ntri=args.ntri, ncuboid=args.ncuboid)
        ntri, nrect, nlt = args.ntri, 6 * args.ncuboid, 4
        wname = f"synthetic {ntri} triangles + {args.ncuboid} cuboids"
    elif args.workload == "cylinders":
        world, lights = scenes.synthetic_cylinders()
        ntri, nrect, nlt = 200 * 32, 200 * 16 + 500 * 6, 4
        wname = "200 upright cylinders (n=16) + 500 cuboids, specularity 100/400"
    elif args.workload == "mirror":
        world, lights = scenes.mirror_bench()
        ntri, nrect, nlt = None, None, 2
        wname = "mirror scene (Main.hs:670-690 `spheres` + 2 000 triangles, reflectivity 0.9 floor strip and sphere)"
    else:
        world, lights = scenes.default_scene()
        ntri, nrect, nlt = 200, 1, 1
        wname = "Main.hs default scene (gaussian height field, linearInterpolation)"
    scene = api.Scene(world, lights, lib=L)
    # N > 1: the bands travel between GPUs, so the frame is 3 bytes per pixel there (RT_OPT_RGB24: the reference's PixelRGB8,
    # what its BMP writer takes); N = 1 keeps the RGBA8 frame of the earlier rounds (`same_frame_as_multi_gpu` has both)
    bpp = 3 if world_size > 1 else 4
    if bpp == 3:
        scene.set_option(_lib.RT_OPT_RGB24, 1)
    if args.filter == "valu":
        scene.set_option(_lib.RT_OPT_FILTER, 1)
    if args.no_cull:
        scene.set_option(_lib.RT_OPT_CULL, 0)
    if ntri is None:
        kinds = [scene.leaf(i)[0] for i in range(scene.num_leaves)]
        ntri, nrect = kinds.count(1.0), kinds.count(0.0)
    nleaves = scene.num_leaves
    t0 = time.perf_counter()
    scene.prepare(w, h)                                           # host index build + upload, outside the timed region
    prepare_wall_ms = (time.perf_counter() - t0) * 1e3

    # ---- bands: rank 0 measures the cost of 32 strips of the frame (HIP events, reduced scale) and everybody cuts the
    # same bands from it
    n1_same_frame_ms = None
    if n == 1:
        bands = [(0, h)]
    else:
        info = torch.zeros(33, dtype=torch.float64)
        if rank == 0:
            info[:32] = torch.tensor(scene.row_costs(w, h, 32, depth=args.depth, spp=args.spp), dtype=torch.float64)
            # the same frame on this GPU alone, same code path: the honest N = 1 point of the scaling curve
            full = torch.empty((h, w, bpp), dtype=torch.uint8, device=dev)
            fhit = torch.empty((h, w), dtype=torch.int32, device=dev)
            for k in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                scene.render_device(w, h, full.data_ptr(), fhit.data_ptr(), depth=args.depth, spp=args.spp, seed=20260355)
                _lib.check(L.rt_synchronize(), "rt_synchronize", L)
                info[32] = (time.perf_counter() - t0) * 1e3
            del full, fhit
            torch.cuda.empty_cache()
        if staged:
            dist.broadcast(info, 0)
        else:
            info = info.to(dev)
            dist.broadcast(info, 0)
            info = info.cpu()
        n1_same_frame_ms = float(info[32])
        # the frame on paper first (every rank computes the same plan): pieces per band and their sizes, rank 0's larger share
        plan_parts, plan_fr, plan_weights, predicted_ms, plan_detail = plan_frame(
            n1_same_frame_ms, n, float(w) * h * bpp / n, args.link_gbps, args.piece_overhead_ms)
        if args.pieces > 0:
            plan_parts, plan_fr = args.pieces, piece_fractions(args.pieces)
        bands = split_rows(info[:32].numpy(), h, n, weights=plan_weights)
    row0, nrows = bands[rank]
    # frame buffer: rank 0 owns the whole image (bands of the other ranks land in it), others own a band
    frame = torch.empty((h if rank == 0 else nrows, w, bpp), dtype=torch.uint8, device=dev)
    mine = frame[row0:row0 + nrows] if rank == 0 else frame
    hit = torch.empty((nrows, w), dtype=torch.int32, device=dev)
    # The renders go to a torch stream that is ALSO torch's current stream while the gather is enqueued: RCCL orders its
    # own stream behind the current stream, so piece k's send waits for piece k's render and for nothing later.  (With the
    # library's private stream, or the legacy default stream, the send would not be ordered behind the render at all.)
    rs = torch.cuda.Stream(dev)
    # N > 1: the pieces of a band alternate between the scene's two lanes (RT_OPT_LANE), each lane on a stream of its own:
    # piece k + 1's kernels fill the tails of piece k's, which takes the cost of cutting a band into pieces to ~0
    # (profiles/r03_piece_overhead.txt).  Piece k's send is enqueued with ITS lane's stream current.
    lanes = [rs, torch.cuda.Stream(dev)]

    # every band is rendered in `pieces` consecutive pieces; piece k of all peers travels to GPU 0 in one group
    # of send/recv while piece k+1 renders, so only the last piece's exchange is exposed.  A piece should take about
    # a millisecond: 6 kernel launches and their tails are then a few per cent of it.
    pieces = plan_parts if n > 1 else 1
    subs = [sub_bands(b, pieces, fractions=plan_fr if n > 1 else None) for b in bands]
    # (the three events of a step are made once and recorded again every step: the last step's are the ones read)
    marks = {k: torch.cuda.Event(enable_timing=True) for k in ("start", "rendered", "rendered_b", "end")}
    lane_done = torch.cuda.Event()
    frame_ptr, hit_ptr = mine.data_ptr(), hit.data_ptr()

    def step(seed=20260355):
        if n == 1:      # one GPU: nothing but the render call in the timed loop (it goes to the stream it is given)
            scene.render_device(w, h, frame_ptr, hit_ptr, depth=args.depth, spp=args.spp, seed=seed, row0=row0, nrows=nrows,
                                stream=rs.cuda_stream)
            return
        pending, host = [], []
        host_frame = torch.empty((h, w, bpp), dtype=torch.uint8) if staged and rank == 0 else None
        rs.record_event(marks["start"])
        lanes[1].wait_event(marks["start"])            # (the previous step ended with lane 1 joined to lane 0)
        for k in range(pieces):
            ln = k & 1
            with torch.cuda.stream(lanes[ln]):
                r0, nr = subs[rank][k]
                piece = mine[r0 - row0:r0 - row0 + nr]
                if nr > 0:
                    scene.render_device(w, h, piece.data_ptr(), hit[r0 - row0:].data_ptr(), depth=args.depth, spp=args.spp,
                                        seed=seed, row0=r0, nrows=nr, stream=lanes[ln].cuda_stream, lane=ln, cont=k > 0)
                if k >= pieces - 2:    # the last piece of each lane
                    lanes[ln].record_event(marks["rendered" if k == pieces - 1 else "rendered_b"])
                group = [s[k] for s in subs]
                if not staged:     # ncclSend/ncclRecv in one group: every peer has a direct xGMI link to GPU 0
                    pending += gather_bands(dist, frame, piece, group, rank, wait=False)
                else:              # gloo rehearsal: the same groups, through host memory
                    lanes[ln].synchronize()
                    host_piece = piece.cpu()
                    host.append((r0, nr, host_piece))
                    pending += gather_bands(dist, host_frame, host_piece, group, rank, wait=False)
        with torch.cuda.stream(rs):
            lanes[1].record_event(lane_done)
            rs.wait_event(lane_done)
            for req in pending:
                req.wait()
            if staged and rank == 0:
                for r0, nr, host_piece in host:
                    host_frame[r0:r0 + nr] = host_piece
                frame.copy_(host_frame)
            rs.record_event(marks["end"])

    def sync():
        torch.cuda.synchronize()
        if n > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # The chip comes out of idle at a low clock and the order of the lights in k_shadow_mfma settles after a few frames:
    # PREWARM untimed frames before the W warm-up steps of the contract (reported in the line as `untimed_frames_before_warmup`).
    # what a caller sees who renders a few frames and stops: ONE warm-up frame, then 5 timed ones, before anything else has run
    step()
    sync()
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    sync()
    ms_after_one_warmup = (time.perf_counter() - t0) / 5 * 1e3
    for _ in range(max(0, PREWARM - 6)):
        step()
    for _ in range(args.warmup):
        step()
    sync()
    shade_ms, prim_ms, stats = [], [], None
    t0 = time.perf_counter()
    # The K steps are enqueued back to back, as a caller renders a sequence of frames; the per-kernel HIP-event times (events
    # the library records on the render stream) are read on every STATS_EVERY-th step and on the last one — reading them
    # waits for the frame, and a wait after every frame costs 0.035 ms of launch latency per frame (1.291 against 1.255 ms).
    for k in range(args.steps):
        step()
        if n == 1 and (k % STATS_EVERY == STATS_EVERY - 1 or k == args.steps - 1):
            stats = scene.stats()
            shade_ms.append(stats["ms_shadow"]); prim_ms.append(stats["ms_primary"])
    sync()
    dt = time.perf_counter() - t0
    # the last step of this rank: render of its band, and what followed until the frame was complete (exposed exchange)
    rank_render_ms = dt / args.steps * 1e3
    if n > 1:
        rank_render_ms = marks["start"].elapsed_time(marks["rendered"])
        if pieces > 1:
            rank_render_ms = max(rank_render_ms, marks["start"].elapsed_time(marks["rendered_b"]))
    rank_total_ms = marks["start"].elapsed_time(marks["end"]) if n > 1 else dt / args.steps * 1e3
    if n > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        if staged:
            tmax = tmax.cpu()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    stats = scene.stats()
    if n > 1:            # untimed pass: ray counts and per-kernel times of this rank's band, summed over its pieces
        keys = ("primary_rays", "shadow_rays", "exact_evals", "ms_shadow", "ms_primary", "ms_setup", "ms_final",
                "mfma_blocks_primary", "mfma_blocks_shadow", "record_tests")
        total = dict.fromkeys(keys, 0.0)
        for r0, nr in subs[rank]:
            if nr > 0:
                scene.render_device(w, h, mine[r0 - row0:].data_ptr(), hit[r0 - row0:].data_ptr(), depth=args.depth, spp=args.spp,
                                    seed=20260355, row0=r0, nrows=nr, stream=rs.cuda_stream, lane=0, cont=False)
                st = scene.stats()
                for key in keys:
                    total[key] += st[key]
        stats.update(total)
    counts = torch.tensor([stats["primary_rays"], stats["shadow_rays"], stats["exact_evals"]], dtype=torch.float64, device=dev)
    shade_t = torch.tensor([stats["ms_shadow"], stats["ms_primary"]], dtype=torch.float64, device=dev)
    per_rank = torch.zeros((n, 2), dtype=torch.float64, device=dev)
    per_rank[rank, 0], per_rank[rank, 1] = rank_render_ms, rank_total_ms
    if n > 1:
        if staged:
            counts, shade_t, per_rank = counts.cpu(), shade_t.cpu(), per_rank.cpu()
        dist.all_reduce(counts)
        dist.all_reduce(shade_t, op=dist.ReduceOp.MAX)
        dist.all_reduce(per_rank)
    verified = None
    if args.verify:
        # a frame nobody has seen yet (another seed has no effect at spp 1, so the buffers are poisoned instead): stale
        # or unordered data in the gathered frame cannot pass
        frame.fill_(0x5a)
        sync()
        step()
        sync()
        if rank == 0:
            alone = torch.empty((h, w, bpp), dtype=torch.uint8, device=dev)
            scene.render_device(w, h, alone.data_ptr(), None, depth=args.depth, spp=args.spp, seed=20260355, stream=rs.cuda_stream,
                                lane=0, cont=False)
            torch.cuda.synchronize()
            verified = bool(torch.equal(alone, frame))
            del alone
    if rank == 0:
        primary, shadow, exact = (float(c) for c in counts.tolist())
        assert primary == float(w) * h * args.spp, (primary, w, h, args.spp)   # every pixel of the frame was traced once per sample
        ms_step = dt / args.steps * 1e3
        per_test = ntri * FLOPS_TRI + nrect * FLOPS_RECT
        ms_shade = float(np.mean(shade_ms)) if shade_ms else float(shade_t[0])
        ms_prim = float(np.mean(prim_ms)) if prim_ms else float(shade_t[1])
        mfma = args.filter == "mfma"
        default_workload = (n == 1 and args.spp == 1 and not args.no_cull and mfma and args.workload == "synthetic"
                            and (w, h, ntri, nrect, args.depth) == (3840, 2160, 10000, 6000, 4))
        pmc = load_pmc("synthetic-10000-1000-3840x2160-d4-spp1") if default_workload else None
        tiles = -(-w // 32) * -(-nrows // 8)
        blocks = -(-(ntri + nrect) // 32)

        def kernel_line(name, rays, ms, scanned, waves):
            u = utilisation(name + ("_mfma" if mfma else ""), ms, float(scanned) * 8, pmc) if n == 1 else {"kernel": name, "launch_ms": ms}
            u["algorithmic_tflops"] = rays / n * per_test / (max(ms, 1e-9) * 1e-3) / 1e12
            u["leaf_blocks_scanned_frac"] = float(scanned) / max(1.0, float(waves * blocks)) if (n == 1 and mfma) else None
            return u
        kp = kernel_line("k_primary", primary, ms_prim, stats["mfma_blocks_primary"], tiles * args.spp)
        ks = kernel_line("k_shadow", shadow, ms_shade, stats["mfma_blocks_shadow"], shadow / 256.0)
        dom, oth = (kp, ks) if ms_prim >= ms_shade else (ks, kp)
        roof = {"kernel": dom["kernel"], "launch_ms": dom["launch_ms"]}
        if n == 1 and mfma:
            vf, mf = dom.get("valu_issue_frac"), dom["mfma_frac"]
            if vf is not None and vf >= mf:
                clock = dom["clock_ghz"]
                roof.update({"bound": "valu_issue", "achieved": vf * SIMDS * clock, "peak": SIMDS * clock,
                             "unit": "G vector-issue cycles/s (1024 SIMDs x clock; a wave64 instruction holds a SIMD's port for its class's cycles)", "frac": vf})
            else:
                roof.update({"bound": "mfma", "achieved": dom["mfma_tflops"], "peak": PEAK_MFMA_F16_TFLOPS, "unit": "TFLOP/s", "frac": mf})
            roof.update({
                "traffic": dom["traffic"],
                "traffic_unit": "HBM bytes per launch: (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 from separate rocprofv3 --pmc passes "
                                "(profiles/r03_pmc.json; x 2 = the gfx950 correction for wide reads, MI355X_MICROARCH.md), "
                                "null when no PMC file matches this workload",
                "valu_issue_frac": vf, "valu_issue_frac_flat4": dom.get("valu_issue_frac_flat4"), "issue_cost_cycles": ISSUE_COST,
                "issue_cost_source": "profiles/r03_issue_costs.txt (tools/microbench/issue_costs.hip, this chip, 4-8 waves per SIMD)",
                "issue_cycles_by_class": dom.get("issue_cycles_by_class"), "insts_by_class": dom.get("insts_by_class"),
                "executed_fp32_tflops": dom.get("executed_fp32_tflops"), "executed_fp64_tflops": dom.get("executed_fp64_tflops"),
                "fp32_peak_frac": dom.get("fp32_peak_frac"), "fp64_peak_frac": dom.get("fp64_peak_frac"), "int_share": dom.get("int_share"),
                "wave_life": dom.get("wave_life"),
                "mfma_frac": mf, "mfma_tflops": dom["mfma_tflops"], "clock_ghz": dom["clock_ghz"],
                "clock_source": dom["clock_source"],
                "pmc_source": (pmc or {}).get("source"), "pmc_command": (pmc or {}).get("command"),
                "algorithmic_tflops": dom["algorithmic_tflops"],
                "speedup_vs_all_pairs": dom["algorithmic_tflops"] / PEAK_FP32_TFLOPS,
                "leaf_blocks_scanned_frac": dom["leaf_blocks_scanned_frac"],
                "other_kernel": {k: oth.get(k) for k in ("kernel", "launch_ms", "valu_issue_frac", "valu_issue_frac_flat4", "issue_cycles_by_class", "executed_fp32_tflops",
                                                         "executed_fp64_tflops", "fp32_peak_frac", "fp64_peak_frac", "int_share", "wave_life", "mfma_frac", "mfma_tflops",
                                                         "traffic", "algorithmic_tflops", "leaf_blocks_scanned_frac")},
                "note": "frac = the larger of the kernel's vector-issue fraction (sum over instruction classes of SQ_INSTS_VALU_<class> x the class's "
                        "measured issue cycles, + 8 cycles per MFMA, / (1024 SIMDs x kernel cycles)) "
                        "and matrix-pipe fraction (MFMAs x 32768 flop / launch time / 2.5 PFLOP/s dense FP16): how busy the issue ports "
                        "are, not how much of the reference's arithmetic was avoided.  executed_fp32_tflops / fp32_peak_frac = the FP32 add, mul and "
                        "fma instructions the kernel really executed (x 64 lanes, fma = 2) against the 157.3 TFLOP/s vector peak north_star names: "
                        "a few per cent, because the pair tests run on the matrix cores and the rest of the issue slots go to integer bookkeeping "
                        "and FP64.  algorithmic_tflops = 53/45 flops per ray x "
                        "triangle/rectangle pair (SURVEY.md 8d) x the pairs the launch decides / launch time; speedup_vs_all_pairs = that "
                        "over the 157.3 TFLOP/s FP32 vector peak a literal all-pairs kernel could not exceed."})
        elif n == 1:
            roof.update({"bound": "valu_fp32", "achieved": dom["algorithmic_tflops"], "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                         "frac": None, "traffic": None,
                         "note": "cross-check build (FP32 vector-ALU kernels): algorithmic rate only; the executed rate is in profiles/README.md"})
        else:
            roof.update({"bound": "valu_issue", "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None,
                         "note": "per-kernel utilisation is reported by the N = 1 line; here launch_ms is the slowest rank's sum over its pieces"})
        out = {
            "metric": "primary Mrays/s (whole node)", "value": primary / (dt / args.steps) / 1e6, "unit": "Mrays/s",
            "n_gpus": n, "steps": args.steps, "warmup": args.warmup, "untimed_frames_before_warmup": PREWARM, "kernel_times_sampled_every": STATS_EVERY, "ms_per_step": ms_step,
            "ms_per_step_after_one_warmup": ms_after_one_warmup,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{wname} ({nleaves} leaves), {nlt} point light(s) + ambient, "
                                   f"{w}x{h}, depth {args.depth}, spp {args.spp}, every ray x leaf pair decided (no hierarchy)"
                                   + (f", {n} row bands of equal measured cost gathered to GPU 0 over " + ("RCCL" if not staged else "gloo via host memory (rehearsal)") if n > 1 else ""),
                       "arithmetic": "hits and colours decided in f64 (the reference's Double); conservative filters in front: "
                                     + (("blocks of 32 leaves against the bound of a wave's 256 ray directions, " if not args.no_cull else "")
                                        + "f16 bounding cones on the matrix cores, then f32 records" if mfma else "f32 records on the vector ALU"),
                       "width": w, "height": h, "leaves": nleaves, "lights": nlt, "spp": args.spp,
                       "shadow_rays_per_frame": shadow, "exact_fp64_evals_per_frame": exact,
                       "record_tests_per_frame": stats.get("record_tests"), "bounce_rays_per_frame": stats.get("bounce_rays"),
                       "kernel_ms": {"k_primary": ms_prim, "k_setup": stats["ms_setup"], "k_shadow": ms_shade, "k_final": stats["ms_final"]}},
            "roofline": roof,
            "scene_prepare_ms": {"host_index_build": stats["prepare_ms"], "upload": stats["upload_ms"], "wall_of_rt_scene_prepare_device": prepare_wall_ms,
                                 "note": "guard chains, bounding cones, direction sort and filter records for 5 apexes, then the upload: once per scene, outside the timed region"},
        }
        if n > 1:
            out["distributed"] = {
                "backend": "RCCL (torch.distributed nccl)" if not staged else "gloo (rehearsal, bands staged through host memory)",
                "world_size": dist.get_world_size(), "bands": bands, "pieces_per_band": pieces, "piece_fractions": plan_fr,
                "model": {"predicted_step_ms": predicted_ms, "predicted_speedup_vs_n1_same_frame": n1_same_frame_ms / predicted_ms,
                          "link_gbps_assumed": args.link_gbps, "piece_overhead_ms_assumed": args.piece_overhead_ms,
                          "band_weights": plan_weights, **plan_detail,
                          "note": "raytrace_amd/distributed.py plan_frame: a sender renders its pieces back to back and its own xGMI link to GPU 0 carries "
                                  "piece i as soon as it is rendered and piece i - 1 has arrived; rank 0 sends nothing and renders a larger band"},
                "rank_render_ms": [float(v) for v in per_rank[:, 0].tolist()],
                "rank_step_ms": [float(v) for v in per_rank[:, 1].tolist()],
                "exposed_gather_ms": float(per_rank[0, 1] - per_rank[:, 0].max()),
                "n1_same_frame_ms": n1_same_frame_ms,
                "speedup_vs_n1_same_frame": n1_same_frame_ms / ms_step,
                "gathered_frame_equals_single_render": verified,
                "note": "rank_render_ms / rank_step_ms: last timed step, events on each rank's render stream; exposed_gather_ms = rank 0's "
                        "step minus the slowest rank's render; n1_same_frame_ms = this very frame rendered by rank 0 alone (same process, "
                        "same code path) before the timed region"}
        elif verified is not None:
            out["config"]["frame_equals_second_render"] = verified
        if n == 1 and mfma and not args.no_extra_legs:
            if not args.no_cull:
                # the same frame with the block test off (every wave scans every block), timed the same way
                scene.set_option(_lib.RT_OPT_CULL, 0)
                step(); sync()
                t0 = time.perf_counter()
                for _ in range(2):
                    step()
                sync()
                bf = (time.perf_counter() - t0) / 2
                bst = scene.stats()
                out["brute_force_scan"] = {"value": primary / bf / 1e6, "unit": "Mrays/s", "ms_per_step": bf * 1e3,
                                           "k_primary_ms": bst["ms_primary"], "k_shadow_ms": bst["ms_shadow"],
                                           "mfma_tflops": float(bst["mfma_blocks_primary"]) * 8 * MFMA_FLOPS / (bst["ms_primary"] * 1e-3) / 1e12,
                                           "mfma_frac": float(bst["mfma_blocks_primary"]) * 8 * MFMA_FLOPS / (bst["ms_primary"] * 1e-3) / 1e12 / PEAK_MFMA_F16_TFLOPS}
                scene.set_option(_lib.RT_OPT_CULL, 1)
            # a SEQUENCE of independent frames with two in flight: consecutive frames alternate between the scene's two lanes
            # (RT_OPT_LANE: own work buffers, wave history and stream each; two frame buffers), so the kernels of frame k + 1
            # fill the tails of frame k's.  Not `value`: a step there is a whole frame that has ended before the next begins.
            if args.spp == 1:
                frame_b, hit_b = torch.empty_like(frame), torch.empty_like(hit)
                fl_streams = [rs, torch.cuda.Stream(dev)]
                fl_out = [(frame_ptr, hit_ptr), (frame_b.data_ptr(), hit_b.data_ptr())]
                nfl = max(20, args.steps)
                for timed_pass in (False, True):
                    sync()
                    t0 = time.perf_counter()
                    for k in range(nfl):
                        scene.render_device(w, h, fl_out[k & 1][0], fl_out[k & 1][1], depth=args.depth, spp=1, seed=20260355,
                                            stream=fl_streams[k & 1].cuda_stream, lane=k & 1, cont=False)
                    sync()
                    fl_ms = (time.perf_counter() - t0) / nfl * 1e3
                out["two_frames_in_flight"] = {"ms_per_frame": fl_ms, "mrays": primary / fl_ms / 1e3, "frames": nfl,
                                               "frames_equal": bool(torch.equal(frame_b, frame) and torch.equal(hit_b, hit)),
                                               "note": "the same frames, consecutive ones alternating between the scene's two lanes (two frame buffers): "
                                                       "throughput of a frame sequence, not the time of a frame; never `value`"}
                scene.set_option(_lib.RT_OPT_LANE, -1).set_option(_lib.RT_OPT_CONTINUE, 0)
                del frame_b, hit_b
            # the call SURVEY.md 8(d) defines the metric on: rt_render into HOST buffers (PCIe included)
            legs = {}
            dests = {"page_locked": (api.host_buffer((h, w, 4), np.uint8), api.host_buffer((h, w), np.int32)),
                     "pageable": (np.zeros((h, w, 4), np.uint8), np.zeros((h, w), np.int32))}
            for name, (rgba_h, hit_h) in dests.items():
                ts = []
                for _ in range(6):
                    t0 = time.perf_counter()
                    scene.render(w, h, depth=args.depth, spp=args.spp, seed=20260355, out_rgba=rgba_h, out_hit=hit_h)
                    ts.append(time.perf_counter() - t0)
                legs[name] = {"ms": min(ts[1:]) * 1e3, "median_ms": float(np.median(ts[1:])) * 1e3, "mrays": primary / min(ts[1:]) / 1e6}
            same = bool((torch.from_numpy(dests["pageable"][0]).to(dev) == frame).all())
            # ... and what the reference's own caller asks for: the image alone, 3 bytes per pixel (PixelRGB8, Main.hs:621)
            scene.set_option(_lib.RT_OPT_RGB24, 1)
            rgb_h = api.host_buffer((h, w, 3), np.uint8)
            ts = []
            for _ in range(6):
                t0 = time.perf_counter()
                scene.render(w, h, depth=args.depth, spp=args.spp, seed=20260355, out_rgba=rgb_h, want_hit=False)
                ts.append(time.perf_counter() - t0)
            legs["page_locked_rgb24_image_only"] = {"ms": min(ts[1:]) * 1e3, "median_ms": float(np.median(ts[1:])) * 1e3, "mrays": primary / min(ts[1:]) / 1e6,
                                                    "equals_rgba_frame": bool((torch.from_numpy(np.asarray(rgb_h)).to(dev) == frame[:, :, :3]).all())}
            scene.set_option(_lib.RT_OPT_RGB24, 0)
            del rgb_h
            out["rt_render"] = {"page_locked": legs["page_locked"], "pageable": legs["pageable"], "frame_equals_device_render": same,
                                "page_locked_rgb24_image_only": legs["page_locked_rgb24_image_only"],
                                "note": "wall time of rt_render(scene, w, h, ...) into caller-owned host buffers, RGBA8 + int32 hit IDs, best of 5 "
                                        "after one warm-up; rendered as two row bands on the scene's two lanes; into page-locked memory k_final "
                                        "stores the colours straight over PCIe.  page_locked_rgb24_image_only: RT_OPT_RGB24, no hit IDs — the 24-bit "
                                        "image the reference's main writes"}
            out["rt_render_mrays"] = legs["page_locked"]["mrays"]
            del dests
            if args.workload == "synthetic" and (w, h) != (MULTI_W, MULTI_H):
                # configs[3]'s frame on this one GPU: the N = 1 point that the N > 1 lines of this script compare with
                big = torch.empty((MULTI_H, MULTI_W, 3), dtype=torch.uint8, device=dev)
                bhit = torch.empty((MULTI_H, MULTI_W), dtype=torch.int32, device=dev)
                scene.set_option(_lib.RT_OPT_RGB24, 1)       # (3 bytes per pixel, as the ranks of the N > 1 runs render it)
                ts = []
                for _ in range(4):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    scene.render_device(MULTI_W, MULTI_H, big.data_ptr(), bhit.data_ptr(), depth=args.depth, spp=args.spp, seed=20260355,
                                        stream=rs.cuda_stream)
                    rs.synchronize()
                    ts.append(time.perf_counter() - t0)
                # the kernels' own times from one more frame on ONE lane (in the timed frames the two halves overlap, and the
                # intervals between a lane's events then hold the other lane's kernels too)
                scene.render_device(MULTI_W, MULTI_H, big.data_ptr(), bhit.data_ptr(), depth=args.depth, spp=args.spp, seed=20260355,
                                    stream=rs.cuda_stream, lane=0, cont=False)
                bst = scene.stats()
                scene.set_option(_lib.RT_OPT_LANE, -1).set_option(_lib.RT_OPT_RGB24, 0)
                out["same_frame_as_multi_gpu"] = {"width": MULTI_W, "height": MULTI_H, "bytes_per_pixel": 3, "ms": min(ts[1:]) * 1e3,
                                                  "mrays": MULTI_W * MULTI_H * args.spp / min(ts[1:]) / 1e6,
                                                  "ms_on_one_lane": bst["ms_total"],
                                                  "kernel_ms_on_one_lane": {"k_primary": bst["ms_primary"], "k_setup": bst["ms_setup"], "k_shadow": bst["ms_shadow"], "k_final": bst["ms_final"]},
                                                  "note": "the frame `bench.py --gpus N` (N > 1) splits across ranks, on this one GPU: divide by it, not by `value`, for a scaling ratio.  "
                                                          "`ms`: as rt_render_device renders it by default, two halves on the scene's two lanes whose kernels fill each other's tails; "
                                                          "`*_on_one_lane`: one more frame with RT_OPT_LANE 0 (kernel after kernel), for the per-kernel times"}
                del big, bhit
        if n == 1 and mfma and not args.no_extra_legs and args.spp == 1:
            cs = cold_start_leg(args.workload, w, h, args.depth)
            if cs:
                out["cold_start"] = cs
                out["cold_start_ms"] = cs["cold_start_ms"]
        if n == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(world, lights, w, h, args.depth)
        print(json.dumps(out), flush=True)
    scene.close()
    if n > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
