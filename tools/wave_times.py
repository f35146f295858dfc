#!/usr/bin/env python3
"""Per-wave timeline of the two big kernels on the benchmark frame, from a -DRT_WAVE_TIMES build
(tools/build_variant.sh wt -DRT_WAVE_TIMES; RT_LIB=raytrace_amd/lib/ab/wt.so python tools/wave_times.py).
Prints, per kernel: waves, duration percentiles (us), the kernel's span, the sum of wave durations / (span x slots)
(how full the wave slots were), and how the wave duration correlates with its work counts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytrace_amd import api, scenes, _lib
path = "/tmp/wave_times.bin"
os.environ["RT_WAVE_TIMES_FILE"] = path
_lib.check(_lib.lib().rt_init(0))
mirror = len(sys.argv) > 1 and sys.argv[1] == "mirror"        # the reflective workload at trace depth 2: k_general (level 1)
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 and not mirror else (3840, 2160)
world, lights = scenes.mirror_bench() if mirror else scenes.synthetic()
sc = api.Scene(world, lights)
sc.set_option(_lib.RT_OPT_HOST_BANDS, 1)           # one band: the records describe the whole frame
if os.environ.get("RT_NO_HISTORY"):                # waves in launch order (default: longest first from the second frame on)
    sc.set_option(_lib.RT_OPT_HISTORY, 0)
for _ in range(3):
    sc.render(w, h, depth=2 if mirror else 4)
    st = sc.stats()
raw = np.fromfile(path, dtype=np.uint64)
rec = raw[8:].reshape(-1, 8)
rec = rec[rec[:, 3] != 0]                                     # (a record per work-group slot; end = 0: not written)
n = len(rec)
if os.environ.get("RT_WAVE_TIMES_KEEP"):                      # keep the raw records (for scheduling experiments off the box)
    np.save(os.environ["RT_WAVE_TIMES_KEEP"], rec)
print(f"{n} wave records; kernel ms: primary {st['ms_primary']:.3f} shadow {st['ms_shadow']:.3f}")
for name, sel in (("k_primary_mfma", (rec[:, 0] & 0xff) == 0), ("k_shadow_mfma", (rec[:, 0] & 0xff) == 1), ("k_general", (rec[:, 0] & 0xff) == 2)):
    r = rec[sel]
    if len(r) == 0:
        continue
    t0, t1 = r[:, 2].astype(np.float64) / 100.0, r[:, 3].astype(np.float64) / 100.0      # us (100 MHz)
    dur = t1 - t0
    span = t1.max() - t0.min()
    slots = 1024 * (3 if name == "k_shadow_mfma" else 4)
    print(f"{name}: {len(r)} waves, span {span:.1f} us, wave duration us: mean {dur.mean():.1f} p50 {np.percentile(dur, 50):.1f} p90 {np.percentile(dur, 90):.1f} "
          f"p99 {np.percentile(dur, 99):.1f} max {dur.max():.1f}; slot occupancy {dur.sum() / (span * slots):.2f}")
    # when do waves finish: how long is the tail after 90 % of the waves have ended
    ends = np.sort(t1) - t0.min()
    print(f"   ends: 50% of waves by {ends[len(ends) // 2]:.1f} us, 90% by {ends[int(len(ends) * 0.9)]:.1f}, 99% by {ends[int(len(ends) * 0.99)]:.1f}, last {ends[-1]:.1f}")
    packed = r[:, 4]
    nb, nr, ne = (packed & 0xffff).astype(np.float64), ((packed >> 16) & 0xffffff).astype(np.float64), (packed >> 40).astype(np.float64)
    ph = [(r[:, 5 + k].astype(np.float64) / 100.0 - t0) for k in range(2)]
    acc = r[:, 7]
    t_s1, t_ex, t_bits, t_loop = ((acc >> (16 * k)) & 0xffff).astype(np.float64) / 100.0 if False else None, None, None, None
    t_s1, t_ex, t_bits, t_loop = [((acc >> np.uint64(16 * k)) & np.uint64(0xffff)).astype(np.float64) / 100.0 for k in range(4)]
    rest = dur - ph[1] - t_bits - t_loop
    print(f"   phases (mean us): ray set-up {ph[0].mean():.1f}, bound + guard proofs {(ph[1] - ph[0]).mean():.1f}; inside the scan: pass bits of blocks with candidates {t_bits.mean():.1f}, "
          f"queueing loop {t_loop.mean():.1f} (of which stage 1 {t_s1.mean():.1f}, exact {t_ex.mean():.1f}, the rest = extraction + pushes), "
          f"everything else (block test, MFMA scan, last flush, finish) {rest.mean():.1f}")
    r = np.stack([r[:, 0], r[:, 1], r[:, 2], r[:, 3], nb, nr, ne], axis=1).astype(np.float64)
    X = np.stack([np.ones(len(r)), r[:, 4].astype(np.float64), r[:, 5].astype(np.float64), r[:, 6].astype(np.float64)], axis=1)
    coef, *_ = np.linalg.lstsq(X, dur, rcond=None)
    pred = X @ coef
    print(f"   per wave: blocks {r[:, 4].mean():.1f} (max {r[:, 4].max()}), record tests {r[:, 5].mean():.0f} (max {r[:, 5].max()}), exact {r[:, 6].mean():.0f} (max {r[:, 6].max()})")
    print(f"   fit dur = {coef[0]:.2f} us + {coef[1]:.3f} us/block + {coef[2] * 64:.3f} us/64 record tests + {coef[3] * 64:.3f} us/64 exact;  R^2 {1 - ((dur - pred) ** 2).sum() / ((dur - dur.mean()) ** 2).sum():.3f}")
    print(f"   shares of mean duration: const {coef[0] / dur.mean():.2f}, blocks {coef[1] * r[:, 4].mean() / dur.mean():.2f}, records {coef[2] * r[:, 5].mean() / dur.mean():.2f}, exact {coef[3] * r[:, 6].mean() / dur.mean():.2f}")
    # how many waves are resident over the kernel's span, and how heavy the waves are along the launch order
    edges = np.linspace(t0.min(), t1.max(), 11)
    conc = [float(np.clip(np.minimum(t1, edges[k + 1]) - np.maximum(t0, edges[k]), 0, None).sum() / (edges[k + 1] - edges[k])) for k in range(10)]
    print("   resident waves in tenths of the span: " + " ".join(f"{c:.0f}" for c in conc) + f"  (slots {slots})")
    order = np.argsort(r[:, 1], kind="stable")
    tenth = np.array_split(order, 10)
    print("   mean wave duration (us) in tenths of the launch order: " + " ".join(f"{dur[ix].mean():.0f}" for ix in tenth))
    print("   mean start (us) in tenths of the launch order: " + " ".join(f"{(t0[ix] - t0.min()).mean():.0f}" for ix in tenth))
