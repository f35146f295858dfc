#!/usr/bin/env python3
"""Wave starts per microsecond and resident waves over the span of k_primary_mfma / k_shadow_mfma, in 20 slices
(records of a -DRT_WAVE_TIMES build, see tools/wave_slots.py).  usage: python tools/wave_rate.py records.npy"""
import numpy as np, sys
rec = np.load(sys.argv[1])
for name, kind, slots in (("primary", 0, 4096), ("shadow", 1, 3072)):
    r = rec[(rec[:, 0] & 0xff) == kind]
    t0 = r[:, 2].astype(np.float64) / 100.0; t1 = r[:, 3].astype(np.float64) / 100.0
    base = t0.min(); t0 -= base; t1 -= base
    span = t1.max()
    print(name, "waves", len(r), "span", span)
    edges = np.linspace(0, span, 21)
    for k in range(20):
        a, b = edges[k], edges[k + 1]
        started = ((t0 >= a) & (t0 < b)).sum()
        resident = np.clip(np.minimum(t1, b) - np.maximum(t0, a), 0, None).sum() / (b - a)
        durs = (t1 - t0)[(t0 >= a) & (t0 < b)]
        print(f"  {a:6.1f}-{b:6.1f} us: starts/us {started / (b - a):6.1f}  resident {resident:6.0f}  mean dur of those started {durs.mean() if len(durs) else 0:6.1f}")
