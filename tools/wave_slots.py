#!/usr/bin/env python3
"""Per hardware wave slot (XCC, SE, CU, SIMD, slot of every wave: HW_ID / XCC_ID in the records of a -DRT_WAVE_TIMES build),
the dead time between a wave's end and the next wave's start, and how evenly the XCDs and CUs were loaded.
usage: RT_WAVE_TIMES_KEEP=records.npy RT_LIB=raytrace_amd/lib/ab/wt.so python tools/wave_times.py; python tools/wave_slots.py records.npy"""
import numpy as np, sys
rec = np.load(sys.argv[1])
for name, kind in (("primary", 0), ("shadow", 1)):
    r = rec[(rec[:, 0] & 0xff) == kind]
    hw = (r[:, 0] >> 16) & 0xffff; xcc = (r[:, 0] >> 32) & 0xf
    slot = hw & 0xf; simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    t0 = r[:, 2].astype(np.float64) / 100; t1 = r[:, 3].astype(np.float64) / 100
    base = t0.min(); t0 -= base; t1 -= base; span = t1.max()
    print(name, "waves", len(r), "span %.1f" % span, "xcc values", np.unique(xcc), "se", np.unique(se), "sh", np.unique(sh), "cu", np.unique(cu), "simd", np.unique(simd), "slot", np.unique(slot))
    key = ((((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd) * 16 + slot
    uk = np.unique(key)
    print("  distinct wave slots used:", len(uk))
    gaps = []; firsts = []; lasts = []
    order = np.lexsort((t0, key))
    k_s, a_s, b_s = key[order], t0[order], t1[order]
    same = k_s[1:] == k_s[:-1]
    g = (a_s[1:] - b_s[:-1])[same]
    mid = (b_s[:-1])[same]
    print("  gap between a wave's end and the next wave's start on the same slot: mean %.2f us, p50 %.2f, p90 %.2f, p99 %.2f, max %.1f; negative: %d" % (g.mean(), np.percentile(g, 50), np.percentile(g, 90), np.percentile(g, 99), g.max(), (g < 0).sum()))
    for lo, hi in ((0, .25), (.25, .5), (.5, .75), (.75, .9), (.9, 1.0)):
        sel = (mid >= lo * span) & (mid < hi * span)
        if sel.any(): print("     ends in [%.2f, %.2f) of the span: %6d gaps, mean %.2f us, p50 %.2f, p90 %.2f" % (lo, hi, sel.sum(), g[sel].mean(), np.percentile(g[sel], 50), np.percentile(g[sel], 90)))
    # per XCD: waves, busy time, last end
    for x in np.unique(xcc):
        m = xcc == x
        print("     xcc %d: %5d waves, busy %.0f slot-us, first start %.1f, last end %.1f, slots %d" % (x, m.sum(), (t1 - t0)[m].sum(), t0[m].min(), t1[m].max(), len(np.unique(key[m]))))
    # per CU: number of slots used
    cukey = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    ucu, cnt = np.unique(cukey, return_counts=True)
    busy = np.array([(t1 - t0)[cukey == c].sum() for c in ucu])
    print("     CUs used: %d; busy slot-us per CU: min %.0f mean %.0f max %.0f; waves per CU min %d max %d" % (len(ucu), busy.min(), busy.mean(), busy.max(), cnt.min(), cnt.max()))
