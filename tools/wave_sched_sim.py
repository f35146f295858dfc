#!/usr/bin/env python3
"""List-scheduler replay of measured wave durations under different dispatch orders (launch order, longest first, mixes),
with work-group i bound to XCD i % 8 and in-order dispatch: what an order could give if freed slots were refilled at once.
usage: python tools/wave_sched_sim.py records.npy   (records of a -DRT_WAVE_TIMES build, see tools/wave_slots.py)"""
import numpy as np, heapq, sys
rec = np.load(sys.argv[1])
def sim(durs, nx=8, slots_per_x=512, gap=0.5, inorder=True):
    # durs in dispatch order
    heaps = [[0.0] * slots_per_x for _ in range(nx)]
    for h in heaps: heapq.heapify(h)
    t_disp = 0.0; end = 0.0; busy = 0.0
    if inorder:
        for i, d in enumerate(durs):
            h = heaps[i % nx]
            free = heapq.heappop(h)
            st = max(t_disp, free + gap)
            t_disp = st + 0.00026
            heapq.heappush(h, st + d)
            end = max(end, st + d); busy += d
    else:   # each XCD works through its own list independently
        for x in range(nx):
            h = heaps[x]
            for d in durs[x::nx]:
                free = heapq.heappop(h)
                st = free + gap
                heapq.heappush(h, st + d)
                end = max(end, st + d); busy += d
    return end, busy / (end * nx * slots_per_x)
for name, kind, occ in (("primary", 0, 4), ("shadow", 1, 3)):
    r = rec[(rec[:, 0] & 0xff) == kind]
    dur = (r[:, 3].astype(np.float64) - r[:, 2].astype(np.float64)) / 100.0
    idx = r[:, 1].astype(np.int64)
    if kind == 1:
        light = (r[:, 0] >> 8).astype(np.int64)
    n = len(dur)
    print(name, n, "waves; sum dur / slots =", dur.sum() / (1024 * occ), "measured span", (r[:, 3].max() - r[:, 2].min()) / 100.0)
    rng = np.random.default_rng(1)
    nat = np.argsort(idx, kind="stable") if kind == 0 else np.lexsort((idx, light))
    orders = {"natural": dur[nat], "LPT": np.sort(dur)[::-1], "random": rng.permutation(dur)}
    # LPT in coarse groups, random inside: top 10% first, bottom 30% last, rest random
    s = np.sort(dur)[::-1]
    mid = s[int(0.1 * n):int(0.7 * n)].copy(); rng.shuffle(mid)
    orders["top10 | random mid | bottom30 desc"] = np.concatenate([s[:int(0.1 * n)], mid, s[int(0.7 * n):]])
    mid = s[int(0.2 * n):int(0.85 * n)].copy(); rng.shuffle(mid)
    orders["top20 | random mid | bottom15 desc"] = np.concatenate([s[:int(0.2 * n)], mid, s[int(0.85 * n):]])
    # snake: LPT dealt to XCDs so that each XCD's list is LPT (same as LPT for round robin) ; interleave heavy/cheap pairs
    half = n // 2
    inter = np.empty(n); inter[0::2] = s[:n - half]; inter[1::2] = s[n - half:][::-1][:half] if False else s[n - half:]
    orders["interleave heavy half / cheap half (both desc)"] = inter
    for k, d in orders.items():
        for gap in (0.5, 2.0):
            e1, o1 = sim(d, slots_per_x=128 * occ, gap=gap, inorder=True)
            e2, o2 = sim(d, slots_per_x=128 * occ, gap=gap, inorder=False)
            print(f"   {k:48s} gap {gap}: in-order round-robin {e1:6.1f} us (occ {o1:.2f}); independent XCD lists {e2:6.1f} us (occ {o2:.2f})")
print("---- finer partitions, primary, in-order round-robin")
r = rec[(rec[:, 0] & 0xff) == 0]
dur = (r[:, 3].astype(np.float64) - r[:, 2].astype(np.float64)) / 100.0
s = np.sort(dur)[::-1]
nat = dur[np.argsort(r[:, 1].astype(np.int64), kind="stable")]
for nx, sp in ((8, 512), (32, 128), (256, 16), (1024, 4)):
    for name, d in (("natural", nat), ("LPT", s)):
        e1, o1 = sim(d, nx=nx, slots_per_x=sp, gap=1.0, inorder=True)
        print(f"   {nx:4d} x {sp:3d} {name:8s}: {e1:6.1f} us occ {o1:.2f}")
