"""Multi-GPU host logic: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

Pixels are independent (reference Main.hs:629-632 is a pure function of the index), so the image is
partitioned into contiguous row bands, one per rank, with no data-path collective; the only exchange
is the final gather of the bands to rank 0 (for the BMP writer), done as grouped point-to-point
send/recv because every peer GPU has its own direct xGMI link to GPU 0 and bands differ in size.
"""
from typing import List, Sequence, Tuple

import numpy as np


def split_rows(cost: Sequence[float], h: int, n: int, align: int = 8, weights: Sequence[float] = None) -> List[Tuple[int, int]]:
    """Split rows [0, h) into n contiguous bands (row0, nrows) of about equal summed cost.

    `cost` is a per-row cost estimate at any vertical resolution (len(cost) rows cover [0, h)).
    Cuts are multiples of `align` (the kernels' tile height); every band has at least `align` rows
    when h allows it.  `weights` (n positive numbers, default all equal): band k gets the share
    weights[k] / sum(weights) of the cost instead of 1 / n — rank 0 of a gather sends nothing, so it
    can render longer than the ranks whose last piece still has to travel (plan_frame).
    """
    if n < 1:
        raise ValueError("split_rows: n must be >= 1")
    cost = np.asarray(cost, dtype=np.float64)
    if cost.ndim != 1 or len(cost) == 0 or (cost < 0).any():
        raise ValueError("split_rows: cost must be a non-empty vector of non-negative numbers")
    if cost.sum() <= 0:
        cost = np.ones_like(cost)
    if weights is None:
        targets = [k / n for k in range(n + 1)]
    else:
        wts = [float(v) for v in weights]
        if len(wts) != n or not all(v > 0 for v in wts):
            raise ValueError("split_rows: weights must be n positive numbers")
        total, run, targets = sum(wts), 0.0, [0.0]
        for v in wts:                # (sequential sums, like the C port)
            run += v
            targets.append(run / total)
    sh = len(cost)
    cum = np.concatenate([[0.0], np.cumsum(cost)])
    cum = cum / cum[-1]              # (sequential sums, so that the C port rt_split_rows agrees bit for bit)
    cuts = [0]
    for k in range(1, n):
        r = int(round(float(np.interp(targets[k], cum, np.arange(sh + 1))) * h / sh))   # cost is piecewise constant per row
        r = (r // align) * align
        lo = min(h, cuts[-1] + align)
        hi = max(lo, h - (n - k) * align)
        cuts.append(min(max(r, lo), hi))
    cuts.append(h)
    return [(cuts[i], cuts[i + 1] - cuts[i]) for i in range(n)]


def piece_fractions(parts: int) -> List[float]:
    """Relative sizes of the pieces of a band: a small first piece (the link starts carrying data early), a small last
    one (what is still travelling when the render has ended is small), full-size ones between: 1 : 2 : ... : 2 : 1."""
    if parts < 1:
        raise ValueError("piece_fractions: parts must be >= 1")
    if parts <= 2:
        return [1.0 / parts] * parts
    units = [1.0] + [2.0] * (parts - 2) + [1.0]
    return [u / sum(units) for u in units]


def predict_step_ms(render_ms: float, transfer_ms: float, fractions: Sequence[float], overhead_ms: float) -> float:
    """When is the frame complete on rank 0?  A sending rank renders its pieces back to back (piece i: render_ms x
    fractions[i] + overhead_ms for its launches and tails) and its link carries piece i (transfer_ms x fractions[i]) as soon
    as it is rendered and piece i - 1 has arrived.  Returns the arrival of the last piece (all senders alike)."""
    t_render, t_link = 0.0, 0.0
    for f in fractions:
        t_render += render_ms * f + overhead_ms
        t_link = max(t_link, t_render) + transfer_ms * f
    return t_link


def plan_frame(frame_ms_one_gpu: float, n: int, band_bytes: float, link_gbps: float, overhead_ms: float = 0.1, max_parts: int = 8):
    """The N-rank frame on paper: pieces per band, rank 0's share, predicted step time.

    frame_ms_one_gpu: the whole frame on one GPU (measured); band_bytes: what one sender puts on its link per frame;
    link_gbps: what one xGMI link carries in one direction under ncclSend/ncclRecv.  Every sender has its own link to
    GPU 0, so the links run side by side and a rank's step = its render pipeline + what of its last piece is still on
    the link.  Rank 0 sends nothing and may render until the last piece arrives: its band gets the weight
    step / (a sender's render time), the others 1.  Returns (parts, fractions, weights, predicted_step_ms, details)."""
    if n <= 1:
        return 1, [1.0], [1.0] * max(n, 1), frame_ms_one_gpu, {"render_ms": frame_ms_one_gpu, "transfer_ms": 0.0}
    best = None
    w0 = 1.0
    for _ in range(3):                                   # rank 0's weight and the senders' band size depend on each other
        share = 1.0 / (w0 + (n - 1))                     # a sender's share of the frame's cost
        render = frame_ms_one_gpu * share
        transfer = band_bytes * share * n / (link_gbps * 1e9) * 1e3      # band_bytes is for an equal split
        best = None
        for parts in range(1, max_parts + 1):
            fr = piece_fractions(parts)
            t = predict_step_ms(render, transfer, fr, overhead_ms)
            if best is None or t < best[0] - 1e-9:
                best = (t, parts, fr)
        w0 = max(1.0, (best[0] - overhead_ms * best[1]) / max(render, 1e-9))
    t, parts, fr = best
    return parts, fr, [w0] + [1.0] * (n - 1), t, {"render_ms": render, "transfer_ms": transfer, "link_gbps": link_gbps,
                                                    "overhead_ms_per_piece": overhead_ms, "rank0_weight": w0}


def sub_bands(band: Tuple[int, int], parts: int, align: int = 8, fractions: Sequence[float] = None) -> List[Tuple[int, int]]:
    """Cut one band (row0, nrows) into exactly `parts` consecutive pieces at multiples of `align` rows; `fractions`
    (default: equal) gives their relative sizes (piece_fractions).

    Every rank renders its band piece by piece and hands each finished piece to the gather while the next
    one renders, so the exchange hides behind the render.  All ranks compute all ranks' pieces with this
    function (piece k of every peer travels in group k); a band too short for `parts` pieces gets empty
    ones, which are skipped by the render and send nothing.
    """
    if parts < 1:
        raise ValueError("sub_bands: parts must be >= 1")
    r0, nr = band
    units = -(-nr // align)                       # pieces are whole tiles except the band's last rows
    if fractions is not None:
        if len(fractions) != parts or not all(f >= 0 for f in fractions) or not sum(fractions) > 0:
            raise ValueError("sub_bands: fractions must be `parts` non-negative numbers")
        tot, run, cum = float(sum(fractions)), 0.0, []
        for f in fractions:
            run += float(f)
            cum.append(run / tot)
    out, done = [], 0
    for k in range(parts):
        if fractions is None:
            upto = min(nr, ((units * (k + 1)) // parts) * align) if k + 1 < parts else nr
        else:
            upto = min(nr, max(done, int(units * cum[k] + 0.5) * align)) if k + 1 < parts else nr
        out.append((r0 + done, upto - done))
        done = upto
    return out


def gather_bands(dist, frame, band, bands: Sequence[Tuple[int, int]], rank: int, wait: bool = True):
    """Send every rank's band to rank 0.

    rank 0: `frame` is the whole [h, w, C] image (its own band already in place); the other bands are
    received straight into their rows.  other ranks: `band` is their [nrows, w, C] band.
    One ncclGroup of send/recv (batch_isend_irecv): the peers' links to GPU 0 run in parallel.
    wait=False returns the requests instead of waiting on them (the caller overlaps the next piece's
    render with this exchange and waits at the end of the frame).
    """
    ops = []
    if rank == 0:
        for peer, (r0, nr) in enumerate(bands):
            if peer != 0 and nr > 0:
                ops.append(dist.P2POp(dist.irecv, frame[r0:r0 + nr], peer))
    elif band.shape[0] > 0:
        ops.append(dist.P2POp(dist.isend, band, 0))
    reqs = dist.batch_isend_irecv(ops) if ops else []
    if not wait:
        return reqs
    for req in reqs:
        req.wait()
    return []
