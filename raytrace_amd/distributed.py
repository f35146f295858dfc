"""Multi-GPU host logic: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

Pixels are independent (reference Main.hs:629-632 is a pure function of the index), so the image is
partitioned into contiguous row bands, one per rank, with no data-path collective; the only exchange
is the final gather of the bands to rank 0 (for the BMP writer), done as grouped point-to-point
send/recv because every peer GPU has its own direct xGMI link to GPU 0 and bands differ in size.
"""
from typing import List, Sequence, Tuple

import numpy as np


def split_rows(cost: Sequence[float], h: int, n: int, align: int = 8) -> List[Tuple[int, int]]:
    """Split rows [0, h) into n contiguous bands (row0, nrows) of about equal summed cost.

    `cost` is a per-row cost estimate at any vertical resolution (len(cost) rows cover [0, h)).
    Cuts are multiples of `align` (the kernels' tile height); every band has at least `align` rows
    when h allows it.
    """
    if n < 1:
        raise ValueError("split_rows: n must be >= 1")
    cost = np.asarray(cost, dtype=np.float64)
    if cost.ndim != 1 or len(cost) == 0 or (cost < 0).any():
        raise ValueError("split_rows: cost must be a non-empty vector of non-negative numbers")
    if cost.sum() <= 0:
        cost = np.ones_like(cost)
    sh = len(cost)
    cum = np.concatenate([[0.0], np.cumsum(cost)])
    cum = cum / cum[-1]              # (sequential sums, so that the C port rt_split_rows agrees bit for bit)
    cuts = [0]
    for k in range(1, n):
        r = int(round(float(np.interp(k / n, cum, np.arange(sh + 1))) * h / sh))   # cost is piecewise constant per row
        r = (r // align) * align
        lo = min(h, cuts[-1] + align)
        hi = max(lo, h - (n - k) * align)
        cuts.append(min(max(r, lo), hi))
    cuts.append(h)
    return [(cuts[i], cuts[i + 1] - cuts[i]) for i in range(n)]


def sub_bands(band: Tuple[int, int], parts: int, align: int = 8) -> List[Tuple[int, int]]:
    """Cut one band (row0, nrows) into exactly `parts` consecutive pieces at multiples of `align` rows.

    Every rank renders its band piece by piece and hands each finished piece to the gather while the next
    one renders, so the exchange hides behind the render.  All ranks compute all ranks' pieces with this
    function (piece k of every peer travels in group k); a band too short for `parts` pieces gets empty
    ones, which are skipped by the render and send nothing.
    """
    if parts < 1:
        raise ValueError("sub_bands: parts must be >= 1")
    r0, nr = band
    units = -(-nr // align)                       # pieces are whole tiles except the band's last rows
    out, done = [], 0
    for k in range(parts):
        upto = min(nr, ((units * (k + 1)) // parts) * align) if k + 1 < parts else nr
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
