"""The N > 1 path on the CPU: two gloo ranks split an image into cost-balanced row bands, each renders its
band (with the oracle standing in for the GPU), the bands are gathered to rank 0 with the same
gather_bands() that bench.py uses over RCCL, and rank 0's image equals the single-process render."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytrace_amd.distributed import gather_bands, piece_fractions, plan_frame, predict_step_ms, split_rows, sub_bands

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_split_rows_properties():
    h = 2160
    for n in (1, 2, 3, 4, 8):
        cost = np.concatenate([np.ones(40), 5 * np.ones(55), np.ones(40)])
        bands = split_rows(cost, h, n)
        assert len(bands) == n and bands[0][0] == 0 and sum(nr for _, nr in bands) == h
        for i in range(1, n):
            assert bands[i][0] == bands[i - 1][0] + bands[i - 1][1] and bands[i][0] % 8 == 0 and bands[i][1] >= 8
        if n > 1:   # balanced: no band costs more than 1.35x the mean
            per_row = np.repeat(cost, h // len(cost))
            costs = [per_row[r0:r0 + nr].sum() for r0, nr in bands]
            assert max(costs) <= 1.35 * np.mean(costs)
    assert split_rows([0, 0, 0], 64, 2) == [(0, 32), (32, 32)]
    assert split_rows([1.0], 16, 2) == [(0, 8), (8, 8)]
    with pytest.raises(ValueError):
        split_rows([], 10, 2)


def test_sub_bands_properties():
    for band, parts in [((664, 496), 4), ((0, 8), 4), ((16, 20), 3), ((0, 0), 2), ((8, 2048), 4), ((0, 1), 1), ((24, 1000), 7)]:
        sb = sub_bands(band, parts)
        assert len(sb) == parts and sb[0][0] == band[0] and sum(nr for _, nr in sb) == band[1]
        for i in range(1, parts):
            assert sb[i][0] == sb[i - 1][0] + sb[i - 1][1] and sb[i][1] >= 0
            assert sb[i][1] == 0 or (sb[i][0] - band[0]) % 8 == 0        # pieces start on the kernels' tile rows
    assert sub_bands((8, 2048), 4) == [(8, 512), (520, 512), (1032, 512), (1544, 512)]
    with pytest.raises(ValueError):
        sub_bands((0, 8), 0)


def test_weighted_split_and_shaped_pieces():
    cost = np.concatenate([np.ones(40), 3 * np.ones(50), np.ones(38)])
    h = 16384
    eq = split_rows(cost, h, 8)
    wt = split_rows(cost, h, 8, weights=[1.25] + [1.0] * 7)
    assert wt[0][1] > eq[0][1] and sum(nr for _, nr in wt) == h and all(r0 % 8 == 0 for r0, _ in wt)
    per_row = np.repeat(cost, h // len(cost))
    c = [per_row[r0:r0 + nr].sum() for r0, nr in wt]
    assert abs(c[0] / np.mean(c[1:]) - 1.25) < 0.03            # rank 0's band costs a quarter more than the others'
    assert split_rows(cost, h, 8, weights=[1.0] * 8) == eq
    with pytest.raises(ValueError):
        split_rows(cost, h, 8, weights=[1.0] * 7)
    # pieces 1 : 2 : 2 : 1
    assert piece_fractions(1) == [1.0] and piece_fractions(2) == [0.5, 0.5]
    assert np.allclose(piece_fractions(4), [1 / 6, 2 / 6, 2 / 6, 1 / 6])
    sb = sub_bands((2048, 2040), 4, fractions=piece_fractions(4))
    assert sb[0][0] == 2048 and sum(nr for _, nr in sb) == 2040 and [nr for _, nr in sb] == [344, 680, 680, 336]
    for band, parts in [((0, 8), 4), ((16, 20), 3), ((0, 0), 2), ((24, 1000), 7), ((0, 1), 1)]:
        sb = sub_bands(band, parts, fractions=piece_fractions(parts))
        assert len(sb) == parts and sb[0][0] == band[0] and sum(nr for _, nr in sb) == band[1] and all(nr >= 0 for _, nr in sb)
        assert all(nr == 0 or (r0 - band[0]) % 8 == 0 for r0, nr in sb)


def test_pipeline_model():
    # one piece: nothing overlaps; many small pieces: the slower of link and render plus one piece of the other
    assert predict_step_ms(3.0, 2.0, [1.0], 0.0) == 5.0
    assert abs(predict_step_ms(3.0, 2.0, [0.25] * 4, 0.0) - 3.5) < 1e-12            # render-bound: + the last quarter on the link
    assert abs(predict_step_ms(2.0, 3.0, [0.25] * 4, 0.0) - 3.5) < 1e-12            # link-bound: + the first quarter's render
    # the 16384 x 16384 frame of configs[3] on 8 GPUs: 23.6 ms on one GPU, 134 MB per sender
    parts, fr, weights, t, detail = plan_frame(23.6, 8, 16384 * 16384 * 4 / 8, 48.0, 0.05)
    assert parts >= 3 and abs(sum(fr) - 1) < 1e-12 and weights[0] > 1.0 and weights[1:] == [1.0] * 7
    assert 23.6 / t > 6.0 and t >= detail["render_ms"]           # what the plan promises at these assumptions
    faster = plan_frame(23.6, 8, 16384 * 16384 * 4 / 8, 64.0, 0.05)[3]
    assert faster < t
    assert plan_frame(5.0, 1, 1e6, 48.0)[3] == 5.0


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    from raytrace_amd import scenes
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w, h = 48, 40
    wl, lights = scenes.stacked_cubes()
    osc = ob.OracleScene(wl, lights)
    # cost estimate from a coarse render, identical on every rank (deterministic)
    coarse = osc.render(24, 20, nthreads=1)["hit_id"]
    bands = split_rows(1.0 + 2.0 * (coarse >= 0).mean(axis=1), h, world)
    r0, nr = bands[rank]
    mine = torch.from_numpy(osc.render(w, h, row0=r0, nrows=nr, nthreads=1)["rgb"].copy())
    frame = torch.zeros((h, w, 3), dtype=torch.uint8) if rank == 0 else None
    if rank == 0:
        frame[r0:r0 + nr] = mine
    gather_bands(dist, frame, mine, bands, rank)
    dist.barrier()
    # the same frame again the way bench.py does it: every band in 3 pieces, piece k of both ranks in group k,
    # nothing waited on until the last piece has been handed over
    piped = torch.zeros((h, w, 3), dtype=torch.uint8) if rank == 0 else None
    subs = [sub_bands(b, 3, fractions=piece_fractions(3)) for b in split_rows(1.0 + 2.0 * (coarse >= 0).mean(axis=1), h, world, weights=[1.5, 1.0])]
    pending, keep = [], []
    for k in range(3):
        p0, pn = subs[rank][k]
        piece = torch.from_numpy(osc.render(w, h, row0=p0, nrows=pn, nthreads=1)["rgb"].copy()) if pn > 0 \
            else torch.zeros((0, w, 3), dtype=torch.uint8)
        keep.append(piece)
        if rank == 0:
            piped[p0:p0 + pn] = piece
        pending += gather_bands(dist, piped, piece, [s[k] for s in subs], rank, wait=False)
    for req in pending:
        req.wait()
    dist.barrier()
    if rank == 0:
        full = osc.render(w, h, nthreads=1)["rgb"]
        np.save(out_path, np.array([int((frame.numpy() != full).sum()), len(bands), int((piped.numpy() != full).sum())]))
    dist.destroy_process_group()


def test_two_rank_band_render_and_gather(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "res.npy")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    res = np.load(out)
    assert res[0] == 0 and res[1] == 2 and res[2] == 0
