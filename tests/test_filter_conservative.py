"""The FP32 filter must never reject a (ray, leaf) pair the reference accepts.  Checked on the CPU by
evaluating the filter records (rt_scene_get_filter_record) in numpy float32, exactly as the kernels do
(same operation order, fused multiply-adds emulated in float64-then-round), for every primary hit of the
oracle — and, for the shadow filter, for hit points that the oracle says are occluded."""
import ctypes as C

import numpy as np
import pytest

import oracle_binding as ob
from raytrace_amd import _lib, api, scenes


def fma32(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def records(sc, apex, w, h):
    L = _lib.lib()
    assert L.rt_scene_prepare_host(sc.handle, w, h) >= 0
    out = (C.c_float * 16)()
    recs = np.zeros((sc.num_leaves, 16), np.float32)
    for i in range(sc.num_leaves):
        assert L.rt_scene_get_filter_record(sc.handle, apex, i, out) == 0
        recs[i] = np.frombuffer(out, np.float32)
    kinds = np.array([int(sc.leaf(i)[0]) for i in range(sc.num_leaves)])
    return recs, kinds


def cone_rows(sc, apex, w, h):
    """First-stage rows of the matrix-core kernels (rt_scene_get_mfma_row), as the FP16 values the MFMA multiplies."""
    L = _lib.lib()
    assert L.rt_scene_prepare_host(sc.handle, w, h) >= 0
    out = (C.c_uint16 * 8)()
    rows = np.zeros((sc.num_leaves, 8), np.float64)
    for i in range(sc.num_leaves):
        assert L.rt_scene_get_mfma_row(sc.handle, apex, i, out) >= 0
        rows[i] = np.frombuffer(out, np.float16).astype(np.float64)
    return rows


def cone_form(row, D):
    """ray_fragment + one row x column of v_mfma_f32_32x32x16_f16 (rt_kernels.hip), for float32 directions D[n, 3]:
    2^10 D^ split into FP16 hi + mid, 16 exact products, summed here in float32 one after the other."""
    D = D.astype(np.float32)
    s = np.float32(1024.0) * (np.float32(1.0) / np.sqrt(fma32(D[:, 0], D[:, 0], fma32(D[:, 1], D[:, 1], D[:, 2] * D[:, 2]))))
    v = (D * s[:, None]).astype(np.float32)
    hi = v.astype(np.float16)
    mid = (v - hi.astype(np.float32)).astype(np.float16)
    acc = np.zeros(len(D), np.float32)
    for part, wv in ((hi, 1024.0), (mid, 0.0)):
        for i in range(3):
            for j in range(2):
                acc = (acc + (row[2 * i + j] * part[:, i].astype(np.float64)).astype(np.float32)).astype(np.float32)
        for j in range(2):
            acc = (acc + np.float32(row[6 + j] * wv)).astype(np.float32)
    return acc


@pytest.mark.parametrize("name,w,h", [("default", 128, 128), ("stacked_cubes", 96, 96), ("cylinder_axes", 100, 80),
                                      ("triangle_example", 120, 66), ("intersection", 96, 96), ("planes", 80, 60)])
def test_eye_cones_accept_every_oracle_hit(name, w, h):
    """Matrix-core first stage, eye apex: the bounding cone of the hit leaf accepts the pixel's ray, and the margin
    is not what saves it (the form stays positive with the margin taken away twice over is NOT required: only > 0)."""
    world, lights = scenes.REFERENCE_SCENES[name]()
    sc = api.Scene(world, lights)
    rows = cone_rows(sc, 0, w, h)
    cam = (C.c_double * 5)()
    _lib.lib().rt_camera_params(w, h, cam)
    hit = ob.OracleScene(world, lights).render(w, h, use_bbox=0)["hit_id"]
    ys, xs = np.nonzero(hit >= 0)
    D = np.stack([(cam[0] * xs.astype(np.float64) + cam[2]), (cam[1] * ys.astype(np.float64) + cam[3]),
                  np.full(len(xs), cam[4])], axis=1).astype(np.float32)
    ids = hit[ys, xs]
    assert len(ids) > 100
    kinds = np.array([int(sc.leaf(i)[0]) for i in range(sc.num_leaves)])
    for leaf in np.unique(ids):
        if kinds[leaf] == 2:
            continue            # spheres have their own loop
        m = ids == leaf
        f = cone_form(rows[leaf], D[m])
        assert (f > 0).all(), f"{name}: leaf {leaf} rejected by its bounding cone (min form {f.min()})"
    sc.close()


def test_eye_cones_are_selective():
    """...and the cones are a filter: for the default scene fewer than 2 % of (pixel, leaf) pairs pass, and a
    leaf's cone passes at most a few times the pixels the leaf really covers."""
    world, lights = scenes.default_scene()
    w = h = 96
    sc = api.Scene(world, lights)
    rows = cone_rows(sc, 0, w, h)
    cam = (C.c_double * 5)()
    _lib.lib().rt_camera_params(w, h, cam)
    ys, xs = np.mgrid[0:h, 0:w]
    D = np.stack([(cam[0] * xs.ravel().astype(np.float64) + cam[2]), (cam[1] * ys.ravel().astype(np.float64) + cam[3]),
                  np.full(w * h, cam[4])], axis=1).astype(np.float32)
    passed = sum(int((cone_form(rows[i], D) > 0).sum()) for i in range(sc.num_leaves))
    assert passed < 0.02 * w * h * sc.num_leaves
    sc.close()


def test_light_cones_accept_every_occluder():
    """Matrix-core first stage, light apex: for hit points of stacked_cubes and every leaf whose interior the segment
    to the light crosses, the leaf's cone (seen from the light) accepts the shadow ray's direction."""
    world, lights = scenes.stacked_cubes()
    w = h = 64
    sc = api.Scene(world, lights)
    o = ob.OracleScene(world, lights).render(w, h, use_bbox=0, want_t=True)
    point_lights = [l for l in lights if isinstance(l, api.PointLight)]
    leaves = [sc.leaf(i) for i in range(sc.num_leaves)]
    checked = 0
    for k, lt in enumerate(point_lights):
        rows = cone_rows(sc, 1 + k, w, h)
        for y in range(0, h, 3):
            for x in range(0, w, 3):
                if o["hit_id"][y, x] < 0:
                    continue
                ray = (C.c_double * 9)()
                ob.lib().orc_camera_ray(w, h, x, y, ray)
                p = np.array(ray[0:3]) + o["t"][y, x] * np.array(ray[3:6])
                tl = np.array(lt.pos) - p
                tl /= np.linalg.norm(tl)
                org = p + 1e-5 * tl
                dist = np.linalg.norm(np.array(lt.pos) - org)
                D = (org - np.array(lt.pos)).astype(np.float32)[None, :]
                for i, lf in enumerate(leaves):
                    n, d = np.array(lf[2:5]), lf[5]
                    den = n @ tl
                    if abs(den) < 1e-9:
                        continue
                    t = -(n @ org + d) / den
                    if not (-1e-4 < t < dist):
                        continue
                    q = org + t * tl - np.array(lf[6:9])
                    e1, e2 = np.array(lf[9:12]), np.array(lf[12:15])
                    if lf[0] == 0:
                        a, b = q @ e1 / lf[15], q @ e2 / lf[16]
                        inside = 0 <= a <= 1 and 0 <= b <= 1
                    else:
                        wv, wu = q @ e2, q @ e1
                        r, s = (lf[15] * wv - lf[17] * wu) / lf[18], (lf[15] * wu - lf[16] * wv) / lf[18]
                        inside = r >= 0 and s >= 0 and r + s <= 1
                    if not inside:
                        continue
                    f = cone_form(rows[i], D)[0]
                    assert f > 0, f"light {k} pixel ({x},{y}) leaf {i}: rejected by its bounding cone ({f})"
                    checked += 1
    assert checked > 500
    sc.close()


def eye_z(rec, kind, px, py):
    """filter4_eye of rt_kernels.hip for one leaf and arrays of pixels."""
    q = []
    for k in range(3):
        b = fma32(np.float32(rec[3 * k + 1]) * np.ones_like(py), py, np.float32(rec[3 * k + 2]) * np.ones_like(py))
        q.append(fma32(np.float32(rec[3 * k]) * np.ones_like(px), px, b))
    if kind == 1:
        return np.minimum(np.minimum(q[0], q[1]), q[2])
    return (q[2] - np.maximum(np.abs(q[0]), np.abs(q[1]))).astype(np.float32)


@pytest.mark.parametrize("name,w,h", [("default", 128, 128), ("stacked_cubes", 96, 96), ("cylinder_axes", 100, 80),
                                      ("triangle_example", 120, 66), ("spec_test", 64, 64)])
def test_eye_filter_accepts_every_oracle_hit(name, w, h):
    world, lights = scenes.REFERENCE_SCENES[name]()
    sc = api.Scene(world, lights)
    recs, kinds = records(sc, 0, w, h)
    cam = (C.c_double * 5)()
    _lib.lib().rt_camera_params(w, h, cam)
    hit = ob.OracleScene(world, lights).render(w, h, use_bbox=0)["hit_id"]
    ys, xs = np.nonzero(hit >= 0)
    px = (cam[0] * xs.astype(np.float64) + cam[2]).astype(np.float32)
    py = (cam[1] * ys.astype(np.float64) + cam[3]).astype(np.float32)
    ids = hit[ys, xs]
    assert len(ids) > 100
    for leaf in np.unique(ids):
        m = ids == leaf
        z = eye_z(recs[leaf], kinds[leaf], px[m], py[m])
        assert (z >= recs[leaf][9]).all(), f"{name}: leaf {leaf} rejected by the FP32 filter"
    sc.close()


def test_eye_filter_is_selective():
    """...and it is a filter: for the default scene far fewer than 1 % of (pixel, leaf) pairs pass."""
    world, lights = scenes.default_scene()
    w = h = 64
    sc = api.Scene(world, lights)
    recs, kinds = records(sc, 0, w, h)
    cam = (C.c_double * 5)()
    _lib.lib().rt_camera_params(w, h, cam)
    ys, xs = np.mgrid[0:h, 0:w]
    px = (cam[0] * xs.ravel().astype(np.float64) + cam[2]).astype(np.float32)
    py = (cam[1] * ys.ravel().astype(np.float64) + cam[3]).astype(np.float32)
    passed = sum(int((eye_z(recs[i], kinds[i], px, py) >= recs[i][9]).sum()) for i in range(sc.num_leaves))
    assert passed < 0.01 * w * h * sc.num_leaves
    sc.close()


def test_light_filter_accepts_every_occluder():
    """Shadow filter (apex = light): every leaf that the oracle's exact test finds
    between a hit point and the light must pass, for each such (pixel, light) of stacked_cubes."""
    world, lights = scenes.stacked_cubes()
    w = h = 64
    sc = api.Scene(world, lights)
    osc = ob.OracleScene(world, lights)
    o = osc.render(w, h, use_bbox=0, want_t=True)
    cam = (C.c_double * 5)()
    _lib.lib().rt_camera_params(w, h, cam)
    point_lights = [l for l in lights if isinstance(l, api.PointLight)]
    leaves = [sc.leaf(i) for i in range(sc.num_leaves)]
    checked = 0
    for k, lt in enumerate(point_lights):
        recs, kinds = records(sc, 1 + k, w, h)
        for y in range(0, h, 3):
            for x in range(0, w, 3):
                if o["hit_id"][y, x] < 0:
                    continue
                ray = (C.c_double * 9)()
                ob.lib().orc_camera_ray(w, h, x, y, ray)
                p = np.array(ray[0:3]) + o["t"][y, x] * np.array(ray[3:6])
                tl = np.array(lt.pos) - p
                tl /= np.linalg.norm(tl)
                org = p + 1e-5 * tl
                dist = np.linalg.norm(np.array(lt.pos) - org)
                D = (org - np.array(lt.pos)).astype(np.float32)
                for i, lf in enumerate(leaves):
                    if i == o["hit_id"][y, x]:
                        continue
                    n, d = np.array(lf[2:5]), lf[5]
                    den = n @ tl
                    if abs(den) < 1e-9:
                        continue
                    t = -(n @ org + d) / den
                    if not (1e-7 < t < dist - 1e-7):
                        continue
                    q = org + t * tl - np.array(lf[6:9])
                    e1, e2 = np.array(lf[9:12]), np.array(lf[12:15])
                    if lf[0] == 0:
                        a, b = q @ e1 / lf[15], q @ e2 / lf[16]
                        inside = 1e-6 < a < 1 - 1e-6 and 1e-6 < b < 1 - 1e-6
                    else:
                        wv, wu = q @ e2, q @ e1
                        r, s = (lf[15] * wv - lf[17] * wu) / lf[18], (lf[15] * wu - lf[16] * wv) / lf[18]
                        inside = r > 1e-6 and s > 1e-6 and r + s < 1 - 1e-6
                    if not inside:
                        continue
                    rec = recs[i]
                    f = [np.float32(np.float64(rec[3 * j]) * D[0] + np.float64(rec[3 * j + 1]) * D[1] + np.float64(rec[3 * j + 2]) * D[2]) for j in range(3)]
                    # third form: W for triangles, N for rectangles (rt_scene.cpp: make_rec, light apex)
                    z = min(f[0], f[1], f[2]) if kinds[i] == 1 else f[2] - max(abs(f[0]), abs(f[1]))
                    assert z >= rec[9], f"light {k} pixel ({x},{y}) leaf {i}: occluder rejected by the shadow filter"
                    # candidate path of k_shadow: the segment light..origin must be kept (N.D >= Kq)
                    nd = np.float32(np.float64(rec[11]) * D[0] + np.float64(rec[12]) * D[1] + np.float64(rec[13]) * D[2])
                    assert nd >= rec[10], f"light {k} pixel ({x},{y}) leaf {i}: occluder dropped by the segment clip"
                    checked += 1
    assert checked > 200
    sc.close()


def test_cone_rows_random_leaves_and_apexes():
    """make_cone_row on random geometry: for random triangles and (also non-orthogonal) rectangles seen from random
    apexes at distances from 1e-3 to 1e3 leaf sizes, every direction from the apex to a point of the accepted polygon
    (corners, edges and interior) passes the FP16 cone form — also when scaled to any length."""
    rng = np.random.default_rng(11)
    L = _lib.lib()
    checked = always = 0
    for case in range(250):
        size = 10.0 ** rng.uniform(-2, 2)
        c = rng.uniform(-50, 50, 3)
        e1, e2 = rng.normal(size=3) * size, rng.normal(size=3) * size
        tri = case % 2 == 0
        if tri:
            world = api.triangle(api.red, tuple(c), tuple(c + e1), tuple(c + e2))
            corners = np.array([c, c + e1, c + e2])
        else:
            world = api.rectangle(api.red, tuple(c), tuple(e1), tuple(e2))
            ww, hh, wh = e1 @ e1, e2 @ e2, e1 @ e2
            p0 = c - 0.5 * e1 - 0.5 * e2
            det = ww * hh - wh * wh
            corners = np.array([p0 + ((s * hh - t * wh) / det) * e1 + ((t * ww - s * wh) / det) * e2
                                for s, t in ((0, 0), (ww, 0), (ww, hh), (0, hh))])
        centre = corners.mean(axis=0)
        apex = centre + rng.normal(size=3) * size * 10.0 ** rng.uniform(-3, 3)
        sc = api.Scene(world, api.pointLight(0.5, 0.5, tuple(apex)))
        assert L.rt_scene_prepare_host(sc.handle, 64, 64) >= 0
        out = (C.c_uint16 * 8)()
        assert L.rt_scene_get_mfma_row(sc.handle, 1, 0, out) >= 0
        row = np.frombuffer(out, np.float16).astype(np.float64)
        sc.close()
        if not row[:6].any():
            always += 1
            assert row[6] > 0          # a constant row must be "always a candidate" for a leaf that can be hit
            continue
        wts = rng.dirichlet(np.ones(len(corners)) * 0.3, size=200)          # concentrated near corners and edges
        pts = np.concatenate([corners, wts @ corners])
        D = (pts - apex) * (10.0 ** rng.uniform(-3, 3, (len(pts), 1)))
        f = cone_form(row, D.astype(np.float32))
        assert (f > 0).all(), f"case {case}: min form {f.min()} ({'triangle' if tri else 'rectangle'})"
        checked += len(pts)
    assert checked > 20000 and always < 120


def test_fp32_records_random_leaves_and_apexes():
    """make_rec on random geometry (light apex): for random triangles and rectangles — also with non-orthogonal edge
    vectors, whose accepted region is the dual parallelogram of Main.hs:238-241 — every direction from the apex through
    a point of the accepted polygon passes the FP32 record test and, with the origin beyond the leaf, the segment clip."""
    rng = np.random.default_rng(12)
    L = _lib.lib()
    checked = 0
    for case in range(250):
        size = 10.0 ** rng.uniform(-1.5, 1.5)
        c = rng.uniform(-30, 30, 3)
        e1, e2 = rng.normal(size=3) * size, rng.normal(size=3) * size
        tri = case % 2 == 0
        if tri:
            world = api.triangle(api.red, tuple(c), tuple(c + e1), tuple(c + e2))
            corners = np.array([c, c + e1, c + e2])
        else:
            world = api.rectangle(api.red, tuple(c), tuple(e1), tuple(e2))
            ww, hh, wh = e1 @ e1, e2 @ e2, e1 @ e2
            p0 = c - 0.5 * e1 - 0.5 * e2
            det = ww * hh - wh * wh
            corners = np.array([p0 + ((s * hh - t * wh) / det) * e1 + ((t * ww - s * wh) / det) * e2
                                for s, t in ((0, 0), (ww, 0), (ww, hh), (0, hh))])
        centre = corners.mean(axis=0)
        apex = centre + rng.normal(size=3) * size * 10.0 ** rng.uniform(-1, 2)
        # a far second leaf stretches the scene box, as other geometry would (the margins depend on |D|max)
        far = api.triangle(api.blue, tuple(centre + 3 * (centre - apex) + 1), tuple(centre + 3 * (centre - apex) - 1), tuple(centre + 3.5 * (centre - apex)))
        sc = api.Scene(api.mconcat([world, far]), api.pointLight(0.5, 0.5, tuple(apex)))
        assert L.rt_scene_prepare_host(sc.handle, 64, 64) >= 0
        out = (C.c_float * 16)()
        assert L.rt_scene_get_filter_record(sc.handle, 1, 0, out) == 0
        rec = np.frombuffer(out, np.float32).copy()
        sc.close()
        wts = rng.dirichlet(np.ones(len(corners)) * 0.3, size=100)
        pts = np.concatenate([corners, wts @ corners])
        D = ((pts - apex) * 1.5).astype(np.float32)                   # shadow-ray origin beyond the leaf, seen from the light
        f = [fma32(np.full(len(D), rec[3 * j], np.float32), D[:, 0], fma32(np.full(len(D), rec[3 * j + 1], np.float32), D[:, 1],
                                                                          (np.float32(rec[3 * j + 2]) * D[:, 2]).astype(np.float32))) for j in range(3)]
        z = np.minimum(np.minimum(f[0], f[1]), f[2]) if tri else (f[2] - np.maximum(np.abs(f[0]), np.abs(f[1]))).astype(np.float32)
        assert (z >= rec[9]).all(), f"case {case} ({'triangle' if tri else 'rectangle'}): min z {z.min()} < {rec[9]}"
        nd = fma32(np.full(len(D), rec[11], np.float32), D[:, 0], fma32(np.full(len(D), rec[12], np.float32), D[:, 1],
                                                                      (np.float32(rec[13]) * D[:, 2]).astype(np.float32)))
        assert (nd >= rec[10]).all(), f"case {case}: segment clip rejects a leaf between light and origin"
        checked += len(D)
    assert checked > 20000


# ---------------------------------------------------------------------------------------------
# Block cones: the matrix-core kernels skip a block of 32 stream positions when its bounding cone misses the bound of
# a wave's ray directions (rt_scene.cpp: block_cone, rt_kernels.hip: culled_scan).

def stream_layout(sc, apex, w, h):
    """(position in the apex's stream, decoded row) of every leaf, and the block cones."""
    L = _lib.lib()
    assert L.rt_scene_prepare_host(sc.handle, w, h) >= 0
    out = (C.c_uint16 * 8)()
    pos = np.zeros(sc.num_leaves, np.int64)
    rows = np.zeros((sc.num_leaves, 8), np.float64)
    for i in range(sc.num_leaves):
        if int(sc.leaf(i)[0]) == 2:
            pos[i] = -1                                                   # spheres are not in the stream
            continue
        pos[i] = L.rt_scene_get_mfma_row(sc.handle, apex, i, out)
        assert pos[i] >= 0
        rows[i] = np.frombuffer(out, np.float16).astype(np.float64)
    cone = (C.c_float * 8)()
    groups = L.rt_scene_get_block_cone(sc.handle, apex, 0, cone)
    assert groups > 0 and groups % 3 == 0
    cones = np.zeros((groups + 2, 8), np.float32)
    for g in range(groups + 2):
        assert L.rt_scene_get_block_cone(sc.handle, apex, g, cone) == groups
        cones[g] = np.frombuffer(cone, np.float32)
    return pos, rows, cones


@pytest.mark.parametrize("name,w,h", [("default", 128, 128), ("stacked_cubes", 96, 96), ("cylinder_axes", 100, 80),
                                      ("intersection", 96, 96), ("adversarial", 64, 64)])
def test_block_cones_contain_their_leaf_cones(name, w, h):
    world, lights = scenes.adversarial() if name == "adversarial" else scenes.REFERENCE_SCENES[name]()
    sc = api.Scene(world, lights)
    napex = 1 + sum(isinstance(l, api.PointLight) for l in lights)
    planar = [i for i in range(sc.num_leaves) if int(sc.leaf(i)[0]) != 2]
    for apex in range(napex):
        pos, rows, cones = stream_layout(sc, apex, w, h)
        assert len(set(pos[planar])) == len(planar)                       # every planar leaf has its own stream position
        assert (cones[-2:, 5] == 1).all()                                 # the padding blocks are "never" blocks
        for i in planar:
            blk = cones[pos[i] // 32]
            axis = np.array([rows[i][0] + rows[i][1], rows[i][2] + rows[i][3], rows[i][4] + rows[i][5]]) / 1024.0
            const = (rows[i][6] + rows[i][7]) / 1024.0
            if not axis.any():                                            # constant row
                if const > 0:
                    assert blk[5] == 2, f"{name} apex {apex}: an 'always' leaf sits in a block that can be skipped"
                continue
            if blk[5] == 2:
                continue
            assert blk[5] == 0
            cs = min(1.0, 2.0 ** -17 - const)                             # row constant = margin - cos(theta)
            A = blk[:3].astype(np.float64)
            between = np.arccos(np.clip(A @ axis / np.linalg.norm(axis), -1, 1))
            T = np.arctan2(float(blk[4]), float(blk[3]))
            # (the row holds cos(theta) to 2^-22: for a tiny leaf, theta read back from it is uncertain by that much)
            slack = np.arccos(cs) - np.arccos(min(1.0, cs + 2.0 ** -22))
            assert between + np.arccos(cs) <= T + 1e-6 + slack, f"{name} apex {apex} leaf {i}: leaf cone sticks out of its block cone"
    sc.close()


def tile_bound(cam, x0, y0):
    """k_primary_mfma's bound of a 32 x 8 pixel tile (a cone), in float32 like the kernel."""
    f = np.float32
    cx, cy, dz = f(cam[0] * (x0 + 15.5) + cam[2]), f(cam[1] * (y0 + 3.5) + cam[3]), f(cam[4])
    hx, hy = f(17.5 * abs(cam[0])), f(5.5 * abs(cam[1]))
    cn = f(1.0) / np.sqrt(fma32(cx, cx, fma32(cy, cy, dz * dz)))
    axis = np.array([cx * cn, cy * cn, dz * cn], np.float32)
    cmin = f(1.0)
    for k in range(4):
        qx, qy = cx + (hx if k & 1 else -hx), cy + (hy if k & 2 else -hy)
        qn = f(1.0) / np.sqrt(fma32(qx, qx, fma32(qy, qy, dz * dz)))
        cmin = min(cmin, f((axis[0] * qx + axis[1] * qy + axis[2] * dz) * qn))
    cr = f(cmin - f(1e-6))
    sr = np.sqrt(max(f(0), f((f(1) - cr) * (f(1) + cr))))
    return axis, cr, f(sr)


def cone_block_passes(axis, cr, sr, blk):
    """culled_scan's test for a cone-shaped bound (p = q)."""
    if blk[5] == 2:
        return True
    if blk[5] != 0:
        return False
    cc = fma32(cr, blk[3], -(sr * blk[4])) - np.float32(4e-6)
    return bool(cc <= 0 or fma32(axis[0], blk[0], fma32(axis[1], blk[1], axis[2] * blk[2])) >= cc)


@pytest.mark.parametrize("name,w,h", [("default", 256, 256), ("stacked_cubes", 200, 120), ("cylinder_axes", 160, 96)])
def test_tile_bound_keeps_the_block_of_every_oracle_hit(name, w, h):
    """Eye apex: for every pixel the oracle hits, the block holding the hit leaf passes the block test of the pixel's
    tile -- and the test is selective (most (tile, block) pairs are skipped in the height field)."""
    world, lights = scenes.REFERENCE_SCENES[name]()
    sc = api.Scene(world, lights)
    pos, rows, cones = stream_layout(sc, 0, w, h)
    cam = (C.c_double * 5)()
    _lib.lib().rt_camera_params(w, h, cam)
    hit = ob.OracleScene(world, lights).render(w, h, use_bbox=0)["hit_id"]
    kinds = np.array([int(sc.leaf(i)[0]) for i in range(sc.num_leaves)])
    passed = total = 0
    for y0 in range(0, h, 8):
        for x0 in range(0, w, 32):
            axis, cr, sr = tile_bound(cam, x0, y0)
            ok = np.array([cone_block_passes(axis, cr, sr, b) for b in cones[:-2]])
            passed += int(ok.sum()); total += len(ok)
            for leaf in np.unique(hit[y0:y0 + 8, x0:x0 + 32]):
                if leaf >= 0 and kinds[leaf] != 2:
                    assert ok[pos[leaf] // 32], f"{name}: tile ({x0},{y0}) skips the block of leaf {leaf}, which it hits"
    if name == "default":
        assert passed < 0.5 * total
    sc.close()


# ---------------------------------------------------------------------------------------------
# Guard boxes a whole pixel tile certainly passes (rt_kernels.hip: tile_guard_mask)

def intersect_bb(box, ray):
    """intersectBB, Main.hs:152-174, in IEEE double with the reference's expression order."""
    x0, y0, z0, x1, y1, z1 = box
    px, py, pz, dx, dy, dz, rx, ry, rz = ray

    def xy(z):
        t = (z - pz) * rz
        a, b = px + t * dx, py + t * dy
        return t >= 0 and x0 <= a <= x1 and y0 <= b <= y1

    def yz(x):
        t = (x - px) * rx
        a, b = py + t * dy, pz + t * dz
        return t >= 0 and y0 <= a <= y1 and z0 <= b <= z1

    def xz(y):
        t = (y - py) * ry
        a, b = px + t * dx, pz + t * dz
        return t >= 0 and x0 <= a <= x1 and z0 <= b <= z1

    return xy(z0) or xy(z1) or xz(y0) or xz(y1) or yz(x0) or yz(x1)


def tile_certainly_passes(cam, x0, y0, box):
    """The per-box part of tile_guard_mask, same arithmetic (IEEE double)."""
    sx, sy, dX, dY, d = cam
    bx0, by0, bz0, bx1, by1, bz1 = box
    xa, xb = sx * float(x0 - 1) + dX, sx * float(x0 + 32) + dX
    ya, yb = sy * float(y0 - 1) + dY, sy * float(y0 + 8) + dY
    XL, XH, YL, YH = min(xa, xb), max(xa, xb), min(ya, yb), max(ya, yb)
    ok = False
    for zf in (bz1, bz0):
        q = zf / d
        sc = 1.0 + q
        lx, hx, ly, hy = XL * sc, XH * sc, YL * sc, YH * sc
        mx = 1e-9 * (abs(bx0) + abs(bx1) + abs(lx) + abs(hx)) + 1e-300
        my = 1e-9 * (abs(by0) + abs(by1) + abs(ly) + abs(hy)) + 1e-300
        ok |= q > 1e-9 and bx0 + mx <= lx and hx <= bx1 - mx and by0 + my <= ly and hy <= by1 - my
    return ok


@pytest.mark.parametrize("name,w,h", [("default", 96, 96), ("stacked_cubes", 128, 72), ("cylinder_axes", 100, 80), ("synthetic", 160, 90)])
def test_tiles_that_certainly_pass_a_guard_box_do(name, w, h):
    """Whenever the tile test says "every ray of this tile passes the box", intersectBB is True for the rays of all its
    pixels; and for the big shared boxes of the synthetic scene the test does say so."""
    world, lights = scenes.synthetic(ntri=300, ncuboid=40) if name == "synthetic" else scenes.REFERENCE_SCENES[name]()
    sc = api.Scene(world, lights)
    L = _lib.lib()
    assert L.rt_scene_prepare_host(sc.handle, w, h) >= 0
    buf = (C.c_double * (6 * 4096))()
    boxes = set()
    for i in range(sc.num_leaves):
        n = L.rt_scene_get_guards(sc.handle, i, buf, 4096)
        for k in range(n):
            boxes.add(tuple(buf[6 * k:6 * k + 6]))
    assert boxes
    cam = (C.c_double * 5)()
    L.rt_camera_params(w, h, cam)
    ray = (C.c_double * 9)()
    said_yes = pairs = 0
    for y0 in range(0, h, 8):
        for x0 in range(0, w, 32):
            rays = []
            for y in range(y0, min(h, y0 + 8)):
                for x in range(x0, min(w, x0 + 32)):
                    ob.lib().orc_camera_ray(w, h, x, y, ray)
                    rays.append(tuple(ray))
            for box in boxes:
                pairs += 1
                if tile_certainly_passes(tuple(cam), x0, y0, box):
                    said_yes += 1
                    assert all(intersect_bb(box, r) for r in rays), f"{name}: tile ({x0},{y0}) box {box}"
    print(f"{name}: {said_yes} of {pairs} (tile, box) pairs certainly pass")
    if name == "synthetic":
        assert said_yes > 0.1 * pairs
    sc.close()


def bundle_certainly_passes(lo, hi, lp, box):
    """The per-box part of bundle_guard_mask (rt_kernels.hip), same arithmetic: IEEE single, every input rounded to it."""
    f32 = np.float32
    lo, hi, lp = [np.asarray(v, dtype=np.float64).astype(f32) for v in (lo, hi, lp)]
    b0, b1 = np.asarray(box[:3]).astype(f32), np.asarray(box[3:]).astype(f32)
    ok = False
    for a in range(3):
        for f in (b0[a], b1[a]):
            n0, n1, d0, d1 = f - lo[a], f - hi[a], lp[a] - lo[a], lp[a] - hi[a]
            S = abs(f) + abs(lp[a]) + abs(lo[a]) + abs(hi[a])
            tiny = f32(1e-5) * S + f32(1e-30)
            far = f32(1e-2) * S + f32(1e-4)
            face = (n0 > tiny and n1 > tiny and d0 > far and d1 > far) or (n0 < -tiny and n1 < -tiny and d0 < -far and d1 < -far)
            if not face:
                continue
            u0, u1 = n0 * (f32(1) / d0), n1 * (f32(1) / d1)
            ul, uh = min(u0, u1), max(u0, u1)
            for k in (1, 2):
                b = (a + k) % 3
                v = [lo[b] + (lp[b] - lo[b]) * ul, lo[b] + (lp[b] - lo[b]) * uh, hi[b] + (lp[b] - hi[b]) * ul, hi[b] + (lp[b] - hi[b]) * uh]
                m = f32(1e-4) * (abs(b0[b]) + abs(b1[b]) + (abs(lp[b]) + abs(lo[b]) + abs(hi[b])) * (f32(1) + uh)) + f32(1e-30)
                face = face and b0[b] + m <= min(v) and max(v) <= b1[b] - m
            ok = ok or face
    return bool(ok)


def test_shadow_bundles_that_certainly_pass_a_guard_box_do():
    """Random hit points, lights and boxes: whenever the wave-level test says every shadow ray passes the box,
    intersectBB (Main.hs:152-174) is True for each of the rays (Main.hs:472: origin hit + 1e-5 L^, direction L^).  The
    boxes are tight around the origins as often as loose, lights sit near face planes, scales go from 1e-2 to 1e3: the
    margins of the FP32 proof are what is being tested."""
    rng = np.random.default_rng(99)
    said_yes = 0
    for trial in range(700):
        scale = 10.0 ** rng.uniform(-2, 3) if trial % 3 == 0 else 1.0
        centre = rng.uniform(-10, 10, 3) * scale
        spread = rng.uniform(0.01, 6, 3) * rng.choice([1.0, 0.05], 3) * scale
        hits = centre + rng.uniform(-1, 1, (64, 3)) * spread
        lp = rng.uniform(-15, 15, 3) * scale
        L = lp - hits
        L /= np.linalg.norm(L, axis=1)[:, None]
        org = hits + 0.00001 * L
        lo, hi = org.min(axis=0), org.max(axis=0)
        rays = [tuple(o) + tuple(d) + tuple(1.0 / d) for o, d in zip(org, L)]
        for k in range(12):
            grow0, grow1 = rng.uniform(-0.5, 8, 3) * scale, rng.uniform(-0.5, 8, 3) * scale
            if k % 3 == 0:                                  # a face a hair beyond the rays' crossings: the margin decides
                grow0 *= rng.choice([1.0, 1e-3, 1e-5], 3); grow1 *= rng.choice([1.0, 1e-3, 1e-5], 3)
            box = tuple(lo - grow0) + tuple(hi + grow1)
            if bundle_certainly_passes(lo, hi, lp, box):
                said_yes += 1
                assert all(intersect_bb(box, r) for r in rays), f"trial {trial}: box {box} light {lp}"
    assert said_yes > 500


def test_general_ray_blocks_cover_every_leaf_and_their_spheres_hold_them():
    """Reflected rays: every planar leaf sits in exactly one block of the spatial order of its kind, and the block's
    bounding sphere holds the whole accepting region of each of its leaves (triangle / parallelogram corners, in double)."""
    L = _lib.lib()
    finite = 0
    for world, lights in (scenes.mirror_room(), scenes.mirror_bench(ntri=700, ncuboid=60), scenes.adversarial()):
        sc = api.Scene(world, lights)
        assert L.rt_scene_prepare_host(sc.handle, 320, 200) >= 0
        seen = []
        for kind in (0, 1):
            sph, ids = (C.c_float * 4)(), (C.c_int32 * 32)()
            nblk = L.rt_scene_get_general_block(sc.handle, kind, 0, sph, ids)
            if nblk <= 0:
                continue
            for b in range(nblk):
                assert L.rt_scene_get_general_block(sc.handle, kind, b, sph, ids) == nblk
                centre, radius = np.array(sph[:3], np.float64), float(sph[3])
                for g in ids:
                    if g < 0:
                        continue
                    seen.append(g)
                    leaf = sc.leaf(g)
                    assert int(leaf[0]) == kind
                    if not np.isfinite(radius):
                        continue
                    finite += 1
                    p0, e1, e2 = np.array(leaf[6:9]), np.array(leaf[9:12]), np.array(leaf[12:15])
                    if kind == 1:
                        corners = [p0, p0 + e1, p0 + e2]
                    else:       # the parallelogram 0 <= dV.w <= ww, 0 <= dV.h <= hh (w, h need not be orthogonal)
                        ww, hh, wh = e1 @ e1, e2 @ e2, e1 @ e2
                        det = ww * hh - wh * wh
                        corners = [p0 + ((s_ * hh - t_ * wh) / det) * e1 + ((t_ * ww - s_ * wh) / det) * e2
                                   for s_, t_ in ((0, 0), (ww, 0), (ww, hh), (0, hh))]
                    for p in corners:
                        assert np.linalg.norm(p - centre) <= radius, (kind, b, g)
        planar = [g for g in range(sc.num_leaves) if int(sc.leaf(g)[0]) in (0, 1)]
        assert sorted(seen) == planar
        sc.close()
    assert finite > 1000


def test_general_ray_blocks_are_tighter_than_runs_of_morton_keys():
    """The blocks of 32 leaves the bounce levels skip are cut top down at medians (rt_scene.cpp: spatial_order).  What a run
    of reflected rays pays for is the cross-section of the blocks' spheres: the sum of r^2 over the triangle blocks of the
    mirror workload's clutter must stay well below what runs of 32 consecutive Morton keys with centroid spheres give
    (the order the library had until round 3: 41.5 blocks looked at per run instead of 14.1)."""
    L = _lib.lib()
    world, lights = scenes.mirror_bench(ntri=700, ncuboid=60)
    sc = api.Scene(world, lights)
    assert L.rt_scene_prepare_host(sc.handle, 320, 200) >= 0
    sph, ids = (C.c_float * 4)(), (C.c_int32 * 32)()
    nblk = L.rt_scene_get_general_block(sc.handle, 1, 0, sph, ids)
    lib_r2, tris = 0.0, []
    for b in range(nblk):
        assert L.rt_scene_get_general_block(sc.handle, 1, b, sph, ids) == nblk
        lib_r2 += float(sph[3]) ** 2
        for g in ids:
            if g >= 0:
                leaf = sc.leaf(g)
                p0, e1, e2 = np.array(leaf[6:9]), np.array(leaf[9:12]), np.array(leaf[12:15])
                tris.append(np.array([p0, p0 + e1, p0 + e2]))
    sc.close()
    tris = np.array(tris)
    assert len(tris) == 700 and nblk == 22
    pts = tris.reshape(-1, 3)
    q = ((tris.mean(axis=1) - pts.min(axis=0)) / (pts.max(axis=0) - pts.min(axis=0)) * 1023).astype(np.uint64)

    def spread(u):
        u = (u | (u << 16)) & 0x030000ff
        u = (u | (u << 8)) & 0x0300f00f
        u = (u | (u << 4)) & 0x030c30c3
        return (u | (u << 2)) & 0x09249249
    order = np.argsort(spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2), kind="stable")
    morton_r2 = 0.0
    for b in range(0, len(order), 32):
        blk = tris[order[b:b + 32]].reshape(-1, 3)
        morton_r2 += float(((blk - blk.mean(axis=0)) ** 2).sum(axis=1).max())
    assert lib_r2 < 0.6 * morton_r2, (lib_r2, morton_r2)


def _split16(v):
    """hi and mid binary16 parts of float32 values, as the kernel and the host split them."""
    v = np.asarray(v, np.float32)
    hi = v.astype(np.float16)
    mid = (v - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float32), mid.astype(np.float32)


def _general_B(o, d, centre, radius):
    """The ray's vector (32 values, K order) in a frame, as k_general builds it: FP32 arithmetic, every value x 2^10,
    split into binary16 hi and mid parts; slot layout of rt_scene.cpp: sphere_row."""
    f = np.float32
    o, d = o.astype(f), d.astype(f)
    u = d * f(1.0 / np.sqrt(f(d @ d)))
    p = o - centre.astype(f)
    sd = f(p @ u)
    invs = f(1024.0) / f(radius) if np.isfinite(radius) else f(0.0)
    q = np.clip((p - sd * u) * invs, -4096, 4096).astype(f)
    qq = f(q @ q) * f(1.0 / 1024.0)
    (qqh, qxh, qyh, qzh), (qqm, qxm, qym, qzm) = _split16([qq, q[0], q[1], q[2]])
    D = np.array([u[0] * u[0], u[1] * u[1], u[2] * u[2], u[0] * u[1], u[0] * u[2], u[1] * u[2]], f) * f(1024.0)
    Dh, Dm = _split16(D)
    one = f(1024.0)
    return np.array([one, one, qqh, qqm, qxh, qxm, qxh, Dm[5],
                     qyh, qym, qyh, qzh, qzm, qzh, Dh[5], 0,
                     Dh[0], Dm[0], Dh[0], Dh[1], Dm[1], Dh[1], Dh[2], Dm[2],
                     Dh[2], Dh[3], Dm[3], Dh[3], Dh[4], Dm[4], Dh[4], Dh[5]], f)


def test_general_rows_pass_every_ray_that_meets_the_leaf():
    """Stage 0 of the bounce levels (rt_scene.cpp: sphere_row, k_general): whenever a ray's line passes through a point of
    a leaf, the linear form of the leaf's row against the ray's vector is > 0 - replayed with binary16 operands and a
    float32 sum, less 2^20 x 1e-5 for the order of the matrix core's accumulation (the margin allows for 9.6e-6).
    Origins anywhere in the scene, also very near the leaf and far away; and the filter is a filter: few random (ray,
    leaf) pairs pass."""
    L = _lib.lib()
    rng = np.random.default_rng(4242)
    checked = passed_random = tried_random = 0
    for world, lights in (scenes.mirror_room(), scenes.mirror_bench(ntri=1500, ncuboid=100), scenes.adversarial()):
        sc = api.Scene(world, lights)
        assert L.rt_scene_prepare_host(sc.handle, 320, 200) >= 0
        leaves = [sc.leaf(g) for g in range(sc.num_leaves)]
        pts = np.array([lf[6:9] for lf in leaves if np.all(np.isfinite(lf[6:15]))])
        lo, hi = pts.min(axis=0) - 1, pts.max(axis=0) + 1
        for kind in (0, 1):
            sph, ids, rows, frame = (C.c_float * 4)(), (C.c_int32 * 32)(), (C.c_uint16 * 1024)(), (C.c_float * 4)()
            nblk = L.rt_scene_get_general_block(sc.handle, kind, 0, sph, ids)
            for b in range(max(nblk, 0)):
                L.rt_scene_get_general_block(sc.handle, kind, b, sph, ids)
                assert L.rt_scene_get_general_rows(sc.handle, kind, b, frame, rows) == nblk
                A = np.frombuffer(bytes(rows), np.float16).reshape(32, 32).astype(np.float32)
                centre, radius = np.array(frame[:3], np.float64), float(frame[3])
                if np.isfinite(sph[3]) and np.isfinite(radius):               # the frame (the super-block's sphere) holds the block's
                    assert np.linalg.norm(np.array(sph[:3], np.float64) - centre) + sph[3] <= radius * (1 + 1e-6)
                for m, g in enumerate(ids):
                    if g < 0:
                        assert A[m, 0] < 0 and not A[m, 1:].any()          # padding: never
                        continue
                    leaf = leaves[g]
                    p0, e1, e2 = np.array(leaf[6:9]), np.array(leaf[9:12]), np.array(leaf[12:15])
                    if not np.all(np.isfinite(np.concatenate([p0, e1, e2]))):
                        assert A[m, 0] > 0 and not A[m, 1:].any()          # no sphere: always
                        continue
                    for trial in range(6):
                        s_, t_ = rng.uniform(0, 1, 2)
                        if kind == 1 and s_ + t_ > 1:
                            s_, t_ = 1 - s_, 1 - t_
                        if kind == 1:
                            p = p0 + s_ * e1 + t_ * e2
                        else:
                            ww, hh, wh = e1 @ e1, e2 @ e2, e1 @ e2
                            det = ww * hh - wh * wh
                            if det == 0:
                                break
                            a_, b_ = s_ * ww, t_ * hh
                            p = p0 + ((a_ * hh - b_ * wh) / det) * e1 + ((b_ * ww - a_ * wh) / det) * e2
                        o = (rng.uniform(lo, hi), p + rng.normal(size=3) * 1e-3, p + rng.normal(size=3) * 0.3,
                             rng.uniform(lo, hi) * 3)[trial % 4]
                        d = (p - o) * rng.choice([1.0, -1.0]) * rng.uniform(0.5, 2.0)      # the line, either way along it
                        if not (d @ d > 1e-12):
                            continue
                        F = np.float32((A[m] * _general_B(o, d, centre, radius)).sum(dtype=np.float32))
                        assert F > 2.0 ** 20 * 1e-5, (kind, b, m, g, trial, float(F))
                        checked += 1
                # rays that have nothing to do with the block
                for trial in range(4):
                    o = rng.uniform(lo, hi)
                    d = rng.normal(size=3)
                    F = (A * _general_B(o, d, centre, radius)[None, :]).sum(axis=1)
                    real = np.array([g >= 0 for g in ids])
                    tried_random += int(real.sum())
                    passed_random += int((F[real] > 0).sum())
        sc.close()
    assert checked > 3000
    assert passed_random < 0.05 * tried_random

