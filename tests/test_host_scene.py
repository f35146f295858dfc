"""Host logic of the product library, no GPU: the C ABI exports every declared symbol, the scene builder
produces the reference's leaf order and constants, guard chains follow the mconcat/mappend tree, errors
are reported like the reference reports them, and rendering without a GPU fails loudly."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

from raytrace_amd import _lib, api, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "raytrace_gpu.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(rt_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 25
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/raytrace_gpu.h but not exported"
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)


def test_camera_params_match_known_answers():
    out = (C.c_double * 5)()
    assert _lib.lib().rt_camera_params(3840, 2160, out) == 0
    assert list(out) == [0.00026041666666666666, -0.000146484375, -0.5, 0.158203125, math.tan(math.pi / 4) * -0.5]


def test_cuboid_is_six_rectangles_in_reference_order_with_outward_normals():
    sc = api.Scene(api.colorcuboid([api.red, api.green], (1.0, 2.0, -5.0), 2.0, 4.0, 6.0), [])
    assert sc.num_leaves == 6
    normals = [tuple(sc.leaf(i)[2:5]) for i in range(6)]
    # top, bottom, front, back, left(+x), right(-x): Main.hs:308-320
    assert normals == [(0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1), (1, 0, 0), (-1, 0, 0)]
    assert [sc.leaf(i)[19] for i in range(6)] == [0, 1, 2, 3, 4, 5]       # one material record per face
    top = sc.leaf(0)
    assert top[0] == 0 and top[5] == -4.0                     # kind rect; d = -(point . normal) = -(2+2)
    assert tuple(top[6:9]) == (0.0, 4.0, -2.0)                # point' = point - w/2 - h/2
    assert (top[15], top[16]) == (4.0, 36.0)                  # ww, hh
    sc.close()


def test_triangle_constants_follow_the_reference():
    pa, pb, pc = (0.0, 1.0, -4.0), (-1.0, 0.0, -4.0), (1.0, 0.0, -4.0)
    sc = api.Scene(api.triangle(api.orange, pa, pb, pc), [])
    leaf = sc.leaf(0)
    u, v = np.subtract(pb, pa), np.subtract(pc, pa)
    uv, uu, vv = float(u @ v), float(u @ u), float(v @ v)
    assert leaf[0] == 1 and tuple(leaf[15:19]) == (uv, uu, vv, uv * uv - uu * vv)
    n = np.cross(u, v)
    assert np.allclose(leaf[2:5], n / np.linalg.norm(n), rtol=0, atol=1e-16)
    sc.close()


def test_cylinder_layout_and_quirks():
    n = 8
    sc = api.Scene(api.cylinder(api.blue, api.red, api.green, (0.0, -2.0, -10.0), n, 1.0, 5.0), [])
    assert sc.num_leaves == 3 * n
    kinds = [sc.leaf(i)[0] for i in range(3 * n)]
    assert kinds == [1] * (2 * n) + [0] * n                   # bottom tris, top tris, mantle rects
    modes = [sc.leaf(i)[1] for i in range(3 * n)]
    assert modes == [0] * (2 * n) + [1] * n                   # mantle rectangles lerp their normal
    # botPoint = point - (0, -h/2, 0) = the TOP centre (Main.hs:735): bottom triangles start at y = -1.5
    assert sc.leaf(0)[7] == -1.5 and sc.leaf(n)[7] == -1.5
    sc.close()


def test_leaf_order_is_depth_first_and_matches_the_oracle_count():
    import oracle_binding as ob
    for name, mk in scenes.REFERENCE_SCENES.items():
        world, lights = mk()
        sc = api.Scene(world, lights)
        assert sc.num_leaves == ob.OracleScene(world, lights).num_leaves, name
        sc.close()


def test_guard_chains_follow_the_tree():
    L = _lib.lib()
    a = api.rectangle(api.red, (0, 0, -2), (1, 0, 0), (0, 1, 0))
    b = api.rectangle(api.green, (3, 0, -2), (1, 0, 0), (0, 1, 0))
    # a single leaf has no mappend node above it
    sc = api.Scene(a, [])
    assert L.rt_scene_prepare_host(sc.handle, 64, 64) == 0
    assert L.rt_scene_get_guards(sc.handle, 0, None, 0) == 0
    sc.close()
    # mconcat [a, b]: a sits under box(a u b); b under box(a u b) and box(b)
    sc = api.Scene(api.mconcat([a, b]), [])
    assert L.rt_scene_prepare_host(sc.handle, 64, 64) == 2
    out = (C.c_double * 12)()
    assert L.rt_scene_get_guards(sc.handle, 0, out, 2) == 1
    assert list(out)[:6] == [-0.5, -0.5, -2.0, 3.5, 0.5, -2.0]
    assert L.rt_scene_get_guards(sc.handle, 1, out, 2) == 2
    assert list(out)[:6] == [2.5, -0.5, -2.0, 3.5, 0.5, -2.0] and list(out)[6:] == [-0.5, -0.5, -2.0, 3.5, 0.5, -2.0]
    sc.close()
    # a `mappend` b: one node, both children under the same box
    sc = api.Scene(a + b, [])
    assert L.rt_scene_prepare_host(sc.handle, 64, 64) == 1
    assert L.rt_scene_get_guards(sc.handle, 0, None, 0) == 1 and L.rt_scene_get_guards(sc.handle, 1, None, 0) == 1
    sc.close()


def test_a_light_may_close_over_a_world_of_its_own():
    """`pointLight world kd ks pos` (Main.hs:467-474): a shadow world is a second shape tree whose leaves live behind the
    traced world's (hit IDs unchanged), have guard chains of their own, and appear only in the streams of the lights that
    name it — never in the eye's, another light's or the reflected rays'."""
    L = _lib.lib()
    wall = api.rectangle(api.red, (0.0, 0.0, -5.0), (4.0, 0.0, 0.0), (0.0, 4.0, 0.0))
    plate = api.rectangle(api.white, (0.0, 1.5, -3.0), (1.0, 0.0, 0.0), (0.0, 0.0, 1.0))
    lid = api.rectangle(api.white, (0.0, 2.5, -3.0), (1.0, 0.0, 0.0), (0.0, 0.0, 1.0))
    other = api.mconcat([plate, lid])
    lights = api.pointLight(other, 0.5, 0.6, (0.0, 3.0, -1.0)) + api.pointLight(0.5, 0.6, (1.0, 3.0, -1.0)) + api.pointLight(other, 0.1, 0.1, (2.0, 3.0, -1.0))
    sc = api.Scene(wall, lights)
    assert L.rt_scene_prepare_host(sc.handle, 64, 64) == 2            # the traced world is one leaf: no box; mconcat [plate, lid]: two
    assert sc.num_leaves == 3                                          # `other` was emitted once for its two lights
    assert L.rt_scene_get_guards(sc.handle, 0, None, 0) == 0 and L.rt_scene_get_guards(sc.handle, 1, None, 0) == 1 and L.rt_scene_get_guards(sc.handle, 2, None, 0) == 2
    row = (C.c_uint16 * 8)()
    has = [[L.rt_scene_get_mfma_row(sc.handle, apex, leaf, row) >= 0 for leaf in range(3)] for apex in range(4)]
    assert has == [[True, False, False], [False, True, True], [True, False, False], [False, True, True]]
    sc.close()
    # the traced world comes first; brackets and worlds nest properly; a light names a world that exists
    h = C.c_void_p()
    assert L.rt_scene_create(C.byref(h)) == 0
    m = api._cmat(api.red)
    rect = lambda: L.rt_add_rectangle(h, C.byref(m), _lib.vec3((0, 0, -5)), _lib.vec3((1, 0, 0)), _lib.vec3((0, 1, 0)))
    assert rect() == 0
    assert L.rt_end_shadow_world(h) < 0 and b"no shadow world" in L.rt_last_error()
    assert L.rt_begin_mconcat(h) == 0 and L.rt_begin_shadow_world(h) < 0 and L.rt_end_mconcat(h) == 0
    assert L.rt_begin_shadow_world(h) == 1 and L.rt_begin_shadow_world(h) < 0
    assert rect() == 1 and L.rt_end_shadow_world(h) == 0
    assert rect() < 0 and b"traced world must be complete" in L.rt_last_error()
    assert L.rt_begin_mconcat(h) < 0
    assert L.rt_add_point_light_world(h, 2, 0.5, 0.5, _lib.vec3((0, 0, 0))) < 0 and b"no such world" in L.rt_last_error()
    assert L.rt_add_point_light_world(h, 1, 0.5, 0.5, _lib.vec3((0, 0, 0))) == 0 and L.rt_add_point_light_world(h, 0, 0.5, 0.5, _lib.vec3((0, 0, 0))) == 1
    assert L.rt_begin_shadow_world(h) == 2 and L.rt_scene_prepare_host(h, 64, 64) < 0 and b"still open" in L.rt_last_error()
    L.rt_scene_destroy(h)


def test_rotated_rectangle_box_uses_two_corners_only():
    """Main.hs:243-245: the `intersection` scene's floor rectangle gets the too-small box."""
    L = _lib.lib()
    world, lights = scenes.intersection()
    sc = api.Scene(world, lights)
    L.rt_scene_prepare_host(sc.handle, 64, 64)
    out = (C.c_double * 6)()
    assert L.rt_scene_get_guards(sc.handle, sc.num_leaves - 1, out, 1) == 2
    assert list(out) == [-1.5, -1.0, -3.5, 1.5, -1.0, -2.5]
    sc.close()


def test_errors_are_reported_not_thrown():
    L = _lib.lib()
    h = C.c_void_p()
    assert L.rt_scene_create(C.byref(h)) == 0
    assert L.rt_add_cuboid(h, (_lib.rt_material * 1)(), 0, _lib.vec3((0, 0, 0)), 1.0, 1.0, 1.0) == -1
    assert b"list of materials must not be empty" in L.rt_last_error()          # Main.hs:302
    assert L.rt_add_rectangle(h, None, None, None, None) == -1
    assert L.rt_end_mconcat(h) == -1
    assert L.rt_begin_mappend(h) == 0
    assert L.rt_add_rectangle(h, C.byref(_lib.rt_material()), _lib.vec3((0, 0, -1)), _lib.vec3((1, 0, 0)), _lib.vec3((0, 1, 0))) == 0
    assert L.rt_end_mappend(h) == -1 and b"exactly two" in L.rt_last_error()
    assert L.rt_scene_prepare_host(h, 16, 16) == -1 and b"still open" in L.rt_last_error()
    assert L.rt_set_normal_mode(h, 0, 5, 0, None, 0) == -1
    assert L.rt_scene_get_leaf(h, 7, (C.c_double * 20)()) == -1
    L.rt_scene_destroy(h)
    with pytest.raises(ValueError):
        api.colorcuboid([], (0, 0, 0), 1, 1, 1)


def test_write_bmp(tmp_path):
    img = np.zeros((3, 5, 3), np.uint8)
    img[0, 0] = (255, 0, 0)
    img[2, 4] = (0, 0, 255)
    p = tmp_path / "t.bmp"
    api.saveBmpImage(p, img)
    raw = p.read_bytes()
    assert raw[:2] == b"BM" and len(raw) == 54 + 16 * 3
    assert int.from_bytes(raw[18:22], "little") == 5 and int.from_bytes(raw[22:26], "little") == 3
    assert raw[28] == 24
    # bottom-up BGR: the last row of the file is image row 0, whose first pixel is red
    assert raw[54 + 32:54 + 35] == bytes([0, 0, 255])
    assert raw[54 + 12:54 + 15] == bytes([255, 0, 0])


def test_render_without_a_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    world, lights = scenes.planes()
    sc = api.Scene(world, lights)
    with pytest.raises(RuntimeError, match="no HIP device|hip"):
        sc.render(16, 16)
    sc.close()


def test_page_locked_allocator_without_a_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    L = _lib.lib()
    assert not L.rt_host_alloc(4096) and b"no HIP device" in L.rt_last_error()
    L.rt_host_free(None)                                   # freeing NULL is a no-op
    with pytest.raises(RuntimeError, match="no HIP device"):
        api.host_buffer((4, 4), np.uint8)
    with pytest.raises(ValueError, match="output array"):    # shape is checked before any device work
        api.Scene(*scenes.planes()).render(16, 16, out_hit=np.zeros((16, 16), np.int64))


def test_cpp_host_mirror_builds_and_fails_loudly_without_gpu(tmp_path):
    """include/raytrace.hpp + examples/main.cpp (the reference's `main`) compile against the C ABI."""
    import subprocess
    import torch
    subprocess.run(["make", "-C", os.path.join(ROOT, "examples")], check=True, capture_output=True)
    exe = os.path.join(ROOT, "examples", "main")
    assert os.path.exists(exe)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the GPU test runs the binary")
    r = subprocess.run([exe, "32", "32", str(tmp_path / "t.bmp")], capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stderr


def test_limits_and_dialect_errors():
    """Misuse of the Talk.hs dialect is reported as an error before any device work.  (Point lights and trace depth have
    no limit worth the name any more: lights are worked off 32 at a time, bounce levels up to 64.)"""
    L = _lib.lib()
    h = C.c_void_p()
    assert L.rt_scene_create(C.byref(h)) == 0
    assert L.rt_scene_set_dialect(h, 7) == -1
    assert L.rt_scene_set_dialect(h, 1) == 0
    m = _lib.rt_material()
    assert L.rt_add_triangle(h, C.byref(m), _lib.vec3((0, 0, -1)), _lib.vec3((1, 0, -1)), _lib.vec3((0, 1, -1))) == -3   # RT_ERR_UNSUPPORTED
    assert b"rectangles only" in L.rt_last_error()
    assert L.rt_add_rectangle(h, C.byref(m), _lib.vec3((0, 0, -1)), _lib.vec3((1, 0, 0)), _lib.vec3((0, 1, 0))) == 0
    assert L.rt_scene_set_dialect(h, 0) == -1 and b"before the first shape" in L.rt_last_error()
    # Talk.hs:185-194: no bounding boxes -> no guard chain
    assert L.rt_scene_prepare_host(h, 16, 16) == 0
    assert L.rt_scene_get_guards(h, 0, None, 0) == 0
    L.rt_scene_destroy(h)


def test_c_band_splitter_agrees_with_split_rows():
    """rt_split_rows (what rt_render_multi cuts its bands with) is a port of distributed.split_rows: same bands."""
    from raytrace_amd.distributed import split_rows
    L = _lib.lib()
    rng = np.random.default_rng(11)
    for trial in range(300):
        ncost = int(rng.integers(1, 70))
        cost = rng.uniform(0, 1, ncost) ** int(rng.integers(1, 4))
        if trial % 5 == 0:
            cost[rng.random(ncost) < 0.5] = 0.0                 # plateaus of the cumulative cost
        if trial % 17 == 0:
            cost[:] = 0.0                                       # all zero: equal bands
        h = int(rng.integers(1, 20000))
        n = int(rng.integers(1, 17))
        align = int(rng.choice([1, 8, 16]))
        r0, nr = (C.c_int * n)(), (C.c_int * n)()
        assert L.rt_split_rows((C.c_double * ncost)(*cost), ncost, h, n, align, r0, nr) == 0
        assert list(zip(r0, nr)) == split_rows(cost, h, n, align), (trial, ncost, h, n, align)
        assert sum(nr) == h and r0[0] == 0 and all(r0[k] + nr[k] == r0[k + 1] for k in range(n - 1))
        # ... and with unequal shares (the collecting device's larger band)
        weights = rng.uniform(0.5, 2.0, n)
        assert L.rt_split_rows_weighted((C.c_double * ncost)(*cost), ncost, h, n, align, (C.c_double * n)(*weights), r0, nr) == 0
        assert list(zip(r0, nr)) == split_rows(cost, h, n, align, weights=weights), (trial, "weighted", ncost, h, n, align)
        assert sum(nr) == h and r0[0] == 0 and all(r0[k] + nr[k] == r0[k + 1] for k in range(n - 1))
    assert L.rt_split_rows((C.c_double * 2)(1.0, -1.0), 2, 10, 2, 8, (C.c_int * 2)(), (C.c_int * 2)()) == -1
    assert L.rt_split_rows_weighted((C.c_double * 2)(1.0, 1.0), 2, 64, 2, 8, (C.c_double * 2)(1.0, 0.0), (C.c_int * 2)(), (C.c_int * 2)()) == -1


def test_scene_options_and_the_cross_check_build():
    """Options are per scene (nothing is read from the environment); the vector-ALU kernel family exists in the
    cross-check build only."""
    sc = api.Scene(*scenes.planes())
    sc.set_option(_lib.RT_OPT_CULL, 0).set_option(_lib.RT_OPT_CULL, 1).set_option(_lib.RT_OPT_HOST_BANDS, 3)
    with pytest.raises(RuntimeError, match="cross-check build"):
        sc.set_option(_lib.RT_OPT_FILTER, 1)
    with pytest.raises(RuntimeError, match="unknown option"):
        sc.set_option(99, 1)
    with pytest.raises(RuntimeError, match=">= 0"):
        sc.set_option(_lib.RT_OPT_HOST_BANDS, -1)
    sc.set_option(_lib.RT_OPT_LANE, 1).set_option(_lib.RT_OPT_LANE, 0).set_option(_lib.RT_OPT_LANE, -1)
    sc.set_option(_lib.RT_OPT_CONTINUE, 1).set_option(_lib.RT_OPT_CONTINUE, 0).set_option(_lib.RT_OPT_HISTORY, 0).set_option(_lib.RT_OPT_HISTORY, 1)
    assert sc.bytes_per_pixel == 4 and sc.set_option(_lib.RT_OPT_RGB24, 1).bytes_per_pixel == 3 and sc.set_option(_lib.RT_OPT_RGB24, 0).bytes_per_pixel == 4
    with pytest.raises(RuntimeError, match="RT_OPT_LANE"):
        sc.set_option(_lib.RT_OPT_LANE, 2)
    sc.close()
    V = _lib.valu_lib()
    sv = api.Scene(*scenes.planes(), lib=V)
    sv.set_option(_lib.RT_OPT_FILTER, 1).set_option(_lib.RT_OPT_FILTER, 0)
    sv.close()
    assert set(_lib.SYMBOLS) <= {n for n in _lib.SYMBOLS if hasattr(V, n)}


def test_changing_the_image_size_rebuilds_the_eye_records_only_when_the_camera_bound_grows():
    L = _lib.lib()
    world, lights = scenes.stacked_cubes()
    sc = api.Scene(world, lights)
    h = sc.handle

    def records(apex):
        out = (C.c_float * 16)()
        rows = []
        for leaf in range(sc.num_leaves):
            assert L.rt_scene_get_filter_record(h, apex, leaf, out) == 0
            rows.append(tuple(out))
        return rows

    assert L.rt_scene_prepare_host(h, 64, 64) > 0
    eye, light = records(0), records(1)
    for w, hh in ((3840, 2160), (1024, 1024), (16, 16), (100, 40)):      # all inside the bound the records were built for
        assert L.rt_scene_prepare_host(h, w, hh) > 0
        assert records(0) == eye and records(1) == light
    assert L.rt_scene_prepare_host(h, 40, 200) > 0                       # a tall image: |posY| goes up to 2.5
    eye2 = records(0)
    assert records(1) == light                                           # the light apexes do not depend on the camera
    assert eye2 != eye
    margins = [(a[9], b[9]) for a, b in zip(eye, eye2)]
    assert all(b <= a for a, b in margins)                               # only the margins widen
    assert all(a[:9] == b[:9] for a, b in zip(eye, eye2))
    sc.close()


def test_normal_callback_registration():
    L = _lib.lib()
    seen = []
    world, lights = api.linearInterpolation(lambda x, y: 0.0, lambda x, y: (seen.append((x, y)) or (0.0, 1.0, 0.0)),
                                            (-1.0, -1.0), (1.0, 1.0), 0.5, (0.0, -2.0, -10.0), 2.0)
    sc = api.Scene(world, lights)
    assert sc.num_leaves == 16 * 2 + 1
    assert [sc.leaf(i)[1] for i in range(sc.num_leaves)] == [3.0] * 32 + [0.0]      # RT_NORMAL_CALLBACK, then the floor
    assert seen == []                                                               # only a render calls it
    assert L.rt_set_normal_callback(sc.handle, 0, 99, (C.c_double * 4)(1, 0, 0, 0), _lib.NORMAL_FN(lambda *a: None), None) == -1
    assert b"out of bounds" in L.rt_last_error()
    sc.close()
    with pytest.raises(TypeError):
        api.linearInterpolation(lambda x, y: 0.0, "sinc", (-1.0, -1.0), (1.0, 1.0), 0.5, (0.0, -2.0, -10.0), 2.0)


def test_host_index_is_the_same_every_time_it_is_built():
    """The apexes' record streams are built on one host thread each (rt_scene.cpp: build_filter_records); they write
    disjoint slices, so two builds of the same scene must agree byte for byte: rows, records, block cones, light order."""
    import ctypes as C
    L = _lib.lib()
    world, lights = scenes.synthetic(ntri=600, ncuboid=60)
    snaps = []
    for _ in range(2):
        sc = api.Scene(world, lights)
        assert L.rt_scene_prepare_host(sc.handle, 640, 360) >= 0
        row, rec, cone = (C.c_uint16 * 8)(), (C.c_float * 16)(), (C.c_float * 8)()
        snap = []
        for apex in range(5):
            groups = L.rt_scene_get_block_cone(sc.handle, apex, 0, cone)
            for g in range(groups + 2):
                L.rt_scene_get_block_cone(sc.handle, apex, g, cone)
                snap.append(bytes(cone))
            for leaf in range(0, sc.num_leaves, 7):
                pos = L.rt_scene_get_mfma_row(sc.handle, apex, leaf, row)
                assert L.rt_scene_get_filter_record(sc.handle, apex, leaf, rec) == 0
                snap.append((pos, bytes(row), bytes(rec)))
        snaps.append(snap)
        sc.close()
    assert snaps[0] == snaps[1]


def test_product_libraries_are_not_measurement_builds():
    """Result-changing measurement switches (RT_NO_GUARDS, RT_X_SCANONLY, ...) compile only with -DRT_MEASUREMENT_BUILD, and
    rt_version() names them: the libraries the tests load must be clean."""
    for L in (_lib.lib(), _lib.valu_lib()):
        v = L.rt_version().decode()
        assert v.startswith("raytrace_amd") and "MEASUREMENT" not in v and "RT_" not in v, v
    assert "cross-check" in _lib.valu_lib().rt_version().decode() and "cross-check" not in _lib.lib().rt_version().decode()
