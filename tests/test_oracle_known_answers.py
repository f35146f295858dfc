"""The oracle against the known-answer table of SURVEY.md §4 (values derived from the reference's
formulas, Main.hs:114, 506-533, 948) and against hand-checkable properties of the reference."""
import ctypes as C
import math

import numpy as np

import oracle_binding as ob
from raytrace_amd import api, scenes


def _cam(w, h):
    out = (C.c_double * 5)()
    ob.lib().orc_camera_params(w, h, out)
    return list(out)


def _ray(w, h, x, y):
    out = (C.c_double * 9)()
    ob.lib().orc_camera_ray(w, h, x, y, out)
    return list(out)


def test_eps():
    eps = ob.lib().orc_eps()
    assert eps.hex() == "0x0.fd1d7d505cd02p-1022"
    assert 1.0 / eps == 4.545454545454545e+307


def test_camera_1024():
    sx, sy, dx, dy, d = _cam(1024, 1024)
    assert (sx, sy, dx, dy) == (0.0009765625, -0.0009765625, -0.5, 0.5)
    assert d == -0.49999999999999994 == math.tan(math.pi / 4) * -0.5
    r = _ray(1024, 1024, 0, 0)
    assert r[3:6] == [-0.5773502691896258, 0.5773502691896258, -0.5773502691896257]
    r = _ray(1024, 1024, 1023, 1023)
    assert r[3:6] == [0.5769736548527653, -0.5769736548527653, -0.5781027618094243]


def test_camera_centre_pixel_is_clamped():
    eps = ob.lib().orc_eps()
    r = _ray(1024, 1024, 512, 512)
    assert r[0:3] == [0.0, 0.0, 0.0]
    assert r[3:6] == [eps, eps, -1.0]          # mkray: 0 -> +eps (Main.hs:112)
    assert r[6] == 1.0 / eps


def test_camera_4k_has_the_h_over_w_squared_quirk():
    sx, sy, dx, dy, d = _cam(3840, 2160)
    assert (sx, sy, dx, dy) == (0.00026041666666666666, -0.000146484375, -0.5, 0.158203125)
    r = _ray(3840, 2160, 0, 0)
    assert r[3:6] == [-0.6900470080000889, 0.2183351861250281, -0.6900470080000888]


def test_grid_steps():
    out = (C.c_double * 16)()
    n = ob.lib().orc_grid_steps(-1.0, 1.0, 0.2, out, 16)
    assert n == 10
    assert list(out)[:10] == [-1.0, -0.8, -0.6000000000000001, -0.4000000000000001, -0.20000000000000007,
                              -5.551115123125783e-17, 0.19999999999999996, 0.39999999999999997, 0.6, 0.8]


def test_default_scene_has_201_leaves_and_python_builder_agrees_with_c_builder():
    L = ob.lib()
    world, lights = scenes.default_scene()
    s = ob.OracleScene(world, lights)
    assert s.num_leaves == 201
    assert lights[0].pos == (5.01, 93.0, -20.0) and (lights[0].diffuse, lights[0].specular) == (0.3, 0.6)
    a = s.render(64, 64)
    h2 = C.c_void_p(L.orc_scene_new())
    root = L.orc_linear_interpolation_gaussian(h2, -1.0, -1.0, 1.0, 1.0, 0.2, (C.c_double * 3)(0.01, -7, -25), 10.0)
    rgb = np.zeros((64, 64, 3), np.uint8)
    hit = np.zeros((64, 64), np.int32)
    assert L.orc_render(h2, root, 64, 64, 4, 1, 0, 0, 64, 2, 1, rgb.ctypes.data, hit.ctypes.data, None, None) == 0
    assert (rgb == a["rgb"]).all() and (hit == a["hit_id"]).all()
    L.orc_scene_free(h2)


def test_single_rectangle_shading_by_hand():
    """One axis-aligned rectangle facing the camera, ambient only: colour = round(f * diffuse)."""
    world = api.rectangle(api.orange, (0, 0, -2), (1, 0, 0), (0, 1, 0))
    o = ob.OracleScene(world, api.ambient(0.5)).render(33, 33)
    assert o["hit_id"][16, 16] == 0 and o["hit_id"][0, 0] == -1
    # 0.5*255 = 127.5 -> 128 (half-even), 0.5*134 = 67, 0
    assert tuple(o["rgb"][16, 16]) == (128, 67, 0)
    assert tuple(o["rgb"][0, 0]) == (0, 0, 0)


def test_round_half_even_and_saturating_add():
    world = api.rectangle(api.Material((255, 5, 3), (0, 0, 0)), (0, 0, -2), (1, 0, 0), (0, 1, 0))
    # 0.5*5 = 2.5 -> 2 ; 0.5*3 = 1.5 -> 2 ; two ambient lights: 128+128 saturates to 255
    o = ob.OracleScene(world, api.ambient(0.5) + api.ambient(0.5)).render(9, 9)
    assert tuple(o["rgb"][4, 4]) == (255, 4, 4)


def test_ties_go_to_the_leftmost_leaf():
    a = api.rectangle(api.red, (0, 0, -2), (1, 0, 0), (0, 1, 0))
    b = api.rectangle(api.green, (0, 0, -2), (1, 0, 0), (0, 1, 0))
    o = ob.OracleScene(api.mconcat([a, b]), api.ambient(1.0)).render(9, 9, flags=True)
    assert o["hit_id"][4, 4] == 0 and tuple(o["rgb"][4, 4]) == (255, 0, 0)
    assert o["flags"][4, 4] & ob.FLAG_TIE


def test_bounding_box_of_a_rotated_rectangle_clips_it():
    """Main.hs:243-245 boxes only two opposite corners: a rectangle whose edge vectors share the xz plane
    is clipped by the mappend guard (Main.hs:209).  The `intersection` scene shows it."""
    world, lights = scenes.intersection()
    s = ob.OracleScene(world, lights)
    a, b = s.render(96, 96), s.render(96, 96, use_bbox=0)
    rect = s.num_leaves - 1
    clipped = (a["hit_id"] != b["hit_id"])
    assert clipped.sum() > 50
    assert (b["hit_id"][clipped] == rect).all() and (a["hit_id"][clipped] != rect).all()


def test_spp_extension_is_deterministic_and_spp1_is_the_reference_ray():
    world, lights = scenes.cubes()
    s = ob.OracleScene(world, lights)
    a, b = s.render(48, 27, spp=4, seed=3), s.render(48, 27, spp=4, seed=3)
    c = s.render(48, 27, spp=4, seed=4)
    assert (a["rgb"] == b["rgb"]).all() and (a["rgb"] != c["rgb"]).any()
    assert (s.render(48, 27, spp=1, seed=99)["rgb"] == s.render(48, 27)["rgb"]).all()


def test_row_bands_and_threads_do_not_change_the_image():
    world, lights = scenes.stacked_cubes()
    s = ob.OracleScene(world, lights)
    full = s.render(40, 30, nthreads=1)
    parts = [s.render(40, 30, row0=r0, nrows=n, nthreads=3) for r0, n in [(0, 7), (7, 13), (20, 10)]]
    assert (np.concatenate([p["rgb"] for p in parts]) == full["rgb"]).all()
    assert (np.concatenate([p["hit_id"] for p in parts]) == full["hit_id"]).all()


def test_talk_dialect_main2():
    """src/Talk.hs main2 (Talk.hs:261-277) under Talk.hs's own rules, 512x512 as in the talk.
    Known by hand: the centre ray (0,0,-1) has its x and y clamped to eps by mkray (Talk.hs:35), so for the floor
    (normal (0,1,0)) frac = eps / -1 and planeLineIsect culls it (Talk.hs:106): black, no hit.  Pixel (128, 400) sees
    only the blue floor (leaf 6: six cube faces come first): red = green = 0, and blue is at least the ambient term
    round(0.1 * 255) = 26 (round half even of 25.5)."""
    import oracle_binding as ob
    from raytrace_amd import scenes
    world, lights = scenes.talkhs_main2()
    o = ob.OracleScene(world, lights).render(512, 512)
    assert o["hit_id"][256, 256] == -1 and (o["rgb"][256, 256] == 0).all()
    assert o["hit_id"][400, 128] == 6
    r, g, b = (int(v) for v in o["rgb"][400, 128])
    assert r == 0 and g == 0 and 26 <= b <= 255
    assert set(np.unique(o["hit_id"])) <= {-1, 0, 1, 2, 3, 4, 5, 6}


# ---- the oracle's one non-literal rewriting: mconcat as a list node with suffix boxes -----------------------------
# Main.hs:196-217 defines only mempty and mappend; `mconcat xs` is the Monoid default foldr mappend mempty, i.e. the chain
# x0 <> (x1 <> (... <> mempty)) whose every node guards with the union box of its suffix.  oracle.c evaluates a list node
# instead (stop at the first failed suffix guard, strict < keeps the left child on ties) and the GPU's guard chains share
# that reading.  Here every mconcat of every reference scene is built as the literal chain of mappend nodes and must
# render the same hit IDs, distances and colours — including `intersection`, where guards clip rectangles.
import pytest


@pytest.mark.parametrize("name,wh,depth", [
    ("default_scene", (96, 96), 4), ("planes", (80, 60), 4), ("axes", (64, 64), 4), ("cubes", (96, 54), 4),
    ("stacked_cubes", (96, 96), 4), ("triangle_example", (90, 50), 4), ("spec_test", (64, 64), 4),
    ("intersection", (128, 128), 4), ("cylinder_axes", (80, 64), 4), ("spheres", (96, 96), 4), ("mirror_room", (96, 72), 5),
    ("adversarial", (96, 72), 4), ("viewport_plane", (64, 48), 4)])
def test_literal_mappend_chain_renders_like_the_list_node(name, wh, depth):
    world, lights = getattr(scenes, name)()
    w, h = wh
    a = ob.OracleScene(world, lights)
    b = ob.OracleScene(world, lights, literal_mconcat=True)
    assert a.num_leaves == b.num_leaves
    ra = a.render(w, h, depth=depth, want_t=True)
    rb = b.render(w, h, depth=depth, want_t=True)
    assert np.array_equal(ra["hit_id"], rb["hit_id"])
    assert np.array_equal(ra["t"], rb["t"])
    assert np.array_equal(ra["rgb"], rb["rgb"])
    if name == "intersection":      # the guards do clip here: the test is not vacuous
        off = ob.OracleScene(world, lights).render(w, h, depth=depth, use_bbox=0)
        assert (off["hit_id"] != ra["hit_id"]).any()


def test_literal_chain_on_a_synthetic_scene_with_cuboids_and_cylinders():
    for world, lights in (scenes.synthetic(ntri=300, ncuboid=30), scenes.synthetic_cylinders(ncyl=6, ncuboid=15)):
        a = ob.OracleScene(world, lights)
        b = ob.OracleScene(world, lights, literal_mconcat=True)
        ra, rb = a.render(96, 54, want_t=True), b.render(96, 54, want_t=True)
        assert np.array_equal(ra["hit_id"], rb["hit_id"]) and np.array_equal(ra["t"], rb["t"]) and np.array_equal(ra["rgb"], rb["rgb"])


# ---- hand-derived pixels (closed form in tests/known_answers.py): they pin the oracle independently of its author's reading
import known_answers as ka


@pytest.mark.parametrize("name", sorted(ka.CASES))
def test_hand_derived_pixel(name):
    world, lights, w, h, (row0, nrows), _ = ka.CASES[name]()
    r = ob.OracleScene(world, lights).render(w, h, depth=4, row0=row0, nrows=nrows)
    ka.check(name, r["hit_id"], r["rgb"], row0)
