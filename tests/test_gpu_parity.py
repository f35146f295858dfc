"""GPU parity: the HIP render path (through the C ABI) against the CPU oracle on the same scenes.

Bars (BASELINE.json north_star): hit-ID buffers bit-exact except pixels within 1e-5 of an
intersection tie; RGB within +-1 LSB per channel on >= 99.9 % of pixels.  The GPU path decides
hits in IEEE double with the reference's expression order (bounding-box guards of Main.hs:209
included), so in practice raw mismatch counts are 0 and colours are bit-identical; the counts
against the oracle run WITHOUT the guards are printed to show how many pixels the guards decide.
"""
import numpy as np
import pytest

import oracle_binding as ob
from raytrace_amd import api, scenes

pytestmark = pytest.mark.gpu

SIZES = {"default": (256, 256), "planes": (160, 120), "axes": (200, 200), "cubes": (256, 144),
         "stacked_cubes": (256, 256), "triangle_example": (250, 130), "spec_test": (192, 192),
         "intersection": (256, 256), "cylinder_axes": (200, 160), "talk_main2": (128, 128), "talkhs_main2": (512, 512), "mirror_room": (256, 192), "spheres": (256, 256),
         "sine_cosine": (240, 160), "viewport_plane": (200, 160)}


def compare(world, lights, w, h, depth=4, label=""):
    osc = ob.OracleScene(world, lights)
    ref = osc.render(w, h, depth=depth, flags=True)
    ref_nobb = osc.render(w, h, depth=depth, use_bbox=0)
    sc = api.Scene(world, lights)
    assert sc.num_leaves == osc.num_leaves
    rgba, hit = sc.render(w, h, depth=depth)
    st = sc.stats()
    sc.close()
    rgb = rgba[:, :, :3]
    assert (rgba[:, :, 3] == 255).all()
    # 1. exact equality with the guard-free oracle
    n_id_nobb = int((hit != ref_nobb["hit_id"]).sum())
    n_rgb_nobb = int((rgb != ref_nobb["rgb"]).any(axis=2).sum())
    # 2. the reference-faithful (guarded) oracle
    bad = hit != ref["hit_id"]
    exempt = (ref["flags"] & ob.FLAG_TIE) != 0
    n_raw, n_masked = int(bad.sum()), int((bad & ~exempt).sum())
    diff = np.abs(rgb.astype(np.int16) - ref["rgb"].astype(np.int16)).max(axis=2)
    frac_1lsb = float((diff <= 1).mean())
    print(f"[{label} {w}x{h}] leaves={sc.num_leaves if False else osc.num_leaves} hit-ID mismatches raw={n_raw} masked={n_masked} "
          f"(no-bbox oracle: {n_id_nobb}); rgb exact={float((diff == 0).mean()):.6f} within1={frac_1lsb:.6f} "
          f"(no-bbox oracle rgb mismatches: {n_rgb_nobb}); stats={st}")
    assert n_masked == 0
    assert frac_1lsb >= 0.999
    return ref, hit, rgb


@pytest.mark.parametrize("name", sorted(SIZES))
def test_reference_scene(gpu_lib, name):
    world, lights = scenes.REFERENCE_SCENES[name]()
    w, h = SIZES[name]
    compare(world, lights, w, h, label=name)


def test_default_scene_1024(gpu_lib):
    """configs[0]: the Main.hs default world/lights at the reference's own 1024x1024."""
    world, lights = scenes.default_scene()
    compare(world, lights, 1024, 1024, label="default")


def test_synthetic_small(gpu_lib):
    """configs[2] scene generator at a size the oracle finishes in seconds."""
    world, lights = scenes.synthetic(ntri=600, ncuboid=60)
    compare(world, lights, 320, 180, label="synthetic-600+60")


def test_synthetic_cylinders_small(gpu_lib):
    world, lights = scenes.synthetic_cylinders(ncyl=12, ncuboid=30)
    compare(world, lights, 320, 180, label="cylinders-12+30")


def test_row_band_equals_full(gpu_lib):
    """Row bands (the multi-GPU partition) reproduce the full image exactly."""
    world, lights = scenes.stacked_cubes()
    sc = api.Scene(world, lights)
    w, h = 200, 150
    full_rgba, full_hit = sc.render(w, h)
    parts = [sc.render(w, h, row0=r0, nrows=n) for r0, n in [(0, 37), (37, 50), (87, 63)]]
    assert (np.concatenate([p[0] for p in parts]) == full_rgba).all()
    assert (np.concatenate([p[1] for p in parts]) == full_hit).all()
    sc.close()


def test_depth0_and_empty(gpu_lib):
    world, lights = scenes.planes()
    sc = api.Scene(world, lights)
    rgba, hit = sc.render(64, 64, depth=0)          # trace 0 = black, Main.hs:542
    assert (hit == -1).all() and (rgba[:, :, :3] == 0).all()
    rgba, hit = sc.render(64, 64, nrows=0)
    assert rgba.shape == (0, 64, 4)
    sc.close()
    sc = api.Scene(api.mempty, lights)               # empty world
    rgba, hit = sc.render(40, 24)
    assert (hit == -1).all() and (rgba[:, :, :3] == 0).all()
    sc.close()


# ---------------------------------------------------------------------------------------------
# golden fixtures (no oracle library needed), supersampling, full-size checks

import glob  # noqa: E402
import os  # noqa: E402

from test_oracle_golden import FILES as GOLDEN_FILES, scene_of  # noqa: E402


@pytest.mark.parametrize("path", GOLDEN_FILES, ids=[os.path.basename(f) for f in GOLDEN_FILES])
def test_golden_fixture(gpu_lib, path):
    world, lights, w, h, kw = scene_of(path)
    g = np.load(path)
    sc = api.Scene(world, lights)
    rgba, hit = sc.render(w, h, depth=4, **kw)
    sc.close()
    assert (hit == g["hit_id"]).all()
    diff = np.abs(rgba[:, :, :3].astype(np.int16) - g["rgb"].astype(np.int16)).max(axis=2)
    assert (diff <= 1).mean() >= 0.999 and (diff == 0).mean() >= 0.999


def test_supersampling_matches_oracle(gpu_lib):
    """spp > 1 is this repo's extension: same jitter hash and same integer mean in both."""
    world, lights = scenes.cylinder_axes()
    osc = ob.OracleScene(world, lights)
    sc = api.Scene(world, lights)
    for spp, seed in [(4, 1), (5, 2), (16, 20260355)]:
        ref = osc.render(120, 90, spp=spp, seed=seed)
        rgba, hit = sc.render(120, 90, spp=spp, seed=seed)
        assert (hit == ref["hit_id"]).all()
        assert (rgba[:, :, :3] == ref["rgb"]).all(), f"spp={spp}"
    sc.close()


def test_default_scene_4k_full_frame_against_oracle(gpu_lib):
    """configs[1]: the Main.hs scene at 3840x2160, every pixel against the oracle."""
    world, lights = scenes.default_scene()
    w, h = 3840, 2160
    sc = api.Scene(world, lights)
    rgba, hit = sc.render(w, h)
    rgba2, hit2 = sc.render(w, h)
    assert (rgba == rgba2).all() and (hit == hit2).all()           # deterministic
    osc = ob.OracleScene(world, lights)
    ref = osc.render(w, h, nthreads=16)                              # the whole frame: every pixel
    assert (hit == ref["hit_id"]).all()
    assert (rgba[:, :, :3] == ref["rgb"]).all()
    # row bands (the multi-GPU partition) tile the frame exactly, also across the 16 Mi-pixel chunking
    top = sc.render(w, h, row0=0, nrows=1000)
    bot = sc.render(w, h, row0=1000, nrows=1160)
    assert (np.concatenate([top[1], bot[1]]) == hit).all() and (np.concatenate([top[0], bot[0]]) == rgba).all()
    sc.close()


def test_synthetic_4k_rows_against_oracle(gpu_lib):
    """configs[2]: 10k triangles + 1k cuboids, 4 lights, 3840x2160 (the benchmark frame, as bench.py renders it): 64
    stratified rows on the oracle — 48 evenly spaced and 16 drawn at random; every row crosses the columns beneath all
    four lights (x = +-6 at z = -10 and -26), where the shadow bundles are widest and the block test has the most to lose."""
    world, lights = scenes.synthetic()
    w, h = 3840, 2160
    sc = api.Scene(world, lights)
    rgba, hit = sc.render(w, h)
    st = sc.stats()
    # k_shadow_mfma dispatches the lights in the order of their measured cost, which changes from the first render (no
    # history) to the next ones: the order must not show in the result
    for _ in range(3):
        again = sc.render(w, h)
        assert (again[0] == rgba).all() and (again[1] == hit).all()
    sc.close()
    rows = sorted(set(int(r) for r in np.linspace(3, h - 4, 48)) | set(int(r) for r in np.random.default_rng(2160).integers(0, h, 16)))
    assert len(rows) >= 60
    assert st["primary_rays"] == w * h and st["bounce_rays"] == 0
    # counted in the kernels: every primary hit costs at least one record test and one exact evaluation
    assert st["record_tests"] >= st["exact_evals"] >= int((hit >= 0).sum())
    osc = ob.OracleScene(world, lights)
    # one oracle call per row (the oracle threads over rows), 16 calls at a time: ctypes releases the GIL
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(16) as pool:
        refs = list(pool.map(lambda r0: osc.render(w, h, row0=r0, nrows=1, nthreads=1), rows))
    for r0, ref in zip(rows, refs):
        assert (hit[r0] == ref["hit_id"][0]).all(), f"row {r0}"
        assert (rgba[r0, :, :3] == ref["rgb"][0]).all(), f"row {r0}"
    assert 0.3 < (hit >= 0).mean() < 0.95


def test_block_test_on_and_off_agree_on_whole_frames(gpu_lib):
    """The block test (stage -1) and the guard proofs are conservative filters whose failure would be a silently
    dropped hit.  With RT_OPT_CULL = 0 every wave scans every block and evaluates every guard box, so whole frames of
    the two must agree bit for bit: the benchmark frame, a band of the 16384-wide frame, the default scene, the
    adversarial scene, and a band of configs[4] at its full 7680 x 4320 x 16 spp (two of its rows also on the oracle)."""
    from raytrace_amd import _lib
    cases = [("synthetic 4K", scenes.synthetic(), 3840, 2160, 1, 0, 2160),
             ("synthetic 16K band", scenes.synthetic(), 16384, 16384, 1, 8000, 64),
             ("default 4K", scenes.default_scene(), 3840, 2160, 1, 0, 2160),
             ("adversarial", scenes.adversarial(), 1600, 1200, 1, 0, 1200),
             ("cylinders 8K x 16 spp band", scenes.synthetic_cylinders(), 7680, 4320, 16, 2000, 64)]
    for label, (world, lights), w, h, spp, row0, nrows in cases:
        sc = api.Scene(world, lights)
        on = sc.render(w, h, spp=spp, seed=20260355, row0=row0, nrows=nrows)
        st_on = sc.stats()
        sc.set_option(_lib.RT_OPT_CULL, 0)
        off = sc.render(w, h, spp=spp, seed=20260355, row0=row0, nrows=nrows)
        st_off = sc.stats()
        sc.close()
        assert (on[1] == off[1]).all(), label
        assert (on[0] == off[0]).all(), label
        assert st_on["shadow_rays"] == st_off["shadow_rays"]
        assert st_on["mfma_blocks_primary"] <= st_off["mfma_blocks_primary"]
        if label.startswith("synthetic"):                                        # the switch does switch something
            assert 20 * st_on["mfma_blocks_primary"] < st_off["mfma_blocks_primary"]
        if spp == 16:
            osc = ob.OracleScene(world, lights)
            for r in (row0 + 5, row0 + 40):
                ref = osc.render(w, h, spp=spp, seed=20260355, row0=r, nrows=1, nthreads=16)
                assert (on[1][r - row0] == ref["hit_id"][0]).all() and (on[0][r - row0, :, :3] == ref["rgb"][0]).all(), f"{label} row {r}"


def test_arbitrary_fnorm_closures(gpu_lib):
    """linearInterpolation with `Just fnorm` for fnorms other than main's gaussian (Main.hs:586-605: the sinc and the
    sine-cosine surfaces): the closure is a host callback on both sides, so results stay bit-identical."""
    for (world, lights), w, h in ((scenes.sine_cosine(True), 200, 120), (scenes.sinc(True), 160, 100)):
        ref, hit, rgb = compare(world, lights, w, h, label="fnorm-callback")
        assert (rgb == ref["rgb"]).all()
    world, lights = scenes.sine_cosine(True)
    plain = api.Scene(*scenes.sine_cosine(False)).render(200, 120)
    with_n = api.Scene(world, lights).render(200, 120)
    assert (plain[1] == with_n[1]).all() and (plain[0] != with_n[0]).any()      # same hits, other shading
    osc = ob.OracleScene(world, lights)
    sc = api.Scene(world, lights)
    ref = osc.render(90, 60, spp=4, seed=5, nthreads=2)
    rgba, hit = sc.render(90, 60, spp=4, seed=5)
    assert (hit == ref["hit_id"]).all() and (rgba[:, :, :3] == ref["rgb"]).all()
    rgba2, hit2 = sc.render(90, 60, spp=4, seed=5, row0=13, nrows=30)            # a band, as a rank renders it
    assert (hit2 == hit[13:43]).all() and (rgba2 == rgba[13:43]).all()
    sc.close()


def test_reflection_off_normals_of_any_length(gpu_lib):
    """An `fnorm` may return a vector of any length (Main.hs:956-965 only fixes its sign), and trace reflects about it as
    it is (Main.hs:550): the reflected rays are then many times longer than a unit vector.  The bounce levels' FP32 stages
    and their "behind the nearest hit so far" skip must not depend on that length."""
    import math
    from raytrace_amd.api import CallbackNormals, black, green, mconcat, orange, rectangle, triangle

    def fnorm(x, y):
        return (0.6 * math.sin(3 * x), 4.0 + math.cos(2 * y), 0.6 * math.sin(x + y))
    mirror = black.replace(reflectivity=0.8, specular=(255, 255, 255), specularity=20)
    a, b, c, d = (-20, -7, 0), (20, -7, 0), (-20, -7, -40), (20, -7, -40)
    two = mconcat([triangle(mirror, a, b, c), triangle(mirror, b, d, c)])
    clutter, lights = scenes.synthetic(ntri=300, ncuboid=40)
    walls = [rectangle(green, (0, 0, -32), (40, 0, 0), (0, 40, 0)), rectangle(orange, (0, 8, -18), (40, 0, 0), (0, 0, 40))]
    world = mconcat([clutter] + walls + [CallbackNormals(two, 0.25, (0.0, -7.0, -20.0), fnorm)])
    ref, hit, rgb = compare(world, lights, 240, 135, depth=4, label="long-normals")
    assert (rgb == ref["rgb"]).all()
    sc = api.Scene(world, lights)
    sc.render(240, 135, depth=4)
    assert sc.stats()["bounce_rays"] > 2000               # (the floor does reflect)
    sc.close()


def test_whole_and_fractional_specular_exponents(gpu_lib):
    """`x ** specularity` (Main.hs:483): k_final takes whole-number exponents by squaring and everything else through pow;
    both against the oracle's libm pow, to the suite's bar (colours within 1 LSB, 99.9 % equal; in practice all equal)."""
    from raytrace_amd.api import mconcat, rectangle, sphere, triangle, white, red, blue
    shapes = []
    for k, sp in enumerate((0.0, 0.5, 1.0, 2.5, 7.0, 33.3, 100.0, 400.0, 4096.0, 5000.0)):
        m = (red if k % 2 else blue).replace(specular=(255, 255, 255), specularity=sp)
        x = -4.5 + k
        shapes.append(rectangle(m, (x, 0.0, -6.0 - 0.1 * k), (0.8, 0, 0.3), (0, 1.6, 0.2)))
        shapes.append(triangle(m, (x - 0.4, -2.0, -5.0), (x + 0.4, -2.0, -5.5), (x, -1.0, -5.2)))
    shapes.append(sphere(white.replace(specular=(255, 255, 255), specularity=12.5), (0.0, 2.2, -6.0), 0.8))
    lights = api.mconcat_lights([api.pointLight(0.5, 0.9, (0.5, 1.0, -1.0)), api.pointLight(0.2, 0.7, (-3.0, -1.0, -2.0)), api.ambient(0.1)])
    ref, hit, rgb = compare(mconcat(shapes), lights, 400, 240, label="specular-exponents")
    assert (np.abs(rgb.astype(np.int16) - ref["rgb"].astype(np.int16)) <= 1).all()
    assert (ref["rgb"].max(axis=2) > 200).sum() > 200             # (highlights are in the picture)


def test_render_multi_equals_single_render(gpu_lib):
    """rt_render_multi on one device: 1, 3 and 5 bands of measured-cost rows, rendered in place and through the
    staging buffer + peer copy path, give the bytes of rt_render; statistics cover the whole frame."""
    from raytrace_amd import _lib
    world, lights = scenes.synthetic(ntri=600, ncuboid=60)
    sc = api.Scene(world, lights)
    w, h = 1000, 531
    ref_rgba, ref_hit = sc.render(w, h)
    one = sc.stats()
    for devices in ([0], [0, 0, 0], [0] * 5):
        for stage in (0, 1):
            sc.set_option(_lib.RT_OPT_MULTI_STAGE, stage)
            rgba, hit = sc.render_multi(w, h, devices)
            assert (rgba == ref_rgba).all() and (hit == ref_hit).all(), (devices, stage)
            st = sc.stats()
            assert st["primary_rays"] == w * h and st["shadow_rays"] == one["shadow_rays"]
    rgba, none = sc.render_multi(w, h, [0, 0], want_hit=False)
    assert none is None and (rgba == ref_rgba).all()
    fresh = api.Scene(world, lights)            # first call of a scene: the lanes' work buffers grow between the pieces of the frame
    rgba, hit = fresh.render_multi(w, h, [0, 0])
    st = fresh.stats()
    assert (rgba == ref_rgba).all() and (hit == ref_hit).all()
    assert st["primary_rays"] == w * h and st["shadow_rays"] == one["shadow_rays"]
    fresh.close()
    ref4 = sc.render(w, h, spp=4, seed=3)
    got4 = sc.render_multi(w, h, [0, 0, 0], spp=4, seed=3)
    assert (ref4[0] == got4[0]).all() and (ref4[1] == got4[1]).all()
    with pytest.raises(RuntimeError, match="out of range"):
        sc.render_multi(w, h, [0, 99])
    sc.close()
    world, lights = scenes.mirror_room()
    sc = api.Scene(world, lights).set_option(_lib.RT_OPT_MULTI_STAGE, 1)
    ref = sc.render(300, 200, depth=4)
    got = sc.render_multi(300, 200, [0, 0, 0], depth=4)
    assert (ref[0] == got[0]).all() and (ref[1] == got[1]).all()
    sc.close()


def test_two_threads_render_distinct_scenes(gpu_lib):
    """Distinct scenes may be rendered from different threads at once: every (scene, device) pair has
    its own streams, events and work buffers."""
    import threading
    jobs = [(scenes.stacked_cubes(), 320, 240), (scenes.cylinder_axes(), 300, 200), (scenes.synthetic(ntri=600, ncuboid=60), 640, 360)]
    refs = [api.Scene(w_, l_).render(w, h) for (w_, l_), w, h in jobs]
    errors = []

    def work(k):
        try:
            (world, lights), w, h = jobs[k]
            sc = api.Scene(world, lights)
            for _ in range(6):
                rgba, hit = sc.render(w, h)
                if not ((rgba == refs[k][0]).all() and (hit == refs[k][1]).all()):
                    errors.append(f"thread {k}: frame differs")
            sc.close()
        except Exception as e:                               # noqa: BLE001
            errors.append(f"thread {k}: {e!r}")

    threads = [threading.Thread(target=work, args=(k,)) for k in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_reflection_depths(gpu_lib):
    """trace n (Main.hs:541-557): n = 1 has no reflection, n = 2..5 add one bounce each; all must match the oracle."""
    world, lights = scenes.mirror_room()
    osc = ob.OracleScene(world, lights)
    sc = api.Scene(world, lights)
    prev = None
    for depth in (1, 2, 3, 5):
        ref = osc.render(160, 120, depth=depth)
        rgba, hit = sc.render(160, 120, depth=depth)
        assert (hit == ref["hit_id"]).all()
        assert (rgba[:, :, :3] == ref["rgb"]).all(), f"depth {depth}"
        if depth == 2:
            assert (rgba != prev).any()          # the first bounce is visible in this scene
        prev = rgba
    # supersampled + reflective
    ref = osc.render(96, 72, depth=4, spp=4, seed=11)
    rgba, hit = sc.render(96, 72, depth=4, spp=4, seed=11)
    assert (rgba[:, :, :3] == ref["rgb"]).all() and (hit == ref["hit_id"]).all()
    sc.close()


def test_cpp_main_matches_oracle(gpu_lib, tmp_path):
    """examples/main (C++ mirror of the reference's main, Main.hs:569-621): BMP pixels and hit IDs vs the oracle."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "main")
    w, h = 200, 120
    bmp, ids = tmp_path / "trace.bmp", tmp_path / "ids.bin"
    r = subprocess.run([exe, str(w), str(h), str(bmp), str(ids)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    raw = bmp.read_bytes()
    row = (3 * w + 3) & ~3
    px = np.frombuffer(raw, np.uint8, offset=54).reshape(h, row)[:, :3 * w].reshape(h, w, 3)[::-1, :, ::-1]
    world, lights = scenes.default_scene()
    ref = ob.OracleScene(world, lights).render(w, h)
    assert (px == ref["rgb"]).all()
    assert (np.fromfile(ids, np.int32).reshape(h, w) == ref["hit_id"]).all()


def test_matrix_core_and_valu_filters_agree_full_size(gpu_lib):
    """Size-independent check at the benchmark sizes: the two first stages are independent conservative filters
    (bounding cones on the matrix cores; per-leaf linear forms on the vector ALU, in libraytrace_gpu_valu.so), and the exact
    FP64 stage decides — so whole frames must agree bit for bit, hit IDs and colours."""
    from raytrace_amd import _lib
    V = _lib.valu_lib()                     # the cross-check build: same sources + the vector-ALU kernel family
    _lib.check(V.rt_init(0), "rt_init", V)
    cases = [(scenes.synthetic(), 3840, 2160, 1), (scenes.synthetic_cylinders(), 1920, 1080, 4), (scenes.default_scene(), 3840, 2160, 1),
             (scenes.adversarial(), 2048, 1536, 1)]
    for (world, lights), w, h, spp in cases:
        sc = api.Scene(world, lights)
        mfma = sc.render(w, h, spp=spp, seed=7)
        sc.close()
        sc = api.Scene(world, lights, lib=V).set_option(_lib.RT_OPT_FILTER, 1)
        valu = sc.render(w, h, spp=spp, seed=7)
        sc.close()
        assert (mfma[1] == valu[1]).all()
        assert (mfma[0] == valu[0]).all()
    with pytest.raises(RuntimeError):       # the product library has no second kernel family
        api.Scene(*scenes.planes()).set_option(_lib.RT_OPT_FILTER, 1)


def test_adversarial_scene(gpu_lib):
    """scenes.adversarial: leaves and lights where a conservative filter is most likely to lose a hit (huge, sliver,
    tiny, degenerate and non-orthogonal leaves; leaves crossing the viewport and eye planes; lights on, in and inside
    leaves; a mirror and a sphere).  Whole images against the oracle, also supersampled."""
    world, lights = scenes.adversarial()
    compare(world, lights, 640, 480, label="adversarial")
    compare(world, lights, 333, 517, depth=2, label="adversarial-odd-size")
    osc = ob.OracleScene(world, lights)
    sc = api.Scene(world, lights)
    ref = osc.render(400, 300, spp=4, seed=3)
    rgba, hit = sc.render(400, 300, spp=4, seed=3)
    sc.close()
    assert (hit == ref["hit_id"]).all() and (rgba[:, :, :3] == ref["rgb"]).all()


def test_16k_image_row_bands_against_oracle(gpu_lib):
    """configs[3]: the synthetic scene at 16384 x 16384, the size that is split into row bands across GPUs.  Bands of
    the image (not the whole 1 GB frame) are rendered exactly as a rank would render them and single rows are checked
    against the oracle; overlapping bands must agree where they overlap."""
    world, lights = scenes.synthetic()
    w = h = 16384
    sc = api.Scene(world, lights)
    a_rgba, a_hit = sc.render(w, h, row0=8000, nrows=200)
    b_rgba, b_hit = sc.render(w, h, row0=8100, nrows=61)            # ragged band: 61 rows is no multiple of the tile height
    sc.close()
    assert (a_hit[100:161] == b_hit).all() and (a_rgba[100:161] == b_rgba).all()
    osc = ob.OracleScene(world, lights)
    for r0 in (8000, 8160):
        ref = osc.render(w, h, row0=r0, nrows=1, nthreads=16)
        assert (a_hit[r0 - 8000] == ref["hit_id"][0]).all(), f"row {r0}"
        assert (a_rgba[r0 - 8000, :, :3] == ref["rgb"][0]).all(), f"row {r0}"


def test_many_point_lights(gpu_lib):
    """Monoid Light (Main.hs:436-441) has no limit: 32 lights fill one word of the per-pixel lit mask, 33 and 70 need
    two and three groups."""
    world, _ = scenes.stacked_cubes()
    rng = np.random.default_rng(5)
    for n in (32, 33, 70):
        lights = api.mconcat_lights([api.pointLight(2.0 / n, 1.6 / n, tuple(map(float, rng.uniform((-8, 2, -12), (8, 9, 2))))) for _ in range(n)]
                                    + [api.ambient(0.05)])
        compare(world, lights, 160, 120, label=f"{n}-lights")
    # ambient lights in between, a mirror scene (the bounce levels take the groups too), supersampling
    world, _ = scenes.mirror_room()
    lights = []
    for k in range(40):
        lights += api.pointLight(0.04, 0.03, tuple(map(float, rng.uniform((-6, -3, -14), (6, 4, -2)))))
        if k % 13 == 0:
            lights += api.ambient(0.02)
    compare(world, lights, 120, 90, depth=3, label="40-lights-mirror")


def test_deep_reflection(gpu_lib):
    """trace n for n beyond the 8 levels the first version kept: a mirror in front of the camera and one behind it,
    depth 12, 20 and 30 (Main.hs:541-557)."""
    mirror = api.black.replace(reflectivity=0.9, specular=(255, 255, 255), specularity=100)
    world = api.mconcat([
        api.rectangle(mirror, (0, 0, -10), (10, 0, 0), (0, 10, 0)),           # normal +z: faces the camera
        api.rectangle(mirror, (0, 0, 2), (0, 10, 0), (10, 0, 0)),             # normal -z: behind the camera
        api.rectangle(api.orange, (0, -3, -4), (0, 0, 14), (12, 0, 0)),
        api.colorcube([api.red, api.green, api.blue], (1.5, -2, -6), 1.2)])
    lights = api.mconcat_lights([api.pointLight(0.7, 0.4, (0, 2.5, -6)), api.ambient(0.15)])
    osc = ob.OracleScene(world, lights)
    sc = api.Scene(world, lights)
    prev = None
    for depth in (8, 12, 20, 30):
        ref = osc.render(160, 100, depth=depth)
        rgba, hit = sc.render(160, 100, depth=depth)
        assert (hit == ref["hit_id"]).all() and (rgba[:, :, :3] == ref["rgb"]).all(), f"depth {depth}"
        if depth in (12, 20, 30):
            assert (rgba != prev).any()              # the deeper levels are visible between the mirrors
        prev = rgba
    assert sc.stats()["bounce_rays"] > 0
    sc.close()


def test_fuzzed_scenes_against_oracle(gpu_lib):
    """Differential fuzzing (tools/fuzz_parity.py): random scenes of every primitive — rectangles with arbitrary, also
    non-orthogonal, edge vectors among them — in random mconcat/mappend trees, lights anywhere, random sizes, depths and
    sample counts.  (A 5 000-scene campaign is logged in profiles/r01_fuzz_parity.txt.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for mode, n in (("main", 150), ("valu", 60), ("talk", 80), ("extreme", 100), ("big", 40), ("height", 40)):
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), str(n), "42", mode], capture_output=True, text=True)
        assert r.returncode == 0, mode + ": " + r.stdout[-2000:] + r.stderr[-2000:]
        assert f"{n} scenes (seed 42" in r.stdout and "), 0 with a mismatch" in r.stdout


def test_band_larger_than_one_chunk(gpu_lib):
    """A band of more than 16 Mi pixels is rendered in chunks of rows (rt_render_device): rows on both sides of the chunk
    boundary, and the last ragged chunk, against the oracle; a mirror scene too (its chunks are a quarter of the size)."""
    from raytrace_amd import _lib
    world, lights = scenes.default_scene()
    w, h = 16384, 4000
    sc = api.Scene(world, lights).set_option(_lib.RT_OPT_HOST_BANDS, 1)   # rt_render would otherwise cut the frame into copy bands smaller than a chunk
    rgba, hit = sc.render(w, h, row0=1500, nrows=1100)             # 18.0 M pixels: chunks of 1024 and 76 rows
    sc.close()
    osc = ob.OracleScene(world, lights)
    for r in (1500, 2523, 2524, 2599):
        ref = osc.render(w, h, row0=r, nrows=1, nthreads=16)
        assert (hit[r - 1500] == ref["hit_id"][0]).all() and (rgba[r - 1500, :, :3] == ref["rgb"][0]).all(), f"row {r}"
    world, lights = scenes.mirror_room()
    w, h = 4096, 1100                                              # 4.5 M pixels: chunks of 1024 and 76 rows when reflective
    sc = api.Scene(world, lights).set_option(_lib.RT_OPT_HOST_BANDS, 1)
    rgba, hit = sc.render(w, h, depth=3)
    sc.close()
    osc = ob.OracleScene(world, lights)
    for r in (0, 1023, 1024, 1099):
        ref = osc.render(w, h, depth=3, row0=r, nrows=1, nthreads=16)
        assert (hit[r] == ref["hit_id"][0]).all() and (rgba[r, :, :3] == ref["rgb"][0]).all(), f"mirror row {r}"


def test_two_live_scenes_interleaved(gpu_lib):
    """Two scenes alive at once, rendered alternately at different sizes: each keeps its own device copy and records."""
    a = api.Scene(*scenes.stacked_cubes())
    b = api.Scene(*scenes.cylinder_axes())
    ra = ob.OracleScene(*scenes.stacked_cubes())
    rb = ob.OracleScene(*scenes.cylinder_axes())
    for (w, h) in ((96, 64), (130, 70), (96, 64)):
        for sc, orc in ((a, ra), (b, rb)):
            rgba, hit = sc.render(w, h)
            ref = orc.render(w, h)
            assert (hit == ref["hit_id"]).all() and (rgba[:, :, :3] == ref["rgb"]).all()
    a.close(); b.close()


def test_host_render_copy_bands_and_page_locked_buffers(gpu_lib):
    """rt_render enqueues the frame as row bands whose copies to the caller's memory overlap the next band's render:
    every band count gives the same bytes, into pageable and into rt_host_alloc'ed memory, with either output alone,
    with supersampling and with a reflective scene, and the statistics describe the whole frame."""
    world, lights = scenes.synthetic(ntri=600, ncuboid=60)
    sc = api.Scene(world, lights)
    from raytrace_amd import _lib
    w, h = 1000, 531                                   # ragged: 531 rows are not a multiple of the 8-row tiles
    sc.set_option(_lib.RT_OPT_HOST_BANDS, 1)
    ref_rgba, ref_hit = sc.render(w, h)
    one = sc.stats()
    ref4_rgba, ref4_hit = sc.render(w, h, spp=4, seed=7)
    osc = ob.OracleScene(world, lights)
    ref = osc.render(w, h, row0=300, nrows=8, nthreads=16)
    assert (ref_hit[300:308] == ref["hit_id"]).all() and (ref_rgba[300:308, :, :3] == ref["rgb"]).all()
    pin_rgba, pin_hit = api.host_buffer((h, w, 4), np.uint8), api.host_buffer((h, w), np.int32)
    for bands in (2, 3, 5, 8, 100):
        sc.set_option(_lib.RT_OPT_HOST_BANDS, bands)
        rgba, hit = sc.render(w, h)
        st = sc.stats()
        assert (rgba == ref_rgba).all() and (hit == ref_hit).all(), bands
        # (the exact-evaluation count depends on which shadow rays share a wave: an occluded ray stops early)
        assert st["primary_rays"] == w * h and st["shadow_rays"] == one["shadow_rays"]
        assert abs(st["exact_evals"] - one["exact_evals"]) <= one["exact_evals"] // 100
        pin_rgba[:] = 0; pin_hit[:] = 0
        sc.render(w, h, out_rgba=pin_rgba, out_hit=pin_hit)
        assert (pin_rgba == ref_rgba).all() and (pin_hit == ref_hit).all(), bands
        rgba, none = sc.render(w, h, want_hit=False)
        assert none is None and (rgba == ref_rgba).all()
        none, hit = sc.render(w, h, want_rgba=False)
        assert none is None and (hit == ref_hit).all()
        rgba, hit = sc.render(w, h, spp=4, seed=7)
        assert (rgba == ref4_rgba).all() and (hit == ref4_hit).all(), bands
        rgba, hit = sc.render(w, h, row0=77, nrows=201)
        assert (rgba == ref_rgba[77:278]).all() and (hit == ref_hit[77:278]).all(), bands
    sc.close()
    world, lights = scenes.mirror_room()
    sc = api.Scene(world, lights)
    sc.set_option(_lib.RT_OPT_HOST_BANDS, 1)
    ref_rgba, ref_hit = sc.render(300, 200, depth=4)
    sc.set_option(_lib.RT_OPT_HOST_BANDS, 4)
    rgba, hit = sc.render(300, 200, depth=4)
    assert (rgba == ref_rgba).all() and (hit == ref_hit).all()
    sc.close()
    with pytest.raises(ValueError):
        api.Scene(*scenes.stacked_cubes()).render(64, 64, out_rgba=np.zeros((64, 64, 3), np.uint8))


def test_pieces_on_alternating_lanes(gpu_lib):
    """RT_OPT_LANE: a frame rendered as pieces that alternate between the scene's two lanes (work buffers + stream each, so
    the pieces overlap on the device) has the bytes of the one-call frame, its statistics cover every piece, and the
    lanes go on working across frames of different sizes; a reflective scene (bounce levels in both lanes) too."""
    from raytrace_amd import _lib
    from hip_buffers import DeviceBuffer, hip
    for (world, lights), (w, h), depth in ((scenes.synthetic(ntri=600, ncuboid=60), (1000, 531), 4), (scenes.mirror_room(), (300, 200), 4),
                                           (scenes.synthetic(ntri=600, ncuboid=60), (640, 96), 4)):
        sc = api.Scene(world, lights)
        ref_rgba, ref_hit = sc.render(w, h, depth=depth)
        one = sc.stats()
        rgba = DeviceBuffer((h, w, 4), np.uint8, fill=0x5a)
        hit = DeviceBuffer((h, w), np.int32, fill=0x5a)
        for cuts in ((0, 200, h), (0, 8, 16, 24, 120, h), (0, h)):
            cuts = sorted(set(min(c, h) for c in cuts))
            for i, (r0, r1) in enumerate(zip(cuts[:-1], cuts[1:])):
                sc.set_option(_lib.RT_OPT_LANE, i & 1).set_option(_lib.RT_OPT_CONTINUE, int(i > 0))
                sc.render_device(w, h, rgba.ptr + 4 * w * r0, hit.ptr + 4 * w * r0, depth=depth, row0=r0, nrows=r1 - r0)
            sc.set_option(_lib.RT_OPT_LANE, -1).set_option(_lib.RT_OPT_CONTINUE, 0)
            st = sc.stats()                                  # (waits for both lanes)
            assert hip().hipDeviceSynchronize() == 0
            assert (rgba.numpy() == ref_rgba).all() and (hit.numpy() == ref_hit).all(), cuts
            assert st["primary_rays"] == w * h and st["shadow_rays"] == one["shadow_rays"] and st["bounce_rays"] == one["bounce_rays"], cuts
        with pytest.raises(RuntimeError, match="RT_OPT_LANE"):
            sc.set_option(_lib.RT_OPT_LANE, 2)
        sc.close()
    # left to itself (RT_OPT_LANE -1) rt_render_device splits a band of 12 Mi pixels or more over the two lanes: same bytes
    # and the same counts as the unsplit call, on the library's stream and on a stream of the caller's, full frame and band
    # (12 Mi pixels for the plain pipeline, 3 Mi with bounce levels: the room of mirrors at 2048 x 1600)
    for (world, lights), (w, h), band in ((scenes.synthetic(ntri=600, ncuboid=60), (4096, 3200), (24, 3171)), (scenes.mirror_room(), (2048, 1600), (8, 1579))):
        sc = api.Scene(world, lights)
        rgba, hit = DeviceBuffer((h, w, 4), np.uint8, fill=0x5a), DeviceBuffer((h, w), np.int32, fill=0x5a)
        sc.set_option(_lib.RT_OPT_LANE, 0)
        sc.render_device(w, h, rgba.ptr, hit.ptr)
        one = sc.stats()
        ref_rgba, ref_hit = rgba.numpy(), hit.numpy()
        sc.set_option(_lib.RT_OPT_LANE, -1)
        for row0, nrows in ((0, h), band):
            rgba2, hit2 = DeviceBuffer((nrows, w, 4), np.uint8, fill=0x5a), DeviceBuffer((nrows, w), np.int32, fill=0x5a)
            sc.render_device(w, h, rgba2.ptr, hit2.ptr, row0=row0, nrows=nrows)
            st = sc.stats()
            assert (rgba2.numpy() == ref_rgba[row0:row0 + nrows]).all() and (hit2.numpy() == ref_hit[row0:row0 + nrows]).all()
            assert st["primary_rays"] == w * nrows and (nrows < h or (st["shadow_rays"] == one["shadow_rays"] and st["bounce_rays"] == one["bounce_rays"]))
            sc.render_device(w, h, None, hit2.ptr, row0=row0, nrows=nrows)                 # hit IDs only, split too
            assert (hit2.numpy() == ref_hit[row0:row0 + nrows]).all()
        sc.close()


def test_consecutive_frames_on_alternating_lanes(gpu_lib):
    """A sequence of frames with two in flight: consecutive frames go to alternating lanes of the scene, each lane on a
    stream of its own and into a frame buffer of its own, nothing waits in between (bench.py: two_frames_in_flight).  Every
    frame has the bytes of the frame rendered alone — also while the other lane renders another size or trace depth, and
    from the second frame of a lane on, when its waves are dispatched by that lane's own history."""
    import ctypes as C
    from raytrace_amd import _lib
    from hip_buffers import DeviceBuffer, hip
    H = hip()
    streams = [C.c_void_p(), C.c_void_p()]
    for st in streams:
        assert H.hipStreamCreateWithFlags(C.byref(st), 1) == 0                      # hipStreamNonBlocking
    for (world, lights), sizes in ((scenes.synthetic(ntri=600, ncuboid=60), ((1000, 531, 4), (640, 400, 4))),
                                   (scenes.mirror_room(), ((300, 200, 4), (300, 200, 2)))):
        sc = api.Scene(world, lights)
        refs = [sc.render(w, h, depth=d) for w, h, d in sizes]
        bufs = [(DeviceBuffer((h, w, 4), np.uint8, fill=0x5a), DeviceBuffer((h, w), np.int32, fill=0x5a)) for w, h, _ in sizes]
        assert H.hipDeviceSynchronize() == 0                                       # (the fills ran on the null stream)
        for frame in range(8):                                                     # lane 0 renders sizes[0], lane 1 sizes[1]
            ln = frame & 1
            w, h, d = sizes[ln]
            sc.render_device(w, h, bufs[ln][0].ptr, bufs[ln][1].ptr, depth=d, stream=streams[ln].value, lane=ln, cont=False)
            if frame in (1, 5, 7):                                                 # after two, six and eight frames: both lanes' last frames
                for st in streams:
                    assert H.hipStreamSynchronize(st) == 0
                for k in range(2):
                    assert (bufs[k][0].numpy() == refs[k][0]).all() and (bufs[k][1].numpy() == refs[k][1]).all(), (frame, k)
                    bufs[k][0].fill(0x5a); bufs[k][1].fill(0x5a)
                assert H.hipDeviceSynchronize() == 0
        sc.set_option(_lib.RT_OPT_LANE, -1).set_option(_lib.RT_OPT_CONTINUE, 0)
        sc.close()
    for st in streams:
        assert H.hipStreamDestroy(st) == 0


def test_lights_with_worlds_of_their_own(gpu_lib):
    """`pointLight world kd ks pos` (Main.hs:467-474): a light tests its shadow rays against the Shape it closed over.  Three
    lights — one over a world of big cuboids, spheres and cylinders that is never traced, one over the traced world,
    one over mempty — on a scene of small triangles and cuboids, and on the room of mirrors (reflected rays must not see
    the lights' worlds either): whole frames against the oracle, bands, 4 samples, the cull off, the Talk.hs dialect."""
    from raytrace_amd import _lib
    rng = np.random.default_rng(77)
    big = [api.cuboid(api.white, tuple(map(float, rng.uniform((-6, -1, -24), (6, 5, -8)))), 2.5, 1.5, 2.0) for _ in range(6)]
    big += [api.sphere(api.white, tuple(map(float, rng.uniform((-6, 0, -20), (6, 5, -8)))), 1.2) for _ in range(4)]
    big += [api.cylinder(api.white, api.white, api.white, (2.0, 2.0, -12.0), 9, 2.0, 1.1)]
    occluders = api.mconcat(big)
    for (world, base_lights), (w, h), depth in ((scenes.synthetic(ntri=500, ncuboid=50), (320, 200), 4), (scenes.mirror_room(), (240, 160), 4)):
        lights = api.mconcat_lights([api.pointLight(occluders, 0.5, 0.6, (1.0, 9.0, -14.0)), api.pointLight(world, 0.3, 0.4, (-4.0, 6.0, -6.0)),
                                     api.pointLight(api.mconcat([]), 0.2, 0.2, (5.0, 3.0, -2.0)), api.pointLight(occluders, 0.2, 0.3, (-7.0, 8.0, -20.0)),
                                     api.ambient(0.1)])
        osc = ob.OracleScene(world, lights)
        ref = osc.render(w, h, depth=depth, nthreads=16)
        plain = ob.OracleScene(world, [api.PointLight(l.diffuse, l.specular, l.pos) if isinstance(l, api.PointLight) else l for l in lights]).render(w, h, depth=depth, nthreads=16)
        assert (ref["rgb"] != plain["rgb"]).any()                      # (the worlds do change the picture)
        sc = api.Scene(world, lights)
        for cull in (1, 0):
            sc.set_option(_lib.RT_OPT_CULL, cull)
            rgba, hit = sc.render(w, h, depth=depth)
            assert (hit == ref["hit_id"]).all() and (rgba[:, :, :3] == ref["rgb"]).all(), cull
        rgba2, hit2 = sc.render(w, h, depth=depth, row0=40, nrows=56)
        assert (rgba2 == rgba[40:96]).all() and (hit2 == hit[40:96]).all()
        ref4 = osc.render(w, h, depth=depth, spp=4, seed=3, row0=64, nrows=16, nthreads=16)
        rgba4, hit4 = sc.render(w, h, depth=depth, spp=4, seed=3, row0=64, nrows=16)
        assert (hit4 == ref4["hit_id"]).all() and (rgba4[:, :, :3] == ref4["rgb"]).all()
        sc.close()
    # Talk.hs's pointLight closes over a world too (Talk.hs:237-243)
    world, lights = scenes.talkhs_main2()
    lights2 = [api.PointLight(l.diffuse, l.specular, l.pos, api.mconcat([])) if isinstance(l, api.PointLight) else l for l in lights]
    ref = ob.OracleScene(world, lights2).render(160, 120)
    sc = api.Scene(world, lights2)
    rgba, hit = sc.render(160, 120)
    sc.close()
    assert (hit == ref["hit_id"]).all() and (rgba[:, :, :3] == ref["rgb"]).all()
    assert (ref["rgb"] != ob.OracleScene(world, lights).render(160, 120)["rgb"]).any()


def test_rgb24_outputs(gpu_lib):
    """RT_OPT_RGB24: the colour outputs are 3 bytes per pixel (the reference's PixelRGB8).  Same colours as the RGBA8
    frame through every way out: rt_render into pageable and page-locked memory with several bands, a band of rows,
    rt_render_device (also the no-alloc entry point and the split over two lanes), 4 samples per pixel, a reflective
    scene (bounce levels), rt_render_multi — for a width that is a multiple of 4 (four pixels = three dwords) and for
    widths that are not (byte stores)."""
    from raytrace_amd import _lib
    from hip_buffers import DeviceBuffer
    for (world, lights), depth in ((scenes.synthetic(ntri=600, ncuboid=60), 4), (scenes.mirror_room(), 4)):
        for w, h in ((1000, 203), (1001, 67), (998, 40)):
            sc = api.Scene(world, lights)
            ref, ref_hit = sc.render(w, h, depth=depth)
            ref4, _ = sc.render(w, h, depth=depth, spp=4, seed=5)
            sc.set_option(_lib.RT_OPT_RGB24, 1)
            for bands in (1, 2, 3):
                sc.set_option(_lib.RT_OPT_HOST_BANDS, bands)
                rgb, hit = sc.render(w, h, depth=depth)
                assert rgb.shape == (h, w, 3) and (rgb == ref[:, :, :3]).all() and (hit == ref_hit).all(), (w, h, bands)
                pin = api.host_buffer((h, w, 3), np.uint8)
                pin[:] = 0x5a
                sc.render(w, h, depth=depth, out_rgba=pin, want_hit=False)
                assert (pin == ref[:, :, :3]).all(), (w, h, bands)
            rgb, _ = sc.render(w, h, depth=depth, row0=16, nrows=19)
            assert (rgb == ref[16:35, :, :3]).all()
            rgb, _ = sc.render(w, h, depth=depth, spp=4, seed=5)
            assert (rgb == ref4[:, :, :3]).all()
            dev = DeviceBuffer((h + 1, w, 3), np.uint8, fill=0x5a)           # (one row of slack: nothing may be written behind the frame)
            sc.render_device(w, h, dev.ptr, None, depth=depth)
            got = dev.numpy()
            assert (got[:h] == ref[:, :, :3]).all() and (got[h] == 0x5a).all()
            odd = DeviceBuffer((3 * w * h + 8,), np.uint8, fill=0x5a)         # a buffer that does not start on a dword: byte stores
            sc.render_device(w, h, odd.ptr + 1, None, depth=depth)
            got = odd.numpy()
            assert got[0] == 0x5a and (got[1:1 + 3 * w * h].reshape(h, w, 3) == ref[:, :, :3]).all() and (got[1 + 3 * w * h:] == 0x5a).all()
            sc.prepare(w, h)
            dev2 = DeviceBuffer((h + 1, w, 3), np.uint8, fill=0x5a)
            sc.render_device_noalloc(w, h, dev2.ptr, None, depth=depth)
            got = dev2.numpy()
            assert (got[:h] == ref[:, :, :3]).all() and (got[h] == 0x5a).all()
            rgb, hit = sc.render_multi(w, h, [0, 0, 0], depth=depth)
            assert rgb.shape == (h, w, 3) and (rgb == ref[:, :, :3]).all() and (hit == ref_hit).all()
            sc.close()
    # a band big enough for rt_render_device to split it over the two lanes
    world, lights = scenes.synthetic(ntri=600, ncuboid=60)
    sc = api.Scene(world, lights)
    w, h = 4096, 3200
    ref, _ = sc.render(w, h)
    sc.set_option(_lib.RT_OPT_RGB24, 1)
    dev = DeviceBuffer((h, w, 3), np.uint8, fill=0x5a)
    sc.render_device(w, h, dev.ptr, None)
    assert (dev.numpy() == ref[:, :, :3]).all()
    sc.close()


def test_wave_history_changes_nothing_but_the_order(gpu_lib):
    """RT_OPT_HISTORY (on by default): from the second render of a band on, the waves of the two big kernels are dispatched
    longest first, by the durations they left behind.  The bytes of frames 2, 3, 4 equal those of frame 1 (no history yet) and
    of a scene that keeps no history, which in turn equal the oracle's; so do a band, a supersampled frame (the later
    samples use the orders of the sample before), a frame of more than one chunk, and the counts."""
    from raytrace_amd import _lib
    world, lights = scenes.synthetic(ntri=600, ncuboid=60)
    w, h = 1000, 531
    plain = api.Scene(world, lights).set_option(_lib.RT_OPT_HISTORY, 0)
    ref_rgba, ref_hit = plain.render(w, h)
    ref_st = plain.stats()
    osc = ob.OracleScene(world, lights)
    ref = osc.render(w, h, row0=256, nrows=8, nthreads=16)
    assert (ref_hit[256:264] == ref["hit_id"]).all() and (ref_rgba[256:264, :, :3] == ref["rgb"]).all()
    ref4 = plain.render(w, h, spp=4, seed=11)
    sc = api.Scene(world, lights)
    for frame in range(4):
        rgba, hit = sc.render(w, h)
        st = sc.stats()
        assert (rgba == ref_rgba).all() and (hit == ref_hit).all(), frame
        assert st["primary_rays"] == ref_st["primary_rays"] and st["shadow_rays"] == ref_st["shadow_rays"], frame
        rgba, hit = sc.render(w, h, row0=77, nrows=201)            # another band: a history of its own
        assert (rgba == ref_rgba[77:278]).all() and (hit == ref_hit[77:278]).all(), frame
        got4 = sc.render(w, h, spp=4, seed=11)
        assert (got4[0] == ref4[0]).all() and (got4[1] == ref4[1]).all(), frame
    sc.close(); plain.close()
    # more than one chunk (16 Mi pixels) per band
    sc = api.Scene(world, lights).set_option(_lib.RT_OPT_HOST_BANDS, 1)
    w, h = 4096, 4200
    first = sc.render(w, h)
    for frame in range(2):
        again = sc.render(w, h)
        assert (again[0] == first[0]).all() and (again[1] == first[1]).all(), frame
    ref = osc.render(w, h, row0=4184, nrows=8, nthreads=16)
    assert (first[1][4184:4192] == ref["hit_id"]).all() and (first[0][4184:4192, :, :3] == ref["rgb"]).all()
    sc.close()


@pytest.mark.parametrize("name", sorted(__import__("known_answers").CASES))
def test_hand_derived_pixel_on_the_gpu(gpu_lib, name):
    """The closed-form pixels of tests/known_answers.py (derived from Main.hs by hand, no oracle involved) through the
    C ABI: the one pin of the GPU path that does not go through the CPU restatement."""
    import known_answers as ka
    world, lights, w, h, (row0, nrows), _ = ka.CASES[name]()
    sc = api.Scene(world, lights)
    rgba, hit = sc.render(w, h, depth=4, row0=row0, nrows=nrows)
    sc.close()
    ka.check(name, hit, rgba, row0)


def test_stack_of_walls(gpu_lib):
    """24 big rectangles one behind the other, all facing the camera: every pixel has 24 true intersections, so a 32 x 8
    tile pushes 6144 pairs through its candidate queues and its exact stage (the queues' wrap-around paths)."""
    m = api.Material((180, 90, 40), (255, 255, 255), 20.0, 0.0)
    walls = [api.rectangle(m.replace(diffuse=(10 * k % 256, 255 - 9 * k % 256, 40 + 5 * k)), (0.0, 0.0, -4.0 - 0.25 * k), (30.0, 0.0, 0.0), (0.0, 30.0, 0.0))
             for k in range(24)]
    lights = api.mconcat_lights([api.pointLight(0.5, 0.4, (3.0, 4.0, 2.0)), api.ambient(0.2)])
    compare(api.mconcat(walls), lights, 160, 100, label="stack-of-walls")


def test_an_fnorm_that_raises_is_reported(gpu_lib):
    """A Python fnorm that raises must not leave a 'successful' render with (0, 0, 0) normals behind (ctypes would only
    print the traceback): the exception comes back from render()."""
    def bad(x, y):
        raise ZeroDivisionError("fnorm failed")
    world, lights = api.linearInterpolation(api.gaussian, bad, (-1.0, -1.0), (1.0, 1.0), 0.5, (0.01, -7.0, -25.0), 10.0)
    sc = api.Scene(world, lights)
    with pytest.raises(RuntimeError, match="fnorm callback raised"):
        sc.render(96, 96)
    sc.close()
