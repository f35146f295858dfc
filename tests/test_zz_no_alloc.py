"""rt_render_device_noalloc (the entry point that allocates nothing: static device scratch) against rt_render_device,
through the C ABI, bit for bit: two calls in a row, the second with a smaller input, then the cases that stress the
shared scratch (a band that needs several sub-bands, supersampling, a mirror scene, a band without a colour buffer)."""
import numpy as np
import pytest

import ctypes as C

from hip_buffers import DeviceBuffer, hip
from raytrace_amd import api, scenes

pytestmark = pytest.mark.gpu


def _pair(sc, w, h, **kw):
    nrows = kw.get("nrows", h - kw.get("row0", 0))
    out = []
    for fn in (sc.render_device, sc.render_device_noalloc):
        rgba = DeviceBuffer((nrows, w, 4), np.uint8, fill=0x5a)
        hit = DeviceBuffer((nrows, w), np.int32, fill=0x5a)
        fn(w, h, rgba.ptr, hit.ptr, **kw)
        out.append((rgba.numpy(), hit.numpy()))
        rgba.free(); hit.free()
    return out


def test_noalloc_entry_point_matches_render_device(gpu_lib):
    world, lights = scenes.synthetic(ntri=600, ncuboid=60)
    sc = api.Scene(world, lights)
    w, h = 1000, 531
    rgba = DeviceBuffer((h, w, 4), np.uint8)
    with pytest.raises(RuntimeError, match="not resident"):                     # nothing was uploaded yet: refused, not allocated
        sc.render_device_noalloc(w, h, rgba.ptr, None)
    sc.prepare(w, h)
    # two calls in a row, the second with a smaller input: the scratch keeps the first call's contents
    (a, ahit), (b, bhit) = _pair(sc, w, h, depth=4)
    assert (a == b).all() and (ahit == bhit).all()
    st = sc.stats()
    assert st["primary_rays"] == w * h and st["ms_total"] == 0.0
    (a, ahit), (b, bhit) = _pair(sc, 333, 117, depth=4)
    assert (a == b).all() and (ahit == bhit).all()
    (a, ahit), (b, bhit) = _pair(sc, w, h, depth=4, row0=77, nrows=201)  # a band, as a rank renders it
    assert (a == b).all() and (ahit == bhit).all()
    (a, ahit), (b, bhit) = _pair(sc, 400, 300, depth=4, spp=4, seed=9)    # supersampled: the accumulator lives in the scratch too
    assert (a == b).all() and (ahit == bhit).all()
    hit = DeviceBuffer((h, w), np.int32)                                        # hit IDs only
    sc.render_device_noalloc(w, h, None, hit.ptr)
    ref = sc.render(w, h)
    assert (hit.numpy() == ref[1]).all()
    sc.close()
    # a frame whose buffers exceed the 64 MiB: rendered as sub-bands of as many 16-row tiles as fit
    world, lights = scenes.synthetic()
    sc = api.Scene(world, lights)
    sc.prepare(3840, 2160)
    (a, ahit), (b, bhit) = _pair(sc, 3840, 2160, depth=4)
    assert (a == b).all() and (ahit == bhit).all()
    assert sc.stats()["primary_rays"] == 3840 * 2160
    sc.close()
    # reflective scene (bounce levels in the scratch), then too wide an image
    world, lights = scenes.mirror_room()
    sc = api.Scene(world, lights)
    sc.prepare(640, 480)
    (a, ahit), (b, bhit) = _pair(sc, 640, 480, depth=4)
    assert (a == b).all() and (ahit == bhit).all()
    sc.prepare(200000, 16)
    wide = DeviceBuffer((16, 200000, 4), np.uint8)
    with pytest.raises(RuntimeError, match="too wide"):
        sc.render_device_noalloc(200000, 16, wide.ptr, None)
    sc.close()
    sc = api.Scene(*scenes.sine_cosine(True))                                   # fnorm callbacks need host staging
    sc.prepare(64, 64)
    with pytest.raises(RuntimeError, match="callbacks"):
        sc.render_device_noalloc(64, 64, rgba.ptr, None)
    sc.close()


def test_noalloc_call_is_capturable_into_a_hip_graph(gpu_lib):
    """Under HIP stream capture (global mode: any allocating or synchronising call fails) the call is recorded into a graph;
    the graph is launched twice and the frame equals rt_render_device's."""
    H = hip()
    world, lights = scenes.synthetic(ntri=600, ncuboid=60)
    sc = api.Scene(world, lights)
    w, h = 1000, 531
    sc.prepare(w, h)
    ref_rgba, ref_hit = DeviceBuffer((h, w, 4), np.uint8), DeviceBuffer((h, w), np.int32)
    sc.render_device(w, h, ref_rgba.ptr, ref_hit.ptr)
    ref = (ref_rgba.numpy(), ref_hit.numpy())
    rgba, hit = DeviceBuffer((h, w, 4), np.uint8, fill=0x5a), DeviceBuffer((h, w), np.int32, fill=0x5a)
    stream, graph, gexec = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert H.hipStreamCreateWithFlags(C.byref(stream), 1) == 0                    # hipStreamNonBlocking
    assert H.hipStreamBeginCapture(stream, 0) == 0                                # hipStreamCaptureModeGlobal: any forbidden call fails
    try:
        sc.render_device_noalloc(w, h, rgba.ptr, hit.ptr, stream=stream.value)
    finally:
        rc = H.hipStreamEndCapture(stream, C.byref(graph))
    assert rc == 0 and graph.value
    # nothing has run yet: the buffers still hold their fill
    assert H.hipDeviceSynchronize() == 0
    assert (hit.numpy() == 0x5a5a5a5a).all()
    assert H.hipGraphInstantiate(C.byref(gexec), graph, None, None, 0) == 0
    for _ in range(2):
        assert H.hipGraphLaunch(gexec, stream) == 0
        assert H.hipStreamSynchronize(stream) == 0
        assert (rgba.numpy() == ref[0]).all() and (hit.numpy() == ref[1]).all()
        assert H.hipMemset(C.c_void_p(hit.ptr), 0x5a, hit.nbytes) == 0 and H.hipDeviceSynchronize() == 0
    assert H.hipGraphExecDestroy(gexec) == 0 and H.hipGraphDestroy(graph) == 0 and H.hipStreamDestroy(stream) == 0
    sc.close()
