"""rt_accumulate_static: the entry point that allocates nothing, against rt_accumulate.

Its work buffers live in a 64 MiB device array of the library instead of hipMalloc'ed memory, so a
frame is rendered in more and smaller batches; the sums must be the same, bit for bit, also on a
second call with a smaller input that finds the previous call's contents in the array.
"""
import ctypes as C

import numpy as np
import pytest

import raytracer_amd as rt
from helpers import load, variant
from hip_mem import CapturedGraph, DeviceBuffer, free_bytes

pytestmark = pytest.mark.gpu


def accumulate_static(dev, cam, opt, begin, end, buf, stream=None):
    return rt.lib.rt_accumulate_static(dev._h, C.byref(cam), C.byref(opt), begin, end, C.c_void_p(buf.address), stream)


def both_ways(scene, opt, begin, end):
    """rt_accumulate_static on a FRESH scene handle -- no rt_accumulate has given it work buffers -- with the free
    device memory read before and after the call, then rt_accumulate on a second handle of the same scene."""
    desc, cam = scene.describe()
    dev = scene.upload()
    n = cam.width * cam.height
    want, got = DeviceBuffer(4 * n), DeviceBuffer(4 * n)
    # one warm-up call on another fresh handle: the runtime loads a kernel's code when it is first launched
    assert accumulate_static(scene.upload(), cam, opt, begin, end, got) == rt.RT_OK, rt.lib.rt_last_error()
    got.to_host()  # (synchronises: nothing of the set-up is still in flight)
    got.zero()
    before = free_bytes()
    rc = accumulate_static(dev, cam, opt, begin, end, got)
    assert rc == rt.RT_OK, rt.lib.rt_last_error()
    b = got.to_host()
    assert free_bytes() == before, f"rt_accumulate_static changed the free device memory by {before - free_bytes()} bytes"
    scene.upload().accumulate(cam, opt, begin, end, want.address)
    a = want.to_host()
    assert a.reshape(-1, 4)[:, 3].max() == end - begin
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"{(a != b).reshape(-1, 4).any(axis=1).sum()} of {n} pixels differ"
    return dev, cam, got


def test_static_work_area_gives_the_sums_of_rt_accumulate():
    rt.init(0)
    # first call: two lights' worth of state per path, several batches (240 x 180 x 6 = 259 200 paths)
    s = variant("mesh_scene", "glass", 240, 180, 6)
    s.add_light([10, 20, 40], 4e9)
    dev, cam, buf = both_ways(s, rt.make_options(seed=61), 0, 6)
    # second call, smaller input, right after: it finds the first call's state in the array
    both_ways(load("mesh_scene", 72, 40, 3), rt.make_options(seed=62), 2, 5)
    # a scene without a mesh, direct light only, then the exhaustive scan
    both_ways(load("default", 96, 64, 4), rt.make_options(indirect=False, seed=63), 0, 4)
    both_ways(load("mesh_scene", 64, 48, 2), rt.make_options(seed=64, cluster_culling=False), 0, 2)
    # the fast arithmetic (its first bounce regenerates the camera rays from one more work buffer), with and without a mesh
    both_ways(load("mesh_scene", 120, 90, 5), rt.make_options(seed=68, arithmetic=1), 1, 5)
    both_ways(load("default", 96, 64, 4), rt.make_options(seed=69, arithmetic=1), 0, 4)
    # continuing a sum: ranges in sample order add up to the whole
    n = cam.width * cam.height
    whole = DeviceBuffer(4 * n)
    parts = DeviceBuffer(4 * n)
    opt = rt.make_options(seed=65)
    assert accumulate_static(dev, cam, opt, 0, 5, whole) == rt.RT_OK
    assert accumulate_static(dev, cam, opt, 0, 2, parts) == rt.RT_OK
    assert accumulate_static(dev, cam, opt, 2, 5, parts) == rt.RT_OK
    assert np.array_equal(whole.to_host().view(np.uint32), parts.to_host().view(np.uint32))


def test_static_work_area_with_three_culling_levels(monkeypatch):
    """The third culling level takes its lists, counters and run table from the same work area."""
    rt.init(0)
    monkeypatch.setenv("RT_AMD_LEVELS", "3")
    monkeypatch.setenv("RT_AMD_CLUSTER_PAIRS", "1")
    both_ways(load("mesh_scene", 96, 64, 3), rt.make_options(seed=67), 0, 3)


def test_static_work_area_ignores_the_profile_option():
    """Timing marks would create events (an allocation): the call takes none, whatever rt_options.profile says."""
    rt.init(0)
    s = load("mesh_scene", 64, 48, 2)
    desc, cam = s.describe()
    dev = s.upload()
    n = cam.width * cam.height
    plain, marked = DeviceBuffer(4 * n), DeviceBuffer(4 * n)
    assert accumulate_static(dev, cam, rt.make_options(seed=66), 0, 2, plain) == rt.RT_OK
    plain.to_host()
    before = free_bytes()
    assert accumulate_static(dev, cam, rt.make_options(seed=66, profile=2), 0, 2, marked) == rt.RT_OK
    assert np.array_equal(plain.to_host().view(np.uint32), marked.to_host().view(np.uint32))
    assert free_bytes() == before
    with pytest.raises(rt.RtError):  # no marks were left for rt_profile_collect
        dev.profile_collect()


def test_static_work_area_under_stream_capture():
    """The callers the entry point is for: the call is captured into a HIP graph (nothing runs, nothing may be
    allocated while a stream captures) and the graph replayed twice; the sums are rt_accumulate's."""
    rt.init(0)
    s = load("mesh_scene", 96, 64, 3)
    desc, cam = s.describe()
    dev = s.upload()
    n = cam.width * cam.height
    opt = rt.make_options(seed=67)
    want, got = DeviceBuffer(4 * n), DeviceBuffer(4 * n)
    scene2 = s.upload()
    scene2.accumulate(cam, opt, 0, 3, want.address)
    a = want.to_host()
    g = CapturedGraph()
    try:
        g.begin()
        rc = accumulate_static(dev, cam, opt, 0, 3, got, g.stream)
        g.end()
        assert rc == rt.RT_OK, rt.lib.rt_last_error()
        assert not got.to_host().any()  # captured, not run
        for _ in range(2):
            got.zero()
            g.launch()
            assert np.array_equal(got.to_host().view(np.uint32), a.view(np.uint32))
    finally:
        g.free()


def test_static_work_area_rejects_what_does_not_fit():
    rt.init(0)
    s = load("mesh_scene", 1 << 20, 2, 1)  # one image row alone needs hundreds of megabytes of path state
    desc, cam = s.describe()
    dev = s.upload()
    buf = DeviceBuffer(16)
    rc = rt.lib.rt_accumulate_static(dev._h, C.byref(cam), C.byref(rt.make_options()), 0, 1, C.c_void_p(buf.address), None)
    assert rc == rt.RT_ERR_INVALID and b"static work area" in rt.lib.rt_last_error()
    assert rt.lib.rt_accumulate_static(None, C.byref(cam), C.byref(rt.make_options()), 0, 1, C.c_void_p(buf.address), None) == rt.RT_ERR_INVALID
    assert rt.lib.rt_accumulate_static(dev._h, C.byref(cam), C.byref(rt.make_options()), 0, 1, None, None) == rt.RT_ERR_INVALID
    assert rt.lib.rt_accumulate_static(dev._h, C.byref(cam), C.byref(rt.make_options()), 3, 3, C.c_void_p(buf.address), None) == rt.RT_OK
