"""Meshes larger than the shipped cat model: more than 64 culling clusters (two culling levels), more than
1 024 (from 16 385 triangles on: three levels, or -- RT_AMD_LEVELS=2 -- two with larger clusters)."""
import os
import sys

import numpy as np
import pytest

import oracle_binding as ob
import raytracer_amd as rt
from helpers import SCENES, same_bits
from hip_mem import DeviceBuffer

sys.path.insert(0, os.path.join(rt.REPO_ROOT, "tools"))
import make_mesh  # noqa: E402

pytestmark = pytest.mark.gpu


def scene_with_mesh(tmp_path, n_lat, n_lon, material, width, height, spp):
    v, n, t = make_mesh.bumpy_sphere(n_lat, n_lon)
    path = tmp_path / f"bumpy_{n_lat}x{n_lon}.obj"
    make_mesh.write_obj(str(path), v, n, t)
    s = rt.Scene()
    s.load_scene(os.path.join(SCENES, "simple.json"))
    s.set_camera_intrinsics(60.0, width, height)
    s.set_num_rays(spp)
    s.add_object(str(path), [0, 2, 8], [1, 1, 1], [10, 25, 0], [0.9, 0.6, 0.3], material, 1.0, 1.5)
    return s, len(t)


@pytest.mark.parametrize("n_lat,n_lon,material,levels", [(32, 62, rt.DIFFUSE, None), (100, 200, rt.DIFFUSE, None), (100, 200, rt.DIFFUSE, "2"),
                                                         (60, 90, rt.GLASS, None), (200, 400, rt.MIRROR, None), (200, 400, rt.MIRROR, "2")])
def test_large_mesh_culled_equals_exhaustive_equals_oracle(tmp_path, monkeypatch, n_lat, n_lon, material, levels):
    if levels:
        monkeypatch.setenv("RT_AMD_LEVELS", levels)
    s, n_tri = scene_with_mesh(tmp_path, n_lat, n_lon, material, 96, 80, 2)
    desc, cam = s.describe()
    assert desc.n_triangles == n_tri == 2 * n_lat * n_lon - 2 * n_lon
    dev = s.upload()
    culled, st_c = dev.render(cam, rt.make_options(seed=3))
    full, st_f = dev.render(cam, rt.make_options(seed=3, cluster_culling=False))
    assert same_bits(culled, full), f"{(culled != full).any(axis=2).sum()} pixels differ"
    assert st_c.triangle_tests < st_f.triangle_tests / 8
    rng = np.random.default_rng(1)
    pixels = rng.choice(cam.width * cam.height, 300, replace=False).astype(np.uint32)
    ref, _, _, _ = ob.render_pixels(desc, cam, ob.options(seed=3), pixels)
    assert same_bits(culled.reshape(-1, 3)[pixels], ref)


@pytest.mark.parametrize("levels", ["2", "3"])
def test_large_mesh_probes(tmp_path, monkeypatch, levels):
    monkeypatch.setenv("RT_AMD_LEVELS", levels)
    s, _ = scene_with_mesh(tmp_path, 100, 200, rt.DIFFUSE, 32, 32, 1)
    desc, _ = s.describe()
    dev = s.upload()
    rng = np.random.default_rng(5)
    o = rng.uniform(-35, 35, (600, 3)).astype(np.float32)
    d = (rng.uniform(-10, 10, (600, 3)) + np.array([0, 2, 8]) - o)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    for culling in (True, False):
        opt = rt.make_options(cluster_culling=culling)
        t, sid, tid, abg = dev.cast(opt, o, d)
        rt_, rsid, rtid, rabg = ob.cast(desc, ob.from_rt_options(opt), o, d)
        assert (rtid >= 0).sum() > 200
        assert np.array_equal(tid, rtid) and np.array_equal(sid, rsid) and same_bits(t, rt_)
        assert same_bits(abg[rtid >= 0], rabg[rtid >= 0])


def test_large_mesh_sums_with_and_without_statistics(tmp_path):
    """A call without statistics leaves the per-cluster counters of a mesh of thousands of clusters alone (they feed the
    statistics only) and clears the counters around them: the sums are those of a call with statistics."""
    s, _ = scene_with_mesh(tmp_path, 200, 400, rt.DIFFUSE, 64, 48, 3)
    _, cam = s.describe()
    dev = s.upload()
    n = cam.width * cam.height
    opt = rt.make_options(seed=9)
    sums = []
    for want_stats in (True, False, False):
        buf = DeviceBuffer(4 * n)
        st = dev.accumulate(cam, opt, 0, 3, buf.address, want_stats=want_stats)
        sums.append(buf.to_host().view(np.uint32).copy())
        assert (st is not None) == want_stats and (st is None or st.triangle_tests > 0)
    assert np.array_equal(sums[0], sums[1]) and np.array_equal(sums[0], sums[2])


def test_three_levels_on_a_work_set_laid_out_for_more_stages(tmp_path):
    """A handle whose work buffers were laid out for a frame of five bounces then renders a frame of two and casts
    probe rays: the counters of all three culling levels of the stages in use are cleared wherever they lie."""
    s, _ = scene_with_mesh(tmp_path, 100, 200, rt.DIFFUSE, 64, 48, 2)
    desc, cam = s.describe()
    used = s.upload()
    used.render(cam, rt.make_options(seed=4))
    shallow = rt.make_options(seed=5, max_depth=2)
    a, _ = used.render(cam, shallow)
    b, _ = s.upload().render(cam, shallow)
    assert same_bits(a, b)
    rng = np.random.default_rng(7)
    o = rng.uniform(-35, 35, (400, 3)).astype(np.float32)
    d = (rng.uniform(-10, 10, (400, 3)) + np.array([0, 2, 8]) - o)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    opt = rt.make_options()
    t, sid, tid, abg = used.cast(opt, o, d)
    rt_, rsid, rtid, rabg = ob.cast(desc, ob.from_rt_options(opt), o, d)
    assert (rtid >= 0).sum() > 100
    assert np.array_equal(tid, rtid) and np.array_equal(sid, rsid) and same_bits(t, rt_)


def test_fast_arithmetic_on_three_culling_levels(tmp_path):
    """The fast arithmetic's triangle stage (fused filter, bounds with the small padding) through all three culling
    levels: direct light without anti-aliasing, so no random number reaches a pixel -- at least 99.9 % of the pixels
    within 1e-4 relative of the exact arithmetic, the rest being hit-or-miss decisions at an edge; the culled frame
    against the exhaustive one in the same arithmetic likewise."""
    from test_gpu_fast_arithmetic import pixels_outside
    s, _ = scene_with_mesh(tmp_path, 100, 200, rt.DIFFUSE, 256, 192, 1)
    _, cam = s.describe()
    dev = s.upload()
    opts = dict(indirect=False, antialiasing=False, seed=2)
    exact, _ = dev.render(cam, rt.make_options(arithmetic=0, **opts))
    fast, st = dev.render(cam, rt.make_options(arithmetic=1, **opts))
    full, st_full = dev.render(cam, rt.make_options(arithmetic=1, cluster_culling=False, **opts))
    assert 0 < st.triangle_tests < st_full.triangle_tests / 8
    out, err = pixels_outside(fast, exact)
    print(f"fast against exact: {out.sum()} of {out.size} pixels outside 1e-4")
    assert out.mean() <= 1e-3 and np.median(err[~out]) < 1e-5
    out, _ = pixels_outside(fast, full)
    print(f"culled against exhaustive, both fast: {out.sum()} of {out.size} pixels outside 1e-4")
    assert out.mean() <= 1e-3
