"""The benchmark configurations at their full sizes, edge cases, and end-to-end
entry points on the MI355X.

At full size the oracle is too slow for whole frames, so each configuration is
checked (1) bit-for-bit against the oracle on a random subset of pixels, (2)
through size-independent properties (sample-range composition, determinism)
and (3) statistically against the reference's own shipped render of the same
configuration where one exists.
"""
import json
import os
import subprocess

import numpy as np
import pytest
from PIL import Image

import oracle_binding as ob
import raytracer_amd as rt
from helpers import SCENES, load, same_bits, variant

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def check_subset(scene, opt, n_check, seed=0):
    """Full-frame HIP render vs the oracle on n_check random pixels, bit for bit."""
    desc, cam = scene.describe()
    dev = scene.upload()
    img, st = dev.render(cam, opt)
    rng = np.random.default_rng(seed)
    pixels = rng.choice(cam.width * cam.height, n_check, replace=False).astype(np.uint32)
    ref, _, _, _ = ob.render_pixels(desc, cam, ob.from_rt_options(opt), pixels)
    got = img.reshape(-1, 3)[pixels]
    assert same_bits(got, ref), f"{(got != ref).any(axis=1).sum()} of {n_check} pixels differ"
    return img, st


def two_light_scene(base, width, height, spp):
    """`base` scene with simple.json's two lights (the reference's two-lights render)."""
    import tempfile

    doc = json.load(open(os.path.join(SCENES, base + ".json")))
    doc["lights"] = json.load(open(os.path.join(SCENES, "simple.json")))["lights"]
    for o in doc.get("objects", []):
        o["filepath"] = os.path.join(rt.ASSETS, "objects", "cat", "cat.obj")
    doc["camera"].update(width=width, height=height, num_rays=spp)
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
        json.dump(doc, f)
    s = rt.Scene()
    s.load_scene(f.name)
    os.unlink(f.name)
    return s


# ---- the five benchmark configurations ------------------------------------

def test_c1_default_direct_aa_full_size():
    check_subset(load("default"), rt.make_options(indirect=False, antialiasing=True, bounding_box=True, seed=1), 6000)


def test_c2_default_indirect_aa_full_size():
    check_subset(load("default"), rt.make_options(indirect=True, antialiasing=True, bounding_box=False, seed=1), 6000)


def test_c3_cat_indirect_aa_bbox_full_size():
    check_subset(load("mesh_scene"), rt.make_options(seed=1), 1500)
    img, st = check_subset(load("mesh_scene"), rt.make_options(seed=1, cluster_culling=False), 1500)
    paths = 512 * 512 * 32
    assert st.paths == paths
    # the recorded algorithmic work of this configuration: 7 174 triangle tests per path
    assert st.triangle_tests / paths == pytest.approx(7174, rel=0.02)
    assert st.sphere_casts / paths == pytest.approx(10.0, rel=0.01)  # recorded: 10 casts per path


@pytest.mark.parametrize("scene,mode", [("mesh_scene", (True, True, True)), ("default", (True, True, False)),
                                        ("simple", (False, True, True))])
def test_work_counters_equal_the_oracles(scene, mode):
    """Casts and mesh walks are counted on the device; the oracle counts the same events.  (Not in the
    mode without indirect light and anti-aliasing: there the reference stops a pixel's sample loop
    after one sample, src/raytrace.cu:496-498, while the device traces every sample and drops the rest.)"""
    s = load(scene, 72, 56, 3)
    desc, cam = s.describe()
    opt = rt.make_options(*mode, seed=17, cluster_culling=False)
    img, st = s.upload().render(cam, opt)
    ref, counters = ob.render_image(desc, cam, ob.from_rt_options(opt))
    assert same_bits(img, ref)
    assert st.sphere_casts == counters["casts"]
    # a handful of mesh walks differ: the oracle also walks for a shadow ray that a sphere already blocks at the same distance
    assert st.triangle_tests == pytest.approx(counters["triangle_tests"], rel=1e-3)


@pytest.mark.parametrize("material", ["glass", "mirror"])
def test_c4_cat_glass_mirror_720p_60spp(material):
    s = variant("mesh_scene", material)
    s.set_camera_intrinsics(60.0, 1280, 720)
    s.set_num_rays(60)
    img, st = check_subset(s, rt.make_options(seed=2), 400)
    assert st.batches >= 2  # 55 M paths do not fit one batch: the batching path ran
    assert st.paths == 1280 * 720 * 60


def test_c5_default_two_lights_3000x2000_1000spp():
    """The configuration at its stated size: 6e9 paths.  A scene without triangles is traced by one
    launch with a lane per pixel (no path state in memory); 1 500 random pixels are compared with the
    oracle bit for bit, and a band of rows rendered through the other arrangement (four samples per
    lane, per-sample colours added by k_accumulate, in two sample ranges) must give the same bits."""
    from hip_mem import DeviceBuffer

    s = two_light_scene("default", 3000, 2000, 1000)
    opt = rt.make_options(bounding_box=False, seed=3)
    img, st = check_subset(s, opt, 1500)
    assert st.paths == 3000 * 2000 * 1000 and st.batches == 1
    assert 5 < st.sphere_casts / st.paths < 16  # 1 camera ray + per diffuse bounce 2 shadow rays + 1 bounce ray, until the weight is zero
    desc, cam = s.describe()
    dev = s.upload()
    n = cam.width * cam.height
    band, out = DeviceBuffer(4 * n), DeviceBuffer(3 * n)
    rows = (1000, 1064)
    dev.accumulate(cam, opt, 0, 391, band.address, rows=rows)
    dev.accumulate(cam, opt, 391, 1000, band.address, rows=rows)
    rt.resolve(band.address, n, out.address)
    got = out.to_host().reshape(cam.height, cam.width, 3)[rows[0]:rows[1]]
    assert np.array_equal(got.view(np.uint32), img[rows[0]:rows[1]].view(np.uint32))


def test_image_of_more_pixels_than_a_batch_holds():
    """6000 x 3000 = 18 M pixels > 2^24 paths per batch: the frame is rendered in chunks of rows."""
    s = load("mesh_scene", 6000, 3000, 2)
    img, st = check_subset(s, rt.make_options(seed=4), 1200)
    assert st.paths == 6000 * 3000 * 2 and st.batches == 4  # two chunks of rows x two samples
    assert img.shape == (3000, 6000, 3)


def test_two_lights_with_mesh_keeps_light_order():
    """Shadow rays of two lights are answered later by the triangle kernel; the
    contributions must still be added in light order."""
    s = two_light_scene("mesh_scene", 96, 96, 4)
    desc, cam = s.describe()
    dev = s.upload()
    for mode in [(True, True, True), (False, True, True), (True, True, False)]:
        opt = rt.make_options(*mode, seed=4)
        img, _ = dev.render(cam, opt)
        ref, _ = ob.render_image(desc, cam, ob.from_rt_options(opt))
        assert same_bits(img, ref)


# ---- size-independent properties at full size -----------------------------

def test_sample_ranges_compose_bitwise_at_full_size():
    from hip_mem import DeviceBuffer

    s = load("mesh_scene")
    desc, cam = s.describe()
    dev = s.upload()
    opt = rt.make_options(seed=6)
    n = cam.width * cam.height
    whole, parts, out = DeviceBuffer(4 * n), DeviceBuffer(4 * n), DeviceBuffer(3 * n)
    dev.accumulate(cam, opt, 0, 32, whole.address)
    for b, e in [(0, 5), (5, 6), (6, 20), (20, 32)]:  # ranges in sample order continue the same sum
        dev.accumulate(cam, opt, b, e, parts.address)
    a, p = whole.to_host(), parts.to_host()
    assert np.array_equal(a.view(np.uint32), p.view(np.uint32))
    assert np.all(a.reshape(-1, 4)[:, 3] == 32)
    # two devices' worth of partial sums, added: the same 32 terms in another association
    lo, hi = DeviceBuffer(4 * n), DeviceBuffer(4 * n)
    dev.accumulate(cam, opt, 0, 16, lo.address)
    dev.accumulate(cam, opt, 16, 32, hi.address)
    np.testing.assert_allclose(lo.to_host() + hi.to_host(), a, rtol=4e-6, atol=0)
    # resolve = gamma(mean): matches the one-call render
    rt.resolve(whole.address, n, out.address)
    img, _ = dev.render(cam, opt)
    assert np.array_equal(out.to_host().view(np.uint32), img.ravel().view(np.uint32))


def test_same_seed_same_image_other_seed_other_image():
    s = load("mesh_scene", 256, 256, 8)
    desc, cam = s.describe()
    dev = s.upload()
    a, _ = dev.render(cam, rt.make_options(seed=10))
    b, _ = dev.render(cam, rt.make_options(seed=10))
    c, _ = dev.render(cam, rt.make_options(seed=11))
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert (a != c).mean() > 0.3  # black channels (pure-colour walls) stay equal
    assert abs(float(a.mean()) - float(c.mean())) < 0.5


# ---- the reference's own renders ------------------------------------------

def _blur(a, r=4):
    out = np.zeros_like(a)
    for dy in range(-r, r + 1):
        for dx in range(-r, r + 1):
            out += np.roll(np.roll(a, dy, 0), dx, 1)
    return out / (2 * r + 1) ** 2


# Every render the reference ships (reference examples/*.png), with the parameters that reproduce it.  The images
# were drawn at different commits of the reference: the bounce limit (MAX_RAY_DEPTH, src/raytrace.cu:12) was 7 for
# the mirror, glass and two-lights images, 3 for the 1280x720 one and 5 (the shipped value) for the others, and
# the two-lights scene had two lights of 1e10 at (+-10, 20, 40).  tools/reference_png_sweep.py sweeps these
# parameters and shows each image singles out one setting (e.g. glass cat, blurred RMSE at depth 5 / 6 / 7 / 8:
# 4.6 / 2.4 / 1.2 / 1.9; refractive index 1.4 / 1.5 / 1.6 at depth 7: 4.1 / 1.2 / 3.7).
TWO_LIGHTS = [{"pos": [-10, 20, 40], "intensity": 1e10}, {"pos": [10, 20, 40], "intensity": 1e10}]
SHIPPED = [
    # file, scene, options, mean tolerance, blurred-RMSE tolerance (8-bit units)
    ("cat_512x512_32rays_antialias-only.png", lambda: load("mesh_scene"), dict(indirect=False), 0.1, 0.3),
    ("cat_mirror_512x512_32rays.png", lambda: variant("mesh_scene", "mirror"),
     dict(indirect=False, antialiasing=False, max_depth=7), 0.02, 0.05),
    ("cat_512x512_32rays_indirect.png", lambda: load("mesh_scene"), dict(), 0.25, 1.5),
    ("cat_glass_512x512_32rays_indirect.png", lambda: variant("mesh_scene", "glass"), dict(max_depth=7), 0.25, 1.8),
    ("cat_720p_60rays_indirect.png", lambda: load("mesh_scene", 1280, 720, 60), dict(max_depth=3), 0.25, 1.2),
    ("default_512x512_32rays_indirect.png", lambda: load("default"), dict(), 0.25, 2.0),
    ("default_512x512_1000rays_indirect_two_lights.png", lambda: with_lights("default", TWO_LIGHTS, 512, 512, 1000),
     dict(max_depth=7), 0.25, 0.6),
]


# Raw (unblurred) per-pixel RMSE bound, in 8-bit levels, for the image that is converged on both sides (1 000 samples
# per pixel here and in the reference): what is left is the two renders' own sampling noise and the 8-bit rounding.
# Measured: 3.11 levels.  (The 32-sample image of the same scene differs from its reference by 13.2 levels raw; noise
# alone would scale that by sqrt(32 / 1000) to 2.4, the rest is the heavy tail of the paths through the glass spheres.)
RAW_RMSE_TOL = {"default_512x512_1000rays_indirect_two_lights.png": 3.5}


def with_lights(base, lights, width, height, spp):
    import tempfile

    doc = json.load(open(os.path.join(SCENES, base + ".json")))
    doc["lights"] = lights
    doc["camera"].update(width=width, height=height, num_rays=spp)
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
        json.dump(doc, f)
    s = rt.Scene()
    s.load_scene(f.name)
    os.unlink(f.name)
    return s


@pytest.mark.parametrize("name,make,options,mean_tol,rmse_tol", SHIPPED, ids=[e[0][:-4] for e in SHIPPED])
def test_matches_the_references_shipped_render(name, make, options, mean_tol, rmse_tol):
    """Statistical comparison (the reference seeded its generator with the wall clock): per-channel means and
    RMSE of the 9x9-blurred images, at the image's full resolution and sample count."""
    s = make()
    desc, cam = s.describe()
    dev = s.upload()
    img, _ = dev.render(cam, rt.make_options(seed=1, **options))
    mine = np.floor(np.minimum(img, 255))
    ref = np.asarray(Image.open(os.path.join(GOLDEN, "reference_renders", name)).convert("RGB"), np.float32)
    assert ref.shape == mine.shape
    mean_diff = np.abs(mine.reshape(-1, 3).astype(np.float64).mean(0) - ref.reshape(-1, 3).astype(np.float64).mean(0))
    rmse = float(np.sqrt(np.mean((_blur(mine) - _blur(ref)) ** 2)))
    raw = float(np.sqrt(np.mean((mine - ref) ** 2)))
    print(f"{name}: mean diff {mean_diff}, blurred rmse {rmse:.3f}, raw rmse {raw:.3f}")
    assert np.all(mean_diff < mean_tol), f"mean diff {mean_diff}"
    assert rmse < rmse_tol, f"blurred rmse {rmse}"
    if name in RAW_RMSE_TOL:
        assert raw < RAW_RMSE_TOL[name], f"raw rmse {raw}"


def test_mirror_cat_matches_the_references_render_pixel_by_pixel():
    """The one deterministic image the reference ships (no anti-aliasing, direct light, no glass: no random
    number reaches a pixel): compared per pixel at the PNG's 8-bit resolution, floor(min(v, 255)) as the
    reference's writer does (src/scene.cu:216-219).  Pixels off by more than one level are listed."""
    s = variant("mesh_scene", "mirror")
    desc, cam = s.describe()
    img, _ = s.upload().render(cam, rt.make_options(indirect=False, antialiasing=False, max_depth=7, seed=1))
    mine = np.floor(np.minimum(img, 255))
    ref = np.asarray(Image.open(os.path.join(GOLDEN, "reference_renders", "cat_mirror_512x512_32rays.png")).convert("RGB"), np.float32)
    level = np.abs(mine - ref).max(axis=2)
    exact, within1 = float((level == 0).mean()), float((level <= 1).mean())
    ys, xs = np.nonzero(level > 1)
    print(f"{exact:.6f} of the pixels equal, {within1:.6f} within one level; off by more:")
    for x, y in zip(xs, ys):
        print(f"  ({x}, {y}): here {mine[y, x].astype(int).tolist()}, reference {ref[y, x].astype(int).tolist()}")
    assert exact >= 0.9999 and within1 >= 0.99995 and len(xs) <= 8
    # with the shipped bounce limit of 5 the same comparison fails: the image pins the limit it was drawn with
    img5, _ = s.upload().render(cam, rt.make_options(indirect=False, antialiasing=False, max_depth=5, seed=1))
    assert (np.abs(np.floor(np.minimum(img5, 255)) - ref).max(axis=2) > 1).sum() > 100


# ---- edge cases -------------------------------------------------------------

@pytest.mark.parametrize("size", [(1, 1), (37, 23), (8, 130), (300, 7)])
def test_resolutions_that_are_not_multiples_of_the_block(size):
    s = load("mesh_scene", size[0], size[1], 2)
    desc, cam = s.describe()
    opt = rt.make_options(seed=8)
    img, _ = s.upload().render(cam, opt)
    ref, _ = ob.render_image(desc, cam, ob.from_rt_options(opt))
    assert img.shape == (size[1], size[0], 3)
    assert same_bits(img, ref)


@pytest.mark.parametrize("depth", [0, 1, 2, 7])
def test_bounce_limits(depth):
    s = load("default", 64, 48, 3)
    desc, cam = s.describe()
    opt = rt.make_options(max_depth=depth, seed=12)
    img, _ = s.upload().render(cam, opt)
    ref, _ = ob.render_image(desc, cam, ob.from_rt_options(opt))
    assert same_bits(img, ref)
    if depth == 0:
        assert not img.any()


def test_shallower_frame_on_a_handle_that_rendered_a_deeper_one():
    """The work buffers keep the layout of the deepest frame rendered so far; a frame with fewer bounces
    then clears and uses only the stages it runs."""
    s = load("mesh_scene", 64, 48, 3)
    desc, cam = s.describe()
    dev = s.upload()
    for depth in (7, 2, 5, 1):
        opt = rt.make_options(max_depth=depth, seed=12)
        img, _ = dev.render(cam, opt)
        ref, _ = ob.render_image(desc, cam, ob.from_rt_options(opt))
        assert same_bits(img, ref), f"depth {depth}"


def test_empty_scene_and_scene_without_lights():
    s = rt.Scene()
    s.set_camera_intrinsics(60.0, 32, 16)
    s.set_num_rays(2)
    desc, cam = s.describe()
    img, st = s.upload().render(cam, rt.make_options())
    assert img.shape == (16, 32, 3) and not img.any() and st.triangle_tests == 0
    s.add_sphere([0, 0, -30], 10, [1, 1, 1], rt.DIFFUSE)
    desc, cam = s.describe()
    img, _ = s.upload().render(cam, rt.make_options())
    assert not img.any()  # hit, but nothing illuminates it
    # light but nothing to hit
    s2 = rt.Scene()
    s2.set_camera_intrinsics(60.0, 16, 16)
    s2.add_light([0, 10, 0], 1e9)
    d2, c2 = s2.describe()
    img2, _ = s2.upload().render(c2, rt.make_options())
    assert not img2.any()


def test_single_triangle_and_odd_triangle_counts(tmp_path):
    """An odd triangle count exercises the padded half of the last triangle pair."""
    p = tmp_path / "fan.obj"
    p.write_text("v -8 -8 0\nv 8 -8 0\nv 0 8 0\nv 12 10 -4\nv -12 10 -4\n"
                 "vn 0 0 1\nf 1//1 2//1 3//1\nf 3//1 2//1 4//1\nf 1//1 3//1 5//1\n")
    for n_faces in (1, 3):
        q = tmp_path / f"m{n_faces}.obj"
        q.write_text("\n".join(p.read_text().splitlines()[: 6 + n_faces]) + "\n")
        s = rt.Scene()
        s.load_scene(os.path.join(SCENES, "simple.json"))
        s.set_camera_intrinsics(60.0, 64, 64)
        s.set_num_rays(3)
        s.add_object(str(q), [0, 0, 20], [1, 1, 1], [0, 0, 0], [1, 0.5, 0.25], rt.DIFFUSE)
        desc, cam = s.describe()
        assert desc.n_triangles == n_faces
        for bbox in (True, False):
            opt = rt.make_options(bounding_box=bbox, seed=13)
            img, st = s.upload().render(cam, opt)
            ref, _ = ob.render_image(desc, cam, ob.from_rt_options(opt))
            assert same_bits(img, ref)
            # a single flat triangle has a zero-thickness box, which the reference's strict
            # min(exit) > max(entry) rule never reports as hit
            assert (st.triangle_tests > 0) == (not bbox or n_faces == 3)


def test_axis_aligned_rays_edges_and_degenerate_rays():
    """Rays along the axes (skipped slabs), rays through shared edges and vertices of the cube's
    quads (barycentrics exactly 0), zero and NaN directions: same verdicts as the oracle."""
    s = rt.Scene()
    s.add_object(os.path.join(rt.ASSETS, "objects", "cube.obj"), [0, 0, 0], [2, 2, 2], [0, 0, 0], [1, 1, 1], rt.DIFFUSE)
    s.add_sphere([0, 0, -1000], 990, [1, 1, 1], rt.DIFFUSE)
    desc, cam = s.describe()
    dev = s.upload()
    g = np.linspace(-2, 2, 9, dtype=np.float32)
    o, d = [], []
    for x in g:
        for y in g:
            o += [[x, y, 10], [10, x, y], [x, 10, y], [x, y, 10]]
            d += [[0, 0, -1], [-1, 0, 0], [0, -1, 0], [0, 0, 1]]
    o += [[0, 0, 10], [0, 0, 10], [0, 0, 0], [5, 5, 5]]
    d += [[0, 0, 0], [np.nan, 0, -1], [0, 1, 0], [-1, -1, -1]]
    o, d = np.array(o, np.float32), np.array(d, np.float32)
    for bbox in (True, False):
        opt = rt.make_options(bounding_box=bbox)
        t, sid, tid, abg = dev.cast(opt, o, d)
        rt_, rsid, rtid, rabg = ob.cast(desc, ob.from_rt_options(opt), o, d)
        assert np.array_equal(sid, rsid) and np.array_equal(tid, rtid)
        assert same_bits(t, rt_)
        assert same_bits(abg[rtid >= 0], rabg[rtid >= 0])
    # with boxes on, rays parallel to an axis never reach the mesh (the reference's slab rule)
    _, _, tid_on, _ = dev.cast(rt.make_options(bounding_box=True), o[:4 * 81], d[:4 * 81])
    _, _, tid_off, _ = dev.cast(rt.make_options(bounding_box=False), o[:4 * 81], d[:4 * 81])
    assert (tid_on >= 0).sum() == 0 and (tid_off >= 0).sum() > 50


def test_three_lights_with_glass_mesh():
    s = variant("mesh_scene", "glass", 80, 60, 3)
    s.add_light([10, 20, 40], 4e9)
    s.add_light([0, 40, 0], 2e9)
    desc, cam = s.describe()
    opt = rt.make_options(seed=14)
    img, _ = s.upload().render(cam, opt)
    ref, _ = ob.render_image(desc, cam, ob.from_rt_options(opt))
    assert desc.n_lights == 3
    assert same_bits(img, ref)


@pytest.mark.parametrize("scene", ["mesh_scene", "default"])
def test_early_termination_holds_across_calls(scene):
    """Direct light without anti-aliasing: a pixel whose first sample hits a diffuse surface ends its sample
    loop there (reference src/raytrace.cu:340-346,496-498).  Sample ranges added one call after the other must
    keep it ended -- the count stays at one sample, carried as -1 -- and give the oracle's image."""
    from hip_mem import DeviceBuffer

    s = load(scene, 80, 60, 5)
    desc, cam = s.describe()
    dev = s.upload()
    opt = rt.make_options(indirect=False, antialiasing=False, seed=19)
    n = cam.width * cam.height
    parts, whole, out = DeviceBuffer(4 * n), DeviceBuffer(4 * n), DeviceBuffer(3 * n)
    for b, e in [(0, 1), (1, 3), (3, 5)]:
        dev.accumulate(cam, opt, b, e, parts.address)
    dev.accumulate(cam, opt, 0, 5, whole.address)
    p, w = parts.to_host().reshape(-1, 4), whole.to_host().reshape(-1, 4)
    assert np.array_equal(p.view(np.uint32), w.view(np.uint32))
    assert set(np.unique(p[:, 3])) <= {-1.0, 5.0} and (p[:, 3] == -1.0).mean() > 0.5
    rt.resolve(parts.address, n, out.address)
    ref, _ = ob.render_image(desc, cam, ob.from_rt_options(opt))
    assert same_bits(out.to_host().reshape(ref.shape), ref)


def test_zero_length_sample_range_adds_nothing():
    from hip_mem import DeviceBuffer

    s = load("default", 32, 32, 4)
    desc, cam = s.describe()
    dev = s.upload()
    buf = DeviceBuffer(4 * 32 * 32)
    dev.accumulate(cam, rt.make_options(), 3, 3, buf.address)
    assert not buf.to_host().any()
    with pytest.raises(rt.RtError):
        dev.accumulate(cam, rt.make_options(), 4, 3, buf.address)


# ---- end-to-end entry points ----------------------------------------------

def test_scene_render_writes_the_clamped_png(tmp_path):
    s = load("mesh_scene", 128, 96, 4)
    s.options = rt.make_options(seed=5)
    out = tmp_path / "image.png"
    img = s.render(str(out), 5)
    desc, cam = s.describe()
    ref, _ = ob.render_image(desc, cam, ob.options(seed=5))
    assert same_bits(img, ref)
    png = np.asarray(Image.open(out).convert("RGB"))
    assert np.array_equal(png, np.minimum(img, 255).astype(np.uint8))  # fminf(v, 255) then truncation
    assert img.max() > 255  # the float image itself is not clamped


def test_render_command_line_tool(tmp_path):
    exe = os.path.join(rt.REPO_ROOT, "raytracer_amd", "render")
    if not os.path.exists(exe):
        pytest.skip("render binary not built")
    out = tmp_path / "cli.png"
    scene = os.path.join(SCENES, "simple.json")
    r = subprocess.run([exe, scene, "-o", str(out), "--seed", "3"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=120, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout
    assert "Finished raytracing in" in r.stdout
    png = np.asarray(Image.open(out).convert("RGB"))
    s = load("simple")
    s.options = rt.make_options(seed=3)
    img = s.render(None, 3)
    assert np.array_equal(png, np.minimum(img, 255).astype(np.uint8))
    bad = subprocess.run([exe, "/no/such/scene.json"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                         timeout=60, cwd=str(tmp_path))
    assert bad.returncode == 1 and "File not found" in bad.stdout


def test_frame_renderer_single_rank_matches_render():
    import torch

    from raytracer_amd.distributed import FrameRenderer

    s = load("mesh_scene", 128, 128, 6)
    desc, cam = s.describe()
    dev = s.upload()
    opt = rt.make_options(seed=15)
    r = FrameRenderer(dev, cam, opt, want_stats=True)
    first = r.render().cpu().numpy().copy()
    second = r.render().cpu().numpy()
    img, _ = dev.render(cam, opt)
    assert np.array_equal(first.view(np.uint32), img.view(np.uint32))
    assert np.array_equal(second.view(np.uint32), img.view(np.uint32))
    assert r.last_stats.triangle_tests > 0
    torch.cuda.synchronize()


def test_row_bands_and_sample_ranges_tile_a_frame():
    """rt_accumulate_rows: 2 sample groups x 3 bands of rows add up to the frame rt_render draws, bit for bit
    inside a sample group and to fp32 addition order across the two."""
    from hip_mem import DeviceBuffer

    from raytracer_amd.distributed import frame_partition

    s = load("mesh_scene", 96, 70, 4)
    desc, cam = s.describe()
    dev = s.upload()
    opt = rt.make_options(seed=23)
    n = cam.width * cam.height
    buf, out = DeviceBuffer(4 * n), DeviceBuffer(3 * n)
    for rank in range(6):
        sb, se, rb, re = frame_partition(rank, 6, 2, cam.height)  # pretend spp = 2 groups: (0,1), (1,2)
        dev.accumulate(cam, opt, 2 * sb, 2 * se, buf.address, rows=(rb, re))
    whole = DeviceBuffer(4 * n)
    dev.accumulate(cam, opt, 0, 4, whole.address)
    assert np.array_equal(buf.to_host().view(np.uint32), whole.to_host().view(np.uint32))
    rt.resolve(buf.address, n, out.address)
    img, _ = dev.render(cam, opt)
    assert np.array_equal(out.to_host().view(np.uint32), img.ravel().view(np.uint32))
    with pytest.raises(rt.RtError):
        dev.accumulate(cam, opt, 0, 1, buf.address, rows=(3, cam.height + 1))
    dev.accumulate(cam, opt, 0, 1, buf.address, rows=(5, 5))  # empty band: nothing to do


def test_frames_in_flight_on_two_streams():
    """Two handles of one scene, two streams: consecutive frames overlap on the device and are still
    the frames a single render call produces."""
    import torch

    from raytracer_amd.distributed import FrameRenderer

    s = load("mesh_scene", 200, 150, 5)
    desc, cam = s.describe()
    dev = s.upload()
    r = FrameRenderer(dev, cam, rt.make_options(seed=50), extra_devs=[s.upload()])
    assert len(r.slots) == 2 and r.streams is not None
    images = [r.render_async(opt=rt.make_options(seed=50 + i)) for i in range(6)]
    r.wait()
    torch.cuda.current_stream().synchronize()
    for i in (4, 5):  # the two slots hold the last two frames
        want, _ = dev.render(cam, rt.make_options(seed=50 + i))
        assert np.array_equal(images[i].cpu().numpy().view(np.uint32), want.view(np.uint32)), f"frame {i}"
    # with statistics: the sampled frame waits for its own stream only
    r.want_stats = True
    img = r.render(opt=rt.make_options(seed=77, profile=True)).cpu().numpy()
    want, _ = dev.render(cam, rt.make_options(seed=77))
    assert np.array_equal(img.view(np.uint32), want.view(np.uint32))
    assert r.last_stats.paths == 200 * 150 * 5 and r.last_stats.mesh_kernel_launches == 6


def _two_rank_worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    # both ranks share the one GPU of the test box, which RCCL refuses; gloo moves the
    # buffers through the host and exercises the same driver code
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from raytracer_amd.distributed import FrameRenderer

        torch.cuda.set_device(0)
        rt.init(0)
        s = load("mesh_scene", 160, 120, 7)
        desc, cam = s.describe()
        dev = s.upload()
        r = FrameRenderer(dev, cam, rt.make_options(seed=31))
        img = r.render().cpu().numpy().copy()
        img2 = r.render().cpu().numpy()
        assert np.array_equal(img, img2)
        np.save(os.path.join(out_dir, f"rank{rank}.npy"), img)
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_the_gpu_reproduce_the_single_rank_frame(tmp_path):
    """Sample split (4 + 3 of 7 samples), HIP accumulate per rank, all-reduce, HIP resolve."""
    import socket

    import torch.multiprocessing as mp

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    mp.start_processes(_two_rank_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    a, b = np.load(tmp_path / "rank0.npy"), np.load(tmp_path / "rank1.npy")
    assert np.array_equal(a, b)
    s = load("mesh_scene", 160, 120, 7)
    desc, cam = s.describe()
    single, _ = s.upload().render(cam, rt.make_options(seed=31))
    # the split changes the association of one fp32 addition per channel
    np.testing.assert_allclose(a, single, rtol=3e-6, atol=1e-4)
    ref, _ = ob.render_image(desc, cam, ob.options(seed=31))
    np.testing.assert_allclose(a, ref, rtol=3e-6, atol=1e-4)


def _pipelined_worker(rank, world, port, out_dir, rank_mode):
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    try:
        from raytracer_amd.distributed import FrameRenderer

        rt.init(0)
        s = load("mesh_scene", 160, 120, 5)
        desc, cam = s.describe()
        dev = s.upload()
        extra = [s.upload()] if rank_mode == "two handles" else []
        r = FrameRenderer(dev, cam, rt.make_options(seed=40), exchange_in_group_of_one=True, extra_devs=extra)
        assert r.overlap and r.exchange and (r.streams is not None) == bool(extra)
        # seven frames with their own seeds through two slots, nothing waited for in between
        images = [r.render_async(opt=rt.make_options(seed=40 + i)) for i in range(7)]
        last_two = [images[5].clone(), images[6].clone()]  # clone is ordered on the current stream: too early on purpose
        r.wait()
        got = [images[5].cpu().numpy().copy(), images[6].cpu().numpy().copy()]
        for i, g in zip((5, 6), got):
            want, _ = dev.render(cam, rt.make_options(seed=40 + i))
            assert np.array_equal(g.view(np.uint32), want.view(np.uint32)), f"frame {i}"
        # the synchronous call returns a finished frame
        img = r.render(opt=rt.make_options(seed=99)).cpu().numpy()
        want, _ = dev.render(cam, rt.make_options(seed=99))
        assert np.array_equal(img.view(np.uint32), want.view(np.uint32))
        del last_two
        np.save(os.path.join(out_dir, f"ok_{rank_mode.replace(' ', '_')}.npy"), np.ones(1))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["side stream", "two handles"])
def test_pipelined_frames_over_rccl_group_of_one(tmp_path, mode):
    """Stream logic of the overlapped path (RCCL collective on its own stream, resolve on a side
    stream, buffer reuse after two frames), rehearsed on the one GPU of the test box."""
    import socket

    import torch.multiprocessing as mp

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    mp.start_processes(_pipelined_worker, args=(1, port, str(tmp_path), mode), nprocs=1, join=True, start_method="spawn")
    assert os.path.exists(tmp_path / f"ok_{mode.replace(' ', '_')}.npy")


def test_timing_marks_collected_after_the_call():
    """rt_options.profile = 2: the call records its marks on the stream and returns; rt_profile_collect reads the
    kernel times later.  They cover the same launches as the blocking statistics of profile = 1."""
    import hip_mem

    s = load("mesh_scene", 128, 96, 4)
    desc, cam = s.describe()
    dev = s.upload()
    n = cam.width * cam.height
    with pytest.raises(rt.RtError):
        dev.profile_collect()  # nothing outstanding
    sums = hip_mem.DeviceBuffer(n * 4)
    sums.zero()
    blocking = dev.accumulate(cam, rt.make_options(seed=3, profile=1), 0, 4, sums.address, want_stats=True)
    first = sums.to_host().copy()
    sums.zero()
    assert dev.accumulate(cam, rt.make_options(seed=3, profile=2), 0, 4, sums.address) is None
    other = hip_mem.DeviceBuffer(n * 4)
    dev.accumulate(cam, rt.make_options(seed=3), 0, 1, other.address)  # a call without marks leaves them alone
    marks = dev.profile_collect()
    assert np.array_equal(sums.to_host().view(np.uint32), first.view(np.uint32))
    assert marks.shade_kernel_launches == blocking.shade_kernel_launches == 5
    assert marks.mesh_kernel_launches == blocking.mesh_kernel_launches == 6
    assert marks.gpu_ms > 0 and marks.shade_kernel_ms > 0 and marks.mesh_kernel_ms > 0 and marks.bin_kernel_ms > 0
    assert marks.shade_kernel_ms + marks.mesh_kernel_ms + marks.bin_kernel_ms <= marks.gpu_ms
    assert marks.paths == 0 and marks.triangle_tests == 0  # timings only
    with pytest.raises(rt.RtError):
        dev.profile_collect()  # already collected
    # a scene without triangles is one kernel: only the whole call is timed
    spheres = load("default", 128, 96, 4)
    _, scam = spheres.describe()
    sdev = spheres.upload()
    sdev.accumulate(scam, rt.make_options(seed=3, profile=2), 0, 4, sums.address)
    smarks = sdev.profile_collect()
    assert smarks.gpu_ms > 0 and smarks.shade_kernel_launches == 0 and smarks.mesh_kernel_launches == 0

