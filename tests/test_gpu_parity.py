"""HIP path vs the CPU oracle on identical inputs (run on the MI355X box).

Both sides draw from the same counter-based generator and use the same fixed
polynomial kernels for log/sin/cos/pow, and both are compiled without
floating-point contraction, so the comparison is BIT-EXACT in every mode.
"""
import os
import numpy as np
import pytest

import oracle_binding as ob
import raytracer_amd as rt
from helpers import MODES, load, same_bits, variant

pytestmark = pytest.mark.gpu


def rays_for(arr, n, seed):
    """Random rays that start inside the room and mostly aim at the mesh."""
    rng = np.random.default_rng(seed)
    o = rng.uniform(-45, 45, size=(n, 3)).astype(np.float32)
    target = rng.uniform(-20, 20, size=(n, 3)).astype(np.float32)
    d = target - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d.astype(np.float32)


@pytest.mark.parametrize("scene_name", ["mesh_scene", "default", "simple"])
@pytest.mark.parametrize("bbox", [True, False])
def test_cast_matches_oracle(scene_name, bbox):
    s = load(scene_name)
    desc, cam = s.describe()
    dev = s.upload()
    o, d = rays_for(None, 20000, 7)
    opt = rt.make_options(bounding_box=bbox)
    t, sid, tid, abg = dev.cast(opt, o, d)
    rt_, rsid, rtid, rabg = ob.cast(desc, ob.from_rt_options(opt), o, d)
    assert np.array_equal(sid, rsid)
    assert np.array_equal(tid, rtid)
    assert same_bits(t, rt_)
    hit = rtid >= 0
    assert same_bits(abg[hit], rabg[hit])
    if scene_name == "mesh_scene":
        assert hit.sum() > 1000  # the mesh really was exercised


@pytest.mark.parametrize("scene_name", ["mesh_scene", "default"])
def test_visible_matches_oracle(scene_name):
    s = load(scene_name)
    desc, cam = s.describe()
    dev = s.upload()
    rng = np.random.default_rng(11)
    o = rng.uniform(-40, 40, size=(20000, 3)).astype(np.float32)
    tgt = np.tile(np.array([[-10, 20, 40]], np.float32), (len(o), 1))
    opt = rt.make_options()
    vis = dev.visible(opt, o, tgt)
    ref = ob.visible(desc, ob.from_rt_options(opt), o, tgt)
    assert np.array_equal(vis, ref)
    assert 0 < ref.sum() < len(ref)


@pytest.mark.parametrize("scene_name", ["default", "mesh_scene"])
def test_visible_segments_that_end_or_start_on_a_sphere(scene_name):
    """Shadow rays are scanned with a shortcut: a sphere whose roots are certainly outside [0, length] skips the
    exact square root.  Segments whose end point or start point lies ON a sphere (to rounding, and a few units
    in the last place to either side) are the cases where "certainly" must not be claimed: the verdicts have
    to be the oracle's, which always takes the exact root."""
    s = load(scene_name)
    desc, cam = s.describe()
    dev = s.upload()
    o, d = rays_for(None, 6000, 21)
    opt = rt.make_options()
    t, sid, tid, _ = ob.cast(desc, ob.from_rt_options(opt), o, d)
    on_sphere = (sid >= 0) & np.isfinite(t)
    o, d, t = o[on_sphere], d[on_sphere], t[on_sphere]
    assert len(o) > 3000
    starts, targets = [], []
    for scale in (1.0, 1.0 - 2.0 ** -23, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -21, 1.0 + 2.0 ** -21, 1.0 - 1e-6, 1.0 + 1e-6):
        hit = (o + (np.float32(scale) * t)[:, None] * d).astype(np.float32)
        starts.append(o)           # ends on (or a hair before / behind) the sphere it hits first
        targets.append(hit)
        starts.append(hit)         # starts there, towards the light and back to where it came from
        targets.append(np.tile(np.array([[-10, 20, 40]], np.float32), (len(o), 1)))
        starts.append(hit)
        targets.append(o)
    a = np.concatenate(starts).astype(np.float32)
    b = np.concatenate(targets).astype(np.float32)
    vis = dev.visible(opt, a, b)
    ref = ob.visible(desc, ob.from_rt_options(opt), a, b)
    assert np.array_equal(vis, ref)
    assert 0.02 < ref.mean() < 0.98


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("scene_name", ["simple", "default", "mesh_scene"])
def test_render_matches_oracle_bitwise(scene_name, mode):
    indirect, aa, bbox = MODES[mode]
    s = load(scene_name, 48, 40, 3)
    desc, cam = s.describe()
    dev = s.upload()
    opt = rt.make_options(indirect, aa, bbox, seed=5)
    img, st = dev.render(cam, opt)
    ref, counters = ob.render_image(desc, cam, ob.from_rt_options(opt))
    assert img.shape == ref.shape
    assert same_bits(img, ref), f"max abs diff {np.nanmax(np.abs(img - ref))}"
    if desc.n_triangles and (indirect or aa):
        # the GPU walks the triangle list for the rays the oracle walks it for,
        # minus the shadow rays a sphere already blocks
        assert 0 < st.triangle_tests <= counters["triangle_tests"]


@pytest.mark.parametrize("material", ["glass", "mirror"])
def test_cat_glass_and_mirror_match_oracle(material):
    s = variant("mesh_scene", material, 40, 40, 2)
    desc, cam = s.describe()
    dev = s.upload()
    opt = rt.make_options(seed=9)
    img, _ = dev.render(cam, opt)
    ref, _ = ob.render_image(desc, cam, ob.from_rt_options(opt))
    assert same_bits(img, ref), f"max abs diff {np.nanmax(np.abs(img - ref))}"


# ---- exhaustive scan vs cluster culling -----------------------------------
# make_options() turns cluster culling on, so every test above runs the culled
# pipeline; the tests below pin the exhaustive scan (the reference's own loop)
# and compare the two modes with each other at full size.

@pytest.mark.parametrize("scene_name", ["mesh_scene", "simple"])
@pytest.mark.parametrize("bbox", [True, False])
def test_exhaustive_scan_cast_matches_oracle(scene_name, bbox):
    s = load(scene_name)
    desc, cam = s.describe()
    dev = s.upload()
    o, d = rays_for(None, 20000, 17)
    opt = rt.make_options(bounding_box=bbox, cluster_culling=False)
    t, sid, tid, abg = dev.cast(opt, o, d)
    rt_, rsid, rtid, rabg = ob.cast(desc, ob.from_rt_options(opt), o, d)
    assert np.array_equal(sid, rsid) and np.array_equal(tid, rtid)
    assert same_bits(t, rt_)
    assert same_bits(abg[rtid >= 0], rabg[rtid >= 0])


@pytest.mark.parametrize("mode", list(MODES))
def test_exhaustive_scan_render_matches_oracle_bitwise(mode):
    indirect, aa, bbox = MODES[mode]
    s = load("mesh_scene", 48, 40, 3)
    desc, cam = s.describe()
    opt = rt.make_options(indirect, aa, bbox, seed=5, cluster_culling=False)
    img, st = s.upload().render(cam, opt)
    ref, counters = ob.render_image(desc, cam, ob.from_rt_options(opt))
    assert same_bits(img, ref)
    if indirect or aa:
        assert 0 < st.triangle_tests <= counters["triangle_tests"]


def test_culled_and_exhaustive_frames_are_identical_at_full_size():
    s = load("mesh_scene")
    desc, cam = s.describe()
    dev = s.upload()
    a, sa = dev.render(cam, rt.make_options(seed=1, cluster_culling=True))
    b, sb = dev.render(cam, rt.make_options(seed=1, cluster_culling=False))
    differing = int((a.view(np.uint32) != b.view(np.uint32)).any(axis=2).sum())
    print(f"culled {sa.triangle_tests / sa.paths:.0f} vs exhaustive {sb.triangle_tests / sb.paths:.0f} triangle tests per path; "
          f"{differing} of {a.shape[0] * a.shape[1]} pixels differ")
    assert differing == 0
    # the default mode also drops the mesh walk of shadow rays whose contribution is exactly zero
    print(f"mesh rays: culled mode {sa.mesh_rays}, exhaustive mode {sb.mesh_rays}")
    assert 0.5 * sb.mesh_rays < sa.mesh_rays < sb.mesh_rays
    assert sa.triangle_tests < 0.35 * sb.triangle_tests


@pytest.mark.parametrize("pairs", ["1", "4", "8", "64", "1000"])
def test_other_cluster_sizes_give_the_same_image(pairs, monkeypatch):
    """1 pair per cluster gives the cat 1 977 clusters: more than the kernel with the second level in LDS takes,
    so the per-cluster lists in global memory (k_bin_groups + k_mesh_clusters), which meshes of more than about
    200 000 triangles use when they are kept on two culling levels.  4 and 8 pairs give more than 64 clusters (two levels, second one in LDS), 64 fewer,
    1000 two clusters."""
    monkeypatch.setenv("RT_AMD_CLUSTER_PAIRS", pairs)
    s = variant("mesh_scene", "glass", 72, 56, 3)
    desc, cam = s.describe()
    dev = s.upload()
    opt = rt.make_options(seed=23)
    img, st = dev.render(cam, opt)
    ref, _ = ob.render_image(desc, cam, ob.from_rt_options(opt))
    assert same_bits(img, ref)
    o, d = rays_for(None, 30000, 5)
    t, sid, tid, abg = dev.cast(opt, o, d)
    rt_, rsid, rtid, rabg = ob.cast(desc, ob.from_rt_options(opt), o, d)
    assert np.array_equal(tid, rtid) and same_bits(t, rt_)


@pytest.mark.parametrize("super_groups", [None, "2", "64"])
def test_three_culling_levels_give_the_same_image(super_groups, monkeypatch):
    """Meshes of more than 1 024 clusters get a third level (super-groups of groups of clusters, k_plan_runs and the
    many-groups instance of k_group_clusters).  Forced onto the cat with one pair per cluster: 1 977 clusters in 124
    groups, in 11 super-groups of 12 by default, in 62 of 2, in 2 of 64."""
    monkeypatch.setenv("RT_AMD_LEVELS", "3")
    monkeypatch.setenv("RT_AMD_CLUSTER_PAIRS", "1")
    if super_groups:
        monkeypatch.setenv("RT_AMD_SUPER_GROUPS", super_groups)
    s = variant("mesh_scene", "glass", 72, 56, 3)
    desc, cam = s.describe()
    dev = s.upload()
    opt = rt.make_options(seed=23)
    img, st = dev.render(cam, opt)
    ref, _ = ob.render_image(desc, cam, ob.from_rt_options(opt))
    assert same_bits(img, ref)
    full, st_full = dev.render(cam, rt.make_options(seed=23, cluster_culling=False))
    assert same_bits(img, full) and 0 < st.triangle_tests < st_full.triangle_tests / 8
    o, d = rays_for(None, 30000, 5)
    t, sid, tid, abg = dev.cast(opt, o, d)
    rt_, rsid, rtid, rabg = ob.cast(desc, ob.from_rt_options(opt), o, d)
    assert np.array_equal(tid, rtid) and same_bits(t, rt_)


@pytest.mark.parametrize("pairs", ["1", "8", "64", "1000"])
def test_axis_parallel_rays_through_the_cluster_filter(pairs, monkeypatch):
    """Rays with one or two exactly-zero direction components, cast without the object boxes, so that
    they reach the cluster filter: origins on both sides of zero on the parallel axes, inside and
    outside the slabs of clusters that straddle zero.  8 pairs per cluster give the cat more than 64
    clusters (two filter levels), 64 fewer (one level), 1000 two clusters."""
    monkeypatch.setenv("RT_AMD_CLUSTER_PAIRS", pairs)
    s = load("mesh_scene")
    desc, cam = s.describe()
    dev = s.upload()
    g = np.linspace(-14, 16, 61, dtype=np.float32)
    o, d = [], []
    for a in g:
        for b in g[::2]:
            o += [[a, b, 40], [a, b, -40], [40, a, b], [-40, a, b], [a, 40, b], [a, -40, b]]
            d += [[0, 0, -1], [0, 0, 1], [-1, 0, 0], [1, 0, 0], [0, -1, 0], [0, 1, 0]]
            # one zero component, the other two oblique
            o += [[a, b, 40], [40, a, b], [a, 40, b]]
            d += [[0, 0.6, -0.8], [-0.8, 0, 0.6], [0.6, -0.8, 0]]
    o, d = np.array(o, np.float32), np.array(d, np.float32)
    opt = rt.make_options(bounding_box=False, cluster_culling=True)
    t, sid, tid, abg = dev.cast(opt, o, d)
    te, side, tide, abge = dev.cast(rt.make_options(bounding_box=False, cluster_culling=False), o, d)
    rt_, rsid, rtid, rabg = ob.cast(desc, ob.from_rt_options(opt), o, d)
    assert (rtid >= 0).sum() > 500  # many of them do hit the cat
    assert np.array_equal(tide, rtid) and same_bits(te, rt_)
    assert np.array_equal(tid, rtid), f"{(tid != rtid).sum()} rays lose or change their triangle under culling"
    assert np.array_equal(sid, rsid) and same_bits(t, rt_)
    assert same_bits(abg[rtid >= 0], rabg[rtid >= 0])
    # the same rays as shadow segments
    vis = dev.visible(opt, o, o + 100 * d)
    assert np.array_equal(vis, ob.visible(desc, ob.from_rt_options(opt), o, o + 100 * d))


def test_random_cos_matches_oracle_including_the_argmin_quirk():
    """random_cos on the device against the oracle's, bit for bit: random normals, normals with a negative
    x (the reference's argmin_abs starts its running minimum at the SIGNED first component, so it answers 0
    whenever N.x < 0, include/vector.cuh:87-98), axis-aligned normals, and N = (-1, 0, 0), whose tangent is
    the zero vector: the direction is NaN and the path ends."""
    rng = np.random.default_rng(3)
    n = 4096
    normals = rng.normal(size=(n, 3)).astype(np.float32)
    normals /= np.linalg.norm(normals, axis=1, keepdims=True)
    special = np.array([[-1, 0, 0], [1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1], [-0.6, 0.8, 0], [-0.6, 0, 0.8],
                        [-1e-3, 0.6, 0.8], [0.6, -0.8, 0], [0.57735026, 0.57735026, 0.57735026]], np.float32)
    normals[:len(special)] = special
    r1 = rng.uniform(0, 1, n).astype(np.float32)
    r2 = rng.uniform(0, 1, n).astype(np.float32)
    r1[20:24] = [1.0, 5.9604645e-08, 0.25, 0.5]  # the ends of the generator's (0, 1] range
    r2[24:28] = [1.0, 5.9604645e-08, 1.0, 0.5]
    got = rt.random_cos(normals, r1, r2)
    want = ob.random_cos(normals, r1, r2)
    assert np.isnan(got[0]).all() and np.isnan(want[0]).all()  # N = (-1, 0, 0)
    assert same_bits(got, want)
    finite = ~np.isnan(got).any(axis=1)
    assert finite.sum() >= n - 1
    np.testing.assert_allclose(np.linalg.norm(got[finite], axis=1), 1.0, atol=2e-6)
    assert (np.einsum("ij,ij->i", got[finite], normals[finite]) >= -1e-6).all()  # in the hemisphere of N


@pytest.mark.parametrize("group", ["4", "6", "16"])
def test_other_group_sizes_give_the_same_image(group, monkeypatch):
    """Groups of 4, 6 and 16 clusters (62, 42 and 16 groups for the cat): the kernel that bins a group's rays
    into its clusters in LDS and scans them handles partly filled box pairs and last groups of any size."""
    monkeypatch.setenv("RT_AMD_GROUP_CLUSTERS", group)
    s = variant("mesh_scene", "glass", 72, 56, 3)
    desc, cam = s.describe()
    dev = s.upload()
    opt = rt.make_options(seed=29)
    img, st = dev.render(cam, opt)
    ref, _ = ob.render_image(desc, cam, ob.from_rt_options(opt))
    assert same_bits(img, ref)
    o, d = rays_for(None, 30000, 6)
    t, sid, tid, abg = dev.cast(rt.make_options(bounding_box=False), o, d)
    rt_, rsid, rtid, rabg = ob.cast(desc, ob.options(bounding_box=False), o, d)
    assert np.array_equal(tid, rtid) and same_bits(t, rt_)


def test_triangle_order_changes_the_work_not_the_image(monkeypatch):
    """The k-d order of the triangles (upload) against the plain Morton order it refines: same pixels and
    same hits bit for bit, fewer triangle tests."""
    s = variant("mesh_scene", "diffuse", 96, 80, 4)
    desc, cam = s.describe()
    opt = rt.make_options(seed=31)
    o, d = rays_for(None, 20000, 9)
    results = {}
    for order in ("kd", "morton"):
        monkeypatch.setenv("RT_AMD_ORDER", order)
        dev = s.upload()
        img, st = dev.render(cam, opt)
        t, sid, tid, abg = dev.cast(rt.make_options(bounding_box=False), o, d)
        results[order] = (img, st.triangle_tests, t, tid)
    ref, _ = ob.render_image(desc, cam, ob.from_rt_options(opt))
    assert same_bits(results["kd"][0], ref) and same_bits(results["morton"][0], ref)
    assert np.array_equal(results["kd"][3], results["morton"][3]) and same_bits(results["kd"][2], results["morton"][2])
    assert results["kd"][1] < 0.8 * results["morton"][1]


def test_counted_triangle_tests_and_footprint_of_a_scene_handle():
    """Full-size headline frame: the ray-cluster pairs counted on the device give about 32 triangle tests per
    path (7 175 in the exhaustive scan; 63 with the clusters in plain Morton order), and the handle's work buffers stay far below what per-cluster ray lists in global
    memory took (18 GB for this frame)."""
    import hip_mem

    s = load("mesh_scene")
    desc, cam = s.describe()
    before = hip_mem.free_bytes()
    dev = s.upload()
    img, st = dev.render(cam, rt.make_options(seed=1))
    used = before - hip_mem.free_bytes()
    print(f"{st.triangle_tests / st.paths:.1f} triangle tests per path, {used / 2**30:.2f} GiB held by the handle")
    assert st.triangle_tests / st.paths == pytest.approx(32.3, rel=0.1)
    assert used < 4 * 2**30
    # (path, bounce) pairs the bounce kernel processed: every path at the first bounce, fewer and fewer afterwards (a
    # closed room: paths end when their weight becomes zero or at the fifth bounce) -- 2.9 bounces per path on this frame
    assert st.paths < st.path_bounces < 5 * st.paths
    assert st.path_bounces / st.paths == pytest.approx(2.94, abs=0.2)
    # a frame of one bounce: exactly one pair per path
    _, st1 = dev.render(cam, rt.make_options(seed=1, max_depth=1))
    assert st1.path_bounces == st1.paths
    # the exact arithmetic fetches a path for every bounce; the fast one keeps a path in its lane while it needs nothing
    # from the triangle stage (most bounces of this frame: the room is large, the cat small), and counts the same bounces
    if "RT_AMD_EXACT_STAYS" not in os.environ:  # (the validation knob runs the exact arithmetic through the fast kernel's control flow)
        assert st.path_records == st.path_bounces
    _, stf = dev.render(cam, rt.make_options(seed=1, arithmetic=1))
    assert stf.path_bounces / stf.paths == pytest.approx(2.94, abs=0.2)
    assert stf.paths < stf.path_records < 0.7 * stf.path_bounces
    print(f"fast arithmetic: {stf.path_records / stf.paths:.2f} fetches for {stf.path_bounces / stf.paths:.2f} bounces per path")
