"""Randomly composed scenes (spheres of every material, 0-2 procedural meshes with their own
transforms and materials, 1-3 lights, a moved and rotated camera): the HIP path, with and without
cluster culling, against the oracle, whole frames, bit for bit."""
import os
import sys

import numpy as np
import pytest

import oracle_binding as ob
import raytracer_amd as rt
from helpers import MODES, same_bits

sys.path.insert(0, os.path.join(rt.REPO_ROOT, "tools"))
import make_mesh  # noqa: E402

pytestmark = pytest.mark.gpu


def random_scene(seed, tmp_path):
    rng = np.random.default_rng(seed)
    s = rt.Scene()
    s.set_camera_intrinsics(float(rng.uniform(40, 80)), 56, 44)
    s.set_num_rays(2)
    # a closed room of six large spheres, as in the reference's scenes, so that paths keep bouncing
    for axis in range(3):
        for sign in (-1.0, 1.0):
            c = [0.0, 0.0, 0.0]
            c[axis] = sign * 1000.0
            s.add_sphere(c, 940.0, rng.uniform(0.2, 1.0, 3), rt.DIFFUSE)
    for _ in range(int(rng.integers(1, 6))):
        kind = int(rng.integers(0, 3))
        s.add_sphere(rng.uniform(-25, 25, 3), float(rng.uniform(2, 9)), rng.uniform(0.1, 1.0, 3), kind, 1.0,
                     float(rng.uniform(1.2, 1.8)))
    for k in range(int(rng.integers(0, 3))):
        n_lat, n_lon = int(rng.integers(5, 40)), int(rng.integers(6, 60))
        v, n, t = make_mesh.bumpy_sphere(n_lat, n_lon, radius=float(rng.uniform(4, 10)), bump=float(rng.uniform(0, 0.3)))
        path = tmp_path / f"mesh_{seed}_{k}.obj"
        make_mesh.write_obj(str(path), v, n, t)
        s.add_object(str(path), rng.uniform(-15, 15, 3), rng.uniform(0.5, 1.5, 3), rng.uniform(0, 360, 3),
                     rng.uniform(0.1, 1.0, 3), int(rng.integers(0, 3)), 1.0, float(rng.uniform(1.2, 1.8)))
    for _ in range(int(rng.integers(1, 4))):
        s.add_light(rng.uniform(-30, 30, 3), float(rng.uniform(1e9, 1e10)))
    s.rotate_camera(*rng.uniform(-20, 20, 3))
    s.transform_camera(*rng.uniform(-10, 10, 2), float(rng.uniform(30, 50)))
    return s


@pytest.mark.parametrize("seed", range(40))
def test_random_scene_matches_the_oracle(seed, tmp_path):
    s = random_scene(seed, tmp_path)
    desc, cam = s.describe()
    dev = s.upload()
    names = list(MODES)
    for mode in (names[seed % len(names)], "indirect_aa_bbox"):
        flags = MODES[mode]
        ref, _ = ob.render_image(desc, cam, ob.options(*flags, seed=seed))
        for culling in (True, False):
            img, _ = dev.render(cam, rt.make_options(*flags, seed=seed, cluster_culling=culling))
            assert same_bits(img, ref), (f"seed {seed} mode {mode} culling {culling}: {(img != ref).any(axis=2).sum()} pixels differ, "
                                         f"{desc.n_triangles} triangles, {desc.n_spheres} spheres, {desc.n_lights} lights")


def test_more_spheres_than_the_lds_cache_holds(tmp_path):
    """100 spheres: the first 64 are read from LDS, the rest from global memory."""
    rng = np.random.default_rng(99)
    s = random_scene(3, tmp_path)
    for _ in range(95):
        s.add_sphere(rng.uniform(-30, 30, 3), float(rng.uniform(0.5, 3)), rng.uniform(0.1, 1.0, 3), int(rng.integers(0, 3)), 1.0, 1.5)
    desc, cam = s.describe()
    assert desc.n_spheres > 100
    opt = rt.make_options(seed=5)
    img, _ = s.upload().render(cam, opt)
    ref, _ = ob.render_image(desc, cam, ob.from_rt_options(opt))
    assert same_bits(img, ref)


def test_more_materials_and_objects_than_the_lds_cache_holds(tmp_path):
    """150 spheres with a material each (the kernels keep 128 materials and 64 spheres in LDS) and
    eleven meshes (8 object boxes in LDS): the entries beyond are read from global memory."""
    rng = np.random.default_rng(5)
    s = random_scene(3, tmp_path)
    for _ in range(150):
        s.add_sphere(rng.uniform(-30, 30, 3), float(rng.uniform(0.5, 2)), rng.uniform(0.1, 1.0, 3), int(rng.integers(0, 3)), 1.0, 1.5)
    v, n, t = make_mesh.bumpy_sphere(6, 9, radius=2.5, bump=0.2)
    path = tmp_path / "small.obj"
    make_mesh.write_obj(str(path), v, n, t)
    for _ in range(11):
        s.add_object(str(path), rng.uniform(-20, 20, 3), rng.uniform(0.7, 1.3, 3), rng.uniform(0, 360, 3),
                     rng.uniform(0.1, 1.0, 3), int(rng.integers(0, 3)), 1.0, 1.5)
    desc, cam = s.describe()
    assert desc.n_materials > 128 and desc.n_boxes > 8 and desc.n_spheres > 64
    dev = s.upload()
    for bbox in (True, False):
        ref, _ = ob.render_image(desc, cam, ob.options(bounding_box=bbox, seed=5))
        for culling in (True, False):
            img, _ = dev.render(cam, rt.make_options(bounding_box=bbox, seed=5, cluster_culling=culling))
            assert same_bits(img, ref)


def test_twelve_lights_with_meshes(tmp_path):
    """Every light's shadow ray may have to wait for the triangle kernel: twelve pending slots per hit."""
    rng = np.random.default_rng(7)
    s = random_scene(1, tmp_path)  # seed 1 has meshes
    for _ in range(10):
        s.add_light(rng.uniform(-30, 30, 3), float(rng.uniform(5e8, 3e9)))
    desc, cam = s.describe()
    assert desc.n_lights >= 11 and desc.n_triangles > 0
    dev = s.upload()
    for flags in (MODES["indirect_aa_bbox"], MODES["aa_bbox"]):
        ref, _ = ob.render_image(desc, cam, ob.options(*flags, seed=9))
        for culling in (True, False):
            img, _ = dev.render(cam, rt.make_options(*flags, seed=9, cluster_culling=culling))
            assert same_bits(img, ref)


def test_degenerate_triangles(tmp_path):
    """Zero-area and repeated-vertex faces have N = 0: the denominator u.N is zero and the
    reference's comparisons reject the NaN / inf results; so must both triangle kernels."""
    p = tmp_path / "degenerate.obj"
    p.write_text("v -6 -6 0\nv 6 -6 0\nv 0 7 0\nv 0 7 0\nv 3 0.5 0\nv 12 2 -3\nvn 0 0 1\n"
                 "f 1//1 2//1 3//1\nf 1//1 1//1 2//1\nf 1//1 5//1 3//1\nf 3//1 4//1 6//1\nf 2//1 6//1 3//1\n")
    s = rt.Scene()
    s.load_scene(os.path.join(rt.ASSETS, "scenes", "simple.json"))
    s.set_camera_intrinsics(60.0, 72, 60)
    s.set_num_rays(3)
    s.add_object(str(p), [0, 0, 15], [1, 1, 1], [0, 20, 0], [0.8, 0.8, 0.2], rt.DIFFUSE)
    desc, cam = s.describe()
    assert desc.n_triangles == 5
    dev = s.upload()
    for bbox in (True, False):
        ref, _ = ob.render_image(desc, cam, ob.options(bounding_box=bbox, seed=2))
        for culling in (True, False):
            img, _ = dev.render(cam, rt.make_options(bounding_box=bbox, seed=2, cluster_culling=culling))
            assert same_bits(img, ref)
