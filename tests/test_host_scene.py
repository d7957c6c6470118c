"""Host scene layer (C++ Scene behind the C ABI): loader, camera, mesh flattening,
error behaviour and the PNG writer.  CPU only.
"""
import json
import math
import os

import numpy as np
import pytest
from PIL import Image

import raytracer_amd as rt
from helpers import SCENES, load

f32 = np.float32


# ---------------------------------------------------------------------------
# An independent numpy statement of the reference's scene set-up
# (reference src/scene.cu:37-67,85-207,224-262,278-413)
# ---------------------------------------------------------------------------

def rotation(alpha, beta, gamma):
    def cs(deg):
        a = f32(float(f32(deg)) * math.pi / 180)
        return f32(math.cos(float(a))), f32(math.sin(float(a)))

    ca, sa = cs(alpha)
    cb, sb = cs(beta)
    cg, sg = cs(gamma)
    rx = np.array([[1, 0, 0], [0, ca, -sa], [0, sa, ca]], f32)
    ry = np.array([[cb, 0, sb], [0, 1, 0], [-sb, 0, cb]], f32)
    rz = np.array([[cg, -sg, 0], [sg, cg, 0], [0, 0, 1]], f32)
    return matmul(matmul(rz, ry), rx)


def matmul(a, b):
    n = a.shape[0]
    c = np.zeros((n, n), f32)
    for i in range(n):
        for j in range(n):
            acc = f32(0)
            for k in range(n):
                acc = f32(acc + f32(a[i, k] * b[k, j]))
            c[i, j] = acc
    return c


def expected_camera(doc):
    cam = doc["camera"]
    e = np.diag([1, -1, -1]).astype(f32)
    rot = matmul(rotation(*cam["rotation"]), e)
    w, h = cam["width"], cam["height"]
    fov = f32(cam["fov"])
    f = f32(f32(w) / f32(2 * f32(math.tan(float(f32(math.pi * float(fov) / float(f32(360.0))))))))
    return rot, np.array(cam["pos"], f32), f, f32(w // 2), f32(h // 2)


def parse_obj(path):
    pos, nrm, faces = [], [], []
    for line in open(path):
        t = line.split()
        if not t:
            continue
        if t[0] == "v":
            pos.append([float(x) for x in t[1:4]])
        elif t[0] == "vn":
            nrm.append([float(x) for x in t[1:4]])
        elif t[0] == "f":
            corners = []
            for c in t[1:]:
                parts = c.split("/")
                vi = int(parts[0])
                ni = int(parts[2]) if len(parts) > 2 and parts[2] else 0
                corners.append((vi - 1 if vi > 0 else len(pos) + vi, ni - 1 if ni > 0 else (len(nrm) + ni if ni else -1)))
            for k in range(1, len(corners) - 1):
                faces.append([corners[0], corners[k], corners[k + 1]])
    return np.array(pos, f32), np.array(nrm, f32), faces


def expected_mesh(obj_path, pos, scale, rot):
    p, n, faces = parse_obj(obj_path)
    m = matmul(rotation(*rot), np.diag(np.array(scale, f32)))  # R * S

    def apply(v):
        out = np.zeros(3, f32)
        for i in range(3):
            acc = f32(0)
            for j in range(3):
                acc = f32(acc + f32(m[i, j] * v[j]))
            out[i] = acc
        return out

    verts, norms = [], []
    for face in faces:
        for vi, ni in face:
            verts.append(apply(p[vi]) + np.array(pos, f32))
            t = apply(n[ni])
            length = f32(math.sqrt(float(f32(f32(f32(t[0] * t[0]) + f32(t[1] * t[1])) + f32(t[2] * t[2])))))
            norms.append(t / length)
    return np.array(verts, f32), np.array(norms, f32)


# ---------------------------------------------------------------------------

@pytest.mark.parametrize("name", ["default", "simple", "mesh_scene"])
def test_camera_and_spheres_match_scene_file(name):
    doc = json.load(open(os.path.join(SCENES, name + ".json")))
    s = load(name)
    desc, cam = s.describe()
    rot, pos, f, cx, cy = expected_camera(doc)
    np.testing.assert_array_equal(np.array(cam.rot, f32).reshape(3, 3), rot)
    np.testing.assert_array_equal(np.array(cam.pos, f32), pos)
    # glibc's tanf may differ from the rounded double tangent by one unit in the last place
    assert abs(cam.fx - f) <= 3e-7 * f and cam.fy == cam.fx and cam.cx == cx and cam.cy == cy
    assert (cam.width, cam.height, cam.num_rays) == (doc["camera"]["width"], doc["camera"]["height"], doc["camera"]["num_rays"])
    a = rt.desc_arrays(desc)
    assert desc.n_spheres == len(doc["spheres"])
    for i, sp in enumerate(doc["spheres"]):
        np.testing.assert_array_equal(a["sphere_pos"][i], np.array(sp["pos"], f32))
        assert a["sphere_radius"][i] == f32(sp["radius"])
        m = a["sphere_material"][i]
        kind = {"diffuse": rt.DIFFUSE, "mirror": rt.MIRROR, "glass": rt.GLASS}[sp["material"]]
        assert a["material_type"][m] == kind
        # colour is read for diffuse only, refractive indices for glass only
        want_colour = sp["color"] if kind == rt.DIFFUSE else [0, 0, 0]
        np.testing.assert_array_equal(a["material_color"][m], np.array(want_colour, f32))
        if kind == rt.GLASS:
            assert (a["material_ro"][m], a["material_ri"][m]) == tuple(f32(x) for x in sp["refractive_index"])
    assert desc.n_lights == len(doc["lights"])
    for i, li in enumerate(doc["lights"]):
        np.testing.assert_array_equal(a["light_pos"][i], np.array(li["pos"], f32))
        assert a["light_intensity"][i] == f32(li["intensity"])


def test_mesh_flattening_matches_independent_statement():
    doc = json.load(open(os.path.join(SCENES, "mesh_scene.json")))
    o = doc["objects"][0]
    s = load("mesh_scene")
    a = s.arrays()
    verts, norms = expected_mesh(os.path.join(rt.ASSETS, "objects", "cat", "cat.obj"), o["pos"], o["scale"], o["rotation"])
    assert a["vertices"].shape == verts.shape == (3 * 3954, 3)
    # rotation entries come from libm cosf/sinf on one side and rounded double cos/sin on the other
    np.testing.assert_allclose(a["vertices"], verts, rtol=0, atol=4e-6)
    np.testing.assert_allclose(a["normals"], norms, rtol=0, atol=3e-7)
    # one fresh vertex and normal per corner, numbered consecutively
    np.testing.assert_array_equal(a["tri_v"].ravel(), np.arange(3 * 3954))
    np.testing.assert_array_equal(a["tri_n"].ravel(), np.arange(3 * 3954))
    np.testing.assert_allclose(a["box_min"][0], a["vertices"].min(0), atol=0)
    np.testing.assert_allclose(a["box_max"][0], a["vertices"].max(0), atol=0)
    np.testing.assert_allclose(np.linalg.norm(a["normals"], axis=1), 1.0, atol=1e-6)


def test_loading_twice_starts_from_scratch():
    s = load("mesh_scene")
    first = s.arrays()
    cam1 = np.array(s.describe()[1].rot)
    s.load_scene(os.path.join(SCENES, "mesh_scene.json"))
    second = s.arrays()
    for k in first:
        np.testing.assert_array_equal(first[k], second[k])
    np.testing.assert_array_equal(cam1, np.array(s.describe()[1].rot))


def test_missing_scene_file_reports_file_not_found():
    s = rt.Scene()
    with pytest.raises(rt.RtError) as e:
        s.load_scene("/nonexistent/scene.json")
    assert e.value.code == -3 and "File not found" in str(e.value)


def test_malformed_scene_file(tmp_path):
    p = tmp_path / "bad.json"
    p.write_text('{"camera": {"pos": [0, 0, 1], ')
    with pytest.raises(rt.RtError) as e:
        rt.Scene().load_scene(str(p))
    assert e.value.code == -4


def test_missing_mesh_file(tmp_path):
    doc = json.load(open(os.path.join(SCENES, "mesh_scene.json")))
    doc["objects"][0]["filepath"] = "no/such/mesh.obj"
    p = tmp_path / "scene.json"
    p.write_text(json.dumps(doc))
    with pytest.raises(rt.RtError) as e:
        rt.Scene().load_scene(str(p))
    assert e.value.code == -3 and "mesh.obj" in str(e.value)


def test_scene_without_objects_or_lights(tmp_path):
    p = tmp_path / "min.json"
    p.write_text(json.dumps({"camera": {"pos": [0, 0, 5], "rotation": [0, 0, 0], "width": 8, "height": 4, "fov": 90,
                                        "num_rays": 1}, "spheres": []}))
    s = rt.Scene()
    s.load_scene(str(p))
    d, c = s.describe()
    assert (d.n_spheres, d.n_triangles, d.n_lights, d.n_boxes) == (0, 0, 0, 0)
    assert c.fx == f32(8 / (2 * math.tan(math.pi / 4))) or abs(c.fx - 4.0) < 1e-5
    assert (c.cx, c.cy) == (4.0, 2.0)


def test_quads_are_split_and_normals_kept():
    s = rt.Scene()
    s.add_object(os.path.join(rt.ASSETS, "objects", "cube.obj"), [1, 2, 3], [2, 2, 2], [0, 0, 0], [1, 0, 0], rt.DIFFUSE)
    a = s.arrays()
    assert a["tri_v"].shape == (12, 3)
    np.testing.assert_array_equal(a["box_min"][0], [-1, 0, 1])
    np.testing.assert_array_equal(a["box_max"][0], [3, 4, 5])
    # every face normal is one of the six axis directions
    assert set(map(tuple, np.abs(a["normals"]).round(6).tolist())) <= {(1, 0, 0), (0, 1, 0), (0, 0, 1)}


def test_faces_without_normals_get_flat_normals(tmp_path):
    p = tmp_path / "tri.obj"
    p.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\nf 1 2 3\nf -4 -3 -1\n")
    s = rt.Scene()
    s.add_object(str(p), [0, 0, 0], [1, 1, 1], [0, 0, 0], [1, 1, 1], rt.DIFFUSE)
    a = s.arrays()
    assert a["tri_v"].shape == (2, 3)
    np.testing.assert_array_equal(a["normals"][:3], [[0, 0, 1]] * 3)
    np.testing.assert_array_equal(a["normals"][3:], [[0, -1, 0]] * 3)  # (1,0,0) x (0,0,1)


def test_groups_become_boxes_with_global_indices(tmp_path):
    p = tmp_path / "two.obj"
    p.write_text("g a\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\ng b\nv 5 5 5\nv 6 5 5\nv 5 6 5\nf 4 5 6\n")
    s = rt.Scene()
    s.add_object(str(p), [0, 0, 0], [1, 1, 1], [0, 0, 0], [1, 1, 1], rt.MIRROR)
    s.add_object(str(p), [10, 0, 0], [1, 1, 1], [0, 0, 0], [1, 1, 1], rt.GLASS, 1.0, 1.5)
    d, _ = s.describe()
    a = rt.desc_arrays(d)
    assert (d.n_boxes, d.n_triangles, d.n_vertices) == (4, 4, 12)
    np.testing.assert_array_equal(a["tri_v"].ravel(), np.arange(12))  # no aliasing between shapes or objects
    np.testing.assert_array_equal(a["box_min"], [[0, 0, 0], [5, 5, 5], [10, 0, 0], [15, 5, 5]])
    assert list(a["material_type"][a["tri_material"]]) == [rt.MIRROR, rt.MIRROR, rt.GLASS, rt.GLASS]


def test_camera_mutators_compose_like_the_reference():
    s = rt.Scene()
    s.set_camera_intrinsics(90.0, 640, 480)
    s.rotate_camera(0, 90, 0)
    s.transform_camera(1, 2, 3)
    s.transform_camera(1, 0, 0)
    _, c = s.describe()
    np.testing.assert_allclose(np.array(c.rot).reshape(3, 3), rotation(0, 90, 0), atol=1e-7)
    assert list(c.pos) == [2, 2, 3]
    assert (c.cx, c.cy, c.width, c.height) == (320, 240, 640, 480)
    assert abs(c.fx - 320.0) < 1e-3 and c.fx == c.fy  # f_y uses the WIDTH too


def test_bad_arguments_are_rejected():
    s = rt.Scene()
    with pytest.raises(rt.RtError):
        s.set_camera_intrinsics(60, 0, 10)
    with pytest.raises(rt.RtError):
        s.add_sphere([0, 0, 0], 1, [1, 1, 1], 7)
    with pytest.raises(rt.RtError):
        s.set_num_rays(0)


def test_png_writer_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(37, 53, 3), dtype=np.uint8)
    p = tmp_path / "x.png"
    rt.write_png(str(p), img)
    back = np.asarray(Image.open(p).convert("RGB"))
    np.testing.assert_array_equal(back, img)
