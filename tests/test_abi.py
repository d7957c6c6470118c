"""The C-ABI library loads without a GPU, exports every function include/rt_amd.h
declares, and rejects bad arguments before touching the device.  CPU only.
"""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import raytracer_amd as rt

HEADER = os.path.join(rt.REPO_ROOT, "include", "rt_amd.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    names = declared_functions()
    assert len(names) >= 24
    lib = C.CDLL(rt.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in rt_amd.h but not exported"
    assert sorted(rt.EXPORTS) == names


def test_library_contains_gfx950_code_object():
    out = subprocess.run(["strings", "-n", "6", rt.LIB_PATH], stdout=subprocess.PIPE, text=True).stdout
    assert "gfx950" in out
    assert "k_mesh" in out and "k_shade" in out


def test_the_product_does_not_link_the_oracle():
    out = subprocess.run(["ldd", rt.LIB_PATH], stdout=subprocess.PIPE, text=True).stdout
    assert "oracle" not in out
    syms = subprocess.run(["nm", "-D", rt.LIB_PATH], stdout=subprocess.PIPE, text=True).stdout
    assert "orc_" not in syms


def test_struct_sizes_match_the_header():
    # computed from the field lists in rt_amd.h with natural alignment on LP64
    assert C.sizeof(rt.Camera) == 9 * 4 + 3 * 4 + 4 * 4 + 3 * 4
    assert C.sizeof(rt.Options) == 8 * 4
    assert C.sizeof(rt.Stats) == 3 * 8 + 2 * 4 + 4 * 8 + 8 + 2 * 4 + 2 * 8
    assert C.sizeof(rt.SceneDesc) == 6 * 8 + 16 * 8  # six counts (padded to 8) and sixteen pointers


def test_upload_rejects_inconsistent_descriptions():
    h = C.c_void_p()
    assert rt.lib.rt_scene_upload(None, C.byref(h)) == -1
    d = rt.SceneDesc()
    d.n_spheres = 3  # count without arrays
    assert rt.lib.rt_scene_upload(C.byref(d), C.byref(h)) == -1
    assert b"null array" in rt.lib.rt_last_error()

    arrays = {
        "sphere_pos": np.zeros((1, 3), np.float32), "sphere_radius": np.ones(1, np.float32),
        "sphere_material": np.array([4], np.int32),  # out of range
        "material_type": np.array([0], np.int32), "material_color": np.ones((1, 3), np.float32),
        "material_ro": np.zeros(1, np.float32), "material_ri": np.zeros(1, np.float32),
    }
    d, keep = rt.desc_from_arrays(arrays)
    assert rt.lib.rt_scene_upload(C.byref(d), C.byref(h)) == -1
    assert b"material" in rt.lib.rt_last_error()

    arrays["sphere_material"] = np.array([0], np.int32)
    arrays.update(vertices=np.zeros((3, 3), np.float32), normals=np.zeros((3, 3), np.float32),
                  tri_v=np.array([[0, 1, 3]], np.int32), tri_n=np.array([[0, 1, 2]], np.int32),
                  tri_material=np.array([0], np.int32))
    d, keep = rt.desc_from_arrays(arrays)
    assert rt.lib.rt_scene_upload(C.byref(d), C.byref(h)) == -1
    assert b"outside" in rt.lib.rt_last_error()

    arrays["tri_v"] = np.array([[0, 1, 2]], np.int32)
    arrays["material_type"] = np.array([5], np.int32)
    d, keep = rt.desc_from_arrays(arrays)
    assert rt.lib.rt_scene_upload(C.byref(d), C.byref(h)) == -1
    assert b"unknown type" in rt.lib.rt_last_error()


def test_render_entry_points_reject_null_arguments():
    cam, opt = rt.Camera(), rt.make_options()
    assert rt.lib.rt_render(None, C.byref(cam), C.byref(opt), None, None) == -1
    assert rt.lib.rt_accumulate(None, C.byref(cam), C.byref(opt), 0, 1, None, None, None) == -1
    assert rt.lib.rt_resolve(None, 16, None, None) == -1
    assert rt.lib.rt_cast(None, C.byref(opt), 1, None, None, None, None, None, None) == -1
    assert rt.lib.rt_visible(None, C.byref(opt), 1, None, None, None) == -1
    assert rt.lib.rt_scene_free(None) == 0


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    """There is no fallback: without the shared library the package cannot be imported."""
    import importlib.util

    src = os.path.join(rt.REPO_ROOT, "raytracer_amd", "__init__.py")
    pkg = tmp_path / "raytracer_amd_copy"
    pkg.mkdir()
    (pkg / "__init__.py").write_text(open(src).read())
    spec = importlib.util.spec_from_file_location("raytracer_amd_copy", pkg / "__init__.py")
    mod = importlib.util.module_from_spec(spec)
    with pytest.raises(ImportError, match="no fallback"):
        spec.loader.exec_module(mod)
