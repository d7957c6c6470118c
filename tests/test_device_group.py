"""Single-process multi-GPU driver (rt_device_group) and the reference-side boundary code.

CPU: the frame partition against the Python driver's, the adapter's build against the reference's
headers.  GPU (one device on the test box): the group code with one member, and with several
members on the same device, which exercises partition, submit threads, the peer exchange and the
order of the additions; the C++ raytrace_spheres() and the adapter through their check programs.
"""
import os
import subprocess

import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd.distributed import frame_partition as py_frame_partition

REFERENCE = "/root/reference"


@pytest.mark.parametrize("world", [1, 2, 3, 4, 6, 8, 16])
def test_frame_partition_matches_the_python_driver(world):
    for spp in (1, 2, 3, 5, 7, 32, 60, 1000):
        for height in (1, 7, 64, 720):
            covered = np.zeros((spp, height), np.int32)
            for rank in range(world):
                got = rt.frame_partition(rank, world, spp, height)
                assert got == tuple(py_frame_partition(rank, world, spp, height))
                covered[got[0]:got[1], got[2]:got[3]] += 1
            assert (covered == 1).all()  # every (sample, row) rendered exactly once
    assert rt.frame_partition(5, 4, 8, 10) == (0, 0, 0, 0)  # rank outside the group: nothing


def test_adapter_builds_against_the_references_headers():
    """integration/raytrace_amd_adapter.cpp is the translation unit a maintainer of the reference adds in
    place of src/raytrace.cu; it must compile with the reference's own struct definitions."""
    if not os.path.isdir(os.path.join(REFERENCE, "include")):
        pytest.skip("the reference tree is not present on this machine")
    r = subprocess.run(["make", "-C", os.path.join(rt.REPO_ROOT, "integration")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    assert os.path.exists(os.path.join(rt.REPO_ROOT, "oracle", "_ref", "adapter_check"))


@pytest.mark.gpu
def test_group_of_one_device_is_rt_render():
    from helpers import load

    s = load("mesh_scene", 160, 120, 5)
    desc, cam = s.describe()
    opt = rt.make_options(seed=71)
    group = rt.DeviceGroup(desc, [0])
    img, st = group.render(cam, opt)
    want, _ = s.upload().render(cam, opt)
    assert np.array_equal(img.view(np.uint32), want.view(np.uint32))
    assert st.paths == 160 * 120 * 5
    again, _ = group.render(cam, rt.make_options(seed=72))  # the group is reusable
    want2, _ = s.upload().render(cam, rt.make_options(seed=72))
    assert np.array_equal(again.view(np.uint32), want2.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("members,spp", [(3, 7), (2, 32), (4, 2), (6, 4)])
def test_group_members_on_one_device_compose_the_frame(members, spp):
    """Several members on device 0 (the peer exchange; RCCL refuses one device twice): each renders its
    rt_frame_partition share, the first adds the others' sums in member order.  The same composition by
    hand -- rt_accumulate_rows per share, rt_add_sums in order, rt_resolve -- must give the same bits."""
    from helpers import load
    from hip_mem import DeviceBuffer

    s = load("mesh_scene", 96, 70, spp)
    desc, cam = s.describe()
    opt = rt.make_options(seed=81)
    img, _ = rt.DeviceGroup(desc, [0] * members).render(cam, opt)
    dev = s.upload()
    n = cam.width * cam.height
    total, out = DeviceBuffer(4 * n), DeviceBuffer(3 * n)
    for rank in range(members):
        sb, se, rb, re = rt.frame_partition(rank, members, spp, cam.height)
        part = total if rank == 0 else DeviceBuffer(4 * n)
        dev.accumulate(cam, opt, sb, se, part.address, rows=(rb, re))
        if rank:
            rt.add_sums(total.address, part.address, 4 * n)
    rt.resolve(total.address, n, out.address)
    want = out.to_host().reshape(cam.height, cam.width, 3)
    assert np.array_equal(img.view(np.uint32), want.view(np.uint32))
    single, _ = dev.render(cam, opt)
    np.testing.assert_allclose(img, single, rtol=4e-6, atol=1e-4)  # fp32 association across members


@pytest.mark.gpu
@pytest.mark.parametrize("members", [2, 3])
def test_pixels_that_end_after_one_sample_compose_across_members(members):
    """Direct light without anti-aliasing: a diffuse first hit ends the pixel's sample loop after one sample
    (reference src/raytrace.cu:340-346, 496-498), which the sums carry as a NEGATIVE sample count.  Each member ends
    the loop on its own first sample, so such a pixel arrives as (members x L, -members): the same colour as the
    single-device render's (L, -1), up to the rounding of (L + ... + L) / members."""
    from helpers import load

    s = load("mesh_scene", 96, 64, 6)
    desc, cam = s.describe()
    opt = rt.make_options(indirect=False, antialiasing=False, seed=7)
    img, _ = rt.DeviceGroup(desc, [0] * members).render(cam, opt)
    single, _ = s.upload().render(cam, opt)
    assert np.all(img >= 0) and np.all(np.isfinite(img))  # (a resolve that divided by the signed count would negate colours)
    np.testing.assert_allclose(img, single, rtol=4e-6, atol=1e-4)
    assert (img == single).mean() > 0.9


_RCCL_REHEARSAL = """
import numpy as np
import raytracer_amd as rt
import sys
sys.path.insert(0, {tests!r})
from helpers import load
s = load("mesh_scene", 128, 96, 4)
desc, cam = s.describe()
opt = rt.make_options(seed=91)
img, _ = rt.DeviceGroup(desc, [0]).render(cam, opt)
want, _ = s.upload().render(cam, opt)
assert np.array_equal(img.view(np.uint32), want.view(np.uint32))
print("rccl rehearsal ok")
"""


@pytest.mark.gpu
def test_rccl_exchange_with_one_rank():
    """RT_AMD_EXCHANGE=rccl makes a group of one device run its exchange through RCCL as well: librccl is
    loaded at run time, ncclCommInitAll builds a one-rank communicator and the frame's sums pass through
    ncclReduce.  (More ranks need more GPUs than the test box has.)  In a process of its own, on the system's
    HIP runtime, which is the one the system's RCCL was built for."""
    import sys

    env = dict(os.environ, RT_AMD_EXCHANGE="rccl", RT_AMD_SYSTEM_HIP="1", PYTHONPATH=rt.REPO_ROOT)
    code = _RCCL_REHEARSAL.format(tests=os.path.join(rt.REPO_ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=110)
    assert r.returncode == 0 and "rccl rehearsal ok" in r.stdout, r.stdout


@pytest.mark.gpu
def test_scene_devices_setting_and_a_missing_device():
    from helpers import load

    s = load("default", 64, 48, 3)
    s.options = rt.make_options(seed=5)
    one = s.render(None, 5)
    s.set_devices(1)
    assert np.array_equal(s.render(None, 5).view(np.uint32), one.view(np.uint32))
    import hip_mem

    count = hip_mem.device_count()
    s.set_devices(count + 1)  # one more than the machine has
    with pytest.raises(rt.RtError) as e:
        s.render(None, 5)
    assert e.value.code == rt.RT_ERR_INVALID and "device" in str(e.value)
    s.set_devices(1)
    assert np.array_equal(s.render(None, 5).view(np.uint32), one.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("scene", ["mesh_scene", "default"])
def test_cpp_raytrace_spheres_equals_scene_render(scene):
    """raytrace_spheres() of the host layer (the reference launcher's signature, arrays of structs) against Scene::render."""
    exe = os.path.join(rt.REPO_ROOT, "raytracer_amd", "check_raytrace_spheres")
    if not os.path.exists(exe):
        pytest.skip("check_raytrace_spheres not built")
    r = subprocess.run([exe, os.path.join(rt.ASSETS, "scenes", scene + ".json"), "96", "64", "3"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=120, cwd=os.path.join(rt.ASSETS, "scenes"))
    assert r.returncode == 0, r.stdout
    assert " 0 of " in r.stdout


@pytest.mark.gpu
def test_reference_side_adapter_renders_what_rt_render_renders():
    """oracle/_ref/adapter_check is integration/raytrace_amd_adapter.cpp + a check program built against the
    reference's headers (16-byte vectors, Material copies per primitive): it lays a scene out in the
    reference's types, calls the adapter's raytrace_spheres() and compares with rt_render."""
    exe = os.path.join(rt.REPO_ROOT, "oracle", "_ref", "adapter_check")
    if not os.path.exists(exe):
        pytest.skip("adapter_check was not built (needs the reference's headers at build time)")
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0, r.stdout
    assert "0 differ from rt_render" in r.stdout and "sizeof(CU_Vector3f)=16" in r.stdout
