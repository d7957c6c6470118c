"""Frame sequences (the animation loop of the reference's README.md:24), float
image output and progress reports."""
import os
import subprocess

import numpy as np
import pytest
from PIL import Image

import raytracer_amd as rt
from helpers import SCENES, load, same_bits
from raytracer_amd.distributed import frame_share


def test_pfm_round_trip_and_errors(tmp_path):
    a = (np.random.default_rng(0).random((5, 7, 3)) * 300 - 10).astype(np.float32)
    a[0, 0, 0] = np.inf
    p = tmp_path / "a.pfm"
    rt.write_pfm(str(p), a)
    head = p.read_bytes()[:20].split(b"\n")
    assert head[0] == b"PF" and head[1] == b"7 5" and float(head[2]) < 0
    assert np.array_equal(rt.read_pfm(str(p)), a)
    with pytest.raises(rt.RtError) as e:
        rt.write_pfm(str(tmp_path / "no" / "dir.pfm"), a)
    assert e.value.code == rt.RT_ERR_IO


def test_frame_share_covers_every_frame_once():
    for world in (1, 2, 3, 8):
        seen = []
        for rank in range(world):
            first, stride = frame_share(rank, world)
            seen += list(range(first, 11, stride))
        assert sorted(seen) == list(range(11))
    with pytest.raises(ValueError):
        frame_share(2, 2)


def test_bad_frame_patterns_are_rejected_before_any_device_work():
    s = load("simple", 16, 16, 1)
    for pattern in ("frame.png", "f_%d_%d.png", "f_%s.png", "f_%5.2f.png"):
        with pytest.raises(rt.RtError, match="pattern"):
            s.render_frames(pattern, 0, 2)
    with pytest.raises(rt.RtError):
        s.render_frames("f_%d.png", 0, 2, stride=0)


STEPS = dict(rotate_step=(0.0, 4.0, 0.0), move_step=(1.5, 0.0, -1.0))


def _single_frames(n, seed0, name="mesh_scene"):
    """Frame i the long way: i camera steps, then Scene.render with seed seed0 + i."""
    s = load(name, 64, 48, 3)
    s.options = rt.make_options(seed=0)
    out = []
    for i in range(n):
        if i:
            s.rotate_camera(*STEPS["rotate_step"])
            s.transform_camera(*STEPS["move_step"])
        out.append(s.render(None, seed0 + i))
    return out


@pytest.mark.gpu
def test_frame_sequence_equals_single_renders(tmp_path):
    want = _single_frames(5, 100)
    s = load("mesh_scene", 64, 48, 3)
    seconds = s.render_frames(str(tmp_path / "f_%03d.pfm"), 100, 5, **STEPS, writers=3)
    assert seconds > 0
    for i in range(5):
        assert same_bits(rt.read_pfm(str(tmp_path / f"f_{i:03d}.pfm")), want[i]), f"frame {i}"
    assert not same_bits(want[0], want[1])
    # PNG output: fminf(v, 255) and truncation, as Scene::render writes it
    s = load("mesh_scene", 64, 48, 3)
    s.render_frames(str(tmp_path / "g_%d.png"), 100, 5, **STEPS)
    for i in range(5):
        png = np.asarray(Image.open(tmp_path / f"g_{i}.png").convert("RGB"))
        assert np.array_equal(png, np.minimum(want[i], 255).astype(np.uint8))


@pytest.mark.gpu
def test_frame_shares_of_two_ranks_make_the_whole_sequence(tmp_path):
    want = _single_frames(5, 7)
    for rank in range(2):
        first, stride = frame_share(rank, 2)
        s = load("mesh_scene", 64, 48, 3)
        s.render_frames(str(tmp_path / f"r{rank}_%d.pfm"), 7, 5, **STEPS, first=first, stride=stride)
    assert sorted(os.listdir(tmp_path)) == ["r0_0.pfm", "r0_2.pfm", "r0_4.pfm", "r1_1.pfm", "r1_3.pfm"]
    for i in range(5):
        assert same_bits(rt.read_pfm(str(tmp_path / f"r{i % 2}_{i}.pfm")), want[i])


@pytest.mark.gpu
def test_unwritable_frame_file_is_reported(tmp_path):
    s = load("simple", 32, 32, 1)
    with pytest.raises(rt.RtError, match="cannot open"):
        s.render_frames(str(tmp_path / "missing" / "f_%d.png"), 0, 3)
    assert s.render_frames(None, 0, 3) > 0  # no pattern: frames are rendered, nothing is written


@pytest.mark.gpu
def test_progress_reports_every_batch():
    s = load("default", 1024, 1024, 20)  # 21 M paths: two batches of at most 2^24
    desc, cam = s.describe()
    dev = s.upload()
    seen = []
    dev.set_progress(lambda done, total: seen.append((done, total)))
    img, st = dev.render(cam, rt.make_options(seed=2))
    assert st.batches == len(seen) == 2
    assert seen[-1] == (1024 * 1024 * 20, 1024 * 1024 * 20) and seen[0][0] < seen[1][0]
    dev.set_progress(None)
    img2, _ = dev.render(cam, rt.make_options(seed=2))
    assert len(seen) == 2 and same_bits(img, img2)


@pytest.mark.gpu
def test_command_line_frame_loop(tmp_path):
    exe = os.path.join(rt.REPO_ROOT, "raytracer_amd", "render")
    r = subprocess.run([exe, os.path.join(SCENES, "simple.json"), "--frames", "3", "--rotate", "0", "5", "0", "--seed", "9",
                        "-o", str(tmp_path / "c_%02d.png")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=120, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout
    assert "Rendered 3 frames" in r.stdout
    s = load("simple")
    s.options = rt.make_options(seed=0)
    for i in range(3):
        if i:
            s.rotate_camera(0, 5, 0)
        img = s.render(None, 9 + i)
        png = np.asarray(Image.open(tmp_path / f"c_{i:02d}.png").convert("RGB"))
        assert np.array_equal(png, np.minimum(img, 255).astype(np.uint8))
