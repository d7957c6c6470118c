"""Committed fixtures (tests/golden/frames.npz, probes.npz; written by
tests/golden/make_fixtures.py with the pinned oracle) against the oracle as built
now (CPU) and against the HIP path through the C ABI (GPU).  Bit-exact.
"""
import os
import sys

import numpy as np
import pytest

import oracle_binding as ob
import raytracer_amd as rt
from helpers import MODES, load, same_bits

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
sys.path.insert(0, GOLDEN)
import make_fixtures as mf  # noqa: E402

FRAMES = np.load(os.path.join(GOLDEN, "frames.npz"))
PROBES = np.load(os.path.join(GOLDEN, "probes.npz"))


def test_fixture_files_cover_every_scene_and_mode():
    assert sorted(FRAMES.files) == sorted(f"{s}/{m}" for s in mf.FRAME_SCENES for m in MODES)
    for name in mf.PROBE_SCENES:
        o, d, tg = mf.probe_rays(name, mf.N_PROBES)
        assert np.array_equal(o, PROBES[f"{name}/origins"]) and np.array_equal(d, PROBES[f"{name}/dirs"])
        assert np.array_equal(tg, PROBES[f"{name}/targets"])
    assert (PROBES["mesh_scene/bbox1/triangle"] >= 0).sum() > 50


@pytest.mark.parametrize("name", list(mf.FRAME_SCENES))
@pytest.mark.parametrize("mode", ["simple_bbox", "indirect_aa_bbox"])
def test_oracle_reproduces_the_frames(name, mode):
    s = mf.FRAME_SCENES[name][0]()  # the description points into the scene's arrays: keep it alive
    desc, cam = s.describe()
    img, _ = ob.render_image(desc, cam, ob.options(*MODES[mode], seed=mf.FRAME_SEED))
    assert same_bits(img, FRAMES[f"{name}/{mode}"])


@pytest.mark.parametrize("name", mf.PROBE_SCENES)
def test_oracle_reproduces_the_probes(name):
    s = load(name)
    desc, _ = s.describe()
    o, d, tg = (PROBES[f"{name}/{k}"] for k in ("origins", "dirs", "targets"))
    for bbox in (0, 1):
        opt = ob.options(bounding_box=bool(bbox))
        t, sid, tid, abg = ob.cast(desc, opt, o, d)
        assert same_bits(t, PROBES[f"{name}/bbox{bbox}/t"])
        assert np.array_equal(sid, PROBES[f"{name}/bbox{bbox}/sphere"])
        assert np.array_equal(tid, PROBES[f"{name}/bbox{bbox}/triangle"])
        assert same_bits(abg, PROBES[f"{name}/bbox{bbox}/abg"])
        assert np.array_equal(ob.visible(desc, opt, o, tg), PROBES[f"{name}/bbox{bbox}/visible"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(mf.FRAME_SCENES))
@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("culling", [True, False])
def test_hip_reproduces_the_frames(name, mode, culling):
    s = mf.FRAME_SCENES[name][0]()
    desc, cam = s.describe()
    img, _ = s.upload().render(cam, rt.make_options(*MODES[mode], seed=mf.FRAME_SEED, cluster_culling=culling))
    want = FRAMES[f"{name}/{mode}"]
    assert same_bits(img, want), f"{(img != want).any(axis=2).sum()} pixels differ"


@pytest.mark.gpu
@pytest.mark.parametrize("name", mf.PROBE_SCENES)
def test_hip_reproduces_the_probes(name):
    dev = load(name).upload()
    o, d, tg = (PROBES[f"{name}/{k}"] for k in ("origins", "dirs", "targets"))
    for bbox in (0, 1):
        opt = rt.make_options(bounding_box=bool(bbox))
        t, sid, tid, abg = dev.cast(opt, o, d)
        want_tid = PROBES[f"{name}/bbox{bbox}/triangle"]
        assert same_bits(t, PROBES[f"{name}/bbox{bbox}/t"])
        assert np.array_equal(sid, PROBES[f"{name}/bbox{bbox}/sphere"])
        assert np.array_equal(tid, want_tid)
        assert same_bits(abg[want_tid >= 0], PROBES[f"{name}/bbox{bbox}/abg"][want_tid >= 0])
        assert np.array_equal(dev.visible(opt, o, tg), PROBES[f"{name}/bbox{bbox}/visible"])
