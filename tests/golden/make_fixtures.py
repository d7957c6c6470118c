#!/usr/bin/env python3
"""Writes tests/golden/frames.npz and tests/golden/probes.npz with the CPU oracle.

    make -C oracle && python tests/golden/make_fixtures.py

The oracle is first pinned to the values recorded from the reference routine
(tests/test_oracle_golden.py); these fixtures then freeze its output for every
mode so that later builds of the oracle (other compiler, other flags) and the
HIP path are both compared against data that no longer moves.  The generator
is keyed on (seed, pixel, sample, dimension), so stochastic frames are exactly
reproducible too.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle_binding as ob  # noqa: E402
from helpers import MODES, load, variant  # noqa: E402

# name -> (scene factory, width, height, samples per pixel)
FRAME_SCENES = {
    "simple": (lambda: load("simple", 96, 96, 2), 96, 96, 2),
    "default": (lambda: load("default", 96, 96, 2), 96, 96, 2),
    "mesh_scene": (lambda: load("mesh_scene", 96, 96, 2), 96, 96, 2),
    "mesh_mirror": (lambda: variant("mesh_scene", "mirror", 80, 80, 2), 80, 80, 2),
    "mesh_glass": (lambda: variant("mesh_scene", "glass", 80, 80, 2), 80, 80, 2),
}
FRAME_SEED = 21
PROBE_SCENES = ("simple", "default", "mesh_scene")
N_PROBES = 768


def probe_rays(name, n):
    """Origins inside the room, directions uniform on the sphere; half of them aimed at the mesh."""
    rng = np.random.default_rng(PROBE_SCENES.index(name) + 1)
    o = rng.uniform(-40, 40, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3))
    aim = rng.uniform(-12, 12, (n, 3)) - o
    d[: n // 2] = aim[: n // 2]
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    targets = rng.uniform(-45, 45, (n, 3)).astype(np.float32)
    return o, d, targets


def main():
    frames = {}
    for name, (make, w, h, spp) in FRAME_SCENES.items():
        s = make()
        desc, cam = s.describe()
        for mode, flags in MODES.items():
            img, _ = ob.render_image(desc, cam, ob.options(*flags, seed=FRAME_SEED))
            frames[f"{name}/{mode}"] = img
            print(name, mode, img.shape, float(img.mean()))
    np.savez_compressed(os.path.join(HERE, "frames.npz"), **frames)

    probes = {}
    for name in PROBE_SCENES:
        s = load(name)
        desc, _ = s.describe()
        o, d, targets = probe_rays(name, N_PROBES)
        probes[f"{name}/origins"], probes[f"{name}/dirs"], probes[f"{name}/targets"] = o, d, targets
        for bbox in (0, 1):
            opt = ob.options(bounding_box=bool(bbox))
            t, sid, tid, abg = ob.cast(desc, opt, o, d)
            vis = ob.visible(desc, opt, o, targets)
            probes[f"{name}/bbox{bbox}/t"], probes[f"{name}/bbox{bbox}/sphere"] = t, sid
            probes[f"{name}/bbox{bbox}/triangle"], probes[f"{name}/bbox{bbox}/abg"] = tid, abg
            probes[f"{name}/bbox{bbox}/visible"] = vis
            print(name, "bbox", bbox, "sphere hits", int((sid >= 0).sum()), "triangle hits", int((tid >= 0).sum()),
                  "visible", int(vis.sum()))
    np.savez_compressed(os.path.join(HERE, "probes.npz"), **probes)


if __name__ == "__main__":
    main()
