#!/usr/bin/env python3
"""rt_options.arithmetic = 1 against arithmetic = 0 on random scenes beyond the test-suite's: per pixel in the modes
without random numbers (direct light, no anti-aliasing; with and without boxes), where the task asks for 1e-4 relative.
Prints, per batch of scenes, the share of pixels outside that tolerance (decision flips at edges) and the median
relative error of the others.    python tools/fast_arithmetic_sweep.py [first] [count]"""
import os
import pathlib
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import importlib.util  # noqa: E402

import raytracer_amd as rt  # noqa: E402

spec = importlib.util.spec_from_file_location("scenes", os.path.join(ROOT, "tests", "test_gpu_random_scenes.py"))
scenes = importlib.util.module_from_spec(spec)
spec.loader.exec_module(scenes)

first = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rt.init(0)
tmp = pathlib.Path(tempfile.mkdtemp())
outside = pixels = 0
worst = (0.0, -1)
medians = []
for seed in range(first, first + count):
    s = scenes.random_scene(seed, tmp)
    desc, cam = s.describe()
    dev = s.upload()
    for bbox in (True, False):
        opts = dict(indirect=False, antialiasing=False, bounding_box=bbox, seed=seed)
        exact, _ = dev.render(cam, rt.make_options(arithmetic=0, **opts))
        fast, _ = dev.render(cam, rt.make_options(arithmetic=1, **opts))
        both_odd = ~np.isfinite(fast) & ~np.isfinite(exact)
        err = np.abs(fast.astype(np.float64) - exact) / np.maximum(np.abs(exact.astype(np.float64)), 1e-2)
        err = np.where(both_odd, 0.0, np.nan_to_num(err, nan=np.inf)).max(axis=2)
        out = err > 1e-4
        outside += int(out.sum())
        pixels += out.size
        if out.mean() > worst[0]:
            worst = (float(out.mean()), seed)
        if (~out).any():
            medians.append(float(np.median(err[~out])))
    if (seed - first) % 50 == 49:
        print(f"{seed - first + 1} scenes: {outside} of {pixels} pixels outside 1e-4 ({outside / pixels:.5%}), worst scene {worst[1]} "
              f"({worst[0]:.4%}), median relative error of the others {np.median(medians):.2e}", flush=True)
print(f"done: {count} scenes x 2, {outside / max(pixels, 1):.5%} of the pixels outside 1e-4 relative, worst scene {worst[0]:.4%}")
