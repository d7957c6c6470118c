#!/usr/bin/env python3
"""Extra parity sweep on the GPU: random scenes beyond the seeds of the test-suite, every mode,
culled and exhaustive, against the CPU oracle, bit for bit.  python tools/parity_sweep.py [first] [count]"""
import os
import pathlib
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import importlib.util  # noqa: E402

import oracle_binding as ob  # noqa: E402
import raytracer_amd as rt  # noqa: E402
from helpers import MODES, same_bits  # noqa: E402

spec = importlib.util.spec_from_file_location("scenes", os.path.join(ROOT, "tests", "test_gpu_random_scenes.py"))
scenes = importlib.util.module_from_spec(spec)
spec.loader.exec_module(scenes)

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rt.init(0)
tmp = pathlib.Path(tempfile.mkdtemp())
bad = 0
for seed in range(first, first + count):
    s = scenes.random_scene(seed, tmp)
    desc, cam = s.describe()
    dev = s.upload()
    for mode, flags in MODES.items():
        ref, _ = ob.render_image(desc, cam, ob.options(*flags, seed=seed))
        for culling in (True, False):
            img, _ = dev.render(cam, rt.make_options(*flags, seed=seed, cluster_culling=culling))
            if not same_bits(img, ref):
                bad += 1
                print(f"MISMATCH seed {seed} mode {mode} culling {culling}: {(img != ref).any(axis=2).sum()} pixels", flush=True)
    if (seed - first) % 20 == 19:
        print(f"{seed - first + 1} scenes, {bad} mismatches", flush=True)
print(f"done: {count} scenes x {len(MODES)} modes x 2, {bad} mismatches")
sys.exit(1 if bad else 0)
