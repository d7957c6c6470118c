#!/usr/bin/env python3
"""Two against three culling levels on procedural meshes: frame time, tests per path, and equality of the images.
usage: three_levels_probe.py [--levels=2|3] [n_lat:n_lon ...]   (512 x 512 x 32 spp, indirect + AA + bbox, exact arithmetic)"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_mesh  # noqa: E402
import raytracer_amd as rt  # noqa: E402

rt.init(0)
tmp = tempfile.mkdtemp()
only = [a[len("--levels="):] for a in sys.argv[1:] if a.startswith("--levels=")]
sizes = [tuple(int(x) for x in a.split(":")) for a in sys.argv[1:] if not a.startswith("--")] or [(64, 128), (100, 200), (200, 400), (400, 800)]
for n_lat, n_lon in sizes:
    v, n, t = make_mesh.bumpy_sphere(n_lat, n_lon)
    path = os.path.join(tmp, f"m{n_lat}.obj")
    make_mesh.write_obj(path, v, n, t)
    s = rt.Scene()
    s.load_scene(os.path.join(rt.ASSETS, "scenes", "simple.json"))
    s.set_camera_intrinsics(60.0, 512, 512)
    s.set_num_rays(32)
    s.add_object(path, [0, 2, 8], [1, 1, 1], [10, 25, 0], [0.9, 0.6, 0.3], rt.DIFFUSE)
    desc, cam = s.describe()
    row = [f"{len(t):7d} triangles"]
    images = []
    for levels in only or ("2", "3"):
        os.environ["RT_AMD_LEVELS"] = levels
        dev = s.upload()
        opt = rt.make_options(seed=1)
        img, _ = dev.render(cam, opt)
        images.append(img)
        ms = []
        for _ in range(3):
            _, st = dev.render(cam, opt)
            ms.append(st.gpu_ms)
        row.append(f"{levels} levels {np.median(ms):8.2f} ms {st.triangle_tests / st.paths:8.0f} tests/path")
        del dev
    if len(images) == 2:
        differ = int((images[0].view(np.uint32) != images[1].view(np.uint32)).any(axis=2).sum())
        row.append(f"{differ} pixels differ")
    print(" | ".join(row), flush=True)
