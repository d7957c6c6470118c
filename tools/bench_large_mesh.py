#!/usr/bin/env python3
"""Times 512x512x32spp indirect+AA+bbox frames of procedural meshes of growing size."""
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
for n_lat, n_lon in [(32, 62), (64, 128), (100, 200), (200, 400), (400, 800)]:
    v, n, t = make_mesh.bumpy_sphere(n_lat, n_lon)
    path = os.path.join(tmp, f"m{n_lat}.obj")
    make_mesh.write_obj(path, v, n, t)
    s = rt.Scene()
    s.load_scene(os.path.join(rt.ASSETS, "scenes", "simple.json"))
    s.set_camera_intrinsics(60.0, 512, 512)
    s.set_num_rays(32)
    s.add_object(path, [0, 2, 8], [1, 1, 1], [10, 25, 0], [0.9, 0.6, 0.3], rt.DIFFUSE)
    desc, cam = s.describe()
    dev = s.upload()
    row = [f"{len(t):7d} triangles"]
    for culling in (True, False):
        if not culling and len(t) > 50000:
            continue
        opt = rt.make_options(seed=1, cluster_culling=culling)
        dev.render(cam, opt)
        ms = []
        for _ in range(3):
            _, st = dev.render(cam, opt)
            ms.append(st.gpu_ms)
        row.append(f"{'culled' if culling else 'exhaustive'} {np.median(ms):8.2f} ms {st.paths / np.median(ms) / 1e3:7.0f} Mpaths/s "
                   f"{st.triangle_tests / st.paths:8.0f} tests/path")
    print(" | ".join(row), flush=True)
