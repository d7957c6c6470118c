#!/usr/bin/env python3
"""Times every published row of the reference's results table and the five
benchmark configurations on one GPU and prints a markdown table.

    python tools/bench_configs.py [--frames 5] [--out gpurun_out/configs.json]

Per row: device time of one frame (HIP events around generate -> resolve, the
reference's own timed region, src/raytrace.cu:572-633) and end-to-end time of
rt_render (adds the image download), median of `--frames` frames after one
warm-up, Mpaths/s = W*H*spp / device time.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=5)
    ap.add_argument("--out", default=None)
    ap.add_argument("--c5-spp", type=int, default=1000)
    ap.add_argument("--arithmetic", type=int, default=0, help="rt_options.arithmetic: 0 exact, 1 fast")
    ap.add_argument("--no-exhaustive", action="store_true", help="skip the rows with cluster culling off")
    ap.add_argument("--only", default="", help="only rows whose name contains this")
    args = ap.parse_args()

    import raytracer_amd as rt
    from helpers import load, variant
    from test_gpu_configs import two_light_scene

    rt.init(0)

    def big(s, w, h, spp):
        s.set_camera_intrinsics(60.0, w, h)
        s.set_num_rays(spp)
        return s

    # name, reference seconds on its GTX 1050 (README.md:74-78) or None, scene factory, (indirect, aa, bbox)
    rows = [
        ("simple, default", 0.05, lambda: load("default"), (False, False, False)),
        ("simple, cat", 0.37, lambda: load("mesh_scene"), (False, False, False)),
        ("simple+bbox, cat", 0.12, lambda: load("mesh_scene"), (False, False, True)),
        ("C1 AA+bbox, default", 0.12, lambda: load("default"), (False, True, True)),
        ("AA+bbox, cat", 4.14, lambda: load("mesh_scene"), (False, True, True)),
        ("C2 indirect+AA, default", 0.23, lambda: load("default"), (True, True, False)),
        ("indirect+AA, cat (no bbox)", 59.38, lambda: load("mesh_scene"), (True, True, False)),
        ("C3 indirect+AA+bbox, cat", 53.77, lambda: load("mesh_scene"), (True, True, True)),
        ("C4 glass cat 1280x720 60spp", None, lambda: big(variant("mesh_scene", "glass"), 1280, 720, 60), (True, True, True)),
        ("C4 mirror cat 1280x720 60spp", None, lambda: big(variant("mesh_scene", "mirror"), 1280, 720, 60), (True, True, True)),
        (f"C5 default two lights 3000x2000 {args.c5_spp}spp", None,
         lambda: two_light_scene("default", 3000, 2000, args.c5_spp), (True, True, False)),
    ]
    results = []
    for name, ref_s, make, mode in rows:
        if args.only and args.only not in name:
            continue
        s = make()
        desc, cam = s.describe()
        dev = s.upload()
        paths = cam.width * cam.height * cam.num_rays
        frames = args.frames if paths < 1e9 else 1
        for culling in (True, False):
            if culling is False and (desc.n_triangles == 0 or args.no_exhaustive):
                continue
            opt = rt.make_options(*mode, seed=1, profile=False, cluster_culling=culling, arithmetic=args.arithmetic)
            dev.render(cam, opt)
            dev_ms, e2e_ms, st = [], [], None
            for _ in range(frames):
                t0 = time.perf_counter()
                _, st = dev.render(cam, opt)
                e2e_ms.append((time.perf_counter() - t0) * 1e3)
                dev_ms.append(st.gpu_ms)
            d, e = float(np.median(dev_ms)), float(np.median(e2e_ms))
            r = {
                "row": name, "cluster_culling": culling, "arithmetic": args.arithmetic, "width": cam.width, "height": cam.height, "spp": cam.num_rays,
                "device_ms": d, "end_to_end_ms": e, "paths": int(st.paths), "mpaths_per_s": st.paths / d / 1e3,
                "triangle_tests_per_path": st.triangle_tests / max(st.paths, 1),
                "reference_gtx1050_s": ref_s, "speedup_vs_reference": (ref_s * 1e3 / d) if ref_s else None,
            }
            results.append(r)
            print(f"| {name} | {'culled' if culling else 'exhaustive'} | {d:.3f} | {e:.3f} | {r['mpaths_per_s']:.0f} | "
                  f"{r['triangle_tests_per_path']:.0f} | {ref_s if ref_s else '-'} | "
                  f"{r['speedup_vs_reference']:.0f}x |" if ref_s else
                  f"| {name} | {'culled' if culling else 'exhaustive'} | {d:.3f} | {e:.3f} | {r['mpaths_per_s']:.0f} | "
                  f"{r['triangle_tests_per_path']:.0f} | - | - |", flush=True)
        del dev
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(results, f, indent=1)


if __name__ == "__main__":
    main()
