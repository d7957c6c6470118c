#!/usr/bin/env python3
"""Which scene parameters reproduce the reference's shipped renders?

The reference's example images (reference examples/*.png, copied as data under
tests/golden/reference_renders/) were drawn at unknown commits of the reference
with a wall-clock seed.  For each image this tool renders the matching scene on
the GPU over a sweep of the parameters the image's name does not fix -- bounce
limit (MAX_RAY_DEPTH, reference src/raytrace.cu:12), anti-aliasing, the glass
indices, the lights -- and prints, per setting, the statistics the test-suite
uses (per-channel mean difference and RMSE of the 9x9-blurred images on the
8-bit scale), for the whole frame and split into the pixels that see the mesh
(primary ray hits a triangle, dilated by 3 pixels) and the rest.  For
deterministic images it also prints the per-pixel agreement.

    python tools/reference_png_sweep.py [mirror|glass|720p|indirect|two_lights|default ...]

Needs a GPU.  Output is a table on stdout.
"""
import json
import os
import sys
import tempfile

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import raytracer_amd as rt  # noqa: E402

RENDERS = os.path.join(ROOT, "tests", "golden", "reference_renders")
SCENES = os.path.join(rt.ASSETS, "scenes")


def scene_from(base, width, height, spp, material=None, ior=None, lights=None):
    doc = json.load(open(os.path.join(SCENES, base + ".json")))
    for o in doc.get("objects", []):
        o["filepath"] = os.path.join(rt.ASSETS, "objects", "cat", "cat.obj")
        if material:
            o["material"] = material
        if ior:
            o["refractive_index"] = list(ior)
    if lights is not None:
        doc["lights"] = lights
    doc["camera"].update(width=width, height=height, num_rays=spp)
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
        json.dump(doc, f)
    s = rt.Scene()
    s.load_scene(f.name)
    os.unlink(f.name)
    return s


def blur(a, r=4):
    out = np.zeros_like(a)
    for dy in range(-r, r + 1):
        for dx in range(-r, r + 1):
            out += np.roll(np.roll(a, dy, 0), dx, 1)
    return out / (2 * r + 1) ** 2


def mesh_mask(dev, cam, dilate=3):
    """Pixels whose undisturbed primary ray hits a triangle, grown by `dilate` pixels."""
    h, w = cam.height, cam.width
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float32)
    d = np.stack([(xs + 0.5 - cam.cx) / cam.fx, (ys + 0.5 - cam.cy) / cam.fy, np.ones_like(xs)], -1).reshape(-1, 3)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rot = np.array(list(cam.rot), np.float32).reshape(3, 3)
    d = (d @ rot.T).astype(np.float32)
    o = np.tile(np.array(list(cam.pos), np.float32), (len(d), 1))
    _, _, tid, _ = dev.cast(rt.make_options(bounding_box=False), o, d)
    m = (tid >= 0).reshape(h, w)
    grown = m.copy()
    for dy in range(-dilate, dilate + 1):
        for dx in range(-dilate, dilate + 1):
            grown |= np.roll(np.roll(m, dy, 0), dx, 1)
    return grown


def compare(img, ref, mask):
    mine = np.floor(np.minimum(img, 255))
    out = {}
    for name, sel in (("all", np.ones_like(mask)), ("mesh", mask), ("rest", ~mask)):
        if not sel.any():
            continue
        md = mine[sel].astype(np.float64).mean(0) - ref[sel].astype(np.float64).mean(0)
        out[name] = md
    diff2 = (blur(mine) - blur(ref)) ** 2
    out["rmse_all"] = float(np.sqrt(diff2.mean()))
    out["rmse_mesh"] = float(np.sqrt(diff2[mask].mean())) if mask.any() else 0.0
    out["rmse_rest"] = float(np.sqrt(diff2[~mask].mean()))
    px = np.abs(mine - ref).max(axis=2)
    out["exact"] = float((px == 0).mean())
    out["within1"] = float((px <= 1).mean())
    return out


def report(label, r):
    f = lambda v: "(" + ", ".join(f"{x:+6.2f}" for x in v) + ")"  # noqa: E731
    print(f"  {label:44s} mean diff all {f(r['all'])} mesh {f(r.get('mesh', [0, 0, 0]))} rest {f(r['rest'])} | "
          f"blurred rmse all {r['rmse_all']:5.2f} mesh {r['rmse_mesh']:5.2f} rest {r['rmse_rest']:5.2f} | "
          f"px exact {r['exact']:.5f} +-1 {r['within1']:.5f}", flush=True)


def load_png(name):
    return np.asarray(Image.open(os.path.join(RENDERS, name)).convert("RGB"), np.float32)


def sweep(title, png, make_scene, settings):
    print(f"== {title}: {png}", flush=True)
    ref = load_png(png)
    cache = {}
    for label, scene_kw, opt_kw in settings:
        key = json.dumps(scene_kw, sort_keys=True)
        if key not in cache:
            s = make_scene(**scene_kw)
            desc, cam = s.describe()
            dev = s.upload()
            cache[key] = (s, cam, dev, mesh_mask(dev, cam) if desc.n_triangles else np.zeros((cam.height, cam.width), bool))
        s, cam, dev, mask = cache[key]
        img, _ = dev.render(cam, rt.make_options(**opt_kw))
        report(label, compare(img, ref, mask))


def main():
    which = set(sys.argv[1:]) or {"mirror", "glass", "720p", "indirect", "two_lights", "default", "aa"}
    rt.init(0)
    depths = (4, 5, 6, 7, 8, 10)
    if "mirror" in which:
        sweep("mirror cat, direct light, no AA (deterministic)", "cat_mirror_512x512_32rays.png",
              lambda **kw: scene_from("mesh_scene", 512, 512, 32, material="mirror", **kw),
              [(f"max_depth {d}", {}, dict(indirect=False, antialiasing=False, max_depth=d)) for d in depths] +
              [("max_depth 7, AA on", {}, dict(indirect=False, antialiasing=True, max_depth=7))])
    if "aa" in which:
        sweep("diffuse cat, direct light + AA", "cat_512x512_32rays_antialias-only.png",
              lambda **kw: scene_from("mesh_scene", 512, 512, 32, **kw),
              [(f"max_depth {d}", {}, dict(indirect=False, antialiasing=True, max_depth=d)) for d in (5, 7)])
    if "glass" in which:
        st = [(f"ior (1, 1.5), max_depth {d}", {}, dict(max_depth=d, seed=1)) for d in depths]
        st += [(f"ior (1, {ri}), max_depth 7", {"ior": (1, ri)}, dict(max_depth=7, seed=1)) for ri in (1.3, 1.4, 1.6)]
        st += [("ior (1, 1.5), max_depth 7, AA off", {}, dict(max_depth=7, antialiasing=False, seed=1)),
               ("ior (1, 1.5), max_depth 7, direct only", {}, dict(max_depth=7, indirect=False, seed=1)),
               ("ior (1, 1.5), max_depth 7, seed 2", {}, dict(max_depth=7, seed=2))]
        sweep("glass cat, indirect + AA", "cat_glass_512x512_32rays_indirect.png",
              lambda **kw: scene_from("mesh_scene", 512, 512, 32, material="glass", **kw), st)
    if "indirect" in which:
        sweep("diffuse cat, indirect + AA", "cat_512x512_32rays_indirect.png",
              lambda **kw: scene_from("mesh_scene", 512, 512, 32, **kw),
              [(f"max_depth {d}", {}, dict(max_depth=d, seed=1)) for d in depths])
    if "720p" in which:
        sweep("diffuse cat 1280x720, 60 rays, indirect + AA", "cat_720p_60rays_indirect.png",
              lambda **kw: scene_from("mesh_scene", 1280, 720, 60, **kw),
              [(f"max_depth {d}", {}, dict(max_depth=d, seed=1)) for d in (1, 2, 3) + depths])
    if "default" in which:
        sweep("default scene, indirect + AA", "default_512x512_32rays_indirect.png",
              lambda **kw: scene_from("default", 512, 512, 32, **kw),
              [(f"max_depth {d}", {}, dict(max_depth=d, seed=1)) for d in depths])
    if "two_lights" in which:
        def lights(i0, i1, x=10):
            return [{"pos": [-x, 20, 40], "intensity": i0}, {"pos": [x, 20, 40], "intensity": i1}]
        st = []
        for d in (5, 7, 10):
            st.append((f"2 x 5e9, max_depth {d}", {"lights": lights(5e9, 5e9)}, dict(max_depth=d, seed=1)))
        for i in (7e9, 1e10, 1.5e10, 2e10):
            st.append((f"2 x {i:.1e}, max_depth 7", {"lights": lights(i, i)}, dict(max_depth=7, seed=1)))
            st.append((f"2 x {i:.1e}, max_depth 10", {"lights": lights(i, i)}, dict(max_depth=10, seed=1)))
        sweep("default scene, two lights, 1000 rays (rendered here with 200)", "default_512x512_1000rays_indirect_two_lights.png",
              lambda **kw: scene_from("default", 512, 512, 200, **kw), st)


if __name__ == "__main__":
    main()
