import os, sys, time
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo') else '.')
import torch
import raytracer_amd as rt
from raytracer_amd.distributed import FrameRenderer
rt.init(0)
scene = rt.Scene(); scene.load_scene(os.path.join(rt.ASSETS, "scenes", "mesh_scene.json"))
desc, cam = scene.describe(); dev = scene.upload()
for spp in (32, 4):
    cam.num_rays = spp
    for profile, stats in ((True, True), (False, True), (False, False)):
        opt = rt.make_options(seed=1, profile=profile)
        r = FrameRenderer(dev, cam, opt, want_stats=stats)
        for _ in range(3): r.render()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 30
        for _ in range(n): r.render()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n * 1e3
        print(f"spp {spp} profile {profile} stats {stats}: {dt:.3f} ms/frame")
