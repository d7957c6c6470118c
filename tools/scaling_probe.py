#!/usr/bin/env python3
"""Rehearses the per-rank load of an N-GPU run on ONE GPU.

For N in 1, 2, 4, 8 the process renders the share of samples rank 0 of N would
get (spp / N per pixel), runs the all-reduce on a one-rank RCCL group (so the
collective's launch path and the stream overlap are exercised; the transfer
itself is missing) and reports ms per frame.  ms(1) / ms(N) is an upper bound
of the strong-scaling factor bench.py --gpus N can reach.

    python tools/scaling_probe.py [N ...] [word ...]   # words select arrangements: serial, overlapped, two, three, four (also five, six, eight);
                                                       # "fast": rt_options.arithmetic = 1
"""
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    import raytracer_amd as rt
    from raytracer_amd.distributed import FrameRenderer, sample_range

    rt.init(0)
    scene = rt.Scene()
    scene.load_scene(os.path.join(rt.ASSETS, "scenes", "mesh_scene.json"))
    desc, cam = scene.describe()
    dev = scene.upload()
    opt = rt.make_options(seed=1, arithmetic=1 if "fast" in sys.argv[1:] else 0)  # word "fast": rt_options.arithmetic = 1, as bench.py runs by default
    out = {}
    shares = [int(a) for a in sys.argv[1:] if a.isdigit()] or [1, 2, 4, 8]
    only = [a for a in sys.argv[1:] if not a.isdigit() and a != "fast"]  # e.g. "serial" or "three": arrangements whose label contains the word
    words = {2: "two", 3: "three", 4: "four", 5: "five", 6: "six", 8: "eight"}
    arrangements = [f"{words[k]} handles" for k in (4, 3, 2)] + [True, False]
    arrangements = [f"{w} handles" for w in only if w in words.values() and f"{w} handles" not in arrangements] + arrangements
    handles = {f"{w} handles": k for k, w in words.items()}
    for n in shares:
        for overlapped in arrangements:
            name = overlapped if isinstance(overlapped, str) else ("overlapped" if overlapped else "serial")
            if only and not any(word in name for word in only):
                continue
            r = extra = None  # frees the previous arrangement's handles (each holds gigabytes of ray lists) before new ones are made
            gc.collect()
            extra = [scene.upload() for _ in range(handles.get(overlapped, 1) - 1)]
            r = FrameRenderer(dev, cam, opt, exchange_in_group_of_one=True, extra_devs=extra)
            r.s_begin, r.s_end = sample_range(0, n, cam.num_rays)  # rank 0's share of an n-rank run; all rows
            step = r.render_async if overlapped else r.render
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            frames = 40
            t0 = time.perf_counter()
            for _ in range(frames):
                step()
            submit_ms = (time.perf_counter() - t0) / frames * 1e3  # host time to queue a frame
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / frames * 1e3
            label = overlapped if isinstance(overlapped, str) else ("overlapped" if overlapped else "serial")
            out[f"share_1_of_{n}_{label.replace(' ', '_')}_ms"] = ms
            print(f"share 1/{n} ({r.s_end - r.s_begin} spp) {label:11s}: {ms:.3f} ms/frame (host queues a frame in {submit_ms:.3f} ms)", flush=True)
    print(json.dumps(out))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
