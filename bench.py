#!/usr/bin/env python3
"""Benchmark of the render hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step is one frame of the workload the headline metric is quoted on: the cat
mesh scene (assets/scenes/mesh_scene.json, 3 954 triangles) at 512x512, 32
samples per pixel, 5 bounces, indirect lighting + anti-aliasing + bounding box.
The scene is resident in HBM before the timed region; every step renders the
whole frame in the library's default mode (cluster culling on).  With N > 1
(one process per GPU, launched by torch.distributed.run) the 32 samples of
every pixel are split over the ranks, each rank accumulates linear radiance for
its sample range, the buffers are summed onto rank 0 with one RCCL reduce and
rank 0 resolves the image: total work is fixed (strong scaling).

Kernel durations for the roofline objects come from HIP events recorded, on the
stream the kernels are launched on, around the kernels of three steps of the
timed region that run alone on the device; the events are read after the
region, so the host never waits inside it.

After the timed region rank 0 also times the exhaustive-scan mode (every
queued ray against every triangle, the reference's own amount of work) and, at
N = 1, the CPU oracle on a bounded sample of the same frame.  Rank 0 prints one
JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Algorithmic work per test of the reference formulation: 47 flop per
# ray-triangle test (3 vector subtractions, 2 cross products, 3 dot products,
# 2 divisions, 1 negation, 2 subtractions; reference src/raytrace.cu:155-164).
FLOP_PER_TRIANGLE_TEST = 47
FLOP_PER_SPHERE_TEST = 19
FLOP_PER_BOX_TEST = 22
PEAK_FP32_VECTOR_TFLOPS = 157.3  # MI355X fp32 vector peak (packed FMA)
# the reference's published time for this configuration: 53.77 s on a GTX 1050
# (reference README.md:78) = 512*512*32 / 53.77 paths per second
BASELINE_MPATHS = 512 * 512 * 32 / 53.77 / 1e6


def cpu_baseline(desc, cam, opt, n_pixels_sample=65536):
    """Times the CPU oracle on a bounded sample of the same frame."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob

    # every CPU this process may run on (the GPU box grants a share of the machine per GPU and
    # exports it as OMP_NUM_THREADS; os.cpu_count() there is the whole machine)
    allowed = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or allowed
    threads = max(1, min(threads, allowed))
    rng = np.random.default_rng(1234)
    pixels = rng.choice(cam.width * cam.height, size=n_pixels_sample, replace=False).astype(np.uint32)
    oopt = ob.from_rt_options(opt)
    t0 = time.perf_counter()
    _, _, _, counters = ob.render_pixels(desc, cam, oopt, pixels, threads=threads)
    dt = time.perf_counter() - t0
    # the same on one thread, on a sixteenth of the sample
    t1 = time.perf_counter()
    _, _, _, single = ob.render_pixels(desc, cam, oopt, pixels[: max(n_pixels_sample // 16, 1)], threads=1)
    dt1 = time.perf_counter() - t1
    return {
        "value": counters["paths"] / dt / 1e6,
        "unit": "Mpaths/s",
        "cores": threads,
        "host_cpus": os.cpu_count(),
        "cpus_allowed": allowed,
        "kind": "port",
        "sample": f"{n_pixels_sample} random pixels x {cam.num_rays} spp of the same frame "
                  f"({counters['paths']} paths, {dt:.1f} s, {counters['triangle_tests'] / counters['paths']:.0f} "
                  f"triangle tests/path)",
        "single_thread_value": single["paths"] / dt1 / 1e6,
        "single_thread_sample": f"{single['paths']} paths, {dt1:.1f} s",
    }


class Meter:
    """Sums the per-frame kernel statistics of the timed steps."""

    def __init__(self):
        self.mesh_ms = self.bin_ms = self.shade_ms = 0.0
        self.launches = self.tests = self.frames = self.shade_launches = self.casts = 0

    def add(self, st):
        if st is None:
            return
        self.mesh_ms += st.mesh_kernel_ms
        self.bin_ms += st.bin_kernel_ms
        self.launches += st.mesh_kernel_launches
        self.tests += st.triangle_tests
        self.shade_ms += st.shade_kernel_ms
        self.shade_launches += st.shade_kernel_launches
        self.casts += st.sphere_casts
        self.frames += 1

    def add_marks(self, marks, counted):
        """A sampling step: the kernel times its marks gave, and the work a frame executes (counted once)."""
        self.mesh_ms += marks.mesh_kernel_ms
        self.bin_ms += marks.bin_kernel_ms
        self.launches += marks.mesh_kernel_launches
        self.shade_ms += marks.shade_kernel_ms
        self.shade_launches += marks.shade_kernel_launches
        self.tests += counted.triangle_tests
        self.casts += counted.sphere_casts
        self.frames += 1

    def roofline(self, kernel, step_ms, traffic):
        """The triangle kernel: 47 flop per ray-triangle test it executed.  (In the default mode the kernel,
        k_group_clusters, also bins every ray of a group into the group's clusters before it scans them; that
        work is not counted.)"""
        achieved = self.tests * FLOP_PER_TRIANGLE_TEST / (self.mesh_ms * 1e-3) / 1e12 if self.mesh_ms > 0 else 0.0
        return {
            "bound": "valu",
            "kernel": kernel,
            "achieved": achieved,
            "peak": PEAK_FP32_VECTOR_TFLOPS,
            "unit": "TFLOP/s",
            "frac": achieved / PEAK_FP32_VECTOR_TFLOPS,
            "traffic": traffic,
            # the same launch against the other roofline: measured HBM bytes per launch (PMC, profiles/) over its duration
            "hbm_gbps": (traffic / (self.mesh_ms / self.launches * 1e-3) / 1e9) if traffic and self.launches and self.mesh_ms > 0 else None,
            "hbm_peak_gbps": 8000.0,
            "avg_launch_ms": self.mesh_ms / self.launches if self.launches else None,
            "launches_per_step": self.launches / max(self.frames, 1),
            "triangle_tests_per_step": self.tests / max(self.frames, 1),
            "kernel_share_of_step": self.mesh_ms / max(self.frames, 1) / step_ms if step_ms else None,
            "binning_ms_per_step": self.bin_ms / max(self.frames, 1),
            "sampled_steps": self.frames,
        }

    def shade_roofline(self, step_ms, n_spheres, n_boxes, generate_casts_per_frame, traffic):
        """The bounce kernel against the same VALU peak.  Algorithmic work = the sphere scans and box
        tests of the rays it casts (19 flop per ray-sphere test, 22 per ray-box test, the reference's
        formulation, src/raytrace.cu:53-146); the shading arithmetic itself is not counted."""
        if not self.shade_launches or self.shade_ms <= 0:
            return None
        casts = self.casts - generate_casts_per_frame * self.frames  # camera rays are cast by k_generate
        flop = casts * (FLOP_PER_SPHERE_TEST * n_spheres + FLOP_PER_BOX_TEST * n_boxes)
        achieved = flop / (self.shade_ms * 1e-3) / 1e12
        hbm = traffic / (self.shade_ms / self.shade_launches * 1e-3) / 1e9 if traffic else None
        return {
            "bound": "valu",
            "kernel": "k_shade",
            "achieved": achieved,
            "peak": PEAK_FP32_VECTOR_TFLOPS,
            "unit": "TFLOP/s",
            "frac": achieved / PEAK_FP32_VECTOR_TFLOPS,
            "traffic": traffic,
            "hbm_gbps": hbm,
            "hbm_peak_gbps": 8000.0,
            "hbm_frac": hbm / 8000.0 if hbm else None,
            "avg_launch_ms": self.shade_ms / self.shade_launches,
            "launches_per_step": self.shade_launches / max(self.frames, 1),
            "casts_per_step": casts / max(self.frames, 1),
            "kernel_share_of_step": self.shade_ms / max(self.frames, 1) / step_ms if step_ms else None,
            "sampled_steps": self.frames,
            "note": "dominant kernel of the step; instruction-issue bound (VALU busy in profiles/), its own arithmetic "
                    "(hit selection, materials, IEEE divisions and square roots, the generator) is not counted as algorithmic work",
        }

    def frame_roofline(self, step_ms, n_spheres, n_boxes, world=1):
        """The whole step on the algorithmic work it executed: triangle tests, sphere tests and box tests of every ray
        (rank 0's counters times the number of ranks: the samples are split evenly), against the peak of all GPUs."""
        if not self.frames or not step_ms:
            return None
        flop = world * (self.tests * FLOP_PER_TRIANGLE_TEST + self.casts * (FLOP_PER_SPHERE_TEST * n_spheres + FLOP_PER_BOX_TEST * n_boxes)) / self.frames
        achieved = flop / (step_ms * 1e-3) / 1e12
        return {"bound": "valu", "kernel": "whole step", "achieved": achieved, "peak": PEAK_FP32_VECTOR_TFLOPS * world, "unit": "TFLOP/s",
                "frac": achieved / (PEAK_FP32_VECTOR_TFLOPS * world), "traffic": None, "gflop_per_step": flop / 1e9}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scene", default="mesh_scene")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exhaustive", action="store_true")
    ap.add_argument("--frames-in-flight", type=int, default=3,
                    help="frames a rank keeps in flight, each on its own stream and scene handle (1 = one after the other)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # RT_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    device = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(device)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend)

    import raytracer_amd as rt
    from raytracer_amd.distributed import FrameRenderer

    rt.init(device)
    scene = rt.Scene()
    scene.load_scene(os.path.join(rt.ASSETS, "scenes", args.scene + ".json"))
    desc, cam = scene.describe()
    dev = scene.upload()
    # every frame in flight needs work buffers of its own: further handles of the same scene
    extra_devs = [scene.upload() for _ in range(max(args.frames_in_flight, 1) - 1)]
    n_pixels = cam.width * cam.height
    spp = cam.num_rays

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run(cluster_culling, steps, warmup):
        opt = rt.make_options(indirect=True, antialiasing=True, bounding_box=True, max_depth=5, seed=1, profile=True,
                              cluster_culling=cluster_culling)
        # only rank 0's image is looked at: the sums are reduced onto it (half an all-reduce's traffic)
        # the marks of a sampling step are recorded on the stream and read after the region (profile = 2)
        opt_marks = rt.make_options(indirect=True, antialiasing=True, bounding_box=True, max_depth=5, seed=1, profile=2,
                                    cluster_culling=cluster_culling)
        renderer = FrameRenderer(dev, cam, opt, want_stats=True, extra_devs=extra_devs, image_on_root_only=True)
        meter = Meter()
        for _ in range(warmup):
            renderer.render()
        # what a frame executes (rays cast, triangle tests): every step renders the same frame, so the counters are
        # read once, here, outside the timed region
        counted = renderer.last_stats
        if counted is None:
            renderer.render()
            counted = renderer.last_stats
        renderer.want_stats = False
        fence()
        pending = []  # scene handles with marks to read
        t0 = time.perf_counter()
        for i in range(steps):
            # Steps just enqueue: consecutive frames overlap on the device (a stream per frame in flight), and with
            # N > 1 so do the exchange of one frame and the kernels of the next.  Three steps of the region (its
            # first, its middle and its last one) are sampling steps: the frame runs ALONE on the device -- its stream
            # waits for the frames in flight, the later frames' streams wait for it -- with HIP events recorded around
            # its kernels on the stream they are launched on.  The host waits for nothing: the events are read after
            # the region.  (The first and the last step are where the device is empty, or has to drain, anyway.)
            sampling = i in (0, steps // 2, steps - 1)
            if sampling:
                handle = renderer.devs[renderer.frame % len(renderer.devs)] if renderer.streams else renderer.dev
                if handle in pending:  # its earlier sampling step finished long ago: reading its marks waits for nothing
                    meter.add_marks(handle.profile_collect(), counted)
                    pending.remove(handle)
                # (with N > 1 a rank's share of a frame is a sixth of a millisecond of work in thirty kernels, and a frame
                # that runs alone costs more than the step itself: there the sampled frames stay in flight beside the
                # others and their kernels' durations are stretched by what runs next to them)
                renderer.render_async(opt=opt_marks, alone=world == 1)
                pending.append(renderer.last_dev)
            else:
                renderer.render_async()
        fence()
        elapsed = time.perf_counter() - t0
        for handle in pending:
            meter.add_marks(handle.profile_collect(), counted)
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()), meter, opt

    # set-up, not a step: one frame per handle so that the work buffers exist whatever --warmup is
    for handle in [dev, *extra_devs]:
        FrameRenderer(handle, cam, rt.make_options(seed=1), depth=1).render()
    fence()

    elapsed, meter, opt = run(True, args.steps, args.warmup)
    # the time of ONE frame on its own (nothing else in flight), beside the throughput figure above
    single_opt = rt.make_options(indirect=True, antialiasing=True, bounding_box=True, max_depth=5, seed=1)
    single = FrameRenderer(dev, cam, single_opt)
    single.render()
    fence()
    single_frames = max(3, min(args.steps, 20))
    t0 = time.perf_counter()
    for _ in range(single_frames):
        single.render()
        torch.cuda.synchronize()
    single_ms = (time.perf_counter() - t0) / single_frames * 1e3
    fence()
    exhaustive = None
    if not args.no_exhaustive:
        ex_steps = max(1, min(args.steps, 5))
        ex_elapsed, ex_meter, _ = run(False, ex_steps, 1)
        exhaustive = (ex_elapsed, ex_meter, ex_steps)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = n_pixels * spp / (elapsed / args.steps) / 1e6
        traffic = {}
        tpath = os.path.join(ROOT, "profiles", "mesh_kernel_traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                traffic = json.load(f)
        line = {
            "metric": "Mpaths/s, cat mesh 512x512 32spp 5-bounce indirect+AA+bbox",
            "value": value,
            "unit": "Mpaths/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "s_per_frame": ms_per_step / 1e3,  # inverse throughput with frames in flight
            "s_per_frame_single": single_ms / 1e3,  # one frame rendered alone, submit to finished (frames_in_flight = 1)
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": value / BASELINE_MPATHS,
            "dtype": "f32",
            "data": "the reference's own scene (assets/scenes/mesh_scene.json and the cat.obj mesh it names, byte-identical copies), "
                    "fixed seed; nothing synthetic",
            "config": {
                "workload": f"{args.scene}.json {cam.width}x{cam.height} {spp}spp depth5 indirect+AA+bbox",
                "triangles": int(desc.n_triangles),
                "mode": "cluster culling on (library default; image bit-identical to the exhaustive scan)",
                "parallelism": f"spp-split x{world} + RCCL reduce of linear radiance onto rank 0" if world > 1 else "single GPU",
                "frames_in_flight": max(args.frames_in_flight, 1),
            },
        }
        from raytracer_amd.distributed import sample_range
        begin, end = sample_range(0, world, spp)  # rank 0's share (the statistics are rank 0's)
        own_samples = end - begin
        # `roofline` names the kernel with the largest share of the step (k_shade in the default mode).  Durations are
        # HIP events on the launch stream around rank 0's launches on the sampling steps; algorithmic work per test as above.
        shade = meter.shade_roofline(ms_per_step, int(desc.n_spheres), int(desc.n_boxes), n_pixels * own_samples,
                                     traffic.get("k_shade_hbm_bytes_per_launch"))
        mesh = meter.roofline("k_group_clusters", ms_per_step, traffic.get("k_group_clusters_hbm_bytes_per_launch"))
        if shade is not None and shade["kernel_share_of_step"] >= (mesh["kernel_share_of_step"] or 0):
            line["roofline"], line["roofline_k_group_clusters"] = shade, mesh
        else:
            line["roofline"], line["roofline_k_shade"] = mesh, shade
        line["roofline_whole_step"] = meter.frame_roofline(ms_per_step, int(desc.n_spheres), int(desc.n_boxes), world)
        line["kernel_durations"] = ("HIP events on the launch stream around the kernels of three steps of the timed region "
                                    + ("that ran alone on the device" if world == 1 else
                                       "in flight beside the other frames (stretched by what ran next to them)")
                                    + ", read after the region")
        if exhaustive is not None:
            ex_elapsed, ex_meter, ex_steps = exhaustive
            line["exhaustive_scan"] = {
                "note": "cluster_culling=0: every queued ray against every triangle, the reference's own work "
                        "(7 175 triangle tests per path)",
                "steps": ex_steps,
                "ms_per_step": ex_elapsed / ex_steps * 1e3,
                "value": n_pixels * spp / (ex_elapsed / ex_steps) / 1e6,
                "unit": "Mpaths/s",
                "roofline": ex_meter.roofline("k_mesh", ex_elapsed / ex_steps * 1e3, traffic.get("hbm_bytes_per_launch")),
            }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(desc, cam, opt)
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
