#!/usr/bin/env python3
"""Benchmark of the render hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config c3] [--arithmetic fast]

A step is one frame of a benchmark configuration.  The default, c3, is the one the headline metric is quoted on:
the cat mesh scene (assets/scenes/mesh_scene.json, 3 954 triangles) at 512x512, 32 samples per pixel, 5 bounces,
indirect lighting + anti-aliasing + bounding box.  The others are the configurations that shard over more GPUs
with more work per rank: c4_glass / c4_mirror (the cat as glass / mirror, 1280x720, 60 samples), c5 (default.json
with two lights, 3000x2000, 1000 samples), and the two small sphere scenes c1 / c2.  The scene is resident in HBM
before the timed region; every step renders the whole frame in the library's default mode (cluster culling on).
With N > 1 (one process per GPU, launched by torch.distributed.run) the samples of every pixel are split over the
ranks, each rank accumulates linear radiance for its sample range, the buffers are summed onto rank 0 with one RCCL
reduce and rank 0 resolves the image: total work is fixed (strong scaling).

`value` is measured with rt_options.arithmetic = 1 (fused multiply-adds, hardware reciprocals and roots: per pixel
within 1e-4 of the bit-exact mode where no random number is involved, equal converged means where one is --
tests/test_gpu_fast_arithmetic.py); the bit-exact mode (--arithmetic exact, the library's default) is timed beside it
and reported as `exact_arithmetic`.

Kernel durations for the roofline objects come from HIP events recorded, on the stream the kernels are launched
on, around the kernels of three steps of the timed region that run alone on the device; the events are read after
the region, so the host never waits inside it.

After the timed region rank 0 also times the other arithmetic, the exhaustive-scan mode (c3 only: every queued ray
against every triangle, the reference's own amount of work) and, at N = 1, the CPU oracle on a bounded sample of
the same frame.  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import tempfile
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
SIMDS, SHADER_HZ = 1024, 2.4e9   # 256 CUs x 4 SIMDs; a wave64 VALU instruction occupies its SIMD for 4 cycles
# the reference's published time for the headline configuration: 53.77 s on a GTX 1050
# (reference README.md:78) = 512*512*32 / 53.77 paths per second
BASELINE_MPATHS = 512 * 512 * 32 / 53.77 / 1e6

# name -> scene file, cat material override, lights override, (width, height, spp) override, (indirect, aa, bbox),
#         steps when --steps is not given, metric label
TWO_LIGHTS_OF_SIMPLE_JSON = "simple"
CONFIGS = {
    "c1": dict(scene="default", size=None, modes=(False, True, True), steps=200,
               label="default.json 512x512 32spp 5-bounce direct+AA"),
    "c2": dict(scene="default", size=None, modes=(True, True, False), steps=200,
               label="default.json 512x512 32spp 5-bounce indirect+AA"),
    "c3": dict(scene="mesh_scene", size=None, modes=(True, True, True), steps=200,
               label="cat mesh 512x512 32spp 5-bounce indirect+AA+bbox"),
    "c4_glass": dict(scene="mesh_scene", material="glass", size=(1280, 720, 60), modes=(True, True, True), steps=30,
                     label="glass cat 1280x720 60spp 5-bounce indirect+AA+bbox"),
    "c4_mirror": dict(scene="mesh_scene", material="mirror", size=(1280, 720, 60), modes=(True, True, True), steps=30,
                      label="mirror cat 1280x720 60spp 5-bounce indirect+AA+bbox"),
    "c5": dict(scene="default", lights_of=TWO_LIGHTS_OF_SIMPLE_JSON, size=(3000, 2000, 1000), modes=(True, True, False), steps=3,
               label="default.json with two lights 3000x2000 1000spp 5-bounce indirect+AA"),
}


def make_scene(rt, cfg):
    """The configuration's scene through the library's own loader (Scene::load_scene)."""
    scenes = os.path.join(rt.ASSETS, "scenes")
    path = os.path.join(scenes, cfg["scene"] + ".json")
    tmp = None
    if cfg.get("material") or cfg.get("lights_of"):
        with open(path) as f:
            doc = json.load(f)
        for o in doc.get("objects", []):
            if cfg.get("material"):
                o["material"] = cfg["material"]
            o["filepath"] = os.path.join(rt.ASSETS, "objects", "cat", "cat.obj")
        if cfg.get("lights_of"):
            with open(os.path.join(scenes, cfg["lights_of"] + ".json")) as f:
                doc["lights"] = json.load(f)["lights"]
        tmp = tempfile.NamedTemporaryFile("w", suffix=".json", delete=False)
        json.dump(doc, tmp)
        tmp.close()
        path = tmp.name
    scene = rt.Scene()
    scene.load_scene(path)
    if tmp is not None:
        os.unlink(tmp.name)
    if cfg.get("size"):
        w, h, spp = cfg["size"]
        scene.set_camera_intrinsics(60.0, w, h)
        scene.set_num_rays(spp)
    return scene


def cpu_baseline(desc, cam, opt, n_pixels_sample=65536):
    """Times the CPU oracle on a bounded sample of the same frame."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob

    # every CPU this process may run on (the GPU box grants a share of the machine per GPU and
    # exports it as OMP_NUM_THREADS; os.cpu_count() there is the whole machine)
    allowed = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or allowed
    threads = max(1, min(threads, allowed))
    # about 10-30 s of CPU work whatever the configuration: scale the pixel sample to the samples per pixel
    n_pixels_sample = int(min(cam.width * cam.height, max(256, n_pixels_sample * 32 // max(cam.num_rays, 1))))
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
        "arithmetic": "the reference's float arithmetic, compiled without contraction (the exact mode's oracle)",
        "sample": f"{n_pixels_sample} random pixels x {cam.num_rays} spp of the same frame "
                  f"({counters['paths']} paths, {dt:.1f} s, {counters['triangle_tests'] / counters['paths']:.0f} "
                  f"triangle tests/path)",
        "single_thread_value": single["paths"] / dt1 / 1e6,
        "single_thread_sample": f"{single['paths']} paths, {dt1:.1f} s",
    }


class Meter:
    """Sums the per-frame kernel statistics of the timed steps."""

    def __init__(self):
        self.mesh_ms = self.bin_ms = self.shade_ms = self.gpu_ms = 0.0
        self.launches = self.tests = self.frames = self.shade_launches = self.casts = 0
        self.path_bounces = self.path_records = self.paths = self.mesh_rays = 0

    def add_marks(self, marks, counted):
        """A sampling step: the kernel times its marks gave, and the work a frame executes (counted once)."""
        self.mesh_ms += marks.mesh_kernel_ms
        self.bin_ms += marks.bin_kernel_ms
        self.launches += marks.mesh_kernel_launches
        self.shade_ms += marks.shade_kernel_ms
        self.shade_launches += marks.shade_kernel_launches
        self.gpu_ms += marks.gpu_ms
        self.tests += counted.triangle_tests
        self.casts += counted.sphere_casts
        self.path_bounces += counted.path_bounces
        self.path_records += counted.path_records
        self.paths += counted.paths
        self.mesh_rays += counted.mesh_rays
        self.frames += 1

    @staticmethod
    def _counters(static, kernel, avg_launch_ms):
        """Hardware-counter figures of `kernel` from profiles/kernel_counters.json (static: measured by a separate
        rocprofv3 --pmc run of this command, not in this run) against this run's measured launch duration."""
        c = (static or {}).get(kernel)
        if not c or not avg_launch_ms:
            return {"traffic": None, "hbm_gbps": None, "hbm_frac": None, "issue_frac": None, "counters": None}
        seconds = avg_launch_ms * 1e-3
        hbm = c["hbm_bytes_per_launch"] / seconds / 1e9
        return {
            "traffic": c["hbm_bytes_per_launch"],
            "hbm_gbps": hbm,
            "hbm_peak_gbps": 8000.0,
            "hbm_frac": hbm / 8000.0,
            # the ceiling that binds these kernels: VALU wave-instructions x 4 cycles over the cycles all SIMDs had
            "issue_frac": c["valu_wave_instructions_per_launch"] * 4 / (SIMDS * SHADER_HZ * seconds),
            "counters": f"static, from profiles/{c['source']} ({c['note']})",
        }

    def mesh_roofline(self, kernel, step_ms, static):
        """The triangle kernel: 47 flop per ray-triangle test it executed.  (In the default mode the kernel,
        k_group_clusters, also bins every ray of a group into the group's clusters before it scans them; that
        work is not counted.)"""
        if not self.launches or self.mesh_ms <= 0:
            return None
        achieved = self.tests * FLOP_PER_TRIANGLE_TEST / (self.mesh_ms * 1e-3) / 1e12
        avg = self.mesh_ms / self.launches
        out = {
            "bound": "valu",
            "kernel": kernel,
            "achieved": achieved,
            "peak": PEAK_FP32_VECTOR_TFLOPS,
            "unit": "TFLOP/s",
            "frac": achieved / PEAK_FP32_VECTOR_TFLOPS,
            "avg_launch_ms": avg,
            "launches_per_step": self.launches / max(self.frames, 1),
            "triangle_tests_per_step": self.tests / max(self.frames, 1),
            "kernel_share_of_step": self.mesh_ms / max(self.frames, 1) / step_ms if step_ms else None,
            "binning_ms_per_step": self.bin_ms / max(self.frames, 1),
            "sampled_steps": self.frames,
        }
        out.update(self._counters(static, kernel, avg))
        return out

    def shade_roofline(self, kernel, step_ms, n_spheres, n_boxes, generate_casts_per_frame, static, fast):
        """The bounce kernel.  With the exact arithmetic it is bound by instruction issue and is priced against the
        VALU peak: algorithmic work = the sphere scans and box tests of the rays it casts (19 flop per ray-sphere
        test, 22 per ray-box test, the reference's formulation, src/raytrace.cu:53-146; the shading arithmetic itself
        is not counted), with issue_frac beside it.  The fast arithmetic is priced against HBM as well, because the
        path state the kernel moved was what bound it (throw-away builds: 56 bytes per path not written, -26 %; 2-3 %
        fewer instructions, nothing) until paths that need nothing from the triangle stage stayed in their lanes:
        algorithmic bytes = what the data layout makes the kernel move -- 76 bytes read per path it fetches (72 at the
        first bounce), 76 written per path it hands to the triangle stage, 16 of colour per path that ends, 32 per ray
        queued.  `bound` names the nearer ceiling by the counters (hbm_frac against issue_frac)."""
        if not self.shade_launches or self.shade_ms <= 0:
            return None
        casts = self.casts - generate_casts_per_frame * self.frames  # camera rays are cast by k_generate
        flop = casts * (FLOP_PER_SPHERE_TEST * n_spheres + FLOP_PER_BOX_TEST * n_boxes)
        tflops = flop / (self.shade_ms * 1e-3) / 1e12
        avg = self.shade_ms / self.shade_launches
        counters = self._counters(static, kernel, avg)
        common = {
            "kernel": kernel,
            "avg_launch_ms": avg,
            "launches_per_step": self.shade_launches / max(self.frames, 1),
            "casts_per_step": casts / max(self.frames, 1),
            "path_bounces_per_step": self.path_bounces / max(self.frames, 1),
            "path_records_per_step": self.path_records / max(self.frames, 1),
            "kernel_share_of_step": self.shade_ms / max(self.frames, 1) / step_ms if step_ms else None,
            "sampled_steps": self.frames,
        }
        valu = {"achieved": tflops, "peak": PEAK_FP32_VECTOR_TFLOPS, "unit": "TFLOP/s", "frac": tflops / PEAK_FP32_VECTOR_TFLOPS}
        if fast and self.path_records:
            # fetched: 76 per record (72 at the first bounce); handed on: 76 per record after the first; ended: 16 per path
            algorithmic = 152 * self.path_records - 64 * self.paths + 32 * self.mesh_rays
            gbps = algorithmic / (self.shade_ms * 1e-3) / 1e9
            hbm = {"achieved": gbps, "peak": 8000.0, "unit": "GB/s", "frac": gbps / 8000.0,
                   "algorithmic_bytes_per_launch": algorithmic / self.shade_launches}
            note = ("dominant kernel of the step, priced against both ceilings: the path state it moves through HBM (152 bytes per "
                    "path fetched and handed on, 32 per queued ray; traffic = HBM bytes per launch by the PMC counters) and the "
                    "sphere scans and box tests of its rays against the VALU peak; issue_frac is the share of the issue slots its "
                    "instructions take")
            # the nearer ceiling by the counters: HBM bytes moved over the HBM peak, or issue slots taken
            if max(counters.get("hbm_frac") or 0.0, hbm["frac"]) >= (counters.get("issue_frac") or 0.0):
                out = dict(hbm, bound="hbm", valu=valu, note=note)
            else:
                out = dict(valu, bound="valu", hbm=hbm, note=note)
        else:
            out = dict(valu, bound="valu",
                       note="dominant kernel of the step; bound by instruction issue (issue_frac), its own arithmetic (hit selection, "
                            "materials, normalisations, the generator) is not counted as algorithmic work")
        out.update(common)
        out.update(counters)
        return out

    def spheres_roofline(self, kernel, step_ms, n_spheres, static):
        """A scene without triangles is one kernel (k_trace_spheres): every ray against every sphere."""
        if not self.frames or self.gpu_ms <= 0:
            return None
        flop = self.casts * FLOP_PER_SPHERE_TEST * n_spheres
        achieved = flop / (self.gpu_ms * 1e-3) / 1e12
        avg = self.gpu_ms / self.frames
        out = {
            "bound": "valu", "kernel": kernel, "achieved": achieved, "peak": PEAK_FP32_VECTOR_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / PEAK_FP32_VECTOR_TFLOPS, "avg_launch_ms": avg, "launches_per_step": 1,
            "casts_per_step": self.casts / self.frames, "kernel_share_of_step": avg / step_ms if step_ms else None,
            "sampled_steps": self.frames,
            "note": "the frame's kernels as one (a lane runs the whole sample and bounce loop in registers); algorithmic work = "
                    "19 flop per ray-sphere test, the shading arithmetic is not counted",
        }
        out.update(self._counters(static, kernel, avg))
        return out

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
    ap.add_argument("--steps", type=int, default=None, help="default: 200 (c1-c3), 30 (c4), 3 (c5)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c3")
    ap.add_argument("--arithmetic", choices=["fast", "exact"], default="fast",
                    help="rt_options.arithmetic of the timed steps (the other one is timed beside it)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exhaustive", action="store_true")
    ap.add_argument("--no-other-arithmetic", action="store_true")
    ap.add_argument("--frames-in-flight", type=int, default=3,
                    help="frames a rank keeps in flight, each on its own stream and scene handle (1 = one after the other)")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    steps = args.steps if args.steps is not None else cfg["steps"]

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
    from raytracer_amd.distributed import FrameRenderer, sample_range

    rt.init(device)
    scene = make_scene(rt, cfg)
    desc, cam = scene.describe()
    has_mesh = int(desc.n_triangles) > 0
    # frames in flight need work buffers of their own (further handles of the same scene); a scene without
    # triangles is one kernel per frame and has no work buffers: one handle
    in_flight = max(args.frames_in_flight, 1) if has_mesh else 1
    dev = scene.upload()
    extra_devs = [scene.upload() for _ in range(in_flight - 1)]
    n_pixels = cam.width * cam.height
    spp = cam.num_rays
    indirect, antialiasing, bounding_box = cfg["modes"]

    def options(arithmetic, cluster_culling=True, profile=False):
        return rt.make_options(indirect=indirect, antialiasing=antialiasing, bounding_box=bounding_box, max_depth=5, seed=1,
                               profile=profile, cluster_culling=cluster_culling, arithmetic=1 if arithmetic == "fast" else 0)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run(arithmetic, cluster_culling, steps, warmup):
        opt = options(arithmetic, cluster_culling, profile=True)
        # only rank 0's image is looked at: the sums are reduced onto it (half an all-reduce's traffic)
        # the marks of a sampling step are recorded on the stream and read after the region (profile = 2)
        opt_marks = options(arithmetic, cluster_culling, profile=2)
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
                renderer.render_async(opt=opt_marks, alone=world == 1 and renderer.streams is not None)
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

    def per_rank_figures(arithmetic):
        """N > 1, outside the timed region: every rank's accumulate of its share alone on its device (min and max
        over the ranks), and the reduce of the sum buffers alone."""
        opt = options(arithmetic)
        r = FrameRenderer(dev, cam, opt, image_on_root_only=True)
        buf = torch.zeros(n_pixels * 4, dtype=torch.float32, device="cuda")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        acc_ms = []
        for _ in range(4):
            fence()
            e0.record()
            if r.s_end > r.s_begin and r.row_end > r.row_begin:
                dev.accumulate(cam, opt, r.s_begin, r.s_end, buf.data_ptr(), torch.cuda.current_stream().cuda_stream,
                               rows=(r.row_begin, r.row_end))
            e1.record()
            torch.cuda.synchronize()
            acc_ms.append(e0.elapsed_time(e1))
        mine = torch.tensor([min(acc_ms[1:])], dtype=torch.float64, device="cuda")
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        acc = [float(g.item()) for g in gathered]
        ex_ms = []
        for _ in range(6):
            fence()
            t0 = time.perf_counter()
            dist.reduce(buf, dst=0)
            torch.cuda.synchronize()
            ex_ms.append((time.perf_counter() - t0) * 1e3)
        ex = torch.tensor([min(ex_ms[1:])], dtype=torch.float64, device="cuda")
        dist.all_reduce(ex, op=dist.ReduceOp.MAX)
        return {"rank_accumulate_ms": {"min": min(acc), "max": max(acc), "per_rank": acc,
                                       "note": "a rank's share of one frame alone on its device, HIP events, best of 3"},
                "exchange_ms": float(ex.item()),
                "exchange_note": f"one reduce of the {n_pixels * 16 / 1e6:.1f} MB sum buffers onto rank 0 alone (host clock around the "
                                 f"call and a device synchronisation, slowest rank, best of 5, backend {backend})"}

    # set-up, not a step: one frame per handle so that the work buffers exist whatever --warmup is
    for handle in [dev, *extra_devs]:
        FrameRenderer(handle, cam, options(args.arithmetic), depth=1).render()
    fence()

    elapsed, meter, opt = run(args.arithmetic, True, steps, args.warmup)
    # the time of ONE frame on its own (nothing else in flight), beside the throughput figure above
    single = FrameRenderer(dev, cam, options(args.arithmetic))
    single.render()
    fence()
    single_frames = max(3, min(steps, 20))
    t0 = time.perf_counter()
    for _ in range(single_frames):
        single.render()
        torch.cuda.synchronize()
    single_ms = (time.perf_counter() - t0) / single_frames * 1e3
    fence()
    other = None
    other_name = "exact" if args.arithmetic == "fast" else "fast"
    if not args.no_other_arithmetic:
        o_steps = max(3, min(steps, 40))
        o_elapsed, o_meter, _ = run(other_name, True, o_steps, 1)
        other = (o_elapsed, o_meter, o_steps)
    exhaustive = None
    if has_mesh and args.config == "c3" and not args.no_exhaustive:
        ex_steps = max(1, min(steps, 5))
        ex_elapsed, ex_meter, _ = run(args.arithmetic, False, ex_steps, 1)
        exhaustive = (ex_elapsed, ex_meter, ex_steps)
    ranks = per_rank_figures(args.arithmetic) if world > 1 else None

    if rank == 0:
        ms_per_step = elapsed / steps * 1e3
        value = n_pixels * spp / (elapsed / steps) / 1e6
        static = {}
        cpath = os.path.join(ROOT, "profiles", "kernel_counters.json")
        if os.path.exists(cpath):
            with open(cpath) as f:
                static = json.load(f).get(args.config, {})
        fast_text = ("fused multiply-adds, hardware reciprocal / square-root / log / sin / cos instructions (rt_options.arithmetic = 1): per "
                     "pixel within 1e-4 relative of the bit-exact mode in the modes without random numbers (>= 99.9 % of the pixels, the "
                     "rest are edge decisions), converged means equal to within 0.01 of an 8-bit level, the reference's seven renders "
                     "reproduced within the exact mode's tolerances (tests/test_gpu_fast_arithmetic.py)")
        exact_text = "the reference's float arithmetic operation by operation (rt_options.arithmetic = 0): image bit-identical to the CPU oracle"
        line = {
            "metric": f"Mpaths/s, {cfg['label']}",
            "value": value,
            "unit": "Mpaths/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "s_per_frame": ms_per_step / 1e3,  # inverse throughput with frames in flight
            "s_per_frame_single": single_ms / 1e3,  # one frame rendered alone, submit to finished (frames_in_flight = 1)
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": value / BASELINE_MPATHS if args.config == "c3" else None,
            "dtype": "f32",
            "data": "the reference's own scene files and cat.obj mesh (byte-identical copies under assets/), fixed seed; nothing synthetic",
            "config": {
                "workload": f"{args.config}: {cfg['label']} ({cfg['scene']}.json {cam.width}x{cam.height} {spp}spp depth5)",
                "triangles": int(desc.n_triangles),
                "mode": "cluster culling on (library default; image bit-identical to the exhaustive scan); arithmetic: "
                        + (fast_text if args.arithmetic == "fast" else exact_text),
                "arithmetic": args.arithmetic,
                "parallelism": f"spp-split x{world} + RCCL reduce of linear radiance onto rank 0" if world > 1 else "single GPU",
                "frames_in_flight": in_flight,
            },
        }
        begin, end = sample_range(0, world, spp)  # rank 0's share (the statistics are rank 0's)
        own_samples = end - begin
        # `roofline` names the kernel with the largest share of the step.  Durations are HIP events on the launch stream
        # around rank 0's launches on the sampling steps; algorithmic work per test as above; hardware counters are static.
        suffix = "_fast" if args.arithmetic == "fast" else ""
        if has_mesh:
            shade = meter.shade_roofline("k_shade" + suffix, ms_per_step, int(desc.n_spheres), int(desc.n_boxes), n_pixels * own_samples, static, args.arithmetic == "fast")
            mesh = meter.mesh_roofline("k_group_clusters", ms_per_step, static)
            if shade is not None and shade["kernel_share_of_step"] >= (mesh["kernel_share_of_step"] or 0):
                line["roofline"], line["roofline_k_group_clusters"] = shade, mesh
            else:
                line["roofline"], line["roofline_k_shade"] = mesh, shade
        else:
            line["roofline"] = meter.spheres_roofline("k_trace_spheres" + suffix, ms_per_step, int(desc.n_spheres), static)
        line["roofline_whole_step"] = meter.frame_roofline(ms_per_step, int(desc.n_spheres), int(desc.n_boxes), world)
        line["kernel_durations"] = ("HIP events on the launch stream around the kernels of three steps of the timed region "
                                    + ("that ran alone on the device" if world == 1 else
                                       "in flight beside the other frames (stretched by what ran next to them)")
                                    + ", read after the region")
        if other is not None:
            o_elapsed, o_meter, o_steps = other
            line[other_name + "_arithmetic"] = {
                "note": exact_text if other_name == "exact" else fast_text,
                "steps": o_steps,
                "ms_per_step": o_elapsed / o_steps * 1e3,
                "value": n_pixels * spp / (o_elapsed / o_steps) / 1e6,
                "unit": "Mpaths/s",
                "vs_baseline": n_pixels * spp / (o_elapsed / o_steps) / 1e6 / BASELINE_MPATHS if args.config == "c3" else None,
            }
        if exhaustive is not None:
            ex_elapsed, ex_meter, ex_steps = exhaustive
            line["exhaustive_scan"] = {
                "note": "cluster_culling=0: every queued ray against every triangle, the reference's own work "
                        "(7 175 triangle tests per path)",
                "steps": ex_steps,
                "ms_per_step": ex_elapsed / ex_steps * 1e3,
                "value": n_pixels * spp / (ex_elapsed / ex_steps) / 1e6,
                "unit": "Mpaths/s",
                "roofline": ex_meter.mesh_roofline("k_mesh", ex_elapsed / ex_steps * 1e3, static),
            }
        if ranks is not None:
            line.update(ranks)
            line["frames_in_flight"] = in_flight
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(desc, cam, options("exact"))
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
