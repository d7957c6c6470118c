"""Multi-GPU frame driver: one process per GPU, samples split over ranks.

The reference renders on one GPU.  Its sample loop (reference
src/raytrace.cu:478-500) is embarrassingly parallel once every sample owns its
random stream, and its only coupling is the per-pixel mean followed by the
non-linear gamma curve.  So every rank renders a contiguous range of the
GLOBAL sample indices of every pixel into a linear (r, g, b, actual_rays)
buffer, the buffers are summed with ONE all-reduce (RCCL over xGMI when the
process group is "nccl"), and the mean + gamma run once on the summed buffer.
The image does not depend on the number of ranks except for the order of the
fp32 additions across rank boundaries.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple


def sample_range(rank: int, world: int, spp: int) -> Tuple[int, int]:
    """Contiguous share [begin, end) of samples 0..spp-1 for `rank` of `world`.

    Shares differ by at most one sample; ranks beyond spp get an empty range.
    """
    if world <= 0 or not 0 <= rank < world or spp < 0:
        raise ValueError(f"bad partition request rank={rank} world={world} spp={spp}")
    base, extra = divmod(spp, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def frame_partition(rank: int, world: int, spp: int, height: int) -> Tuple[int, int, int, int]:
    """(sample_begin, sample_end, row_begin, row_end) of one frame for `rank` of `world`.

    Samples are split first (perfectly balanced, every rank sees all pixels).  Only when there are
    more ranks than samples per pixel is the image also cut into bands of rows: the ranks form
    S sample groups x B bands with S the largest divisor of `world` not above spp.
    """
    if world <= 0 or not 0 <= rank < world or spp < 0 or height < 0:
        raise ValueError(f"bad partition request rank={rank} world={world} spp={spp} height={height}")
    groups = max(d for d in range(1, world + 1) if world % d == 0 and d <= max(spp, 1))
    bands = world // groups
    s_begin, s_end = sample_range(rank % groups, groups, spp)
    r_begin, r_end = sample_range(rank // groups, bands, height)
    return s_begin, s_end, r_begin, r_end


def frame_share(rank: int, world: int) -> Tuple[int, int]:
    """(first, stride) for Scene.render_frames: rank r of N draws frames r, r + N, r + 2N, ...

    A frame sequence shards over whole frames (every frame has its own seed and camera pose), so
    there is no exchange step at all: each rank writes its own files.
    """
    if world <= 0 or not 0 <= rank < world:
        raise ValueError(f"bad partition request rank={rank} world={world}")
    return rank, world


class _Slot:
    """Buffers of one frame in flight."""

    def __init__(self, torch, n_pixels, device):
        self.sum = torch.zeros(n_pixels * 4, dtype=torch.float32, device=device)
        self.image = torch.empty(n_pixels * 3, dtype=torch.float32, device=device)
        self.done = None  # event: the frame that last used these buffers has been resolved


class FrameRenderer:
    """Renders whole frames of one scene with the samples split over a process group.

    accumulate(cam, opt, s_begin, s_end, sum_tensor, rows) must ADD the linear
    colour sums and sample counts of the sample range, for the image rows
    rows = (begin, end), into sum_tensor (4 floats per pixel of the whole
    image); resolve(sum_tensor, out_tensor) must write gamma(mean).  The
    fourth float of a pixel is its sample count, and its SIGN carries meaning in
    sums the HIP library wrote: a negative count marks a pixel whose sample loop
    the reference ends after the first sample (direct light without
    anti-aliasing, a diffuse first hit), so a resolve that divides must divide
    by abs(count).  The
    defaults call the HIP library on the current CUDA stream; the CPU test
    suite injects host implementations to exercise the partition + reduction
    logic under the gloo backend.

    render() returns a finished-in-stream-order image.  render_async() keeps
    frames in flight; call wait() (or synchronize the device) before reading the
    images it returned.  Two arrangements:

    * one scene handle (`dev`): with the "nccl" backend the all-reduce and the
      resolve of frame i run on RCCL's stream and a side stream while the
      current stream already accumulates frame i + 1, so a sequence of frames
      costs max(compute, exchange) per frame instead of their sum;
    * several handles of the same scene (`extra_devs`: each has its own work
      buffers): frame i runs entirely -- accumulate, exchange, resolve -- on
      stream i mod n with handle i mod n, so the kernels of consecutive frames
      overlap on the device.  Short frames (a rank's share at 8 GPUs is about a
      million paths) are bound by the start-up and drain of their ~20 kernels;
      a second frame in flight fills those gaps.
    """

    def __init__(self, dev, cam, opt, *, group=None, tensor_device="cuda",
                 accumulate: Optional[Callable] = None, resolve: Optional[Callable] = None,
                 want_stats: bool = False, depth: int = 2, exchange_in_group_of_one: bool = False,
                 extra_devs=(), image_on_root_only: bool = False):
        import torch
        import torch.distributed as dist

        self.torch = torch
        self.dist = dist
        self.dev, self.cam, self.opt = dev, cam, opt
        self.group = group
        self.distributed = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if self.distributed else 0
        self.world = dist.get_world_size(group) if self.distributed else 1
        self.n_pixels = cam.width * cam.height
        self.s_begin, self.s_end, self.row_begin, self.row_end = frame_partition(self.rank, self.world, cam.num_rays, cam.height)
        if depth < 1:
            raise ValueError("depth must be at least 1")
        self.on_gpu = str(tensor_device).startswith("cuda")
        self.devs = [dev, *extra_devs]
        if len(self.devs) > 1 and not self.on_gpu:
            raise ValueError("several scene handles need device tensors")
        # one stream per handle; a slot is only ever used with the handle and stream of its index
        self.streams = [torch.cuda.Stream() for _ in self.devs] if len(self.devs) > 1 else None
        self.slots = [_Slot(torch, self.n_pixels, tensor_device) for _ in range(len(self.devs) if self.streams else depth)]
        self.frame = 0
        # overlap needs a collective that runs on a device stream of its own
        self.overlap = bool(self.on_gpu and self.distributed and dist.get_backend(group) == "nccl")
        self.side = torch.cuda.Stream() if self.overlap and not self.streams else None
        # a group of one has nothing to exchange; rehearsals of the stream logic can ask for the collective anyway
        self.exchange = self.world > 1 or (exchange_in_group_of_one and self.distributed)
        # image_on_root_only: the sums are reduced onto the group's first rank, which alone resolves them
        # (half the traffic of an all-reduce); the other ranks' image tensors are then not written
        self.root_only = bool(image_on_root_only and self.distributed)
        self.root = dist.get_global_rank(group, 0) if (self.root_only and group is not None) else 0
        self.want_stats = want_stats
        self.last_stats = None
        self._accumulate = accumulate or self._hip_accumulate
        self._resolve = resolve or self._hip_resolve
        self._active_dev = dev
        self.last_dev = dev  # the scene handle the most recent frame was rendered with

    # the buffers of the most recent frame (kept for callers that look at them)
    @property
    def sum(self):
        return self.slots[(self.frame - 1) % len(self.slots)].sum

    @property
    def image(self):
        return self.slots[(self.frame - 1) % len(self.slots)].image

    def _stream(self) -> int:
        return self.torch.cuda.current_stream().cuda_stream

    def _hip_accumulate(self, cam, opt, s_begin, s_end, sum_tensor, rows):
        self.last_stats = self._active_dev.accumulate(cam, opt, s_begin, s_end, sum_tensor.data_ptr(), self._stream(),
                                                      want_stats=self.want_stats, rows=rows)

    def _hip_resolve(self, sum_tensor, out_tensor):
        import raytracer_amd as rt

        rt.resolve(sum_tensor.data_ptr(), self.n_pixels, out_tensor.data_ptr(), self._stream())

    def render_async(self, cam=None, opt=None, alone=False):
        """Enqueues one frame; returns its (H, W, 3) image tensor, valid after wait().

        alone (several handles only): the frame runs with nothing beside it on the device -- its stream waits for
        the frames in flight, the streams of the later frames wait for it -- without the host waiting for anything.
        For timing a frame's kernels undisturbed (opt with profile = 2, then last_dev.profile_collect())."""
        torch = self.torch
        cam = cam if cam is not None else self.cam
        opt = opt if opt is not None else self.opt
        k = self.frame % len(self.slots)
        slot = self.slots[k]
        self.frame += 1
        if self.streams:
            stream = self.streams[k]
            stream.wait_stream(torch.cuda.current_stream())  # after whatever the caller queued before
            others = [x for x in self.streams if x is not stream] if alone else []
            for other in others:
                stream.wait_stream(other)
            self._active_dev = self.last_dev = self.devs[k]
            with torch.cuda.stream(stream):
                self._whole_frame_on_current_stream(slot, cam, opt)
            self._active_dev = self.dev
            for other in others:
                other.wait_stream(stream)
            return slot.image.view(cam.height, cam.width, 3)
        self.last_dev = self.dev
        if slot.done is not None:  # the frame that used these buffers `depth` frames ago
            torch.cuda.current_stream().wait_event(slot.done)
            slot.done = None
        slot.sum.zero_()
        if self.s_end > self.s_begin and self.row_end > self.row_begin:
            self._accumulate(cam, opt, self.s_begin, self.s_end, slot.sum, (self.row_begin, self.row_end))
        if self.exchange and self.overlap:
            # the collective is ordered after the accumulate; the current stream does not wait for it
            work = self._sum_over_ranks(slot.sum, async_op=True)
            with torch.cuda.stream(self.side):
                work.wait()  # makes the side stream wait for the collective
                self._resolve_where_wanted(slot)
                slot.done = self.side.record_event()
        else:
            self._exchange_blocking(slot)
            self._resolve_where_wanted(slot)
        return slot.image.view(cam.height, cam.width, 3)

    def _sum_over_ranks(self, tensor, async_op=False):
        if self.root_only:
            return self.dist.reduce(tensor, dst=self.root, group=self.group, async_op=async_op)
        return self.dist.all_reduce(tensor, group=self.group, async_op=async_op)

    def _resolve_where_wanted(self, slot) -> None:
        if not self.root_only or self.rank == 0:
            self._resolve(slot.sum, slot.image)

    def _exchange_blocking(self, slot) -> None:
        if self.exchange:
            # host-staged backends (gloo) copy the tensor themselves: hand it over finished.
            # (Seen on one GPU shared by two gloo ranks: with the accumulate still in flight
            # the collective stalled for seconds.)
            if self.on_gpu:
                self.torch.cuda.current_stream().synchronize()
            self._sum_over_ranks(slot.sum)

    def _whole_frame_on_current_stream(self, slot, cam, opt) -> None:
        """Stream-ordered after the previous frame of this slot: same stream."""
        slot.sum.zero_()
        if self.s_end > self.s_begin and self.row_end > self.row_begin:
            self._accumulate(cam, opt, self.s_begin, self.s_end, slot.sum, (self.row_begin, self.row_end))
        if self.exchange and self.overlap:
            # RCCL's stream waits for this one; wait() then makes this stream (only) wait for RCCL
            self._sum_over_ranks(slot.sum, async_op=True).wait()
        else:
            self._exchange_blocking(slot)
        self._resolve_where_wanted(slot)

    def wait(self) -> None:
        """Orders the current stream after every frame enqueued so far."""
        current = self.torch.cuda.current_stream() if self.on_gpu else None
        if self.streams:
            for stream in self.streams:
                current.wait_stream(stream)
        elif self.overlap:
            current.wait_stream(self.side)

    def render(self, cam=None, opt=None):
        """One frame; returns the (H, W, 3) image tensor, ordered on the current stream."""
        image = self.render_async(cam, opt)
        self.wait()
        return image
