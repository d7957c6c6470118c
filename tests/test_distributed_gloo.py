"""The multi-GPU frame driver (raytracer_amd.distributed) under the gloo backend
with two CPU ranks.  The HIP calls are replaced by host implementations built
on the oracle, so what is exercised here is the partition of the sample range,
the all-reduce of the linear (r, g, b, count) buffers and the single resolve.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from raytracer_amd.distributed import FrameRenderer, sample_range  # noqa: E402


def test_sample_range_partitions_exactly():
    for spp in (0, 1, 5, 32, 60, 1000):
        for world in (1, 2, 3, 4, 7, 8, 16):
            covered = []
            sizes = []
            for r in range(world):
                b, e = sample_range(r, world, spp)
                assert 0 <= b <= e <= spp
                covered += list(range(b, e))
                sizes.append(e - b)
            assert covered == list(range(spp))
            assert max(sizes) - min(sizes) <= 1
    assert [sample_range(r, 8, 60) for r in range(8)] == [(0, 8), (8, 16), (16, 24), (24, 32), (32, 39), (39, 46), (46, 53), (53, 60)]
    with pytest.raises(ValueError):
        sample_range(2, 2, 8)


def _host_callbacks(desc, n_pixels):
    import oracle_binding as ob

    pixels = np.arange(n_pixels, dtype=np.uint32)

    def accumulate(cam, opt, s_begin, s_end, sum_tensor):
        _, ssum, count, _ = ob.render_pixels(desc, cam, ob.from_rt_options(opt), pixels, s_begin, s_end, threads=2)
        buf = sum_tensor.view(n_pixels, 4).numpy()
        buf[:, :3] += ssum
        buf[:, 3] += count

    def resolve(sum_tensor, out_tensor):
        buf = sum_tensor.view(n_pixels, 4).numpy()
        mean = (np.float32(1) / buf[:, 3:4]) * buf[:, :3]
        out_tensor.view(n_pixels, 3).numpy()[:] = ob.math_probe("gamma", mean.ravel()).reshape(-1, 3)

    return accumulate, resolve


def _worker(rank, world, port, spp, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import raytracer_amd as rt
        from helpers import load

        s = load("default", 24, 16, spp)
        desc, cam = s.describe()
        opt = rt.make_options(seed=21)
        acc, res = _host_callbacks(desc, cam.width * cam.height)
        r = FrameRenderer(None, cam, opt, tensor_device="cpu", accumulate=acc, resolve=res)
        assert (r.rank, r.world) == (rank, world)
        first = r.render().clone()
        second = r.render()  # buffers are reused: a second frame must not accumulate onto the first
        assert torch.equal(first, second)
        np.save(os.path.join(out_dir, f"img{rank}.npy"), first.numpy())
        np.save(os.path.join(out_dir, f"range{rank}.npy"), np.array([r.s_begin, r.s_end]))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("spp", [5, 1])
def test_two_ranks_reproduce_the_single_rank_frame(tmp_path, spp):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), spp, str(tmp_path)), nprocs=world, join=True)

    import oracle_binding as ob
    import raytracer_amd as rt
    from helpers import load

    s = load("default", 24, 16, spp)
    desc, cam = s.describe()
    ref, _ = ob.render_image(desc, cam, ob.from_rt_options(rt.make_options(seed=21)))
    imgs = [np.load(tmp_path / f"img{r}.npy") for r in range(world)]
    ranges = [tuple(np.load(tmp_path / f"range{r}.npy")) for r in range(world)]
    assert ranges == [sample_range(r, world, spp) for r in range(world)]
    assert np.array_equal(imgs[0], imgs[1])  # every rank ends with the same image
    # the split changes only the order of a few fp32 additions
    np.testing.assert_allclose(imgs[0], ref, rtol=2e-6, atol=1e-4)
    if spp == 1:  # rank 1 had nothing to render: the sum is rank 0's alone, bit for bit
        assert np.array_equal(imgs[0].view(np.uint32), ref.view(np.uint32))
