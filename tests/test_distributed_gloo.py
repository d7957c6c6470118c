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

from raytracer_amd.distributed import FrameRenderer, frame_partition, sample_range  # noqa: E402


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


def test_frame_partition_covers_every_sample_of_every_row_once():
    for spp in (1, 2, 3, 4, 32):
        for world in (1, 2, 3, 4, 6, 8):
            for height in (1, 5, 16):
                seen = np.zeros((height, spp), np.int32)
                for r in range(world):
                    sb, se, rb, re = frame_partition(r, world, spp, height)
                    seen[rb:re, sb:se] += 1
                assert (seen == 1).all(), (spp, world, height)
    # more samples than ranks: no bands
    assert [frame_partition(r, 4, 32, 512) for r in range(4)] == [(8 * r, 8 * r + 8, 0, 512) for r in range(4)]
    # 8 ranks, 4 samples: 4 sample groups x 2 bands
    assert frame_partition(5, 8, 4, 512) == (1, 2, 256, 512)
    with pytest.raises(ValueError):
        frame_partition(3, 3, 4, 16)


def _host_callbacks(desc, n_pixels):
    import oracle_binding as ob

    def accumulate(cam, opt, s_begin, s_end, sum_tensor, rows):
        pixels = np.arange(rows[0] * cam.width, rows[1] * cam.width, dtype=np.uint32)
        _, ssum, count, _ = ob.render_pixels(desc, cam, ob.from_rt_options(opt), pixels, s_begin, s_end, threads=2)
        buf = sum_tensor.view(n_pixels, 4).numpy()
        buf[pixels, :3] += ssum
        buf[pixels, 3] += count

    def resolve(sum_tensor, out_tensor):
        buf = sum_tensor.view(n_pixels, 4).numpy()
        mean = (np.float32(1) / np.abs(buf[:, 3:4])) * buf[:, :3]  # (the HIP library's sums mark early-ended pixels by a negative count)
        out_tensor.view(n_pixels, 3).numpy()[:] = ob.math_probe("gamma", mean.ravel()).reshape(-1, 3)

    return accumulate, resolve


def _worker(rank, world, port, spp, out_dir, root_only=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import raytracer_amd as rt
        from helpers import load

        s = load("default", 24, 16, spp)
        desc, cam = s.describe()
        opt = rt.make_options(seed=21)
        acc, res = _host_callbacks(desc, cam.width * cam.height)
        r = FrameRenderer(None, cam, opt, tensor_device="cpu", accumulate=acc, resolve=res, image_on_root_only=root_only)
        assert (r.rank, r.world) == (rank, world)
        first = r.render().clone()
        second = r.render()  # buffers are reused: a second frame must not accumulate onto the first
        if not root_only or rank == 0:
            assert torch.equal(first, second)
        np.save(os.path.join(out_dir, f"img{rank}.npy"), first.numpy())
        np.save(os.path.join(out_dir, f"range{rank}.npy"), np.array([r.s_begin, r.s_end]))
        np.save(os.path.join(out_dir, f"rows{rank}.npy"), np.array([r.row_begin, r.row_end]))
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
    rows = [tuple(np.load(tmp_path / f"rows{r}.npy")) for r in range(world)]
    assert [ranges[r] + rows[r] for r in range(world)] == [frame_partition(r, world, spp, 16) for r in range(world)]
    assert np.array_equal(imgs[0], imgs[1])  # every rank ends with the same image
    # the split changes only the order of a few fp32 additions
    np.testing.assert_allclose(imgs[0], ref, rtol=2e-6, atol=1e-4)
    if spp == 1:  # one sample: the ranks took the two halves of the rows, every pixel's sum is one rank's
        assert rows == [(0, 8), (8, 16)]
        assert np.array_equal(imgs[0].view(np.uint32), ref.view(np.uint32))



def test_three_ranks_two_samples_use_row_bands(tmp_path):
    """More ranks than samples per pixel: 1 sample group x 3 bands of rows (3 has no divisor <= 2 but 1)."""
    world, spp = 3, 2
    mp.spawn(_worker, args=(world, _free_port(), spp, str(tmp_path)), nprocs=world, join=True)

    import oracle_binding as ob
    import raytracer_amd as rt
    from helpers import load

    s = load("default", 24, 16, spp)
    desc, cam = s.describe()
    ref, _ = ob.render_image(desc, cam, ob.from_rt_options(rt.make_options(seed=21)))
    rows = [tuple(np.load(tmp_path / f"rows{r}.npy")) for r in range(world)]
    assert rows == [(0, 6), (6, 11), (11, 16)]
    imgs = [np.load(tmp_path / f"img{r}.npy") for r in range(world)]
    assert all(np.array_equal(imgs[0], i) for i in imgs[1:])
    # every pixel's samples were added by one rank, in order: bit-identical to the single-rank frame
    assert np.array_equal(imgs[0].view(np.uint32), ref.view(np.uint32))


def test_sums_reduced_onto_the_first_rank_only(tmp_path):
    """image_on_root_only: a reduce instead of an all-reduce; the first rank's image is the all-reduce one's,
    the other ranks resolve nothing."""
    world, spp = 2, 5
    mp.spawn(_worker, args=(world, _free_port(), spp, str(tmp_path), True), nprocs=world, join=True)
    root = np.load(tmp_path / "img0.npy")
    other_dir = tmp_path / "all"
    other_dir.mkdir()
    mp.spawn(_worker, args=(world, _free_port(), spp, str(other_dir)), nprocs=world, join=True)
    assert np.array_equal(root.view(np.uint32), np.load(other_dir / "img0.npy").view(np.uint32))

