"""rt_options.arithmetic = 1: fused multiply-adds and the hardware's reciprocal, square-root and
log / sin / cos instructions in the path kernels, judged by the tolerances the task states for this path --
per pixel within 1e-4 relative where no random number reaches a pixel, converged means within a stated RMSE
where they do -- against the exact arithmetic (rt_options.arithmetic = 0), which the rest of the suite shows
to equal the CPU oracle bit for bit, and against the oracle itself and the reference's own renders.
"""
import os

import numpy as np
import pytest
from PIL import Image

import oracle_binding as ob
import raytracer_amd as rt
from helpers import load, variant
from test_gpu_configs import GOLDEN, SHIPPED, _blur

pytestmark = pytest.mark.gpu

REL_TOL = 1e-4   # per channel, relative to the exact value ...
ABS_FLOOR = 1e-2  # ... or to this, on the 0..255 scale of the output (1/25 600 of full scale)


def render(scene, **options):
    desc, cam = scene.describe()
    img, _ = scene.upload().render(cam, rt.make_options(**options))
    return img


def pixels_outside(fast, exact):
    """Pixels with a channel off by more than REL_TOL (NaN and inf count as equal when both sides have them)."""
    both_odd = ~np.isfinite(fast) & ~np.isfinite(exact)
    err = np.abs(fast.astype(np.float64) - exact.astype(np.float64)) / np.maximum(np.abs(exact.astype(np.float64)), ABS_FLOOR)
    err = np.where(both_odd, 0.0, np.nan_to_num(err, nan=np.inf))
    return err.max(axis=2) > REL_TOL, err.max(axis=2)


@pytest.mark.parametrize("bbox", [True, False], ids=["bbox", "no_bbox"])
@pytest.mark.parametrize("scene", ["default", "simple", "mesh_scene", "mirror_cat_depth7"])
def test_deterministic_modes_agree_per_pixel_within_1e4(scene, bbox):
    """Direct light without anti-aliasing at the shipped 512 x 512: a pixel's value does not depend on a random
    number (glass aside, whose reflect-or-refract draw is the same in both arithmetics).  At least 99.9 % of the
    pixels agree within 1e-4 relative; the others are hit-or-miss decisions that flip at an edge, and are listed."""
    s = variant("mesh_scene", "mirror") if scene == "mirror_cat_depth7" else load(scene)
    depth = 7 if scene == "mirror_cat_depth7" else 5
    opts = dict(indirect=False, antialiasing=False, bounding_box=bbox, max_depth=depth, seed=1)
    exact = render(s, arithmetic=0, **opts)
    fast = render(s, arithmetic=1, **opts)
    out, err = pixels_outside(fast, exact)
    inside = err[~out]
    ys, xs = np.nonzero(out)
    print(f"{scene} bbox={bbox}: {out.sum()} of {out.size} pixels outside 1e-4 ({out.mean():.5%}); "
          f"median relative error of the others {np.median(inside):.2e}, largest {inside.max():.2e}")
    for x, y in list(zip(xs, ys))[:12]:
        print(f"  ({x}, {y}): exact {exact[y, x].tolist()}, fast {fast[y, x].tolist()}")
    assert out.mean() <= 1e-3
    assert np.median(inside) < 1e-5


@pytest.mark.parametrize("depth", [0, 1, 2])
@pytest.mark.parametrize("scene", ["mesh_scene", "default"])
def test_shallow_frames_agree_per_pixel(scene, depth):
    """Bounce limits of 0, 1 and 2: with the fast arithmetic the first bounce computes the camera rays again instead
    of reading them back, which a frame without bounces cannot do (its camera kernel writes the path state itself)."""
    s = load(scene, 160, 120, 3)
    opts = dict(indirect=True, antialiasing=False, bounding_box=True, max_depth=depth, seed=4)
    exact = render(s, arithmetic=0, **opts)
    fast = render(s, arithmetic=1, **opts)
    if depth == 0:
        assert not exact.any() and not fast.any()  # no bounce, no light (reference src/raytrace.cu:317: the loop does not run)
        return
    out, _ = pixels_outside(fast, exact)
    # indirect light: a pixel whose path flips a decision draws another direction, so a few more pixels differ than in the direct modes
    assert out.mean() <= 5e-3, f"{out.sum()} of {out.size} pixels outside 1e-4"


CONVERGED = [
    # scene, options, RMSE bound and mean bound in 8-bit levels of the clamped image
    ("default", lambda: load("default", 64, 64, 4096), dict(), 0.25, 0.03),
    ("cat", lambda: load("mesh_scene", 64, 64, 4096), dict(), 0.25, 0.03),
    ("glass_cat", lambda: variant("mesh_scene", "glass", 64, 64, 4096), dict(max_depth=7), 0.35, 0.03),
    ("default_direct_aa", lambda: load("default", 64, 64, 4096), dict(indirect=False), 0.25, 0.03),
]


@pytest.mark.parametrize("name,make,options,rmse_tol,mean_tol", CONVERGED, ids=[c[0] for c in CONVERGED])
def test_converged_means_agree(name, make, options, rmse_tol, mean_tol):
    """64 x 64 at 4 096 samples per pixel with random numbers in play (anti-aliasing, indirect light, glass): the
    two arithmetics draw the same numbers, so their paths part only where a decision flips.  Stated and asserted:
    the per-pixel RMSE and the per-channel mean difference of the clamped images, in 8-bit levels."""
    s = make()
    exact = np.minimum(render(s, arithmetic=0, seed=3, **options), 255.0)
    fast = np.minimum(render(s, arithmetic=1, seed=3, **options), 255.0)
    rmse = float(np.sqrt(np.mean((fast.astype(np.float64) - exact) ** 2)))
    mean = np.abs(fast.reshape(-1, 3).astype(np.float64).mean(0) - exact.reshape(-1, 3).astype(np.float64).mean(0))
    print(f"{name}: RMSE {rmse:.4f} levels, mean difference {mean} levels")
    assert rmse < rmse_tol
    assert np.all(mean < mean_tol)


def test_converged_mean_against_the_oracle_itself():
    """The same statement against the CPU oracle (not against the exact mode on the GPU): default.json, 64 x 64,
    1 024 samples per pixel, indirect light and anti-aliasing."""
    s = load("default", 64, 64, 1024)
    desc, cam = s.describe()
    opt = rt.make_options(seed=5, arithmetic=1)
    fast = np.minimum(render(s, seed=5, arithmetic=1), 255.0)
    ref, _ = ob.render_image(desc, cam, ob.from_rt_options(opt))
    ref = np.minimum(ref, 255.0)
    rmse = float(np.sqrt(np.mean((fast.astype(np.float64) - ref) ** 2)))
    mean = np.abs(fast.reshape(-1, 3).astype(np.float64).mean(0) - ref.reshape(-1, 3).astype(np.float64).mean(0))
    print(f"fast arithmetic vs oracle: RMSE {rmse:.4f} levels, mean difference {mean} levels")
    assert rmse < 0.5 and np.all(mean < 0.05)


@pytest.mark.parametrize("name,make,options,mean_tol,rmse_tol", SHIPPED, ids=[e[0][:-4] for e in SHIPPED])
def test_fast_arithmetic_matches_the_references_shipped_render(name, make, options, mean_tol, rmse_tol):
    """The seven renders the reference ships, with the tolerances the exact mode is held to."""
    s = make()
    img = render(s, seed=1, arithmetic=1, **options)
    mine = np.floor(np.minimum(img, 255))
    ref = np.asarray(Image.open(os.path.join(GOLDEN, "reference_renders", name)).convert("RGB"), np.float32)
    mean_diff = np.abs(mine.reshape(-1, 3).astype(np.float64).mean(0) - ref.reshape(-1, 3).astype(np.float64).mean(0))
    rmse = float(np.sqrt(np.mean((_blur(mine) - _blur(ref)) ** 2)))
    print(f"{name}: mean diff {mean_diff}, blurred rmse {rmse:.3f}")
    assert np.all(mean_diff < mean_tol) and rmse < rmse_tol


def test_fast_arithmetic_matches_the_references_mirror_cat_pixel_by_pixel():
    """The reference's one deterministic render at its 8-bit resolution, as for the exact mode."""
    s = variant("mesh_scene", "mirror")
    img = render(s, indirect=False, antialiasing=False, max_depth=7, seed=1, arithmetic=1)
    mine = np.floor(np.minimum(img, 255))
    ref = np.asarray(Image.open(os.path.join(GOLDEN, "reference_renders", "cat_mirror_512x512_32rays.png")).convert("RGB"), np.float32)
    level = np.abs(mine - ref).max(axis=2)
    exact, within1, off = float((level == 0).mean()), float((level <= 1).mean()), int((level > 1).sum())
    print(f"{exact:.6f} of the pixels equal, {within1:.6f} within one level, {off} off by more")
    assert exact >= 0.9999 and within1 >= 0.99995 and off <= 8


def test_the_two_routes_of_the_fast_arithmetic_give_the_same_bits():
    """Batches of 1.5 M paths and more keep paths in their lanes from bounce to bounce (k_shade_fast); smaller ones take
    a launch per bounce, the first of which computes the camera rays again.  Same arithmetic per path in the same order:
    a frame rendered as one batch and as sample ranges too small for the first route must give the same sums bit for bit."""
    from hip_mem import DeviceBuffer

    s = load("mesh_scene", 512, 512, 8)  # 2.1 M paths in one batch
    desc, cam = s.describe()
    dev = s.upload()
    n = cam.width * cam.height
    for options in (dict(seed=4), dict(seed=4, max_depth=7), dict(seed=4, antialiasing=False)):
        opt = rt.make_options(arithmetic=1, **options)
        whole, parts = DeviceBuffer(4 * n), DeviceBuffer(4 * n)
        st = dev.accumulate(cam, opt, 0, 8, whole.address, want_stats=True)
        assert st.path_records < st.path_bounces or options.get("max_depth", 5) < 2  # (the batch did take the first route)
        for b, e in [(0, 3), (3, 4), (4, 8)]:  # 0.8 M, 0.3 M, 1.0 M paths
            dev.accumulate(cam, opt, b, e, parts.address)
        a, p = whole.to_host(), parts.to_host()
        assert np.array_equal(a.view(np.uint32), p.view(np.uint32)), options
    # and the statistics say which route a frame took
    _, big = dev.render(cam, rt.make_options(arithmetic=1, seed=4))
    small = load("mesh_scene", 256, 256, 8)
    _, few = small.upload().render(small.describe()[1], rt.make_options(arithmetic=1, seed=4))
    assert big.path_records < 0.7 * big.path_bounces and few.path_records == few.path_bounces


def test_unknown_arithmetic_is_rejected():
    s = load("simple", 16, 16, 1)
    desc, cam = s.describe()
    with pytest.raises(rt.RtError):
        s.upload().render(cam, rt.make_options(arithmetic=2))
