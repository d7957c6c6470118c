"""The counter-based generator and the fixed polynomial kernels (rt_math.h), CPU only."""
import json
import math
import os

import numpy as np

import oracle_binding as ob

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def mix32(x):
    x = np.asarray(x, dtype=np.uint64) & 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x846CA68B) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def np_uniform(seed, pixel, sample, dim):
    """Independent numpy statement of rt_rng_make_key / rt_rng_uniform."""
    a = mix32(np.uint64(seed ^ 0x3C6EF372))
    kp = mix32(a ^ np.asarray(pixel, np.uint64))
    ks = mix32(((a + 0x9E3779B9) & 0xFFFFFFFF) ^ np.asarray(sample, np.uint64))
    w = mix32(((kp ^ ((np.asarray(dim, np.uint64) * 0x9E3779B9) & 0xFFFFFFFF)) + ks) & 0xFFFFFFFF)
    return ((w >> 8) + 1).astype(np.float64) * 2.0 ** -24


def test_uniform_matches_independent_statement():
    for seed, pixel, sample in [(1, 0, 0), (5, 12345, 31), (0xFFFFFFFF, 262143, 999)]:
        got = ob.uniforms(seed, pixel, sample, 3, 257)
        want = np_uniform(seed, pixel, sample, np.arange(3, 260))
        assert np.array_equal(got.astype(np.float64), want)


def np_words(seed, pixel, sample, dim):
    a = mix32(np.uint64(seed ^ 0x3C6EF372))
    kp = mix32(a ^ np.asarray(pixel, np.uint64))
    ks = mix32(((a + 0x9E3779B9) & 0xFFFFFFFF) ^ np.asarray(sample, np.uint64))
    return mix32(((kp ^ ((np.asarray(dim, np.uint64) * 0x9E3779B9) & 0xFFFFFFFF)) + ks) & 0xFFFFFFFF)


def test_no_two_pixels_draw_shifted_copies_of_a_stream():
    """With the dimension and the pixel key combined by the same operation, pixels whose keys differ by j times
    the dimension stride would draw the same numbers j dimensions apart (about two thousand such pairs per
    sample index in a 512 x 512 frame within 60 dimensions).  Here: among all 262 144 pixels of a frame, no
    pair shares two consecutive words at any shift of up to 24 dimensions."""
    pixels = np.arange(512 * 512, dtype=np.uint64)
    base = (np_words(1, pixels, 0, 0) << np.uint64(32)) | np_words(1, pixels, 0, 1)
    for shift in range(1, 25):
        other = (np_words(1, pixels, 0, shift) << np.uint64(32)) | np_words(1, pixels, 0, shift + 1)
        assert np.intersect1d(base, other).size == 0, f"streams {shift} dimensions apart coincide"


def test_uniform_range_is_half_open_at_zero():
    u = ob.uniforms(9, 77, 3, 0, 200000)
    assert u.min() > 0.0 and u.max() <= 1.0
    # multiples of 2^-24, so log(u) is finite and u is exact in binary32
    assert np.array_equal(u * 2 ** 24, np.round(u * 2 ** 24))


def test_uniform_statistics():
    n = 400000
    u = np_uniform(3, 1000, 5, np.arange(n))
    assert abs(u.mean() - 0.5) < 4 * math.sqrt(1 / 12 / n)
    assert abs(u.var() - 1 / 12) < 1e-3
    hist, _ = np.histogram(u, bins=64, range=(0, 1))
    chi2 = ((hist - n / 64) ** 2 / (n / 64)).sum()
    assert chi2 < 64 + 5 * math.sqrt(2 * 64)
    # neighbouring dimensions, pixels and samples are uncorrelated
    assert abs(np.corrcoef(u[:-1], u[1:])[0, 1]) < 0.01
    a = np_uniform(3, np.arange(n), 5, 7)
    b = np_uniform(3, np.arange(n) + 1, 5, 7)
    assert abs(np.corrcoef(a, b)[0, 1]) < 0.01
    c = np_uniform(3, 4242, np.arange(n), 0)
    d = np_uniform(3, 4242, np.arange(n) + 1, 0)
    assert abs(np.corrcoef(c, d)[0, 1]) < 0.01
    assert abs(np.corrcoef(a, c)[0, 1]) < 0.01


def test_streams_differ_between_seeds_pixels_samples():
    base = ob.uniforms(1, 10, 0, 0, 64)
    for other in (ob.uniforms(2, 10, 0, 0, 64), ob.uniforms(1, 11, 0, 0, 64), ob.uniforms(1, 10, 1, 0, 64)):
        assert (base == other).mean() < 0.05


def test_box_mueller_integral_known_answer():
    """The reference's Monte-Carlo demo (test/integrate.cu): importance-sample cos(xyz) over
    [-pi/2, pi/2]^3 with a unit Gaussian built by Box-Muller from (0,1] uniforms."""
    with open(os.path.join(GOLDEN, "anchors.json")) as f:
        want = json.load(f)["integrate_known_answer"]["value"]
    n = 1 << 20
    u = np_uniform(17, np.arange(n)[:, None], 0, np.arange(6)[None, :])
    r = np.sqrt(-2 * np.log(u[:, 0::2]))
    g = r * np.cos(2 * np.pi * u[:, 1::2])  # x, y, z ~ N(0, 1)
    inside = np.all(np.abs(g) <= np.pi / 2, axis=1)
    pdf = (2 * np.pi) ** -1.5 * np.exp(-(g ** 2).sum(1) / 2)
    f = np.where(inside, np.cos(g.prod(1)) / pdf, 0.0)
    est, err = f.mean(), f.std() / math.sqrt(n)
    assert abs(est - want) < 4 * err + 1e-3
    assert err < 0.05


def test_polynomial_log():
    x = np.concatenate([np.arange(1, 1 << 16) * 2.0 ** -24, np.linspace(2.0 ** -8, 1.0, 200001)]).astype(np.float32)
    got = ob.math_probe("log", x).astype(np.float64)
    assert np.max(np.abs(got - np.log(x.astype(np.float64)))) < 6e-7


def test_polynomial_sincos():
    r = (np.arange(1, (1 << 20) + 1) * 2.0 ** -20)
    a = (2 * np.pi * r).astype(np.float32)  # the angles the sampler forms
    s = ob.math_probe("sin", a).astype(np.float64)
    c = ob.math_probe("cos", a).astype(np.float64)
    ad = a.astype(np.float64)
    assert np.max(np.abs(s - np.sin(ad))) < 2e-7
    assert np.max(np.abs(c - np.cos(ad))) < 2e-7
    assert np.min(np.maximum(np.abs(s), np.abs(c))) > 0.7070  # the bound the jitter loop's quick reject uses


def test_polynomial_gamma():
    x = np.concatenate([np.linspace(0, 1, 5001), np.geomspace(1e-6, 1e6, 20001)]).astype(np.float32)
    got = ob.math_probe("gamma", x).astype(np.float64)
    want = x.astype(np.float64) ** np.float64(np.float32(1 / 2.2))
    assert np.max(np.abs(got - want) / np.maximum(want, 1e-30)) < 2e-6
    special = ob.math_probe("gamma", np.array([0.0, np.inf, np.nan, -1.0], np.float32))
    assert special[0] == 0.0 and np.isposinf(special[1]) and np.isnan(special[2]) and np.isnan(special[3])


def test_jitter_quick_reject_is_exact():
    """r1 < 0.77 always fails the half-pixel test, so skipping its evaluation changes nothing."""
    r1 = np.float32(0.77)
    radius = np.sqrt(np.float32(-2) * ob.math_probe("log", [r1])[0])
    assert radius * 0.7070 > 0.5
    below = (np.arange(1, int(0.77 * 2 ** 24), 997) * 2.0 ** -24).astype(np.float32)
    assert np.all(np.sqrt(-2 * ob.math_probe("log", below)) * 0.7070 > 0.5)
