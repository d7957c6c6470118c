/*
 * rt_math.h -- deterministic sampling arithmetic shared by the HIP kernels and
 * by CPU code (valid C99 and HIP C++).
 *
 * Everything in here is built from IEEE-754 binary32/binary64 +,-,*,/ and
 * integer operations only, so that a translation unit compiled WITHOUT
 * floating-point contraction (-ffp-contract=off, no fast-math) produces the
 * same bits on an x86-64 host and on a gfx950 device.  It replaces two things
 * the reference takes from the CUDA toolkit and that have no portable,
 * reproducible definition:
 *
 *   - cuRAND XORWOW (curand_init / curand_uniform,
 *     reference src/raytrace.cu:18-30,34-35,254-255,397) -> a stateless
 *     counter-based generator keyed on (seed, pixel, sample, dimension);
 *   - the fast intrinsics __logf/__cosf/__sinf/__powf
 *     (reference src/raytrace.cu:37-41,46-49,260-261) -> fixed polynomial
 *     kernels with absolute error below 3e-7 on the ranges the sampler uses.
 */
#ifndef RT_MATH_H
#define RT_MATH_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define RT_HD __host__ __device__ static inline
#else
#define RT_HD static inline
#endif

/* ---- bit casts ---------------------------------------------------------- */

RT_HD uint32_t rt_f2u(float f) {
    union { float f; uint32_t u; } c;
    c.f = f;
    return c.u;
}

RT_HD float rt_u2f(uint32_t u) {
    union { float f; uint32_t u; } c;
    c.u = u;
    return c.f;
}

/* ---- counter-based generator ------------------------------------------- */

/* 32-bit finaliser (two odd multiplies, three xor-shifts); a bijection. */
RT_HD uint32_t rt_mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

/* Key pair of one path: kp depends on (seed, pixel), ks on (seed, sample).
 * `sample` is the GLOBAL sample index of the pixel, so a frame split over
 * several devices by sample range draws exactly the numbers a single device
 * would draw. */
typedef struct {
    uint32_t kp;
    uint32_t ks;
} rt_rng_key;

RT_HD rt_rng_key rt_rng_make_key(uint32_t seed, uint32_t pixel, uint32_t sample) {
    rt_rng_key k;
    uint32_t a = rt_mix32(seed ^ 0x3c6ef372U);
    k.kp = rt_mix32(a ^ pixel);
    k.ks = rt_mix32((a + 0x9e3779b9U) ^ sample);
    return k;
}

/* The dim-th 32-bit word of a path's stream.  The dimension enters by XOR and
 * the sample key by addition: with both combined the same way (say kp + dim *
 * c) two pixels whose keys differ by a multiple j * c would draw the same
 * stream j dimensions apart, and among a few hundred thousand pixels such
 * pairs exist.  (kp ^ dim * c) == (kp' ^ (dim + j) * c) cannot hold along a
 * run of dimensions. */
RT_HD uint32_t rt_rng_word(rt_rng_key k, uint32_t dim) {
    return rt_mix32((k.kp ^ (dim * 0x9e3779b9U)) + k.ks);
}

/* Uniform in (0,1]: ((w>>8)+1) * 2^-24.  Every value is exactly representable,
 * 0 is excluded (Box-Muller takes its logarithm) and 1 is included, which is
 * the range of curand_uniform. */
RT_HD float rt_rng_uniform(rt_rng_key k, uint32_t dim) {
    uint32_t w = rt_rng_word(k, dim);
    return (float)((w >> 8) + 1U) * 5.9604644775390625e-08f;
}

/* ---- natural logarithm, x normal and positive --------------------------- */

RT_HD float rt_logf(float x) {
    /* x = m * 2^e with m in [sqrt(1/2), sqrt(2)) */
    uint32_t ix = rt_f2u(x);
    int32_t e = (int32_t)(ix - 0x3f3504f3U) >> 23;
    float m = rt_u2f(ix - ((uint32_t)e << 23));
    float f = m - 1.0f;
    float s = f / (2.0f + f);
    float z = s * s;
    /* log(m) = 2s (1 + z/3 + z^2/5 + z^3/7 + z^4/9), |s| <= 0.1716 */
    float p = 0.11111111f;
    p = p * z + 0.14285715f;
    p = p * z + 0.2f;
    p = p * z + 0.33333334f;
    p = p * z;
    float lm = 2.0f * s + (2.0f * s) * p;
    float fe = (float)e;
    /* ln2 split: hi has 16 significant bits so fe*hi is exact for |e| < 256 */
    return fe * 0.693145751953125f + (fe * 1.42860682030941723e-06f + lm);
}

/* ---- exp(t) for finite t ------------------------------------------------ */

RT_HD float rt_expf(float t) {
    if (t < -87.0f) return 0.0f;
    if (t > 88.0f) t = 88.0f;
    float kf = t * 1.44269504f;
    int32_t k = (int32_t)(kf + (kf < 0.0f ? -0.5f : 0.5f));
    float fk = (float)k;
    float r = (t - fk * 0.693145751953125f) - fk * 1.42860682030941723e-06f;
    /* degree-6 Taylor on |r| <= 0.3466: remainder < 2e-8 relative */
    float p = 1.3888889e-03f;
    p = p * r + 8.3333333e-03f;
    p = p * r + 4.1666668e-02f;
    p = p * r + 1.6666667e-01f;
    p = p * r + 0.5f;
    p = p * r + 1.0f;
    p = p * r + 1.0f;
    return rt_u2f(rt_f2u(p) + ((uint32_t)k << 23));
}

/* x^(1/2.2) for the output transfer curve (reference gamma_correct,
 * src/raytrace.cu:44-51: __powf(c, (float)(1.f/2.2))).  Zero maps to zero,
 * negative and NaN inputs to NaN, +inf to +inf, like powf. */
RT_HD float rt_gamma_encode(float x) {
    const float exponent = (float)(1.f / 2.2);
    if (!(x >= 0.0f)) return rt_u2f(0x7fc00000U);
    if (x == 0.0f) return 0.0f;
    if (x > 3.0e38f) return x;
    float scale = 0.0f;
    if (x < 1.1754944e-38f) { /* subnormal: lift by 2^24 */
        x = x * 16777216.0f;
        scale = -16.635532f; /* ln(2^-24) */
    }
    return rt_expf(exponent * (rt_logf(x) + scale));
}

/* ---- sine and cosine of an angle in [0, 8] radians ---------------------- */

RT_HD void rt_sincosf(float a, float* sn, float* cs) {
    /* quadrant k = nearest multiple of pi/2; Cody-Waite in three pieces,
     * the first with 8 significant bits so that k*P1 is exact */
    int32_t k = (int32_t)(a * 0.63661977f + 0.5f);
    float fk = (float)k;
    float r = ((a - fk * 1.5703125f) - fk * 4.837512969970703125e-4f) - fk * 7.54978995489188216e-8f;
    float z = r * r;
    float ps = -1.9515295891e-4f;
    ps = ps * z + 8.3321608736e-3f;
    ps = ps * z - 1.6666654611e-1f;
    float s = r + r * (z * ps);
    float pc = 2.443315711809948e-5f;
    pc = pc * z - 1.388731625493765e-3f;
    pc = pc * z + 4.166664568298827e-2f;
    float c = (1.0f - 0.5f * z) + (z * z) * pc;
    switch (k & 3) {
        case 0: *sn = s;  *cs = c;  break;
        case 1: *sn = c;  *cs = -s; break;
        case 2: *sn = -s; *cs = -c; break;
        default: *sn = -c; *cs = s; break;
    }
}

#endif /* RT_MATH_H */
