// path_kernels.hip.h -- the path kernels: camera samples (k_generate), one bounce (k_shade), accumulation and resolve, and the known-answer test kernels.
// Part of the device half of librt_amd.so; included by pathtrace.hip only.
#pragma once

#include "device_scene.hip.h"

namespace {

// ------------------------------------------------------- path kernels

struct Frame {
    float rot[9];
    float pos[3];
    float fx, fy, cx, cy;
    uint32_t width, n_pixels; // n_pixels = pixels rendered: the whole image, or a band of its rows
    uint32_t pixel_offset;  // index of the band's first pixel in the image
    uint32_t n_paths;       // paths in this batch = n_pixels * samples in batch
    uint32_t sample_begin;  // global index of the batch's first sample
    uint32_t seed;
    int32_t indirect, antialiasing, bounding_box, max_depth;
    int32_t count_casts;    // count the rays tested against the spheres (statistics)
    uint32_t min_waves;     // kernels whose paths stay in their lanes: the fewest wavefronts a long input is spread over (stretch_of)
    int32_t prune_zero_terms; // work whose result can only add exact zeros is dropped: the mesh walk of a shadow
                              // ray with a zero contribution, the rest of a path whose weight is zero (not in
                              // the exhaustive mode, which keeps the reference's amount of work)
};

// Path p of a batch is sample p / n_pixels of pixel p % n_pixels.  A batch holds at most 2^24 paths, so p and n_pixels
// are exact floats: the quotient by a reciprocal is off by at most one (error below 3 / n_pixels of a quotient below
// 2^24 / n_pixels; n_pixels of 1 and 2 have exact reciprocals), which the remainder shows and one step either way
// repairs -- a dozen instructions where an integer division and a remainder by an unknown divisor take forty.
struct PathId {
    uint32_t pixel, sample;
};
__device__ __forceinline__ PathId path_id(const Frame& f, uint32_t p) {
    static_assert(kMaxPathsPerBatch <= (1u << 24), "path indices must be exact in single precision");
    const float inv = __builtin_amdgcn_rcpf((float)f.n_pixels);
    uint32_t q = (uint32_t)((float)p * inv);
    int32_t r = (int32_t)(p - q * f.n_pixels);
    if (r < 0) {
        q -= 1u;
        r += (int32_t)f.n_pixels;
    } else if (r >= (int32_t)f.n_pixels) {
        q += 1u;
        r -= (int32_t)f.n_pixels;
    }
    return {f.pixel_offset + (uint32_t)r, f.sample_begin + q};
}
__device__ __forceinline__ uint32_t path_pixel(const Frame& f, uint32_t p) { return path_id(f, p).pixel; }

// State write-back of a freshly generated ray whose nearest sphere (ts, sid)
// and mesh verdict are known, and queueing for k_mesh.  All lanes of the
// wavefront must call it.
__device__ __forceinline__ uint32_t commit_ray(const Work& w, uint32_t counter, bool alive, uint32_t p, V3 start, V3 ray,
                                               float ts, int sid, bool walk) {
    if (alive) {
        w.ray_o[p] = make_float4(start.x, start.y, start.z, ts);
        w.ray_o.second(p) = make_float4(ray.x, ray.y, ray.z, __int_as_float(sid));
        w.hit[p] = kNoHit;
    }
    return enqueue(w, counter, wave_shard(), alive && walk, start, ray, p, ts);
}

// Nearest sphere + box test for a freshly generated ray, then commit_ray.
template <class Cache>
__device__ __forceinline__ void launch_ray(const DeviceScene& sc, const Cache& cache, const Work& w, const Frame& f, uint32_t counter,
                                           bool alive, uint32_t p, V3 start, V3 ray) {
    int sid = -1;
    float ts = INFINITY;
    bool walk = false;
    if (alive) {
        ts = nearest_sphere(sc, cache, ray, start, sid);
        // (the kernels that call this have registers to spare for the shorter box test)
        walk = needs_mesh<true>(sc, cache, f.bounding_box, ray, start);
    }
    commit_ray(w, counter, alive, p, start, ray, ts, sid, walk);
}

// Camera sample of every path of the batch (reference src/raytrace.cu:479-494).
//
// With anti-aliasing the pixel offset is a Box-Muller pair that is redrawn
// until both components are within half a pixel (src/raytrace.cu:33-42,
// 483-486): 6.8 draws per path on average, but the slowest of 64 lanes needs
// about 26, so a wavefront that keeps one path per lane idles three quarters
// of its lanes.  A lane therefore owns kGenPaths paths and searches their
// accepted draws back to back (first loop: lanes stay busy until the last
// path of the slowest lane), and only the accepted draw of each path is
// evaluated with the reference's arithmetic (second loop, all lanes together).
//
// The search classifies a draw with hardware transcendentals: r1 < 0.77 means
// radius > 0.722, so one component exceeds 0.5 (max(|cos|,|sin|) >= 0.7071);
// otherwise the approximate offsets (absolute error far below 1e-4) decide,
// except within 1e-4 of the half-pixel limit, where the reference's own
// arithmetic is evaluated.
constexpr uint32_t kGenPaths = 4; // at most; short batches use fewer so that the grid still fills the chip

__device__ __forceinline__ void jitter_exact(const rt_rng_key& key, uint32_t dim, float& dx, float& dy) {
    const float r1 = rt_rng_uniform(key, dim);
    const float r2 = rt_rng_uniform(key, dim + 1);
    const float radius = sqrt_rn(-2 * rt_logf(r1));
    const float angle = (float)(2 * M_PI * (double)r2);
    float sn, cs;
    rt_sincosf(angle, &sn, &cs);
    dx = radius * cs * 1.f;
    dy = radius * sn * 1.f;
}

// The same Box-Muller pair from the hardware's log2 / sin / cos (arguments in turns), as the reference takes it
// from __logf / __cosf / __sinf: absolute error of the offsets far below 1e-4 of a pixel.
__device__ __forceinline__ void jitter_fast(const rt_rng_key& key, uint32_t dim, float& dx, float& dy) {
    const float r1 = rt_rng_uniform(key, dim);
    const float r2 = rt_rng_uniform(key, dim + 1);
    const float radius = __builtin_amdgcn_sqrtf(-1.3862944f * __builtin_amdgcn_logf(r1)); // -2 ln 2 * log2(r1)
    dx = radius * __builtin_amdgcn_cosf(r2);
    dy = radius * __builtin_amdgcn_sinf(r2);
}

template <class M> __device__ __forceinline__ void jitter_m(const rt_rng_key& key, uint32_t dim, float& dx, float& dy) {
    if constexpr (M::kExact) jitter_exact(key, dim, dx, dy); else jitter_fast(key, dim, dx, dy);
}

// One trip of the search for a path's accepted draw.  Three draws of four fail the cheap test (r1 < 0.77) and a
// wavefront that evaluates one draw per lane and trip runs the expensive rest -- second number, logarithm, root, sine,
// cosine -- in every trip for the quarter of its lanes that pass.  So a trip looks at the first numbers of the next
// kJitterLook draws from `dim` on and evaluates the first of them that passes, which two lanes of three have.
// Accepted: returns true with `dim` at that draw.  Otherwise `dim` moves past every draw the trip has settled (the
// evaluated one and those before it, or all kJitterLook).  A draw is accepted when both offsets are within half a
// pixel (src/raytrace.cu:483-486).  ExactMath: classified with the hardware transcendentals where that is certain, by
// the reference's own arithmetic within 1e-4 of the limit.  FastMath: the hardware values are the arithmetic.  The
// draws are looked at in order, so the accepted one is the first the reference's loop accepts.
#ifndef RT_JITTER_LOOK
#define RT_JITTER_LOOK 4
#endif
constexpr int kJitterLook = RT_JITTER_LOOK; // (measured on the headline frame, 2 / 3 / 4 / 6 / 8 draws per trip: k_generate 158 / 149 / 146 / 148 / 156 us)
template <class M> __device__ __forceinline__ bool jitter_search(const rt_rng_key& key, uint32_t& dim) {
    // r1 = ((w >> 8) + 1) * 2^-24 >= 0.77f exactly when the generator's word w >= kPass: 0.77f * 2^24 = 12 918 455.36,
    // so (w >> 8) + 1 >= 12 918 456 (rt_rng_uniform; the comparison on the word saves the conversion per draw)
    constexpr uint32_t kPass = 12918455u << 8;
    uint32_t first = kJitterLook, word = 0;
#pragma unroll
    for (int i = kJitterLook - 1; i >= 0; i--) {
        const uint32_t w = rt_rng_word(key, dim + 2 * i);
        if (w >= kPass) { first = (uint32_t)i; word = w; }
    }
    const uint32_t d = dim + 2 * first;
    if (first == (uint32_t)kJitterLook) { dim = d; return false; }
    dim = d + 2;
    const float r1 = (float)((word >> 8) + 1U) * 5.9604644775390625e-08f; // rt_rng_uniform(key, d)
    const float r2 = rt_rng_uniform(key, d + 1);
    float radius;
    if constexpr (M::kExact) radius = __builtin_sqrtf(-1.3862944f * __builtin_amdgcn_logf(r1)); // -2 ln 2 * log2(r1)
    else radius = __builtin_amdgcn_sqrtf(-1.3862944f * __builtin_amdgcn_logf(r1));
    const float m = fmaxf(fabsf(radius * __builtin_amdgcn_cosf(r2)), fabsf(radius * __builtin_amdgcn_sinf(r2))); // argument in turns
    bool ok;
    if constexpr (!M::kExact) {
        ok = !(m > 0.5f);
    } else {
        ok = m < 0.4999f;
        if (!ok && !(m > 0.5001f)) { // within 1e-4 of the limit: the reference's own arithmetic decides
            float dx, dy;
            jitter_exact(key, d, dx, dy);
            ok = !(fabsf(dx) > 0.5f || fabsf(dy) > 0.5f);
        }
    }
    if (ok) dim = d;
    return ok;
}

// Every camera ray starts at the eye: the start-dependent half of the sphere test is the same for all paths
// and is computed once per workgroup (same operations, same bits).  All threads must call it; the caller
// synchronises the workgroup before eye_oc is read.
template <class M, class Cache>
__device__ __forceinline__ void stage_eye(const DeviceScene& sc, const Cache& cache, const Frame& f, float4* eye_oc) {
    if (threadIdx.x < min(sc.n_spheres, kSphereCache)) {
        const V3 oc = mk(f.pos[0], f.pos[1], f.pos[2]) - xyz(cache.sphere[threadIdx.x]);
        eye_oc[threadIdx.x] = make_float4(oc.x, oc.y, oc.z, oc_norm2_m<M>(oc));
    }
}

// The camera ray through pixel (ux, uy) (reference src/raytrace.cu:479-494); with anti-aliasing the pixel offset is
// the draw at dimension `dim` of the path's stream, and `dim` moves past it.
template <class M>
__device__ __forceinline__ V3 camera_ray(const Frame& f, const rt_rng_key& key, uint32_t ux, uint32_t uy, uint32_t& dim) {
    float dx = 0.f, dy = 0.f;
    if (f.antialiasing) {
        jitter_m<M>(key, dim, dx, dy);
        dim += 2;
    }
    const float px = (float)ux + 0.5f + dx;
    const float py = (float)uy + 0.5f + dy;
    const V3 dir = normalize_m<M>(mk(div_m<M>(px - f.cx, f.fx), div_m<M>(py - f.cy, f.fy), 1.f));
    return mk(dot_m<M>(mk(f.rot[0], f.rot[1], f.rot[2]), dir),
              dot_m<M>(mk(f.rot[3], f.rot[4], f.rot[5]), dir),
              dot_m<M>(mk(f.rot[6], f.rot[7], f.rot[8]), dir));
}

// nearest_sphere for a ray that starts at the eye, with the halves stage_eye prepared
template <class M, class Cache>
__device__ __forceinline__ float nearest_sphere_from_eye(const DeviceScene& sc, const Cache& cache, const float4* eye_oc, V3 start, V3 ray, int& sid) {
    float ts = INFINITY;
    sid = -1;
    const uint32_t cached = min(sc.n_spheres, kSphereCache);
    for (uint32_t i = 0; i < cached; i++) {
        const float4 e = eye_oc[i];
        sphere_step<M>(dot_m<M>(ray, xyz(e)), e.w, cache.sphere[i].w, i, ts, sid);
    }
    for (uint32_t i = cached; i < sc.n_spheres; i++) {
        const float4 sph = sc.sphere[i];
        const V3 oc = start - xyz(sph);
        sphere_step<M>(dot_m<M>(ray, oc), oc_norm2_m<M>(oc), sph.w, i, ts, sid);
    }
    return ts;
}

// kFits: the scene's tables fit the LDS cache (SceneCacheT): no code and no pointers for entries beyond it.
template <uint32_t kPaths, class M, bool kFits>
// regenerate: leave (accepted draw, walks the mesh) per path instead of the ray; the first bounce then computes the
// camera ray again (k_shade_fast_first).  (An argument of its own: one more word in Frame costs the bounce kernels,
// which are at the limit of their scalar registers, two dozen spills.)
__global__ __launch_bounds__(kBlock) void k_generate(DeviceScene sc, Work w, Frame f, int regenerate) {
    __shared__ SceneCacheT<kFits> cache;
    __shared__ uint32_t key_kp[kPaths][kBlock], key_ks[kPaths][kBlock];
    __shared__ uint32_t accepted[kPaths][kBlock]; // first dimension of the accepted draw
    __shared__ float4 eye_oc[kSphereCache];       // per cached sphere: OC = eye - centre, |OC| * |OC|
    stage_scene(sc, cache);
    stage_eye<M>(sc, cache, f, eye_oc);
    __syncthreads();
    const uint32_t base = blockIdx.x * (kBlock * kPaths) + threadIdx.x;
    // the three arrays are written and read by the lanes of ONE wavefront (its 64 columns), in program order: no barrier needed
#pragma unroll
    for (uint32_t r = 0; r < kPaths; r++) {
        const uint32_t p = base + r * kBlock;
        if (p < f.n_paths) {
            const PathId id = path_id(f, p);
            const rt_rng_key key = rt_rng_make_key(f.seed, id.pixel, id.sample);
            key_kp[r][threadIdx.x] = key.kp;
            key_ks[r][threadIdx.x] = key.ks;
            accepted[r][threadIdx.x] = 0;
        }
    }
    if (f.antialiasing) {
        // The accepted draw of every path of the wavefront (64 * kPaths of them).  A search takes 6.8 draws on average
        // and several times that now and then, so the lanes do not keep the searches of their own paths: a lane that has
        // found one takes the next path nobody has started (in slot order, by ballot and prefix count -- no atomics), and
        // the wavefront is done after about total draws / 64 trips instead of after its unluckiest lane's kPaths searches.
        // Which lane searches changes nothing about what is found.
        const uint32_t lane = __lane_id(), wave_first = threadIdx.x - lane;
        const uint32_t block_base = blockIdx.x * (kBlock * kPaths);
        constexpr uint32_t kSlots = 64 * kPaths; // slot j of the wavefront: path block_base + (j / 64) * kBlock + wave_first + j % 64
        auto valid = [&](uint32_t j) { return j < kSlots && block_base + (j >> 6) * kBlock + wave_first + (j & 63u) < f.n_paths; };
        uint32_t j = lane, next = 64, dim = 0;
        bool busy = valid(j);
        rt_rng_key key;
        key.kp = key_kp[0][threadIdx.x];
        key.ks = key_ks[0][threadIdx.x];
        for (;;) {
            bool found = false;
            if (busy) {
                found = jitter_search<M>(key, dim);
                if (found) accepted[j >> 6][wave_first + (j & 63u)] = dim;
            }
            const unsigned long long free_lanes = __ballot(!busy || found);
            if (next >= kSlots) {
                if (free_lanes == ~0ULL) break; // nothing left to start, nothing running
                if (found) busy = false;
                continue;
            }
            if (!busy || found) { // take the next slot nobody has started
                j = next + (uint32_t)__popcll(free_lanes & ((1ULL << lane) - 1ULL));
                busy = valid(j);
                dim = 0;
                if (busy) {
                    key.kp = key_kp[j >> 6][wave_first + (j & 63u)];
                    key.ks = key_ks[j >> 6][wave_first + (j & 63u)];
                }
            }
            next += (uint32_t)__popcll(free_lanes);
        }
    }
    __syncthreads(); // (the draws were found by other lanes)
    uint32_t n_casts = 0;
    for (uint32_t r = 0; r < kPaths; r++) {
        const uint32_t p = base + r * kBlock;
        const bool alive = p < f.n_paths;
        V3 start = mk(f.pos[0], f.pos[1], f.pos[2]);
        V3 ray = mk(0.f, 0.f, 1.f);
        uint32_t first_dim = 0, dim = 0;
        if (alive) {
            const uint32_t pixel = path_pixel(f, p);
            const uint32_t ux = pixel % f.width, uy = pixel / f.width;
            rt_rng_key key;
            key.kp = key_kp[r][threadIdx.x];
            key.ks = key_ks[r][threadIdx.x];
            first_dim = dim = f.antialiasing ? accepted[r][threadIdx.x] : 0u;
            ray = camera_ray<M>(f, key, ux, uy, dim);
        }
        int sid = -1;
        float ts = INFINITY;
        bool walk = false;
        if (alive) {
            ts = nearest_sphere_from_eye<M>(sc, cache, eye_oc, start, ray, sid);
            walk = needs_mesh<true, M>(sc, cache, f.bounding_box, ray, start);
        }
        if (regenerate) { // the first bounce computes the ray again: it needs the accepted draw and, if the ray walks the mesh, a hit slot
            if (alive) {
                w.gen[p] = first_dim | (walk ? kGenWalks : 0u);
                if (walk) w.hit[p] = kNoHit;
            }
            enqueue(w, 0, wave_shard(), alive && walk, start, ray, p, ts);
        } else {
            if (alive) {
                w.thr[p] = make_float4(1.f, 1.f, 1.f, __uint_as_float(dim));
                w.rad[p] = make_float4(0.f, 0.f, 0.f, __uint_as_float(F_ALIVE));
            }
            commit_ray(w, 0, alive, p, start, ray, ts, sid, walk);
        }
        n_casts += (uint32_t)__popcll(__ballot(alive));
    }
    if (f.count_casts) add_casts(w, 0, n_casts);
}

// Adds the light contributions a path left pending at its previous hit, in
// light order, now that k_mesh has answered its shadow rays.
// (The caller may have fetched the first light's pending record and shadow result already.)
__device__ __forceinline__ V3 settle_pending(const DeviceScene& sc, const Work& w, uint32_t n_paths, uint32_t p, V3 L,
                                             bool have_first = false, float4 first_pend = make_float4(0, 0, 0, 0),
                                             unsigned long long first_key = 0) {
    for (uint32_t l = 0; l < sc.n_lights; l++) {
        const bool fetched = have_first && l == 0;
        const float4 pe = fetched ? first_pend : w.pend[(size_t)p * sc.n_lights + l];
        if (pe.w == PEND_NONE) continue;
        bool visible = true;
        if (pe.w != PEND_ALWAYS) {
            const unsigned long long key = fetched ? first_key : w.hit[shadow_base(w) + (size_t)p * sc.n_lights + l];
            visible = !(__uint_as_float((uint32_t)(key >> 32)) < pe.w);
        }
        if (visible) L = L + mk(pe.x, pe.y, pe.z);
    }
    return L;
}

// cosine-weighted direction about N (reference random_cos,
// src/raytrace.cu:253-296, with argmin_abs of include/vector.cuh:87-98: the
// running minimum starts at the signed first component)
template <class M = ExactMath>
__device__ __forceinline__ V3 random_cos(float r1, float r2, V3 N) {
    const float tmp1 = sqrt_m<M>(1 - r2);
    float sn, cs;
    if constexpr (M::kExact) {
        const float tmp2 = (float)(2 * M_PI * (double)r1);
        rt_sincosf(tmp2, &sn, &cs);
    } else { // the hardware's sine and cosine take the angle in turns
        sn = __builtin_amdgcn_sinf(r1);
        cs = __builtin_amdgcn_cosf(r1);
    }
    const float x = tmp1 * cs;
    const float y = tmp1 * sn;
    const float z = sqrt_m<M>(r2);
    const float n[3] = {N.x, N.y, N.z};
    float m = n[0];
    int k = 0;
#pragma unroll
    for (int i = 0; i < 3; i++)
        if (fabsf(n[i]) < m) { m = fabsf(n[i]); k = i; }
    V3 T1;
    if (k == 0) T1 = mk(0.f, N.z, -N.y);       // (i, j) = (1, 2)
    else if (k == 1) T1 = mk(-N.z, 0.f, N.x);  // (i, j) = (2, 0)
    else T1 = mk(N.y, -N.x, 0.f);              // (i, j) = (0, 1)
    T1 = normalize_m<M>(T1);
    const V3 T2 = normalize_m<M>(cross_m<M>(N, T1));
    if constexpr (M::kExact) return (x * T1 + y * T2) + z * N;
    else return along_m<M>(along_m<M>(x * T1, y, T2), z, N);
}

template <class M = ExactMath>
__device__ __forceinline__ V3 reflect(V3 ray, V3 N) {
    if constexpr (M::kExact) return ray - (2 * dot(ray, N)) * N;
    else return along_m<M>(ray, -2.f * fast_dot(ray, N), N);
}

// The material's response to a hit at P with shading normal N (reference get_color,
// src/raytrace.cu:338-437): the point and direction the path continues from, whether the lights are
// evaluated here (`lit`: diffuse hit), whether the path goes on (`alive`) and whether its weight
// picks up the surface colour (`tinted`).  `ray` is the incoming direction on entry.  N comes back
// flipped when a glass surface is left from the inside.
template <class M = ExactMath, class Cache>
__device__ __forceinline__ void respond_to_hit(const DeviceScene& sc, const Cache& cache, const Frame& f, const rt_rng_key& rng,
                                               int depth, int m, V3 P, V3& N, V3& start, V3& ray, V3 weight, V3& colour,
                                               uint32_t& dim, uint32_t& flags, bool& alive, bool& lit, bool& tinted) {
    const float4 mc = material_color(sc, cache, (uint32_t)m);
    const int type = __float_as_int(mc.w);
    colour = xyz(mc);
    V3 next_weight = weight;
    alive = true;

    if (type == RT_DIFFUSE) {
        if (!f.indirect && depth == 0) flags |= F_EARLY;
        if (f.antialiasing) flags &= ~F_EARLY;
        lit = true;
        start = offset_m<M>(P, 1.f, N);
        if (!f.indirect) {
            alive = false;
        } else {
            next_weight = weight * colour;
            tinted = true;
            if (depth + 1 < f.max_depth) { // (the last bounce casts no ray: its direction would be drawn for nothing)
                const float r1 = rt_rng_uniform(rng, dim);
                const float r2 = rt_rng_uniform(rng, dim + 1);
                dim += 2;
                ray = random_cos<M>(r1, r2, N);
            }
        }
    } else if (type == RT_MIRROR) {
        const V3 rr = reflect<M>(ray, N);
        start = offset_m<M>(P, 1.f, N);
        ray = rr;
    } else { // RT_GLASS
        const float2 ior = material_ior(sc, cache, (uint32_t)m);
        float ro = ior.x, ri = ior.y;
        float wn = dot_m<M>(ray, N);
        // Schlick reflectance (reference src/raytrace.cu:388-394)
        const float k0 = div_m<M>((ro - ri) * (ro - ri), (ro + ri) * (ro + ri));
        const float om = 1.f - fabsf(wn);
        const float om2 = om * om;
        const float R = k0 + (1.f - k0) * om2 * om2 * om;
        const float r = rt_rng_uniform(rng, dim);
        dim += 1;
        if (r < R) {
            const V3 rr = reflect<M>(ray, N);
            start = offset_m<M>(P, 1.f, N);
            ray = rr;
        } else {
            if (wn > 0) { // leaving the medium: swap indices, flip the normal
                const float sw = ro; ro = ri; ri = sw;
                N = -1.f * N;
                wn = dot_m<M>(ray, N);
            }
            const float eta = div_m<M>(ro, ri);
            const float k = 1.f - eta * eta * (1.f - wn * wn);
            if (k < 0) { // total internal reflection
                const V3 rr = reflect<M>(ray, N);
                start = offset_m<M>(P, 1.f, N);
                ray = rr;
            } else {
                const V3 wt_t = eta * (ray - wn * N);
                const V3 wt_n = (-sqrt_m<M>(k)) * N;
                start = offset_m<M>(P, -1.f, N);
                ray = wt_t + wt_n;
            }
        }
    }
    if (depth + 1 >= f.max_depth) alive = false;
    // A weight of exactly zero in all three channels (a pure red wall seen from a pure
    // green one) stays zero, and every later term of the path is weight * direct light
    // with a finite direct light: the path has nothing more to add.
    if (f.prune_zero_terms && next_weight.x == 0.f && next_weight.y == 0.f && next_weight.z == 0.f) alive = false;
}

// One bounce of every live path (reference get_color loop body,
// src/raytrace.cu:317-442).  `depth` is the bounce index; counter depth+1
// receives the rays this bounce queues for k_mesh.
//
// The kernel has two parts.  The first, divergent by material, finds the hit
// point, its normal and the bounce ray.  The second is shared by all lanes:
// the sphere scans and box tests of the shadow rays (diffuse hits) and of the
// bounce ray.  The first light's shadow ray and the bounce ray of a diffuse
// hit leave the same point, so their scans run as one (nearest_sphere_pair).
// kContinue: a path whose bounce ray cannot meet a triangle (it misses every object box) and that leaves no light term
// waiting for a shadow ray has nothing to ask of the triangle stage: its next bounce needs no other kernel.  Such a
// path does not go through memory: it stays in its lane and takes its next bounce at once.  Three quarters of the
// (path, bounce) pairs of a typical frame are of that kind, and the kernel is bound by the path state it writes and
// reads.  The lanes that fall free (their path ended, or was written for the triangle stage) take the next paths of
// the wavefront's own stretch of the input, so that a wavefront stays full until that stretch is used up.  The
// arithmetic per path is the same in the same order, so the image does not change by a bit.
// With kContinue `depth` counts the kernels a path has been through rather than its bounces: the paths of one list
// are at different bounces (each one's index rides in its flags, above kBounceShift), all of them at `depth` or
// beyond, so max_depth launches still see every path to its end.  A path is written at most once per kernel, always
// to the list and queue of stage depth + 1, and a wavefront writes at most the paths of its own stretch there.
// kRegenerate (first bounce of a launch per bounce, k_generate's `regenerate`): the camera ray is computed again from what
// k_generate left in Work::gen instead of being read back.
// kFits: the scene's tables fit the LDS cache (SceneCacheT).
template <class M, bool kContinue = false, bool kRegenerate = false, bool kFits = false>
__device__ __forceinline__ void shade_bounce(const DeviceScene& sc, const Work& w, const Frame& f, int depth) {
    static_assert(!(kContinue && kRegenerate), "lanes that fetch a few paths at a time would compute their rays with a few lanes");
    __shared__ SceneCacheT<kFits> cache;
    __shared__ float4 eye_oc[kRegenerate ? kSphereCache : 1];
    // The first bounce sees every path; later ones walk the list of paths the previous bounce
    // left alive, so that the lanes of a wavefront are all busy however many paths have ended.
    // A list entry is the path index, with kLivePending set when light terms of the previous
    // hit await their shadow rays.
    // The path's ray and hit slot sit at its position in the list (`index`); the camera rays of the
    // first bounce at the path's own index.
    uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    uint32_t index = p;
    bool pending = false;
    __shared__ ShardTable table; // kContinue: the shards of the input list
    uint32_t cursor = 0, cursor_end = 0; // kContinue: the wavefront's stretch of the input, in entries of the walk through the shards
    if constexpr (!kContinue) {
        if (depth > 0) {
            const ShardMap map = load_shard_map(w.live_count + (size_t)depth * kShards * kCountStride);
            if (blockIdx.x >= map.n_ray_blocks) return; // the whole workgroup
            bool active;
            index = (uint32_t)(depth & 1) * w.live_total + ray_position(map, blockIdx.x, w.live_capacity, active);
            const uint32_t entry = active ? w.live[index] : 0xffffffffu;
            p = active ? (entry & ~kLivePending) : 0xffffffffu;
            pending = active && (entry & kLivePending) != 0;
        }
    } else {
        uint32_t n_in = f.n_paths;
        if (depth > 0) {
            const ShardMap map = load_shard_map(w.live_count + (size_t)depth * kShards * kCountStride);
            n_in = map.n_ray_blocks * kBlock;
            if (threadIdx.x < kShards) { // (stage_scene's barrier comes before the first use)
                table.count[threadIdx.x] = map.my_count;
                table.before[threadIdx.x] = map.before;
            }
            static_assert(kShards <= 64, "the first wavefront fills the table");
        }
        const uint32_t stretch = stretch_of(n_in, f.min_waves);
        const uint32_t wave = wave_index();
        if ((size_t)blockIdx.x * (blockDim.x >> 6) * stretch >= n_in) return; // the whole workgroup
        const size_t first = (size_t)wave * stretch;
        cursor = (uint32_t)(first < n_in ? first : n_in);
        cursor_end = (uint32_t)(first + stretch < n_in ? first + stretch : n_in);
        p = 0xffffffffu;
    }
    stage_scene(sc, cache);
    if constexpr (kRegenerate) {
        stage_eye<M>(sc, cache, f, eye_oc);
        __syncthreads();
    }
    // the path's state on entry: colour and flags, ray (origin + nearest-sphere distance, direction + sphere index),
    // weight + next random dimension, the triangle stage's answer; read from memory or kept from the previous trip of
    // the loop below (kContinue)
    float4 rad = make_float4(0.f, 0.f, 0.f, 0.f), ro4 = rad, rd4 = rad, th = rad;
    unsigned long long key = kNoHit;
    const uint32_t counter = (uint32_t)depth + 1u;
    const uint32_t shard = wave_shard();
    bool fresh = p < f.n_paths; // the lane has to fetch its path's state
    for (;;) { // (one trip unless kContinue)
    uint32_t at_index = index; // where the state of a path that is fetched lies
    float4 first_pend = make_float4(0.f, 0.f, 0.f, PEND_NONE); // (fetched with the state: the first light term that waits, its shadow ray's answer)
    unsigned long long first_key = kNoHit;
    if constexpr (kContinue) {
        // the free lanes take the next entries of the stretch, in lane order
        const unsigned long long free_lanes = __ballot(p >= f.n_paths);
        const uint32_t left = cursor_end - cursor, n_free = (uint32_t)__popcll(free_lanes);
        const uint32_t n_take = n_free < left ? n_free : left;
        const uint32_t rank = (uint32_t)__popcll(free_lanes & ((1ULL << __lane_id()) - 1ULL));
        const bool take = p >= f.n_paths && rank < n_take;
        if (depth == 0) {
            if (take) p = at_index = cursor + rank;
        } else if (n_take) {
            bool active;
            const uint32_t at = list_position(table, cursor, cursor + rank, w.live_capacity, active);
            if (take) {
                at_index = (uint32_t)(depth & 1) * w.live_total + at;
                const uint32_t entry = active ? w.live[at_index] : 0xffffffffu;
                p = active ? (entry & ~kLivePending) : 0xffffffffu;
                pending = active && (entry & kLivePending) != 0;
            }
        }
        cursor += n_take;
        if (__ballot(p < f.n_paths) == 0ULL) {
            if (cursor == cursor_end) break; // (the whole wavefront)
            continue; // (entries past the end of a shard's last block)
        }
        fresh = take && p < f.n_paths;
    }
    if (fresh) {
        if constexpr (kRegenerate) {
            const uint32_t g = w.gen[p];
            if (g & kGenWalks) key = w.hit[p];
            const PathId id = path_id(f, p);
            const uint32_t pixel = id.pixel;
            const uint32_t ux = pixel % f.width, uy = pixel / f.width;
            uint32_t dim = g & ~kGenWalks;
            const V3 eye = mk(f.pos[0], f.pos[1], f.pos[2]);
            const V3 cray = camera_ray<M>(f, rt_rng_make_key(f.seed, pixel, id.sample), ux, uy, dim);
            int csid;
            const float cts = nearest_sphere_from_eye<M>(sc, cache, eye_oc, eye, cray, csid);
            rad = make_float4(0.f, 0.f, 0.f, __uint_as_float(F_ALIVE));
            th = make_float4(1.f, 1.f, 1.f, __uint_as_float(dim));
            ro4 = make_float4(eye.x, eye.y, eye.z, cts);
            rd4 = make_float4(cray.x, cray.y, cray.z, __int_as_float(csid));
        } else {
            // the whole state of the path in one round trip to memory
            rad = w.rad[p];
            ro4 = w.ray_o[at_index], rd4 = w.ray_o.second(at_index), th = w.thr[at_index];
            key = w.hit[at_index];
            if (pending) {
                first_pend = w.pend[(size_t)p * sc.n_lights];
                first_key = w.hit[shadow_base(w) + (size_t)p * sc.n_lights];
            }
        }
    }
    bool alive = false; // the path continues with a bounce ray
    bool lit = false;   // diffuse hit: the lights are evaluated at `start`
    bool open = false;  // the path was alive on entry: its radiance record is rewritten
    V3 start = mk(0, 0, 0), ray = mk(0, 0, 1);
    V3 P = mk(0, 0, 0), N = mk(0, 0, 1), L = mk(0, 0, 0);
    V3 weight = mk(0, 0, 0), colour = mk(0, 0, 0); // path weight before this hit, surface colour
    uint32_t flags = 0;
    uint32_t next_dim = 0;     // next random dimension of the path after this bounce
    bool tinted = false;       // the weight after this bounce is weight * colour (diffuse hit with indirect light)
    float ts = INFINITY;       // the bounce ray's nearest sphere
    int sid = -1;
    bool at_once = false;      // the path goes on at once, in this kernel: nothing of it is written
    float4 next_thr = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p < f.n_paths) {
        flags = __float_as_uint(rad.w);
        if (flags & F_ALIVE) {
            open = true;
            int bounce = depth;
            if constexpr (kContinue) {
                bounce = (int)(flags >> kBounceShift);
                flags += 1u << kBounceShift;
            }
            L = mk(rad.x, rad.y, rad.z);
            if (flags & F_PENDING) {
                L = settle_pending(sc, w, f.n_paths, p, L, pending, first_pend, first_key);
                flags &= ~F_PENDING;
            }
            start = xyz(ro4);
            ray = xyz(rd4);
            const float ts = ro4.w;
            int sid = __float_as_int(rd4.w);
            const float tt = __uint_as_float((uint32_t)(key >> 32));
            int tid = (int)(uint32_t)key; // 0xffffffff -> -1

            // nearer of the two, spheres win ties (reference get_intersection,
            // src/raytrace.cu:194-215; `tmp` starts as the triangle id)
            float min_dist = INFINITY;
            int tmp = tid;
            if (ts < min_dist) { min_dist = ts; tmp = tid; tid = -1; }
            if (tt < min_dist) { min_dist = tt; sid = -1; tid = tmp; }

            if (sid >= 0 || tid >= 0) {
                P = isinf(min_dist) ? mk(INFINITY, INFINITY, INFINITY) : along_m<M>(start, min_dist, ray);
                int m;
                if (sid >= 0) {
                    m = sphere_material(sc, cache, (uint32_t)sid);
                    N = normalize_m<M>(P - xyz(sphere_at(sc, cache, (uint32_t)sid)));
                } else {
                    float alpha, beta, gamma;
                    barycentrics<M>(sc, tid, ray, start, alpha, beta, gamma);
                    const float4 s0 = sc.tri_shade[3 * tid + 0], s1 = sc.tri_shade[3 * tid + 1], s2 = sc.tri_shade[3 * tid + 2];
                    m = __float_as_int(s0.w);
                    if constexpr (M::kExact) N = normalize((alpha * xyz(s0) + beta * xyz(s1)) + gamma * xyz(s2));
                    else N = normalize_m<M>(along_m<M>(along_m<M>(alpha * xyz(s0), beta, xyz(s1)), gamma, xyz(s2)));
                }
                weight = xyz(th);
                uint32_t dim = __float_as_uint(th.w);
                const PathId id = path_id(f, p);
                const rt_rng_key rng = rt_rng_make_key(f.seed, id.pixel, id.sample);
                respond_to_hit<M>(sc, cache, f, rng, bounce, m, P, N, start, ray, weight, colour, dim, flags, alive, lit, tinted);
                next_dim = dim;
            }
        }
    }

    // ---- shared part: shadow rays of diffuse hits, then the bounce ray ----
    bool walk = false;
    bool bounce_probed = false;
    bool deferred = false;
    // the last light's shadow ray is queued together with the bounce ray
    bool last_want = false;
    V3 last_sray = mk(0, 0, 1);
    float last_dist = 0.f;
    uint32_t last_slot = 0;
    for (uint32_t l = 0; l < sc.n_lights; l++) {
        // Direct light (reference src/raytrace.cu:351-360) and its shadow ray
        // (is_visible, src/raytrace.cu:218-244).  Lanes without a diffuse hit
        // carry placeholders and are masked by `lit`.
        const float4 li = light_at(sc, cache, l);
        const V3 lp = xyz(li);
        const V3 sp_ = lp - P;
        V3 wi, sray;
        float d, dist;
        if constexpr (M::kExact) {
            d = norm(sp_);
            wi = (1.f / d) * sp_;
            sray = normalize(lp - start);
            dist = norm(lp - start);
        } else { // one reciprocal square root per vector serves its length and its direction
            const float d2 = fast_dot(sp_, sp_), rd = __builtin_amdgcn_rsqf(d2);
            d = d2 * rd;
            wi = rd * sp_;
            const V3 ls = lp - start;
            const float l2 = fast_dot(ls, ls), rl = __builtin_amdgcn_rsqf(l2);
            dist = l2 * rl;
            sray = rl * ls;
        }
        const float nwi = fmaxf(dot_m<M>(N, wi), 0.f);
        float ds = INFINITY;
        int shadow_sid = -1;
        bool shadow_walk = false;
        if (l == 0) {
            nearest_sphere_pair<M>(sc, cache, start, lit, sray, dist, ds, shadow_sid, alive, ray, ts, sid);
            needs_mesh_pair<M>(sc, cache, f.bounding_box, start, lit && !(ds < dist), sray, shadow_walk, alive, ray, walk);
            bounce_probed = true;
        } else if (lit) {
            ds = nearest_sphere_shadow<M>(sc, cache, sray, start, dist, shadow_sid);
            shadow_walk = !(ds < dist) && needs_mesh<false, M>(sc, cache, f.bounding_box, sray, start);
        }
        const size_t slot = (size_t)p * sc.n_lights + l;
        if (lit) {
            const bool blocked = ds < dist;
            // ((I / (4 pi^2 d^2) * colour) * visible) * N.wi, the first
            // factor in double (reference src/raytrace.cu:360)
            V3 direct;
            if constexpr (M::kExact) {
                const double q = (double)li.w / (4 * M_PI * M_PI * (double)d * (double)d);
                direct = nwi * ((blocked ? 0.f : 1.f) * scale_d(q, colour));
            } else {
                const float q = li.w * __builtin_amdgcn_rcpf((float)(4 * M_PI * M_PI) * d * d);
                direct = (blocked ? 0.f : nwi * q) * colour;
            }
            const V3 c = weight * direct;
            // A contribution of exactly zero (light behind the surface, black surface) is the
            // same added or not: what the triangles would say about its shadow ray does not
            // matter.  (L is a sum of non-negative terms, never -0, so L + 0 has L's bits.)
            if (f.prune_zero_terms && c.x == 0.f && c.y == 0.f && c.z == 0.f) shadow_walk = false;
            if (shadow_walk) {
                if (!deferred) // first pending light: clear the earlier slots
                    for (uint32_t e = 0; e < l; e++) w.pend[(size_t)p * sc.n_lights + e].w = PEND_NONE;
                deferred = true;
                w.pend[slot] = make_float4(c.x, c.y, c.z, dist);
                w.hit[shadow_base(w) + slot] = kNoHit;
            } else if (deferred) {
                w.pend[slot] = make_float4(c.x, c.y, c.z, PEND_ALWAYS);
            } else {
                L = L + c;
            }
        }
        if (l + 1 < sc.n_lights) {
            enqueue(w, counter, shard, lit && shadow_walk, start, sray, (uint32_t)(shadow_base(w) + slot), dist);
        } else {
            last_want = lit && shadow_walk;
            last_sray = sray;
            last_dist = dist;
            last_slot = (uint32_t)(shadow_base(w) + slot);
        }
    }
    if (f.count_casts) {
        add_casts(w, counter, (uint32_t)__popcll(__ballot(lit)) * sc.n_lights + (uint32_t)__popcll(__ballot(alive)));
        add_bounces(w, (uint32_t)__popcll(__ballot(open)));
    }
    if (deferred) flags |= F_PENDING;
    if (!bounce_probed && alive) { // a scene without lights
        ts = nearest_sphere<M>(sc, cache, ray, start, sid);
        walk = needs_mesh<false, M>(sc, cache, f.bounding_box, ray, start);
    }
    at_once = kContinue && alive && !walk && !deferred;
    if constexpr (kContinue) {
        // The stretch is used up and few paths are left in the wavefront: they go to the next launch's list like the
        // others (a trip costs a wavefront the same with 10 lanes as with 64; there they fill lanes again).
        if (cursor == cursor_end && (uint32_t)__popcll(__ballot(at_once)) < kKeepLanes) at_once = false;
    }
    if (open && !at_once) {
        if (!alive) flags &= ~F_ALIVE;
        w.rad[p] = make_float4(L.x, L.y, L.z, __uint_as_float(flags));
    }
    // (the product again rather than three registers kept across the lights: same operands, same bits)
    const V3 next_weight = tinted ? weight * colour : weight;
    next_thr = make_float4(next_weight.x, next_weight.y, next_weight.z, __uint_as_float(next_dim));
    finish_bounce(w, counter, shard, last_want, start, last_sray, last_slot, last_dist, alive && !at_once, walk, ray, ts, sid,
                  next_thr, p | (deferred ? kLivePending : 0u));
    if constexpr (!kContinue) {
        break;
    } else {
        // a path that goes on at once keeps its lane; the others' lanes are free
        rad = make_float4(L.x, L.y, L.z, __uint_as_float(flags));
        ro4 = make_float4(start.x, start.y, start.z, ts);
        rd4 = make_float4(ray.x, ray.y, ray.z, __int_as_float(sid));
        th = next_thr;
        key = kNoHit; // no triangle can be met: the ray misses every object box
        pending = false;
        if (!at_once) p = 0xffffffffu;
    }
    } // trips
}

// The bounce kernels.  kFits: the scene's tables fit the LDS cache (every scene the reference ships; the host checks the
// counts): no code and no pointers for entries beyond it -- k_shade_fast 2 965 -> 2 404 instructions, 170 -> 93 scalar
// registers spilled to vector lanes, no scratch, 904 -> 850 us per headline frame.
// (At least five wavefronts per SIMD: 96 VGPRs.  One register more would cost a fifth of the occupancy.)
template <bool kFits>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(5, 8))) void k_shade(DeviceScene sc, Work w, Frame f, int depth) {
    shade_bounce<ExactMath, false, false, kFits>(sc, w, f, depth);
}

// rt_options.arithmetic = 1: batches of kStaysFromPaths paths or more ...
template <bool kFits>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(5, 8))) void k_shade_fast(DeviceScene sc, Work w, Frame f, int depth) {
    shade_bounce<FastMath, true, false, kFits>(sc, w, f, depth);
}

// ... and smaller ones, which have too few wavefronts for long stretches and are not bound by memory: a launch per bounce,
// every path of a list in a lane of its own.
template <bool kFits>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(5, 8))) void k_shade_fast_each(DeviceScene sc, Work w, Frame f, int depth) {
    shade_bounce<FastMath, false, false, kFits>(sc, w, f, depth);
}

// ... whose first bounce computes the camera ray again instead of reading it back (k_generate's `regenerate`)
template <bool kFits>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(5, 8))) void k_shade_fast_first(DeviceScene sc, Work w, Frame f) {
    shade_bounce<FastMath, false, true, kFits>(sc, w, f, 0);
}

// The control flow of k_shade_fast with the arithmetic of k_shade: what the tests hold against the oracle bit for bit
// (RT_AMD_EXACT_STAYS, a validation knob; slower than k_shade, which has the registers for one trip only).
template <bool kFits>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(5, 8))) void k_shade_stays(DeviceScene sc, Work w, Frame f, int depth) {
    shade_bounce<ExactMath, true, false, kFits>(sc, w, f, depth);
}

// ---- scenes without triangles: the whole sample loop in registers -------
//
// Without a mesh no ray ever waits for the triangle stage, so nothing has to go through memory:
// a lane owns a PIXEL and runs the reference's own loop nest (samples, then bounces;
// src/raytrace.cu:478-500 and :317-442) with the path in registers, adding each sample's colour
// to the pixel's running sum in sample order.  No queue, no path state, no batches: one launch
// renders all samples of a band of rows.  The arithmetic per bounce is k_shade's (respond_to_hit
// and the same sphere scans), so the sums are the ones k_generate / k_shade / k_accumulate produce.
//
// The anti-aliasing draws of a group of samples are searched back to back, as in k_generate: kTraceGroup
// samples where a lane owns that many samples of a pixel, kTraceSearch where it owns the pixel (the longer
// the run of searches, the closer the lanes of a wavefront finish together: 3 % of a 1 000-sample frame).
constexpr uint32_t kTraceGroup = 4, kTraceSearch = 16;

template <class M, bool kFits>
__global__ __launch_bounds__(kBlock) void k_trace_spheres(DeviceScene sc, Frame f, uint32_t n_samples, float4* __restrict__ sum,
                                                          float4* __restrict__ rad, uint32_t* __restrict__ cast_count) {
    __shared__ SceneCacheT<kFits> cache;
    __shared__ uint32_t accepted[kTraceSearch][kBlock]; // first dimension of the accepted draw, per sample of the group
    __shared__ float4 eye_oc[kSphereCache];            // per cached sphere: OC = eye - centre, |OC| * |OC|
    stage_scene(sc, cache);
    if (threadIdx.x < min(sc.n_spheres, kSphereCache)) {
        const V3 oc = mk(f.pos[0], f.pos[1], f.pos[2]) - xyz(cache.sphere[threadIdx.x]);
        eye_oc[threadIdx.x] = make_float4(oc.x, oc.y, oc.z, oc_norm2_m<M>(oc));
    }
    __syncthreads();
    // rad == nullptr: a lane owns a pixel and all n_samples samples, and keeps the running sum.
    // Otherwise (frames of few pixels, which would not fill the chip that way): a lane owns
    // kTraceGroup consecutive samples of a pixel and writes each sample's colour and flags to
    // rad[sample * n_pixels + pixel]; k_accumulate then adds them up in sample order.
    const bool buffered = rad != nullptr;
    const uint32_t id = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t n_groups = (n_samples + kTraceGroup - 1) / kTraceGroup;
    const bool real = buffered ? id < f.n_pixels * n_groups : id < f.n_pixels; // lanes past the end do nothing
    const uint32_t px = real ? (buffered ? id % f.n_pixels : id) : 0;
    const uint32_t s_first = buffered && real ? (id / f.n_pixels) * kTraceGroup : 0;
    const uint32_t s_end = buffered ? min(s_first + kTraceGroup, n_samples) : n_samples;
    const uint32_t pixel = f.pixel_offset + px;
    const uint32_t ux = pixel % f.width, uy = pixel / f.width;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!buffered && real) acc = sum[pixel];
    // the reference's terminate_early: the sample loop of this pixel has ended (in an earlier call
    // if the count arrives negative, see k_accumulate)
    bool done = !real || acc.w < 0.f;
    const bool ended_before = done;
    uint32_t n_casts = 0;
    const V3 eye = mk(f.pos[0], f.pos[1], f.pos[2]);

    const uint32_t search = buffered ? kTraceGroup : kTraceSearch;
    for (uint32_t s0 = s_first; s0 < s_end; s0 += search) {
        if (__ballot(!done) == 0) break;
        const uint32_t group = min(search, s_end - s0);
        if (f.antialiasing) { // the accepted draw of every sample of the group (see k_generate)
            uint32_t r = 0, dim = 0;
            rt_rng_key key = rt_rng_make_key(f.seed, pixel, f.sample_begin + s0);
            while (r < group) {
                if (jitter_search<M>(key, dim)) {
                    accepted[r][threadIdx.x] = dim;
                    r++;
                    dim = 0;
                    if (r < group) key = rt_rng_make_key(f.seed, pixel, f.sample_begin + s0 + r);
                }
            }
        }
        for (uint32_t r = 0; r < group; r++) {
            if (__ballot(!done) == 0) break;
            const rt_rng_key rng = rt_rng_make_key(f.seed, pixel, f.sample_begin + s0 + r);
            // camera sample (reference src/raytrace.cu:479-494), as in k_generate
            uint32_t dim = 0;
            float dx = 0.f, dy = 0.f;
            if (f.antialiasing) {
                dim = accepted[r][threadIdx.x];
                jitter_m<M>(rng, dim, dx, dy);
                dim += 2;
            }
            const float fpx = (float)ux + 0.5f + dx;
            const float fpy = (float)uy + 0.5f + dy;
            const V3 dir = normalize_m<M>(mk(div_m<M>(fpx - f.cx, f.fx), div_m<M>(fpy - f.cy, f.fy), 1.f));
            V3 ray = mk(dot_m<M>(mk(f.rot[0], f.rot[1], f.rot[2]), dir),
                        dot_m<M>(mk(f.rot[3], f.rot[4], f.rot[5]), dir),
                        dot_m<M>(mk(f.rot[6], f.rot[7], f.rot[8]), dir));
            V3 start = eye;
            int sid = -1;
            float ts = INFINITY;
            {
                const uint32_t cached = min(sc.n_spheres, kSphereCache);
                for (uint32_t i = 0; i < cached; i++) {
                    const float4 e = eye_oc[i];
                    sphere_step<M>(dot_m<M>(ray, xyz(e)), e.w, cache.sphere[i].w, i, ts, sid);
                }
                for (uint32_t i = cached; i < sc.n_spheres; i++) {
                    const float4 sph = sc.sphere[i];
                    const V3 oc = start - xyz(sph);
                    sphere_step<M>(dot_m<M>(ray, oc), oc_norm2_m<M>(oc), sph.w, i, ts, sid);
                }
            }
            n_casts += (uint32_t)__popcll(__ballot(!done));

            // the bounce loop (reference get_color, src/raytrace.cu:317-442)
            V3 L = mk(0, 0, 0), weight = mk(1.f, 1.f, 1.f);
            uint32_t flags = 0;
            bool open = !done; // the path still bounces
            for (int depth = 0; depth < f.max_depth; depth++) {
                if (__ballot(open) == 0) break;
                bool alive = false, lit = false, tinted = false;
                V3 P = mk(0, 0, 0), N = mk(0, 0, 1), colour = mk(0, 0, 0);
                if (open && sid >= 0) {
                    float min_dist = INFINITY;
                    if (ts < min_dist) min_dist = ts;
                    P = isinf(min_dist) ? mk(INFINITY, INFINITY, INFINITY) : along_m<M>(start, min_dist, ray);
                    const int m = sphere_material(sc, cache, (uint32_t)sid);
                    N = normalize_m<M>(P - xyz(sphere_at(sc, cache, (uint32_t)sid)));
                    respond_to_hit<M>(sc, cache, f, rng, depth, m, P, N, start, ray, weight, colour, dim, flags, alive, lit, tinted);
                }
                // shadow rays of a diffuse hit and the bounce ray, as in k_shade
                ts = INFINITY;
                sid = -1;
                bool bounce_probed = false;
                for (uint32_t l = 0; l < sc.n_lights; l++) {
                    if (l > 0 && __ballot(lit) == 0) break;
                    const float4 li = light_at(sc, cache, l);
                    const V3 lp = xyz(li);
                    const V3 sp_ = lp - P;
                    V3 wi, sray;
                    float d, dist;
                    if constexpr (M::kExact) {
                        d = norm(sp_);
                        wi = (1.f / d) * sp_;
                        sray = normalize(lp - start);
                        dist = norm(lp - start);
                    } else {
                        const float d2 = fast_dot(sp_, sp_), rd = __builtin_amdgcn_rsqf(d2);
                        d = d2 * rd;
                        wi = rd * sp_;
                        const V3 ls = lp - start;
                        const float l2 = fast_dot(ls, ls), rl = __builtin_amdgcn_rsqf(l2);
                        dist = l2 * rl;
                        sray = rl * ls;
                    }
                    const float nwi = fmaxf(dot_m<M>(N, wi), 0.f);
                    float ds = INFINITY;
                    int shadow_sid = -1;
                    if (l == 0) {
                        nearest_sphere_pair<M>(sc, cache, start, lit, sray, dist, ds, shadow_sid, alive, ray, ts, sid);
                        bounce_probed = true;
                    } else if (lit) {
                        ds = nearest_sphere_shadow<M>(sc, cache, sray, start, dist, shadow_sid);
                    }
                    if (lit) {
                        const bool blocked = ds < dist;
                        if constexpr (M::kExact) {
                            const double q = (double)li.w / (4 * M_PI * M_PI * (double)d * (double)d);
                            const V3 direct = nwi * ((blocked ? 0.f : 1.f) * scale_d(q, colour));
                            L = L + weight * direct;
                        } else {
                            const float q = li.w * __builtin_amdgcn_rcpf((float)(4 * M_PI * M_PI) * d * d);
                            L = L + weight * ((blocked ? 0.f : nwi * q) * colour);
                        }
                    }
                }
                if (!bounce_probed && alive) ts = nearest_sphere<M>(sc, cache, ray, start, sid); // a scene without lights
                n_casts += (uint32_t)__popcll(__ballot(lit)) * sc.n_lights + (uint32_t)__popcll(__ballot(alive));
                if (tinted) weight = weight * colour;
                open = alive;
            }
            if (buffered) {
                if (real) rad[(size_t)(s0 + r) * f.n_pixels + px] = make_float4(L.x, L.y, L.z, __uint_as_float(flags));
            } else if (!done) {
                acc.x += L.x;
                acc.y += L.y;
                acc.z += L.z;
                acc.w += 1.f;
                if (!f.indirect && (flags & F_EARLY)) done = true;
            }
        }
    }
    if (!buffered && real && !ended_before) {
        if (done) acc.w = -acc.w;
        sum[pixel] = acc;
    }
    if (f.count_casts && n_casts && __lane_id() == 0) {
        const uint32_t shard = wave_index() % kCastShards;
        atomicAdd(&cast_count[(size_t)shard * kCountStride], n_casts);
    }
}

// Per pixel: add the batch's sample colours to the running sum in sample
// order, as the reference's sample loop does (src/raytrace.cu:478-499).
// sum[pixel] = (r, g, b, actual_rays).
__global__ __launch_bounds__(kBlock) void k_accumulate(DeviceScene sc, Work w, Frame f, float4* __restrict__ sum) {
    const uint32_t pixel = blockIdx.x * kBlock + threadIdx.x;
    if (pixel >= f.n_pixels) return;
    float4 acc = sum[f.pixel_offset + pixel];
    const uint32_t n_samples = f.n_paths / f.n_pixels;
    // A negative sample count marks a pixel whose sample loop the reference has ended
    // (terminate_early, src/raytrace.cu:340-346,496-498): later batches and sample ranges add nothing.
    if (acc.w < 0.f) return;
    // four records are requested at a time (the additions themselves stay in sample order)
    bool done = false;
    for (uint32_t s0 = 0; s0 < n_samples && !done; s0 += 4) {
        float4 rads[4];
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) rads[k] = w.rad[(size_t)min(s0 + k, n_samples - 1) * f.n_pixels + pixel];
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
            if (done || s0 + k >= n_samples) break;
            const uint32_t p = (s0 + k) * f.n_pixels + pixel;
            const uint32_t flags = __float_as_uint(rads[k].w);
            V3 L = mk(rads[k].x, rads[k].y, rads[k].z);
            if (flags & F_PENDING) L = settle_pending(sc, w, f.n_paths, p, L);
            acc.x += L.x;
            acc.y += L.y;
            acc.z += L.z;
            acc.w += 1.f;
            if (!f.indirect && (flags & F_EARLY)) done = true;
        }
    }
    if (done) acc.w = -acc.w;
    sum[f.pixel_offset + pixel] = acc;
}

// gamma((1/actual_rays) * colour) (reference src/raytrace.cu:44-51,500)
__global__ __launch_bounds__(kBlock) void k_resolve(const float4* __restrict__ sum, uint32_t n_pixels, float* __restrict__ out) {
    const uint32_t pixel = blockIdx.x * kBlock + threadIdx.x;
    if (pixel >= n_pixels) return;
    const float4 s = sum[pixel];
    const float inv = 1.f / fabsf(s.w); // the sign only says that the pixel's sample loop ended early
    out[3 * pixel + 0] = rt_gamma_encode(inv * s.x);
    out[3 * pixel + 1] = rt_gamma_encode(inv * s.y);
    out[3 * pixel + 2] = rt_gamma_encode(inv * s.z);
}

// dst += src, element by element (the sums of another device arriving on this one)
__global__ __launch_bounds__(kBlock) void k_add_sums(float4* __restrict__ dst, const float4* __restrict__ src, uint32_t n_vec4,
                                                     float* __restrict__ dst_tail, const float* __restrict__ src_tail, uint32_t n_tail) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n_vec4) {
        float4 a = dst[i];
        const float4 b = src[i];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        dst[i] = a;
    }
    if (i < n_tail) dst_tail[i] += src_tail[i];
}

// ------------------------------------------- known-answer test kernels

// random_cos for n (normal, r1, r2) triples: the function k_shade and k_trace_spheres call
__global__ __launch_bounds__(kBlock) void k_probe_random_cos(uint32_t n, const float* __restrict__ normals, const float* __restrict__ r1,
                                                             const float* __restrict__ r2, float* __restrict__ out) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const V3 d = random_cos(r1[i], r2[i], mk(normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]));
    out[3 * i] = d.x; out[3 * i + 1] = d.y; out[3 * i + 2] = d.z;
}

// rays -> state + queue, through the same launch_ray the renderer uses
__global__ __launch_bounds__(kBlock) void k_probe_rays(DeviceScene sc, Work w, Frame f, const float* __restrict__ origins,
                                                       const float* __restrict__ dirs, int as_segments) {
    __shared__ SceneCache cache;
    stage_scene(sc, cache);
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    const bool alive = p < f.n_paths;
    V3 start = mk(0, 0, 0), ray = mk(0, 0, 1);
    if (alive) {
        start = mk(origins[3 * p], origins[3 * p + 1], origins[3 * p + 2]);
        ray = mk(dirs[3 * p], dirs[3 * p + 1], dirs[3 * p + 2]);
        if (as_segments) { // `dirs` holds targets (reference is_visible)
            const V3 target = ray;
            ray = normalize(target - start);
            w.thr[p] = make_float4(norm(target - start), 0.f, 0.f, 0.f);
        }
    }
    if (as_segments) { // the shadow-ray scan of k_shade: its result only compares with the segment's length
        int sid = -1;
        float ts = INFINITY;
        bool walk = false;
        if (alive) {
            ts = nearest_sphere_shadow(sc, cache, ray, start, w.thr[p].x, sid);
            walk = needs_mesh<true>(sc, cache, f.bounding_box, ray, start);
        }
        commit_ray(w, 0, alive, p, start, ray, ts, sid, walk);
    } else {
        launch_ray(sc, cache, w, f, 0, alive, p, start, ray);
    }
}

__global__ __launch_bounds__(kBlock) void k_probe_results(DeviceScene sc, Work w, Frame f, int as_segments, float* __restrict__ out_t,
                                                          int32_t* __restrict__ out_sphere, int32_t* __restrict__ out_triangle,
                                                          float* __restrict__ out_abg, int32_t* __restrict__ out_visible) {
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= f.n_paths) return;
    const float4 ro4 = w.ray_o[p], rd4 = w.ray_o.second(p);
    const float ts = ro4.w;
    int sid = __float_as_int(rd4.w);
    const unsigned long long key = w.hit[p];
    const float tt = __uint_as_float((uint32_t)(key >> 32));
    int tid = (int)(uint32_t)key;
    if (as_segments) {
        const float dist = w.thr[p].x;
        out_visible[p] = (ts < dist || tt < dist) ? 0 : 1;
        return;
    }
    float min_dist = INFINITY;
    int tmp = tid;
    if (ts < min_dist) { min_dist = ts; tmp = tid; tid = -1; }
    if (tt < min_dist) { min_dist = tt; sid = -1; tid = tmp; }
    out_t[p] = min_dist;
    out_sphere[p] = sid;
    out_triangle[p] = tid;
    if (tid >= 0) {
        float a, b, g;
        barycentrics(sc, tid, xyz(rd4), xyz(ro4), a, b, g);
        out_abg[3 * p] = a; out_abg[3 * p + 1] = b; out_abg[3 * p + 2] = g;
    }
}

} // namespace
