// mesh_kernels.hip.h -- the triangle kernels: exhaustive scan (k_mesh) and cluster culling (k_bin64, then k_group_clusters; with k_bin_groups and k_plan_runs between them for meshes of more than 1 024 clusters).
// Part of the device half of librt_amd.so; included by pathtrace.hip only.
#pragma once

#include "device_scene.hip.h"

namespace {

// ------------------------------------------------------------- k_mesh

typedef float f2 __attribute__((ext_vector_type(2)));

// Nearest hit of a lane so far.  `t` is the ray's cut-off distance while
// index < 0 (nearest sphere hit for a bounce ray, light distance for a shadow
// ray: a triangle at or beyond it cannot change the outcome, because spheres
// win ties, src/raytrace.cu:203-212, and a shadow ray asks for t < distance,
// :240).  Once a triangle is held, `t` is its distance to within a few 1e-7
// (relative) and the exact value is the IEEE quotient num / den, which
// hit_distance() evaluates once, after the scan.  `index` is the triangle's
// POSITION in the scan order; tri_orig maps it to the caller's index.
struct Nearest {
    float t, num, den;
    int index;
};

// tri_orig[pos] through the vector memory path.  `pos` is wave-uniform inside
// the scan, so the compiler would use a scalar load, and waiting for a scalar
// load also waits for the prefetch of the next triangle pair (scalar loads
// return out of order: the only wait is "all of them").  Used on the rare
// exact-tie path only.
__device__ __forceinline__ int orig_index(const uint32_t* __restrict__ tri_orig, int pos) {
    const uint32_t* p = tri_orig + pos;
    uint32_t v;
    asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return (int)v;
}

__device__ __forceinline__ Nearest nearest_init(float cutoff) { Nearest n; n.t = cutoff; n.num = 0.f; n.den = 1.f; n.index = -1; return n; }
__device__ __forceinline__ float hit_distance(const Nearest& n) { return n.num / n.den; }

// Exact comparison (reference src/raytrace.cu:164-174): t = (A-O).N / u.N must
// be positive and nearer than the nearest so far; among equal distances the
// lower caller's triangle index wins, as in the reference's in-order scan.
__device__ __forceinline__ void take_if_nearer_exact(float ao_dot_n, float d, uint32_t pos, const uint32_t* __restrict__ tri_orig, Nearest& best) {
    const float t = ao_dot_n / d;
    const float held = best.index >= 0 ? hit_distance(best) : best.t;
    if (t > 0 && (t < held || (t == held && best.index >= 0 && orig_index(tri_orig, (int)pos) < orig_index(tri_orig, best.index)))) {
        best.t = t; best.num = ao_dot_n; best.den = d; best.index = (int)pos;
    }
}

// Exact tail of one ray-triangle test (reference src/raytrace.cu:161-174),
// entered only by lanes the division-free test could neither reject nor
// accept.
__device__ __forceinline__ void exact_hit(float nb, float ng, float d, float ao_dot_n, uint32_t pos, const uint32_t* __restrict__ tri_orig, Nearest& best) {
    const float beta = nb / d;
    const float gamma = -ng / d;
    const float alpha = 1.f - beta - gamma;
    if (alpha >= 0 && beta >= 0 && gamma >= 0) take_if_nearer_exact(ao_dot_n, d, pos, tri_orig, best);
}

// A lane the division-free test could not reject.  `lo` is the smallest of the
// approximate alpha, beta, gamma (each within 1e-6 of the exact value when the
// point is inside).  lo >= 1e-5 means the reference's exact values are all
// positive, so the hit is accepted without computing them.  Its approximate
// distance t_ = (A-O).N * rcp(u.N) is within 3e-7 (relative) of the exact
// quotient, as is the held one, so a factor 1 + 2e-6 between them decides the
// order without a division; a negative t_ has a non-positive exact quotient.
// Near-ties, values too small for the error bound and NaNs are decided
// exactly.  Anything closer to an edge takes exact_hit.
__device__ __forceinline__ void candidate(float lo, float nb, float ng, float d, float rcp, float ao_dot_n, uint32_t pos,
                                          const uint32_t* __restrict__ tri_orig, Nearest& best) {
    if (lo >= 1e-5f) {
        const float t_ = ao_dot_n * rcp;
        if (t_ > best.t * 1.000002f || t_ < 0.f) return;
        if (t_ * 1.000002f < best.t && t_ > 1e-30f) {
            best.t = t_; best.num = ao_dot_n; best.den = d; best.index = (int)pos;
        } else {
            take_if_nearer_exact(ao_dot_n, d, pos, tri_orig, best);
        }
    } else {
        exact_hit(nb, ng, d, ao_dot_n, pos, tri_orig, best);
    }
}

// Twelve packed values of one triangle pair: A.xyz, e1.xyz, e2.xyz, N.xyz.
struct PairData {
    f2 v[12];
};

__device__ __forceinline__ PairData load_pair(const f2* __restrict__ tri2, uint32_t j) {
    PairData p;
    const f2* __restrict__ g = tri2 + 12 * (size_t)j;
#pragma unroll
    for (int k = 0; k < 12; k++) p.v[k] = g[k];
    return p;
}

// One ray (lane) against the two triangles of a pair.  The numerators e2.q,
// e1.q and the denominator u.N are computed with the reference's operation
// order (src/raytrace.cu:155-162), two triangles at a time in packed fp32.
// The division-free test uses an approximate reciprocal: a hit the reference
// accepts has alpha, beta, gamma in [0, 1], where the approximations are off
// by < 1e-6, so the 1e-5 margin never drops one; a NaN passes.
__device__ __forceinline__ void test_pair(const PairData& p, const uint32_t* __restrict__ tri_orig, uint32_t pos, float ox, float oy, float oz,
                                          float ux, float uy, float uz, Nearest& best) {
    const f2 Ax = p.v[0], Ay = p.v[1], Az = p.v[2];
    const f2 e1x = p.v[3], e1y = p.v[4], e1z = p.v[5];
    const f2 e2x = p.v[6], e2y = p.v[7], e2z = p.v[8];
    const f2 Nx = p.v[9], Ny = p.v[10], Nz = p.v[11];
    const f2 aox = Ax - ox, aoy = Ay - oy, aoz = Az - oz;
    const f2 qx = aoy * uz - aoz * uy;
    const f2 qy = aoz * ux - aox * uz;
    const f2 qz = aox * uy - aoy * ux;
    const f2 d = (ux * Nx + uy * Ny) + uz * Nz;
    const f2 nb = (e2x * qx + e2y * qy) + e2z * qz; // e2 . q
    const f2 ng = (e1x * qx + e1y * qy) + e1z * qz; // e1 . q
    f2 rcp;
    rcp.x = __builtin_amdgcn_rcpf(d.x);
    rcp.y = __builtin_amdgcn_rcpf(d.y);
    const f2 b_ = nb * rcp;
    const f2 g_ = -ng * rcp;
    const f2 a_ = (1.f - b_) - g_;
    const float lo0 = fminf(fminf(a_.x, b_.x), g_.x);
    const float lo1 = fminf(fminf(a_.y, b_.y), g_.y);
    const bool c0 = !(lo0 < -1e-5f), c1 = !(lo1 < -1e-5f);
    if (c0 || c1) {
        // one divergent region for both triangles: a lane brings the first of its candidates
        // (lower position first, as in the scan), the rare second one follows
        const f2 aon = (aox * Nx + aoy * Ny) + aoz * Nz;
        if (c0) candidate(lo0, nb.x, ng.x, d.x, rcp.x, aon.x, pos, tri_orig, best);
        else    candidate(lo1, nb.y, ng.y, d.y, rcp.y, aon.y, pos + 1, tri_orig, best);
        if (c0 && c1) candidate(lo1, nb.y, ng.y, d.y, rcp.y, aon.y, pos + 1, tri_orig, best);
    }
}

// test_pair for a lane that meets this (ray, pair) once and reports every hit it finds instead of
// keeping the nearest: emit(h, numerator, denominator) is called for triangle h of the pair when the
// reference's test (src/raytrace.cu:161-174) accepts it with a distance t = numerator / denominator in
// (0, cutoff).  The caller merges hits of one ray with the atomic minimum on (t, caller's triangle index),
// which is the reference's "nearest, first triangle wins ties" whatever the order of arrival.  Same
// division-free filter and the same exact tail as test_pair / candidate; `live` = false masks the lane.
template <typename Emit>
__device__ __forceinline__ void pair_hits(const PairData& p, bool live, float ox, float oy, float oz, float ux, float uy, float uz,
                                          float cutoff, Emit&& emit) {
    const f2 Ax = p.v[0], Ay = p.v[1], Az = p.v[2];
    const f2 e1x = p.v[3], e1y = p.v[4], e1z = p.v[5];
    const f2 e2x = p.v[6], e2y = p.v[7], e2z = p.v[8];
    const f2 Nx = p.v[9], Ny = p.v[10], Nz = p.v[11];
    const f2 aox = Ax - ox, aoy = Ay - oy, aoz = Az - oz;
    const f2 qx = aoy * uz - aoz * uy;
    const f2 qy = aoz * ux - aox * uz;
    const f2 qz = aox * uy - aoy * ux;
    const f2 d = (ux * Nx + uy * Ny) + uz * Nz;
    const f2 nb = (e2x * qx + e2y * qy) + e2z * qz; // e2 . q
    const f2 ng = (e1x * qx + e1y * qy) + e1z * qz; // e1 . q
    f2 rcp;
    rcp.x = __builtin_amdgcn_rcpf(d.x);
    rcp.y = __builtin_amdgcn_rcpf(d.y);
    const f2 b_ = nb * rcp;
    const f2 g_ = -ng * rcp;
    const f2 a_ = (1.f - b_) - g_;
    const float lo0 = fminf(fminf(a_.x, b_.x), g_.x);
    const float lo1 = fminf(fminf(a_.y, b_.y), g_.y);
    const bool c0 = live && !(lo0 < -1e-5f), c1 = live && !(lo1 < -1e-5f);
    if (c0 || c1) {
        const f2 aon = (aox * Nx + aoy * Ny) + aoz * Nz;
        auto one = [&](int h, float lo, float nb_, float ng_, float d_, float rcp_, float aon_) {
            bool inside = lo >= 1e-5f; // the exact barycentrics are certainly positive
            if (!inside) {             // near an edge (or NaN): the reference's own quotients decide
                const float beta = nb_ / d_;
                const float gamma = -ng_ / d_;
                const float alpha = 1.f - beta - gamma;
                inside = alpha >= 0 && beta >= 0 && gamma >= 0;
            }
            if (inside) {
                // the approximate distance is within 3e-7 (relative) of the exact quotient: a factor 1 + 2e-6
                // decides 0 < t < cutoff without a division, anything closer (or NaN) is decided exactly
                const float t_ = aon_ * rcp_;
                bool valid = t_ * 1.000002f < cutoff && t_ > 1e-30f;
                if (!valid && !(t_ > cutoff * 1.000002f || t_ < 0.f)) {
                    const float t = aon_ / d_;
                    valid = t > 0 && t < cutoff;
                }
                if (valid) emit(h, aon_, d_);
            }
        };
        if (c0) one(0, lo0, nb.x, ng.x, d.x, rcp.x, aon.x);
        if (c1) one(1, lo1, nb.y, ng.y, d.y, rcp.y, aon.y);
    }
}

// The division-free filter of pair_hits alone: c0 / c1 say whether the ray's line may meet triangle 0 / 1 of the pair
// (the same packed arithmetic, so the same verdicts).
// kFused (the fast arithmetic, rt_options.arithmetic = 1): products are fused into the sums that follow them -- a third
// fewer instructions, numerators and denominator a few units in the last place of their terms away from the reference's.
// A ray that meets a triangle within that distance of its edge may then be judged the other way, which the fast
// arithmetic's tolerance covers and the exact arithmetic's bit-for-bit promise does not.
template <bool kFused>
__device__ __forceinline__ void pair_filter(const PairData& p, bool live, float ox, float oy, float oz, float ux, float uy, float uz,
                                            bool& c0, bool& c1) {
    const f2 Ax = p.v[0], Ay = p.v[1], Az = p.v[2];
    const f2 e1x = p.v[3], e1y = p.v[4], e1z = p.v[5];
    const f2 e2x = p.v[6], e2y = p.v[7], e2z = p.v[8];
    const f2 Nx = p.v[9], Ny = p.v[10], Nz = p.v[11];
    const f2 aox = Ax - ox, aoy = Ay - oy, aoz = Az - oz;
    f2 qx, qy, qz, d, nb, ng;
    if constexpr (kFused) {
        auto fma2 = [](f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); };
        qx = fma2(aoy, (f2)(uz), -(aoz * uy));
        qy = fma2(aoz, (f2)(ux), -(aox * uz));
        qz = fma2(aox, (f2)(uy), -(aoy * ux));
        d = fma2((f2)(ux), Nx, fma2((f2)(uy), Ny, uz * Nz));
        nb = fma2(e2x, qx, fma2(e2y, qy, e2z * qz)); // e2 . q
        ng = fma2(e1x, qx, fma2(e1y, qy, e1z * qz)); // e1 . q
    } else {
        qx = aoy * uz - aoz * uy;
        qy = aoz * ux - aox * uz;
        qz = aox * uy - aoy * ux;
        d = (ux * Nx + uy * Ny) + uz * Nz;
        nb = (e2x * qx + e2y * qy) + e2z * qz; // e2 . q
        ng = (e1x * qx + e1y * qy) + e1z * qz; // e1 . q
    }
    f2 rcp;
    rcp.x = __builtin_amdgcn_rcpf(d.x);
    rcp.y = __builtin_amdgcn_rcpf(d.y);
    const f2 b_ = nb * rcp;
    const f2 g_ = -ng * rcp;
    const f2 a_ = (1.f - b_) - g_;
    const float lo0 = fminf(fminf(a_.x, b_.x), g_.x);
    const float lo1 = fminf(fminf(a_.y, b_.y), g_.y);
    c0 = live && !(lo0 < -1e-5f);
    c1 = live && !(lo1 < -1e-5f);
}

// One (ray, triangle) that pair_filter let through, decided the way pair_hits decides it: the same expressions on
// one triangle's twelve numbers (t[0..12): A.xyz, e1.xyz, e2.xyz, N.xyz) in unpacked arithmetic, which rounds like
// the packed one.  emit(numerator, denominator) is called when the reference's test accepts the triangle with
// 0 < t = numerator / denominator < cutoff.
template <typename Emit>
__device__ __forceinline__ void triangle_hit(const float (&t)[12], bool live, float ox, float oy, float oz, float ux, float uy, float uz,
                                             float cutoff, Emit&& emit) {
    const float aox = t[0] - ox, aoy = t[1] - oy, aoz = t[2] - oz;
    const float qx = aoy * uz - aoz * uy;
    const float qy = aoz * ux - aox * uz;
    const float qz = aox * uy - aoy * ux;
    const float d = (ux * t[9] + uy * t[10]) + uz * t[11];
    const float nb = (t[6] * qx + t[7] * qy) + t[8] * qz; // e2 . q
    const float ng = (t[3] * qx + t[4] * qy) + t[5] * qz; // e1 . q
    const float rcp = __builtin_amdgcn_rcpf(d);
    const float b_ = nb * rcp;
    const float g_ = -ng * rcp;
    const float a_ = (1.f - b_) - g_;
    const float lo = fminf(fminf(a_, b_), g_);
    if (live && !(lo < -1e-5f)) {
        const float aon = (aox * t[9] + aoy * t[10]) + aoz * t[11];
        bool inside = lo >= 1e-5f; // the exact barycentrics are certainly positive
        if (!inside) {             // near an edge (or NaN): the reference's own quotients decide
            const float beta = nb / d;
            const float gamma = -ng / d;
            const float alpha = 1.f - beta - gamma;
            inside = alpha >= 0 && beta >= 0 && gamma >= 0;
        }
        if (inside) {
            const float t_ = aon * rcp;
            bool valid = t_ * 1.000002f < cutoff && t_ > 1e-30f;
            if (!valid && !(t_ > cutoff * 1.000002f || t_ < 0.f)) {
                const float tt = aon / d;
                valid = tt > 0 && tt < cutoff;
            }
            if (valid) emit(aon, d);
        }
    }
}

// Lane l of every wavefront holds shard l's ray count and the number of
// 256-ray blocks in the shards before it.
struct ShardMap {
    uint32_t my_count, before, n_ray_blocks;
};

__device__ __forceinline__ ShardMap load_shard_map(const uint32_t* __restrict__ q_count) {
    ShardMap m;
    const uint32_t lane = __lane_id();
    m.my_count = q_count[lane * kCountStride];
    const uint32_t my_blocks = (m.my_count + kBlock - 1) / kBlock;
    uint32_t incl = my_blocks;
#pragma unroll
    for (int step = 1; step < 64; step <<= 1) {
        const uint32_t up = __shfl_up(incl, step);
        if ((int)lane >= step) incl += up;
    }
    m.before = incl - my_blocks;
    m.n_ray_blocks = __shfl(incl, 63);
    return m;
}

// Queue position of this lane's ray in ray block `ray_block`; lanes past the
// end of the shard get the shard's last ray and active = false.
__device__ __forceinline__ uint32_t ray_position(const ShardMap& m, uint32_t ray_block, uint32_t shard_capacity, bool& active) {
    const uint32_t shard = (uint32_t)__popcll(__ballot(m.before <= ray_block)) - 1u; // `before` is non-decreasing
    const uint32_t shard_first = __shfl(m.before, (int)shard);
    const uint32_t count = __shfl(m.my_count, (int)shard);
    const uint32_t r = (ray_block - shard_first) * kBlock + threadIdx.x;
    active = r < count;
    return shard * shard_capacity + (active ? r : count - 1);
}

// The same for entry v of the walk through the shards (block v / kBlock, entry v % kBlock of that block), where the
// lanes of a wavefront ask for entries of [v_first, v_first + 64): of at most two blocks.  The map is read from LDS
// (what lane l of load_shard_map holds at [l]) rather than kept in two registers of a kernel that has none to spare.
// v_first is the same in every lane and so is the flow of control.
struct ShardTable {
    uint32_t count[kShards], before[kShards];
};
__device__ __forceinline__ uint32_t list_position(const ShardTable& t, uint32_t v_first, uint32_t v, uint32_t shard_capacity, bool& active) {
    const uint32_t block_a = v_first / kBlock;
    const uint32_t mine = t.before[__lane_id()];
    const uint32_t shard_a = (uint32_t)__popcll(__ballot(mine <= block_a)) - 1u; // `before` is non-decreasing
    const uint32_t shard_b = (uint32_t)__popcll(__ballot(mine <= block_a + 1u)) - 1u;
    const bool second = v / kBlock != block_a;
    const uint32_t shard = second ? shard_b : shard_a;
    const uint32_t shard_first = t.before[shard], count = t.count[shard];
    const uint32_t r = (v / kBlock - shard_first) * kBlock + v % kBlock;
    active = r < count;
    return shard * shard_capacity + (active ? r : 0u);
}

// Entries of its input that one wavefront of a kernel with paths that stay in their lanes takes (a multiple of 64):
// long stretches keep the wavefronts full for most of their life, short ones leave enough wavefronts to fill the device.
constexpr uint32_t kMaxStretch = 512;
__host__ __device__ inline uint32_t stretch_of(uint32_t n_in, uint32_t min_waves) {
    const uint32_t want = n_in / (64u * min_waves);
    return 64u * (want < 1u ? 1u : want > kMaxStretch / 64u ? kMaxStretch / 64u : want);
}
// (Measured on the 8.4 M paths of the headline frame, stretches of 256 / 512 / 1 024: 4 690 / 4 795 / 4 733 Mpaths/s;
// on 4.2 M and 2.1 M paths 512 is the best as well, so the lists of a million entries and more get the longest
// stretches and only shorter ones are spread thinner.)
constexpr uint32_t kMinStretchWaves = 2048; // 2 wavefronts per SIMD
// (Measured, 0 / 16 / 24 / 32 / 48 / 64: 5 130 / 5 185 / 5 193 / 5 205 / 5 205 / 5 206 Mpaths/s on the headline frame.)
constexpr uint32_t kKeepLanes = 32; // fewest paths a wavefront whose stretch is used up goes on with
// Batches below this many paths take the kernel with a launch per bounce (measured with three frames in flight, 1.05 M /
// 2.1 M paths: 0.31 / 0.53 ms per frame that way, 0.33 / 0.51 ms with paths that stay in their lanes).
constexpr uint32_t kStaysFromPaths = 1536u * 1024u;

// One ray per lane against triangle pairs [first, last): two pairs per trip,
// two SGPR sets.  Scalar loads return out of order, so the only wait available
// is "all of them": each set is therefore requested right after an explicit
// wait and just before the OTHER set is tested, which leaves exactly one
// request in flight during a test.  Indices past the end are clamped and
// their tests skipped.
__device__ __forceinline__ void scan_pairs(const f2* __restrict__ tri2, const uint32_t* __restrict__ tri_orig, uint32_t first,
                                           uint32_t last, float ox, float oy, float oz, float ux, float uy, float uz,
                                           Nearest& best) {
    PairData cur = load_pair(tri2, first);
    for (uint32_t j = first; j < last; j += 2) {
        __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0): `cur` has landed
        const PairData nxt = load_pair(tri2, min(j + 1, last - 1));
        __builtin_amdgcn_sched_barrier(0); // keep the request ahead of the test it overlaps
        test_pair(cur, tri_orig, 2 * j, ox, oy, oz, ux, uy, uz, best);
        __builtin_amdgcn_s_waitcnt(0xC07F); // `nxt` has landed
        cur = load_pair(tri2, min(j + 2, last - 1));
        __builtin_amdgcn_sched_barrier(0);
        if (j + 1 < last) test_pair(nxt, tri_orig, 2 * j + 2, ox, oy, oz, ux, uy, uz, best);
    }
}

// Every queued ray against every triangle (reference src/raytrace.cu:148-175).
//
// A lane owns a ray; the triangle index is wave-uniform, so triangle data
// arrives by scalar loads and stays in SGPRs.  Two triangles are tested per
// iteration with packed fp32 instructions (v_pk_mul_f32 / v_pk_add_f32 work on
// a register pair at the issue cost of one instruction): tri_pair holds every
// quantity as (triangle 2j, triangle 2j+1), the ray components are broadcast.
//
// A job is 256 rays x one chunk of triangle pairs.  The nearest hit per ray is
// merged across chunks with a 64-bit atomic minimum on (t bits, triangle),
// which orders by distance first and triangle index second: the reference's
// "first triangle wins ties".  Jobs are dealt so that the chunks of one ray
// block run on workgroups with equal blockIdx % 8, which share an L2.
__global__ __launch_bounds__(kBlock) void k_mesh(const f2* __restrict__ tri2, const uint32_t* __restrict__ tri_orig, uint32_t n_tri,
                                                 const float4* __restrict__ q_o, const float4* __restrict__ q_d,
                                                 const uint32_t* __restrict__ q_count, uint32_t shard_capacity,
                                                 unsigned long long* __restrict__ hit) {
    const ShardMap map = load_shard_map(q_count);
    if (map.n_ray_blocks == 0) return;
    const uint32_t n_pairs = (n_tri + 1) / 2;
    const uint32_t n_chunks = (n_pairs + kChunkPairs - 1) / kChunkPairs;
    const uint32_t xcd = blockIdx.x % 8;
    const uint32_t per_xcd = ((map.n_ray_blocks + 7) / 8) * n_chunks;
    for (uint32_t q = blockIdx.x / 8; q < per_xcd; q += gridDim.x / 8) {
        const uint32_t ray_block = (q / n_chunks) * 8 + xcd;
        if (ray_block >= map.n_ray_blocks) continue;
        const uint32_t chunk = q % n_chunks;
        bool active;
        const uint32_t at = ray_position(map, ray_block, shard_capacity, active);
        const float4 o4 = q_o[2 * (size_t)at];
        const float4 d4 = q_d[2 * (size_t)at];
        Nearest best = nearest_init(d4.w);
        const uint32_t first = chunk * kChunkPairs;
        scan_pairs(tri2, tri_orig, first, min(first + kChunkPairs, n_pairs), o4.x, o4.y, o4.z, d4.x, d4.y, d4.z, best);
        if (active && best.index >= 0) {
            const unsigned long long key = ((unsigned long long)__float_as_uint(hit_distance(best)) << 32) | tri_orig[best.index];
            atomicMin(&hit[__float_as_uint(o4.w)], key);
        }
    }
}

// ---- cluster culling ------------------------------------------------------
//
// The reference tests every triangle once any object box is hit
// (src/raytrace.cu:136-148).  A triangle can only be hit by a ray that passes
// through the (padded) bounds of the cluster of 16 or 32 spatially adjacent
// triangles it belongs to, before the ray's cut-off distance.  The clusters are
// put into groups of consecutive clusters (neighbours in space, the triangles
// being in k-d order) with the union of their bounds -- a two-level
// bounding-volume hierarchy walked breadth first:
//
//   k_bin64            every queued ray against the (at most 64) group boxes
//                      -> per-group lists of queue positions in global memory
//   k_group_clusters   meshes of up to 64 groups of 16 clusters: a workgroup
//                      bins a stretch of one group's list into that group's
//                      clusters in LDS and scans the clusters' triangles itself
//   three levels       more than 1 024 clusters: k_bin64 against up to 64 super-groups
//                      of groups, k_bin_groups (a super-group's list against its
//                      groups -> per-group lists), k_plan_runs, k_group_clusters
//   k_bin_groups +     (RT_AMD_LEVELS=2: large meshes on two levels, groups of up to 64
//   k_mesh_clusters    clusters) per-cluster lists in global memory, job offsets from
//                      the list lengths (plan_jobs), the same pair scan per job
//
// The image is the exhaustive scan's, bit for bit, unless the reference's own
// rounding makes it accept a triangle for a ray that misses the triangle by
// more than the padding (a grazing ray whose barycentrics are rounding noise);
// rt_options.cluster_culling = 0 selects the exhaustive scan.

// Index of the last entry of first[0..n) that is <= job (entries are
// non-decreasing; empty ranges share their successor's value).
// A 64-ary search by the whole wavefront (all lanes must call it with the same job): two LDS reads
// instead of the twelve dependent ones of a binary search over 4 096 entries.
__device__ __forceinline__ uint32_t owner_of_job(const uint32_t* first, uint32_t n, uint32_t job) {
    const uint32_t lane = __lane_id();
    const uint32_t coarse = lane * 64u;
    const uint32_t v = coarse < n ? first[coarse] : 0xffffffffu;
    const uint32_t block = (uint32_t)__popcll(__ballot(v <= job)) - 1u; // first[0] = 0 <= job
    const uint32_t fine = block * 64u + lane;
    const uint32_t u = fine < n ? first[fine] : 0xffffffffu;
    return block * 64u + (uint32_t)__popcll(__ballot(u <= job)) - 1u;
}

// first_job[i] = number of jobs (of job_entries list entries) of the lists before list i,
// first_job[n] = their total, from the list lengths (count[i * kCountStride]);
// n <= kMaxClusters.  Every workgroup of a kernel that walks the lists makes
// its own copy in LDS (`scan` is kBlock words of scratch): the lengths sit in
// L2 and the scan is a few microseconds, less than a launch of its own would
// cost.  All threads of the workgroup must call it.
__device__ __forceinline__ void plan_jobs(const uint32_t* __restrict__ count, uint32_t n, uint32_t* first_job, uint32_t* scan,
                                          uint32_t job_entries = kBlock) {
    const uint32_t t = threadIdx.x;
    const uint32_t per = (n + kBlock - 1) / kBlock; // consecutive lists per thread, at most 16
    uint32_t mine = 0;
    for (uint32_t k = 0; k < per; k++) {
        const uint32_t i = t * per + k;
        if (i < n) mine += (count[(size_t)i * kCountStride] + job_entries - 1) / job_entries;
    }
    scan[t] = mine;
    __syncthreads();
    for (uint32_t off = 1; off < kBlock; off <<= 1) {
        const uint32_t v = t >= off ? scan[t - off] : 0;
        __syncthreads();
        scan[t] += v;
        __syncthreads();
    }
    uint32_t run = scan[t] - mine;
    for (uint32_t k = 0; k < per; k++) {
        const uint32_t i = t * per + k;
        if (i < n) {
            first_job[i] = run;
            run += (count[(size_t)i * kCountStride] + job_entries - 1) / job_entries;
        }
    }
    if (t == kBlock - 1) first_job[n] = scan[t];
    __syncthreads();
}

// The hardware minimum / maximum: a NaN operand is dropped (minNum / maxNum).
// Written as instructions because the library functions first re-quiet every
// operand (v_max x, x), which doubles the cost of a slab test.

// One ray prepared for slab tests: 1/u, origin * (1/u), cut-off distance.
struct SlabRay {
    float ix, iy, iz, ox, oy, oz, cutoff;
};

__device__ __forceinline__ SlabRay slab_ray(float4 o4, float4 d4) {
    SlabRay r;
    // hardware reciprocals (one unit in the last place: nothing next to the padding of the bounds);
    // +-inf for an axis-parallel ray
    r.ix = __builtin_amdgcn_rcpf(d4.x); r.iy = __builtin_amdgcn_rcpf(d4.y); r.iz = __builtin_amdgcn_rcpf(d4.z);
    r.ox = o4.x * r.ix; r.oy = o4.y * r.iy; r.oz = o4.z * r.iz;
    // A ray (nearly) parallel to a slab has an infinite origin term, and bound * (1/u) - inf is -inf
    // for BOTH bounds of a box that straddles zero: the box would be dropped although the ray runs
    // inside the slab.  Such an axis is taken out of the test instead: with a NaN origin term both
    // slab parameters are NaN, which the minimum / maximum of the three axes ignore.  (The ray may
    // then pass outside the slab; the filter only has to keep every box a ray can hit.)
    const float nan = __uint_as_float(0x7fc00000u);
    if (!(fabsf(r.ox) < INFINITY)) r.ox = nan;
    if (!(fabsf(r.oy) < INFINITY)) r.oy = nan;
    if (!(fabsf(r.oz) < INFINITY)) r.oz = nan;
    r.cutoff = d4.w;
    return r;
}

// Slab tests of R rays per lane against box pairs [first, last), last - first
// <= 16: bit 2(k - first) + h of bits[i] is set when ray i is inside box h of
// pair k somewhere between its start and its cut-off distance.  The slab parameters come from packed
// FMAs (t = bound * (1/u) - origin * (1/u)); this is only a filter and need not
// follow the reference's rounding.  NaN parameters (the axes slab_ray took out
// of the test, or 0 * inf) are dropped by the minimum / maximum: that axis does
// not constrain the ray, which errs towards "touches".  The box
// data of an iteration (scalar loads) serves all R rays.
template <int R>
__device__ __forceinline__ void slab_bits(const f2* __restrict__ box_pair, uint32_t first, uint32_t last, const SlabRay (&ray)[R],
                                          uint32_t (&bits)[R]) {
#pragma unroll
    for (int i = 0; i < R; i++) bits[i] = 0;
    for (uint32_t k = first; k < last; k++) {
        const f2* __restrict__ b = box_pair + 6 * (size_t)k;
        const f2 b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3], b4 = b[4], b5 = b[5];
#pragma unroll
        for (int i = 0; i < R; i++) {
            const SlabRay& r = ray[i];
            const f2 ax = __builtin_elementwise_fma(b0, (f2)(r.ix), (f2)(-r.ox)), bx = __builtin_elementwise_fma(b3, (f2)(r.ix), (f2)(-r.ox));
            const f2 ay = __builtin_elementwise_fma(b1, (f2)(r.iy), (f2)(-r.oy)), by = __builtin_elementwise_fma(b4, (f2)(r.iy), (f2)(-r.oy));
            const f2 az = __builtin_elementwise_fma(b2, (f2)(r.iz), (f2)(-r.oz)), bz = __builtin_elementwise_fma(b5, (f2)(r.iz), (f2)(-r.oz));
            // entry, not before the ray's start; exit, not beyond its cut-off distance; the ray touches the box when
            // the interval is not empty.  The verdicts are shifted into bits[i] through the carry: the first box
            // tested ends up in the highest of the bits used.
            const float in0 = hw_max3(hw_min(ax.x, bx.x), hw_min(ay.x, by.x), hw_max(hw_min(az.x, bz.x), 0.f));
            const float out0 = hw_min3(hw_max(ax.x, bx.x), hw_max(ay.x, by.x), hw_min(hw_max(az.x, bz.x), r.cutoff));
            const float in1 = hw_max3(hw_min(ax.y, bx.y), hw_min(ay.y, by.y), hw_max(hw_min(az.y, bz.y), 0.f));
            const float out1 = hw_min3(hw_max(ax.y, bx.y), hw_max(ay.y, by.y), hw_min(hw_max(az.y, bz.y), r.cutoff));
            asm("v_cmp_ge_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
                "v_cmp_ge_f32 vcc, %3, %4\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
                : "+v"(bits[i]) : "v"(out0), "v"(in0), "v"(out1), "v"(in1) : "vcc");
        }
    }
    // box 2 (k - first) + h to bit 2 (k - first) + h
    const uint32_t used = 2 * (last - first);
#pragma unroll
    for (int i = 0; i < R; i++) bits[i] = used ? __brev(bits[i]) >> (32 - used) : 0u;
}

// The boxes (up to 64, as a bit mask) that S rays per lane enter before their
// cut-off distances; the rays are read from the queue at positions at[i] and
// a ray with !active[i] gets an empty mask.  The box data of an iteration
// (scalar loads) serves all S rays.
template <int S>
__device__ __forceinline__ void touched_boxes(const f2* __restrict__ box_pair, uint32_t n_boxes, const bool* active, const uint32_t* at,
                                              const float4* __restrict__ q_o, const float4* __restrict__ q_d, unsigned long long* mine) {
    SlabRay ray[S];
#pragma unroll
    for (int i = 0; i < S; i++) ray[i] = slab_ray(q_o[2 * (size_t)at[i]], q_d[2 * (size_t)at[i]]);
    const uint32_t n_pairs = (n_boxes + 1) / 2;
    uint32_t bits_lo[S], bits_hi[S]; // boxes 0-31 and 32-63
    slab_bits<S>(box_pair, 0, min(n_pairs, 16u), ray, bits_lo);
    slab_bits<S>(box_pair, 16, max(n_pairs, 16u), ray, bits_hi);
    // an odd box count leaves the second half of the last entry without a box: drop its bit
    const unsigned long long real = n_boxes >= 64 ? ~0ULL : (1ULL << n_boxes) - 1ULL;
#pragma unroll
    for (int i = 0; i < S; i++) mine[i] = active[i] ? ((((unsigned long long)bits_hi[i] << 32) | bits_lo[i]) & real) : 0ULL;
}

// Appends R queue positions per lane to the lists of the boxes in their masks:
// every lane walks the set bits of its masks twice -- to count per box in LDS
// (`slot`, 64 words), and, after the workgroup has reserved list space with ONE
// global atomic per box, to take its positions.  The reservations are what
// bounds the binning kernels: a counter sustains only 100-200 atomics per
// microsecond, so the more rays a workgroup brings per reservation, the better
// (as long as enough workgroups stay busy).  All threads of the workgroup must
// call it.
template <int R>
__device__ __forceinline__ void append_to_lists(uint32_t n_boxes, const unsigned long long (&mine)[R], const uint32_t (&at)[R],
                                                uint32_t* __restrict__ count, uint32_t* __restrict__ list, uint32_t list_capacity,
                                                uint32_t* slot) {
    if (threadIdx.x < 64) slot[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < R; i++)
        for (unsigned long long m = mine[i]; m; m &= m - 1) atomicAdd(&slot[(uint32_t)__ffsll((long long)m) - 1u], 1u);
    __syncthreads();
    if (threadIdx.x < n_boxes) {
        const uint32_t total = slot[threadIdx.x];
        slot[threadIdx.x] = total ? atomicAdd(&count[threadIdx.x * kCountStride], total) : 0;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < R; i++)
        for (unsigned long long m = mine[i]; m; m &= m - 1) {
            const uint32_t c = (uint32_t)__ffsll((long long)m) - 1u;
            list[(size_t)c * list_capacity + atomicAdd(&slot[c], 1u)] = at[i];
        }
    __syncthreads();
}

// Bins R rays per lane (R = 1 or a multiple of 2; slab tests two rays at a time).
template <int R>
__device__ __forceinline__ void bin_rays64(const f2* __restrict__ box_pair, uint32_t n_boxes, const bool (&active)[R], const uint32_t (&at)[R],
                                           const float4* __restrict__ q_o, const float4* __restrict__ q_d,
                                           uint32_t* __restrict__ count, uint32_t* __restrict__ list, uint32_t list_capacity,
                                           uint32_t* slot) {
    unsigned long long mine[R];
    if constexpr (R == 1) {
        touched_boxes<1>(box_pair, n_boxes, active, at, q_o, q_d, mine);
    } else {
        static_assert(R % 2 == 0, "rays per lane: 1 or even");
#pragma unroll
        for (int i = 0; i < R; i += 2) touched_boxes<2>(box_pair, n_boxes, active + i, at + i, q_o, q_d, mine + i);
    }
    append_to_lists<R>(n_boxes, mine, at, count, list, list_capacity, slot);
}

// First level: the queued rays of one tile (256-ray blocks [first_block,
// first_block + blocks_in_tile)) against the clusters of a mesh of at most 64
// clusters, or against the groups of a larger one.  A workgroup takes R
// consecutive ray blocks at a time (R rays per lane) and reserves list space
// once for all of them; fewer keep more workgroups busy when the queue is
// short.
template <int R>
__global__ __launch_bounds__(kBlock) void k_bin64(const f2* __restrict__ box_pair, uint32_t n_boxes,
                                                  const float4* __restrict__ q_o, const float4* __restrict__ q_d,
                                                  const uint32_t* __restrict__ q_count, uint32_t shard_capacity,
                                                  uint32_t first_block, uint32_t blocks_in_tile,
                                                  uint32_t* __restrict__ count, uint32_t* __restrict__ list, uint32_t list_capacity) {
    __shared__ uint32_t slot[64];
    const ShardMap map = load_shard_map(q_count);
    const uint32_t end = min(first_block + blocks_in_tile, map.n_ray_blocks);
    for (uint32_t ray_block = first_block + R * blockIdx.x; ray_block < end; ray_block += R * gridDim.x) {
        bool active[R];
        uint32_t at[R];
        at[0] = ray_position(map, ray_block, shard_capacity, active[0]);
#pragma unroll
        for (int i = 1; i < R; i++) {
            if (ray_block + i < end) {
                at[i] = ray_position(map, ray_block + i, shard_capacity, active[i]);
            } else {
                at[i] = at[0];
                active[i] = false;
            }
        }
        bin_rays64<R>(box_pair, n_boxes, active, at, q_o, q_d, count, list, list_capacity, slot);
    }
}

// Second level: the rays on each group's list against the (at most 64)
// clusters of that group.  A job is R x 256 list entries of one group (R rays per lane).
template <int R>
__global__ __launch_bounds__(kBlock) void k_bin_groups(const f2* __restrict__ cl_pair, uint32_t n_clusters, uint32_t n_groups, uint32_t group_clusters,
                                                       const float4* __restrict__ q_o, const float4* __restrict__ q_d,
                                                       const uint32_t* __restrict__ grp_count, const uint32_t* __restrict__ grp_list,
                                                       uint32_t* __restrict__ cl_count, uint32_t* __restrict__ cl_list, uint32_t list_capacity) {
    __shared__ uint32_t slot[64];
    __shared__ uint32_t first_job[64 + 1];
    __shared__ uint32_t scan[kBlock];
    plan_jobs(grp_count, n_groups, first_job, scan, R * kBlock);
    const uint32_t n_jobs = first_job[n_groups];
    for (uint32_t job = blockIdx.x; job < n_jobs; job += gridDim.x) {
        const uint32_t g = owner_of_job(first_job, n_groups, job);
        const uint32_t count = grp_count[g * kCountStride];
        bool active[R];
        uint32_t at[R];
#pragma unroll
        for (int i = 0; i < R; i++) {
            const uint32_t e = ((job - first_job[g]) * R + i) * kBlock + threadIdx.x;
            active[i] = e < count;
            at[i] = grp_list[(size_t)g * list_capacity + (active[i] ? e : count - 1)];
        }
        const uint32_t first_cluster = g * group_clusters; // even: a group starts a box pair
        bin_rays64<R>(cl_pair + 6 * (size_t)(first_cluster / 2), min(group_clusters, n_clusters - first_cluster), active, at, q_o, q_d,
                      cl_count + (size_t)first_cluster * kCountStride, cl_list + (size_t)first_cluster * list_capacity, list_capacity, slot);
    }
}

// ---- second level and triangle scan in one kernel ---------------------------
//
// For meshes of up to kFusedMaxGroups * kFusedGroup clusters the groups hold at most kFusedGroup
// clusters each, and the rays on a group's list are binned into that group's clusters and scanned
// against the clusters' triangles by the same workgroup, through LDS: no per-cluster lists in
// global memory, no global reservations, no job table, one kernel less per triangle stage.
//
// A workgroup of T threads takes a contiguous run of one group's list (a few passes of R * T
// entries).  A pass has two phases:
//
//   bin    every lane fetches its entries' rays, keeps them in LDS (32 bytes each) and tests them
//          against the group's cluster boxes: a 16-bit mask per entry, also in LDS;
//   scan   the wavefronts draw (cluster, half of the pass) units from a counter.  A unit's entries
//          are compacted into the wavefront's own list, and the wavefront then works as 8 triangle
//          pairs x 8 rays: a lane keeps ONE pair of the cluster in registers for the whole unit
//          (clusters of more than 8 pairs: one chunk of 8 after the other) and meets a new ray, read
//          from LDS, every trip -- 128 ray-triangle tests per trip, the packed arithmetic of k_mesh.
//
// Compared with a lane per ray and a wave-uniform triangle (k_mesh, where thousands of rays share
// one long triangle stream): no chain of dependent scalar loads per unit (a cluster is 8 steps
// long, and the three workgroups of a CU walk different groups, which defeats the scalar cache),
// rays come in eights instead of sixty-fours, so a unit of 37 rays wastes 3 slots instead of 27 and
// nothing has to be carried from pass to pass, and the rays are read from memory once per pass.
//
// A lane's hit is final for that (ray, pair): it goes to the ray's slot with the 64-bit atomic
// minimum on (t bits, caller's triangle index) that merges chunks and clusters anyway.  The exact
// quotient of an accepted hit is deferred (`pend`): a lane flushes it when its next hit arrives
// or the pass ends, so the division and the atomic run about once per pass, not once per trip.
struct PendingHit {
    float num, den;
    uint32_t orig;   // caller's index of the triangle; 0xffffffff: none pending
    uint32_t slot;
};

__device__ __forceinline__ void flush_hit(PendingHit& pend, unsigned long long* __restrict__ hit) {
    if (pend.orig != 0xffffffffu) {
        const unsigned long long key = ((unsigned long long)__float_as_uint(pend.num / pend.den) << 32) | pend.orig;
        atomicMin(&hit[pend.slot], key);
        pend.orig = 0xffffffffu;
    }
}

// The run table of k_group_clusters for meshes of more than kFusedMaxGroups groups (up to kMaxClusters of them), made once
// per launch by ONE workgroup instead of by every workgroup of the consumer -- the counters sit on a line each, and
// thousands of workgroups reading thousands of lines would cost more than the stage: plan[g] = runs of the groups before
// group g, plan[n_groups] = their total, plan[n_groups + 1] = entries per run (a multiple of pass_entries, such that every
// workgroup of a grid of consumer_grid gets about one run; the rule of k_group_clusters' own plan).
__global__ __launch_bounds__(1024) void k_plan_runs(const uint32_t* __restrict__ grp_count, uint32_t n_groups, uint32_t consumer_grid,
                                                    uint32_t pass_entries, uint32_t* __restrict__ plan) {
    constexpr uint32_t kPer = kMaxClusters / 1024; // consecutive groups per thread
    __shared__ uint32_t wave_sum[16];
    const uint32_t t = threadIdx.x, lane = __lane_id(), wave = t >> 6;
    uint32_t count[kPer], sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < kPer; k++) {
        const uint32_t i = t * kPer + k;
        count[k] = i < n_groups ? grp_count[(size_t)i * kCountStride] : 0u;
        sum += count[k];
    }
#pragma unroll
    for (int step = 1; step < 64; step <<= 1) sum += __shfl_xor(sum, step);
    if (lane == 0) wave_sum[wave] = sum;
    __syncthreads();
    uint32_t total = 0;
#pragma unroll
    for (uint32_t v = 0; v < 16; v++) total += wave_sum[v];
    uint32_t passes = (total + consumer_grid * pass_entries - 1) / (consumer_grid * pass_entries);
    passes = min(max(passes, 1u), 16u);
    const uint32_t run_entries = passes * pass_entries;
    uint32_t mine = 0;
#pragma unroll
    for (uint32_t k = 0; k < kPer; k++) mine += (count[k] + run_entries - 1) / run_entries;
    uint32_t incl = mine;
#pragma unroll
    for (int step = 1; step < 64; step <<= 1) {
        const uint32_t up = __shfl_up(incl, step);
        if ((int)lane >= step) incl += up;
    }
    __syncthreads(); // every thread has read the totals
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    uint32_t run = incl - mine;
    for (uint32_t v = 0; v < wave; v++) run += wave_sum[v];
#pragma unroll
    for (uint32_t k = 0; k < kPer; k++) {
        const uint32_t i = t * kPer + k;
        if (i < n_groups) plan[i] = run;
        run += (count[k] + run_entries - 1) / run_entries;
    }
    if (t == 1023) { // the groups past n_groups added nothing: the total
        plan[n_groups] = run;
        plan[n_groups + 1] = run_entries;
    }
}

// kMany: more than kFusedMaxGroups groups (three culling levels) -- the run table comes from k_plan_runs (`plan`, read
// from memory: a workgroup looks up the one or two runs it takes) instead of being computed by every workgroup.
// kFused: the filter of a trip in fused arithmetic (pair_filter; the fast arithmetic only).
template <int R, int T, int U, bool kMany, bool kFused>
__global__ __launch_bounds__(T) void k_group_clusters(const f2* __restrict__ tri2, const uint32_t* __restrict__ tri_orig, uint32_t n_tri,
                                                      uint32_t n_clusters, uint32_t cluster_pairs, uint32_t n_groups,
                                                      uint32_t group_clusters, const f2* __restrict__ cl_pair,
                                                      const float4* __restrict__ q_o, const float4* __restrict__ q_d,
                                                      const uint32_t* __restrict__ grp_count, const uint32_t* __restrict__ grp_list,
                                                      uint32_t list_capacity, unsigned long long* __restrict__ hit,
                                                      uint32_t* __restrict__ cl_stat, const uint32_t* __restrict__ plan) {
    constexpr uint32_t kPass = R * T;                    // list entries per pass
    constexpr uint32_t kWaves = T / 64;
    constexpr uint32_t kUnit = kPass / U;                // entries a scan unit looks at
    static_assert(kUnit % 64 == 0 && kPass <= 65536, "units are whole 64-entry blocks; entry numbers are 16 bits wide");
    __shared__ float4 ray_s[2 * kPass];                  // this pass's rays: origin + hit slot, direction + cut-off
    __shared__ uint16_t mask_s[kPass];                   // the clusters of the group each entry's ray touches
    __shared__ uint16_t refs_s[kWaves][kUnit];           // per wavefront: the entries of the unit it scans
    // per wavefront: the (ray, triangle) candidates the filter has let through and nobody has decided yet --
    // entry of the pass | (pair within the group, triangle of the pair) << 10
    constexpr uint32_t kCandCap = 256;
    static_assert(kPass <= 1024, "a candidate keeps its entry of the pass in ten bits");
    __shared__ uint32_t cand_s[kWaves][kCandCap];
    __shared__ uint32_t first_run[kMany ? 1 : kFusedMaxGroups + 1];  // runs of the groups before each group
    __shared__ uint32_t run_entries_s, next_unit;

    // ---- plan: the groups' lists are cut into runs of `run_entries` entries (a multiple of kPass),
    // sized so that every workgroup of the grid gets about one run; every workgroup computes the same plan
    const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6;
    if (!kMany && wave == 0) {
        const uint32_t mine = lane < n_groups ? grp_count[lane * kCountStride] : 0;
        uint32_t total = mine;
#pragma unroll
        for (int step = 1; step < 64; step <<= 1) total += __shfl_xor(total, step);
        uint32_t passes = (total + gridDim.x * kPass - 1) / (gridDim.x * kPass);
        passes = min(max(passes, 1u), 16u);
        const uint32_t run_entries = passes * kPass;
        const uint32_t runs = (mine + run_entries - 1) / run_entries;
        uint32_t incl = runs;
#pragma unroll
        for (int step = 1; step < 64; step <<= 1) {
            const uint32_t up = __shfl_up(incl, step);
            if ((int)lane >= step) incl += up;
        }
        if (lane < n_groups) first_run[lane] = incl - runs;
        if (lane == 63) first_run[n_groups] = incl; // lanes past n_groups add nothing: the total
        if (lane == 0) run_entries_s = run_entries;
    }
    __syncthreads();
    const uint32_t n_runs = kMany ? plan[n_groups] : first_run[n_groups], run_entries = kMany ? plan[n_groups + 1] : run_entries_s;
    const uint32_t n_pairs = (n_tri + 1) / 2;
    const uint32_t my_pair = lane & 7u, my_ray = lane >> 3;
    const unsigned long long below = (1ULL << lane) - 1ULL;

    for (uint32_t run = blockIdx.x; run < n_runs; run += gridDim.x) {
        // the group this run belongs to (first_run is non-decreasing; n_groups <= 64: one ballot)
        uint32_t g, runs_before;
        if constexpr (kMany) {
            g = owner_of_job(plan, n_groups, run);
            runs_before = plan[g];
        } else {
            const uint32_t fr = lane < n_groups ? first_run[lane] : 0xffffffffu;
            g = (uint32_t)__popcll(__ballot(fr <= run)) - 1u;
            runs_before = first_run[g];
        }
        const uint32_t count = grp_count[(size_t)g * kCountStride];
        const uint32_t run_begin = (run - runs_before) * run_entries;
        const uint32_t run_end = min(run_begin + run_entries, count);
        const uint32_t first_cluster = g * group_clusters; // even: a group starts a box pair
        const uint32_t boxes = min(group_clusters, n_clusters - first_cluster);
        const uint32_t real = (1u << boxes) - 1u;          // an odd count leaves half a box pair without a box
        const uint32_t* __restrict__ my_list = grp_list + (size_t)g * list_capacity;

        for (uint32_t pass = run_begin; pass < run_end; pass += kPass) {
            // ---- bin: this pass's rays into LDS, their cluster masks beside them
            SlabRay ray[R];
            bool active[R];
#pragma unroll
            for (int i = 0; i < R; i++) {
                const uint32_t e = pass + i * T + threadIdx.x;
                active[i] = e < run_end;
                const uint32_t pos = my_list[active[i] ? e : run_end - 1];
                const float4 o4 = q_o[2 * (size_t)pos], d4 = q_d[2 * (size_t)pos];
                ray_s[2 * (i * T + threadIdx.x)] = o4;
                ray_s[2 * (i * T + threadIdx.x) + 1] = d4;
                ray[i] = slab_ray(o4, d4);
            }
            uint32_t bits[R];
            slab_bits<R>(cl_pair + 6 * (size_t)(first_cluster / 2), 0, (boxes + 1) / 2, ray, bits);
#pragma unroll
            for (int i = 0; i < R; i++) mask_s[i * T + threadIdx.x] = (uint16_t)(active[i] ? (bits[i] & real) : 0u);
            if (threadIdx.x == 0) next_unit = 0;
            __syncthreads();

            // ---- scan: (cluster, part of the pass) units.  A wavefront works as 8 pairs x 8 rays.
            // The filter of a trip lets a few of its 128 (ray, triangle) tests through -- the ray's line meets the
            // triangle -- in nearly every trip; deciding them on the spot would run the rest of the test with a
            // handful of lanes per trip.  They are queued instead and decided 64 at a time, a lane each.
            const uint32_t n_units = U * boxes;
            const uint32_t group_first_pair = first_cluster * cluster_pairs;
            PendingHit pend;
            pend.num = 0.f; pend.den = 1.f; pend.orig = 0xffffffffu; pend.slot = 0;
            uint32_t n_cand = 0;
            auto decide = [&](uint32_t at, bool on) { // the candidate at cand_s[wave][at]
                const uint32_t code = cand_s[wave][at];
                const uint32_t e = code & 1023u, th = code >> 10;
                const size_t tri = 2 * (size_t)group_first_pair + th;
                const float4 o4 = ray_s[2 * e], d4 = ray_s[2 * e + 1];
                const float* __restrict__ g = reinterpret_cast<const float*>(tri2) + 24 * (tri >> 1) + (tri & 1u);
                float t[12];
#pragma unroll
                for (int k = 0; k < 12; k++) t[k] = g[2 * k];
                const uint32_t og = tri_orig[tri];
                triangle_hit(t, on, o4.x, o4.y, o4.z, d4.x, d4.y, d4.z, d4.w, [&](float num, float den) {
                    flush_hit(pend, hit);
                    pend.num = num; pend.den = den; pend.slot = __float_as_uint(o4.w);
                    pend.orig = og;
                });
            };
            for (;;) {
                uint32_t unit = 0;
                if (lane == 0) unit = atomicAdd(&next_unit, 1u);
                unit = (uint32_t)__builtin_amdgcn_readfirstlane((int)unit);
                if (unit >= n_units) break;
                const uint32_t c = unit / U, part = unit % U;
                const uint32_t first = (first_cluster + c) * cluster_pairs;
                const uint32_t last = min(first + cluster_pairs, n_pairs) - 1u; // the cluster's last pair
                // this lane's pair of the cluster's first eight, requested before the work that hides its latency;
                // lanes past the cluster's end repeat its last pair: the same hits once more, which the atomic minimum absorbs
                uint32_t j = min(first + my_pair, last);
                PairData pd = load_pair(tri2, j);
                uint2 orig = reinterpret_cast<const uint2*>(tri_orig)[j];
                // the unit's entries, in entry order, into this wavefront's list
                uint32_t n = 0;
#pragma unroll
                for (uint32_t b = 0; b < kUnit / 64; b++) {
                    const uint32_t e = part * kUnit + b * 64 + lane;
                    const bool touches = (mask_s[e] >> c) & 1u;
                    const unsigned long long m = __ballot(touches);
                    if (touches) refs_s[wave][n + (uint32_t)__popcll(m & below)] = (uint16_t)e;
                    n += (uint32_t)__popcll(m);
                }
                if (n == 0) continue;
                if (cl_stat && lane == 0) atomicAdd(&cl_stat[(size_t)(first_cluster + c) * kCountStride], n); // statistics
                for (uint32_t chunk = first;;) {
                    for (uint32_t k0 = 0; k0 < n; k0 += 8) {
                        const uint32_t k = k0 + my_ray;
                        const uint32_t e = refs_s[wave][min(k, n - 1u)];
                        const float4 o4 = ray_s[2 * e], d4 = ray_s[2 * e + 1];
                        bool c0, c1;
                        pair_filter<kFused>(pd, k < n, o4.x, o4.y, o4.z, d4.x, d4.y, d4.z, c0, c1);
                        const unsigned long long m0 = __ballot(c0), m1 = __ballot(c1);
                        if (m0 | m1) {
                            if (n_cand + 128u <= kCandCap) {
                                const uint32_t code = e | ((j - group_first_pair) << 11);
                                const uint32_t n0 = (uint32_t)__popcll(m0);
                                if (c0) cand_s[wave][n_cand + (uint32_t)__popcll(m0 & below)] = code;
                                if (c1) cand_s[wave][n_cand + n0 + (uint32_t)__popcll(m1 & below)] = code | 1024u;
                                n_cand += n0 + (uint32_t)__popcll(m1);
                            } else { // (the queue is full: a unit of many trips whose candidates come thick) decided on the spot
                                pair_hits(pd, k < n, o4.x, o4.y, o4.z, d4.x, d4.y, d4.z, d4.w, [&](int h, float num, float den) {
                                    flush_hit(pend, hit);
                                    pend.num = num; pend.den = den; pend.slot = __float_as_uint(o4.w);
                                    pend.orig = h ? orig.y : orig.x;
                                });
                            }
                        }
                    }
                    chunk += 8; // clusters of more than eight pairs: the next eight
                    if (chunk > last) break;
                    j = min(chunk + my_pair, last);
                    pd = load_pair(tri2, j);
                    orig = reinterpret_cast<const uint2*>(tri_orig)[j];
                }
                while (n_cand >= 64u) { // between units, when the trip's registers are free
                    n_cand -= 64u;
                    decide(n_cand + lane, true);
                }
            }
            if (n_cand) decide(min(lane, n_cand - 1u), lane < n_cand); // what is left belongs to this pass's rays
            flush_hit(pend, hit);
            __syncthreads();
        }
    }
}

// Statistics: the triangle tests of a batch -- over all stage slots, (rays that met a cluster) x (triangles of the cluster,
// cluster_tris except for the mesh's last one) -- added to *out.
__global__ __launch_bounds__(kBlock) void k_count_tests(const uint32_t* __restrict__ cl_count, size_t n_entries, uint32_t n_clusters,
                                                        uint32_t cluster_tris, uint32_t n_tri, unsigned long long* __restrict__ out) {
    unsigned long long sum = 0;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n_entries; i += (size_t)gridDim.x * kBlock) {
        const uint32_t c = (uint32_t)(i % n_clusters);
        sum += (unsigned long long)cl_count[i * kCountStride] * min(cluster_tris, n_tri - c * cluster_tris);
    }
#pragma unroll
    for (int step = 1; step < 64; step <<= 1) sum += __shfl_xor(sum, step);
    if (__lane_id() == 0 && sum) atomicAdd(out, sum);
}

__global__ __launch_bounds__(kBlock) void k_mesh_clusters(const f2* __restrict__ tri2, const uint32_t* __restrict__ tri_orig, uint32_t n_tri,
                                                          uint32_t n_clusters, uint32_t cluster_pairs,
                                                          const float4* __restrict__ q_o, const float4* __restrict__ q_d,
                                                          const uint32_t* __restrict__ cl_count,
                                                          const uint32_t* __restrict__ cl_list, uint32_t list_capacity,
                                                          unsigned long long* __restrict__ hit) {
    __shared__ uint32_t first_job[kMaxClusters + 1];
    __shared__ uint32_t scan[kBlock];
    plan_jobs(cl_count, n_clusters, first_job, scan);
    const uint32_t n_jobs = first_job[n_clusters];
    const uint32_t n_pairs = (n_tri + 1) / 2;
    // A job is a few hundred instructions, less than the three dependent accesses (job table, list
    // entry, ray) take, and a wavefront waits in order.  So the loop is a pipeline: in one trip it
    // requests the list entries of the job after next, requests the rays of the next job (whose list
    // entries were requested a trip ago) and scans the current job (whose rays were requested a trip ago).
    struct Where { // stage A: the job's list entries, requested
        uint32_t cluster, entry, count, at;
    };
    struct Job {   // stage B: the job's rays, requested
        uint32_t cluster;
        bool active;
        float4 o4, d4;
    };
    auto locate = [&](uint32_t job) {
        Where a;
        a.cluster = owner_of_job(first_job, n_clusters, job);
        a.entry = (job - first_job[a.cluster]) * kBlock + threadIdx.x;
        a.count = cl_count[a.cluster * kCountStride];
        // within the list's capacity (a multiple of kBlock) even past its end; such lanes are masked below
        a.at = cl_list[(size_t)a.cluster * list_capacity + a.entry];
        return a;
    };
    auto request = [&](const Where& a) {
        Job j;
        j.cluster = a.cluster;
        j.active = a.entry < a.count;
        const uint32_t at = j.active ? a.at : 0u; // lanes past the end read the queue's first record and drop the result
        j.o4 = q_o[2 * (size_t)at];
        j.d4 = q_d[2 * (size_t)at];
        return j;
    };
    uint32_t job = blockIdx.x;
    if (job >= n_jobs) return;
    Where where = locate(job);
    Job cur = request(where);
    job += gridDim.x;
    bool more = job < n_jobs;
    if (more) where = locate(job);
    for (;;) {
        Job next = cur;
        if (more) next = request(where);
        job += gridDim.x;
        const bool more_after = more && job < n_jobs;
        if (more_after) where = locate(job);
        Nearest best = nearest_init(cur.d4.w);
        const uint32_t first = cur.cluster * cluster_pairs;
        scan_pairs(tri2, tri_orig, first, min(first + cluster_pairs, n_pairs), cur.o4.x, cur.o4.y, cur.o4.z, cur.d4.x, cur.d4.y, cur.d4.z, best);
        if (cur.active && best.index >= 0) {
            const unsigned long long key = ((unsigned long long)__float_as_uint(hit_distance(best)) << 32) | tri_orig[best.index];
            atomicMin(&hit[__float_as_uint(cur.o4.w)], key);
        }
        if (!more) break;
        cur = next;
        more = more_after;
    }
}

} // namespace
