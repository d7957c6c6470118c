// device_scene.hip.h -- constants, the scene and work-buffer layouts in HBM, vector arithmetic in the reference's operation order, sphere and box intersection, the ray queue.
// Part of the device half of librt_amd.so; included by pathtrace.hip only.
#pragma once

#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>

#include "../../include/rt_amd.h"
#include "rt_math.h"

namespace {

constexpr uint32_t kBlock = 256;            // threads per workgroup (4 wavefronts)
constexpr uint32_t kWavesPerBlock = kBlock / 64;
constexpr uint32_t kChunkPairs = 256;       // triangle pairs per k_mesh job (512 triangles)
constexpr uint32_t kMeshGrid = 256 * 8;     // persistent k_mesh grid: 8 workgroups per CU
constexpr uint32_t kShards = 64;            // ray-queue shards = lanes of a wavefront
// Every atomically updated counter sits on a 128-byte line of its own: atomics
// to one line serialise, whatever word of it they touch.
constexpr uint32_t kCountStride = 32;
constexpr uint32_t kCastShards = 1024;      // statistics counters per path kernel: one atomic per wavefront, spread over 1024 lines
constexpr uint64_t kNoHit = 0x7f800000ffffffffULL; // (t = +inf, triangle = none)
constexpr uint32_t kMaxPathsPerBatch = 1u << 24;
constexpr uint32_t kMaxLights = 64;
constexpr uint32_t kClusterPairs = 8;       // triangle pairs per culling cluster of a small mesh (16 triangles; 8 triangles are 2 % faster on the cat
                                            // mesh in k-d order, but twice the groups double the provision for the per-group ray lists: +1.1 GB)
constexpr uint32_t kFusedClusterPairsMax = 100; // clusters grow with the mesh up to this size to stay within k_group_clusters' 1 024 clusters
constexpr uint32_t kMaxClusters = 64 * 64;  // two levels of at most 64 boxes each; larger meshes get larger clusters
constexpr uint32_t kFusedGroup = 16;        // clusters per group at most where k_group_clusters is used ...
constexpr uint32_t kFusedMaxGroups = 64;    // ... which is for meshes of up to 64 such groups
constexpr size_t kThreeLevelsFromPairs = kFusedMaxGroups * kFusedGroup * kClusterPairs + 1; // triangle pairs from which a mesh gets three culling levels (clusters of 16 triangles,
                                                 // groups of 16 clusters, up to 64 super-groups of up to 64 groups)
constexpr size_t kClusterListBudget = size_t(24) << 30; // most bytes the per-cluster ray lists may take (of 288 GB)

constexpr uint32_t kGenWalks = 0x80000000u;    // in Work::gen: the camera ray is queued for the first triangle stage
constexpr uint32_t kLivePending = 0x80000000u; // in an entry of the live-path list: the path has F_PENDING set

constexpr uint32_t F_ALIVE = 1u;   // the path still bounces
constexpr uint32_t F_PENDING = 2u; // pend[] holds light contributions of the last hit
constexpr uint32_t F_EARLY = 4u;   // the reference's terminate_early for this sample
constexpr uint32_t kBounceShift = 8; // flags >> kBounceShift: the path's bounce index (kept by the kernels whose paths run ahead of the launches)

constexpr float PEND_NONE = -2.f;   // slot unused
constexpr float PEND_ALWAYS = -1.f; // contribution known to be visible

// Structure-of-arrays scene in HBM.  Every array is read with a wave-uniform
// index except tri_shade / mat_* at shading time.
struct DeviceScene {
    const float4* sphere;     // centre.xyz, radius
    const int32_t* sphere_mat;
    const float4* mat_color;  // colour.xyz, type as int bits
    const float2* mat_ior;    // ro, ri
    const float4* tri_geom;   // 3 per triangle: A.xyz e1.x | e1.yz e2.xy | e2.z N.xyz
    const float2* tri_pair;   // 12 per PAIR of triangles (2j, 2j+1): A.x A.y A.z e1.xyz e2.xyz N.xyz,
                              // each as (value of 2j, value of 2j+1); k_mesh streams this one
    const uint32_t* tri_orig; // caller's index of the triangle at each position of tri_pair
                              // (triangles are stored in a k-d order of their centroids, see rt_scene_upload)
    const float2* cl_pair;    // padded bounds of each cluster of cluster_pairs consecutive pairs,
                              // 6 per PAIR of clusters (2k, 2k+1): lo.x lo.y lo.z hi.x hi.y hi.z
    const float2* grp_pair;   // the same for groups of group_clusters consecutive clusters (meshes of more than 64 clusters)
    const float4* tri_shade;  // 3 per triangle: n0.xyz material | n1.xyz - | n2.xyz -
    const float4* box_min;
    const float4* box_max;
    const float4* light;      // position.xyz, intensity
    uint32_t n_spheres, n_triangles, n_boxes, n_lights, n_materials;
    uint32_t n_clusters, cluster_pairs, n_groups, group_clusters;
    uint32_t fused_groups; // groups of at most 16 clusters: k_group_clusters bins and scans them in one kernel
};

// Per-batch path state and queues.
// One float4 of a record of kStride float4.
template <int kStride>
struct PathField {
    float4* base;
    __device__ __forceinline__ float4& operator[](size_t i) const { return base[kStride * i]; }
    // the float4 after it in the same record (the path kernels reach ray_d and q_d through ray_o and q_o: a pointer
    // less to keep in scalar registers, which they are short of)
    __device__ __forceinline__ float4& second(size_t i) const { return base[kStride * i + 1]; }
};

struct Work {
    // Per LIVE path, at its position in the list of live paths (two buffers of live_total entries:
    // bounce d reads buffer d & 1 at index (d & 1) * live_total + position and fills the other one;
    // the camera rays are at the path's own index in buffer 0).  What a bounce reads of its paths'
    // rays is therefore contiguous however many paths have ended.
    PathField<2> ray_o;   // origin.xyz, nearest-sphere distance
    PathField<2> ray_d;   // direction.xyz, nearest-sphere index as int bits   (one 32-byte record with ray_o)
    float4* thr;          // albedo.xyz, next random dimension as uint bits
    // Per path, at the path's index:
    float4* rad;     // colour L.xyz, F_* flags as uint bits
    float4* pend;    // n_lights per path: contribution.xyz, light distance or PEND_*
    // (t bits << 32) | triangle per ray: 2 * live_total slots for the bounce rays (indexed like ray_o),
    // then paths * n_lights slots for the shadow rays (path * n_lights + light)
    unsigned long long* hit;
    // the ray queue: 32-byte records (origin.xyz, hit slot as uint bits | direction.xyz, cut-off
    // distance), so that a gather through a cluster list touches one line per ray;
    // entry i is q_o[2 i] and q_d[2 i], with q_d = q_o + 1
    float4* q_o;
    float4* q_d;
    uint32_t* q_count; // kShards counters (kCountStride words apart) per k_mesh stage of a batch
    uint32_t* cast_count; // statistics: kCastShards counters (kCountStride words apart) per path kernel of a batch
    uint32_t* bounce_count; // statistics: kCastShards counters more, right after cast_count's: (path, bounce) pairs of a batch
    uint32_t shard_capacity; // queue entries reserved per shard
    // cluster culling: per stage slot, the queue positions of the rays that touch each cluster
    uint32_t* cl_count;      // n_clusters counters per slot
    uint32_t* cl_list;       // n_clusters lists of list_capacity queue positions (reused by every slot)
    uint32_t* grp_count;     // the same three for the groups of clusters
    uint32_t* grp_list;
    uint32_t list_capacity;  // = rays binned per tile
    // the paths that are still alive, for the next bounce: kShards shards like the ray queue,
    // two buffers (bounce d reads buffer d & 1 and fills the other)
    uint32_t* live;
    uint32_t* live_count;    // kShards counters per bounce
    // Per path, when the first bounce regenerates the camera rays (k_generate's `regenerate`) instead of reading them back:
    // first random dimension of the accepted anti-aliasing draw, kGenWalks set when the camera ray walks the mesh
    uint32_t* gen;
    uint32_t live_capacity;  // path indices reserved per shard
    uint32_t live_total;     // = kShards * live_capacity: entries of one buffer
};

struct V3 {
    float x, y, z;
};

__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
// scalar of type double: product in double, one rounding to float
// (reference include/vector.cuh:153-160 with S = double)
__device__ __forceinline__ V3 scale_d(double s, V3 a) {
    return mk((float)(s * (double)a.x), (float)(s * (double)a.y), (float)(s * (double)a.z));
}
// in-order dot product.  The reference starts the sum at 0.f
// (include/vector.cuh:179-183); 0.f + p equals p except that it turns -0 into
// +0, which no consumer of these dots can observe (they feed squares,
// comparisons against 0 and quotients that are rejected when the divisor is
// a zero of either sign).
__device__ __forceinline__ float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// the same with the explicit leading zero, for the few places that divide by
// the result of a dot with an axis vector
__device__ __forceinline__ float dot_z(V3 a, V3 b) { return ((0.f + a.x * b.x) + a.y * b.y) + a.z * b.z; }
// Correctly rounded square root -- the value sqrtf returns -- without the range scaling the
// compiler wraps around every sqrtf: v_sqrt_f32 is within one unit in the last place, and the
// exact residuals x - s * neighbour decide between s and its two neighbours.  Below 2^-96 the
// residuals would leave the normal range, so a wavefront with such an input (zero included)
// takes the library routine for all its lanes: both routes give the same value above it.
__device__ __forceinline__ float sqrt_rn(float x) {
    if (__ballot(x < 0x1p-96f) != 0) return sqrtf(x);
    float s = __builtin_amdgcn_sqrtf(x);
    const float down = __int_as_float(__float_as_int(s) - 1), up = __int_as_float(__float_as_int(s) + 1);
    const float r_down = __builtin_fmaf(-down, s, x), r_up = __builtin_fmaf(-up, s, x);
    s = r_down <= 0.f ? down : s;
    s = r_up > 0.f ? up : s;
    return s;
}

__device__ __forceinline__ float norm(V3 a) { return sqrt_rn((a.x * a.x + a.y * a.y) + a.z * a.z); }
__device__ __forceinline__ V3 normalize(V3 a) {
    const float n = norm(a);
    return mk(a.x / n, a.y / n, a.z / n);
}
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

// ---- the two arithmetics of the path kernels --------------------------------
//
// ExactMath is the reference's float arithmetic operation by operation: separately rounded products
// and sums, IEEE square roots and quotients, the double-precision scalars of src/raytrace.cu:358,360
// and the polynomial log / sin / cos of rt_math.h -- the image equals the CPU restatement's bit for bit.
// FastMath (rt_options.arithmetic = 1) computes the same expressions the way the hardware likes them:
// fused multiply-adds, v_rsq_f32 / v_rcp_f32 / v_sqrt_f32 (one unit in the last place each) instead of
// the IEEE expansions, |OC|^2 as a dot product instead of a square root squared, the light term in
// single precision, hardware log / sin / cos for the samplers -- as the reference itself does on its
// hardware (nvcc contracts by default; __logf, __cosf, __sinf, __powf in src/raytrace.cu:37-41,46-49).
// Every quantity is within a few 1e-7 (relative) of the exact one; pixels differ by more only where a
// ray's hit-or-miss decision flips at an edge.
struct ExactMath { static constexpr bool kExact = true; };
struct FastMath { static constexpr bool kExact = false; };

__device__ __forceinline__ float fast_dot(V3 a, V3 b) { return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)); }
template <class M> __device__ __forceinline__ float dot_m(V3 a, V3 b) {
    if constexpr (M::kExact) return dot(a, b); else return fast_dot(a, b);
}
template <class M> __device__ __forceinline__ float sqrt_m(float x) {
    if constexpr (M::kExact) return sqrt_rn(x); else return __builtin_amdgcn_sqrtf(x);
}
template <class M> __device__ __forceinline__ float div_m(float a, float b) {
    if constexpr (M::kExact) return a / b; else return a * __builtin_amdgcn_rcpf(b);
}
template <class M> __device__ __forceinline__ float norm_m(V3 a) {
    if constexpr (M::kExact) return norm(a); else return __builtin_amdgcn_sqrtf(fast_dot(a, a));
}
template <class M> __device__ __forceinline__ V3 normalize_m(V3 a) {
    if constexpr (M::kExact) {
        return normalize(a);
    } else {
        const float r = __builtin_amdgcn_rsqf(fast_dot(a, a));
        return mk(a.x * r, a.y * r, a.z * r);
    }
}
template <class M> __device__ __forceinline__ V3 cross_m(V3 a, V3 b) {
    if constexpr (M::kExact) return cross(a, b);
    else return mk(__builtin_fmaf(a.y, b.z, -(a.z * b.y)), __builtin_fmaf(a.z, b.x, -(a.x * b.z)), __builtin_fmaf(a.x, b.y, -(a.y * b.x)));
}
// a + s * b
template <class M> __device__ __forceinline__ V3 along_m(V3 a, float s, V3 b) {
    if constexpr (M::kExact) return a + s * b;
    else return mk(__builtin_fmaf(s, b.x, a.x), __builtin_fmaf(s, b.y, a.y), __builtin_fmaf(s, b.z, a.z));
}
// a + 0.01 * N with the reference's double-precision scalar (src/raytrace.cu:358,372,379)
template <class M> __device__ __forceinline__ V3 offset_m(V3 a, float sign, V3 N) {
    if constexpr (M::kExact) return sign > 0 ? a + scale_d(0.01, N) : a - scale_d(0.01, N);
    else return along_m<M>(a, sign * 0.01f, N);
}
// |OC|^2 as the reference computes it: the norm, squared (src/raytrace.cu:62-66)
template <class M> __device__ __forceinline__ float oc_norm2_m(V3 oc) {
    if constexpr (M::kExact) { const float n = norm(oc); return n * n; } else return fast_dot(oc, oc);
}
__device__ __forceinline__ V3 xyz(float4 v) { return mk(v.x, v.y, v.z); }

// ---------------------------------------------------------- intersection

// Scene data that every path reads with wave-uniform or small indices, staged
// in LDS by the path kernels.  An LDS read is a broadcast of about a hundred
// cycles; a global load in the middle of a kernel (which the compiler cannot
// turn into a scalar load in kernels that also write global memory) is a round
// trip to L2, and the few resident wavefronts of these register-heavy kernels
// cannot hide a dozen of those in a row.  Entries beyond the cached counts are
// read from global memory.
constexpr uint32_t kSphereCache = 64, kMatCache = 128, kLightCache = 8, kBoxCache = 8;

// kFits: the scene's tables are known to fit (the host checked the counts): no entry is ever fetched from global memory,
// and the kernels carry neither the code nor the pointers for it.
template <bool kFitsScene>
struct SceneCacheT {
    static constexpr bool kFits = kFitsScene;
    float4 sphere[kSphereCache];
    float4 mat_color[kMatCache];
    float2 mat_ior[kMatCache];
    int32_t sphere_mat[kSphereCache];
    float4 light[kLightCache];
    float4 box_min[kBoxCache], box_max[kBoxCache];
};
using SceneCache = SceneCacheT<false>;

// All threads of the workgroup must call it.  Every thread copies at most one entry of one
// array, so the whole cache costs one round trip to memory.
static_assert(kSphereCache + kMatCache + kLightCache + kBoxCache <= kBlock, "one thread per cached entry");
template <class Cache>
__device__ __forceinline__ void stage_scene(const DeviceScene& sc, Cache& c) {
    for (uint32_t first = 0; first < kBlock; first += blockDim.x) { // (one trip in a workgroup of kBlock threads)
        uint32_t i = first + threadIdx.x;
        if (i < kSphereCache) {
            if (i < sc.n_spheres) {
                c.sphere[i] = sc.sphere[i];
                c.sphere_mat[i] = sc.sphere_mat[i];
            }
        } else if ((i -= kSphereCache) < kMatCache) {
            if (i < sc.n_materials) {
                c.mat_color[i] = sc.mat_color[i];
                c.mat_ior[i] = sc.mat_ior[i];
            }
        } else if ((i -= kMatCache) < kLightCache) {
            if (i < sc.n_lights) c.light[i] = sc.light[i];
        } else if ((i -= kLightCache) < kBoxCache) {
            if (i < sc.n_boxes) {
                c.box_min[i] = sc.box_min[i];
                c.box_max[i] = sc.box_max[i];
            }
        }
    }
    __syncthreads();
}

// Single entries: the LDS read is unconditional (index wrapped into the cache) and the rare entry
// beyond the cache replaces it, so that the common case is a plain LDS read.
template <class Cache>
__device__ __forceinline__ float4 sphere_at(const DeviceScene& sc, const Cache& c, uint32_t i) {
    float4 v = c.sphere[i % kSphereCache];
    if (!Cache::kFits && i >= kSphereCache) v = sc.sphere[i];
    return v;
}
template <class Cache>
__device__ __forceinline__ int sphere_material(const DeviceScene& sc, const Cache& c, uint32_t i) {
    int v = c.sphere_mat[i % kSphereCache];
    if (!Cache::kFits && i >= kSphereCache) v = sc.sphere_mat[i];
    return v;
}
template <class Cache>
__device__ __forceinline__ float4 material_color(const DeviceScene& sc, const Cache& c, uint32_t m) {
    float4 v = c.mat_color[m % kMatCache];
    if (!Cache::kFits && m >= kMatCache) v = sc.mat_color[m];
    return v;
}
template <class Cache>
__device__ __forceinline__ float2 material_ior(const DeviceScene& sc, const Cache& c, uint32_t m) {
    float2 v = c.mat_ior[m % kMatCache];
    if (!Cache::kFits && m >= kMatCache) v = sc.mat_ior[m];
    return v;
}
template <class Cache>
__device__ __forceinline__ float4 light_at(const DeviceScene& sc, const Cache& c, uint32_t l) {
    float4 v = c.light[l % kLightCache];
    if (!Cache::kFits && l >= kLightCache) v = sc.light[l];
    return v;
}

// body(i, sphere i) for every sphere in index order: the cached ones from LDS, the rest from global memory.
template <typename Body, class Cache>
__device__ __forceinline__ void for_each_sphere(const DeviceScene& sc, const Cache& c, Body&& body) {
    const uint32_t cached = min(sc.n_spheres, kSphereCache);
    for (uint32_t i = 0; i < cached; i++) body(i, c.sphere[i]);
    if constexpr (!Cache::kFits)
        for (uint32_t i = cached; i < sc.n_spheres; i++) body(i, sc.sphere[i]);
}

// body(box_min o, box_max o) for every object box in index order.
template <typename Body, class Cache>
__device__ __forceinline__ void for_each_box(const DeviceScene& sc, const Cache& c, Body&& body) {
    const uint32_t cached = min(sc.n_boxes, kBoxCache);
    for (uint32_t o = 0; o < cached; o++) body(c.box_min[o], c.box_max[o]);
    if constexpr (!Cache::kFits)
        for (uint32_t o = cached; o < sc.n_boxes; o++) body(sc.box_min[o], sc.box_max[o]);
}

// One sphere of the reference's scan (get_distance, src/raytrace.cu:59-84)
// given b = u . OC and n2 = |OC| * |OC|.
template <class M = ExactMath>
__device__ __forceinline__ void sphere_step(float b, float n2, float R, uint32_t i, float& min_dist, int& id) {
    float delta;
    if constexpr (M::kExact) delta = b * b - n2 + R * R; else delta = __builtin_fmaf(R, R, __builtin_fmaf(b, b, -n2));
    if (delta >= 0) {
        const float sq = sqrt_m<M>(delta);
        const float t2 = -b + sq;
        if (t2 >= 0) {
            const float t1 = -b - sq;
            const float t = t1 >= 0 ? t1 : t2;
            if (id < 0 || t < min_dist) { min_dist = t; id = (int)i; }
        }
    }
}

// The same step for a SHADOW ray, of which the caller uses nothing but "is any sphere hit nearer
// than the light" (reference is_visible, src/raytrace.cu:218-244: get_distance(...) < distance to the
// light): a sphere that certainly has no root in [0, dist] cannot change that answer and is skipped
// without the exact square root; every other sphere takes sphere_step's arithmetic.  "Certainly":
// v_sqrt_f32 is within one unit in the last place, so the roots computed from it are within
// 3.1e-7 * (|b| + root of delta) of the exact code's; the band is 1e-6 * (|b| + root + dist)
// (+ 1e-15 for a subnormal delta), and a root passes when it is below -band or above dist + band --
// tested as |t - dist / 2| > dist / 2 + band, which is false for NaN or infinite operands (they
// take the exact step).  With both roots outside, the exact code finds t2 < 0 (no hit), or t1 > dist,
// or t1 < 0 and t2 > dist: never a hit nearer than the light.  On return min_dist < dist has the
// value it has after the exact scan; min_dist and id themselves are only those of the spheres
// that took the exact step.
template <class M = ExactMath>
__device__ __forceinline__ void sphere_step_shadow(float b, float n2, float R, uint32_t i, float dist, float half_dist, float& min_dist,
                                                   int& id) {
    if constexpr (!M::kExact) { // the hardware square root is the arithmetic itself: nothing to certify
        sphere_step<M>(b, n2, R, i, min_dist, id);
        return;
    }
    const float delta = b * b - n2 + R * R;
    if (delta >= 0) {
        const float root = __builtin_amdgcn_sqrtf(delta);
        const float h = half_dist + (1e-6f * ((fabsf(b) + root) + dist) + 1e-15f);
        const float t1 = -b - root, t2 = -b + root;
        const bool out1 = fabsf(t1 - half_dist) > h, out2 = fabsf(t2 - half_dist) > h;
        const bool clear = out1 && out2;
        if (!clear) {
            const float sq = sqrt_rn(delta);
            const float e2 = -b + sq;
            if (e2 >= 0) {
                const float e1 = -b - sq;
                const float t = e1 >= 0 ? e1 : e2;
                if (id < 0 || t < min_dist) { min_dist = t; id = (int)i; }
            }
        }
    }
}

// Nearest sphere along a ray (reference get_distance, src/raytrace.cu:53-87).
template <class M = ExactMath, class Cache>
__device__ __forceinline__ float nearest_sphere(const DeviceScene& sc, const Cache& c, V3 ray, V3 start, int& id) {
    float min_dist = INFINITY;
    for_each_sphere(sc, c, [&](uint32_t i, float4 s) {
        const V3 oc = start - xyz(s);
        sphere_step<M>(dot_m<M>(ray, oc), oc_norm2_m<M>(oc), s.w, i, min_dist, id);
    });
    return min_dist;
}

// The scan of a shadow ray towards a light at distance light_dist: the returned value compares with
// light_dist (<) as nearest_sphere's does, and says nothing else (sphere_step_shadow).
template <class M = ExactMath, class Cache>
__device__ __forceinline__ float nearest_sphere_shadow(const DeviceScene& sc, const Cache& c, V3 ray, V3 start, float light_dist, int& id) {
    float min_dist = INFINITY;
    const float half = 0.5f * light_dist;
    for_each_sphere(sc, c, [&](uint32_t i, float4 s) {
        const V3 oc = start - xyz(s);
        sphere_step_shadow<M>(dot_m<M>(ray, oc), oc_norm2_m<M>(oc), s.w, i, light_dist, half, min_dist, id);
    });
    return min_dist;
}

// One scan for two rays that leave the same point -- the shadow ray (a) towards a light at distance
// light_dist and the bounce ray (b) of one hit: OC and its norm depend on the start only and are
// computed once per sphere.  A ray whose flag is false keeps (inf, id); dist_a is good for the
// comparison with light_dist only (sphere_step_shadow).
template <class M = ExactMath, class Cache>
__device__ __forceinline__ void nearest_sphere_pair(const DeviceScene& sc, const Cache& c, V3 start, bool do_a, V3 ray_a,
                                                    float light_dist, float& dist_a, int& id_a, bool do_b, V3 ray_b, float& dist_b,
                                                    int& id_b) {
    dist_a = INFINITY;
    dist_b = INFINITY;
    const float half = 0.5f * light_dist;
    for_each_sphere(sc, c, [&](uint32_t i, float4 s) {
        const V3 oc = start - xyz(s);
        const float n2 = oc_norm2_m<M>(oc);
        if (do_a) sphere_step_shadow<M>(dot_m<M>(ray_a, oc), n2, s.w, i, light_dist, half, dist_a, id_a);
        if (do_b) sphere_step<M>(dot_m<M>(ray_b, oc), n2, s.w, i, dist_b, id_b);
    });
}

// Hardware minimum / maximum without the canonicalising extra instruction fminf / fmaxf bring.
// (A NaN operand is dropped in favour of the other one.)
__device__ __forceinline__ float hw_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float hw_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float hw_min3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float hw_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float hw_max3_abs(float a, float b, float c) { float r; asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// One box of the scan below for a ray, given lo = box_min - start, hi = box_max - start.
__device__ __forceinline__ bool box_hit(V3 lo, V3 hi, V3 ray) {
    const V3 axis[3] = {mk(1, 0, 0), mk(0, 1, 0), mk(0, 0, 1)};
    float t0[3] = {0.f, 0.f, 0.f}, t1[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const float rdn = dot_z(ray, axis[i]);
        if (rdn != 0) {
            const float a = dot_z(lo, axis[i]) / rdn;
            const float b = dot_z(hi, axis[i]) / rdn;
            if (a < b) { t0[i] = a; t1[i] = b; }
            else       { t0[i] = b; t1[i] = a; }
        }
    }
    // running min / max seeded with element 0 (reference include/vector.cuh:69-85)
    float mn = t1[0];
    if (t1[1] < mn) mn = t1[1];
    if (t1[2] < mn) mn = t1[2];
    float mx = t0[0];
    if (t0[1] > mx) mx = t0[1];
    if (t0[2] > mx) mx = t0[2];
    return mn > mx;
}

// The verdict of box_hit from reciprocals instead of IEEE quotients: 1 hit, 0 miss, -1 too close
// to call.  A quotient a * rcp(r) is within 2.4e-7 (relative) of the IEEE quotient a / r, so
// min(exit) - max(entry) is off by less than 1e-6 times the largest of the six quotients; a gap
// larger than that has the sign the exact arithmetic gives it.  Non-finite inputs or quotients
// make the band NaN or infinite and so end in "too close to call", which the caller answers with
// box_hit.  finite_probe is any sum over the components of lo and hi (NaN or inf if one of them is).
__device__ __forceinline__ int box_verdict(V3 lo, V3 hi, V3 ray, float finite_probe) {
    const float ix = __builtin_amdgcn_rcpf(ray.x), iy = __builtin_amdgcn_rcpf(ray.y), iz = __builtin_amdgcn_rcpf(ray.z);
    float ax = lo.x * ix, bx = hi.x * ix, ay = lo.y * iy, by = hi.y * iy, az = lo.z * iz, bz = hi.z * iz;
    if (ray.x == 0) ax = bx = 0.f; // a ray parallel to a slab leaves that axis at [0, 0]
    if (ray.y == 0) ay = by = 0.f;
    if (ray.z == 0) az = bz = 0.f;
    const float mn = hw_min3(hw_max(ax, bx), hw_max(ay, by), hw_max(az, bz));
    const float mx = hw_max3(hw_min(ax, bx), hw_min(ay, by), hw_min(az, bz));
    const float largest = hw_max(hw_max3_abs(ax, bx, ay), hw_max3_abs(by, az, bz));
    const float probe = ((ray.x + ray.y) + ray.z) + finite_probe;
    const float band = (1e-6f * largest + 1e-30f) + 0.f * probe;
    const float gap = mn - mx;
    if (!(fabsf(gap) > band)) return -1;
    return gap > 0.f ? 1 : 0;
}

// box_hit with reciprocals (FastMath): the same slabs, the same [0, 0] for an axis the ray is parallel to, the
// same verdict min(exit) > max(entry); quotients within 2.4e-7 of the IEEE ones.
__device__ __forceinline__ bool box_hit_fast(V3 lo, V3 hi, V3 ray) {
    const float ix = __builtin_amdgcn_rcpf(ray.x), iy = __builtin_amdgcn_rcpf(ray.y), iz = __builtin_amdgcn_rcpf(ray.z);
    float ax = lo.x * ix, bx = hi.x * ix, ay = lo.y * iy, by = hi.y * iy, az = lo.z * iz, bz = hi.z * iz;
    if (ray.x == 0) ax = bx = 0.f;
    if (ray.y == 0) ay = by = 0.f;
    if (ray.z == 0) az = bz = 0.f;
    const float mn = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    const float mx = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    return mn > mx;
}

// (Out of line: the rare exact answer must not cost the callers registers.)
__device__ __attribute__((noinline)) bool box_hit_out_of_line(float lx, float ly, float lz, float hx, float hy, float hz, float rx,
                                                              float ry, float rz) {
    return box_hit(mk(lx, ly, lz), mk(hx, hy, hz), mk(rx, ry, rz));
}

__device__ __forceinline__ bool box_hit_via_verdict(V3 lo, V3 hi, V3 ray, float finite_probe) {
    const int v = box_verdict(lo, hi, ray, finite_probe);
    if (v < 0) return box_hit_out_of_line(lo.x, lo.y, lo.z, hi.x, hi.y, hi.z, ray.x, ray.y, ray.z); // rare
    return v != 0;
}

// Slab test against every box until one is hit (reference
// src/raytrace.cu:103-145): per axis the entry/exit parameters stay 0 when
// the ray is parallel to the slab, there is no t > 0 requirement, and the
// verdict is min(exit) > max(entry).
// kViaVerdict: decide by box_verdict where that is certain (fewer instructions, more registers).
template <bool kViaVerdict = false, class M = ExactMath, class Cache>
__device__ __forceinline__ bool any_box_hit(const DeviceScene& sc, const Cache& c, V3 ray, V3 start) {
    bool hit = false;
    for_each_box(sc, c, [&](float4 lo, float4 hi) {
        if (!hit) {
            const V3 l = xyz(lo) - start, h = xyz(hi) - start;
            if constexpr (!M::kExact) hit = box_hit_fast(l, h, ray);
            else if (kViaVerdict) hit = box_hit_via_verdict(l, h, ray, ((l.x + l.y) + (l.z + h.x)) + (h.y + h.z));
            else hit = box_hit(l, h, ray);
        }
    });
    return hit;
}

// Appends the rays of the lanes with want == true to the k_mesh queue: one
// atomic per wavefront, slots in lane order.  The queue is split into kShards
// shards with a counter each and a wavefront always uses the shard of its own
// index, which spreads the returning atomics over 64 addresses (one address
// serialises at well under 200 atomics per microsecond).
// Returns the queue position of this lane's ray (0 when want == false).
// (the wavefront's index in the grid, for workgroups of any size)
__device__ __forceinline__ uint32_t wave_index() { return blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); }
__device__ __forceinline__ uint32_t wave_shard() { return wave_index() % kShards; }

__device__ __forceinline__ uint32_t enqueue(const Work& w, uint32_t counter, uint32_t shard, bool want, V3 start, V3 ray, uint32_t slot, float cutoff) {
    const unsigned long long mask = __ballot(want);
    if (mask == 0) return 0;
    const uint32_t lane = __lane_id();
    const uint32_t leader = (uint32_t)__ffsll((long long)mask) - 1u;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(&w.q_count[(counter * kShards + shard) * kCountStride], (uint32_t)__popcll(mask));
    base = __shfl(base, (int)leader);
    uint32_t at = 0;
    if (want) {
        at = shard * w.shard_capacity + base + (uint32_t)__popcll(mask & ((1ULL << lane) - 1ULL));
        w.q_o[2 * (size_t)at] = make_float4(start.x, start.y, start.z, __uint_as_float(slot));
        w.q_o[2 * (size_t)at + 1] = make_float4(ray.x, ray.y, ray.z, cutoff);
    }
    return at;
}

// Index of the first shadow-ray slot in Work::hit.
__device__ __forceinline__ size_t shadow_base(const Work& w) { return 2 * (size_t)w.live_total; }

// The end of a bounce for a wavefront, ALL of whose lanes must call it.  A path that goes on
// (alive) gets a position in the list of live paths that bounce `counter` reads: its list entry,
// its bounce ray (start, ray_b, nearest sphere ts_b / sid_b) and an empty hit slot are written
// there, and the ray is queued for triangle stage `counter` if it has to walk the mesh (walk_b).
// A shadow ray (want_a) is queued with it.  The two reservations are issued together by lane 0,
// so the wavefront waits for one round trip to the counters instead of three.
__device__ __forceinline__ void finish_bounce(const Work& w, uint32_t counter, uint32_t shard, bool want_a, V3 start, V3 ray_a, uint32_t slot_a,
                                              float cutoff_a, bool alive, bool walk_b, V3 ray_b, float ts_b, int sid_b, float4 thr,
                                              uint32_t entry) {
    const bool want_b = alive && walk_b;
    const unsigned long long mask_a = __ballot(want_a), mask_b = __ballot(want_b), mask_l = __ballot(alive);
    const uint32_t n_a = (uint32_t)__popcll(mask_a), n_b = (uint32_t)__popcll(mask_b), n_l = (uint32_t)__popcll(mask_l);
    const uint32_t lane = __lane_id();
    uint32_t base_q = 0, base_l = 0;
    if (lane == 0) {
        if (n_a + n_b) base_q = atomicAdd(&w.q_count[(counter * kShards + shard) * kCountStride], n_a + n_b);
        if (n_l) base_l = atomicAdd(&w.live_count[(counter * kShards + shard) * kCountStride], n_l);
    }
    base_q = __shfl(base_q, 0);
    base_l = __shfl(base_l, 0);
    const unsigned long long below = (1ULL << lane) - 1ULL;
    if (want_a) {
        const size_t at = shard * w.shard_capacity + base_q + (uint32_t)__popcll(mask_a & below);
        w.q_o[2 * at] = make_float4(start.x, start.y, start.z, __uint_as_float(slot_a));
        w.q_o[2 * at + 1] = make_float4(ray_a.x, ray_a.y, ray_a.z, cutoff_a);
    }
    if (alive) {
        const uint32_t index = (counter & 1u) * w.live_total + shard * w.live_capacity + base_l + (uint32_t)__popcll(mask_l & below);
        w.live[index] = entry;
        w.ray_o[index] = make_float4(start.x, start.y, start.z, ts_b);
        w.ray_o.second(index) = make_float4(ray_b.x, ray_b.y, ray_b.z, __int_as_float(sid_b));
        w.thr[index] = thr;
        w.hit[index] = kNoHit;
        if (walk_b) {
            const size_t at = shard * w.shard_capacity + base_q + n_a + (uint32_t)__popcll(mask_b & below);
            w.q_o[2 * at] = make_float4(start.x, start.y, start.z, __uint_as_float(index));
            w.q_o[2 * at + 1] = make_float4(ray_b.x, ray_b.y, ray_b.z, ts_b);
        }
    }
}

// Statistics: adds n (the wavefront's number of rays tested against the
// spheres in this kernel) to one of the kernel's counters.  All lanes pass the
// same n.
__device__ __forceinline__ void add_casts(const Work& w, uint32_t counter, uint32_t n) {
    if (n && __lane_id() == 0) {
        const uint32_t shard = wave_index() % kCastShards;
        atomicAdd(&w.cast_count[((size_t)counter * kCastShards + shard) * kCountStride], n);
    }
}

// Statistics: adds n (the wavefront's number of paths that were alive on entry to a bounce) to the batch's count of
// (path, bounce) pairs.  All lanes pass the same n.
__device__ __forceinline__ void add_bounces(const Work& w, uint32_t n) {
    if (n && __lane_id() == 0) {
        const uint32_t shard = wave_index() % kCastShards;
        atomicAdd(&w.bounce_count[(size_t)shard * kCountStride], n);
    }
}

// True when the ray has to walk the triangle list (reference
// src/raytrace.cu:103-146: with boxes enabled only after a box hit).
template <bool kViaVerdict = false, class M = ExactMath, class Cache>
__device__ __forceinline__ bool needs_mesh(const DeviceScene& sc, const Cache& c, int use_boxes, V3 ray, V3 start) {
    if (sc.n_triangles == 0) return false;
    if (!use_boxes) return true;
    return any_box_hit<kViaVerdict, M>(sc, c, ray, start);
}

// needs_mesh for two rays that leave the same point; box - start is shared.
template <class M = ExactMath, class Cache>
__device__ __forceinline__ void needs_mesh_pair(const DeviceScene& sc, const Cache& c, int use_boxes, V3 start, bool do_a, V3 ray_a,
                                                bool& walk_a, bool do_b, V3 ray_b, bool& walk_b) {
    walk_a = false;
    walk_b = false;
    if (sc.n_triangles == 0) return;
    if (!use_boxes) { walk_a = do_a; walk_b = do_b; return; }
    for_each_box(sc, c, [&](float4 bmin, float4 bmax) {
        const V3 lo = xyz(bmin) - start;
        const V3 hi = xyz(bmax) - start;
        if constexpr (M::kExact) {
            if (do_a && !walk_a) walk_a = box_hit(lo, hi, ray_a);
            if (do_b && !walk_b) walk_b = box_hit(lo, hi, ray_b);
        } else {
            if (do_a && !walk_a) walk_a = box_hit_fast(lo, hi, ray_a);
            if (do_b && !walk_b) walk_b = box_hit_fast(lo, hi, ray_b);
        }
    });
}

// Barycentrics of a known hit, by the expressions of the reference's triangle
// loop (src/raytrace.cu:155-164).
template <class M = ExactMath>
__device__ __forceinline__ void barycentrics(const DeviceScene& sc, int tri, V3 ray, V3 start, float& alpha, float& beta, float& gamma) {
    const float4 g0 = sc.tri_geom[3 * tri + 0], g1 = sc.tri_geom[3 * tri + 1], g2 = sc.tri_geom[3 * tri + 2];
    const V3 A = mk(g0.x, g0.y, g0.z), e1 = mk(g0.w, g1.x, g1.y), e2 = mk(g1.z, g1.w, g2.x), N = mk(g2.y, g2.z, g2.w);
    const V3 q = cross_m<M>(A - start, ray);
    if constexpr (M::kExact) {
        const float d = dot(ray, N);
        beta = dot(e2, q) / d;
        gamma = -dot(e1, q) / d;
    } else {
        const float r = __builtin_amdgcn_rcpf(fast_dot(ray, N));
        beta = fast_dot(e2, q) * r;
        gamma = -fast_dot(e1, q) * r;
    }
    alpha = 1.f - beta - gamma;
}

} // namespace
