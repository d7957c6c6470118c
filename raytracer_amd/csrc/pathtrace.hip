// pathtrace.hip -- gfx950 kernels and the device half of the C ABI.
//
// The reference runs one CUDA thread per pixel through the whole sample and
// bounce loop (reference src/raytrace.cu:446-507).  Here a frame is a batch of
// independent PATHS (one per pixel and sample) advanced one bounce at a time:
//
//   k_generate           camera sample, nearest sphere, box test
//   for each bounce:
//     k_mesh             every queued ray against every triangle
//     k_shade            hit selection, material response, shadow rays,
//                        next ray, nearest sphere, box test
//   k_mesh               shadow rays of the last bounce
//   k_accumulate         per pixel: sum the sample colours in sample order
//
// Only rays that pass the bounding-box test are queued for k_mesh; the queue
// is dense, so all 64 lanes of a wavefront walk the triangle list with work
// to do.  In k_mesh a lane owns a ray and the triangle index is wave-uniform,
// so triangle data comes in through scalar loads and lives in SGPRs.
//
// Arithmetic follows the reference expression by expression and this file is
// compiled with -ffp-contract=off, so sums and products round separately as
// they do in the reference's float code.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/rt_amd.h"
#include "rt_math.h"

// ------------------------------------------------------------ error store

static thread_local char g_error[512] = "";

void rt_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
}

extern "C" const char* rt_last_error(void) { return g_error; }

#define HIP_TRY(expr)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            rt_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return RT_ERR_DEVICE;                                                                 \
        }                                                                                         \
    } while (0)

// ---------------------------------------------------------- device layout

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
constexpr uint32_t kClusterPairs = 32;      // triangle pairs per culling cluster (64 triangles)
constexpr uint32_t kGroupClusters = 64;     // clusters per group (second level of the culling hierarchy)
constexpr uint32_t kMaxClusters = kGroupClusters * 64; // 4096 clusters = 262 144 triangles; larger meshes get larger clusters
constexpr size_t kClusterListBudget = size_t(24) << 30; // most bytes the per-cluster ray lists may take (of 288 GB)

constexpr uint32_t F_ALIVE = 1u;   // the path still bounces
constexpr uint32_t F_PENDING = 2u; // pend[] holds light contributions of the last hit
constexpr uint32_t F_EARLY = 4u;   // the reference's terminate_early for this sample

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
                              // (triangles are stored in Morton order of their centroids)
    const float2* cl_pair;    // padded bounds of each cluster of cluster_pairs consecutive pairs,
                              // 6 per PAIR of clusters (2k, 2k+1): lo.x lo.y lo.z hi.x hi.y hi.z
    const float2* grp_pair;   // the same for groups of kGroupClusters consecutive clusters (meshes of more than 64 clusters)
    const float4* tri_shade;  // 3 per triangle: n0.xyz material | n1.xyz - | n2.xyz -
    const float4* box_min;
    const float4* box_max;
    const float4* light;      // position.xyz, intensity
    uint32_t n_spheres, n_triangles, n_boxes, n_lights;
    uint32_t n_clusters, cluster_pairs, n_groups;
};

// Per-batch path state and queues.
struct Work {
    float4* ray_o;   // origin.xyz, nearest-sphere distance
    float4* ray_d;   // direction.xyz, nearest-sphere index as int bits
    float4* thr;     // albedo.xyz, next random dimension as uint bits
    float4* rad;     // colour L.xyz, F_* flags as uint bits
    float4* pend;    // n_lights per path: contribution.xyz, light distance or PEND_*
    unsigned long long* hit; // paths + paths*n_lights slots: (t bits << 32) | triangle
    float4* q_o;     // queued ray origin.xyz, hit slot as uint bits
    float4* q_d;     // queued ray direction.xyz, cut-off distance
    uint32_t* q_count; // kShards counters (kCountStride words apart) per k_mesh stage of a batch
    uint32_t* cast_count; // statistics: kCastShards counters (kCountStride words apart) per path kernel of a batch
    uint32_t shard_capacity; // queue entries reserved per shard
    // cluster culling: per stage slot, the queue positions of the rays that touch each cluster
    uint32_t* cl_count;      // n_clusters counters per slot
    uint32_t* cl_list;       // n_clusters lists of list_capacity queue positions (reused by every slot)
    uint32_t* grp_count;     // the same three for the groups of clusters
    uint32_t* grp_list;
    uint32_t list_capacity;  // = rays binned per tile
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
__device__ __forceinline__ float norm(V3 a) { return sqrtf((a.x * a.x + a.y * a.y) + a.z * a.z); }
__device__ __forceinline__ V3 normalize(V3 a) {
    const float n = norm(a);
    return mk(a.x / n, a.y / n, a.z / n);
}
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ V3 xyz(float4 v) { return mk(v.x, v.y, v.z); }

// ---------------------------------------------------------- intersection

// The first kSphereCache spheres of the scene, staged in LDS by the kernels
// that test every sphere for every path: the index is wave-uniform, so an LDS
// read is a broadcast, whereas a global load (which the compiler cannot turn
// into a scalar load in kernels that also write global memory) stalls every
// iteration on the vector memory path.
constexpr uint32_t kSphereCache = 64;

struct Spheres {
    const float4* lds;
    const float4* global;
    uint32_t cached, n;
    __device__ __forceinline__ float4 operator[](uint32_t i) const { return i < cached ? lds[i] : global[i]; }
};

// All threads of the workgroup must call it.
__device__ __forceinline__ Spheres stage_spheres(const DeviceScene& sc, float4* lds) {
    Spheres s;
    s.lds = lds;
    s.global = sc.sphere;
    s.n = sc.n_spheres;
    s.cached = s.n < kSphereCache ? s.n : kSphereCache;
    for (uint32_t i = threadIdx.x; i < s.cached; i += kBlock) lds[i] = sc.sphere[i];
    __syncthreads();
    return s;
}

// One sphere of the reference's scan (get_distance, src/raytrace.cu:59-84)
// given b = u . OC and n2 = |OC| * |OC|.
__device__ __forceinline__ void sphere_step(float b, float n2, float R, uint32_t i, float& min_dist, int& id) {
    const float delta = b * b - n2 + R * R;
    if (delta >= 0) {
        const float sq = sqrtf(delta);
        const float t2 = -b + sq;
        if (t2 >= 0) {
            const float t1 = -b - sq;
            const float t = t1 >= 0 ? t1 : t2;
            if (id < 0 || t < min_dist) { min_dist = t; id = (int)i; }
        }
    }
}

// Nearest sphere along a ray (reference get_distance, src/raytrace.cu:53-87).
__device__ __forceinline__ float nearest_sphere(const Spheres& sp, V3 ray, V3 start, int& id) {
    float min_dist = INFINITY;
    for (uint32_t i = 0; i < sp.n; i++) {
        const float4 s = sp[i];
        const V3 oc = start - xyz(s);
        const float ocn = norm(oc);
        sphere_step(dot(ray, oc), ocn * ocn, s.w, i, min_dist, id);
    }
    return min_dist;
}

// The same scan for two rays that leave the same point (a shadow ray and the
// bounce ray of one hit): OC and its norm depend on the start only and are
// computed once per sphere.  A ray whose flag is false keeps (inf, id).
__device__ __forceinline__ void nearest_sphere_pair(const Spheres& sp, V3 start, bool do_a, V3 ray_a, float& dist_a, int& id_a,
                                                    bool do_b, V3 ray_b, float& dist_b, int& id_b) {
    dist_a = INFINITY;
    dist_b = INFINITY;
    for (uint32_t i = 0; i < sp.n; i++) {
        const float4 s = sp[i];
        const V3 oc = start - xyz(s);
        const float ocn = norm(oc);
        const float n2 = ocn * ocn;
        if (do_a) sphere_step(dot(ray_a, oc), n2, s.w, i, dist_a, id_a);
        if (do_b) sphere_step(dot(ray_b, oc), n2, s.w, i, dist_b, id_b);
    }
}

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

// Slab test against every box until one is hit (reference
// src/raytrace.cu:103-145): per axis the entry/exit parameters stay 0 when
// the ray is parallel to the slab, there is no t > 0 requirement, and the
// verdict is min(exit) > max(entry).
__device__ __forceinline__ bool any_box_hit(const DeviceScene& sc, V3 ray, V3 start) {
    for (uint32_t o = 0; o < sc.n_boxes; o++)
        if (box_hit(xyz(sc.box_min[o]) - start, xyz(sc.box_max[o]) - start, ray)) return true;
    return false;
}

// Appends the rays of the lanes with want == true to the k_mesh queue: one
// atomic per wavefront, slots in lane order.  The queue is split into kShards
// shards with a counter each and a wavefront always uses the shard of its own
// index, which spreads the returning atomics over 64 addresses (one address
// serialises at well under 200 atomics per microsecond).
// Returns the queue position of this lane's ray (0 when want == false).
__device__ __forceinline__ uint32_t enqueue(const Work& w, uint32_t counter, bool want, V3 start, V3 ray, uint32_t slot, float cutoff) {
    const unsigned long long mask = __ballot(want);
    if (mask == 0) return 0;
    const uint32_t lane = __lane_id();
    const uint32_t shard = (blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)) % kShards;
    const uint32_t leader = (uint32_t)__ffsll((long long)mask) - 1u;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(&w.q_count[(counter * kShards + shard) * kCountStride], (uint32_t)__popcll(mask));
    base = __shfl(base, (int)leader);
    uint32_t at = 0;
    if (want) {
        at = shard * w.shard_capacity + base + (uint32_t)__popcll(mask & ((1ULL << lane) - 1ULL));
        w.q_o[at] = make_float4(start.x, start.y, start.z, __uint_as_float(slot));
        w.q_d[at] = make_float4(ray.x, ray.y, ray.z, cutoff);
    }
    return at;
}

// Statistics: adds n (the wavefront's number of rays tested against the
// spheres in this kernel) to one of the kernel's counters.  All lanes pass the
// same n.
__device__ __forceinline__ void add_casts(const Work& w, uint32_t counter, uint32_t n) {
    if (n && __lane_id() == 0) {
        const uint32_t shard = (blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)) % kCastShards;
        atomicAdd(&w.cast_count[((size_t)counter * kCastShards + shard) * kCountStride], n);
    }
}

// True when the ray has to walk the triangle list (reference
// src/raytrace.cu:103-146: with boxes enabled only after a box hit).
__device__ __forceinline__ bool needs_mesh(const DeviceScene& sc, int use_boxes, V3 ray, V3 start) {
    if (sc.n_triangles == 0) return false;
    if (!use_boxes) return true;
    return any_box_hit(sc, ray, start);
}

// needs_mesh for two rays that leave the same point; box - start is shared.
__device__ __forceinline__ void needs_mesh_pair(const DeviceScene& sc, int use_boxes, V3 start, bool do_a, V3 ray_a, bool& walk_a,
                                                bool do_b, V3 ray_b, bool& walk_b) {
    walk_a = false;
    walk_b = false;
    if (sc.n_triangles == 0) return;
    if (!use_boxes) { walk_a = do_a; walk_b = do_b; return; }
    for (uint32_t o = 0; o < sc.n_boxes; o++) {
        const V3 lo = xyz(sc.box_min[o]) - start;
        const V3 hi = xyz(sc.box_max[o]) - start;
        if (do_a && !walk_a) walk_a = box_hit(lo, hi, ray_a);
        if (do_b && !walk_b) walk_b = box_hit(lo, hi, ray_b);
    }
}

// Barycentrics of a known hit, by the expressions of the reference's triangle
// loop (src/raytrace.cu:155-164).
__device__ __forceinline__ void barycentrics(const DeviceScene& sc, int tri, V3 ray, V3 start, float& alpha, float& beta, float& gamma) {
    const float4 g0 = sc.tri_geom[3 * tri + 0], g1 = sc.tri_geom[3 * tri + 1], g2 = sc.tri_geom[3 * tri + 2];
    const V3 A = mk(g0.x, g0.y, g0.z), e1 = mk(g0.w, g1.x, g1.y), e2 = mk(g1.z, g1.w, g2.x), N = mk(g2.y, g2.z, g2.w);
    const V3 q = cross(A - start, ray);
    const float d = dot(ray, N);
    beta = dot(e2, q) / d;
    gamma = -dot(e1, q) / d;
    alpha = 1.f - beta - gamma;
}

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
        const float4 o4 = q_o[at];
        const float4 d4 = q_d[at];
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
// through the (padded) bounds of the cluster of 64 spatially adjacent
// triangles it belongs to, before the ray's cut-off distance.  k_bin64 lists,
// per cluster, the queued rays that do; k_mesh_clusters turns the list lengths
// into job offsets (plan_jobs) and runs the same pair scan as k_mesh over each
// cluster's list.  A mesh of more than 64 clusters gets a second level: groups
// of 64 consecutive clusters (neighbours in space, the triangles being in
// Morton order) with the union of their bounds; k_bin64 lists the rays per
// group and k_bin_groups bins each group's list into its clusters, so a ray
// is tested against 64 boxes per level instead of against every cluster --
// a 64-ary bounding-volume hierarchy walked breadth first.  The image is the exhaustive scan's, bit for bit, unless the
// reference's own rounding makes it accept a triangle for a ray that misses
// the triangle by more than the padding (a grazing ray whose barycentrics are
// rounding noise); rt_options.cluster_culling = 0 selects the exhaustive scan.

// Index of the last entry of first[0..n) that is <= job (entries are
// non-decreasing; empty ranges share their successor's value).
__device__ __forceinline__ uint32_t owner_of_job(const uint32_t* first, uint32_t n, uint32_t job) {
    uint32_t lo = 0, hi = n - 1;
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) / 2;
        if (first[mid] <= job) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// first_job[i] = number of 256-entry jobs of the lists before list i,
// first_job[n] = their total, from the list lengths (count[i * kCountStride]);
// n <= kMaxClusters.  Every workgroup of a kernel that walks the lists makes
// its own copy in LDS (`scan` is kBlock words of scratch): the lengths sit in
// L2 and the scan is a few microseconds, less than a launch of its own would
// cost.  All threads of the workgroup must call it.
__device__ __forceinline__ void plan_jobs(const uint32_t* __restrict__ count, uint32_t n, uint32_t* first_job, uint32_t* scan) {
    const uint32_t t = threadIdx.x;
    const uint32_t per = (n + kBlock - 1) / kBlock; // consecutive lists per thread, at most 16
    uint32_t mine = 0;
    for (uint32_t k = 0; k < per; k++) {
        const uint32_t i = t * per + k;
        if (i < n) mine += (count[(size_t)i * kCountStride] + kBlock - 1) / kBlock;
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
            run += (count[(size_t)i * kCountStride] + kBlock - 1) / kBlock;
        }
    }
    if (t == kBlock - 1) first_job[n] = scan[t];
    __syncthreads();
}

// The hardware minimum / maximum: a NaN operand is dropped (minNum / maxNum).
// Written as instructions because the library functions first re-quiet every
// operand (v_max x, x), which doubles the cost of a slab test.
__device__ __forceinline__ float hw_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float hw_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float hw_min3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float hw_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// One ray prepared for slab tests: 1/u, origin * (1/u), cut-off distance.
struct SlabRay {
    float ix, iy, iz, ox, oy, oz, cutoff;
};

__device__ __forceinline__ SlabRay slab_ray(float4 o4, float4 d4) {
    SlabRay r;
    r.ix = 1.f / d4.x; r.iy = 1.f / d4.y; r.iz = 1.f / d4.z; // +-inf for an axis-parallel ray
    r.ox = o4.x * r.ix; r.oy = o4.y * r.iy; r.oz = o4.z * r.iz;
    r.cutoff = d4.w;
    return r;
}

// Slab tests of R rays per lane against box pairs [first, last), last - first
// <= 16: bit 2(k - first) + h of bits[i] is set when ray i enters box h of
// pair k before its cut-off distance.  The slab parameters come from packed
// FMAs (t = bound * (1/u) - origin * (1/u)); this is only a filter and need not
// follow the reference's rounding.  NaN (inf - inf on an axis-parallel ray) is
// dropped by the minimum / maximum, which errs towards "touches".  The box
// data of an iteration (scalar loads) serves all R rays.
template <int R>
__device__ __forceinline__ void slab_bits(const f2* __restrict__ box_pair, uint32_t first, uint32_t last, const SlabRay (&ray)[R],
                                          uint32_t (&bits)[R]) {
#pragma unroll
    for (int i = 0; i < R; i++) bits[i] = 0;
    for (uint32_t k = first; k < last; k++) {
        const f2* __restrict__ b = box_pair + 6 * (size_t)k;
        const f2 b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3], b4 = b[4], b5 = b[5];
        const uint32_t bit = 1u << (2 * (k - first)); // wave-uniform
#pragma unroll
        for (int i = 0; i < R; i++) {
            const SlabRay& r = ray[i];
            const f2 ax = __builtin_elementwise_fma(b0, (f2)(r.ix), (f2)(-r.ox)), bx = __builtin_elementwise_fma(b3, (f2)(r.ix), (f2)(-r.ox));
            const f2 ay = __builtin_elementwise_fma(b1, (f2)(r.iy), (f2)(-r.oy)), by = __builtin_elementwise_fma(b4, (f2)(r.iy), (f2)(-r.oy));
            const f2 az = __builtin_elementwise_fma(b2, (f2)(r.iz), (f2)(-r.oz)), bz = __builtin_elementwise_fma(b5, (f2)(r.iz), (f2)(-r.oz));
            const float in0 = hw_max3(hw_min(ax.x, bx.x), hw_min(ay.x, by.x), hw_min(az.x, bz.x));
            const float out0 = hw_min3(hw_max(ax.x, bx.x), hw_max(ay.x, by.x), hw_max(az.x, bz.x));
            const float in1 = hw_max3(hw_min(ax.y, bx.y), hw_min(ay.y, by.y), hw_min(az.y, bz.y));
            const float out1 = hw_min3(hw_max(ax.y, bx.y), hw_max(ay.y, by.y), hw_max(az.y, bz.y));
            const bool t0 = (out0 >= hw_max(in0, 0.f)) & (in0 <= r.cutoff);
            const bool t1 = (out1 >= hw_max(in1, 0.f)) & (in1 <= r.cutoff);
            bits[i] |= (t0 ? bit : 0u) | (t1 ? bit << 1 : 0u);
        }
    }
}

// Bins R rays per lane (queue positions at[i], ignored when !active[i])
// against up to 64 padded boxes and appends each position to the list of every
// box its ray enters before its cut-off distance.  One pass over the boxes: a
// lane keeps the boxes a ray touches as a 64-bit mask; then every lane walks
// the set bits of its masks twice -- to count per box in LDS (`slot`, 64
// words), and, after the workgroup has reserved list space with one atomic per
// box, to take its positions.  All threads of the workgroup must call it.
template <int R>
__device__ __forceinline__ void bin_prepared64(const f2* __restrict__ box_pair, uint32_t n_boxes, const bool (&active)[R], const uint32_t (&at)[R],
                                               const SlabRay (&ray)[R], uint32_t* __restrict__ count, uint32_t* __restrict__ list,
                                               uint32_t list_capacity, uint32_t* slot) {
    if (threadIdx.x < 64) slot[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t n_pairs = (n_boxes + 1) / 2;
    // boxes 0-31 and 32-63 each ray touches
    uint32_t bits_lo[R], bits_hi[R];
    slab_bits<R>(box_pair, 0, min(n_pairs, 16u), ray, bits_lo);
    slab_bits<R>(box_pair, 16, max(n_pairs, 16u), ray, bits_hi);
    // an odd box count leaves the second half of the last entry without a box: drop its bit
    const unsigned long long real = n_boxes >= 64 ? ~0ULL : (1ULL << n_boxes) - 1ULL;
    unsigned long long mine[R];
#pragma unroll
    for (int i = 0; i < R; i++) {
        mine[i] = active[i] ? ((((unsigned long long)bits_hi[i] << 32) | bits_lo[i]) & real) : 0ULL;
        for (unsigned long long m = mine[i]; m; m &= m - 1) atomicAdd(&slot[(uint32_t)__ffsll((long long)m) - 1u], 1u);
    }
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

// The same for rays read back from the queue.
template <int R>
__device__ __forceinline__ void bin_rays64(const f2* __restrict__ box_pair, uint32_t n_boxes, const bool (&active)[R], const uint32_t (&at)[R],
                                           const float4* __restrict__ q_o, const float4* __restrict__ q_d,
                                           uint32_t* __restrict__ count, uint32_t* __restrict__ list, uint32_t list_capacity,
                                           uint32_t* slot) {
    SlabRay ray[R];
#pragma unroll
    for (int i = 0; i < R; i++) ray[i] = slab_ray(q_o[at[i]], q_d[at[i]]);
    bin_prepared64<R>(box_pair, n_boxes, active, at, ray, count, list, list_capacity, slot);
}

// First level: the queued rays of one tile (256-ray blocks [first_block,
// first_block + blocks_in_tile)) against the clusters of a mesh of at most 64
// clusters, or against the groups of a larger one.  A workgroup takes R
// consecutive ray blocks at a time (R rays per lane): two halve the barriers
// and reservations per ray, one keeps more workgroups busy when the queue is
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
// clusters of that group.  A job is 256 list entries of one group.
__global__ __launch_bounds__(kBlock) void k_bin_groups(const f2* __restrict__ cl_pair, uint32_t n_clusters, uint32_t n_groups,
                                                       const float4* __restrict__ q_o, const float4* __restrict__ q_d,
                                                       const uint32_t* __restrict__ grp_count, const uint32_t* __restrict__ grp_list,
                                                       uint32_t* __restrict__ cl_count, uint32_t* __restrict__ cl_list, uint32_t list_capacity) {
    __shared__ uint32_t slot[64];
    __shared__ uint32_t first_job[64 + 1];
    __shared__ uint32_t scan[kBlock];
    plan_jobs(grp_count, n_groups, first_job, scan);
    const uint32_t n_jobs = first_job[n_groups];
    for (uint32_t job = blockIdx.x; job < n_jobs; job += gridDim.x) {
        const uint32_t g = owner_of_job(first_job, n_groups, job);
        const uint32_t count = grp_count[g * kCountStride];
        const uint32_t e = (job - first_job[g]) * kBlock + threadIdx.x;
        const bool active[1] = {e < count};
        const uint32_t at[1] = {grp_list[(size_t)g * list_capacity + (active[0] ? e : count - 1)]};
        const uint32_t first_cluster = g * kGroupClusters;
        bin_rays64<1>(cl_pair + 6 * (size_t)(first_cluster / 2), min(kGroupClusters, n_clusters - first_cluster), active, at, q_o, q_d,
                   cl_count + (size_t)first_cluster * kCountStride, cl_list + (size_t)first_cluster * list_capacity, list_capacity, slot);
    }
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
    for (uint32_t job = blockIdx.x; job < n_jobs; job += gridDim.x) {
        const uint32_t c = owner_of_job(first_job, n_clusters, job);
        const uint32_t count = cl_count[c * kCountStride];
        const uint32_t e = (job - first_job[c]) * kBlock + threadIdx.x;
        const bool active = e < count;
        const uint32_t at = cl_list[(size_t)c * list_capacity + (active ? e : count - 1)];
        const float4 o4 = q_o[at];
        const float4 d4 = q_d[at];
        Nearest best = nearest_init(d4.w);
        const uint32_t first = c * cluster_pairs;
        scan_pairs(tri2, tri_orig, first, min(first + cluster_pairs, n_pairs), o4.x, o4.y, o4.z, d4.x, d4.y, d4.z, best);
        if (active && best.index >= 0) {
            const unsigned long long key = ((unsigned long long)__float_as_uint(hit_distance(best)) << 32) | tri_orig[best.index];
            atomicMin(&hit[__float_as_uint(o4.w)], key);
        }
    }
}

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
};

__device__ __forceinline__ uint32_t path_pixel(const Frame& f, uint32_t p) { return f.pixel_offset + p % f.n_pixels; }
__device__ __forceinline__ uint32_t path_sample(const Frame& f, uint32_t p) { return f.sample_begin + p / f.n_pixels; }

// State write-back of a freshly generated ray whose nearest sphere (ts, sid)
// and mesh verdict are known, and queueing for k_mesh.  All lanes of the
// wavefront must call it.
__device__ __forceinline__ uint32_t commit_ray(const Work& w, uint32_t counter, bool alive, uint32_t p, V3 start, V3 ray,
                                               float ts, int sid, bool walk) {
    if (alive) {
        w.ray_o[p] = make_float4(start.x, start.y, start.z, ts);
        w.ray_d[p] = make_float4(ray.x, ray.y, ray.z, __int_as_float(sid));
        w.hit[p] = kNoHit;
    }
    return enqueue(w, counter, alive && walk, start, ray, p, ts);
}

// Nearest sphere + box test for a freshly generated ray, then commit_ray.
__device__ __forceinline__ void launch_ray(const DeviceScene& sc, const Spheres& sp, const Work& w, const Frame& f, uint32_t counter,
                                           bool alive, uint32_t p, V3 start, V3 ray) {
    int sid = -1;
    float ts = INFINITY;
    bool walk = false;
    if (alive) {
        ts = nearest_sphere(sp, ray, start, sid);
        walk = needs_mesh(sc, f.bounding_box, ray, start);
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
    const float radius = sqrtf(-2 * rt_logf(r1));
    const float angle = (float)(2 * M_PI * (double)r2);
    float sn, cs;
    rt_sincosf(angle, &sn, &cs);
    dx = radius * cs * 1.f;
    dy = radius * sn * 1.f;
}

template <uint32_t kPaths>
__global__ __launch_bounds__(kBlock) void k_generate(DeviceScene sc, Work w, Frame f) {
    __shared__ float4 sphere_lds[kSphereCache];
    __shared__ uint32_t key_kp[kPaths][kBlock], key_ks[kPaths][kBlock];
    __shared__ uint32_t accepted[kPaths][kBlock]; // first dimension of the accepted draw
    const Spheres sp = stage_spheres(sc, sphere_lds);
    const uint32_t base = blockIdx.x * (kBlock * kPaths) + threadIdx.x;
    // only this lane reads what it writes to the three arrays: no barrier needed
    uint32_t owned = 0;
#pragma unroll
    for (uint32_t r = 0; r < kPaths; r++) {
        const uint32_t p = base + r * kBlock;
        if (p < f.n_paths) {
            const rt_rng_key key = rt_rng_make_key(f.seed, path_pixel(f, p), path_sample(f, p));
            key_kp[r][threadIdx.x] = key.kp;
            key_ks[r][threadIdx.x] = key.ks;
            accepted[r][threadIdx.x] = 0;
            owned = r + 1; // paths are consecutive: a lane owns slots 0 .. owned-1
        }
    }
    if (f.antialiasing) {
        uint32_t r = 0, dim = 0;
        rt_rng_key key;
        key.kp = key_kp[0][threadIdx.x];
        key.ks = key_ks[0][threadIdx.x];
        while (r < owned) {
            const float r1 = rt_rng_uniform(key, dim);
            bool accept = false;
            if (r1 >= 0.77f) {
                const float r2 = rt_rng_uniform(key, dim + 1);
                const float radius = __builtin_sqrtf(-1.3862944f * __builtin_amdgcn_logf(r1)); // -2 ln 2 * log2(r1)
                const float m = fmaxf(fabsf(radius * __builtin_amdgcn_cosf(r2)), fabsf(radius * __builtin_amdgcn_sinf(r2))); // argument in turns
                if (m < 0.4999f) {
                    accept = true;
                } else if (!(m > 0.5001f)) {
                    float dx, dy;
                    jitter_exact(key, dim, dx, dy);
                    accept = !(fabsf(dx) > 0.5f || fabsf(dy) > 0.5f);
                }
            }
            if (accept) {
                accepted[r][threadIdx.x] = dim;
                r++;
                dim = 0;
                if (r < owned) {
                    key.kp = key_kp[r][threadIdx.x];
                    key.ks = key_ks[r][threadIdx.x];
                }
            } else {
                dim += 2;
            }
        }
    }
    uint32_t n_casts = 0;
    for (uint32_t r = 0; r < kPaths; r++) {
        const uint32_t p = base + r * kBlock;
        const bool alive = p < f.n_paths;
        V3 start = mk(f.pos[0], f.pos[1], f.pos[2]);
        V3 ray = mk(0.f, 0.f, 1.f);
        if (alive) {
            const uint32_t pixel = path_pixel(f, p);
            const uint32_t ux = pixel % f.width, uy = pixel / f.width;
            uint32_t dim = 0;
            float dx = 0.f, dy = 0.f;
            if (f.antialiasing) {
                rt_rng_key key;
                key.kp = key_kp[r][threadIdx.x];
                key.ks = key_ks[r][threadIdx.x];
                dim = accepted[r][threadIdx.x];
                jitter_exact(key, dim, dx, dy);
                dim += 2;
            }
            const float px = (float)ux + 0.5f + dx;
            const float py = (float)uy + 0.5f + dy;
            const V3 dir = normalize(mk((px - f.cx) / f.fx, (py - f.cy) / f.fy, 1.f));
            ray = mk(dot(mk(f.rot[0], f.rot[1], f.rot[2]), dir),
                     dot(mk(f.rot[3], f.rot[4], f.rot[5]), dir),
                     dot(mk(f.rot[6], f.rot[7], f.rot[8]), dir));
            w.thr[p] = make_float4(1.f, 1.f, 1.f, __uint_as_float(dim));
            w.rad[p] = make_float4(0.f, 0.f, 0.f, __uint_as_float(F_ALIVE));
        }
        launch_ray(sc, sp, w, f, 0, alive, p, start, ray);
        n_casts += (uint32_t)__popcll(__ballot(alive));
    }
    if (f.count_casts) add_casts(w, 0, n_casts);
}

// Adds the light contributions a path left pending at its previous hit, in
// light order, now that k_mesh has answered its shadow rays.
__device__ __forceinline__ V3 settle_pending(const DeviceScene& sc, const Work& w, uint32_t n_paths, uint32_t p, V3 L) {
    for (uint32_t l = 0; l < sc.n_lights; l++) {
        const float4 pe = w.pend[(size_t)p * sc.n_lights + l];
        if (pe.w == PEND_NONE) continue;
        bool visible = true;
        if (pe.w != PEND_ALWAYS) {
            const unsigned long long key = w.hit[(size_t)n_paths + (size_t)p * sc.n_lights + l];
            visible = !(__uint_as_float((uint32_t)(key >> 32)) < pe.w);
        }
        if (visible) L = L + mk(pe.x, pe.y, pe.z);
    }
    return L;
}

// cosine-weighted direction about N (reference random_cos,
// src/raytrace.cu:253-296, with argmin_abs of include/vector.cuh:87-98: the
// running minimum starts at the signed first component)
__device__ __forceinline__ V3 random_cos(float r1, float r2, V3 N) {
    const float tmp1 = sqrtf(1 - r2);
    const float tmp2 = (float)(2 * M_PI * (double)r1);
    float sn, cs;
    rt_sincosf(tmp2, &sn, &cs);
    const float x = tmp1 * cs;
    const float y = tmp1 * sn;
    const float z = sqrtf(r2);
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
    T1 = normalize(T1);
    const V3 T2 = normalize(cross(N, T1));
    return (x * T1 + y * T2) + z * N;
}

__device__ __forceinline__ V3 reflect(V3 ray, V3 N) { return ray - (2 * dot(ray, N)) * N; }

// One bounce of every live path (reference get_color loop body,
// src/raytrace.cu:317-442).  `depth` is the bounce index; counter depth+1
// receives the rays this bounce queues for k_mesh.
//
// The kernel has two parts.  The first, divergent by material, finds the hit
// point, its normal and the bounce ray.  The second is shared by all lanes:
// the sphere scans and box tests of the shadow rays (diffuse hits) and of the
// bounce ray.  The first light's shadow ray and the bounce ray of a diffuse
// hit leave the same point, so their scans run as one (nearest_sphere_pair).
// (At least five wavefronts per SIMD: 96 VGPRs.  One register more would cost a fifth of the occupancy.)
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(5, 8))) void k_shade(DeviceScene sc, Work w, Frame f, int depth) {
    __shared__ float4 sphere_lds[kSphereCache];
    const Spheres sp = stage_spheres(sc, sphere_lds);
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t counter = (uint32_t)depth + 1u;
    bool alive = false; // the path continues with a bounce ray
    bool lit = false;   // diffuse hit: the lights are evaluated at `start`
    bool open = false;  // the path was alive on entry: its radiance record is rewritten
    V3 start = mk(0, 0, 0), ray = mk(0, 0, 1);
    V3 P = mk(0, 0, 0), N = mk(0, 0, 1), L = mk(0, 0, 0);
    V3 weight = mk(0, 0, 0), colour = mk(0, 0, 0); // path weight before this hit, surface colour
    uint32_t flags = 0;
    if (p < f.n_paths) {
        const float4 rad = w.rad[p];
        flags = __float_as_uint(rad.w);
        if (flags & F_ALIVE) {
            open = true;
            L = mk(rad.x, rad.y, rad.z);
            if (flags & F_PENDING) {
                L = settle_pending(sc, w, f.n_paths, p, L);
                flags &= ~F_PENDING;
            }
            const float4 ro4 = w.ray_o[p], rd4 = w.ray_d[p];
            start = xyz(ro4);
            ray = xyz(rd4);
            const float ts = ro4.w;
            int sid = __float_as_int(rd4.w);
            const unsigned long long key = w.hit[p];
            const float tt = __uint_as_float((uint32_t)(key >> 32));
            int tid = (int)(uint32_t)key; // 0xffffffff -> -1

            // nearer of the two, spheres win ties (reference get_intersection,
            // src/raytrace.cu:194-215; `tmp` starts as the triangle id)
            float min_dist = INFINITY;
            int tmp = tid;
            if (ts < min_dist) { min_dist = ts; tmp = tid; tid = -1; }
            if (tt < min_dist) { min_dist = tt; sid = -1; tid = tmp; }

            if (sid >= 0 || tid >= 0) {
                P = isinf(min_dist) ? mk(INFINITY, INFINITY, INFINITY) : start + min_dist * ray;
                int m;
                if (sid >= 0) {
                    m = sc.sphere_mat[sid];
                    N = normalize(P - xyz(sp[(uint32_t)sid]));
                } else {
                    float alpha, beta, gamma;
                    barycentrics(sc, tid, ray, start, alpha, beta, gamma);
                    const float4 s0 = sc.tri_shade[3 * tid + 0], s1 = sc.tri_shade[3 * tid + 1], s2 = sc.tri_shade[3 * tid + 2];
                    m = __float_as_int(s0.w);
                    N = normalize((alpha * xyz(s0) + beta * xyz(s1)) + gamma * xyz(s2));
                }
                const float4 mc = sc.mat_color[m];
                const int type = __float_as_int(mc.w);
                colour = xyz(mc);
                const float4 th = w.thr[p];
                weight = xyz(th);
                V3 next_weight = weight;
                uint32_t dim = __float_as_uint(th.w);
                const rt_rng_key rng = rt_rng_make_key(f.seed, path_pixel(f, p), path_sample(f, p));
                alive = true;

                if (type == RT_DIFFUSE) {
                    if (!f.indirect && depth == 0) flags |= F_EARLY;
                    if (f.antialiasing) flags &= ~F_EARLY;
                    lit = true;
                    start = P + scale_d(0.01, N);
                    if (!f.indirect) {
                        alive = false;
                    } else {
                        next_weight = weight * colour;
                        const float r1 = rt_rng_uniform(rng, dim);
                        const float r2 = rt_rng_uniform(rng, dim + 1);
                        dim += 2;
                        ray = random_cos(r1, r2, N);
                    }
                } else if (type == RT_MIRROR) {
                    const V3 rr = reflect(ray, N);
                    start = P + scale_d(0.01, N);
                    ray = rr;
                } else { // RT_GLASS
                    const float2 ior = sc.mat_ior[m];
                    float ro = ior.x, ri = ior.y;
                    float wn = dot(ray, N);
                    // Schlick reflectance (reference src/raytrace.cu:388-394)
                    const float k0 = (ro - ri) * (ro - ri) / ((ro + ri) * (ro + ri));
                    const float om = 1.f - fabsf(wn);
                    const float om2 = om * om;
                    const float R = k0 + (1.f - k0) * om2 * om2 * om;
                    const float r = rt_rng_uniform(rng, dim);
                    dim += 1;
                    if (r < R) {
                        const V3 rr = reflect(ray, N);
                        start = P + scale_d(0.01, N);
                        ray = rr;
                    } else {
                        if (wn > 0) { // leaving the medium: swap indices, flip the normal
                            const float sw = ro; ro = ri; ri = sw;
                            N = -1.f * N;
                            wn = dot(ray, N);
                        }
                        const float eta = ro / ri;
                        const float k = 1.f - eta * eta * (1.f - wn * wn);
                        if (k < 0) { // total internal reflection
                            const V3 rr = reflect(ray, N);
                            start = P + scale_d(0.01, N);
                            ray = rr;
                        } else {
                            const V3 wt_t = eta * (ray - wn * N);
                            const V3 wt_n = (-sqrtf(k)) * N;
                            start = P - scale_d(0.01, N);
                            ray = wt_t + wt_n;
                        }
                    }
                }
                if (depth + 1 >= f.max_depth) alive = false;
                w.thr[p] = make_float4(next_weight.x, next_weight.y, next_weight.z, __uint_as_float(dim));
            }
        }
    }

    // ---- shared part: shadow rays of diffuse hits, then the bounce ray ----
    float ts = INFINITY;
    int sid = -1;
    bool walk = false;
    bool bounce_probed = false;
    bool deferred = false;
    for (uint32_t l = 0; l < sc.n_lights; l++) {
        // Direct light (reference src/raytrace.cu:351-360) and its shadow ray
        // (is_visible, src/raytrace.cu:218-244).  Lanes without a diffuse hit
        // carry placeholders and are masked by `lit`.
        const float4 li = sc.light[l];
        const V3 lp = xyz(li);
        const V3 sp_ = lp - P;
        const float d = norm(sp_);
        const V3 wi = (1.f / d) * sp_;
        const float nwi = fmaxf(dot(N, wi), 0.f);
        const V3 sray = normalize(lp - start);
        const float dist = norm(lp - start);
        float ds = INFINITY;
        int shadow_sid = -1;
        bool shadow_walk = false;
        if (l == 0) {
            nearest_sphere_pair(sp, start, lit, sray, ds, shadow_sid, alive, ray, ts, sid);
            needs_mesh_pair(sc, f.bounding_box, start, lit && !(ds < dist), sray, shadow_walk, alive, ray, walk);
            bounce_probed = true;
        } else if (lit) {
            ds = nearest_sphere(sp, sray, start, shadow_sid);
            shadow_walk = !(ds < dist) && needs_mesh(sc, f.bounding_box, sray, start);
        }
        const size_t slot = (size_t)p * sc.n_lights + l;
        if (lit) {
            const bool blocked = ds < dist;
            // ((I / (4 pi^2 d^2) * colour) * visible) * N.wi, the first
            // factor in double (reference src/raytrace.cu:360)
            const double q = (double)li.w / (4 * M_PI * M_PI * (double)d * (double)d);
            const V3 direct = nwi * ((blocked ? 0.f : 1.f) * scale_d(q, colour));
            const V3 c = weight * direct;
            if (shadow_walk) {
                if (!deferred) // first pending light: clear the earlier slots
                    for (uint32_t e = 0; e < l; e++) w.pend[(size_t)p * sc.n_lights + e].w = PEND_NONE;
                deferred = true;
                w.pend[slot] = make_float4(c.x, c.y, c.z, dist);
                w.hit[(size_t)f.n_paths + slot] = kNoHit;
            } else if (deferred) {
                w.pend[slot] = make_float4(c.x, c.y, c.z, PEND_ALWAYS);
            } else {
                L = L + c;
            }
        }
        enqueue(w, counter, lit && shadow_walk, start, sray, (uint32_t)(f.n_paths + slot), dist);
    }
    if (f.count_casts)
        add_casts(w, counter, (uint32_t)__popcll(__ballot(lit)) * sc.n_lights + (uint32_t)__popcll(__ballot(alive)));
    if (deferred) flags |= F_PENDING;
    if (open) {
        if (!alive) flags &= ~F_ALIVE;
        w.rad[p] = make_float4(L.x, L.y, L.z, __uint_as_float(flags));
    }
    if (!bounce_probed && alive) { // a scene without lights
        ts = nearest_sphere(sp, ray, start, sid);
        walk = needs_mesh(sc, f.bounding_box, ray, start);
    }
    commit_ray(w, counter, alive, p, start, ray, ts, sid, walk);
}

// Per pixel: add the batch's sample colours to the running sum in sample
// order, as the reference's sample loop does (src/raytrace.cu:478-499).
// sum[pixel] = (r, g, b, actual_rays).
__global__ __launch_bounds__(kBlock) void k_accumulate(DeviceScene sc, Work w, Frame f, float4* __restrict__ sum) {
    const uint32_t pixel = blockIdx.x * kBlock + threadIdx.x;
    if (pixel >= f.n_pixels) return;
    float4 acc = sum[f.pixel_offset + pixel];
    const uint32_t n_samples = f.n_paths / f.n_pixels;
    for (uint32_t s = 0; s < n_samples; s++) {
        const uint32_t p = s * f.n_pixels + pixel;
        const float4 rad = w.rad[p];
        const uint32_t flags = __float_as_uint(rad.w);
        V3 L = mk(rad.x, rad.y, rad.z);
        if (flags & F_PENDING) L = settle_pending(sc, w, f.n_paths, p, L);
        acc.x += L.x;
        acc.y += L.y;
        acc.z += L.z;
        acc.w += 1.f;
        if (!f.indirect && (flags & F_EARLY)) break;
    }
    sum[f.pixel_offset + pixel] = acc;
}

// gamma((1/actual_rays) * colour) (reference src/raytrace.cu:44-51,500)
__global__ __launch_bounds__(kBlock) void k_resolve(const float4* __restrict__ sum, uint32_t n_pixels, float* __restrict__ out) {
    const uint32_t pixel = blockIdx.x * kBlock + threadIdx.x;
    if (pixel >= n_pixels) return;
    const float4 s = sum[pixel];
    const float inv = 1.f / s.w;
    out[3 * pixel + 0] = rt_gamma_encode(inv * s.x);
    out[3 * pixel + 1] = rt_gamma_encode(inv * s.y);
    out[3 * pixel + 2] = rt_gamma_encode(inv * s.z);
}

// ------------------------------------------- known-answer test kernels

// rays -> state + queue, through the same launch_ray the renderer uses
__global__ __launch_bounds__(kBlock) void k_probe_rays(DeviceScene sc, Work w, Frame f, const float* __restrict__ origins,
                                                       const float* __restrict__ dirs, int as_segments) {
    __shared__ float4 sphere_lds[kSphereCache];
    const Spheres sp = stage_spheres(sc, sphere_lds);
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
    launch_ray(sc, sp, w, f, 0, alive, p, start, ray);
}

__global__ __launch_bounds__(kBlock) void k_probe_results(DeviceScene sc, Work w, Frame f, int as_segments, float* __restrict__ out_t,
                                                          int32_t* __restrict__ out_sphere, int32_t* __restrict__ out_triangle,
                                                          float* __restrict__ out_abg, int32_t* __restrict__ out_visible) {
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= f.n_paths) return;
    const float4 ro4 = w.ray_o[p], rd4 = w.ray_d[p];
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

// ------------------------------------------------------------- host side

struct rt_scene {
    DeviceScene d{};
    std::vector<void*> allocations;
    // batch work buffers, grown on demand
    Work w{};
    std::vector<void*> work_allocations;
    uint32_t work_paths = 0;
    uint32_t n_counters = 0;
    uint32_t tiles = 1;                  // cluster-list tiles per k_mesh stage
    size_t counter_words = 0;            // queue, cluster and group counters: one block starting at w.q_count
    std::vector<uint32_t> cluster_tris;  // triangles per culling cluster
    // profiling events
    std::vector<hipEvent_t> events;   // pool, reused by every call
    std::vector<char> event_tags;     // tag of the events in use, in record order
    hipEvent_t frame_begin = nullptr, frame_end = nullptr;
    // rt_render's device buffers (sums and output image), grown on demand
    float* frame_sum = nullptr;
    float* frame_out = nullptr;
    size_t frame_pixels = 0;
    // progress reports per finished batch
    rt_progress_fn progress = nullptr;
    void* progress_user = nullptr;
};

namespace {

template <typename T>
int upload(rt_scene* s, const std::vector<T>& host, const T** out) {
    void* dev = nullptr;
    const size_t bytes = (host.empty() ? 1 : host.size()) * sizeof(T);
    HIP_TRY(hipMalloc(&dev, bytes));
    s->allocations.push_back(dev);
    if (!host.empty()) HIP_TRY(hipMemcpy(dev, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = static_cast<const T*>(dev);
    return RT_OK;
}

int validate(const rt_scene_desc* d) {
    if (!d) { rt_set_error("scene description is null"); return RT_ERR_INVALID; }
    if ((d->n_spheres && !(d->sphere_pos && d->sphere_radius && d->sphere_material)) ||
        (d->n_materials && !(d->material_type && d->material_color && d->material_ro && d->material_ri)) ||
        (d->n_vertices && !(d->vertices && d->normals)) ||
        (d->n_triangles && !(d->tri_v && d->tri_n && d->tri_material)) ||
        (d->n_boxes && !(d->box_min && d->box_max)) || (d->n_lights && !(d->light_pos && d->light_intensity))) {
        rt_set_error("scene description has a non-zero count with a null array");
        return RT_ERR_INVALID;
    }
    if (d->n_lights > kMaxLights) { rt_set_error("at most %u lights are supported", kMaxLights); return RT_ERR_INVALID; }
    if (d->n_triangles > 0x7fffffffu / 3 || d->n_spheres > 0x7fffffffu) { rt_set_error("scene too large"); return RT_ERR_INVALID; }
    for (uint32_t i = 0; i < d->n_materials; i++)
        if (d->material_type[i] < RT_DIFFUSE || d->material_type[i] > RT_GLASS) {
            rt_set_error("material %u has unknown type %d", i, d->material_type[i]);
            return RT_ERR_INVALID;
        }
    for (uint32_t i = 0; i < d->n_spheres; i++)
        if (d->sphere_material[i] < 0 || (uint32_t)d->sphere_material[i] >= d->n_materials) {
            rt_set_error("sphere %u refers to material %d of %u", i, d->sphere_material[i], d->n_materials);
            return RT_ERR_INVALID;
        }
    for (uint32_t i = 0; i < d->n_triangles; i++) {
        if (d->tri_material[i] < 0 || (uint32_t)d->tri_material[i] >= d->n_materials) {
            rt_set_error("triangle %u refers to material %d of %u", i, d->tri_material[i], d->n_materials);
            return RT_ERR_INVALID;
        }
        for (int k = 0; k < 3; k++)
            if (d->tri_v[3 * i + k] < 0 || (uint32_t)d->tri_v[3 * i + k] >= d->n_vertices || d->tri_n[3 * i + k] < 0 ||
                (uint32_t)d->tri_n[3 * i + k] >= d->n_vertices) {
                rt_set_error("triangle %u refers to a vertex or normal outside [0, %u)", i, d->n_vertices);
                return RT_ERR_INVALID;
            }
    }
    return RT_OK;
}

int check_frame(const rt_scene* scene, const rt_camera* cam, const rt_options* opt) {
    if (!scene || !cam || !opt) { rt_set_error("null scene, camera or options"); return RT_ERR_INVALID; }
    if (cam->width == 0 || cam->height == 0 || (uint64_t)cam->width * cam->height > (1ull << 28)) {
        rt_set_error("unsupported resolution %ux%u", cam->width, cam->height);
        return RT_ERR_INVALID;
    }
    if (opt->max_depth < 0 || opt->max_depth > 64) { rt_set_error("max_depth %d outside [0, 64]", opt->max_depth); return RT_ERR_INVALID; }
    return RT_OK;
}

void free_work(rt_scene* s) {
    for (void* p : s->work_allocations) (void)hipFree(p);
    s->work_allocations.clear();
    s->work_paths = 0;
    s->n_counters = 0;
    std::memset(&s->w, 0, sizeof s->w);
}

template <typename T>
int work_alloc(rt_scene* s, T** out, size_t count) {
    void* dev = nullptr;
    HIP_TRY(hipMalloc(&dev, (count ? count : 1) * sizeof(T)));
    s->work_allocations.push_back(dev);
    *out = static_cast<T*>(dev);
    return RT_OK;
}

// Work buffers for batches of up to n_paths paths and max_depth bounces.
int ensure_work(rt_scene* s, uint32_t n_paths, uint32_t n_counters) {
    if (s->work_paths >= n_paths && s->n_counters >= n_counters) return RT_OK;
    free_work(s);
    const size_t nl = s->d.n_lights ? s->d.n_lights : 1;
    const size_t slots = (size_t)n_paths * (1 + nl);
    // a shard receives the rays of every kShards-th wavefront of a path kernel
    const size_t n_waves = (((size_t)n_paths + kBlock - 1) / kBlock) * kWavesPerBlock;
    // k_shade: a wavefront queues at most 64 * (1 + lights) rays; k_generate: 64 * kGenPaths, from fewer wavefronts
    size_t shard_capacity = ((n_waves + kShards - 1) / kShards) * 64 * (1 + nl);
    for (size_t r = 2; r <= kGenPaths; r *= 2) {
        const size_t gen_waves = (((size_t)n_paths + kBlock * r - 1) / (kBlock * r)) * kWavesPerBlock;
        shard_capacity = std::max(shard_capacity, ((gen_waves + kShards - 1) / kShards) * 64 * r);
    }
    const size_t queue_entries = shard_capacity * kShards;
    if (queue_entries > 0xffffffffu) { rt_set_error("batch too large for the ray queue"); return RT_ERR_INVALID; }
    // cluster lists: one tile of rays at a time if the whole queue would not fit the budget
    const size_t nc = s->d.n_clusters, ng = s->d.n_groups;
    size_t list_capacity = 0, tiles = 1;
    if (nc > 1) {
        // a tile is a range of 256-ray blocks; every shard may end in a partial block
        const size_t max_blocks = ((shard_capacity + kBlock - 1) / kBlock) * kShards;
        size_t budget = kClusterListBudget, free_bytes = 0, total_bytes = 0;
        if (hipMemGetInfo(&free_bytes, &total_bytes) == hipSuccess && free_bytes / 4 < budget) budget = free_bytes / 4;
        list_capacity = budget / (4 * (nc + ng)) / kBlock * kBlock;
        if (list_capacity > max_blocks * kBlock) list_capacity = max_blocks * kBlock;
        if (list_capacity < kBlock) list_capacity = kBlock;
        tiles = (max_blocks + list_capacity / kBlock - 1) / (list_capacity / kBlock);
    }
    const size_t slots_cl = (size_t)n_counters * tiles;
    const size_t q_words = (size_t)n_counters * kShards * kCountStride;
    const size_t cl_words = slots_cl * nc * kCountStride, grp_words = slots_cl * ng * kCountStride;
    int rc;
    if ((rc = work_alloc(s, &s->w.ray_o, n_paths)) || (rc = work_alloc(s, &s->w.ray_d, n_paths)) ||
        (rc = work_alloc(s, &s->w.thr, n_paths)) || (rc = work_alloc(s, &s->w.rad, n_paths)) ||
        (rc = work_alloc(s, &s->w.pend, (size_t)n_paths * nl)) || (rc = work_alloc(s, &s->w.hit, slots)) ||
        (rc = work_alloc(s, &s->w.q_o, queue_entries)) || (rc = work_alloc(s, &s->w.q_d, queue_entries)) ||
        (rc = work_alloc(s, &s->w.q_count, q_words + cl_words + grp_words)) || // one block: cleared with one fill
        (rc = work_alloc(s, &s->w.cast_count, (size_t)n_counters * kCastShards * kCountStride)) ||
        (rc = work_alloc(s, &s->w.cl_list, nc * list_capacity)) ||
        (rc = work_alloc(s, &s->w.grp_list, ng * list_capacity))) {
        free_work(s);
        return rc;
    }
    s->w.cl_count = s->w.q_count + q_words;
    s->w.grp_count = s->w.cl_count + cl_words;
    s->counter_words = q_words + cl_words + grp_words;
    s->w.shard_capacity = (uint32_t)shard_capacity;
    s->w.list_capacity = (uint32_t)list_capacity;
    s->tiles = (uint32_t)tiles;
    s->work_paths = n_paths;
    s->n_counters = n_counters;
    return RT_OK;
}

Frame make_frame(const rt_camera* cam, const rt_options* opt) {
    Frame f;
    std::memcpy(f.rot, cam->rot, sizeof f.rot);
    std::memcpy(f.pos, cam->pos, sizeof f.pos);
    f.fx = cam->fx; f.fy = cam->fy; f.cx = cam->cx; f.cy = cam->cy;
    f.width = cam->width;
    f.n_pixels = cam->width * cam->height;
    f.pixel_offset = 0;
    f.n_paths = 0;
    f.sample_begin = 0;
    f.seed = opt->seed;
    f.indirect = opt->indirect != 0;
    f.antialiasing = opt->antialiasing != 0;
    f.bounding_box = opt->bounding_box != 0;
    f.max_depth = opt->max_depth;
    f.count_casts = 0;
    return f;
}

// One k_mesh stage: the queued rays of `counter` against the mesh, either by
// the exhaustive scan or through the cluster lists.
int launch_mesh(rt_scene* s, uint32_t counter, hipStream_t stream, bool profile, bool cull, uint32_t batch_paths) {
    if (s->d.n_triangles == 0) return RT_OK;
    // profiling marks: 'B' binning starts, 'M' triangle kernel starts, 'E' it ended
    auto mark = [&](char tag) -> int {
        if (!profile) return RT_OK;
        if (s->event_tags.size() == s->events.size()) {
            hipEvent_t e = nullptr;
            HIP_TRY(hipEventCreate(&e));
            s->events.push_back(e);
        }
        HIP_TRY(hipEventRecord(s->events[s->event_tags.size()], stream));
        s->event_tags.push_back(tag);
        return RT_OK;
    };
    int rc;
    const f2* tri2 = reinterpret_cast<const f2*>(s->d.tri_pair);
    const uint32_t* q_count = s->w.q_count + (size_t)counter * kShards * kCountStride;
    if (cull && s->d.n_clusters > 1) {
        const uint32_t nc = s->d.n_clusters;
        const uint32_t blocks_per_tile = s->w.list_capacity / kBlock;
        for (uint32_t tile = 0; tile < s->tiles; tile++) {
            const size_t slot = (size_t)counter * s->tiles + tile;
            uint32_t* cl_count = s->w.cl_count + slot * nc * kCountStride;
            if ((rc = mark('B')) != RT_OK) return rc;
            const f2* cl_pair = reinterpret_cast<const f2*>(s->d.cl_pair);
            // two ray blocks per workgroup, except for short batches (measured: better from about 2 M paths up)
            const bool wide = batch_paths >= 1536u * 1024u;
            auto bin64 = wide ? k_bin64<2> : k_bin64<1>;
            if (s->d.n_groups == 0) {
                hipLaunchKernelGGL(bin64, dim3(kMeshGrid), dim3(kBlock), 0, stream, cl_pair, nc,
                                   (const float4*)s->w.q_o, (const float4*)s->w.q_d, q_count, s->w.shard_capacity,
                                   tile * blocks_per_tile, blocks_per_tile, cl_count, s->w.cl_list, s->w.list_capacity);
            } else {
                const uint32_t ng = s->d.n_groups;
                uint32_t* grp_count = s->w.grp_count + slot * ng * kCountStride;
                hipLaunchKernelGGL(bin64, dim3(kMeshGrid), dim3(kBlock), 0, stream, reinterpret_cast<const f2*>(s->d.grp_pair), ng,
                                   (const float4*)s->w.q_o, (const float4*)s->w.q_d, q_count, s->w.shard_capacity,
                                   tile * blocks_per_tile, blocks_per_tile, grp_count, s->w.grp_list, s->w.list_capacity);
                hipLaunchKernelGGL(k_bin_groups, dim3(kMeshGrid), dim3(kBlock), 0, stream, cl_pair, nc, ng,
                                   (const float4*)s->w.q_o, (const float4*)s->w.q_d, (const uint32_t*)grp_count,
                                   (const uint32_t*)s->w.grp_list, cl_count, s->w.cl_list, s->w.list_capacity);
            }
            if ((rc = mark('M')) != RT_OK) return rc;
            hipLaunchKernelGGL(k_mesh_clusters, dim3(kMeshGrid), dim3(kBlock), 0, stream, tri2, s->d.tri_orig, s->d.n_triangles, nc,
                               s->d.cluster_pairs, (const float4*)s->w.q_o, (const float4*)s->w.q_d, (const uint32_t*)cl_count,
                               (const uint32_t*)s->w.cl_list, s->w.list_capacity, s->w.hit);
            if ((rc = mark('E')) != RT_OK) return rc;
        }
    } else {
        if ((rc = mark('M')) != RT_OK) return rc;
        hipLaunchKernelGGL(k_mesh, dim3(kMeshGrid), dim3(kBlock), 0, stream, tri2, s->d.tri_orig, s->d.n_triangles,
                           (const float4*)s->w.q_o, (const float4*)s->w.q_d, q_count, s->w.shard_capacity, s->w.hit);
        if ((rc = mark('E')) != RT_OK) return rc;
    }
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

} // namespace

extern "C" int rt_init(int32_t device) {
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) { rt_set_error("rt_init: device %d of %d", device, count); return RT_ERR_INVALID; }
    HIP_TRY(hipSetDevice(device));
    return RT_OK;
}

extern "C" int rt_scene_upload(const rt_scene_desc* d, rt_scene** out) {
    if (!out) { rt_set_error("rt_scene_upload: null output"); return RT_ERR_INVALID; }
    *out = nullptr;
    int rc = validate(d);
    if (rc != RT_OK) return rc;

    std::vector<float4> sphere(d->n_spheres);
    std::vector<int32_t> sphere_mat(d->sphere_material, d->sphere_material + d->n_spheres);
    for (uint32_t i = 0; i < d->n_spheres; i++)
        sphere[i] = make_float4(d->sphere_pos[3 * i], d->sphere_pos[3 * i + 1], d->sphere_pos[3 * i + 2], d->sphere_radius[i]);
    std::vector<float4> mat_color(d->n_materials);
    std::vector<float2> mat_ior(d->n_materials);
    for (uint32_t i = 0; i < d->n_materials; i++) {
        float type_bits;
        std::memcpy(&type_bits, &d->material_type[i], sizeof type_bits);
        mat_color[i] = make_float4(d->material_color[3 * i], d->material_color[3 * i + 1], d->material_color[3 * i + 2], type_bits);
        mat_ior[i] = make_float2(d->material_ro[i], d->material_ri[i]);
    }
    // Ray-independent part of the reference's per-test work, done once:
    // e1 = B - A, e2 = C - A, N = e1 x e2 (reference src/raytrace.cu:155-158);
    // plain float subtractions and products, so the stored values are the
    // ones the reference recomputes for every ray.
    std::vector<float4> geom(3 * (size_t)d->n_triangles), shade(3 * (size_t)d->n_triangles);
    float mesh_lo[3] = {INFINITY, INFINITY, INFINITY}, mesh_hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = 0; i < d->n_triangles; i++) {
        const float* A = d->vertices + 3 * d->tri_v[3 * i + 0];
        const float* B = d->vertices + 3 * d->tri_v[3 * i + 1];
        const float* C = d->vertices + 3 * d->tri_v[3 * i + 2];
        const float e1[3] = {B[0] - A[0], B[1] - A[1], B[2] - A[2]};
        const float e2[3] = {C[0] - A[0], C[1] - A[1], C[2] - A[2]};
        const float N[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
        geom[3 * i + 0] = make_float4(A[0], A[1], A[2], e1[0]);
        geom[3 * i + 1] = make_float4(e1[1], e1[2], e2[0], e2[1]);
        geom[3 * i + 2] = make_float4(e2[2], N[0], N[1], N[2]);
        float mat_bits;
        std::memcpy(&mat_bits, &d->tri_material[i], sizeof mat_bits);
        for (int k = 0; k < 3; k++) {
            const float* n = d->normals + 3 * d->tri_n[3 * i + k];
            shade[3 * i + k] = make_float4(n[0], n[1], n[2], k == 0 ? mat_bits : 0.f);
            const float* v = d->vertices + 3 * d->tri_v[3 * i + k];
            for (int a = 0; a < 3; a++) {
                if (v[a] < mesh_lo[a]) mesh_lo[a] = v[a];
                if (v[a] > mesh_hi[a]) mesh_hi[a] = v[a];
            }
        }
    }
    // Scan order of the triangle kernel: Morton order of the centroids, so
    // that consecutive triangles are neighbours in space and a cluster of them
    // has tight bounds.  Hits carry the caller's index (tri_orig) and ties are
    // resolved on it, so the order does not change any result.
    std::vector<uint32_t> order(d->n_triangles);
    {
        std::vector<uint64_t> keyed(d->n_triangles);
        auto spread = [](uint32_t x) {
            x &= 0x3ff;
            x = (x | (x << 16)) & 0x030000ff;
            x = (x | (x << 8)) & 0x0300f00f;
            x = (x | (x << 4)) & 0x030c30c3;
            x = (x | (x << 2)) & 0x09249249;
            return x;
        };
        for (uint32_t i = 0; i < d->n_triangles; i++) {
            uint32_t code = 0;
            for (int a = 0; a < 3; a++) {
                double c = 0;
                for (int k = 0; k < 3; k++) c += d->vertices[3 * d->tri_v[3 * i + k] + a];
                const double extent = (double)mesh_hi[a] - mesh_lo[a];
                double u = extent > 0 ? (c / 3 - mesh_lo[a]) / extent : 0.0;
                if (!(u >= 0)) u = 0;  // also catches NaN coordinates
                if (u > 1) u = 1;
                code |= spread((uint32_t)(u * 1023.0)) << a;
            }
            keyed[i] = ((uint64_t)code << 32) | i;
        }
        std::sort(keyed.begin(), keyed.end());
        for (uint32_t i = 0; i < d->n_triangles; i++) order[i] = (uint32_t)keyed[i];
    }
    // The same twelve numbers per triangle, interleaved two triangles at a time
    // for the packed-fp32 inner loop; an odd count is padded with an all-zero
    // triangle, whose zero denominator can never produce a hit.
    const size_t n_pairs = ((size_t)d->n_triangles + 1) / 2;
    std::vector<float2> pair(12 * n_pairs, make_float2(0.f, 0.f));
    std::vector<uint32_t> orig(2 * n_pairs, 0xffffffffu);
    for (uint32_t pos = 0; pos < d->n_triangles; pos++) {
        const uint32_t i = order[pos];
        orig[pos] = i;
        const float4 g0 = geom[3 * i + 0], g1 = geom[3 * i + 1], g2 = geom[3 * i + 2];
        const float v[12] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w, g2.x, g2.y, g2.z, g2.w};
        for (int k = 0; k < 12; k++) {
            float2& dst = pair[12 * (size_t)(pos / 2) + k];
            if (pos & 1) dst.y = v[k]; else dst.x = v[k];
        }
    }
    // Clusters of consecutive pairs and their bounds, padded by 0.05 plus a
    // thousandth of the cluster's diagonal.
    uint32_t cluster_pairs = kClusterPairs;
    if (const char* env = std::getenv("RT_AMD_CLUSTER_PAIRS")) { // tuning / test knob: pairs per cluster
        const long v = std::strtol(env, nullptr, 10);
        if (v >= 1 && v <= 4096) cluster_pairs = (uint32_t)v;
    }
    while ((n_pairs + cluster_pairs - 1) / cluster_pairs > kMaxClusters) cluster_pairs *= 2;
    const uint32_t n_clusters = (uint32_t)((n_pairs + cluster_pairs - 1) / cluster_pairs);
    std::vector<float4> cl_min(n_clusters), cl_max(n_clusters);
    std::vector<uint32_t> cluster_tris(n_clusters, 0);
    for (uint32_t c = 0; c < n_clusters; c++) {
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        const size_t begin = (size_t)c * cluster_pairs * 2;
        const size_t end = std::min<size_t>(begin + (size_t)cluster_pairs * 2, d->n_triangles);
        cluster_tris[c] = (uint32_t)(end - begin);
        for (size_t pos = begin; pos < end; pos++)
            for (int k = 0; k < 3; k++) {
                const float* v = d->vertices + 3 * d->tri_v[3 * order[pos] + k];
                for (int a = 0; a < 3; a++) {
                    if (v[a] < lo[a]) lo[a] = v[a];
                    if (v[a] > hi[a]) hi[a] = v[a];
                }
            }
        const float diag = std::sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2]));
        const float pad = 0.05f + 1e-3f * diag;
        cl_min[c] = make_float4(lo[0] - pad, lo[1] - pad, lo[2] - pad, 0.f);
        cl_max[c] = make_float4(hi[0] + pad, hi[1] + pad, hi[2] + pad, 0.f);
    }
    // the same bounds, two clusters per entry for k_bin64; the unused half of
    // the last entry of an odd count is never read as a cluster (k_bin64 skips it)
    std::vector<float2> cl_pair(6 * (((size_t)n_clusters + 1) / 2), make_float2(1e30f, 1e30f));
    for (uint32_t c = 0; c < ((n_clusters + 1) / 2) * 2; c++) {
        const bool real = c < n_clusters;
        const float v[6] = {real ? cl_min[c].x : 1e30f, real ? cl_min[c].y : 1e30f, real ? cl_min[c].z : 1e30f,
                            real ? cl_max[c].x : -1e30f, real ? cl_max[c].y : -1e30f, real ? cl_max[c].z : -1e30f};
        for (int k = 0; k < 6; k++) {
            float2& dst = cl_pair[6 * (size_t)(c / 2) + k];
            if (c & 1) dst.y = v[k]; else dst.x = v[k];
        }
    }
    // groups of kGroupClusters consecutive clusters with the union of their bounds
    const uint32_t n_groups = n_clusters > 64 ? (n_clusters + kGroupClusters - 1) / kGroupClusters : 0;
    std::vector<float2> grp_pair(6 * (((size_t)n_groups + 1) / 2), make_float2(0.f, 0.f));
    for (uint32_t g = 0; g < n_groups; g++) {
        float v[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (uint32_t c = g * kGroupClusters; c < std::min(n_clusters, (g + 1) * kGroupClusters); c++) {
            const float lo[3] = {cl_min[c].x, cl_min[c].y, cl_min[c].z}, hi[3] = {cl_max[c].x, cl_max[c].y, cl_max[c].z};
            for (int a = 0; a < 3; a++) {
                if (lo[a] < v[a]) v[a] = lo[a];
                if (hi[a] > v[3 + a]) v[3 + a] = hi[a];
            }
        }
        for (int k = 0; k < 6; k++) {
            float2& dst = grp_pair[6 * (size_t)(g / 2) + k];
            if (g & 1) dst.y = v[k]; else dst.x = v[k];
        }
    }
    std::vector<float4> box_min(d->n_boxes), box_max(d->n_boxes), light(d->n_lights);
    for (uint32_t i = 0; i < d->n_boxes; i++) {
        box_min[i] = make_float4(d->box_min[3 * i], d->box_min[3 * i + 1], d->box_min[3 * i + 2], 0.f);
        box_max[i] = make_float4(d->box_max[3 * i], d->box_max[3 * i + 1], d->box_max[3 * i + 2], 0.f);
    }
    for (uint32_t i = 0; i < d->n_lights; i++)
        light[i] = make_float4(d->light_pos[3 * i], d->light_pos[3 * i + 1], d->light_pos[3 * i + 2], d->light_intensity[i]);

    rt_scene* s = new rt_scene();
    s->d.n_spheres = d->n_spheres;
    s->d.n_triangles = d->n_triangles;
    s->d.n_boxes = d->n_boxes;
    s->d.n_lights = d->n_lights;
    s->d.n_clusters = n_clusters;
    s->d.cluster_pairs = cluster_pairs;
    s->d.n_groups = n_groups;
    s->cluster_tris = cluster_tris;
    if ((rc = upload(s, sphere, &s->d.sphere)) || (rc = upload(s, sphere_mat, &s->d.sphere_mat)) ||
        (rc = upload(s, mat_color, &s->d.mat_color)) || (rc = upload(s, mat_ior, &s->d.mat_ior)) ||
        (rc = upload(s, geom, &s->d.tri_geom)) || (rc = upload(s, pair, &s->d.tri_pair)) || (rc = upload(s, orig, &s->d.tri_orig)) ||
        (rc = upload(s, cl_pair, &s->d.cl_pair)) || (rc = upload(s, grp_pair, &s->d.grp_pair)) || (rc = upload(s, shade, &s->d.tri_shade)) ||
        (rc = upload(s, box_min, &s->d.box_min)) || (rc = upload(s, box_max, &s->d.box_max)) ||
        (rc = upload(s, light, &s->d.light))) {
        rt_scene_free(s);
        return rc;
    }
    *out = s;
    return RT_OK;
}

extern "C" int rt_scene_free(rt_scene* s) {
    if (!s) return RT_OK;
    free_work(s);
    for (void* p : s->allocations) (void)hipFree(p);
    for (hipEvent_t e : s->events) (void)hipEventDestroy(e);
    if (s->frame_begin) (void)hipEventDestroy(s->frame_begin);
    if (s->frame_end) (void)hipEventDestroy(s->frame_end);
    if (s->frame_sum) (void)hipFree(s->frame_sum);
    if (s->frame_out) (void)hipFree(s->frame_out);
    delete s;
    return RT_OK;
}

extern "C" int rt_set_progress(rt_scene* s, rt_progress_fn fn, void* user) {
    if (!s) { rt_set_error("rt_set_progress: null scene"); return RT_ERR_INVALID; }
    s->progress = fn;
    s->progress_user = fn ? user : nullptr;
    return RT_OK;
}

extern "C" int rt_accumulate(rt_scene* s, const rt_camera* cam, const rt_options* opt, uint32_t sample_begin,
                             uint32_t sample_end, float* d_sum, void* stream_, rt_stats* stats) {
    return rt_accumulate_rows(s, cam, opt, sample_begin, sample_end, 0, cam ? cam->height : 0, d_sum, stream_, stats);
}

extern "C" int rt_accumulate_rows(rt_scene* s, const rt_camera* cam, const rt_options* opt, uint32_t sample_begin,
                                  uint32_t sample_end, uint32_t row_begin, uint32_t row_end, float* d_sum, void* stream_,
                                  rt_stats* stats) {
    int rc = check_frame(s, cam, opt);
    if (rc != RT_OK) return rc;
    if (!d_sum || sample_end < sample_begin) { rt_set_error("rt_accumulate: bad sample range or null buffer"); return RT_ERR_INVALID; }
    if (row_end < row_begin || row_end > cam->height) { rt_set_error("rt_accumulate_rows: rows [%u, %u) of %u", row_begin, row_end, cam->height); return RT_ERR_INVALID; }
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    Frame f = make_frame(cam, opt);
    f.n_pixels = (row_end - row_begin) * cam->width;
    f.pixel_offset = row_begin * cam->width;
    if (stats) std::memset(stats, 0, sizeof *stats);
    if (f.n_pixels == 0) return RT_OK;
    const uint32_t n_pixels = f.n_pixels;
    const uint32_t n_samples = sample_end - sample_begin;
    if (stats) std::memset(stats, 0, sizeof *stats);
    if (n_samples == 0) return RT_OK;

    uint32_t per_batch = kMaxPathsPerBatch / n_pixels;
    if (per_batch == 0) per_batch = 1;
    if (per_batch > n_samples) per_batch = n_samples;
    const uint32_t n_counters = (uint32_t)opt->max_depth + 1u;
    if ((rc = ensure_work(s, per_batch * n_pixels, n_counters)) != RT_OK) return rc;

    const bool profile = stats && opt->profile;
    f.count_casts = stats != nullptr;
    const bool cull = opt->cluster_culling != 0 && s->d.n_clusters > 1;
    const size_t cl_words = cull ? (size_t)n_counters * s->tiles * s->d.n_clusters * kCountStride : 0;
    std::vector<uint32_t> cl_counts(cl_words);
    uint64_t tests = 0;
    s->event_tags.clear();
    if (stats) {
        if (!s->frame_begin) HIP_TRY(hipEventCreate(&s->frame_begin));
        if (!s->frame_end) HIP_TRY(hipEventCreate(&s->frame_end));
        HIP_TRY(hipEventRecord(s->frame_begin, stream));
    }
    std::vector<uint32_t> counts((size_t)n_counters * kShards * kCountStride);
    std::vector<uint32_t> cast_counts(stats ? (size_t)n_counters * kCastShards * kCountStride : 0);
    uint64_t mesh_rays = 0, paths = 0, sphere_casts = 0;
    uint32_t batches = 0;

    for (uint32_t first = 0; first < n_samples; first += per_batch) {
        const uint32_t in_batch = (n_samples - first < per_batch) ? n_samples - first : per_batch;
        f.n_paths = in_batch * n_pixels;
        f.sample_begin = sample_begin + first;
        const dim3 grid((f.n_paths + kBlock - 1) / kBlock), block(kBlock);
        HIP_TRY(hipMemsetAsync(s->w.q_count, 0, s->counter_words * sizeof(uint32_t), stream));
        if (stats) HIP_TRY(hipMemsetAsync(s->w.cast_count, 0, cast_counts.size() * sizeof(uint32_t), stream));
        // paths per lane: as many as still leave 2 048 workgroups (8 per CU)
        if (f.n_paths >= 2048u * kBlock * 4)
            hipLaunchKernelGGL(k_generate<4>, dim3((f.n_paths + kBlock * 4 - 1) / (kBlock * 4)), block, 0, stream, s->d, s->w, f);
        else if (f.n_paths >= 2048u * kBlock * 2)
            hipLaunchKernelGGL(k_generate<2>, dim3((f.n_paths + kBlock * 2 - 1) / (kBlock * 2)), block, 0, stream, s->d, s->w, f);
        else
            hipLaunchKernelGGL(k_generate<1>, grid, block, 0, stream, s->d, s->w, f);
        HIP_TRY(hipGetLastError());
        for (int depth = 0; depth < opt->max_depth; depth++) {
            if ((rc = launch_mesh(s, (uint32_t)depth, stream, profile, cull, f.n_paths)) != RT_OK) return rc;
            hipLaunchKernelGGL(k_shade, grid, block, 0, stream, s->d, s->w, f, depth);
            HIP_TRY(hipGetLastError());
        }
        if ((rc = launch_mesh(s, (uint32_t)opt->max_depth, stream, profile, cull, f.n_paths)) != RT_OK) return rc;
        hipLaunchKernelGGL(k_accumulate, dim3((n_pixels + kBlock - 1) / kBlock), block, 0, stream, s->d, s->w, f,
                           reinterpret_cast<float4*>(d_sum));
        HIP_TRY(hipGetLastError());
        batches++;
        paths += f.n_paths;
        if (s->progress) {
            HIP_TRY(hipStreamSynchronize(stream));
            s->progress(paths, (uint64_t)n_samples * n_pixels, s->progress_user);
        }
        if (stats) { // the counters are reused by the next batch: read them now
            HIP_TRY(hipMemcpyAsync(counts.data(), s->w.q_count, counts.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipMemcpyAsync(cast_counts.data(), s->w.cast_count, cast_counts.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            if (cull) HIP_TRY(hipMemcpyAsync(cl_counts.data(), s->w.cl_count, cl_words * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            uint64_t batch_rays = 0;
            for (size_t i = 0; i < counts.size(); i += kCountStride) batch_rays += counts[i];
            for (size_t i = 0; i < cast_counts.size(); i += kCountStride) sphere_casts += cast_counts[i];
            mesh_rays += batch_rays;
            if (cull) {
                for (size_t i = 0; i < cl_words; i += kCountStride) tests += (uint64_t)cl_counts[i] * s->cluster_tris[(i / kCountStride) % s->d.n_clusters];
            } else {
                tests += batch_rays * s->d.n_triangles;
            }
        }
    }
    if (stats) {
        HIP_TRY(hipEventRecord(s->frame_end, stream));
        HIP_TRY(hipEventSynchronize(s->frame_end));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, s->frame_begin, s->frame_end));
        stats->gpu_ms = ms;
        for (size_t i = 0; i + 1 < s->event_tags.size(); i++) {
            if (s->event_tags[i] == 'E') continue;
            float k = 0.f;
            HIP_TRY(hipEventElapsedTime(&k, s->events[i], s->events[i + 1]));
            if (s->event_tags[i] == 'B') {
                stats->bin_kernel_ms += k;
            } else {
                stats->mesh_kernel_ms += k;
                stats->mesh_kernel_launches++;
            }
        }
        stats->batches = batches;
        stats->paths = paths;
        stats->mesh_rays = mesh_rays;
        stats->triangle_tests = tests;
        stats->sphere_casts = sphere_casts;
    }
    return RT_OK;
}

extern "C" int rt_resolve(const float* d_sum, uint32_t n_pixels, float* d_out_rgb, void* stream_) {
    if (!d_sum || !d_out_rgb || n_pixels == 0) { rt_set_error("rt_resolve: bad argument"); return RT_ERR_INVALID; }
    hipLaunchKernelGGL(k_resolve, dim3((n_pixels + kBlock - 1) / kBlock), dim3(kBlock), 0, static_cast<hipStream_t>(stream_),
                       reinterpret_cast<const float4*>(d_sum), n_pixels, d_out_rgb);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

extern "C" int rt_render(rt_scene* s, const rt_camera* cam, const rt_options* opt, float* out_rgb, rt_stats* stats) {
    int rc = check_frame(s, cam, opt);
    if (rc != RT_OK) return rc;
    if (!out_rgb) { rt_set_error("rt_render: null output"); return RT_ERR_INVALID; }
    const uint32_t n_pixels = cam->width * cam->height;
    if (s->frame_pixels < n_pixels) { // the buffers stay with the scene: no allocation per frame
        if (s->frame_sum) (void)hipFree(s->frame_sum);
        if (s->frame_out) (void)hipFree(s->frame_out);
        s->frame_sum = s->frame_out = nullptr;
        s->frame_pixels = 0;
        if (hipMalloc(reinterpret_cast<void**>(&s->frame_sum), (size_t)n_pixels * 4 * sizeof(float)) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&s->frame_out), (size_t)n_pixels * 3 * sizeof(float)) != hipSuccess) {
            if (s->frame_sum) (void)hipFree(s->frame_sum);
            s->frame_sum = nullptr;
            (void)hipGetLastError();
            rt_set_error("rt_render: out of device memory");
            return RT_ERR_NOMEM;
        }
        s->frame_pixels = n_pixels;
    }
    HIP_TRY(hipMemsetAsync(s->frame_sum, 0, (size_t)n_pixels * 4 * sizeof(float), nullptr));
    if ((rc = rt_accumulate(s, cam, opt, 0, cam->num_rays, s->frame_sum, nullptr, stats)) != RT_OK) return rc;
    if ((rc = rt_resolve(s->frame_sum, n_pixels, s->frame_out, nullptr)) != RT_OK) return rc;
    if (hipMemcpy(out_rgb, s->frame_out, (size_t)n_pixels * 3 * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) {
        rt_set_error("rt_render: device to host copy failed");
        return RT_ERR_DEVICE;
    }
    return RT_OK;
}

namespace {

int probe(rt_scene* s, const rt_options* opt, uint32_t n, const float* a, const float* b, int as_segments, float* out_t,
          int32_t* out_sphere, int32_t* out_triangle, float* out_abg, int32_t* out_visible) {
    if (!s || !opt || !a || !b) { rt_set_error("probe: null argument"); return RT_ERR_INVALID; }
    if (n == 0) return RT_OK;
    if (n > kMaxPathsPerBatch) { rt_set_error("probe: at most %u rays per call", kMaxPathsPerBatch); return RT_ERR_INVALID; }
    int rc = ensure_work(s, n, 2);
    if (rc != RT_OK) return rc;
    rt_camera cam;
    std::memset(&cam, 0, sizeof cam);
    cam.width = n;
    cam.height = 1;
    Frame f = make_frame(&cam, opt);
    f.n_paths = n;
    float *d_a = nullptr, *d_b = nullptr, *d_t = nullptr, *d_abg = nullptr;
    int32_t *d_s = nullptr, *d_tr = nullptr, *d_v = nullptr;
    std::vector<void*> tmp;
    auto grab = [&](void** p, size_t bytes) { hipError_t e = hipMalloc(p, bytes); if (e == hipSuccess) tmp.push_back(*p); return e; };
    auto release = [&]() { for (void* p : tmp) (void)hipFree(p); };
    hipError_t e = hipSuccess;
    if ((e = grab((void**)&d_a, 3 * (size_t)n * 4)) || (e = grab((void**)&d_b, 3 * (size_t)n * 4)) || (e = grab((void**)&d_t, (size_t)n * 4)) ||
        (e = grab((void**)&d_abg, 3 * (size_t)n * 4)) || (e = grab((void**)&d_s, (size_t)n * 4)) || (e = grab((void**)&d_tr, (size_t)n * 4)) ||
        (e = grab((void**)&d_v, (size_t)n * 4))) {
        release();
        rt_set_error("probe: hipMalloc failed: %s", hipGetErrorString(e));
        return RT_ERR_NOMEM;
    }
    const dim3 grid((n + kBlock - 1) / kBlock), block(kBlock);
    rc = RT_OK;
    do {
        if (hipMemcpy(d_a, a, 3 * (size_t)n * 4, hipMemcpyHostToDevice) || hipMemcpy(d_b, b, 3 * (size_t)n * 4, hipMemcpyHostToDevice) ||
            hipMemset(d_abg, 0, 3 * (size_t)n * 4) || hipMemset(s->w.q_count, 0, 2 * kShards * kCountStride * sizeof(uint32_t))) { rc = RT_ERR_DEVICE; break; }
        hipLaunchKernelGGL(k_probe_rays, grid, block, 0, nullptr, s->d, s->w, f, (const float*)d_a, (const float*)d_b, as_segments);
        const bool cull = opt->cluster_culling != 0 && s->d.n_clusters > 1;
        if (cull && hipMemset(s->w.cl_count, 0, (size_t)2 * s->tiles * s->d.n_clusters * kCountStride * sizeof(uint32_t))) { rc = RT_ERR_DEVICE; break; }
        if (cull && s->d.n_groups && hipMemset(s->w.grp_count, 0, (size_t)2 * s->tiles * s->d.n_groups * kCountStride * sizeof(uint32_t))) { rc = RT_ERR_DEVICE; break; }
        if (launch_mesh(s, 0, nullptr, false, cull, n) != RT_OK) { rc = RT_ERR_DEVICE; break; }
        hipLaunchKernelGGL(k_probe_results, grid, block, 0, nullptr, s->d, s->w, f, as_segments, d_t, d_s, d_tr, d_abg, d_v);
        if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { rc = RT_ERR_DEVICE; break; }
        if (as_segments) {
            if (hipMemcpy(out_visible, d_v, (size_t)n * 4, hipMemcpyDeviceToHost)) rc = RT_ERR_DEVICE;
        } else {
            std::vector<float> abg(3 * (size_t)n);
            std::vector<int32_t> tri(n);
            if (hipMemcpy(out_t, d_t, (size_t)n * 4, hipMemcpyDeviceToHost) || hipMemcpy(out_sphere, d_s, (size_t)n * 4, hipMemcpyDeviceToHost) ||
                hipMemcpy(tri.data(), d_tr, (size_t)n * 4, hipMemcpyDeviceToHost) || hipMemcpy(abg.data(), d_abg, 3 * (size_t)n * 4, hipMemcpyDeviceToHost)) {
                rc = RT_ERR_DEVICE;
                break;
            }
            for (uint32_t i = 0; i < n; i++) {
                out_triangle[i] = tri[i];
                if (tri[i] >= 0 && out_abg) { out_abg[3 * i] = abg[3 * i]; out_abg[3 * i + 1] = abg[3 * i + 1]; out_abg[3 * i + 2] = abg[3 * i + 2]; }
            }
        }
    } while (0);
    if (rc != RT_OK) rt_set_error("probe: a HIP call failed: %s", hipGetErrorString(hipGetLastError()));
    release();
    return rc;
}

} // namespace

extern "C" int rt_cast(rt_scene* s, const rt_options* opt, uint32_t n, const float* origins, const float* dirs, float* out_t,
                       int32_t* out_sphere, int32_t* out_triangle, float* out_abg) {
    if (n && (!out_t || !out_sphere || !out_triangle)) { rt_set_error("rt_cast: null output"); return RT_ERR_INVALID; }
    return probe(s, opt, n, origins, dirs, 0, out_t, out_sphere, out_triangle, out_abg, nullptr);
}

extern "C" int rt_visible(rt_scene* s, const rt_options* opt, uint32_t n, const float* origins, const float* targets,
                          int32_t* out_visible) {
    if (n && !out_visible) { rt_set_error("rt_visible: null output"); return RT_ERR_INVALID; }
    return probe(s, opt, n, origins, targets, 1, nullptr, nullptr, nullptr, nullptr, out_visible);
}
