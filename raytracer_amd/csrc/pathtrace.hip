// pathtrace.hip -- the device half of the C ABI: scene upload, work buffers,
// kernel launches (rt_scene_upload, rt_accumulate, rt_resolve, rt_render,
// rt_cast, rt_visible).  The gfx950 kernels live in three headers:
//
//   device_scene.hip.h   layouts in HBM, vector arithmetic, sphere / box tests, ray queue
//   mesh_kernels.hip.h   k_mesh (exhaustive scan), k_bin64 / k_group_clusters / k_bin_groups / k_mesh_clusters (culling)
//   path_kernels.hip.h   k_generate, k_shade, k_accumulate, k_resolve, known-answer test kernels
//
// The reference runs one CUDA thread per pixel through the whole sample and
// bounce loop (reference src/raytrace.cu:446-507).  Here a frame is a batch of
// independent PATHS (one per pixel and sample) advanced one bounce at a time:
//
//   k_generate           camera sample, nearest sphere, box test
//   for each bounce:
//     triangle stage     the queued rays against the mesh: k_mesh, or
//                        k_bin64 and k_group_clusters (more than 1 024 clusters: k_bin64, k_bin_groups,
//                        k_plan_runs, k_group_clusters -- three culling levels)
//     k_shade            hit selection, material response, shadow rays,
//                        next ray, nearest sphere, box test
//   triangle stage       shadow rays of the last bounce
//   k_accumulate         per pixel: sum the sample colours in sample order
//
// Only rays that pass the bounding-box test are queued for the triangle stage;
// the queue is dense, so all 64 lanes of a wavefront walk triangles with work
// to do.  There a lane owns a ray and the triangle index is wave-uniform, so
// triangle data comes in through scalar loads and lives in SGPRs.
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
#include <mutex>
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

#include "device_scene.hip.h"
#include "mesh_kernels.hip.h"
#include "path_kernels.hip.h"


// ------------------------------------------------------------- host side

// The work buffers of one batch in flight: path state, ray queue, live lists, counters and cluster
// lists, all carved from ONE block of device memory (layout_work).
struct WorkSet {
    Work w{};
    uint32_t paths = 0;        // batches of up to this many paths fit
    uint32_t n_counters = 0;   // ... with up to this many triangle stages
    uint32_t tiles = 1;        // cluster-list tiles per triangle stage
    size_t counter_words = 0;  // queue, live, cluster and group counters: one block starting at w.q_count
    size_t bytes = 0;          // size of the block
    // three culling levels (rt_scene::n_super > 0): the lists of the super-groups, their counters (n_super per stage
    // slot, right after w.grp_count's) and the run table k_plan_runs leaves for k_group_clusters
    uint32_t* sup_count = nullptr;
    uint32_t* sup_list = nullptr;
    uint32_t* run_plan = nullptr;
    unsigned long long* tests_sum = nullptr; // statistics: triangle tests of a batch, summed on the device (k_count_tests)
};

struct rt_scene {
    DeviceScene d{};
    int device = -1;                     // the HIP device the scene was uploaded to
    std::vector<void*> allocations;
    // batch work buffers, grown on demand
    WorkSet ws{};
    void* work_block = nullptr;
    float4* sphere_rad = nullptr;        // per-sample colours of a scene without triangles (accumulate_spheres)
    size_t sphere_rad_paths = 0;
    // Three culling levels (meshes of more than kFusedMaxGroups groups of clusters): super-groups of super_groups
    // consecutive groups with the union of their bounds, at most 64 of at most 64.  0: two levels.
    uint32_t n_super = 0, super_groups = 0;
    const float2* sup_pair = nullptr;
    // The bounds of clusters, groups and super-groups once more with the small padding the fast arithmetic needs
    // (rt_scene_upload); d.cl_pair, d.grp_pair and sup_pair carry the padding of the exact arithmetic.
    const float2* cl_pair_tight = nullptr;
    const float2* grp_pair_tight = nullptr;
    const float2* sup_pair_tight = nullptr;
    // profiling events
    std::vector<hipEvent_t> events;   // pool, reused by every call
    std::vector<char> event_tags;     // tag of the events in use, in record order
    hipEvent_t frame_begin = nullptr, frame_end = nullptr;
    bool marks_pending = false;       // a call with rt_options.profile == 2 left timing marks for rt_profile_collect
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
    if (opt->arithmetic != 0 && opt->arithmetic != 1) { rt_set_error("arithmetic %d is neither 0 (exact) nor 1 (fast)", opt->arithmetic); return RT_ERR_INVALID; }
    return RT_OK;
}

void free_work(rt_scene* s) {
    if (s->work_block) (void)hipFree(s->work_block);
    s->work_block = nullptr;
    s->ws = WorkSet{};
}

// Hands out consecutive 256-byte aligned pieces of a block; with a null base it only adds up sizes.
struct Carver {
    unsigned char* base;
    size_t used = 0;
    template <typename T>
    T* take(size_t count) {
        used = (used + 255) & ~size_t(255);
        T* p = base ? reinterpret_cast<T*>(base + used) : nullptr;
        used += (count ? count : 1) * sizeof(T);
        return p;
    }
};

// Lays the work buffers for batches of up to n_paths paths and n_counters triangle stages out in
// `arena` (null: only ws->bytes and the capacities are computed).  The cluster lists get at most
// list_budget bytes; a triangle stage is cut into tiles of rays when that does not cover the queue.
// The scene's tables fit the LDS cache of the path kernels (SceneCacheT): their kFits instances apply.
bool scene_fits_cache(const DeviceScene& d) {
    return d.n_spheres <= kSphereCache && d.n_materials <= kMatCache && d.n_lights <= kLightCache && d.n_boxes <= kBoxCache;
}

// Fewest wavefronts a long input of a kernel whose paths stay in their lanes is spread over (stretch_of).
// (Validation knob: small frames get short stretches; RT_AMD_STRETCH_WAVES=1 gives them the longest.)
uint32_t stretch_min_waves() {
    static const int forced = [] { const char* e = std::getenv("RT_AMD_STRETCH_WAVES"); return e ? std::atoi(e) : 0; }();
    return forced > 0 ? (uint32_t)forced : kMinStretchWaves;
}

int layout_work(const DeviceScene& d, uint32_t n_super, uint32_t n_paths, uint32_t n_counters, size_t list_budget, unsigned char* arena, WorkSet* ws) {
    const size_t nl = d.n_lights ? d.n_lights : 1;
    // a shard receives the rays of every kShards-th wavefront of a path kernel
    // (k_shade walks its list of live paths shard by shard, every shard ending in a partial workgroup:
    // up to kShards workgroups more than the paths need)
    // (or of every kShards-th workgroup of the kernels that reserve by workgroup: what is reserved here covers both)
    const size_t n_groups_of_paths = ((size_t)n_paths + kBlock - 1) / kBlock + kShards;
    // (a wavefront of the kernels whose paths stay in their lanes writes up to a stretch of S entries of its input, and
    // of w = ceil(n / S) wavefronts a shard serves ceil(w / kShards): less than n / kShards + 2 S entries; S grows with n)
    const size_t longest_stretch = stretch_of((uint32_t)std::min<size_t>(n_groups_of_paths * kBlock, 0xffffffffu), stretch_min_waves());
    const size_t per_shard = ((n_groups_of_paths + kShards - 1) / kShards) * kBlock + 2 * longest_stretch;
    // k_shade: a wavefront queues at most 64 * (1 + lights) rays; k_generate: 64 * kGenPaths, from fewer wavefronts
    size_t shard_capacity = per_shard * (1 + nl);
    for (size_t r = 2; r <= kGenPaths; r *= 2) {
        const size_t gen_waves = (((size_t)n_paths + kBlock * r - 1) / (kBlock * r)) * kWavesPerBlock;
        shard_capacity = std::max(shard_capacity, ((gen_waves + kShards - 1) / kShards) * 64 * r);
    }
    const size_t queue_entries = shard_capacity * kShards;
    if (queue_entries > 0xffffffffu) { rt_set_error("batch too large for the ray queue"); return RT_ERR_INVALID; }
    // cluster lists: one tile of rays at a time if the whole queue would not fit the budget
    const size_t nc = d.n_clusters, ng = d.n_groups;
    const size_t lists_nc = d.fused_groups ? 0 : nc; // k_group_clusters keeps the per-cluster lists in LDS
    size_t list_capacity = 0, tiles = 1;
    if (nc > 1) {
        // a tile is a range of 256-ray blocks; every shard may end in a partial block
        const size_t max_blocks = ((shard_capacity + kBlock - 1) / kBlock) * kShards;
        list_capacity = list_budget / (4 * (lists_nc + ng + n_super)) / kBlock * kBlock;
        if (list_capacity > max_blocks * kBlock) list_capacity = max_blocks * kBlock;
        if (list_capacity < kBlock) list_capacity = kBlock;
        tiles = (max_blocks + list_capacity / kBlock - 1) / (list_capacity / kBlock);
    }
    const size_t slots_cl = (size_t)n_counters * tiles;
    const size_t q_words = (size_t)n_counters * kShards * kCountStride;
    const size_t cl_words = slots_cl * nc * kCountStride, grp_words = slots_cl * (ng + n_super) * kCountStride;
    const size_t live_capacity = per_shard; // a wavefront keeps at most its 64 paths, a workgroup its 256
    const size_t live_total = kShards * live_capacity; // >= n_paths: buffer 0 also holds the camera rays at the paths' own indices
    if (2 * live_total + (size_t)n_paths * nl > 0xffffffffu) { rt_set_error("batch too large for 32-bit hit slots"); return RT_ERR_INVALID; }
    Carver c{arena};
    Work& w = ws->w;
    float4* rays = c.take<float4>(4 * live_total); // (ray_o, ray_d) records, two buffers of live_total
    w.ray_o.base = rays;
    w.ray_d.base = rays + 1;
    w.thr = c.take<float4>(2 * live_total);
    w.rad = c.take<float4>(n_paths);
    w.gen = c.take<uint32_t>(n_paths);
    w.pend = c.take<float4>((size_t)n_paths * nl);
    w.hit = c.take<unsigned long long>(2 * live_total + (size_t)n_paths * nl);
    w.q_o = c.take<float4>(2 * queue_entries);
    w.q_d = w.q_o + 1;
    w.q_count = c.take<uint32_t>(2 * q_words + cl_words + grp_words); // one block: cleared with one fill
    w.live_count = w.q_count + q_words;
    w.cl_count = w.live_count + q_words;
    w.grp_count = w.cl_count + cl_words;
    w.live = c.take<uint32_t>(2 * kShards * live_capacity);
    w.cast_count = c.take<uint32_t>((size_t)(n_counters + 1) * kCastShards * kCountStride);
    w.bounce_count = w.cast_count + (size_t)n_counters * kCastShards * kCountStride;
    w.cl_list = c.take<uint32_t>(lists_nc * list_capacity);
    w.grp_list = c.take<uint32_t>(ng * list_capacity);
    ws->sup_count = w.grp_count + slots_cl * ng * kCountStride;
    ws->sup_list = c.take<uint32_t>(n_super * list_capacity);
    ws->run_plan = c.take<uint32_t>(n_super ? kMaxClusters + 2 : 0);
    ws->tests_sum = c.take<unsigned long long>(1);
    w.live_total = (uint32_t)live_total;
    w.live_capacity = (uint32_t)live_capacity;
    w.shard_capacity = (uint32_t)shard_capacity;
    w.list_capacity = (uint32_t)list_capacity;
    ws->counter_words = 2 * q_words + cl_words + grp_words;
    ws->tiles = (uint32_t)tiles;
    ws->paths = n_paths;
    ws->n_counters = n_counters;
    ws->bytes = c.used;
    return RT_OK;
}

// The scene's own work buffers for batches of up to n_paths paths and n_counters triangle stages.
int ensure_work(rt_scene* s, uint32_t n_paths, uint32_t n_counters) {
    if (s->ws.paths >= n_paths && s->ws.n_counters >= n_counters) return RT_OK;
    n_paths = std::max(n_paths, s->ws.paths);          // never shrink one dimension to grow the other
    n_counters = std::max(n_counters, s->ws.n_counters);
    free_work(s);
    size_t budget = kClusterListBudget, free_bytes = 0, total_bytes = 0;
    if (const char* env = std::getenv("RT_AMD_LIST_GIB")) { // tuning knob
        const long v = std::strtol(env, nullptr, 10);
        if (v >= 1 && v <= 128) budget = (size_t)v << 30;
    }
    if (hipMemGetInfo(&free_bytes, &total_bytes) == hipSuccess && free_bytes / 4 < budget) budget = free_bytes / 4;
    WorkSet ws;
    int rc = layout_work(s->d, s->n_super, n_paths, n_counters, budget, nullptr, &ws);
    if (rc != RT_OK) return rc;
    void* block = nullptr;
    if (hipMalloc(&block, ws.bytes) != hipSuccess) {
        (void)hipGetLastError();
        rt_set_error("out of device memory: %zu bytes of work buffers for batches of %u paths", ws.bytes, n_paths);
        return RT_ERR_NOMEM;
    }
    if ((rc = layout_work(s->d, s->n_super, n_paths, n_counters, budget, static_cast<unsigned char*>(block), &ws)) != RT_OK) {
        (void)hipFree(block);
        return rc;
    }
    s->work_block = block;
    s->ws = ws;
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
    f.prune_zero_terms = 0;
    f.min_waves = stretch_min_waves();
    return f;
}

// Profiling mark on the stream: 'B' binning starts, 'M' triangle kernel starts, 'E' it ended,
// 'S' k_shade starts.
int mark_event(rt_scene* s, hipStream_t stream, char tag) {
    if (s->event_tags.size() == s->events.size()) {
        hipEvent_t e = nullptr;
        HIP_TRY(hipEventCreate(&e));
        s->events.push_back(e);
    }
    HIP_TRY(hipEventRecord(s->events[s->event_tags.size()], stream));
    s->event_tags.push_back(tag);
    return RT_OK;
}

// One k_mesh stage: the queued rays of `counter` against the mesh, either by
// the exhaustive scan or through the cluster lists.
// fast: rt_options.arithmetic = 1 (the triangle filter of k_group_clusters in fused arithmetic).
// ray_bound: most rays the stage's queue can hold (0: unknown) -- the camera rays of a batch are at most its paths --,
// which spares a mesh whose lists need several tiles per stage the launches of tiles that are certainly empty.
int launch_mesh(rt_scene* s, const WorkSet& ws, uint32_t counter, hipStream_t stream, bool profile, bool cull, uint32_t batch_paths,
                bool count_tests, bool fast, size_t ray_bound = 0) {
    const Work& w = ws.w;
    if (s->d.n_triangles == 0) return RT_OK;
    auto mark = [&](char tag) -> int { return profile ? mark_event(s, stream, tag) : RT_OK; };
    int rc;
    const f2* tri2 = reinterpret_cast<const f2*>(s->d.tri_pair);
    const uint32_t* q_count = w.q_count + (size_t)counter * kShards * kCountStride;
    if (cull && s->d.n_clusters > 1) {
        const uint32_t nc = s->d.n_clusters;
        const uint32_t blocks_per_tile = w.list_capacity / kBlock;
        uint32_t tiles = ws.tiles;
        if (ray_bound) // every shard may end in a partial block
            tiles = (uint32_t)std::min<size_t>(tiles, ((ray_bound + kBlock - 1) / kBlock + kShards + blocks_per_tile - 1) / blocks_per_tile);
        for (uint32_t tile = 0; tile < tiles; tile++) {
            const size_t slot = (size_t)counter * ws.tiles + tile;
            uint32_t* cl_count = w.cl_count + slot * nc * kCountStride;
            if ((rc = mark('B')) != RT_OK) return rc;
            const f2* cl_pair = reinterpret_cast<const f2*>(fast ? s->cl_pair_tight : s->d.cl_pair);
            // two ray blocks per workgroup, except for short batches (measured: better from about 2 M paths up)
            bool wide = batch_paths >= 1536u * 1024u;
            bool wider = batch_paths >= 6u * 1024u * 1024u;
            // Half the resident workgroups for batches below 6 M paths: every workgroup of these grids pays a
            // set-up (shard map, job table) whether it finds work or not, and with several frames in flight
            // the short stages of small batches are mostly set-up.  (Measured with three frames in flight:
            // 2 048 workgroups are best at 8.4 M paths, 1 024 at 4.2 M and below -- 9 % at 1 M; a frame
            // rendered alone loses 2 % against 4 096.)
            static const int forced_grid = [] { const char* e = std::getenv("RT_AMD_BIN_GRID"); return e ? std::atoi(e) : 0; }(); // tuning knob
            const uint32_t bin_grid = forced_grid > 0 ? (uint32_t)forced_grid : batch_paths >= 6u * 1024u * 1024u ? kMeshGrid : kMeshGrid / 2, cl_grid = bin_grid;
            static const int forced = [] { const char* e = std::getenv("RT_AMD_BIN_RAYS"); return e ? std::atoi(e) : 0; }(); // tuning knob
            if (forced == 1) wide = wider = false;
            if (forced == 2) { wide = true; wider = false; }
            if (forced == 4) wide = wider = true;
            auto bin64 = wider ? k_bin64<4> : wide ? k_bin64<2> : k_bin64<1>;
            const uint32_t ng = s->d.n_groups;
            uint32_t* grp_count = w.grp_count + slot * ng * kCountStride;
            const f2* grp_pair = reinterpret_cast<const f2*>(fast ? s->grp_pair_tight : s->d.grp_pair);
            static const int fused_grid = [] { const char* e = std::getenv("RT_AMD_FUSED_GRID"); return e ? std::atoi(e) : 0; }(); // tuning knob
            if (s->n_super) {
                // three levels: the rays against the super-groups, each super-group's list against its groups (lists in
                // global memory both), then the groups' lists against their clusters and the clusters' triangles
                // (k_group_clusters, the last level in LDS), with the table of runs made once by k_plan_runs
                uint32_t* sup_count = ws.sup_count + slot * s->n_super * kCountStride;
                hipLaunchKernelGGL(bin64, dim3(bin_grid), dim3(kBlock), 0, stream, reinterpret_cast<const f2*>(fast ? s->sup_pair_tight : s->sup_pair), s->n_super,
                                   (const float4*)w.q_o, (const float4*)w.q_d, q_count, w.shard_capacity,
                                   tile * blocks_per_tile, blocks_per_tile, sup_count, ws.sup_list, w.list_capacity);
                auto bin_groups = wider ? k_bin_groups<4> : wide ? k_bin_groups<2> : k_bin_groups<1>;
                hipLaunchKernelGGL(bin_groups, dim3(bin_grid), dim3(kBlock), 0, stream, grp_pair, ng, s->n_super, s->super_groups,
                                   (const float4*)w.q_o, (const float4*)w.q_d, (const uint32_t*)sup_count,
                                   (const uint32_t*)ws.sup_list, grp_count, w.grp_list, w.list_capacity);
                if ((rc = mark('M')) != RT_OK) return rc;
                const uint32_t grid = fused_grid > 0 ? (uint32_t)fused_grid : batch_paths >= 6u * 1024u * 1024u ? 6144u : batch_paths >= 2u * 1024u * 1024u ? 3072u : 1024u;
                hipLaunchKernelGGL(k_plan_runs, dim3(1), dim3(1024), 0, stream, (const uint32_t*)grp_count, ng, grid, 1024u, ws.run_plan);
                hipLaunchKernelGGL((fast ? k_group_clusters<2, 512, 2, true, true> : k_group_clusters<2, 512, 2, true, false>), dim3(grid), dim3(512), 0, stream,
                                   tri2, s->d.tri_orig, s->d.n_triangles, nc,
                                   s->d.cluster_pairs, ng, s->d.group_clusters, cl_pair, (const float4*)w.q_o, (const float4*)w.q_d,
                                   (const uint32_t*)grp_count, (const uint32_t*)w.grp_list, w.list_capacity, w.hit,
                                   count_tests ? cl_count : (uint32_t*)nullptr, (const uint32_t*)ws.run_plan);
                if ((rc = mark('E')) != RT_OK) return rc;
                continue;
            }
            hipLaunchKernelGGL(bin64, dim3(bin_grid), dim3(kBlock), 0, stream, grp_pair, ng,
                               (const float4*)w.q_o, (const float4*)w.q_d, q_count, w.shard_capacity,
                               tile * blocks_per_tile, blocks_per_tile, grp_count, w.grp_list, w.list_capacity);
            if (s->d.fused_groups) {
                // second level and triangle scan in one kernel: the per-cluster lists live in LDS
                if ((rc = mark('M')) != RT_OK) return rc;
                // 512 threads bin 1 024 list entries per pass (measured on the cat frame, 256 / 512 / 1 024 threads:
                // 1.03 / 0.91 / 1.13 ms per frame).  Many more workgroups than fit the chip: a run is then one or
                // two passes and the runs even out (768 / 3 072 / 6 144 workgroups: 1.20 / 0.90 / 0.87 ms).
                const uint32_t grid = fused_grid > 0 ? (uint32_t)fused_grid : batch_paths >= 6u * 1024u * 1024u ? 6144u : batch_paths >= 2u * 1024u * 1024u ? 3072u : 1024u;
                hipLaunchKernelGGL((fast ? k_group_clusters<2, 512, 2, false, true> : k_group_clusters<2, 512, 2, false, false>), dim3(grid), dim3(512), 0, stream,
                                   tri2, s->d.tri_orig, s->d.n_triangles, nc,
                                   s->d.cluster_pairs, ng, s->d.group_clusters, cl_pair, (const float4*)w.q_o, (const float4*)w.q_d,
                                   (const uint32_t*)grp_count, (const uint32_t*)w.grp_list, w.list_capacity, w.hit,
                                   count_tests ? cl_count : (uint32_t*)nullptr, (const uint32_t*)nullptr);
            } else {
                auto bin_groups = wider ? k_bin_groups<4> : wide ? k_bin_groups<2> : k_bin_groups<1>;
                hipLaunchKernelGGL(bin_groups, dim3(bin_grid), dim3(kBlock), 0, stream, cl_pair, nc, ng, s->d.group_clusters,
                                   (const float4*)w.q_o, (const float4*)w.q_d, (const uint32_t*)grp_count,
                                   (const uint32_t*)w.grp_list, cl_count, w.cl_list, w.list_capacity);
                if ((rc = mark('M')) != RT_OK) return rc;
                hipLaunchKernelGGL(k_mesh_clusters, dim3(cl_grid), dim3(kBlock), 0, stream, tri2, s->d.tri_orig, s->d.n_triangles, nc,
                                   s->d.cluster_pairs, (const float4*)w.q_o, (const float4*)w.q_d, (const uint32_t*)cl_count,
                                   (const uint32_t*)w.cl_list, w.list_capacity, w.hit);
            }
            if ((rc = mark('E')) != RT_OK) return rc;
        }
    } else {
        if ((rc = mark('M')) != RT_OK) return rc;
        hipLaunchKernelGGL(k_mesh, dim3(kMeshGrid), dim3(kBlock), 0, stream, tri2, s->d.tri_orig, s->d.n_triangles,
                           (const float4*)w.q_o, (const float4*)w.q_d, q_count, w.shard_capacity, w.hit);
        if ((rc = mark('E')) != RT_OK) return rc;
    }
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

} // namespace

// ---- rt_accumulate_static: the same frame on a fixed work area -------------
//
// For callers that may not allocate device memory (code under HIP graph capture, serving loops,
// harnesses that forbid it): the work buffers rt_accumulate takes from hipMalloc come from one
// file-scope device array instead.  Every GPU the library runs on holds its own copy of the array,
// at its own address, until the process ends; the name `scratch` is only a host-side stand-in, so
// host code uses the address hipGetSymbolAddress returns for the current device and nothing else.
constexpr size_t kStaticWorkBytes = size_t(64) << 20;
static __device__ __align__(256) unsigned char scratch[kStaticWorkBytes];

namespace {

constexpr int kMaxDevices = 64;
unsigned char* g_static_work[kMaxDevices]; // device address of `scratch` per device, once resolved
std::mutex g_static_work_mutex;

// The current device's copy of `scratch`.
int static_work_area(unsigned char** out) {
    int device = -1;
    HIP_TRY(hipGetDevice(&device));
    if (device < 0 || device >= kMaxDevices) { rt_set_error("static work area: device %d out of range", device); return RT_ERR_INVALID; }
    std::lock_guard<std::mutex> lock(g_static_work_mutex);
    if (!g_static_work[device]) {
        void* p = nullptr;
        HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) { rt_set_error("static work area: null symbol address on device %d", device); return RT_ERR_DEVICE; }
        g_static_work[device] = static_cast<unsigned char*>(p);
    }
    *out = g_static_work[device];
    return RT_OK;
}

} // namespace

extern "C" int rt_init(int32_t device) {
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) { rt_set_error("rt_init: device %d of %d", device, count); return RT_ERR_INVALID; }
    HIP_TRY(hipSetDevice(device));
    // set-up time: resolve this device's copy of the static work area, so that rt_accumulate_static finds it.
    // A failure here is not rt_init's: it is reported by rt_accumulate_static, the only call that needs the area.
    unsigned char* area = nullptr;
    if (static_work_area(&area) != RT_OK) g_error[0] = '\0';
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
    const size_t n_pairs = ((size_t)d->n_triangles + 1) / 2;
    // Clusters of consecutive pairs and their bounds, padded by 0.05 plus a
    // thousandth of the cluster's diagonal.
    // Cluster size: 16 triangles.  Up to 1 024 clusters are the two levels of k_group_clusters (64 groups of 16).
    // (RT_AMD_LEVELS=2 keeps larger meshes on two levels as well: the clusters then grow with the mesh up to about
    // 200 000 triangles, and beyond that clusters of 32 triangles or more get per-cluster lists in global memory --
    // measured against three levels: 39 600 triangles 5.4 against 4.3 ms, 638 400 triangles 14.7 against 7.1 ms.)
    // From kThreeLevelsFromPairs pairs on the clusters stay small and a third level is put on top: super-groups of about
    // sqrt(groups) consecutive groups, binned through lists in global memory like the groups of the two-level path.
    bool three_levels = n_pairs >= kThreeLevelsFromPairs;
    if (const char* env = std::getenv("RT_AMD_LEVELS")) three_levels = std::atoi(env) == 3; // tuning / test knob
    uint32_t cluster_pairs = kClusterPairs;
    if (!three_levels) {
        cluster_pairs = std::max<uint32_t>(kClusterPairs, (uint32_t)((n_pairs + kFusedMaxGroups * kFusedGroup - 1) / (kFusedMaxGroups * kFusedGroup)));
        if (cluster_pairs > kFusedClusterPairsMax) cluster_pairs = 16;
    }
    if (const char* env = std::getenv("RT_AMD_CLUSTER_PAIRS")) { // tuning / test knob: pairs per cluster
        const long v = std::strtol(env, nullptr, 10);
        if (v >= 1 && v <= 4096) cluster_pairs = (uint32_t)v;
    }
    while ((n_pairs + cluster_pairs - 1) / cluster_pairs > (three_levels ? kFusedGroup * kMaxClusters : kMaxClusters)) cluster_pairs *= 2;
    // (a mesh that fits the two levels of k_group_clusters needs no third)
    if ((n_pairs + cluster_pairs - 1) / cluster_pairs <= kFusedMaxGroups * kFusedGroup) three_levels = false;
    uint32_t super_groups = 0, n_super = 0;
    if (three_levels) {
        const uint32_t groups = (uint32_t)(((n_pairs + cluster_pairs - 1) / cluster_pairs + kFusedGroup - 1) / kFusedGroup); // <= kMaxClusters
        super_groups = 2 * (((uint32_t)std::ceil(std::sqrt((double)groups)) + 1) / 2); // even: a super-group starts a box pair
        if (const char* env = std::getenv("RT_AMD_SUPER_GROUPS")) { // tuning / test knob
            const long v = std::strtol(env, nullptr, 10);
            if (v >= 2 && v <= 64 && v % 2 == 0 && (groups + v - 1) / v <= 64) super_groups = (uint32_t)v;
        }
        n_super = (groups + super_groups - 1) / super_groups;
    }
    // Scan order of the triangle kernel: consecutive triangles are neighbours in
    // space, so that a cluster of them has tight bounds -- Morton order of the
    // centroids first (what tiny meshes and RT_AMD_ORDER=morton keep), refined
    // into a k-d order below.  Hits carry the caller's index (tri_orig) and ties
    // are resolved on it, so the order does not change any result.
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
    // Then a k-d ordering on top of it: the triangles are split at the median of their centroids
    // along the axis that leaves the two parts the smallest bounds, at multiples of the group size (16 clusters) while a part
    // is larger than a group (of the super-group size while it is larger than that) and at multiples of the cluster size below that, so that every
    // cluster and every group of 16 consecutive clusters (and every super-group) is a compact cell.  Clusters of
    // consecutive Morton codes straddle the curve's jumps; the cells' bounds are tighter, so a ray
    // touches fewer of them (cat mesh, headline frame: 32 instead of 63 triangle tests per path).
    const char* order_env = std::getenv("RT_AMD_ORDER"); // test knob, read at every upload
    const bool kd_order = !(order_env && std::strcmp(order_env, "morton") == 0);
    if (kd_order && d->n_triangles > 2 * cluster_pairs) {
        std::vector<float> cen(3 * (size_t)d->n_triangles);
        for (uint32_t i = 0; i < d->n_triangles; i++)
            for (int a = 0; a < 3; a++) {
                float c = 0.f;
                for (int k = 0; k < 3; k++) c += d->vertices[3 * d->tri_v[3 * i + k] + a];
                cen[3 * (size_t)i + a] = c == c ? c : 0.f; // NaN coordinates sort as 0
            }
        const size_t cluster_tris = 2 * (size_t)cluster_pairs, group_tris = kFusedGroup * cluster_tris;
        const size_t super_tris = three_levels ? super_groups * group_tris : ~size_t(0);
        std::vector<std::pair<size_t, size_t>> todo{{0, d->n_triangles}};
        while (!todo.empty()) {
            const size_t lo = todo.back().first, hi = todo.back().second;
            todo.pop_back();
            const size_t range = hi - lo;
            if (range <= cluster_tris) continue;
            const size_t unit = range > super_tris ? super_tris : range > group_tris ? group_tris : cluster_tris;
            const size_t units = (range + unit - 1) / unit;
            const size_t mid = lo + (units / 2) * unit;
            // the axis whose median split gives the two parts the smallest total surface area of their bounds
            auto area_of = [&](size_t b, size_t e) {
                float lo3[3] = {INFINITY, INFINITY, INFINITY}, hi3[3] = {-INFINITY, -INFINITY, -INFINITY};
                for (size_t k = b; k < e; k++)
                    for (int c = 0; c < 3; c++) {
                        const float* v = d->vertices + 3 * d->tri_v[3 * order[k] + c];
                        for (int a = 0; a < 3; a++) {
                            if (v[a] < lo3[a]) lo3[a] = v[a];
                            if (v[a] > hi3[a]) hi3[a] = v[a];
                        }
                    }
                const float ex = hi3[0] - lo3[0], ey = hi3[1] - lo3[1], ez = hi3[2] - lo3[2];
                const float area = ex * ey + ey * ez + ez * ex;
                return area == area ? area : INFINITY;
            };
            auto split_on = [&](int axis) {
                std::nth_element(order.begin() + lo, order.begin() + mid, order.begin() + hi, [&](uint32_t x, uint32_t y) {
                    const float cx = cen[3 * (size_t)x + axis], cy = cen[3 * (size_t)y + axis];
                    return cx < cy || (cx == cy && x < y);
                });
            };
            int axis = 0;
            float best_area = INFINITY;
            for (int a = 0; a < 3; a++) {
                split_on(a);
                const float area = area_of(lo, mid) + area_of(mid, hi);
                if (area < best_area) { best_area = area; axis = a; }
            }
            if (axis != 2) split_on(axis);
            todo.push_back({lo, mid});
            todo.push_back({mid, hi});
        }
    }
    // The same twelve numbers per triangle, interleaved two triangles at a time
    // for the packed-fp32 inner loop; an odd count is padded with an all-zero
    // triangle, whose zero denominator can never produce a hit.
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
    const uint32_t n_clusters = (uint32_t)((n_pairs + cluster_pairs - 1) / cluster_pairs);
    // Groups of consecutive clusters with the union of their bounds.  Up to kFusedMaxGroups * kFusedGroup
    // clusters: groups of kFusedGroup clusters, binned and scanned by one kernel (k_group_clusters).
    // Beyond: about sqrt(n) groups of sqrt(n) clusters (an even number, so that a group starts a box
    // pair), at most 64 of at most 64, with per-cluster lists in global memory.
    const bool fused = n_clusters > 1 && (n_clusters <= kFusedMaxGroups * kFusedGroup || three_levels);
    uint32_t group_clusters = fused ? kFusedGroup : 2 * (((uint32_t)std::ceil(std::sqrt((double)n_clusters)) + 1) / 2);
    if (const char* env = std::getenv("RT_AMD_GROUP_CLUSTERS")) { // tuning / test knob
        const long v = std::strtol(env, nullptr, 10);
        if (v >= 2 && v <= (fused ? (long)kFusedGroup : 64) && v % 2 == 0 && (n_clusters + v - 1) / v <= 64) group_clusters = (uint32_t)v;
    }
    const uint32_t n_groups = n_clusters > 1 ? (n_clusters + group_clusters - 1) / group_clusters : 0;
    // The bounds of the clusters (padded by pad_abs plus pad_rel of the cluster's diagonal), of the groups and of the
    // super-groups.  Built twice: with the padding the exact arithmetic needs -- 0.05 plus a thousandth of the diagonal,
    // so that no triangle the reference's own rounding accepts for a ray just outside its true bounds is lost -- and
    // with a small one for the fast arithmetic, whose filter only has to absorb its own rounding (relative 1e-6 of the
    // coordinates at most): a ten-thousandth of the largest coordinate or extent.  Tighter bounds: 3 % fewer list
    // entries and tests on the cat.
    auto build_bounds = [&](float pad_abs, float pad_rel, std::vector<float2>& cl_pair, std::vector<float2>& grp_pair,
                            std::vector<float2>& sup_pair) {
        std::vector<float4> cl_min(n_clusters), cl_max(n_clusters);
        for (uint32_t c = 0; c < n_clusters; c++) {
            float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
            const size_t begin = (size_t)c * cluster_pairs * 2;
            const size_t end = std::min<size_t>(begin + (size_t)cluster_pairs * 2, d->n_triangles);
            for (size_t pos = begin; pos < end; pos++)
                for (int k = 0; k < 3; k++) {
                    const float* v = d->vertices + 3 * d->tri_v[3 * order[pos] + k];
                    for (int a = 0; a < 3; a++) {
                        if (v[a] < lo[a]) lo[a] = v[a];
                        if (v[a] > hi[a]) hi[a] = v[a];
                    }
                }
            const float diag = std::sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2]));
            const float pad = pad_abs + pad_rel * diag;
            cl_min[c] = make_float4(lo[0] - pad, lo[1] - pad, lo[2] - pad, 0.f);
            cl_max[c] = make_float4(hi[0] + pad, hi[1] + pad, hi[2] + pad, 0.f);
        }
        // the same bounds, two clusters per entry for k_bin64; the unused half of
        // the last entry of an odd count is never read as a cluster (k_bin64 skips it)
        cl_pair.assign(6 * (((size_t)n_clusters + 1) / 2), make_float2(1e30f, 1e30f));
        for (uint32_t c = 0; c < ((n_clusters + 1) / 2) * 2; c++) {
            const bool real = c < n_clusters;
            const float v[6] = {real ? cl_min[c].x : 1e30f, real ? cl_min[c].y : 1e30f, real ? cl_min[c].z : 1e30f,
                                real ? cl_max[c].x : -1e30f, real ? cl_max[c].y : -1e30f, real ? cl_max[c].z : -1e30f};
            for (int k = 0; k < 6; k++) {
                float2& dst = cl_pair[6 * (size_t)(c / 2) + k];
                if (c & 1) dst.y = v[k]; else dst.x = v[k];
            }
        }
        grp_pair.assign(6 * (((size_t)n_groups + 1) / 2), make_float2(0.f, 0.f));
        for (uint32_t g = 0; g < n_groups; g++) {
            float v[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
            for (uint32_t c = g * group_clusters; c < std::min(n_clusters, (g + 1) * group_clusters); c++) {
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
        // Three levels: super-groups of consecutive groups, the union of their clusters' bounds.
        sup_pair.assign(6 * (((size_t)n_super + 1) / 2), make_float2(0.f, 0.f));
        for (uint32_t u = 0; u < n_super; u++) {
            float v[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
            const uint32_t per = super_groups * group_clusters;
            for (uint32_t c = u * per; c < std::min(n_clusters, (u + 1) * per); c++) {
                const float lo[3] = {cl_min[c].x, cl_min[c].y, cl_min[c].z}, hi[3] = {cl_max[c].x, cl_max[c].y, cl_max[c].z};
                for (int a = 0; a < 3; a++) {
                    if (lo[a] < v[a]) v[a] = lo[a];
                    if (hi[a] > v[3 + a]) v[3 + a] = hi[a];
                }
            }
            for (int k = 0; k < 6; k++) {
                float2& dst = sup_pair[6 * (size_t)(u / 2) + k];
                if (u & 1) dst.y = v[k]; else dst.x = v[k];
            }
        }
    };
    std::vector<float2> cl_pair, grp_pair, sup_pair, cl_pair_tight, grp_pair_tight, sup_pair_tight;
    build_bounds(0.05f, 1e-3f, cl_pair, grp_pair, sup_pair);
    float scale = 0.f; // largest coordinate or extent of the mesh
    for (int a = 0; a < 3; a++)
        if (d->n_triangles) scale = std::max(scale, std::max(std::max(std::fabs(mesh_lo[a]), std::fabs(mesh_hi[a])), mesh_hi[a] - mesh_lo[a]));
    if (!(scale < INFINITY)) scale = 0.f; // (non-finite coordinates: the exact padding's bounds are as good as any)
    build_bounds(1e-4f * scale, 1e-5f, cl_pair_tight, grp_pair_tight, sup_pair_tight);
    std::vector<float4> box_min(d->n_boxes), box_max(d->n_boxes), light(d->n_lights);
    for (uint32_t i = 0; i < d->n_boxes; i++) {
        box_min[i] = make_float4(d->box_min[3 * i], d->box_min[3 * i + 1], d->box_min[3 * i + 2], 0.f);
        box_max[i] = make_float4(d->box_max[3 * i], d->box_max[3 * i + 1], d->box_max[3 * i + 2], 0.f);
    }
    for (uint32_t i = 0; i < d->n_lights; i++)
        light[i] = make_float4(d->light_pos[3 * i], d->light_pos[3 * i + 1], d->light_pos[3 * i + 2], d->light_intensity[i]);

    rt_scene* s = new rt_scene();
    if (hipGetDevice(&s->device) != hipSuccess) s->device = -1;
    s->d.n_spheres = d->n_spheres;
    s->d.n_triangles = d->n_triangles;
    s->d.n_boxes = d->n_boxes;
    s->d.n_lights = d->n_lights;
    s->d.n_materials = d->n_materials;
    s->d.n_clusters = n_clusters;
    s->d.cluster_pairs = cluster_pairs;
    s->d.n_groups = n_groups;
    s->d.group_clusters = group_clusters;
    s->d.fused_groups = fused ? 1u : 0u;
    s->n_super = n_super;
    s->super_groups = super_groups;
    if ((rc = upload(s, sphere, &s->d.sphere)) || (rc = upload(s, sphere_mat, &s->d.sphere_mat)) ||
        (rc = upload(s, mat_color, &s->d.mat_color)) || (rc = upload(s, mat_ior, &s->d.mat_ior)) ||
        (rc = upload(s, geom, &s->d.tri_geom)) || (rc = upload(s, pair, &s->d.tri_pair)) || (rc = upload(s, orig, &s->d.tri_orig)) ||
        (rc = upload(s, cl_pair, &s->d.cl_pair)) || (rc = upload(s, grp_pair, &s->d.grp_pair)) || (rc = upload(s, sup_pair, &s->sup_pair)) ||
        (rc = upload(s, cl_pair_tight, &s->cl_pair_tight)) || (rc = upload(s, grp_pair_tight, &s->grp_pair_tight)) ||
        (rc = upload(s, sup_pair_tight, &s->sup_pair_tight)) || (rc = upload(s, shade, &s->d.tri_shade)) ||
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
    if (s->sphere_rad) (void)hipFree(s->sphere_rad);
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

namespace {

// Rows per chunk and samples per batch of a band of `rows` rows, for batches of at most max_paths
// paths: a band of more pixels than that is rendered in chunks of rows, and a chunk in batches of samples.
struct BatchShape {
    uint32_t rows_per_chunk, per_batch;
};

BatchShape batch_shape(uint32_t width, uint32_t rows, uint32_t n_samples, uint32_t max_paths) {
    BatchShape b;
    b.rows_per_chunk = std::max(1u, std::min(rows, max_paths / width));
    b.per_batch = std::max(1u, max_paths / (b.rows_per_chunk * width));
    if (b.per_batch > n_samples) b.per_batch = n_samples;
    return b;
}

// Waits for the frame-end mark and turns the marks of the call into times: the whole call, and per kind of
// kernel ('B' binning starts, 'M' triangle kernel starts, 'S' k_shade starts, 'E' it ended).
int read_marks(rt_scene* s, rt_stats* stats) {
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
        } else if (s->event_tags[i] == 'S') {
            stats->shade_kernel_ms += k;
            stats->shade_kernel_launches++;
        } else {
            stats->mesh_kernel_ms += k;
            stats->mesh_kernel_launches++;
        }
    }
    return RT_OK;
}

// The body of rt_accumulate_rows on a given set of work buffers, which must hold batches of
// shape.rows_per_chunk * width * shape.per_batch paths with max_depth + 1 triangle stages.
// With stats == nullptr and no progress callback it only enqueues work and allocates nothing.
int accumulate_batches(rt_scene* s, const WorkSet& ws, BatchShape shape, const rt_camera* cam, const rt_options* opt,
                       uint32_t sample_begin, uint32_t sample_end, uint32_t row_begin, uint32_t row_end, float* d_sum,
                       hipStream_t stream, rt_stats* stats, bool report_progress) {
    int rc;
    const Work& w = ws.w;
    Frame f = make_frame(cam, opt);
    const uint32_t n_samples = sample_end - sample_begin;
    const uint32_t rows_per_chunk = shape.rows_per_chunk, per_batch = shape.per_batch;
    const uint32_t n_counters = (uint32_t)opt->max_depth + 1u;
    const uint64_t total_paths = (uint64_t)n_samples * (row_end - row_begin) * cam->width;

    // profile == 2 without statistics: the timing marks are recorded on the stream and read later (rt_profile_collect),
    // so that the call neither waits for the device nor keeps the host from queueing the next frame
    const bool deferred = !stats && opt->profile == 2;
    const bool profile = (stats && opt->profile) || deferred;
    f.count_casts = stats != nullptr;
    const bool cull = opt->cluster_culling != 0 && s->d.n_clusters > 1;
    f.prune_zero_terms = opt->cluster_culling != 0;
    // (rays on a cluster's list) x (triangles of the cluster), summed on the device: a large mesh has tens of thousands of
    // counters, a line each, per stage and tile
    const size_t cl_entries = cull && stats ? (size_t)n_counters * ws.tiles * s->d.n_clusters : 0;
    uint64_t tests = 0;
    if (stats || deferred) { // (a call without marks leaves those of an earlier call alone)
        s->event_tags.clear();
        s->marks_pending = false;
        if (!s->frame_begin) HIP_TRY(hipEventCreate(&s->frame_begin));
        if (!s->frame_end) HIP_TRY(hipEventCreate(&s->frame_end));
        HIP_TRY(hipEventRecord(s->frame_begin, stream));
    }
    std::vector<uint32_t> counts(stats ? (size_t)n_counters * kShards * kCountStride : 0);
    std::vector<uint32_t> bounce_counts(stats ? (size_t)kCastShards * kCountStride : 0);
    std::vector<uint32_t> live_counts(counts.size());
    uint64_t path_bounces = 0, path_records = 0;
    std::vector<uint32_t> cast_counts(stats ? (size_t)n_counters * kCastShards * kCountStride : 0);
    uint64_t mesh_rays = 0, paths = 0, sphere_casts = 0;
    uint32_t batches = 0;

    for (uint32_t chunk_row = row_begin; chunk_row < row_end; chunk_row += rows_per_chunk)
    for (uint32_t first = 0; first < n_samples; first += per_batch) {
        const uint32_t n_pixels = std::min(rows_per_chunk, row_end - chunk_row) * cam->width;
        f.n_pixels = n_pixels;
        f.pixel_offset = chunk_row * cam->width;
        const uint32_t in_batch = (n_samples - first < per_batch) ? n_samples - first : per_batch;
        f.n_paths = in_batch * n_pixels;
        f.sample_begin = sample_begin + first;
        const dim3 grid((f.n_paths + kBlock - 1) / kBlock), block(kBlock);
        const size_t cl_count_words = (size_t)(w.grp_count - w.cl_count);
        if (ws.n_counters == n_counters && s->d.fused_groups && !stats && cl_count_words >= (1u << 20)) {
            // k_group_clusters touches the per-cluster counters for the statistics only, and a mesh of tens of thousands
            // of clusters has hundreds of megabytes of them (a line each, per stage and tile): the blocks around them
            HIP_TRY(hipMemsetAsync(w.q_count, 0, (size_t)(w.cl_count - w.q_count) * sizeof(uint32_t), stream));
            HIP_TRY(hipMemsetAsync(w.grp_count, 0, (ws.counter_words - (size_t)(w.grp_count - w.q_count)) * sizeof(uint32_t), stream));
        } else if (ws.n_counters == n_counters) { // the counters are one block: one fill
            HIP_TRY(hipMemsetAsync(w.q_count, 0, ws.counter_words * sizeof(uint32_t), stream));
        } else { // a work set laid out for more bounces than this frame has: only the stages it runs
            const size_t q_bytes = (size_t)n_counters * kShards * kCountStride * sizeof(uint32_t);
            HIP_TRY(hipMemsetAsync(w.q_count, 0, q_bytes, stream));
            HIP_TRY(hipMemsetAsync(w.live_count, 0, q_bytes, stream));
            if (cull) {
                const size_t slots = (size_t)n_counters * ws.tiles * kCountStride * sizeof(uint32_t);
                HIP_TRY(hipMemsetAsync(w.cl_count, 0, slots * s->d.n_clusters, stream));
                if (s->d.n_groups) HIP_TRY(hipMemsetAsync(w.grp_count, 0, slots * s->d.n_groups, stream));
                if (s->n_super) HIP_TRY(hipMemsetAsync(ws.sup_count, 0, slots * s->n_super, stream));
            }
        }
        if (stats) {
            HIP_TRY(hipMemsetAsync(w.cast_count, 0, cast_counts.size() * sizeof(uint32_t), stream));
            HIP_TRY(hipMemsetAsync(w.bounce_count, 0, bounce_counts.size() * sizeof(uint32_t), stream));
        }
        // paths per lane: as many as still leave 2 048 workgroups (8 per CU)
        const bool fast = opt->arithmetic != 0;
        const bool fits = scene_fits_cache(s->d);
        // The bounce kernel of the fast arithmetic.  Large batches of frames with indirect light and object boxes: paths stay
        // in their lanes while they need nothing from the triangle stage (k_shade_fast).  Otherwise a launch per bounce, the first
        // of which computes the camera rays again instead of reading them back (a frame without bounces has none).
        // (RT_AMD_EXACT_STAYS: the control flow of the former with the exact arithmetic, for the tests against the oracle.)
        static const bool exact_stays = std::getenv("RT_AMD_EXACT_STAYS") != nullptr;
        // (without the object boxes every ray of a scene with triangles walks the mesh: no path could stay)
        const bool stays = (fast && opt->indirect != 0 && opt->bounding_box != 0 && f.n_paths >= kStaysFromPaths) || exact_stays;
        const int regenerate = fast && !stays && opt->max_depth >= 1;
        if (f.n_paths >= 2048u * kBlock * 4)
            hipLaunchKernelGGL((fast ? (fits ? k_generate<4, FastMath, true> : k_generate<4, FastMath, false>) : (fits ? k_generate<4, ExactMath, true> : k_generate<4, ExactMath, false>)), dim3((f.n_paths + kBlock * 4 - 1) / (kBlock * 4)), block, 0, stream, s->d, w, f, regenerate);
        else if (f.n_paths >= 2048u * kBlock * 2)
            hipLaunchKernelGGL((fast ? (fits ? k_generate<2, FastMath, true> : k_generate<2, FastMath, false>) : (fits ? k_generate<2, ExactMath, true> : k_generate<2, ExactMath, false>)), dim3((f.n_paths + kBlock * 2 - 1) / (kBlock * 2)), block, 0, stream, s->d, w, f, regenerate);
        else
            hipLaunchKernelGGL((fast ? (fits ? k_generate<1, FastMath, true> : k_generate<1, FastMath, false>) : (fits ? k_generate<1, ExactMath, true> : k_generate<1, ExactMath, false>)), grid, block, 0, stream, s->d, w, f, regenerate);
        HIP_TRY(hipGetLastError());
        for (int depth = 0; depth < opt->max_depth; depth++) {
            if ((rc = launch_mesh(s, ws, (uint32_t)depth, stream, profile, cull, f.n_paths, stats != nullptr, fast, depth == 0 ? f.n_paths : 0)) != RT_OK) return rc;
            if (profile && (rc = mark_event(s, stream, 'S')) != RT_OK) return rc;
            if (depth == 0 && regenerate) {
                hipLaunchKernelGGL((fits ? k_shade_fast_first<true> : k_shade_fast_first<false>), grid, block, 0, stream, s->d, w, f);
            } else {
                // (a wavefront of the kernels whose paths stay in their lanes takes stretch_of() paths; how many the lists
                // of the later launches hold is not known here)
                const uint32_t per_group = kWavesPerBlock * stretch_of(f.n_paths, f.min_waves);
                const dim3 first_grid = stays ? dim3((f.n_paths + per_group - 1) / per_group) : grid;
                auto bounce = exact_stays && !fast ? (fits ? k_shade_stays<true> : k_shade_stays<false>)
                              : !fast              ? (fits ? k_shade<true> : k_shade<false>)
                              : !stays             ? (fits ? k_shade_fast_each<true> : k_shade_fast_each<false>)
                                                   : (fits ? k_shade_fast<true> : k_shade_fast<false>);
                hipLaunchKernelGGL(bounce,
                                   depth == 0 ? first_grid : dim3(grid.x + kShards), block, 0, stream, s->d, w, f, depth);
            }
            if (profile && (rc = mark_event(s, stream, 'E')) != RT_OK) return rc;
            HIP_TRY(hipGetLastError());
        }
        if ((rc = launch_mesh(s, ws, (uint32_t)opt->max_depth, stream, profile, cull, f.n_paths, stats != nullptr, opt->arithmetic != 0)) != RT_OK) return rc;
        hipLaunchKernelGGL(k_accumulate, dim3((n_pixels + kBlock - 1) / kBlock), block, 0, stream, s->d, w, f,
                           reinterpret_cast<float4*>(d_sum));
        HIP_TRY(hipGetLastError());
        batches++;
        paths += f.n_paths;
        if (report_progress && s->progress) {
            HIP_TRY(hipStreamSynchronize(stream));
            s->progress(paths, total_paths, s->progress_user);
        }
        if (stats) { // the counters are reused by the next batch: read them now
            HIP_TRY(hipMemcpyAsync(counts.data(), w.q_count, counts.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipMemcpyAsync(live_counts.data(), w.live_count, live_counts.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipMemcpyAsync(bounce_counts.data(), w.bounce_count, bounce_counts.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipMemcpyAsync(cast_counts.data(), w.cast_count, cast_counts.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            unsigned long long batch_tests = 0;
            if (cull) {
                HIP_TRY(hipMemsetAsync(ws.tests_sum, 0, sizeof(unsigned long long), stream));
                hipLaunchKernelGGL(k_count_tests, dim3((unsigned)std::min<size_t>((cl_entries + kBlock - 1) / kBlock, 1024)), block, 0, stream,
                                   (const uint32_t*)w.cl_count, cl_entries, s->d.n_clusters, 2 * s->d.cluster_pairs, s->d.n_triangles, ws.tests_sum);
                HIP_TRY(hipMemcpyAsync(&batch_tests, ws.tests_sum, sizeof batch_tests, hipMemcpyDeviceToHost, stream));
            }
            HIP_TRY(hipStreamSynchronize(stream));
            uint64_t batch_rays = 0;
            for (size_t i = 0; i < counts.size(); i += kCountStride) batch_rays += counts[i];
            for (size_t i = 0; i < cast_counts.size(); i += kCountStride) sphere_casts += cast_counts[i];
            mesh_rays += batch_rays;
            for (size_t i = 0; i < bounce_counts.size(); i += kCountStride) path_bounces += bounce_counts[i];
            // the first launch fetches every path of the batch, launch d > 0 the paths on its list
            if (opt->max_depth > 0) path_records += f.n_paths;
            for (int d = 1; d < opt->max_depth; d++)
                for (uint32_t shard = 0; shard < kShards; shard++) path_records += live_counts[((size_t)d * kShards + shard) * kCountStride];
            if (cull) {
                tests += batch_tests;
            } else {
                tests += batch_rays * s->d.n_triangles;
            }
        }
    }
    if (deferred) {
        HIP_TRY(hipEventRecord(s->frame_end, stream));
        s->marks_pending = true;
    }
    if (stats) {
        HIP_TRY(hipEventRecord(s->frame_end, stream));
        if ((rc = read_marks(s, stats)) != RT_OK) return rc;
        stats->batches = batches;
        stats->paths = paths;
        stats->mesh_rays = mesh_rays;
        stats->path_bounces = path_bounces;
        stats->path_records = path_records;
        stats->triangle_tests = tests;
        stats->sphere_casts = sphere_casts;
    }
    return RT_OK;
}

// rt_accumulate_rows for a scene without triangles: k_trace_spheres runs the sample and bounce
// loops in registers, so there is no path state and no queue.  A band of many pixels is rendered
// by one launch, a lane per pixel.  A band of few pixels would not fill the chip that way: there a
// lane takes four samples of a pixel and the per-sample colours go through `rad` (16 bytes per
// path, batches of at most 2^24 paths) to k_accumulate, which adds them in sample order -- the
// same sums, bit for bit.  rad == nullptr selects the first arrangement whatever the size.
// `cast_count` (kCastShards counters) is only touched when stats are wanted.
constexpr uint32_t kDirectPixels = 1u << 20; // from this many pixels on, a lane per pixel fills the chip

int accumulate_spheres(rt_scene* s, const rt_camera* cam, const rt_options* opt, uint32_t sample_begin, uint32_t sample_end,
                       uint32_t row_begin, uint32_t row_end, float* d_sum, hipStream_t stream, rt_stats* stats,
                       float4* rad, uint32_t* cast_count, bool report_progress) {
    Frame f = make_frame(cam, opt);
    f.count_casts = stats != nullptr;
    f.prune_zero_terms = opt->cluster_culling != 0;
    const uint32_t n_samples = sample_end - sample_begin;
    const uint64_t total_paths = (uint64_t)n_samples * (row_end - row_begin) * cam->width;
    const bool fast = opt->arithmetic != 0;
    const bool deferred = !stats && opt->profile == 2; // marks for rt_profile_collect: here only the whole call is timed
    if (stats || deferred) {
        s->event_tags.clear();
        s->marks_pending = false;
        if (!s->frame_begin) HIP_TRY(hipEventCreate(&s->frame_begin));
        if (!s->frame_end) HIP_TRY(hipEventCreate(&s->frame_end));
        if (stats) HIP_TRY(hipMemsetAsync(cast_count, 0, (size_t)kCastShards * kCountStride * sizeof(uint32_t), stream));
        HIP_TRY(hipEventRecord(s->frame_begin, stream));
    }
    uint64_t paths = 0;
    uint32_t launches = 0;
    auto after_launch = [&](uint64_t n_paths) -> int {
        HIP_TRY(hipGetLastError());
        launches++;
        paths += n_paths;
        if (report_progress && s->progress) {
            HIP_TRY(hipStreamSynchronize(stream));
            s->progress(paths, total_paths, s->progress_user);
        }
        return RT_OK;
    };
    int rc;
    const bool fits = scene_fits_cache(s->d);
    auto trace = fast ? (fits ? k_trace_spheres<FastMath, true> : k_trace_spheres<FastMath, false>)
                      : (fits ? k_trace_spheres<ExactMath, true> : k_trace_spheres<ExactMath, false>);
    if (!rad) {
        f.n_pixels = (row_end - row_begin) * cam->width;
        f.pixel_offset = row_begin * cam->width;
        // one launch for all samples; with a progress callback, launches of about 2^24 paths
        uint32_t per_launch = n_samples;
        if (report_progress && s->progress) per_launch = std::max(1u, std::min(n_samples, kMaxPathsPerBatch / std::max(f.n_pixels, 1u)));
        for (uint32_t first = 0; first < n_samples; first += per_launch) {
            const uint32_t n = std::min(per_launch, n_samples - first);
            f.sample_begin = sample_begin + first;
            hipLaunchKernelGGL(trace, dim3((f.n_pixels + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, s->d, f, n,
                               reinterpret_cast<float4*>(d_sum), (float4*)nullptr, cast_count);
            if ((rc = after_launch((uint64_t)n * f.n_pixels)) != RT_OK) return rc;
        }
    } else {
        const BatchShape shape = batch_shape(cam->width, row_end - row_begin, n_samples, kMaxPathsPerBatch);
        Work w{};
        w.rad = rad;
        for (uint32_t chunk_row = row_begin; chunk_row < row_end; chunk_row += shape.rows_per_chunk)
        for (uint32_t first = 0; first < n_samples; first += shape.per_batch) {
            f.n_pixels = std::min(shape.rows_per_chunk, row_end - chunk_row) * cam->width;
            f.pixel_offset = chunk_row * cam->width;
            const uint32_t n = std::min(shape.per_batch, n_samples - first);
            f.sample_begin = sample_begin + first;
            f.n_paths = n * f.n_pixels;
            const uint32_t lanes = f.n_pixels * ((n + kTraceGroup - 1) / kTraceGroup);
            hipLaunchKernelGGL(trace, dim3((lanes + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, s->d, f, n,
                               reinterpret_cast<float4*>(d_sum), rad, cast_count);
            hipLaunchKernelGGL(k_accumulate, dim3((f.n_pixels + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, s->d, w, f,
                               reinterpret_cast<float4*>(d_sum));
            if ((rc = after_launch(f.n_paths)) != RT_OK) return rc;
        }
    }
    if (deferred) {
        HIP_TRY(hipEventRecord(s->frame_end, stream));
        s->marks_pending = true;
    }
    if (stats) {
        HIP_TRY(hipEventRecord(s->frame_end, stream));
        std::vector<uint32_t> counts((size_t)kCastShards * kCountStride);
        HIP_TRY(hipMemcpyAsync(counts.data(), cast_count, counts.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipEventSynchronize(s->frame_end));
        HIP_TRY(hipStreamSynchronize(stream));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, s->frame_begin, s->frame_end));
        stats->gpu_ms = ms;
        stats->batches = launches;
        stats->paths = paths;
        for (size_t i = 0; i < counts.size(); i += kCountStride) stats->sphere_casts += counts[i];
    }
    return RT_OK;
}

// The per-sample colour buffer of the second arrangement, for batches of up to n_paths paths.
int ensure_sphere_rad(rt_scene* s, size_t n_paths) {
    if (s->sphere_rad_paths >= n_paths) return RT_OK;
    if (s->sphere_rad) (void)hipFree(s->sphere_rad);
    s->sphere_rad = nullptr;
    s->sphere_rad_paths = 0;
    if (hipMalloc(reinterpret_cast<void**>(&s->sphere_rad), n_paths * sizeof(float4)) != hipSuccess) {
        (void)hipGetLastError();
        rt_set_error("out of device memory: %zu bytes of per-sample colours", n_paths * sizeof(float4));
        return RT_ERR_NOMEM;
    }
    s->sphere_rad_paths = n_paths;
    return RT_OK;
}

int check_accumulate(const rt_scene* s, const rt_camera* cam, const rt_options* opt, uint32_t sample_begin, uint32_t sample_end,
                     uint32_t row_begin, uint32_t row_end, const float* d_sum) {
    int rc = check_frame(s, cam, opt);
    if (rc != RT_OK) return rc;
    if (!d_sum || sample_end < sample_begin) { rt_set_error("rt_accumulate: bad sample range or null buffer"); return RT_ERR_INVALID; }
    if (row_end < row_begin || row_end > cam->height) { rt_set_error("rt_accumulate_rows: rows [%u, %u) of %u", row_begin, row_end, cam->height); return RT_ERR_INVALID; }
    return RT_OK;
}

} // namespace

extern "C" int rt_accumulate_rows(rt_scene* s, const rt_camera* cam, const rt_options* opt, uint32_t sample_begin,
                                  uint32_t sample_end, uint32_t row_begin, uint32_t row_end, float* d_sum, void* stream_,
                                  rt_stats* stats) {
    int rc = check_accumulate(s, cam, opt, sample_begin, sample_end, row_begin, row_end, d_sum);
    if (rc != RT_OK) return rc;
    if (stats) std::memset(stats, 0, sizeof *stats);
    if (row_end == row_begin || sample_end == sample_begin) return RT_OK;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (s->d.n_triangles == 0) { // no mesh: no path state and no queue
        if (stats && (rc = ensure_work(s, 1, 1)) != RT_OK) return rc; // the statistics counters
        const uint32_t band_pixels = (row_end - row_begin) * cam->width;
        // without indirect light and anti-aliasing most pixels stop after one sample, which only the lane-per-pixel loop exploits
        const bool direct = band_pixels >= kDirectPixels || (!opt->indirect && !opt->antialiasing) || sample_end - sample_begin == 1;
        if (!direct) {
            const BatchShape shape = batch_shape(cam->width, row_end - row_begin, sample_end - sample_begin, kMaxPathsPerBatch);
            if ((rc = ensure_sphere_rad(s, (size_t)shape.per_batch * shape.rows_per_chunk * cam->width)) != RT_OK) return rc;
        }
        return accumulate_spheres(s, cam, opt, sample_begin, sample_end, row_begin, row_end, d_sum, stream, stats,
                                  direct ? nullptr : s->sphere_rad, stats ? s->ws.w.cast_count : nullptr, true);
    }
    // A batch is at most 2^24 paths.
    const BatchShape shape = batch_shape(cam->width, row_end - row_begin, sample_end - sample_begin, kMaxPathsPerBatch);
    if ((rc = ensure_work(s, shape.per_batch * shape.rows_per_chunk * cam->width, (uint32_t)opt->max_depth + 1u)) != RT_OK) return rc;
    return accumulate_batches(s, s->ws, shape, cam, opt, sample_begin, sample_end, row_begin, row_end, d_sum, stream, stats, true);
}

extern "C" int rt_accumulate_static(rt_scene* s, const rt_camera* cam, const rt_options* opt, uint32_t sample_begin,
                                    uint32_t sample_end, float* d_sum, void* stream_) {
    int rc = check_accumulate(s, cam, opt, sample_begin, sample_end, 0, cam ? cam->height : 0, d_sum);
    if (rc != RT_OK) return rc;
    if (cam->height == 0 || sample_end == sample_begin) return RT_OK;
    // Timing marks (rt_options.profile) need events, and creating an event allocates: this call takes none.
    rt_options quiet = *opt;
    quiet.profile = 0;
    opt = &quiet;
    if (s->d.n_triangles == 0) // no mesh: the paths live in registers and need no work area at all
        return accumulate_spheres(s, cam, opt, sample_begin, sample_end, 0, cam->height, d_sum, static_cast<hipStream_t>(stream_),
                                  nullptr, nullptr, nullptr, false);
    int device = -1;
    HIP_TRY(hipGetDevice(&device));
    if (device != s->device) { // the area is the current device's copy; the scene's arrays live where it was uploaded
        rt_set_error("rt_accumulate_static: the scene lives on device %d, the current device is %d", s->device, device);
        return RT_ERR_INVALID;
    }
    unsigned char* area = nullptr;
    if ((rc = static_work_area(&area)) != RT_OK) return rc;
    // The largest batch whose work buffers fit the area (at most half of it for cluster lists): the
    // frame is rendered in as many batches of rows and samples as that takes.
    const uint32_t n_counters = (uint32_t)opt->max_depth + 1u;
    uint32_t max_paths = kMaxPathsPerBatch;
    WorkSet ws;
    BatchShape shape;
    for (;;) {
        shape = batch_shape(cam->width, cam->height, sample_end - sample_begin, max_paths);
        const uint64_t n = (uint64_t)shape.per_batch * shape.rows_per_chunk * cam->width;
        if (n <= kMaxPathsPerBatch && layout_work(s->d, s->n_super, (uint32_t)n, n_counters, kStaticWorkBytes / 2, nullptr, &ws) == RT_OK &&
            ws.bytes <= kStaticWorkBytes)
            break;
        if (n <= cam->width) {
            rt_set_error("rt_accumulate_static: one image row of %u pixels with this scene (%u lights, %u clusters) needs more "
                         "than the static work area of %zu bytes", cam->width, s->d.n_lights, s->d.n_clusters, kStaticWorkBytes);
            return RT_ERR_INVALID;
        }
        max_paths = (uint32_t)std::max<uint64_t>(n / 2, cam->width);
    }
    if ((rc = layout_work(s->d, s->n_super, ws.paths, n_counters, kStaticWorkBytes / 2, area, &ws)) != RT_OK) return rc;
    return accumulate_batches(s, ws, shape, cam, opt, sample_begin, sample_end, 0, cam->height, d_sum,
                              static_cast<hipStream_t>(stream_), nullptr, false);
}

extern "C" int rt_resolve(const float* d_sum, uint32_t n_pixels, float* d_out_rgb, void* stream_) {
    if (!d_sum || !d_out_rgb || n_pixels == 0) { rt_set_error("rt_resolve: bad argument"); return RT_ERR_INVALID; }
    hipLaunchKernelGGL(k_resolve, dim3((n_pixels + kBlock - 1) / kBlock), dim3(kBlock), 0, static_cast<hipStream_t>(stream_),
                       reinterpret_cast<const float4*>(d_sum), n_pixels, d_out_rgb);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

extern "C" int rt_add_sums(float* d_dst, const float* d_src, uint32_t n_floats, void* stream_) {
    if (!d_dst || !d_src) { rt_set_error("rt_add_sums: null buffer"); return RT_ERR_INVALID; }
    if (n_floats == 0) return RT_OK;
    const uint32_t n_vec4 = n_floats / 4, n_tail = n_floats % 4;
    hipLaunchKernelGGL(k_add_sums, dim3((std::max(n_vec4, n_tail) + kBlock - 1) / kBlock), dim3(kBlock), 0, static_cast<hipStream_t>(stream_),
                       reinterpret_cast<float4*>(d_dst), reinterpret_cast<const float4*>(d_src), n_vec4, d_dst + 4 * (size_t)n_vec4,
                       d_src + 4 * (size_t)n_vec4, n_tail);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

extern "C" int rt_profile_collect(rt_scene* s, rt_stats* stats) {
    if (!s || !stats) { rt_set_error("rt_profile_collect: null argument"); return RT_ERR_INVALID; }
    if (!s->marks_pending) { rt_set_error("rt_profile_collect: no call with rt_options.profile = 2 left marks on this handle"); return RT_ERR_INVALID; }
    std::memset(stats, 0, sizeof *stats);
    s->marks_pending = false;
    return read_marks(s, stats);
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
    const Work& w = s->ws.w;
    rt_camera cam;
    std::memset(&cam, 0, sizeof cam);
    cam.width = n;
    cam.height = 1;
    Frame f = make_frame(&cam, opt);
    f.n_paths = n;
    // one block for the seven arrays of the call: a (3n) | b (3n) | abg (3n) | t (n) | sphere (n) | triangle (n) | visible (n)
    float* block_mem = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&block_mem), 13 * (size_t)n * 4) != hipSuccess) {
        rt_set_error("probe: hipMalloc failed: %s", hipGetErrorString(hipGetLastError()));
        return RT_ERR_NOMEM;
    }
    float *d_a = block_mem, *d_b = d_a + 3 * (size_t)n, *d_abg = d_b + 3 * (size_t)n, *d_t = d_abg + 3 * (size_t)n;
    int32_t *d_s = reinterpret_cast<int32_t*>(d_t + n), *d_tr = d_s + n, *d_v = d_tr + n;
    auto release = [&]() { (void)hipFree(block_mem); };
    const dim3 grid((n + kBlock - 1) / kBlock), block(kBlock);
    rc = RT_OK;
    do {
        if (hipMemcpy(d_a, a, 3 * (size_t)n * 4, hipMemcpyHostToDevice) || hipMemcpy(d_b, b, 3 * (size_t)n * 4, hipMemcpyHostToDevice) ||
            hipMemset(d_abg, 0, 3 * (size_t)n * 4) || hipMemset(w.q_count, 0, 2 * kShards * kCountStride * sizeof(uint32_t))) { rc = RT_ERR_DEVICE; break; }
        hipLaunchKernelGGL(k_probe_rays, grid, block, 0, nullptr, s->d, w, f, (const float*)d_a, (const float*)d_b, as_segments);
        const bool cull = opt->cluster_culling != 0 && s->d.n_clusters > 1;
        if (cull && hipMemset(w.cl_count, 0, (size_t)2 * s->ws.tiles * s->d.n_clusters * kCountStride * sizeof(uint32_t))) { rc = RT_ERR_DEVICE; break; }
        if (cull && s->d.n_groups && hipMemset(w.grp_count, 0, (size_t)2 * s->ws.tiles * s->d.n_groups * kCountStride * sizeof(uint32_t))) { rc = RT_ERR_DEVICE; break; }
        if (cull && s->n_super && hipMemset(s->ws.sup_count, 0, (size_t)2 * s->ws.tiles * s->n_super * kCountStride * sizeof(uint32_t))) { rc = RT_ERR_DEVICE; break; }
        if (launch_mesh(s, s->ws, 0, nullptr, false, cull, n, false, opt->arithmetic != 0) != RT_OK) { rc = RT_ERR_DEVICE; break; }
        hipLaunchKernelGGL(k_probe_results, grid, block, 0, nullptr, s->d, w, f, as_segments, d_t, d_s, d_tr, d_abg, d_v);
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

extern "C" int rt_random_cos(uint32_t n, const float* normals, const float* r1, const float* r2, float* out_dirs) {
    if (n && (!normals || !r1 || !r2 || !out_dirs)) { rt_set_error("rt_random_cos: null argument"); return RT_ERR_INVALID; }
    if (n == 0) return RT_OK;
    float* d = nullptr; // normals (3n) | r1 (n) | r2 (n) | out (3n)
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d), (size_t)n * 8 * sizeof(float)));
    int rc = RT_OK;
    if (hipMemcpy(d, normals, (size_t)n * 12, hipMemcpyHostToDevice) || hipMemcpy(d + 3 * (size_t)n, r1, (size_t)n * 4, hipMemcpyHostToDevice) ||
        hipMemcpy(d + 4 * (size_t)n, r2, (size_t)n * 4, hipMemcpyHostToDevice)) {
        rc = RT_ERR_DEVICE;
    } else {
        hipLaunchKernelGGL(k_probe_random_cos, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, nullptr, n, (const float*)d,
                           (const float*)(d + 3 * (size_t)n), (const float*)(d + 4 * (size_t)n), d + 5 * (size_t)n);
        if (hipGetLastError() != hipSuccess || hipMemcpy(out_dirs, d + 5 * (size_t)n, (size_t)n * 12, hipMemcpyDeviceToHost)) rc = RT_ERR_DEVICE;
    }
    if (rc != RT_OK) rt_set_error("rt_random_cos: a HIP call failed: %s", hipGetErrorString(hipGetLastError()));
    (void)hipFree(d);
    return rc;
}

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
