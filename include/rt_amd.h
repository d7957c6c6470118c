/*
 * rt_amd.h -- C ABI of the MI355X (gfx950) path tracer.
 *
 * The library is a drop-in for the render path of wak31415/raytracer:
 * everything the reference does between Scene::render() and the returned
 * image (reference src/scene.cu:209-222 -> src/raytrace.cu:509-644) sits
 * behind these entry points.  All signatures are plain C: pointers, sizes
 * and POD structs.  Every function returns 0 on success or a negative
 * RT_ERR_* code; rt_last_error() gives the message of the calling thread's
 * most recent failure.
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_OK 0
#define RT_ERR_INVALID (-1)  /* bad argument / inconsistent description   */
#define RT_ERR_DEVICE (-2)   /* a HIP call failed (message has the detail) */
#define RT_ERR_IO (-3)       /* file missing or unreadable                 */
#define RT_ERR_PARSE (-4)    /* malformed scene / mesh file                */
#define RT_ERR_NOMEM (-5)

/* material_type of the reference (include/scene_objects.cuh:4-8) */
enum { RT_DIFFUSE = 0, RT_MIRROR = 1, RT_GLASS = 2 };

/* Host-side scene description: the reference's raytrace_spheres() argument
 * list (src/raytrace.cu:509-522) as flat arrays.  The Material copy the
 * reference keeps in every Sphere and Triangle (include/scene_objects.cuh:
 * 24-43) is an index into one material table here.  All pointers are host
 * pointers owned by the caller; rt_scene_upload copies what it needs. */
typedef struct rt_scene_desc {
    uint32_t n_spheres;
    const float* sphere_pos;        /* 3 * n_spheres */
    const float* sphere_radius;     /* n_spheres */
    const int32_t* sphere_material; /* n_spheres, index into materials */
    uint32_t n_materials;
    const int32_t* material_type;   /* RT_DIFFUSE / RT_MIRROR / RT_GLASS */
    const float* material_color;    /* 3 * n_materials */
    const float* material_ro;       /* refractive index outside */
    const float* material_ri;       /* refractive index inside  */
    uint32_t n_vertices;
    const float* vertices;          /* 3 * n_vertices */
    const float* normals;           /* 3 * n_vertices */
    uint32_t n_triangles;
    const int32_t* tri_v;           /* 3 * n_triangles, into vertices */
    const int32_t* tri_n;           /* 3 * n_triangles, into normals  */
    const int32_t* tri_material;    /* n_triangles */
    uint32_t n_boxes;               /* the reference's obj_count */
    const float* box_min;           /* 3 * n_boxes */
    const float* box_max;           /* 3 * n_boxes */
    uint32_t n_lights;
    const float* light_pos;         /* 3 * n_lights */
    const float* light_intensity;   /* n_lights */
} rt_scene_desc;

/* What the reference passes to its kernel from struct Camera
 * (src/raytrace.cu:538-539,590-595): the rotation block and translation
 * column of the extrinsic matrix E, the four intrinsics the kernel reads
 * (K[0][0], K[1][1], K[0][2], K[1][2]; include/projection_helpers.cuh:69-75),
 * the resolution and the samples per pixel. */
typedef struct rt_camera {
    float rot[9]; /* row-major */
    float pos[3];
    float fx, fy, cx, cy;
    uint32_t width, height, num_rays;
} rt_camera;

/* The reference's compile-time switches INDIRECT_LIGHTING / ANTIALIASING /
 * BOUNDING_BOX and MAX_RAY_DEPTH (src/raytrace.cu:12-16) as run-time
 * options.  seed replaces time(0)+seed (src/raytrace.cu:574): the same seed
 * always gives the same image. */
typedef struct rt_options {
    int32_t indirect;
    int32_t antialiasing;
    int32_t bounding_box;
    int32_t max_depth; /* 5 in the reference */
    uint32_t seed;
    int32_t profile;   /* 1: kernel timings in the call's rt_stats (the call then waits for its work);
                          2, with stats == NULL: timing marks are recorded on the stream and nothing
                          is read -- rt_profile_collect fetches them later */
    /* 0: every queued ray is tested against every triangle, the reference's
     * scan (src/raytrace.cu:148-175), and every path runs all its bounces.
     * Non-zero: triangles are grouped into clusters of 16 or 32 neighbours and a ray
     * skips the clusters whose padded bounds it misses; work that can only add
     * exact zeros is dropped (the mesh walk of a shadow ray whose contribution
     * is zero, the remaining bounces of a path whose weight is zero in all
     * three channels).  Same image unless the reference's rounding accepts a
     * triangle for a ray that misses it by more than the padding, or a path of
     * zero weight would have met a light at distance exactly zero (the
     * reference then adds 0 x inf = NaN). */
    int32_t cluster_culling;
    /* 0 (the default): the reference's float arithmetic operation by operation -- separately rounded
     * products and sums, IEEE square roots and quotients, the double-precision scalars of
     * src/raytrace.cu:358,360 -- with fixed polynomial log / sin / cos; the image equals a C restatement
     * of the reference compiled without contraction, bit for bit, on any device and batch layout.
     * 1: the same expressions as the hardware computes them fastest -- fused multiply-adds, reciprocal,
     * reciprocal-square-root and square-root instructions (one unit in the last place), hardware
     * log / sin / cos in the samplers -- which is also how the reference's own build computes (nvcc
     * contracts by default and the kernel calls __logf / __cosf / __sinf / __powf, src/raytrace.cu:37-41,
     * 46-49).  Per-pixel values without random numbers agree with mode 0 to about 1e-5 relative except
     * where a hit-or-miss decision flips at an edge; images with random numbers agree in the mean. */
    int32_t arithmetic;
} rt_options;

/* Work executed by the most recent render / accumulate call. */
typedef struct rt_stats {
    double gpu_ms;             /* first kernel start -> last kernel end        */
    double mesh_kernel_ms;     /* triangle kernel, summed over launches (profile != 0 only) */
    double bin_kernel_ms;      /* cluster binning + planning kernels (profile != 0 only)   */
    uint32_t mesh_kernel_launches;
    uint32_t batches;
    uint64_t paths;
    uint64_t mesh_rays;        /* rays that walked the triangle list           */
    uint64_t triangle_tests;   /* ray-triangle tests the triangle kernel executed */
    uint64_t sphere_casts;     /* rays tested against every sphere: camera, bounce and shadow rays */
    double shade_kernel_ms;    /* the bounce kernel (hit selection, materials, lights, sphere scans), summed
                                  over launches (profile != 0 only) */
    uint32_t shade_kernel_launches;
    uint32_t reserved;
    uint64_t path_bounces;     /* (path, bounce) pairs the bounce kernel processed: the paths alive on entry to a bounce
                                  (0 for a scene without triangles, whose frame is one kernel with the paths in registers) */
    uint64_t path_records;     /* of those, the pairs whose path came from memory: every path at its first bounce and every
                                  path a bounce wrote out for the triangle stage -- the unit the kernel's memory traffic
                                  scales with.  Equal to path_bounces with the exact arithmetic; with the fast arithmetic a
                                  path that needs nothing from the triangle stage stays in its lane for the next bounce. */
} rt_stats;

/* Device-resident scene plus the work buffers of the frames rendered from it.
 * One handle serves one frame at a time (calls on a handle must not overlap,
 * from host threads or on different streams); independent handles -- also of
 * the same scene -- can render concurrently on different streams. */
typedef struct rt_scene rt_scene;

/* Selects the HIP device this thread renders on (hipSetDevice). */
int rt_init(int32_t device);
const char* rt_last_error(void);

/* Replaces the per-call cudaMalloc + cudaMemcpy block of the reference
 * (src/raytrace.cu:541-555): builds the structure-of-arrays device copy
 * once, so several frames can be rendered from it. */
int rt_scene_upload(const rt_scene_desc* desc, rt_scene** out_scene);
int rt_scene_free(rt_scene* scene);

/* Replaces raytrace_spheres() (src/raytrace.cu:509-644) for one device:
 * renders cam->num_rays samples per pixel and writes the reference's output
 * value per pixel, gamma((1/actual_rays) * colour) on the 0..255+ scale,
 * unclamped, to out_rgb (host, 3 * width * height floats, row 0 on top). */
int rt_render(rt_scene* scene, const rt_camera* cam, const rt_options* opt,
              float* out_rgb, rt_stats* stats);

/* Multi-device building blocks (no counterpart in the single-GPU
 * reference; they split its sample loop, src/raytrace.cu:478-500).
 * rt_accumulate renders the GLOBAL sample indices [sample_begin, sample_end)
 * of every pixel and ADDS the linear colour sums and the sample counts to
 * d_sum (DEVICE memory, 4 * width * height floats: r, g, b, actual_rays),
 * in sample order.  rt_resolve turns such a buffer -- after the per-device
 * buffers were summed -- into the output image on the device.  `stream` is
 * a hipStream_t (NULL = default stream); both calls only enqueue work.
 * A NEGATIVE sample count marks a pixel whose sample loop the reference ends
 * after the sample just added (terminate_early, src/raytrace.cu:340-346,
 * 496-498: direct light without anti-aliasing, diffuse first hit): later
 * calls on the same buffer leave the pixel alone, rt_resolve divides by the
 * magnitude. */
int rt_accumulate(rt_scene* scene, const rt_camera* cam, const rt_options* opt,
                  uint32_t sample_begin, uint32_t sample_end, float* d_sum,
                  void* stream, rt_stats* stats);
int rt_resolve(const float* d_sum, uint32_t n_pixels, float* d_out_rgb, void* stream);
/* The same for the image rows [row_begin, row_end) only (d_sum still covers
 * the whole image; the other rows are left alone): lets more devices than
 * there are samples per pixel share a frame as row bands x sample ranges. */
int rt_accumulate_rows(rt_scene* scene, const rt_camera* cam, const rt_options* opt,
                       uint32_t sample_begin, uint32_t sample_end, uint32_t row_begin,
                       uint32_t row_end, float* d_sum, void* stream, rt_stats* stats);

/* The kernel timings of the last rt_accumulate / rt_accumulate_rows call made
 * on this handle with rt_options.profile = 2 and stats == NULL (the reference
 * times its one kernel with events around the launch, src/raytrace.cu:572-633,
 * and waits for it; a call of that kind only records its marks on the stream,
 * so the host goes on queueing work): waits for that call's work and fills
 * gpu_ms and the *_kernel_ms / *_launches fields, the rest is zero (a scene
 * without triangles is traced by one kernel: only gpu_ms is filled).
 * RT_ERR_INVALID when no such call is outstanding. */
int rt_profile_collect(rt_scene* scene, rt_stats* stats);

/* d_dst[i] += d_src[i] for n_floats floats of DEVICE memory on the current
 * device (16-byte aligned buffers): adds the sums another device produced,
 * after they were copied here, before rt_resolve. */
int rt_add_sums(float* d_dst, const float* d_src, uint32_t n_floats, void* stream);

/* The share of one frame that device `rank` of `world` renders:
 * out = {sample_begin, sample_end, row_begin, row_end}.  Samples are split
 * first (every device sees all pixels: perfectly balanced); only with more
 * devices than samples per pixel is the image also cut into bands of rows
 * (sample groups x bands, groups = the largest divisor of world <= spp). */
void rt_frame_partition(uint32_t rank, uint32_t world, uint32_t spp, uint32_t height, uint32_t out[4]);

/* One frame over several GPUs of one node from ONE process (the reference
 * renders on one GPU; this splits its sample loop, src/raytrace.cu:478-500).
 * rt_device_group_create uploads the scene to every listed device (HIP device
 * indices; the first one collects the result).  rt_device_group_render has
 * every device accumulate its rt_frame_partition share (one host thread per
 * device), sums the linear buffers on the first device -- ncclReduce over
 * RCCL when librccl can be loaded at run time, otherwise (or with
 * RT_AMD_EXCHANGE=peer in the environment, or when a device is listed twice)
 * peer copies followed by additions in device order -- resolves there and
 * copies the image to out_rgb (host, 3 * width * height floats).  stats, if
 * not NULL, are the first device's.  The image is the one rt_render draws up
 * to the order of the fp32 additions across devices. */
typedef struct rt_device_group rt_device_group;
int rt_device_group_create(const rt_scene_desc* desc, const int32_t* devices, uint32_t n_devices,
                           rt_device_group** out_group);
int rt_device_group_render(rt_device_group* group, const rt_camera* cam, const rt_options* opt,
                           float* out_rgb, rt_stats* stats);
int rt_device_group_free(rt_device_group* group);

/* rt_accumulate for callers that cannot allocate device memory (code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc).  It
 * has no counterpart in the reference.  The work buffers rt_accumulate takes
 * from hipMalloc come from a 64 MiB device array of the library instead (every
 * GPU the library runs on holds its copy until the process ends), so the frame
 * is rendered in more, smaller batches -- a few ten thousand paths each; the
 * sums are the ones rt_accumulate produces, bit for bit.  The call issues no
 * hipMalloc / hipFree and only enqueues work on `stream`; it takes no
 * statistics, reports no progress and ignores rt_options.profile (timing marks
 * need events, whose creation allocates).  RT_ERR_INVALID when the current
 * device is not the one the scene was uploaded to.  RT_ERR_INVALID when a single image row
 * of the frame does not fit the array (several hundred thousand pixels wide,
 * or a scene with very many lights or clusters).  Call rt_init on the device
 * first (it resolves the array's address there; otherwise the first call
 * does).  All calls on one GPU share the array: calls that overlap on a GPU --
 * from several threads, streams or scene handles -- race. */
int rt_accumulate_static(rt_scene* scene, const rt_camera* cam, const rt_options* opt,
                         uint32_t sample_begin, uint32_t sample_end, float* d_sum, void* stream);

/* Replaces the reference's progress bar (src/raytrace.cu:601-627, a mapped
 * counter the host polls while the kernel runs).  A frame is rendered in
 * batches of at most 2^24 paths; when a callback is set, rt_accumulate and
 * rt_render wait for each batch and then call fn(paths_done, paths_total,
 * user) on the calling thread.  fn == NULL removes it (the default: batches
 * are only enqueued). */
typedef void (*rt_progress_fn)(uint64_t paths_done, uint64_t paths_total, void* user);
int rt_set_progress(rt_scene* scene, rt_progress_fn fn, void* user);

/* Per-function entry points for known-answer tests.
 * rt_cast = get_intersection (src/raytrace.cu:179-216) for n rays given as
 * host arrays; out_t is +inf on a miss, ids are -1 when that kind is not
 * the nearest hit, out_abg (3n) is written for triangle hits only.
 * rt_visible = is_visible (src/raytrace.cu:218-244) for n segments. */
int rt_cast(rt_scene* scene, const rt_options* opt, uint32_t n, const float* origins,
            const float* dirs, float* out_t, int32_t* out_sphere, int32_t* out_triangle,
            float* out_abg);
int rt_visible(rt_scene* scene, const rt_options* opt, uint32_t n, const float* origins,
               const float* targets, int32_t* out_visible);
/* rt_random_cos = random_cos with CU_Vector::argmin_abs (src/raytrace.cu:253-296,
 * include/vector.cuh:87-98) on the device, for n triples of a normal (3n
 * floats) and the two uniforms the reference draws; out_dirs receives 3n
 * floats.  All host arrays. */
int rt_random_cos(uint32_t n, const float* normals, const float* r1, const float* r2, float* out_dirs);

/* ---- host scene API: C view of class Scene ------------------------------
 * Mirrors the reference's public Scene interface (include/scene.cuh:8-31);
 * each function names the method it wraps. */
typedef struct rt_host_scene rt_host_scene;

int rt_host_scene_create(rt_host_scene** out);                       /* Scene::Scene      */
int rt_host_scene_destroy(rt_host_scene* s);                          /* Scene::~Scene     */
int rt_host_scene_load(rt_host_scene* s, const char* json_path);      /* load_scene        */
int rt_host_scene_set_camera_intrinsics(rt_host_scene* s, float fov, size_t width, size_t height);
int rt_host_scene_rotate_camera(rt_host_scene* s, float alpha, float beta, float gamma);
int rt_host_scene_transform_camera(rt_host_scene* s, float x, float y, float z);
int rt_host_scene_set_num_rays(rt_host_scene* s, uint32_t num_rays);
int rt_host_scene_add_sphere(rt_host_scene* s, const float pos[3], float radius,
                             const float color[3], int32_t material_type, float ro, float ri);
int rt_host_scene_add_object(rt_host_scene* s, const char* obj_path, const float pos[3],
                             const float scale[3], const float rotation[3],
                             const float color[3], int32_t material_type, float ro, float ri);
int rt_host_scene_add_light(rt_host_scene* s, const float pos[3], float intensity);
int rt_host_scene_set_options(rt_host_scene* s, const rt_options* opt);
/* Scene::settings.devices: render() and render_frames() split every frame over
 * HIP devices 0 .. n_devices-1 through an rt_device_group (1 = the current
 * device only, the default). */
int rt_host_scene_set_devices(rt_host_scene* s, uint32_t n_devices);
/* Scene::render: renders on the current device, clamps to 8 bits like the
 * reference (src/scene.cu:214-220) and writes an RGB PNG.  png_path may be
 * NULL to skip the file. */
int rt_host_scene_render(rt_host_scene* s, const char* png_path, size_t seed);
/* The frame loop the reference's README describes (README.md:24: move the
 * camera with rotate_camera / transform_camera between calls of
 * Scene::render and pass the frame number as the seed).  Frame i (0-based)
 * is drawn after i applications of rotate_step (degrees) and move_step, with
 * seed first_seed + i, into the file named by `pattern` (a printf pattern
 * with one integer conversion, e.g. "out/frame_%04d.png"; ".pfm" selects a
 * float image instead of a PNG).  Only frames i = first, first + stride, ...
 * are rendered (one process per GPU: first = rank, stride = number of
 * ranks); the camera still takes every step.  The scene stays on the device
 * for the whole sequence and files are encoded by `writers` host threads
 * while the next frame renders.  out_seconds (may be NULL) receives the wall
 * time of the loop including the last file. */
int rt_host_scene_render_frames(rt_host_scene* s, const char* pattern, size_t first_seed,
                                uint32_t n_frames, const float rotate_step[3],
                                const float move_step[3], uint32_t first, uint32_t stride,
                                uint32_t writers, double* out_seconds);
/* Borrowed views, valid until the scene is modified or destroyed. */
int rt_host_scene_describe(rt_host_scene* s, rt_scene_desc* out_desc, rt_camera* out_cam);
const float* rt_host_scene_image(rt_host_scene* s, uint32_t* out_width, uint32_t* out_height);

/* Encodes an 8-bit RGB image as PNG (the job stbi_write_png does in the
 * reference, src/scene.cu:221). */
int rt_write_png(const char* path, uint32_t width, uint32_t height, const uint8_t* rgb);
/* Writes the unclamped float image (3 * width * height floats, row 0 on top,
 * the layout rt_render returns) as a little-endian colour PFM, for
 * convergence studies the reference's 8-bit PNG cannot serve. */
int rt_write_pfm(const char* path, uint32_t width, uint32_t height, const float* rgb);

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H */
