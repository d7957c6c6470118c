/*
 * pathtrace_oracle.h -- TEST INFRASTRUCTURE, not product code.
 *
 * Scalar CPU restatement of the reference's per-pixel path-trace routine
 * (reference src/raytrace.cu:33-507).  Only tests, the smoke check and the
 * benchmark's cpu_baseline leg may use it; the renderer never links it.
 *
 * Parity status: the reference routine cannot be compiled in this build
 * environment (it needs the CUDA toolkit's cuRAND and runtime headers), and
 * the reference ships no golden vectors for its renderer.  This restatement is
 * therefore pinned by (1) anchor pixels recorded earlier from a host compile of
 * the reference routine (deterministic mode; values and provenance in
 * tests/golden/anchors.json) and (2) the reference's shipped example renders
 * (statistical comparison only: they were drawn with a wall-clock seed).  The
 * random-number stream is this repository's own (rt_math.h), because cuRAND
 * XORWOW has no public definition here: parity of stochastic modes against
 * the reference is statistical, PER-SAMPLE PARITY WITH THE REFERENCE IS
 * UNPINNED.
 */
#ifndef PATHTRACE_ORACLE_H
#define PATHTRACE_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_DIFFUSE = 0, ORC_MIRROR = 1, ORC_GLASS = 2 };

/* Flat scene arrays.  Counts and meanings follow the argument list of the
 * reference's raytrace_spheres() (src/raytrace.cu:509-522); the per-sphere and
 * per-triangle Material copies of the reference (include/scene_objects.cuh:
 * 24-43) are expressed as an index into one material table. */
typedef struct orc_scene {
    uint32_t n_spheres;
    const float* sphere_pos;        /* 3 * n_spheres */
    const float* sphere_radius;     /* n_spheres */
    const int32_t* sphere_material; /* n_spheres */
    uint32_t n_materials;
    const int32_t* material_type;   /* ORC_DIFFUSE / MIRROR / GLASS */
    const float* material_color;    /* 3 * n_materials */
    const float* material_ro;       /* refractive index outside */
    const float* material_ri;       /* refractive index inside */
    uint32_t n_vertices;
    const float* vertices;          /* 3 * n_vertices */
    const float* normals;           /* 3 * n_vertices */
    uint32_t n_triangles;
    const int32_t* tri_v;           /* 3 * n_triangles, into vertices */
    const int32_t* tri_n;           /* 3 * n_triangles, into normals */
    const int32_t* tri_material;    /* n_triangles */
    uint32_t n_boxes;
    const float* box_min;           /* 3 * n_boxes */
    const float* box_max;           /* 3 * n_boxes */
    uint32_t n_lights;
    const float* light_pos;         /* 3 * n_lights */
    const float* light_intensity;   /* n_lights */
} orc_scene;

/* cam_rot = rotation block of the extrinsic matrix, row-major; fx, fy, cx, cy
 * = K[0][0], K[1][1], K[0][2], K[1][2] (reference src/raytrace.cu:538-539,
 * include/projection_helpers.cuh:69-75). */
typedef struct orc_camera {
    float rot[9];
    float pos[3];
    float fx, fy, cx, cy;
    uint32_t width, height, num_rays;
} orc_camera;

/* The reference's three compile-time switches (src/raytrace.cu:14-16) and
 * MAX_RAY_DEPTH (:12) as run-time options.  use_libm selects glibc's
 * logf/sinf/cosf/powf instead of rt_math.h's fixed polynomials. */
typedef struct orc_options {
    int32_t indirect;
    int32_t antialiasing;
    int32_t bounding_box;
    int32_t max_depth;
    uint32_t seed;
    int32_t use_libm;
} orc_options;

/* Work actually executed, summed over the rendered paths. */
typedef struct orc_counters {
    uint64_t paths;
    uint64_t bounce_iterations;
    uint64_t casts;           /* closest-hit + shadow casts */
    uint64_t sphere_tests;
    uint64_t box_tests;
    uint64_t mesh_walks;      /* casts that walked the triangle list */
    uint64_t triangle_tests;
    uint64_t uniforms;
} orc_counters;

/* Renders the listed pixels (pixel = y*width + x) over global sample indices
 * [sample_begin, sample_end).  out_rgb (3n, may be NULL) receives the
 * reference's output value gamma((1/actual_rays) * colour) on the 0..255+
 * scale, unclamped; out_sum (3n, may be NULL) the linear colour sum and
 * out_count (n, may be NULL) actual_rays.  Returns 0, or -1 on bad input. */
int orc_render_pixels(const orc_scene* scene, const orc_camera* cam, const orc_options* opt,
                      const uint32_t* pixels, uint32_t n_pixels,
                      uint32_t sample_begin, uint32_t sample_end,
                      float* out_rgb, float* out_sum, uint32_t* out_count,
                      orc_counters* counters, int32_t n_threads);

/* get_intersection (src/raytrace.cu:179-216) for n rays.  out_t = distance
 * (+inf on a miss); ids are -1 when that kind was not the nearest hit;
 * out_abg = barycentrics of a triangle hit (3n, untouched otherwise). */
int orc_cast(const orc_scene* scene, const orc_options* opt, uint32_t n,
             const float* origins, const float* dirs,
             float* out_t, int32_t* out_sphere, int32_t* out_triangle, float* out_abg);

/* is_visible (src/raytrace.cu:218-244) for n (origin, target) pairs. */
int orc_visible(const orc_scene* scene, const orc_options* opt, uint32_t n,
                const float* origins, const float* targets, int32_t* out_visible);

/* random_cos (src/raytrace.cu:253-296) for n (normal, r1, r2) triples. */
int orc_random_cos(uint32_t n, const float* normals, const float* r1, const float* r2,
                   int32_t use_libm, float* out_dirs);

/* Primary ray of (pixel, sample): anti-aliasing jitter + pixel_to_camera +
 * normalise + rotate (src/raytrace.cu:479-492).  out_dims = uniforms drawn. */
int orc_primary_ray(const orc_camera* cam, const orc_options* opt, uint32_t pixel,
                    uint32_t sample, float* out_dir, uint32_t* out_dims);

/* The generator's dim-th uniform of (seed, pixel, sample), for n dims. */
int orc_uniforms(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t first_dim,
                 uint32_t n, float* out);

/* The fixed polynomial kernels of rt_math.h, for accuracy tests:
 * kind 0 = log, 1 = sin, 2 = cos, 3 = output transfer curve x^(1/2.2). */
int orc_math_probe(int32_t kind, uint32_t n, const float* in, float* out);

#ifdef __cplusplus
}
#endif
#endif
