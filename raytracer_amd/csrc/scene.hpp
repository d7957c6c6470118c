// scene.hpp -- host-side scene container with the reference's public Scene
// interface (reference include/scene.cuh:8-31) and the plain-data types its
// signatures use (reference include/scene_objects.cuh, include/vector.cuh).
//
// The layouts are this library's own: a vector is three floats (the
// reference's CU_Vector carries an extra `int N`), and the scene is kept as
// flat arrays that map one-to-one onto rt_scene_desc, so rendering needs no
// conversion pass.
#pragma once

#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rt_amd.h"

struct Vec3f {
    float v[3];
    Vec3f() : v{0.f, 0.f, 0.f} {}
    Vec3f(float x, float y, float z) : v{x, y, z} {}
    const float& operator[](int i) const { return v[i]; }
    float& operator[](int i) { return v[i]; }
};
typedef Vec3f CU_Vector3f; // the name the reference's signatures use

struct Mat3 {
    float m[9]; // row-major
    const float& operator[](int i) const { return m[i]; }
    float& operator[](int i) { return m[i]; }
};

struct Mat4 {
    float m[16]; // row-major
    const float& operator[](int i) const { return m[i]; }
    float& operator[](int i) { return m[i]; }
    float& operator()(int r, int c) { return m[4 * r + c]; }
    void set_identity();
    Mat3 get_rotation() const;       // upper-left 3x3 block
    Vec3f get_translation() const;   // last column
};

enum material_type { DIFFUSE = 0, MIRROR, GLASS };

// What class Scene throws besides the reference's std::invalid_argument("File not found"): the
// message plus the RT_ERR_* class the C view of the class reports (rt_host_scene_*).
struct SceneError : std::runtime_error {
    int code;
    SceneError(int code_, const std::string& message) : std::runtime_error(message), code(code_) {}
};

struct Material {
    material_type type = DIFFUSE;
    Vec3f color;
    float ro = 0.f; // refractive index outside
    float ri = 0.f; // refractive index inside
};

struct Sphere {
    Vec3f pos;
    float radius = 0.f;
    Material material;
};

struct Light {
    Vec3f pos;
    float I = 0.f;
};

struct BoundingBox {
    Vec3f _min;
    Vec3f _max;
};

struct Triangle {
    int v[3]; // vertex indices
    int n[3]; // normal indices
    int group = 0;
    Material material;
};

struct Camera {
    Mat4 E;  // extrinsics: rotation block + position column
    Mat3 K;  // intrinsics: f_x, f_y on the diagonal, c_x, c_y in the last column
    unsigned width = 512;
    unsigned height = 512;
    unsigned num_rays = 1;
};

// The reference's three mode macros and its bounce limit
// (reference src/raytrace.cu:12-16) as run-time settings; the defaults are the
// reference's shipped configuration.
struct RenderSettings {
    bool indirect_lighting = true;
    bool antialiasing = true;
    bool bounding_box = true;
    int max_ray_depth = 5;
    // not in the reference: skip clusters of triangles whose bounds a ray misses
    // (false = the reference's exhaustive triangle scan)
    bool cluster_culling = true;
    // not in the reference's source, but what its build does: fused multiply-adds and the hardware's reciprocal,
    // square-root and log / sin / cos instructions instead of the bit-exact restatement (rt_options.arithmetic = 1)
    bool fast_arithmetic = false;
    // not in the reference: split every frame over HIP devices 0 .. devices-1 of this node
    // (samples per pixel first, bands of rows beyond that; sums exchanged over RCCL or peer copies)
    unsigned devices = 1;
};

class Scene {
public:
    Scene();
    ~Scene();
    Scene(const Scene&) = delete;
    Scene& operator=(const Scene&) = delete;

    // Reads a scene file; throws std::invalid_argument("File not found") when
    // it cannot be opened and std::runtime_error on malformed input.  Unlike
    // the reference it starts from an empty scene and an identity camera each
    // time, so loading twice does not accumulate.
    void load_scene(std::string filepath);

    // Renders camera->num_rays samples per pixel on the current device and
    // writes an 8-bit RGB PNG.  The image depends only on the scene and seed.
    void render(std::string imagepath, size_t seed);

    // The frame loop of the reference's README (README.md:24): frame i is drawn
    // after i camera steps (rotate_camera(rotate_step), then
    // transform_camera(move_step)) with seed first_seed + i and written to the
    // file `pattern` names for i ("frame_%04d.png"; ".pfm" = float image;
    // empty = no files).  Frames first, first + stride, ... are rendered, the
    // camera takes every step.  Files are encoded by `writers` host threads
    // while the next frame renders.  Returns the wall time in seconds.
    double render_frames(std::string pattern, size_t first_seed, unsigned n_frames, CU_Vector3f rotate_step,
                         CU_Vector3f move_step, unsigned first = 0, unsigned stride = 1, unsigned writers = 4);

    void set_camera_intrinsics(float fov, size_t width, size_t height);

    void rotate_camera(float alpha, float beta, float gamma); // degrees, R = Rz Ry Rx
    void rotate_camera(CU_Vector3f rotation);
    void transform_camera(float x, float y, float z);
    void transform_camera(CU_Vector3f direction);

    void add_sphere(CU_Vector3f pos, float radius, CU_Vector3f color, material_type type, float ro, float ri);
    // Throws std::runtime_error when the mesh file cannot be read (the
    // reference prints the loader's message and calls exit(1)).
    void add_object(std::string obj_path, CU_Vector3f pos, CU_Vector3f scale, CU_Vector3f rotation, Material material);
    void add_light(CU_Vector3f pos, float intensity);

    std::vector<Sphere> get_spheres();
    size_t get_sphere_count();
    Camera* get_camera();

    // ---- additions over the reference interface ----
    RenderSettings settings;
    bool verbose = true; // loader statistics and the timing line on stdout

    size_t get_triangle_count() const { return tri_material_.size(); }
    size_t get_vertex_count() const { return vertices_.size() / 3; }
    std::vector<BoundingBox> get_bounding_boxes() const;
    const float* get_image() const { return image_.data(); } // 3 floats per pixel, unclamped
    const rt_stats& last_stats() const { return stats_; }

    // Views of the flat arrays for rt_scene_upload / rt_render; valid until
    // the scene changes.
    void describe(rt_scene_desc* desc, rt_camera* cam) const;

private:
    int add_material(const Material& m);
    void release_device_scene();
    void sync_device_scene(); // uploads the scene when it changed since the last upload
    int render_frame(const rt_camera& cam, const rt_options& opt); // on the device(s) the settings name

    std::vector<float> sphere_pos_, sphere_radius_;
    std::vector<int32_t> sphere_material_;
    std::vector<int32_t> material_type_;
    std::vector<float> material_color_, material_ro_, material_ri_;
    std::vector<float> vertices_, normals_;
    std::vector<int32_t> tri_v_, tri_n_, tri_material_, tri_group_;
    std::vector<float> box_min_, box_max_;
    std::vector<float> light_pos_, light_intensity_;

    Camera* camera;
    std::vector<float> image_;
    std::string scene_dir_; // directory of the last scene file, for mesh lookup
    rt_scene* device_scene_ = nullptr;
    rt_device_group* device_group_ = nullptr; // instead of device_scene_ when settings.devices > 1
    unsigned group_devices_ = 0;
    bool device_scene_stale_ = true;
    rt_stats stats_{};
};

// The function the reference's Scene::render calls (reference
// src/scene.cu:21-35,210; src/raytrace.cu:509-522), for callers that hold the
// scene as arrays of structs.  `normals` has vertex_count entries like
// `vertices`; `image` receives width*height output values.  The seed is used
// as given.  Returns an RT_* status instead of void.
int raytrace_spheres(Sphere* spheres, size_t sphere_count, Triangle* triangles, size_t triangle_count,
                     CU_Vector3f* vertices, CU_Vector3f* normals, BoundingBox* bounding_boxes,
                     size_t vertex_count, size_t obj_count, Light* lights, size_t light_count,
                     CU_Vector3f* image, Camera* camera, size_t seed,
                     const RenderSettings* settings = nullptr);
