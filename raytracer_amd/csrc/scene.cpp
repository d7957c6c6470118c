// scene.cpp -- Scene loading, camera set-up, mesh flattening and the render
// entry point.  The arithmetic follows the reference statement by statement
// where it decides pixel values (camera matrices, mesh transform, bounding
// boxes); citations are to the reference repository.
#include "scene.hpp"

#include <zlib.h>

#include <chrono>
#include <cmath>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <stdexcept>

#include "json_mini.hpp"
#include "obj_mini.hpp"

// ---------------------------------------------------------------- matrices

void Mat4::set_identity() {
    std::memset(m, 0, sizeof m);
    for (int i = 0; i < 4; i++) m[4 * i + i] = 1.f;
}

Mat3 Mat4::get_rotation() const {
    Mat3 r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r.m[3 * i + j] = m[4 * i + j];
    return r;
}

Vec3f Mat4::get_translation() const { return Vec3f(m[3], m[7], m[11]); }

namespace {

// 4x4 product, every element accumulated from 0.f over k = 0..3
// (reference include/matrix.cuh:97-119).
Mat4 matmul(const Mat4& a, const Mat4& b) {
    Mat4 c;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            float acc = 0.f;
            for (int k = 0; k < 4; k++) acc += a.m[4 * i + k] * b.m[4 * k + j];
            c.m[4 * i + j] = acc;
        }
    return c;
}

// Euler angles in degrees, R = Rz * Ry * Rx; the angle is converted in double
// and narrowed at the cosf/sinf call (reference src/scene.cu:37-63).
Mat4 rotation_matrix(float alpha, float beta, float gamma) {
    Mat4 rx, ry, rz;
    rx.set_identity();
    ry.set_identity();
    rz.set_identity();

    rx(1, 1) = cosf(alpha * M_PI / 180);
    rx(1, 2) = -sinf(alpha * M_PI / 180);
    rx(2, 1) = -rx(1, 2);
    rx(2, 2) = rx(1, 1);

    ry(0, 0) = cosf(beta * M_PI / 180);
    ry(0, 2) = sinf(beta * M_PI / 180);
    ry(2, 0) = -ry(0, 2);
    ry(2, 2) = ry(0, 0);

    rz(0, 0) = cosf(gamma * M_PI / 180);
    rz(0, 1) = -sinf(gamma * M_PI / 180);
    rz(1, 0) = -rz(0, 1);
    rz(1, 1) = rz(0, 0);

    return matmul(matmul(rz, ry), rx);
}

// 3x3 * vec3 and the affine 4x4 * vec3 (reference include/matrix.cuh:67-95).
Vec3f apply3(const Mat3& m, const Vec3f& v) {
    Vec3f r;
    for (int i = 0; i < 3; i++) {
        float acc = 0.f;
        for (int j = 0; j < 3; j++) acc += m.m[3 * i + j] * v[j];
        r[i] = acc;
    }
    return r;
}

Vec3f apply_affine(const Mat4& m, const Vec3f& v) {
    Vec3f r;
    for (int i = 0; i < 3; i++) {
        float acc = 0.f;
        for (int j = 0; j < 3; j++) acc += m.m[4 * i + j] * v[j];
        r[i] = acc + m.m[4 * i + 3];
    }
    return r;
}

Vec3f sub(const Vec3f& a, const Vec3f& b) { return Vec3f(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }

Vec3f cross(const Vec3f& a, const Vec3f& b) {
    return Vec3f(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
}

// in-order sum of squares, sqrtf, then one division per component
// (reference include/vector.cuh:100-113)
Vec3f normalized(const Vec3f& a) {
    float acc = 0.f;
    for (int i = 0; i < 3; i++) acc += a[i] * a[i];
    const float n = sqrtf(acc);
    return Vec3f(a[0] / n, a[1] / n, a[2] / n);
}

std::string directory_of(const std::string& path) {
    size_t cut = path.find_last_of('/');
    if (cut == std::string::npos) return ".";
    if (cut == 0) return "/";
    return path.substr(0, cut);
}

bool readable(const std::string& path) {
    std::ifstream f(path);
    return f.is_open();
}

// The reference opens mesh paths relative to the working directory (its
// scenes say "../assets/..." and expect to run from build/).  That is tried
// first; then the path is tried below the scene file's directory and each of
// its ancestors, with leading "../" components dropped.
std::string locate_mesh(const std::string& path, const std::string& scene_dir) {
    if (readable(path) || path.empty() || path[0] == '/') return path;
    std::string rest = path;
    while (rest.compare(0, 3, "../") == 0) rest = rest.substr(3);
    std::string dir = scene_dir.empty() ? "." : scene_dir;
    for (int level = 0; level < 6; level++) {
        std::string candidate = dir + "/" + rest;
        if (readable(candidate)) return candidate;
        if (level == 0) {
            candidate = dir + "/" + path;
            if (readable(candidate)) return candidate;
        }
        dir += "/..";
    }
    return path;
}

Vec3f vec3_from(const rt::json::Value& a) { return Vec3f(a[0].as_float(), a[1].as_float(), a[2].as_float()); }

// material fields of a sphere or object entry (reference src/scene.cu:122-142,
// 165-191): colour only for diffuse, refractive indices only for glass
Material material_from(const rt::json::Value& node) {
    Material m;
    const std::string& kind = node["material"].as_string();
    if (kind == "diffuse") {
        m.type = DIFFUSE;
        m.color = vec3_from(node["color"]);
    } else if (kind == "mirror") {
        m.type = MIRROR;
    } else if (kind == "glass") {
        m.type = GLASS;
        m.ro = node["refractive_index"][0].as_float();
        m.ri = node["refractive_index"][1].as_float();
    }
    return m;
}

} // namespace

// ------------------------------------------------------------------- Scene

Scene::Scene() {
    camera = new Camera();
    camera->E.set_identity();
    std::memset(camera->K.m, 0, sizeof camera->K.m);
    set_camera_intrinsics(60.f, 512, 512);
    image_.assign(size_t(3) * camera->width * camera->height, 0.f);
}

Scene::~Scene() {
    release_device_scene();
    delete camera;
}

void Scene::release_device_scene() {
    if (device_scene_) rt_scene_free(device_scene_);
    if (device_group_) rt_device_group_free(device_group_);
    device_scene_ = nullptr;
    device_group_ = nullptr;
    group_devices_ = 0;
    device_scene_stale_ = true;
}

void Scene::load_scene(std::string filepath) {
    std::ifstream ifs(filepath);
    if (!ifs.is_open()) throw std::invalid_argument("File not found");
    std::stringstream buffer;
    buffer << ifs.rdbuf();
    rt::json::Value parsed;
    try {
        parsed = rt::json::parse(buffer.str());
    } catch (const std::exception& e) {
        throw SceneError(RT_ERR_PARSE, e.what());
    }
    const rt::json::Value& jf = parsed;

    sphere_pos_.clear(); sphere_radius_.clear(); sphere_material_.clear();
    material_type_.clear(); material_color_.clear(); material_ro_.clear(); material_ri_.clear();
    vertices_.clear(); normals_.clear();
    tri_v_.clear(); tri_n_.clear(); tri_material_.clear(); tri_group_.clear();
    box_min_.clear(); box_max_.clear();
    light_pos_.clear(); light_intensity_.clear();
    device_scene_stale_ = true;
    scene_dir_ = directory_of(filepath);

    // the camera looks down -z with y up: E starts as diag(1,-1,-1,1)
    // (reference src/scene.cu:86-87)
    camera->E.set_identity();
    camera->E[1 * 4 + 1] = -1.f;
    camera->E[2 * 4 + 2] = -1.f;

    const rt::json::Value& cam = jf["camera"];
    Vec3f camera_pos = vec3_from(cam["pos"]);
    Vec3f camera_rot = vec3_from(cam["rotation"]);
    unsigned width = static_cast<unsigned>(cam["width"].as_double());
    unsigned height = static_cast<unsigned>(cam["height"].as_double());
    float fov = cam["fov"].as_float();

    set_camera_intrinsics(fov, width, height);
    rotate_camera(camera_rot);
    transform_camera(camera_pos);
    camera->num_rays = static_cast<unsigned>(cam["num_rays"].as_double());

    const rt::json::Value& spheres = jf["spheres"];
    for (size_t i = 0; i < spheres.size(); i++) {
        const rt::json::Value& s = spheres[i];
        Material m = material_from(s);
        add_sphere(vec3_from(s["pos"]), s["radius"].as_float(), m.color, m.type, m.ro, m.ri);
    }

    const rt::json::Value& objects = jf["objects"];
    for (size_t i = 0; i < objects.size(); i++) {
        const rt::json::Value& o = objects[i];
        add_object(o["filepath"].as_string(), vec3_from(o["pos"]), vec3_from(o["scale"]),
                   vec3_from(o["rotation"]), material_from(o));
    }

    const rt::json::Value& lights = jf["lights"];
    for (size_t i = 0; i < lights.size(); i++)
        add_light(vec3_from(lights[i]["pos"]), lights[i]["intensity"].as_float());

    image_.assign(size_t(3) * camera->width * camera->height, 0.f);
}

// f_x = f_y = width / (2 tan(fov/2)) with fov in degrees (horizontal),
// principal point at the integer image centre (reference src/scene.cu:224-242).
void Scene::set_camera_intrinsics(float fov, size_t width, size_t height) {
    camera->K[0 * 3 + 0] = width / (2 * tanf(M_PI * fov / 360.f));
    camera->K[1 * 3 + 1] = width / (2 * tanf(M_PI * fov / 360.f));
    camera->K[2 * 3 + 2] = 1.f;
    camera->K[0 * 3 + 2] = width / 2;
    camera->K[1 * 3 + 2] = height / 2;
    camera->width = static_cast<unsigned>(width);
    camera->height = static_cast<unsigned>(height);
    image_.assign(size_t(3) * width * height, 0.f);
}

void Scene::rotate_camera(float alpha, float beta, float gamma) {
    camera->E = matmul(rotation_matrix(alpha, beta, gamma), camera->E);
}

void Scene::rotate_camera(CU_Vector3f rotation) { rotate_camera(rotation[0], rotation[1], rotation[2]); }

void Scene::transform_camera(float x, float y, float z) {
    camera->E[0 * 4 + 3] += x;
    camera->E[1 * 4 + 3] += y;
    camera->E[2 * 4 + 3] += z;
}

void Scene::transform_camera(CU_Vector3f direction) { transform_camera(direction[0], direction[1], direction[2]); }

int Scene::add_material(const Material& m) {
    material_type_.push_back(static_cast<int32_t>(m.type));
    for (int i = 0; i < 3; i++) material_color_.push_back(m.color[i]);
    material_ro_.push_back(m.ro);
    material_ri_.push_back(m.ri);
    return static_cast<int>(material_type_.size()) - 1;
}

void Scene::add_sphere(CU_Vector3f pos, float radius, CU_Vector3f color, material_type type, float ro, float ri) {
    Material m;
    m.type = type;
    m.color = color;
    m.ro = ro;
    m.ri = ri;
    for (int i = 0; i < 3; i++) sphere_pos_.push_back(pos[i]);
    sphere_radius_.push_back(radius);
    sphere_material_.push_back(add_material(m));
    device_scene_stale_ = true;
}

// Mesh flattening (reference src/scene.cu:278-413): every face contributes
// three fresh transformed vertices and three normals, and every shape of the
// file one axis-aligned bounding box.  Indices are global here; the reference
// restarts them per shape, which is only meaningful for its single-shape
// scenes.
void Scene::add_object(std::string obj_path, CU_Vector3f translation, CU_Vector3f scale, CU_Vector3f rotation, Material material) {
    // v' = R * S * v + t (reference src/scene.cu:280-284, include/matrix.cuh:41-55)
    Mat4 transform;
    transform.set_identity();
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 4; j++) transform.m[4 * i + j] *= scale[i];
    transform = matmul(rotation_matrix(rotation[0], rotation[1], rotation[2]), transform);
    for (int i = 0; i < 3; i++) transform.m[4 * i + 3] += translation[i];
    const Mat3 linear = transform.get_rotation();

    rt::obj::Mesh mesh;
    try {
        mesh = rt::obj::load(locate_mesh(obj_path, scene_dir_));
    } catch (const std::exception& e) {
        const std::string what = e.what();
        throw SceneError(what.find("Cannot open") != std::string::npos ? RT_ERR_IO : RT_ERR_PARSE, "TinyObjReader: " + what);
    }

    const int material_index = add_material(material);
    const size_t first_box = box_min_.size() / 3;

    for (size_t s = 0; s < mesh.shapes.size(); s++) {
        const std::vector<rt::obj::Corner>& corners = mesh.shapes[s].corners;
        Vec3f bmax(-INFINITY, -INFINITY, -INFINITY);
        Vec3f bmin(INFINITY, INFINITY, INFINITY);

        for (size_t f = 0; f + 2 < corners.size(); f += 3) {
            const int base = static_cast<int>(vertices_.size() / 3);
            bool has_normal = true;
            Vec3f face_normals[3];
            Vec3f face_vertices[3];

            for (int k = 0; k < 3; k++) {
                const rt::obj::Corner& c = corners[f + k];
                Vec3f vtx(mesh.positions[3 * c.vertex + 0], mesh.positions[3 * c.vertex + 1], mesh.positions[3 * c.vertex + 2]);
                vtx = apply_affine(transform, vtx);
                // an update of the minimum skips the maximum check
                // (reference src/scene.cu:330-334)
                for (int i = 0; i < 3; i++) {
                    if (vtx[i] < bmin[i]) bmin[i] = vtx[i];
                    else if (vtx[i] > bmax[i]) bmax[i] = vtx[i];
                }
                if (c.normal >= 0)
                    face_normals[k] = Vec3f(mesh.normals[3 * c.normal + 0], mesh.normals[3 * c.normal + 1], mesh.normals[3 * c.normal + 2]);
                else
                    has_normal = false;
                face_vertices[k] = vtx;
                for (int i = 0; i < 3; i++) vertices_.push_back(vtx[i]);
            }
            for (int k = 0; k < 3; k++) {
                Vec3f n;
                if (has_normal) {
                    n = face_normals[k];
                } else {
                    // flat normal from the TRANSFORMED corners, then transformed
                    // once more like a stored normal (reference src/scene.cu:372-374)
                    n = cross(sub(face_vertices[1], face_vertices[0]), sub(face_vertices[2], face_vertices[0]));
                }
                n = normalized(apply3(linear, n));
                for (int i = 0; i < 3; i++) normals_.push_back(n[i]);
            }
            for (int k = 0; k < 3; k++) {
                tri_v_.push_back(base + k);
                tri_n_.push_back(base + k);
            }
            tri_group_.push_back(static_cast<int32_t>(s));
            tri_material_.push_back(material_index);
        }
        for (int i = 0; i < 3; i++) { box_min_.push_back(bmin[i]); box_max_.push_back(bmax[i]); }
    }
    device_scene_stale_ = true;

    if (verbose) {
        // the reference reports the first box of the scene (src/scene.cu:394-411)
        if (!box_min_.empty()) {
            std::cout << "Minimum bound: " << box_min_[0] << ", " << box_min_[1] << ", " << box_min_[2] << ", " << std::endl;
            std::cout << "Maximum bound: " << box_max_[0] << ", " << box_max_[1] << ", " << box_max_[2] << ", " << std::endl;
        }
        std::cout << "Loaded " << box_min_.size() / 3 << " objects." << std::endl;
        std::cout << "Loaded " << vertices_.size() / 3 << " vertices." << std::endl;
        std::cout << "Loaded " << normals_.size() / 3 << " normals." << std::endl;
        std::cout << "Loaded " << tri_material_.size() << " triangles." << std::endl;
    }
    (void)first_box;
}

void Scene::add_light(CU_Vector3f pos, float intensity) {
    for (int i = 0; i < 3; i++) light_pos_.push_back(pos[i]);
    light_intensity_.push_back(intensity);
    device_scene_stale_ = true;
}

std::vector<Sphere> Scene::get_spheres() {
    std::vector<Sphere> out(sphere_radius_.size());
    for (size_t i = 0; i < out.size(); i++) {
        out[i].pos = Vec3f(sphere_pos_[3 * i], sphere_pos_[3 * i + 1], sphere_pos_[3 * i + 2]);
        out[i].radius = sphere_radius_[i];
        const int m = sphere_material_[i];
        out[i].material.type = static_cast<material_type>(material_type_[m]);
        out[i].material.color = Vec3f(material_color_[3 * m], material_color_[3 * m + 1], material_color_[3 * m + 2]);
        out[i].material.ro = material_ro_[m];
        out[i].material.ri = material_ri_[m];
    }
    return out;
}

size_t Scene::get_sphere_count() { return sphere_radius_.size(); }
Camera* Scene::get_camera() { return camera; }

std::vector<BoundingBox> Scene::get_bounding_boxes() const {
    std::vector<BoundingBox> out(box_min_.size() / 3);
    for (size_t i = 0; i < out.size(); i++) {
        out[i]._min = Vec3f(box_min_[3 * i], box_min_[3 * i + 1], box_min_[3 * i + 2]);
        out[i]._max = Vec3f(box_max_[3 * i], box_max_[3 * i + 1], box_max_[3 * i + 2]);
    }
    return out;
}

void Scene::describe(rt_scene_desc* d, rt_camera* c) const {
    if (d) {
        std::memset(d, 0, sizeof *d);
        d->n_spheres = static_cast<uint32_t>(sphere_radius_.size());
        d->sphere_pos = sphere_pos_.data();
        d->sphere_radius = sphere_radius_.data();
        d->sphere_material = sphere_material_.data();
        d->n_materials = static_cast<uint32_t>(material_type_.size());
        d->material_type = material_type_.data();
        d->material_color = material_color_.data();
        d->material_ro = material_ro_.data();
        d->material_ri = material_ri_.data();
        d->n_vertices = static_cast<uint32_t>(vertices_.size() / 3);
        d->vertices = vertices_.data();
        d->normals = normals_.data();
        d->n_triangles = static_cast<uint32_t>(tri_material_.size());
        d->tri_v = tri_v_.data();
        d->tri_n = tri_n_.data();
        d->tri_material = tri_material_.data();
        d->n_boxes = static_cast<uint32_t>(box_min_.size() / 3);
        d->box_min = box_min_.data();
        d->box_max = box_max_.data();
        d->n_lights = static_cast<uint32_t>(light_intensity_.size());
        d->light_pos = light_pos_.data();
        d->light_intensity = light_intensity_.data();
    }
    if (c) {
        // what the reference's launcher extracts from Camera (src/raytrace.cu:538-539)
        const Mat3 r = camera->E.get_rotation();
        const Vec3f t = camera->E.get_translation();
        for (int i = 0; i < 9; i++) c->rot[i] = r.m[i];
        for (int i = 0; i < 3; i++) c->pos[i] = t[i];
        c->fx = camera->K[0];
        c->fy = camera->K[4];
        c->cx = camera->K[2];
        c->cy = camera->K[5];
        c->width = camera->width;
        c->height = camera->height;
        c->num_rays = camera->num_rays;
    }
}

namespace {

rt_options options_from(const RenderSettings& s, size_t seed) {
    rt_options o;
    std::memset(&o, 0, sizeof o);
    o.indirect = s.indirect_lighting;
    o.antialiasing = s.antialiasing;
    o.bounding_box = s.bounding_box;
    o.max_depth = s.max_ray_depth;
    o.seed = static_cast<uint32_t>(seed);
    o.cluster_culling = s.cluster_culling;
    o.arithmetic = s.fast_arithmetic ? 1 : 0;
    return o;
}

// fminf(x, 255) then a truncating cast (reference src/scene.cu:216-219);
// NaN maps to 255 through fminf, negative values are clamped to 0 here
// (the reference's cast of a negative float is undefined).
std::vector<uint8_t> quantize(const float* image, size_t count) {
    std::vector<uint8_t> out(count);
    for (size_t i = 0; i < count; i++) {
        float v = fminf(image[i], 255);
        out[i] = v > 0.f ? static_cast<uint8_t>(v) : 0;
    }
    return out;
}

bool ends_with(const std::string& s, const char* suffix) {
    const size_t n = std::strlen(suffix);
    return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}

// ".pfm" -> the unclamped float image; anything else -> the reference's 8-bit PNG
void write_image(const std::string& path, uint32_t width, uint32_t height, const float* image) {
    int rc;
    if (ends_with(path, ".pfm")) {
        rc = rt_write_pfm(path.c_str(), width, height, image);
    } else {
        std::vector<uint8_t> rgb = quantize(image, size_t(3) * width * height);
        rc = rt_write_png(path.c_str(), width, height, rgb.data());
    }
    if (rc != RT_OK) throw SceneError(rc, rt_last_error());
}

// Host threads that encode finished frames while the device renders the next.
class FrameWriters {
public:
    explicit FrameWriters(unsigned n) : limit_(2 * size_t(n ? n : 1)) {
        for (unsigned i = 0; i < (n ? n : 1); i++) threads_.emplace_back([this] { run(); });
    }
    ~FrameWriters() { finish(); }

    // Blocks while 2 frames per thread are already waiting, which bounds the memory held.
    void submit(std::string path, uint32_t width, uint32_t height, std::vector<float> image) {
        std::unique_lock<std::mutex> lock(m_);
        room_.wait(lock, [this] { return jobs_.size() < limit_; });
        jobs_.push_back(Job{std::move(path), width, height, std::move(image)});
        work_.notify_one();
    }

    // Waits for every file; returns the first failure's message ("" = all written).
    std::string finish() {
        {
            std::lock_guard<std::mutex> lock(m_);
            closed_ = true;
        }
        work_.notify_all();
        for (std::thread& t : threads_) if (t.joinable()) t.join();
        return error_;
    }

private:
    struct Job {
        std::string path;
        uint32_t width, height;
        std::vector<float> image;
    };

    void run() {
        for (;;) {
            Job job;
            {
                std::unique_lock<std::mutex> lock(m_);
                work_.wait(lock, [this] { return closed_ || !jobs_.empty(); });
                if (jobs_.empty()) return;
                job = std::move(jobs_.front());
                jobs_.pop_front();
                room_.notify_one();
            }
            try {
                write_image(job.path, job.width, job.height, job.image.data());
            } catch (const std::exception& e) {
                std::lock_guard<std::mutex> lock(m_);
                if (error_.empty()) error_ = e.what();
            }
        }
    }

    std::mutex m_;
    std::condition_variable work_, room_;
    std::deque<Job> jobs_;
    std::vector<std::thread> threads_;
    std::string error_;
    size_t limit_;
    bool closed_ = false;
};

// printf pattern with exactly one integer conversion -> file name of frame i
std::string frame_path(const std::string& pattern, unsigned frame) {
    int conversions = 0;
    for (size_t i = 0; i < pattern.size(); i++) {
        if (pattern[i] != '%') continue;
        if (i + 1 < pattern.size() && pattern[i + 1] == '%') { i++; continue; }
        size_t j = i + 1;
        while (j < pattern.size() && (pattern[j] == '0' || pattern[j] == '-')) j++;
        while (j < pattern.size() && pattern[j] >= '0' && pattern[j] <= '9') j++;
        if (j >= pattern.size() || (pattern[j] != 'd' && pattern[j] != 'u' && pattern[j] != 'i'))
            throw SceneError(RT_ERR_INVALID, "frame pattern: only %d, %u, %i (with width) are accepted");
        conversions++;
        i = j;
    }
    if (conversions != 1) throw SceneError(RT_ERR_INVALID, "frame pattern needs exactly one integer conversion, e.g. frame_%04d.png");
    std::vector<char> buf(pattern.size() + 32);
    std::snprintf(buf.data(), buf.size(), pattern.c_str(), frame);
    return std::string(buf.data());
}

// The reference's progress line (src/raytrace.cu:601-627): a 50-column bar,
// the percentage and the predicted remaining time.
struct ProgressBar {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    bool drawn = false;

    static void report(uint64_t done, uint64_t total, void* user) {
        ProgressBar* self = static_cast<ProgressBar*>(user);
        if (done >= total && !self->drawn) return; // single-batch frame: nothing to show
        const double frac = total ? double(done) / double(total) : 1.0;
        const double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - self->t0).count();
        const double remaining = frac > 0 ? elapsed * (1.0 - frac) / frac : 0.0;
        char bar[51];
        const int filled = static_cast<int>(frac * 50);
        for (int i = 0; i < 50; i++) bar[i] = i < filled ? '=' : (i == filled ? '>' : ' ');
        bar[50] = 0;
        std::printf("\r[%s] %3d %% | Remaining: %.1f s   ", bar, static_cast<int>(frac * 100), remaining);
        if (done >= total) std::printf("\n");
        std::fflush(stdout);
        self->drawn = true;
    }
};

} // namespace

void Scene::sync_device_scene() {
    const unsigned devices = settings.devices > 1 ? settings.devices : 1;
    const bool have = devices > 1 ? (device_group_ && group_devices_ == devices) : device_scene_ != nullptr;
    if (device_scene_stale_ || !have) {
        rt_scene_desc desc;
        describe(&desc, nullptr);
        release_device_scene();
        int rc;
        if (devices > 1) {
            std::vector<int32_t> list(devices);
            for (unsigned i = 0; i < devices; i++) list[i] = static_cast<int32_t>(i);
            rc = rt_device_group_create(&desc, list.data(), devices, &device_group_);
            group_devices_ = devices;
        } else {
            rc = rt_scene_upload(&desc, &device_scene_);
        }
        if (rc != RT_OK) throw SceneError(rc, rt_last_error());
        device_scene_stale_ = false;
    }
}

int Scene::render_frame(const rt_camera& cam, const rt_options& opt) {
    if (device_group_) return rt_device_group_render(device_group_, &cam, &opt, image_.data(), &stats_);
    return rt_render(device_scene_, &cam, &opt, image_.data(), &stats_);
}

void Scene::render(std::string imagepath, size_t seed) {
    rt_camera cam;
    describe(nullptr, &cam);
    sync_device_scene();
    rt_options opt = options_from(settings, seed);
    image_.assign(size_t(3) * cam.width * cam.height, 0.f);

    ProgressBar bar;
    if (verbose && device_scene_) rt_set_progress(device_scene_, &ProgressBar::report, &bar);
    auto t0 = std::chrono::steady_clock::now();
    const int rc = render_frame(cam, opt);
    if (device_scene_) rt_set_progress(device_scene_, nullptr, nullptr);
    if (rc != RT_OK) throw SceneError(rc, rt_last_error());
    double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (verbose) std::printf("Finished raytracing in %.3f seconds.\n", seconds);

    if (!imagepath.empty()) write_image(imagepath, cam.width, cam.height, image_.data());
}

double Scene::render_frames(std::string pattern, size_t first_seed, unsigned n_frames, CU_Vector3f rotate_step,
                            CU_Vector3f move_step, unsigned first, unsigned stride, unsigned writers) {
    if (stride == 0) throw SceneError(RT_ERR_INVALID, "render_frames: stride must be at least 1");
    if (!pattern.empty()) (void)frame_path(pattern, 0); // reject a bad pattern before any work
    sync_device_scene();
    auto t0 = std::chrono::steady_clock::now();
    {
        FrameWriters pool(pattern.empty() ? 1 : writers);
        for (unsigned i = 0; i < n_frames; i++) {
            if (i > 0) { // the pose of frame i does not depend on which frames this process draws
                rotate_camera(rotate_step);
                transform_camera(move_step);
            }
            if (i < first || (i - first) % stride != 0) continue;
            rt_camera cam;
            describe(nullptr, &cam);
            rt_options opt = options_from(settings, first_seed + i);
            image_.assign(size_t(3) * cam.width * cam.height, 0.f);
            const int rc = render_frame(cam, opt);
            if (rc != RT_OK) throw SceneError(rc, rt_last_error());
            if (!pattern.empty()) pool.submit(frame_path(pattern, i), cam.width, cam.height, image_);
        }
        const std::string failure = pool.finish();
        if (!failure.empty()) throw SceneError(RT_ERR_IO, failure);
    }
    const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (verbose) std::printf("Rendered %u frames in %.3f seconds.\n", n_frames > first ? (n_frames - first + stride - 1) / stride : 0, seconds);
    return seconds;
}

int raytrace_spheres(Sphere* spheres, size_t sphere_count, Triangle* triangles, size_t triangle_count,
                     CU_Vector3f* vertices, CU_Vector3f* normals, BoundingBox* bounding_boxes,
                     size_t vertex_count, size_t obj_count, Light* lights, size_t light_count,
                     CU_Vector3f* image, Camera* camera, size_t seed, const RenderSettings* settings) {
    // arrays of structs -> the flat description; one material entry per
    // sphere and per triangle, exactly the copies the reference carries
    std::vector<float> sphere_pos(3 * sphere_count), sphere_radius(sphere_count);
    std::vector<int32_t> sphere_material(sphere_count);
    const size_t n_mat = sphere_count + triangle_count;
    std::vector<int32_t> mat_type(n_mat);
    std::vector<float> mat_color(3 * n_mat), mat_ro(n_mat), mat_ri(n_mat);
    auto put_material = [&](size_t slot, const Material& m) {
        mat_type[slot] = m.type;
        for (int i = 0; i < 3; i++) mat_color[3 * slot + i] = m.color[i];
        mat_ro[slot] = m.ro;
        mat_ri[slot] = m.ri;
    };
    for (size_t i = 0; i < sphere_count; i++) {
        for (int k = 0; k < 3; k++) sphere_pos[3 * i + k] = spheres[i].pos[k];
        sphere_radius[i] = spheres[i].radius;
        sphere_material[i] = static_cast<int32_t>(i);
        put_material(i, spheres[i].material);
    }
    std::vector<int32_t> tri_v(3 * triangle_count), tri_n(3 * triangle_count), tri_m(triangle_count);
    for (size_t i = 0; i < triangle_count; i++) {
        for (int k = 0; k < 3; k++) { tri_v[3 * i + k] = triangles[i].v[k]; tri_n[3 * i + k] = triangles[i].n[k]; }
        tri_m[i] = static_cast<int32_t>(sphere_count + i);
        put_material(sphere_count + i, triangles[i].material);
    }
    std::vector<float> box_min(3 * obj_count), box_max(3 * obj_count);
    for (size_t i = 0; i < obj_count; i++)
        for (int k = 0; k < 3; k++) { box_min[3 * i + k] = bounding_boxes[i]._min[k]; box_max[3 * i + k] = bounding_boxes[i]._max[k]; }
    std::vector<float> light_pos(3 * light_count), light_i(light_count);
    for (size_t i = 0; i < light_count; i++) {
        for (int k = 0; k < 3; k++) light_pos[3 * i + k] = lights[i].pos[k];
        light_i[i] = lights[i].I;
    }

    rt_scene_desc d;
    std::memset(&d, 0, sizeof d);
    d.n_spheres = static_cast<uint32_t>(sphere_count);
    d.sphere_pos = sphere_pos.data(); d.sphere_radius = sphere_radius.data(); d.sphere_material = sphere_material.data();
    d.n_materials = static_cast<uint32_t>(n_mat);
    d.material_type = mat_type.data(); d.material_color = mat_color.data(); d.material_ro = mat_ro.data(); d.material_ri = mat_ri.data();
    d.n_vertices = static_cast<uint32_t>(vertex_count);
    d.vertices = reinterpret_cast<const float*>(vertices);
    d.normals = reinterpret_cast<const float*>(normals);
    d.n_triangles = static_cast<uint32_t>(triangle_count);
    d.tri_v = tri_v.data(); d.tri_n = tri_n.data(); d.tri_material = tri_m.data();
    d.n_boxes = static_cast<uint32_t>(obj_count);
    d.box_min = box_min.data(); d.box_max = box_max.data();
    d.n_lights = static_cast<uint32_t>(light_count);
    d.light_pos = light_pos.data(); d.light_intensity = light_i.data();

    rt_camera c;
    const Mat3 r = camera->E.get_rotation();
    const Vec3f t = camera->E.get_translation();
    for (int i = 0; i < 9; i++) c.rot[i] = r.m[i];
    for (int i = 0; i < 3; i++) c.pos[i] = t[i];
    c.fx = camera->K[0]; c.fy = camera->K[4]; c.cx = camera->K[2]; c.cy = camera->K[5];
    c.width = camera->width; c.height = camera->height; c.num_rays = camera->num_rays;

    RenderSettings defaults;
    rt_options opt = options_from(settings ? *settings : defaults, seed);

    rt_scene* scene = nullptr;
    int rc = rt_scene_upload(&d, &scene);
    if (rc != RT_OK) return rc;
    rc = rt_render(scene, &c, &opt, reinterpret_cast<float*>(image), nullptr);
    rt_scene_free(scene);
    return rc;
}

// ------------------------------------------------------------ PNG encoding

namespace {

void put_be32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back(uint8_t(x >> 24)); v.push_back(uint8_t(x >> 16)); v.push_back(uint8_t(x >> 8)); v.push_back(uint8_t(x));
}

void put_chunk(std::vector<uint8_t>& file, const char tag[4], const std::vector<uint8_t>& body) {
    put_be32(file, static_cast<uint32_t>(body.size()));
    std::vector<uint8_t> tagged(tag, tag + 4);
    tagged.insert(tagged.end(), body.begin(), body.end());
    file.insert(file.end(), tagged.begin(), tagged.end());
    put_be32(file, static_cast<uint32_t>(crc32(0L, tagged.data(), static_cast<uInt>(tagged.size()))));
}

} // namespace

// thread-local message store shared with the device half of the library
void rt_set_error(const char* fmt, ...);

extern "C" int rt_write_png(const char* path, uint32_t width, uint32_t height, const uint8_t* rgb) {
    if (!path || !rgb || width == 0 || height == 0) { rt_set_error("rt_write_png: bad argument"); return RT_ERR_INVALID; }
    const size_t stride = size_t(3) * width;
    std::vector<uint8_t> raw((stride + 1) * height);
    for (uint32_t y = 0; y < height; y++) {
        raw[(stride + 1) * y] = 0; // filter type: none
        std::memcpy(&raw[(stride + 1) * y + 1], rgb + stride * y, stride);
    }
    uLongf packed_size = compressBound(static_cast<uLong>(raw.size()));
    std::vector<uint8_t> packed(packed_size);
    if (compress2(packed.data(), &packed_size, raw.data(), static_cast<uLong>(raw.size()), 6) != Z_OK) {
        rt_set_error("rt_write_png: zlib compression failed");
        return RT_ERR_NOMEM;
    }
    packed.resize(packed_size);

    std::vector<uint8_t> file = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> header;
    put_be32(header, width);
    put_be32(header, height);
    header.push_back(8); // bit depth
    header.push_back(2); // colour type: RGB
    header.push_back(0); header.push_back(0); header.push_back(0);
    put_chunk(file, "IHDR", header);
    put_chunk(file, "IDAT", packed);
    put_chunk(file, "IEND", std::vector<uint8_t>());

    std::FILE* f = std::fopen(path, "wb");
    if (!f) { rt_set_error("rt_write_png: cannot open %s", path); return RT_ERR_IO; }
    size_t written = std::fwrite(file.data(), 1, file.size(), f);
    std::fclose(f);
    if (written != file.size()) { rt_set_error("rt_write_png: short write to %s", path); return RT_ERR_IO; }
    return RT_OK;
}

extern "C" int rt_write_pfm(const char* path, uint32_t width, uint32_t height, const float* rgb) {
    if (!path || !rgb || width == 0 || height == 0) { rt_set_error("rt_write_pfm: bad argument"); return RT_ERR_INVALID; }
    std::FILE* f = std::fopen(path, "wb");
    if (!f) { rt_set_error("rt_write_pfm: cannot open %s", path); return RT_ERR_IO; }
    // "PF" = three channels; a negative scale marks little-endian samples; rows run bottom to top
    bool ok = std::fprintf(f, "PF\n%u %u\n-1.0\n", width, height) > 0;
    const size_t row = size_t(3) * width;
    for (uint32_t y = height; ok && y-- > 0;) ok = std::fwrite(rgb + row * y, sizeof(float), row, f) == row;
    ok = (std::fclose(f) == 0) && ok;
    if (!ok) { rt_set_error("rt_write_pfm: short write to %s", path); return RT_ERR_IO; }
    return RT_OK;
}

// ------------------------------------------------- C view of class Scene

struct rt_host_scene {
    Scene scene;
};

namespace {

template <typename F>
int guarded(const char* what, F&& body) {
    try {
        body();
        return RT_OK;
    } catch (const SceneError& e) { // carries its class: a device call's own status, a parse or file error
        rt_set_error("%s: %s", what, e.what());
        return e.code;
    } catch (const std::invalid_argument& e) { // the reference's "File not found"
        rt_set_error("%s: %s", what, e.what());
        return RT_ERR_IO;
    } catch (const std::bad_alloc&) {
        rt_set_error("%s: out of memory", what);
        return RT_ERR_NOMEM;
    } catch (const std::exception& e) {
        rt_set_error("%s: %s", what, e.what());
        return RT_ERR_INVALID;
    }
}

Vec3f vec3_of(const float* p) { return Vec3f(p[0], p[1], p[2]); }

} // namespace

extern "C" {

int rt_host_scene_create(rt_host_scene** out) {
    if (!out) { rt_set_error("rt_host_scene_create: null output"); return RT_ERR_INVALID; }
    return guarded("rt_host_scene_create", [&] {
        *out = new rt_host_scene();
        const char* v = std::getenv("RT_VERBOSE");
        (*out)->scene.verbose = v && v[0] == '1';
    });
}

int rt_host_scene_destroy(rt_host_scene* s) {
    delete s;
    return RT_OK;
}

int rt_host_scene_load(rt_host_scene* s, const char* json_path) {
    if (!s || !json_path) { rt_set_error("rt_host_scene_load: null argument"); return RT_ERR_INVALID; }
    return guarded("rt_host_scene_load", [&] { s->scene.load_scene(json_path); });
}

int rt_host_scene_set_camera_intrinsics(rt_host_scene* s, float fov, size_t width, size_t height) {
    if (!s || width == 0 || height == 0) { rt_set_error("rt_host_scene_set_camera_intrinsics: bad argument"); return RT_ERR_INVALID; }
    return guarded("rt_host_scene_set_camera_intrinsics", [&] { s->scene.set_camera_intrinsics(fov, width, height); });
}

int rt_host_scene_rotate_camera(rt_host_scene* s, float alpha, float beta, float gamma) {
    if (!s) { rt_set_error("rt_host_scene_rotate_camera: null scene"); return RT_ERR_INVALID; }
    return guarded("rt_host_scene_rotate_camera", [&] { s->scene.rotate_camera(alpha, beta, gamma); });
}

int rt_host_scene_transform_camera(rt_host_scene* s, float x, float y, float z) {
    if (!s) { rt_set_error("rt_host_scene_transform_camera: null scene"); return RT_ERR_INVALID; }
    return guarded("rt_host_scene_transform_camera", [&] { s->scene.transform_camera(x, y, z); });
}

int rt_host_scene_set_num_rays(rt_host_scene* s, uint32_t num_rays) {
    if (!s || num_rays == 0) { rt_set_error("rt_host_scene_set_num_rays: bad argument"); return RT_ERR_INVALID; }
    s->scene.get_camera()->num_rays = num_rays;
    return RT_OK;
}

int rt_host_scene_add_sphere(rt_host_scene* s, const float pos[3], float radius, const float color[3],
                             int32_t type, float ro, float ri) {
    if (!s || !pos || !color || type < 0 || type > 2) { rt_set_error("rt_host_scene_add_sphere: bad argument"); return RT_ERR_INVALID; }
    return guarded("rt_host_scene_add_sphere", [&] {
        s->scene.add_sphere(vec3_of(pos), radius, vec3_of(color), static_cast<material_type>(type), ro, ri);
    });
}

int rt_host_scene_add_object(rt_host_scene* s, const char* obj_path, const float pos[3], const float scale[3],
                             const float rotation[3], const float color[3], int32_t type, float ro, float ri) {
    if (!s || !obj_path || !pos || !scale || !rotation || !color || type < 0 || type > 2) {
        rt_set_error("rt_host_scene_add_object: bad argument");
        return RT_ERR_INVALID;
    }
    return guarded("rt_host_scene_add_object", [&] {
        Material m;
        m.type = static_cast<material_type>(type);
        m.color = vec3_of(color);
        m.ro = ro;
        m.ri = ri;
        s->scene.add_object(obj_path, vec3_of(pos), vec3_of(scale), vec3_of(rotation), m);
    });
}

int rt_host_scene_add_light(rt_host_scene* s, const float pos[3], float intensity) {
    if (!s || !pos) { rt_set_error("rt_host_scene_add_light: bad argument"); return RT_ERR_INVALID; }
    return guarded("rt_host_scene_add_light", [&] { s->scene.add_light(vec3_of(pos), intensity); });
}

int rt_host_scene_set_options(rt_host_scene* s, const rt_options* opt) {
    if (!s || !opt || opt->max_depth < 0) { rt_set_error("rt_host_scene_set_options: bad argument"); return RT_ERR_INVALID; }
    s->scene.settings.indirect_lighting = opt->indirect != 0;
    s->scene.settings.antialiasing = opt->antialiasing != 0;
    s->scene.settings.bounding_box = opt->bounding_box != 0;
    s->scene.settings.max_ray_depth = opt->max_depth;
    s->scene.settings.cluster_culling = opt->cluster_culling != 0;
    s->scene.settings.fast_arithmetic = opt->arithmetic != 0;
    return RT_OK;
}

int rt_host_scene_set_devices(rt_host_scene* s, uint32_t n_devices) {
    if (!s || n_devices == 0 || n_devices > 64) { rt_set_error("rt_host_scene_set_devices: bad argument"); return RT_ERR_INVALID; }
    s->scene.settings.devices = n_devices;
    return RT_OK;
}

int rt_host_scene_render(rt_host_scene* s, const char* png_path, size_t seed) {
    if (!s) { rt_set_error("rt_host_scene_render: null scene"); return RT_ERR_INVALID; }
    return guarded("rt_host_scene_render", [&] { s->scene.render(png_path ? png_path : "", seed); });
}

int rt_host_scene_render_frames(rt_host_scene* s, const char* pattern, size_t first_seed, uint32_t n_frames,
                                const float rotate_step[3], const float move_step[3], uint32_t first, uint32_t stride,
                                uint32_t writers, double* out_seconds) {
    if (!s || stride == 0) { rt_set_error("rt_host_scene_render_frames: null scene or zero stride"); return RT_ERR_INVALID; }
    const Vec3f rot = rotate_step ? vec3_of(rotate_step) : Vec3f();
    const Vec3f mov = move_step ? vec3_of(move_step) : Vec3f();
    int rc = guarded("rt_host_scene_render_frames", [&] {
        const double seconds = s->scene.render_frames(pattern ? pattern : "", first_seed, n_frames, rot, mov, first, stride, writers);
        if (out_seconds) *out_seconds = seconds;
    });
    return rc;
}

int rt_host_scene_describe(rt_host_scene* s, rt_scene_desc* out_desc, rt_camera* out_cam) {
    if (!s) { rt_set_error("rt_host_scene_describe: null scene"); return RT_ERR_INVALID; }
    s->scene.describe(out_desc, out_cam);
    return RT_OK;
}

const float* rt_host_scene_image(rt_host_scene* s, uint32_t* out_width, uint32_t* out_height) {
    if (!s) return nullptr;
    if (out_width) *out_width = s->scene.get_camera()->width;
    if (out_height) *out_height = s->scene.get_camera()->height;
    return s->scene.get_image();
}

} // extern "C"
