// check_raytrace_spheres.cpp -- calls raytrace_spheres() (scene.cpp), the function with the
// signature of the reference's launcher (reference src/scene.cu:21-35, src/raytrace.cu:509-522),
// with a scene in arrays of structs, and compares its image with the one Scene::render draws from
// the same scene file and seed.  usage: check_raytrace_spheres scene.json [width height rays]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <vector>

#include "scene.hpp"

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: check_raytrace_spheres scene.json [width height rays]\n"); return 2; }
    try {
        Scene s;
        s.verbose = false;
        s.load_scene(argv[1]);
        if (argc >= 5) {
            s.set_camera_intrinsics(60.f, std::strtoul(argv[2], nullptr, 10), std::strtoul(argv[3], nullptr, 10));
            s.get_camera()->num_rays = static_cast<unsigned>(std::strtoul(argv[4], nullptr, 10));
        }
        const size_t seed = 9;
        s.render("", seed);
        Camera* cam = s.get_camera();
        const size_t n_pixels = size_t(cam->width) * cam->height;
        std::vector<float> want(s.get_image(), s.get_image() + 3 * n_pixels);

        // the flat description back into arrays of structs, one Material copy per sphere and triangle
        rt_scene_desc d;
        s.describe(&d, nullptr);
        auto material = [&](int32_t m) {
            Material out;
            out.type = static_cast<material_type>(d.material_type[m]);
            out.color = Vec3f(d.material_color[3 * m], d.material_color[3 * m + 1], d.material_color[3 * m + 2]);
            out.ro = d.material_ro[m];
            out.ri = d.material_ri[m];
            return out;
        };
        std::vector<Sphere> spheres(d.n_spheres);
        for (uint32_t i = 0; i < d.n_spheres; i++) {
            spheres[i].pos = Vec3f(d.sphere_pos[3 * i], d.sphere_pos[3 * i + 1], d.sphere_pos[3 * i + 2]);
            spheres[i].radius = d.sphere_radius[i];
            spheres[i].material = material(d.sphere_material[i]);
        }
        std::vector<Triangle> triangles(d.n_triangles);
        for (uint32_t i = 0; i < d.n_triangles; i++) {
            for (int k = 0; k < 3; k++) { triangles[i].v[k] = d.tri_v[3 * i + k]; triangles[i].n[k] = d.tri_n[3 * i + k]; }
            triangles[i].material = material(d.tri_material[i]);
        }
        std::vector<Vec3f> vertices(d.n_vertices), normals(d.n_vertices);
        for (uint32_t i = 0; i < d.n_vertices; i++) {
            vertices[i] = Vec3f(d.vertices[3 * i], d.vertices[3 * i + 1], d.vertices[3 * i + 2]);
            normals[i] = Vec3f(d.normals[3 * i], d.normals[3 * i + 1], d.normals[3 * i + 2]);
        }
        std::vector<BoundingBox> boxes = s.get_bounding_boxes();
        std::vector<Light> lights(d.n_lights);
        for (uint32_t i = 0; i < d.n_lights; i++) {
            lights[i].pos = Vec3f(d.light_pos[3 * i], d.light_pos[3 * i + 1], d.light_pos[3 * i + 2]);
            lights[i].I = d.light_intensity[i];
        }
        std::vector<Vec3f> image(n_pixels);
        const int rc = raytrace_spheres(spheres.data(), spheres.size(), triangles.data(), triangles.size(), vertices.data(), normals.data(),
                                        boxes.data(), vertices.size(), boxes.size(), lights.data(), lights.size(), image.data(), cam, seed,
                                        &s.settings);
        if (rc != RT_OK) { std::fprintf(stderr, "raytrace_spheres: %d %s\n", rc, rt_last_error()); return 2; }
        size_t differing = 0;
        for (size_t i = 0; i < n_pixels; i++)
            for (int k = 0; k < 3; k++)
                if (std::memcmp(&image[i].v[k], &want[3 * i + k], 4) != 0) differing++;
        std::printf("check_raytrace_spheres: %ux%u x %u rays, %zu spheres, %zu triangles: %zu of %zu values differ from Scene::render\n",
                    cam->width, cam->height, cam->num_rays, spheres.size(), triangles.size(), differing, 3 * n_pixels);
        return differing == 0 ? 0 : 1;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 2;
    }
}
