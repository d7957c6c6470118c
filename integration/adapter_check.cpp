// adapter_check.cpp -- exercises raytrace_amd_adapter.cpp with the REFERENCE's own types.
//
// Built only where the reference's headers are available (integration/Makefile puts the binary
// under oracle/_ref/, which is not kept in history).  It lays a small scene out in the reference's
// arrays of structs (16-byte CU_Vector3f, Material copies in every Sphere and Triangle; reference
// include/scene_objects.cuh), fills struct Camera the way the reference's Scene does
// (src/scene.cu:86-87,224-262), calls raytrace_spheres() -- the adapter -- and then renders the
// same numbers through rt_scene_desc / rt_render directly.  Both images must agree bit for bit.
// Writes the adapter's image as raw floats to argv[1] when given, so that a test can compare it
// with the CPU oracle.
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <cstdio>
#include <vector>

#include "scene_objects.cuh"
#include "rt_amd.h"

void raytrace_spheres(Sphere* spheres, size_t sphere_count, Triangle* triangles, size_t triangle_count,
                      CU_Vector3f* vertices, CU_Vector3f* normals, BoundingBox* bounding_boxes,
                      size_t vertex_count, size_t obj_count, Light* lights, size_t light_count,
                      CU_Vector3f* image, Camera* camera, size_t seed);

static Material material(material_type type, float r, float g, float b, float ro = 0.f, float ri = 0.f) {
    Material m;
    m.type = type;
    m.color = CU_Vector3f(r, g, b);
    m.ro = ro;
    m.ri = ri;
    return m;
}

int main(int argc, char** argv) {
    const unsigned W = 96, H = 64, SPP = 4;
    // room of six large spheres plus one mirror and one glass ball
    struct { float x, y, z, R; Material m; } balls[] = {
        {0, 0, -1000, 940, material(DIFFUSE, 0, 1, 0)},      {0, -1000, 0, 990, material(DIFFUSE, 0, 0, 1)},
        {0, 0, 1000, 940, material(DIFFUSE, 1, 0, 1)},       {0, 1000, 0, 940, material(DIFFUSE, 1, 0, 0)},
        {-1000, 0, 0, 940, material(DIFFUSE, 0.25f, 1, 1)},  {1000, 0, 0, 940, material(DIFFUSE, 1, 1, 0.25f)},
        {-22, 0, 0, 9, material(MIRROR, 0, 0, 0)},           {22, 0, 5, 9, material(GLASS, 0, 0, 0, 1.f, 1.5f)},
    };
    std::vector<Sphere> spheres;
    for (auto& b : balls) {
        Sphere s;
        s.pos = CU_Vector3f(b.x, b.y, b.z);
        s.radius = b.R;
        s.material = b.m;
        spheres.push_back(s);
    }
    // a tetrahedron, de-indexed like the reference's loader does (three fresh vertices per face)
    const float P[4][3] = {{0, 12, 0}, {-10, -8, 8}, {10, -8, 8}, {0, -8, -10}};
    const int F[4][3] = {{0, 1, 2}, {0, 2, 3}, {0, 3, 1}, {1, 3, 2}};
    std::vector<CU_Vector3f> vertices, normals;
    std::vector<Triangle> triangles;
    BoundingBox box;
    box._min = CU_Vector3f(1e30f, 1e30f, 1e30f);
    box._max = CU_Vector3f(-1e30f, -1e30f, -1e30f);
    for (int f = 0; f < 4; f++) {
        float e1[3], e2[3], n[3];
        for (int k = 0; k < 3; k++) { e1[k] = P[F[f][1]][k] - P[F[f][0]][k]; e2[k] = P[F[f][2]][k] - P[F[f][0]][k]; }
        n[0] = e1[1] * e2[2] - e1[2] * e2[1]; n[1] = e1[2] * e2[0] - e1[0] * e2[2]; n[2] = e1[0] * e2[1] - e1[1] * e2[0];
        const float len = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        Triangle t;
        for (int k = 0; k < 3; k++) {
            const float* p = P[F[f][k]];
            vertices.push_back(CU_Vector3f(p[0], p[1], p[2]));
            normals.push_back(CU_Vector3f(n[0] / len, n[1] / len, n[2] / len));
            t.v[k] = t.n[k] = 3 * f + k;
            for (int a = 0; a < 3; a++) {
                if (p[a] < box._min[a]) box._min[a] = p[a];
                if (p[a] > box._max[a]) box._max[a] = p[a];
            }
        }
        t.group = 0;
        t.material = material(DIFFUSE, 1, 1, 1);
        triangles.push_back(t);
    }
    Light light;
    light.pos = CU_Vector3f(-10, 20, 40);
    light.I = 1e10f;

    Camera cam;
    cam.E.set_identity();
    cam.E(1, 1) = -1.f;
    cam.E(2, 2) = -1.f;
    cam.E(2, 3) = 55.f;
    const float fl = W / (2 * tanf(M_PI * 60.f / (2 * 180)));
    cam.K.set_identity();
    cam.K(0, 0) = fl; cam.K(1, 1) = fl; cam.K(0, 2) = W / 2; cam.K(1, 2) = H / 2;
    cam.width = W; cam.height = H; cam.num_rays = SPP;

    std::vector<CU_Vector3f> image(W * H);
    raytrace_spheres(spheres.data(), spheres.size(), triangles.data(), triangles.size(), vertices.data(), normals.data(), &box,
                     vertices.size(), 1, &light, 1, image.data(), &cam, 7);

    // the same scene straight through the C ABI
    std::vector<float> sp, sr, mc, mro, mri, vv, vn, bmin, bmax;
    std::vector<int32_t> sm, mt, tv, tn, tm;
    auto mat = [&](const Material& m) { mt.push_back(m.type); for (int k = 0; k < 3; k++) mc.push_back(m.color[k]); mro.push_back(m.ro); mri.push_back(m.ri); return (int32_t)mt.size() - 1; };
    for (auto& b : balls) { sp.push_back(b.x); sp.push_back(b.y); sp.push_back(b.z); sr.push_back(b.R); sm.push_back(mat(b.m)); }
    for (size_t i = 0; i < vertices.size(); i++) for (int k = 0; k < 3; k++) { vv.push_back(vertices[i][k]); vn.push_back(normals[i][k]); }
    for (auto& t : triangles) { for (int k = 0; k < 3; k++) { tv.push_back(t.v[k]); tn.push_back(t.n[k]); } tm.push_back(mat(t.material)); }
    for (int k = 0; k < 3; k++) { bmin.push_back(box._min[k]); bmax.push_back(box._max[k]); }
    const float lp[3] = {-10, 20, 40}, li[1] = {1e10f};
    rt_scene_desc d = {};
    d.n_spheres = (uint32_t)sr.size(); d.sphere_pos = sp.data(); d.sphere_radius = sr.data(); d.sphere_material = sm.data();
    d.n_materials = (uint32_t)mt.size(); d.material_type = mt.data(); d.material_color = mc.data(); d.material_ro = mro.data(); d.material_ri = mri.data();
    d.n_vertices = (uint32_t)vertices.size(); d.vertices = vv.data(); d.normals = vn.data();
    d.n_triangles = (uint32_t)triangles.size(); d.tri_v = tv.data(); d.tri_n = tn.data(); d.tri_material = tm.data();
    d.n_boxes = 1; d.box_min = bmin.data(); d.box_max = bmax.data();
    d.n_lights = 1; d.light_pos = lp; d.light_intensity = li;
    rt_camera c = {};
    const float rot[9] = {1, 0, 0, 0, -1, 0, 0, 0, -1};
    memcpy(c.rot, rot, sizeof rot);
    c.pos[2] = 55.f;
    c.fx = c.fy = fl; c.cx = W / 2; c.cy = H / 2; c.width = W; c.height = H; c.num_rays = SPP;
    rt_options o = {};
    o.indirect = o.antialiasing = o.bounding_box = o.cluster_culling = 1;
    o.max_depth = 5;
    o.seed = 7;
    rt_scene* scene = nullptr;
    std::vector<float> direct(3 * W * H);
    if (rt_scene_upload(&d, &scene) != RT_OK || rt_render(scene, &c, &o, direct.data(), nullptr) != RT_OK) {
        std::fprintf(stderr, "adapter_check: %s\n", rt_last_error());
        return 2;
    }
    rt_scene_free(scene);

    size_t differing = 0, lit = 0;
    std::vector<float> flat(3 * W * H);
    for (size_t i = 0; i < (size_t)W * H; i++)
        for (int k = 0; k < 3; k++) {
            flat[3 * i + k] = image[i][k];
            if (memcmp(&flat[3 * i + k], &direct[3 * i + k], 4) != 0) differing++;
            if (flat[3 * i + k] > 0) lit++;
        }
    if (argc > 1) {
        FILE* f = fopen(argv[1], "wb");
        if (!f || fwrite(flat.data(), 4, flat.size(), f) != flat.size()) { std::fprintf(stderr, "adapter_check: cannot write %s\n", argv[1]); return 2; }
        fclose(f);
    }
    std::printf("adapter_check: %ux%u, %zu values lit, %zu differ from rt_render; sizeof(CU_Vector3f)=%zu sizeof(Sphere)=%zu sizeof(Triangle)=%zu\n",
                W, H, lit, differing, sizeof(CU_Vector3f), sizeof(Sphere), sizeof(Triangle));
    return differing == 0 && lit > W * H ? 0 : 1;
}
