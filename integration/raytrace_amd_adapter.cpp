// raytrace_amd_adapter.cpp -- compiled with the REFERENCE's headers
//
// Drop-in definition of the function the reference's Scene::render calls
// (declared reference src/scene.cu:21-35, defined src/raytrace.cu:509-644):
// converts the reference's arrays of structs to rt_scene_desc and renders
// through librt_amd.so.  Build it in place of src/raytrace.cu.
#include <math.h>                // the reference's headers use memset / sqrt / abs without including
#include <stdlib.h>              // anything themselves (nvcc pulls the CUDA runtime headers in implicitly)
#include <string.h>
#include <cstdio>
#include <vector>

#include "scene_objects.cuh"     // reference types: Sphere, Triangle, CU_Vector3f (16 bytes), ...
#include "rt_amd.h"

void raytrace_spheres(Sphere* spheres, size_t sphere_count, Triangle* triangles, size_t triangle_count,
                      CU_Vector3f* vertices, CU_Vector3f* normals, BoundingBox* bounding_boxes,
                      size_t vertex_count, size_t obj_count, Light* lights, size_t light_count,
                      CU_Vector3f* image, Camera* camera, size_t seed)
{
    auto put3 = [](std::vector<float>& v, const CU_Vector3f& a) { v.push_back(a[0]); v.push_back(a[1]); v.push_back(a[2]); };
    std::vector<float> sp, sr, mc, mro, mri, vv, vn, bmin, bmax, lp, li;
    std::vector<int32_t> sm, mt, tv, tn, tm;
    auto add_material = [&](const Material& m) {
        mt.push_back(m.type); put3(mc, m.color); mro.push_back(m.ro); mri.push_back(m.ri);
        return (int32_t)mt.size() - 1;
    };
    for (size_t i = 0; i < sphere_count; i++) { put3(sp, spheres[i].pos); sr.push_back(spheres[i].radius); sm.push_back(add_material(spheres[i].material)); }
    for (size_t i = 0; i < vertex_count; i++) { put3(vv, vertices[i]); put3(vn, normals[i]); }
    for (size_t i = 0; i < triangle_count; i++) {
        for (int k = 0; k < 3; k++) { tv.push_back(triangles[i].v[k]); tn.push_back(triangles[i].n[k]); }
        tm.push_back(add_material(triangles[i].material));
    }
    for (size_t i = 0; i < obj_count; i++) { put3(bmin, bounding_boxes[i]._min); put3(bmax, bounding_boxes[i]._max); }
    for (size_t i = 0; i < light_count; i++) { put3(lp, lights[i].pos); li.push_back(lights[i].I); }

    rt_scene_desc d = {};
    d.n_spheres = sphere_count;  d.sphere_pos = sp.data(); d.sphere_radius = sr.data(); d.sphere_material = sm.data();
    d.n_materials = mt.size();   d.material_type = mt.data(); d.material_color = mc.data(); d.material_ro = mro.data(); d.material_ri = mri.data();
    d.n_vertices = vertex_count; d.vertices = vv.data(); d.normals = vn.data();
    d.n_triangles = triangle_count; d.tri_v = tv.data(); d.tri_n = tn.data(); d.tri_material = tm.data();
    d.n_boxes = obj_count;       d.box_min = bmin.data(); d.box_max = bmax.data();
    d.n_lights = light_count;    d.light_pos = lp.data(); d.light_intensity = li.data();

    rt_camera c = {};
    CU_Matrix<3> R = camera->E.get_rotation();          // src/raytrace.cu:538
    CU_Vector3f  t = camera->E.get_translation();       // src/raytrace.cu:539
    for (int i = 0; i < 9; i++) c.rot[i] = R[i];
    for (int i = 0; i < 3; i++) c.pos[i] = t[i];
    c.fx = camera->K[0]; c.fy = camera->K[4]; c.cx = camera->K[2]; c.cy = camera->K[5];
    c.width = camera->width; c.height = camera->height; c.num_rays = camera->num_rays;

    rt_options o = {};                                   // the three #defines + MAX_RAY_DEPTH, src/raytrace.cu:12-16
    o.indirect = 1; o.antialiasing = 1; o.bounding_box = 1; o.max_depth = 5;
    o.seed = (uint32_t)seed;
    o.cluster_culling = 1;                               // 0 = test every triangle, as the reference does

    rt_scene* scene = nullptr;
    std::vector<float> rgb(3 * (size_t)c.width * c.height);
    if (rt_scene_upload(&d, &scene) || rt_render(scene, &c, &o, rgb.data(), nullptr)) {
        std::fprintf(stderr, "rt_amd: %s\n", rt_last_error());
    } else {
        for (size_t i = 0; i < rgb.size() / 3; i++) image[i] = CU_Vector3f(rgb[3*i], rgb[3*i+1], rgb[3*i+2]);
    }
    rt_scene_free(scene);
}
