// obj_mini.hpp -- Wavefront OBJ reader for the subset Scene::add_object uses.
//
// The reference reads meshes with tinyobj::ObjReader in its default
// configuration (reference src/scene.cu:289-303): positions, normals and
// texture coordinates, faces triangulated, one shape per `g` / `o` group.
// This reader yields the same view: attribute arrays plus, per shape, a flat
// list of triangle corners.  Polygons with more than three corners are split
// as a fan around their first corner; the reference's tinyobjloader version is
// not pinned, so its choice of diagonal for quads is not known.
#pragma once

#include <cstdlib>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace rt {
namespace obj {

struct Corner {
    int vertex = -1;   // index into positions (0-based), always valid
    int texcoord = -1; // -1 when the face gives none
    int normal = -1;   // -1 when the face gives none
};

struct Shape {
    std::string name;
    std::vector<Corner> corners; // 3 per triangle
};

struct Mesh {
    std::vector<float> positions; // xyz
    std::vector<float> normals;   // xyz
    std::vector<float> texcoords; // uv
    std::vector<Shape> shapes;
};

namespace detail {

inline const char* skip_blank(const char* p) {
    while (*p == ' ' || *p == '\t') p++;
    return p;
}

inline bool read_float(const char*& p, float& out) {
    p = skip_blank(p);
    char* end = nullptr;
    double v = std::strtod(p, &end);
    if (end == p) return false;
    out = static_cast<float>(v);
    p = end;
    return true;
}

// OBJ indices are 1-based; negative ones count back from the end.
inline int resolve(long idx, size_t count) {
    if (idx > 0) return static_cast<int>(idx - 1);
    if (idx < 0) return static_cast<int>(static_cast<long>(count) + idx);
    return -1;
}

inline bool read_corner(const char*& p, Corner& c, size_t nv, size_t nt, size_t nn) {
    p = skip_blank(p);
    if (*p == '\0' || *p == '\r' || *p == '\n' || *p == '#') return false;
    char* end = nullptr;
    long v = std::strtol(p, &end, 10);
    if (end == p) return false;
    c = Corner();
    c.vertex = resolve(v, nv);
    p = end;
    if (*p == '/') {
        p++;
        if (*p != '/') {
            long t = std::strtol(p, &end, 10);
            if (end != p) { c.texcoord = resolve(t, nt); p = end; }
        }
        if (*p == '/') {
            p++;
            long n = std::strtol(p, &end, 10);
            if (end != p) { c.normal = resolve(n, nn); p = end; }
        }
    }
    return true;
}

} // namespace detail

// Throws std::runtime_error when the file cannot be opened or a face refers
// to a vertex that does not exist.
inline Mesh load(const std::string& path) {
    std::ifstream in(path);
    if (!in.is_open()) throw std::runtime_error("Cannot open file [" + path + "]");

    Mesh mesh;
    Shape current;
    auto flush = [&]() {
        if (!current.corners.empty()) mesh.shapes.push_back(current);
        current.corners.clear();
    };

    std::string line;
    std::vector<Corner> polygon;
    while (std::getline(in, line)) {
        const char* p = detail::skip_blank(line.c_str());
        if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
            p += 2;
            float x, y, z;
            if (detail::read_float(p, x) && detail::read_float(p, y) && detail::read_float(p, z)) {
                mesh.positions.push_back(x); mesh.positions.push_back(y); mesh.positions.push_back(z);
            }
        } else if (p[0] == 'v' && p[1] == 'n' && (p[2] == ' ' || p[2] == '\t')) {
            p += 3;
            float x, y, z;
            if (detail::read_float(p, x) && detail::read_float(p, y) && detail::read_float(p, z)) {
                mesh.normals.push_back(x); mesh.normals.push_back(y); mesh.normals.push_back(z);
            }
        } else if (p[0] == 'v' && p[1] == 't' && (p[2] == ' ' || p[2] == '\t')) {
            p += 3;
            float u = 0.f, v = 0.f;
            detail::read_float(p, u);
            detail::read_float(p, v);
            mesh.texcoords.push_back(u); mesh.texcoords.push_back(v);
        } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
            p += 2;
            polygon.clear();
            Corner c;
            while (detail::read_corner(p, c, mesh.positions.size() / 3, mesh.texcoords.size() / 2, mesh.normals.size() / 3)) {
                if (c.vertex < 0 || static_cast<size_t>(c.vertex) >= mesh.positions.size() / 3)
                    throw std::runtime_error("face refers to a missing vertex in [" + path + "]");
                if (c.normal >= 0 && static_cast<size_t>(c.normal) >= mesh.normals.size() / 3)
                    throw std::runtime_error("face refers to a missing normal in [" + path + "]");
                polygon.push_back(c);
            }
            for (size_t k = 1; k + 1 < polygon.size(); k++) {
                current.corners.push_back(polygon[0]);
                current.corners.push_back(polygon[k]);
                current.corners.push_back(polygon[k + 1]);
            }
        } else if ((p[0] == 'g' || p[0] == 'o') && (p[1] == ' ' || p[1] == '\t' || p[1] == '\0' || p[1] == '\r')) {
            flush();
            const char* name = detail::skip_blank(p + 1);
            current.name.assign(name);
            while (!current.name.empty() && (current.name.back() == '\r' || current.name.back() == ' ')) current.name.pop_back();
        }
        // mtllib, usemtl, s and comments carry nothing the renderer uses
    }
    flush();
    return mesh;
}

} // namespace obj
} // namespace rt
