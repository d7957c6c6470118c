#!/usr/bin/env python3
"""Writes a procedural closed mesh (a bumpy sphere) as a Wavefront OBJ with
smooth normals: make_mesh.py out.obj [n_lat] [n_lon] -> 2 * n_lat * n_lon - 2 * n_lon triangles.
Used to exercise mesh sizes the shipped cat model (3 954 triangles) does not reach."""
import sys

import numpy as np


def bumpy_sphere(n_lat, n_lon, radius=12.0, bump=0.12, lobes=5):
    """Vertices, normals (per vertex, from the analytic gradient), triangles (0-based)."""
    theta = np.linspace(0.0, np.pi, n_lat + 1)[1:-1]          # interior latitudes
    phi = np.linspace(0.0, 2 * np.pi, n_lon, endpoint=False)
    t, p = np.meshgrid(theta, phi, indexing="ij")
    r = radius * (1.0 + bump * np.sin(lobes * t) * np.cos(lobes * p))
    ring = np.stack([r * np.sin(t) * np.cos(p), r * np.cos(t), r * np.sin(t) * np.sin(p)], axis=-1).reshape(-1, 3)
    verts = np.concatenate([[[0.0, radius, 0.0]], ring, [[0.0, -radius, 0.0]]]).astype(np.float32)
    tris = []
    n_rings = n_lat - 1
    idx = lambda i, j: 1 + i * n_lon + (j % n_lon)
    for j in range(n_lon):
        tris.append((0, idx(0, j + 1), idx(0, j)))
        tris.append((len(verts) - 1, idx(n_rings - 1, j), idx(n_rings - 1, j + 1)))
    for i in range(n_rings - 1):
        for j in range(n_lon):
            a, b, c, d = idx(i, j), idx(i, j + 1), idx(i + 1, j), idx(i + 1, j + 1)
            tris.append((a, b, c))
            tris.append((b, d, c))
    tris = np.array(tris, np.int32)
    # area-weighted vertex normals
    fn = np.cross(verts[tris[:, 1]] - verts[tris[:, 0]], verts[tris[:, 2]] - verts[tris[:, 0]])
    normals = np.zeros_like(verts)
    for k in range(3):
        np.add.at(normals, tris[:, k], fn)
    normals /= np.maximum(np.linalg.norm(normals, axis=1, keepdims=True), 1e-20)
    return verts, normals.astype(np.float32), tris


def write_obj(path, verts, normals, tris):
    with open(path, "w") as f:
        for v in verts:
            f.write(f"v {v[0]:.6f} {v[1]:.6f} {v[2]:.6f}\n")
        for n in normals:
            f.write(f"vn {n[0]:.6f} {n[1]:.6f} {n[2]:.6f}\n")
        for t in tris + 1:
            f.write(f"f {t[0]}//{t[0]} {t[1]}//{t[1]} {t[2]}//{t[2]}\n")


if __name__ == "__main__":
    n_lat = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    n_lon = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    v, n, t = bumpy_sphere(n_lat, n_lon)
    write_obj(sys.argv[1], v, n, t)
    print(f"{sys.argv[1]}: {len(v)} vertices, {len(t)} triangles")
