#!/usr/bin/env python3
"""Fixtures from the reference's OWN mesh assets (the only real inputs it ships for core/src/geometry/mesh.rs:58-140
and acc_grid.rs:36-83).  Run in the build container, where /root/reference exists:

    python tools/gen_ref_fixtures.py

Reads /root/reference/assets/meshes/{cube,ico_sphere,monkeysmooth,suzanne,suzanne_flat}.ply AS DATA and writes

  tests/golden/ref_meshes.json        per mesh: the counts its header declares, triangle count, sha256 of the parsed
                                      (N, 9) f64 position / normal arrays, bounds (hex floats), and for the mesh as loaded
                                      and as baked by cli_old's translation (0, -0.3, 2.9): grid resolution, table sizes,
                                      sha256 of cells / mapping_table — or "panics" where the reference's index
                                      `x + res.x*(y + z*res.z)` (acc_grid.rs:61, Q5) runs past the cell array
  tests/golden/ref_mesh_<name>.npz    the parsed arrays (tri_pos, tri_nrm), so that the GPU box — which has no
                                      /root/reference — can build scenes from the reference's meshes

Everything here is computed by a THIRD implementation (plain Python / numpy below), independent of the two PLY loaders
(raymond_amd/scene.py, raymond_amd/host/raymond.cpp), of the product's grid builders and of the oracle; the tests hold all
of them against these values.  No PLY text is stored in the repo.
"""
import hashlib
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
SRC = "/root/reference/assets/meshes"
NAMES = ["cube", "ico_sphere", "monkeysmooth", "suzanne", "suzanne_flat"]
BAKE = (0.0, -0.3, 2.9)  # cli_old/src/main.rs:61


def parse_ply(path):
    """Header counts + vertex table + triangle index table, straight from the file's lines."""
    with open(path) as f:
        lines = f.read().split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    declared = {}
    props = []
    i = 0
    while True:
        tok = lines[i].split()
        i += 1
        if tok[0] == "element":
            declared[tok[1]] = int(tok[2])
            current = tok[1]
        elif tok[0] == "property" and current == "vertex":
            props.append(tok[-1])
        elif tok[0] == "end_header":
            break
    nv = declared["vertex"]
    verts = np.array([[float(t) for t in lines[i + k].split()] for k in range(nv)], dtype=np.float64)
    faces = [[int(t) for t in l.split()] for l in lines[i + nv :]]
    tris = np.array([f[1:4] for f in faces if f[0] == 3], dtype=np.int64)
    return declared, props, verts, tris, len(faces)


def trunc_usize(v):
    """f64 -> usize as num-traits NumCast does it: truncate toward zero, None outside (-1, 2^64) or NaN."""
    if not (v > -1.0 and v < 18446744073709551616.0):
        return None
    return int(v)


def build_grid(tri_pos):
    """AccGrid::build_from_mesh (acc_grid.rs:6-83) in plain Python; returns a dict or {"panics": reason}."""
    n = tri_pos.shape[0]
    pts = tri_pos.reshape(-1, 3)
    seed_min, seed_max = np.array([125125.0, 1251251.0, 12512512.0]), np.array([-123125.0, -125123.0, -512123.0])  # mesh.rs:124-125
    bmin, bmax = np.minimum(seed_min, pts.min(axis=0)), np.maximum(seed_max, pts.max(axis=0))
    size = bmax - bmin
    volume = abs(size[0] * size[1] * size[2])
    density = math.pow((3.0 * float(n)) / volume, 1.0 / 3.0)
    res = [int(abs(size[a]) * density) for a in range(3)]
    out = {"bounds_min": [float(v).hex() for v in bmin], "bounds_max": [float(v).hex() for v in bmax], "resolution": res}
    if min(res) == 0:
        out["panics"] = "zero resolution"
        return out
    cell = [size[a] / float(res[a]) for a in range(3)]
    out["cell_size"] = [float(v).hex() for v in cell]
    n_cells = res[0] * res[1] * res[2]
    naive = [[] for _ in range(n_cells)]
    for index in range(n):
        p = tri_pos[index].reshape(3, 3)
        tmin = np.minimum(seed_min, p.min(axis=0))  # triangle.rs:70-84 uses the same seeds
        tmax = np.maximum(seed_max, p.max(axis=0))
        cmin, cmax = [], []
        for a in range(3):
            lo, hi = trunc_usize((tmin[a] - bmin[a]) / cell[a]), trunc_usize((tmax[a] - bmin[a]) / cell[a])
            if lo is None or hi is None:
                out["panics"] = "cast"
                return out
            cmin.append(min(max(lo, 0), res[a] - 1))
            cmax.append(min(max(hi, 0), res[a] - 1))
        for z in range(cmin[2], cmax[2] + 1):
            for y in range(cmin[1], cmax[1] + 1):
                for x in range(cmin[0], cmax[0] + 1):
                    idx = x + res[0] * (y + z * res[2])  # Q5: res.z where res.y is meant
                    if idx >= n_cells:
                        out["panics"] = "cell index %d >= %d" % (idx, n_cells)
                        return out
                    naive[idx].append(index)
    cells, mapping = [], []
    for c in naive:
        cells.append(len(mapping))
        mapping.append(len(c))
        mapping.extend(c)
    cells, mapping = np.asarray(cells, dtype=np.uint32), np.asarray(mapping, dtype=np.uint32)
    out.update(n_cells=n_cells, n_mapping=int(mapping.size), non_empty_cells=int(sum(1 for c in naive if c)),
               cells_sha256=hashlib.sha256(cells.tobytes()).hexdigest(), mapping_sha256=hashlib.sha256(mapping.tobytes()).hexdigest())
    return out


def main():
    if not os.path.isdir(SRC):
        sys.exit("%s is not present (this script runs in the build container only)" % SRC)
    out = {}
    for name in NAMES:
        declared, props, verts, tris, n_face_lines = parse_ply(os.path.join(SRC, name + ".ply"))
        assert verts.shape == (declared["vertex"], len(props)) and n_face_lines == declared["face"]
        pos = verts[:, 0:3][tris].reshape(-1, 9)  # Triangle(v[i], v[j], v[k]) (mesh.rs:110-114)
        nrm = verts[:, 3:6][tris].reshape(-1, 9)
        baked = pos + np.tile(np.asarray(BAKE), 3)[None, :]  # Mesh::bake_transform (mesh.rs:48-56)
        out[name] = {
            "header_vertices": declared["vertex"], "header_faces": declared["face"], "vertex_properties": props,
            "triangles": int(pos.shape[0]),
            "tri_pos_sha256": hashlib.sha256(pos.tobytes()).hexdigest(), "tri_nrm_sha256": hashlib.sha256(nrm.tobytes()).hexdigest(),
            "baked_tri_pos_sha256": hashlib.sha256(baked.tobytes()).hexdigest(),
            "grid": build_grid(pos), "grid_baked": build_grid(baked),
        }
        np.savez_compressed(os.path.join(GOLD, "ref_mesh_%s.npz" % name), tri_pos=pos, tri_nrm=nrm)
        g = out[name]["grid_baked"]
        print(name, out[name]["triangles"], "triangles; baked grid", g.get("resolution"), g.get("panics", "ok"), flush=True)
    with open(os.path.join(GOLD, "ref_meshes.json"), "w") as f:
        json.dump({"_source": "/root/reference/assets/meshes/*.ply read as data by tools/gen_ref_fixtures.py", "bake_translation": list(BAKE), "meshes": out}, f, indent=1)


if __name__ == "__main__":
    main()
