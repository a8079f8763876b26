"""The reference's own mesh assets (assets/meshes/*.ply) as pins for Mesh::load_ply / bake_transform / find_mesh_bounds
(core/src/geometry/mesh.rs:48-140) and AccGrid::build_from_mesh (core/src/geometry/acc_grid.rs:6-83).

tests/golden/ref_meshes.json and ref_mesh_<name>.npz were written by tools/gen_ref_fixtures.py from those files with a
third, plain-Python implementation.  Held against them here: the oracle's loader and grid builder, the product's Python and
C++ PLY loaders, and the product's host grid builder.  The loaders need the PLY files themselves and so run only where
/root/reference exists (the build container); everything else runs from the committed arrays.
"""
import ctypes as C
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

from raymond_amd import abi
from raymond_amd.scene import AccGrid, Mesh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
REF_MESHES = "/root/reference/assets/meshes"
with open(os.path.join(GOLD, "ref_meshes.json")) as _f:
    FIX = json.load(_f)
NAMES = sorted(FIX["meshes"])
BAKE = tuple(FIX["bake_translation"])
needs_reference = pytest.mark.skipif(not os.path.isdir(REF_MESHES), reason="the reference checkout is only present in the build container")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def fixture_mesh(name):
    z = np.load(os.path.join(GOLD, "ref_mesh_%s.npz" % name))
    return Mesh(z["tri_pos"], z["tri_nrm"])


def hexes(v):
    return [float(x).hex() for x in v]


def test_fixture_arrays_match_their_digests():
    for name in NAMES:
        m, fx = fixture_mesh(name), FIX["meshes"][name]
        assert len(m) == fx["triangles"]
        assert sha(m.tri_pos) == fx["tri_pos_sha256"] and sha(m.tri_nrm) == fx["tri_nrm_sha256"]
    # what SURVEY.md records about the assets: 12 / 80 / 507 / 967 / 967 faces, the first two with UVs
    assert [FIX["meshes"][n]["header_faces"] for n in ("cube", "ico_sphere", "monkeysmooth", "suzanne", "suzanne_flat")] == [12, 80, 507, 967, 967]
    assert FIX["meshes"]["cube"]["vertex_properties"][-2:] == ["s", "t"] and FIX["meshes"]["suzanne"]["vertex_properties"][-1] == "nz"


@needs_reference
@pytest.mark.parametrize("name", NAMES)
def test_ply_loaders_on_the_reference_assets(name, oracle, product_lib, tmp_path):
    """Mesh::load_ply + bake_transform: oracle, Python mirror and C++ mirror parse the reference's files to the fixture's bits."""
    fx = FIX["meshes"][name]
    path = os.path.join(REF_MESHES, name + ".ply")
    rc, om, (bmin, bmax) = oracle.load_ply(path)
    assert rc == 0 and len(om) == fx["triangles"] == fx["header_faces"]
    assert sha(om.tri_pos) == fx["tri_pos_sha256"] and sha(om.tri_nrm) == fx["tri_nrm_sha256"]
    assert hexes(bmin) == fx["grid"]["bounds_min"] and hexes(bmax) == fx["grid"]["bounds_max"]  # find_mesh_bounds, mesh.rs:123-140
    rc, ob, (bmin, bmax) = oracle.load_ply(path, translate=BAKE)
    assert rc == 0 and sha(ob.tri_pos) == fx["baked_tri_pos_sha256"] and sha(ob.tri_nrm) == fx["tri_nrm_sha256"]
    assert hexes(bmin) == fx["grid_baked"]["bounds_min"] and hexes(bmax) == fx["grid_baked"]["bounds_max"]
    pm = Mesh.load_ply(path)
    assert sha(pm.tri_pos) == fx["tri_pos_sha256"] and sha(pm.tri_nrm) == fx["tri_nrm_sha256"]
    pm.bake_transform(BAKE)
    assert sha(pm.tri_pos) == fx["baked_tri_pos_sha256"]
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "raymond_amd", "host")], check=True)
    out = tmp_path / "m.bin"
    subprocess.run([os.path.join(ROOT, "raymond_amd", "host", "raymond_cli"), "ply", path, str(out)], check=True, capture_output=True)  # loads + bakes (0, -0.3, 2.9)
    raw = np.fromfile(out)
    n = fx["triangles"] * 9
    assert sha(raw[:n]) == fx["baked_tri_pos_sha256"] and sha(raw[n:]) == fx["tri_nrm_sha256"]


def test_oracle_loader_refuses_what_the_reference_panics_on(oracle):
    head = "ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nelement face 1\nend_header\n"
    v = "0 0 0 0 0 1\n1 0 0 0 0 1\n0 1 0 0 0 1\n"
    rc, m, _ = oracle.load_ply(text=head + v + "3 0 1 2\n4 0 1 2 2\n")
    assert rc == 0 and len(m) == 1  # the quad is dropped (:116)
    assert np.array_equal(m.tri_pos[0], [0, 0, 0, 1, 0, 0, 0, 1, 0]) and np.array_equal(m.tri_nrm[0], [0, 0, 1] * 3)
    for bad in (
        head + v + "3 0 1 5\n",          # vertex index out of range (:104)
        head + v + "3 0 1\n",            # values[3] (:99)
        head + v + "\n3 0 1 2\n",        # empty face line: values[0] (:97)
        head + v + "3 0 1 -2\n",         # parse::<u32> (:95)
        head + v[:12] + "3 0 1 2\n",     # fewer vertex lines than declared: the face line is parsed as a vertex with 4 values (:84)
        head.replace("end_header\n", "\nend_header\n") + v,  # empty header line: tokens.next().unwrap() (:68)
        head + "0 0 0 0 0 0x1p0\n" + v[12:],  # hexadecimal float is not f64::from_str syntax
    ):
        assert oracle.load_ply(text=bad)[0] == 1, bad
    assert oracle.load_ply("/nonexistent/file.ply")[0] == 1


@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("baked", [False, True])
def test_grid_builders_on_the_reference_assets(name, baked, oracle, product_lib):
    """AccGrid::build_from_mesh: oracle and the product's host builder give the fixture's tables — or the reference's
    out-of-bounds panic (suzanne.ply: res.z > res.y, so `x + res.x*(y + z*res.z)` leaves the cell array, Q5)."""
    fx = FIX["meshes"][name]["grid_baked" if baked else "grid"]
    mesh = fixture_mesh(name)
    if baked:
        mesh.bake_transform(BAKE)
        assert sha(mesh.tri_pos) == FIX["meshes"][name]["baked_tri_pos_sha256"]
    rc, og = oracle.grid_build(mesh)
    if "panics" in fx:
        assert rc == 5
        h = C.c_void_p()
        st = product_lib.rmd_grid_build_from_mesh(mesh.tri_pos.ctypes.data_as(C.c_void_p), mesh.tri_nrm.ctypes.data_as(C.c_void_p), len(mesh), C.byref(h))
        assert st == abi.RMD_ERR_GRID_INDEX
        return
    assert rc == 0
    pg = AccGrid.build_from_mesh(mesh)
    for g in (og, pg):
        assert [int(v) for v in g.resolution] == fx["resolution"]
        assert hexes(g.bbox_min) == fx["bounds_min"] and hexes(g.bbox_max) == fx["bounds_max"] and hexes(g.cell_size) == fx["cell_size"]
        assert g.cells.size == fx["n_cells"] and g.mapping_table.size == fx["n_mapping"]
        assert sha(g.cells) == fx["cells_sha256"] and sha(g.mapping_table) == fx["mapping_sha256"]
        assert sha(g.tri_pos) == sha(mesh.tri_pos) and sha(g.tri_nrm) == sha(mesh.tri_nrm)


def test_oracle_tonemap_known_answers(oracle):
    """cli_old/src/main.rs:161-181 after await's division (src/trace.rs:95): known answers of the restatement."""
    acc = np.array([[1.5 * 7, 0.0, 1e9], [np.nan, 0.5, 0.5], [-1e-3, 0.2, 0.2], [0.7, 0.7, 0.7]])
    out = oracle.resolve_tonemap(acc, 7)
    assert out[0].tolist() == [227, 0, 255]  # the ceiling of examples/ReflectiveSpheres.png: trunc(255 * (1 - e^-1.5)^(1/2.2)) = 227
    assert out[1].tolist() == [0, 0, 0]      # one NaN channel: cast::<u8>() is None for the whole pixel, it stays (0, 0, 0) (:176-181)
    assert out[2].tolist() == [0, 0, 0]      # negative radiance: (negative).powf(1/2.2) is NaN
    p = 0.7 / 7
    assert out[3].tolist() == [int(255.0 * (1.0 - np.exp(-p)) ** (1 / 2.2))] * 3
