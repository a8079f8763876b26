"""Host grid builder (rmd_grid_build_from_mesh) vs the oracle's AccGrid::build_from_mesh — byte equality.

Reference: core/src/geometry/acc_grid.rs:6-83.
"""
import numpy as np
import pytest

from raymond_amd import abi, lib, scenes
from raymond_amd.scene import AccGrid, Mesh


def _equal(a, b):
    assert np.array_equal(a.resolution, b.resolution)
    assert a.bbox_min.tobytes() == b.bbox_min.tobytes() and a.bbox_max.tobytes() == b.bbox_max.tobytes()
    assert a.cell_size.tobytes() == b.cell_size.tobytes()
    assert a.cells.tobytes() == b.cells.tobytes()
    assert a.mapping_table.tobytes() == b.mapping_table.tobytes()
    assert a.tri_pos.tobytes() == b.tri_pos.tobytes() and a.tri_nrm.tobytes() == b.tri_nrm.tobytes()


@pytest.mark.parametrize("n", [4, 13, 40])
def test_lumpy_mesh_grid_matches_oracle(oracle, product_lib, n):
    mesh = scenes.lumpy_sphere_mesh(n)
    mesh.bake_transform((0.0, -0.3, 2.9))
    rc, og = oracle.grid_build(mesh)
    assert rc == 0
    pg = AccGrid.build_from_mesh(mesh)
    _equal(pg, og)
    # structure: every run is [count, ascending triangle indices]
    for c in range(0, pg.cells.size, max(1, pg.cells.size // 257)):
        off = pg.cells[c]
        cnt = pg.mapping_table[off]
        run = pg.mapping_table[off + 1 : off + 1 + cnt]
        assert np.all(np.diff(run.astype(np.int64)) > 0)


def test_random_soup_grid_matches_oracle(oracle, product_lib):
    rng = np.random.default_rng(7)
    centers = rng.uniform(-1, 1, size=(500, 1, 3)) * np.array([1.0, 0.8, 0.5])
    tri = centers + rng.normal(scale=0.05, size=(500, 3, 3))
    mesh = Mesh(tri.reshape(-1, 9), rng.normal(size=(500, 9)))
    rc, og = oracle.grid_build(mesh)
    assert rc == 0
    _equal(AccGrid.build_from_mesh(mesh), og)


def test_full_size_standin_grid_matches_oracle(oracle, product_lib):
    """The ~100k-triangle GoldDragon stand-in of configs C3-C5."""
    mesh = scenes.lumpy_sphere_mesh(91)
    assert len(mesh) == 99372
    mesh.bake_transform((0.0, -0.3, 2.9))
    rc, og = oracle.grid_build(mesh)
    assert rc == 0
    pg = AccGrid.build_from_mesh(mesh)
    _equal(pg, og)
    assert pg.resolution[2] <= pg.resolution[1]  # otherwise the res.z index quirk (Q5) runs off the array


def test_q5_index_panic_is_an_error_not_a_crash(oracle, product_lib):
    """A mesh deeper (z) than tall (y) makes `x + res.x*(y + z*res.z)` exceed the cell array: the
    reference panics at acc_grid.rs:61; both builders must report it as status 5."""
    mesh = scenes.lumpy_sphere_mesh(6, extent=(2.0, 0.5, 3.0))
    rc, _ = oracle.grid_build(mesh)
    assert rc == abi.RMD_ERR_GRID_INDEX
    with pytest.raises(lib.RaymondError) as e:
        AccGrid.build_from_mesh(mesh)
    assert e.value.status == abi.RMD_ERR_GRID_INDEX


def test_degenerate_inputs_rejected(product_lib):
    import ctypes as C

    h = C.c_void_p()
    assert product_lib.rmd_grid_build_from_mesh(None, None, 0, C.byref(h)) == abi.RMD_ERR_INVALID_ARGUMENT
    flat = Mesh(np.zeros((2, 9)), np.zeros((2, 9)))  # zero volume -> zero resolution
    with pytest.raises(lib.RaymondError):
        AccGrid.build_from_mesh(flat)
