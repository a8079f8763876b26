"""Host-side mirror of the reference's scene data model (inputs to the hot path).

Names follow the reference: `Scene{objects}` / `Object{geometry, material}` (core/src/scene.rs:33-45),
`Geometry::{Plane, Sphere, Grid}` (:9-13), `Material::{Diffuse, Metal, Emission}` (core/src/lib.rs:21-26),
`Mesh{triangles, bounding_box}` + `bake_transform` (core/src/geometry/mesh.rs:10-56) and
`AccGrid::build_from_mesh` (core/src/geometry/acc_grid.rs:36-83).  No arithmetic of the per-pixel
path lives here; `flatten()` turns a Scene into the POD arrays of include/raymond_hip.h.
"""
import ctypes as C

import numpy as np

from . import abi


# ---------------------------------------------------------------- Material (core/src/lib.rs:21-26)
class Material:
    def __init__(self, kind, color, roughness=0.0, aux=(0.0, 0.0, 0.0, 0.0, 0.0)):
        self.kind = kind
        self.color = tuple(float(c) for c in color)
        self.roughness = float(roughness)
        self.aux = tuple(float(a) for a in aux)

    @staticmethod
    def Diffuse(color, roughness):
        return Material(abi.RMD_MAT_DIFFUSE, color, roughness)

    @staticmethod
    def Metal(color, roughness):
        return Material(abi.RMD_MAT_METAL, color, roughness)

    @staticmethod
    def Emission(e, v2=(1.0, 1.0, 1.0), f1=0.0, f2=0.0):
        return Material(abi.RMD_MAT_EMISSION, e, 0.0, (v2[0], v2[1], v2[2], f1, f2))


# ---------------------------------------------------------------- Geometry (core/src/scene.rs:9-13)
class Plane:  # core/src/geometry/primitives/plane.rs:5-8
    def __init__(self, origin, normal):
        self.origin = tuple(float(c) for c in origin)
        self.normal = tuple(float(c) for c in normal)


class Sphere:  # core/src/geometry/primitives/sphere.rs:5-8
    def __init__(self, origin, radius):
        self.origin = tuple(float(c) for c in origin)
        self.radius = float(radius)


class Mesh:
    """Triangle soup: tri_pos / tri_nrm are (N, 9) float64 (v0 v1 v2 / n0 n1 n2).  mesh.rs:10-13."""

    def __init__(self, tri_pos, tri_nrm):
        self.tri_pos = np.ascontiguousarray(tri_pos, dtype=np.float64).reshape(-1, 9)
        self.tri_nrm = np.ascontiguousarray(tri_nrm, dtype=np.float64).reshape(-1, 9)
        assert self.tri_pos.shape == self.tri_nrm.shape

    @staticmethod
    def load_ply(path):
        """Mesh::load_ply (mesh.rs:58-121): ASCII PLY; only `element vertex N` is read from the header, vertex lines are
        `x y z nx ny nz [s t]`, face lines `3 i j k`; faces with another vertex count are dropped (:116).  A malformed file
        raises (the reference panics on its unwrap()s)."""
        with open(path) as f:
            lines = f.read().splitlines()  # like Rust's str::lines(): no trailing empty element
        it = iter(lines)
        n_vertices = 0
        for line in it:
            tok = line.split()
            if not tok:
                raise ValueError("load_ply: empty header line")
            if tok[0] == "element" and len(tok) > 2 and tok[1] == "vertex":
                n_vertices = int(tok[2])
            elif tok[0] == "end_header":
                break
        verts = []
        for _ in range(n_vertices):
            v = [float(t) for t in next(it).split()]
            verts.append((v[0:3], v[3:6]))
        pos, nrm = [], []
        for line in it:
            v = [int(t) for t in line.split()]
            if not v:
                raise ValueError("load_ply: empty face line")  # values[0] panics in the reference
            if v[0] != 3:
                continue
            tri = [verts[i] for i in v[1:4]]
            pos.append(sum((t[0] for t in tri), []))
            nrm.append(sum((t[1] for t in tri), []))
        return Mesh(np.array(pos, dtype=np.float64).reshape(-1, 9), np.array(nrm, dtype=np.float64).reshape(-1, 9))

    def bake_transform(self, translate):
        """mesh.rs:48-56: position += translate for every vertex (bounds are recomputed by the grid build)."""
        t = np.asarray(translate, dtype=np.float64)
        self.tri_pos = self.tri_pos + np.tile(t, 3)[None, :]

    def __len__(self):
        return self.tri_pos.shape[0]


class AccGrid:
    """acc_grid.rs:27-33 in the compact layout of rmd_grid_desc (u32 cells / mapping_table)."""

    def __init__(self, bbox_min, bbox_max, resolution, cell_size, cells, mapping_table, tri_pos, tri_nrm):
        self.bbox_min = np.asarray(bbox_min, dtype=np.float64).copy()
        self.bbox_max = np.asarray(bbox_max, dtype=np.float64).copy()
        self.resolution = np.asarray(resolution, dtype=np.uint32).copy()
        self.cell_size = np.asarray(cell_size, dtype=np.float64).copy()
        self.cells = np.ascontiguousarray(cells, dtype=np.uint32)
        self.mapping_table = np.ascontiguousarray(mapping_table, dtype=np.uint32)
        self.tri_pos = np.ascontiguousarray(tri_pos, dtype=np.float64).reshape(-1, 9)
        self.tri_nrm = np.ascontiguousarray(tri_nrm, dtype=np.float64).reshape(-1, 9)

    @staticmethod
    def from_desc(desc):
        """Deep-copies the arrays a (library- or oracle-owned) rmd_grid_desc points at."""
        nt = int(desc.n_tris)
        return AccGrid(
            list(desc.bbox_min),
            list(desc.bbox_max),
            list(desc.resolution),
            list(desc.cell_size),
            np.ctypeslib.as_array(desc.cells, shape=(int(desc.n_cells),)).copy(),
            np.ctypeslib.as_array(desc.mapping_table, shape=(int(desc.n_mapping),)).copy(),
            np.ctypeslib.as_array(desc.tri_pos, shape=(nt * 9,)).copy(),
            np.ctypeslib.as_array(desc.tri_nrm, shape=(nt * 9,)).copy(),
        )

    @staticmethod
    def build_from_mesh(mesh, ctx=None):
        """AccGrid::build_from_mesh through the product's host builder (rmd_grid_build_from_mesh), or on the GPU of
        `ctx` (rmd_grid_build_from_mesh_gpu) — both give byte-identical tables."""
        from . import lib as _lib

        L = _lib.load()
        handle = C.c_void_p()
        pos, nrm = mesh.tri_pos.ctypes.data_as(C.c_void_p), mesh.tri_nrm.ctypes.data_as(C.c_void_p)
        if ctx is None:
            _lib.check(L.rmd_grid_build_from_mesh(pos, nrm, len(mesh), C.byref(handle)))
        else:
            _lib.check(L.rmd_grid_build_from_mesh_gpu(ctx.handle, pos, nrm, len(mesh), C.byref(handle)), ctx.handle)
        try:
            desc = abi.GridDesc()
            _lib.check(L.rmd_grid_build_describe(handle, C.byref(desc)))
            return AccGrid.from_desc(desc)
        finally:
            L.rmd_grid_build_destroy(handle)

    def desc(self):
        d = abi.GridDesc()
        d.bbox_min[:] = self.bbox_min.tolist()
        d.bbox_max[:] = self.bbox_max.tolist()
        d.resolution[:] = [int(v) for v in self.resolution]
        d.cell_size[:] = self.cell_size.tolist()
        d.cells = self.cells.ctypes.data_as(C.POINTER(C.c_uint32))
        d.n_cells = self.cells.size
        d.mapping_table = self.mapping_table.ctypes.data_as(C.POINTER(C.c_uint32))
        d.n_mapping = self.mapping_table.size
        d.tri_pos = self.tri_pos.ctypes.data_as(C.POINTER(C.c_double))
        d.tri_nrm = self.tri_nrm.ctypes.data_as(C.POINTER(C.c_double))
        d.n_tris = self.tri_pos.shape[0]
        return d


class Grid:  # Geometry::Grid(Arc<AccGrid>)
    def __init__(self, acc_grid):
        self.grid = acc_grid


class Object:  # core/src/scene.rs:33-37
    def __init__(self, geometry, material):
        self.geometry = geometry
        self.material = material


class Scene:  # core/src/scene.rs:42-52
    def __init__(self):
        self.objects = []

    def flatten(self):
        """-> (objects: (abi.Object * n), grids: (abi.GridDesc * g), keepalive list).  Object order is kept."""
        grids = []
        objs = (abi.Object * max(1, len(self.objects)))()
        for i, o in enumerate(self.objects):
            r = objs[i]
            g = o.geometry
            if isinstance(g, Plane):
                r.geometry_kind = abi.RMD_GEOM_PLANE
                r.origin[:] = g.origin
                r.normal[:] = g.normal
            elif isinstance(g, Sphere):
                r.geometry_kind = abi.RMD_GEOM_SPHERE
                r.origin[:] = g.origin
                r.radius = g.radius
            elif isinstance(g, Grid):
                r.geometry_kind = abi.RMD_GEOM_GRID
                if g.grid not in grids:
                    grids.append(g.grid)
                r.grid_index = grids.index(g.grid)
            else:
                raise TypeError("unknown geometry %r" % (g,))
            m = o.material
            r.material.kind = m.kind
            r.material.color[:] = m.color
            r.material.roughness = m.roughness
            r.material.emission_aux[:] = m.aux
        descs = (abi.GridDesc * max(1, len(grids)))()
        for i, g in enumerate(grids):
            descs[i] = g.desc()
        return objs, len(self.objects), descs, len(grids), grids


# ---------------------------------------------------------------- Settings (src/trace.rs:32-55)
class Transform:  # src/transform.rs:4-14
    def __init__(self, position=(0.0, 0.0, 0.0)):
        self.position = tuple(float(c) for c in position)

    @staticmethod
    def identity():
        return Transform()


class CameraSettings:
    def __init__(self, backbuffer_width, backbuffer_height, fov_vert, transform=None, focal_length=2.5, aperture_radius=0.0):
        self.backbuffer_width = int(backbuffer_width)
        self.backbuffer_height = int(backbuffer_height)
        self.fov_vert = float(fov_vert)
        self.transform = transform or Transform.identity()
        self.focal_length = float(focal_length)
        self.aperture_radius = float(aperture_radius)

    def pod(self):
        c = abi.Camera()
        c.backbuffer_width = self.backbuffer_width
        c.backbuffer_height = self.backbuffer_height
        c.fov_vert = self.fov_vert
        c.position[:] = self.transform.position
        c.focal_length = self.focal_length
        c.aperture_radius = self.aperture_radius
        return c


class Settings:
    """src/trace.rs:42-55 plus the RNG seed the reference lacks."""

    def __init__(self, camera_settings, sample_count, tile_size=(32, 32), bounce_limit=5, samples_per_iteration=0,
                 worker_count=None, seed=0x5EED0001, use_dof=False, trace_black_paths=False, end_black_paths=False):
        self.camera_settings = camera_settings
        self.sample_count = int(sample_count)
        self.tile_size = (int(tile_size[0]), int(tile_size[1]))
        self.bounce_limit = int(bounce_limit)
        self.samples_per_iteration = int(samples_per_iteration)
        self.worker_count = worker_count  # number of GPUs (contexts) here; None = 1
        self.seed = int(seed)
        # The reference's loop always calls the pinhole generate_primary_ray (src/trace.rs:199) whatever
        # camera_settings.aperture_radius holds; use_dof=True opts into generate_primary_ray_with_dof (:335-360).
        self.use_dof = bool(use_dof)
        # A path whose throughput has become exactly (0, 0, 0) (raymond_hip.h: RMD_RENDER_*_BLACK_PATHS).  Default: reference-identical —
        # ended in scenes without grids (provably the same samples), traced on in scenes with a grid (a later mesh vertex may make the
        # reference's sample 0 x NaN = NaN).  end_black_paths=True ends them in grid scenes too; trace_black_paths=True never ends one.
        self.trace_black_paths = bool(trace_black_paths)
        self.end_black_paths = bool(end_black_paths)

    def pod(self, sample_begin=0, sample_count=None):
        s = abi.Settings()
        s.bounce_limit = self.bounce_limit
        s.sample_begin = int(sample_begin)
        s.sample_count = self.sample_count if sample_count is None else int(sample_count)
        s.seed = self.seed
        s.flags = ((abi.RMD_RENDER_DOF if self.use_dof else 0) | (abi.RMD_RENDER_TRACE_BLACK_PATHS if self.trace_black_paths else 0)
                   | (abi.RMD_RENDER_END_BLACK_PATHS if self.end_black_paths else 0))
        return s


def generate_tiles(width, height, tile_size):
    """Tile generation order of render_tiled (src/trace.rs:142-173): column-major, edge tiles clamped."""
    tw, th = tile_size
    tiles = []
    x = y = 0
    while True:
        x1 = min(x + tw, width)
        y1 = min(y + th, height)
        tiles.append((x, y, x1 - x, y1 - y))
        y += th
        if y >= height:
            y = 0
            x += tw
        if x >= width:
            break
    return tiles


def tile_array(tiles):
    arr = (abi.TileRect * max(1, len(tiles)))()
    for i, (l, t, w, h) in enumerate(tiles):
        arr[i].left, arr[i].top, arr[i].width, arr[i].height = l, t, w, h
    return arr
