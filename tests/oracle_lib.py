"""ctypes loader for oracle/liboracle.so — the CPU restatement of the reference path.

Test infrastructure: imported only from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Never imported by raymond_amd.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from raymond_amd import abi

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(_ROOT, "oracle")
ORACLE_PATH = os.path.join(ORACLE_DIR, "liboracle.so")
ORACLE_FAST_PATH = os.path.join(ORACLE_DIR, "liboracle_fast.so")  # the same source without its work counters, -O3: bench.py's cpu_baseline

_P = C.POINTER
_vp = C.c_void_p
_sz = C.c_size_t
_SIGS = {
    "orc_philox4x32_10": (None, [_vp, _vp, _vp]),
    "orc_block_uniforms": (None, [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, _vp]),
    "orc_sphere_intersect": (None, [_sz, _vp, _vp, _vp, _vp]),
    "orc_sphere_normal": (None, [_sz, _vp, _vp, _vp, _vp]),
    "orc_plane_intersect": (None, [_sz, _vp, _vp, _vp, _vp]),
    "orc_aabb_intersect": (None, [_sz, _vp, _vp, _vp, _vp]),
    "orc_triangle_intersect": (None, [_sz, _vp, _vp, _vp, _vp]),
    "orc_triangle_normal": (None, [_sz, _vp, _vp, _vp, _vp, _vp]),
    "orc_onb": (None, [_sz, _vp, _vp, _vp]),
    "orc_cosine_hemisphere": (None, [_sz, _vp, _vp, _vp, _vp]),
    "orc_importance_sample_ggx": (None, [_sz, _vp, _vp, _vp, _vp, _vp]),
    "orc_ggx_distribution": (None, [_sz, _vp, _vp, _vp, _vp]),
    "orc_geometry_smith": (None, [_sz, _vp, _vp, _vp, _vp, _vp]),
    "orc_fresnel_schlick": (None, [_sz, _vp, _vp, _vp]),
    "orc_primary_ray": (None, [_sz, _P(abi.Camera), _vp, _vp, _vp]),
    "orc_grid_build": (C.c_int32, [_vp, _vp, C.c_uint64, _P(_vp)]),
    "orc_grid_describe": (None, [_vp, _P(abi.GridDesc)]),
    "orc_grid_destroy": (None, [_vp]),
    "orc_scene_create": (_vp, [_P(abi.Object), C.c_uint32, _P(abi.GridDesc), C.c_uint32]),
    "orc_scene_destroy": (None, [_vp]),
    "orc_scene_intersect": (None, [_vp, _sz, _vp, _vp, _vp, _vp]),
    "orc_grid_intersect": (None, [_vp, C.c_uint32, _sz, _vp, _vp, _vp, _vp]),
    "orc_trace_sample": (C.c_int32, [_vp, _P(abi.Camera), _P(abi.Settings), C.c_uint32, C.c_uint32, C.c_uint32, _vp, _vp, _vp]),
    "orc_trace_samples": (None, [_vp, _P(abi.Camera), _P(abi.Settings), _sz, _vp, _vp, _vp]),
    "orc_render_tiles": (None, [_vp, _P(abi.Camera), _P(abi.Settings), _P(abi.TileRect), C.c_uint32, _vp, C.c_uint32]),
    "orc_mesh_load_ply": (C.c_int32, [C.c_char_p, _P(_vp)]),
    "orc_mesh_load_ply_text": (C.c_int32, [C.c_char_p, _sz, _P(_vp)]),
    "orc_mesh_bake_transform": (None, [_vp, _vp]),
    "orc_mesh_size": (C.c_uint64, [_vp]),
    "orc_mesh_arrays": (None, [_vp, _P(_vp), _P(_vp)]),
    "orc_mesh_bounds": (None, [_vp, _vp, _vp]),
    "orc_mesh_destroy": (None, [_vp]),
    "orc_resolve_tonemap": (None, [_vp, _sz, C.c_double, C.c_double, C.c_double, _vp]),
    "orc_set_mutation": (C.c_int32, [C.c_int32]),
    "orc_count_black_paths": (None, [C.c_int32]),
    "orc_counters_reset": (None, []),
    "orc_counters_get": (None, [_vp]),
}

_lib = None
_fast = None


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def _open(path):
    if not os.path.exists(path):
        build()
    lib = C.CDLL(path)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


def load(fast=False):
    """liboracle.so (the checker, counting its work) or, fast=True, liboracle_fast.so (un-instrumented, for timing)."""
    global _lib, _fast
    if fast:
        if _fast is None:
            _fast = _open(ORACLE_FAST_PATH)
        return _fast
    if _lib is None:
        _lib = _open(ORACLE_PATH)
    return _lib


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class OracleScene:
    """Owns an orc_scene built from a raymond_amd.scene.Scene."""

    def __init__(self, scene, fast=False):
        self.lib = load(fast)
        objs, n, descs, ng, keep = scene.flatten()
        self._keep = (objs, descs, keep)
        self.handle = self.lib.orc_scene_create(objs, n, descs, ng)
        assert self.handle

    def close(self):
        if self.handle:
            self.lib.orc_scene_destroy(self.handle)
            self.handle = None

    def __del__(self):
        self.close()

    def trace_samples(self, cam, settings, xy, samples):
        xy = np.ascontiguousarray(xy, dtype=np.uint32).reshape(-1, 2)
        samples = np.ascontiguousarray(samples, dtype=np.uint32)
        out = np.zeros((xy.shape[0], 3))
        c, s = cam.pod(), settings.pod()
        self.lib.orc_trace_samples(self.handle, C.byref(c), C.byref(s), xy.shape[0], ptr(xy), ptr(samples), ptr(out))
        return out

    def trace_sample_path(self, cam, settings, x, y, sample):
        rgb = np.zeros(3)
        po = np.full(abi.RMD_MAX_BOUNCE_LIMIT + 1, -2, dtype=np.int32)
        ps = np.zeros(abi.RMD_MAX_BOUNCE_LIMIT + 1, dtype=np.uint32)
        c, s = cam.pod(), settings.pod()
        n = self.lib.orc_trace_sample(self.handle, C.byref(c), C.byref(s), x, y, sample, ptr(rgb), ptr(po), ptr(ps))
        return rgb, po[:n].copy(), ps[:n].copy()

    def render_tiles(self, cam, settings, tiles, accum=None, sample_begin=0, sample_count=None, threads=0):
        from raymond_amd.scene import tile_array

        W, H = cam.backbuffer_width, cam.backbuffer_height
        if accum is None:
            accum = np.zeros((H, W, 3))
        c, s = cam.pod(), settings.pod(sample_begin, sample_count)
        self.lib.orc_render_tiles(self.handle, C.byref(c), C.byref(s), tile_array(tiles), len(tiles), ptr(accum), threads)
        return accum

    def scene_intersect(self, rays):
        rays = f64(rays).reshape(-1, 6)
        n = rays.shape[0]
        obj = np.zeros(n, dtype=np.int32)
        t = np.zeros(n)
        sub = np.zeros(n, dtype=np.uint32)
        self.lib.orc_scene_intersect(self.handle, n, ptr(rays), ptr(obj), ptr(t), ptr(sub))
        return obj, t, sub

    def grid_intersect(self, g, rays):
        rays = f64(rays).reshape(-1, 6)
        n = rays.shape[0]
        hit = np.zeros(n, dtype=np.int32)
        t = np.zeros(n)
        tri = np.zeros(n, dtype=np.uint32)
        self.lib.orc_grid_intersect(self.handle, g, n, ptr(rays), ptr(hit), ptr(t), ptr(tri))
        return hit, t, tri


def grid_build(mesh):
    """Oracle's AccGrid::build_from_mesh -> raymond_amd.scene.AccGrid (arrays copied)."""
    from raymond_amd.scene import AccGrid

    lib = load()
    h = C.c_void_p()
    rc = lib.orc_grid_build(ptr(mesh.tri_pos), ptr(mesh.tri_nrm), len(mesh), C.byref(h))
    if rc != 0:
        return rc, None
    d = abi.GridDesc()
    lib.orc_grid_describe(h, C.byref(d))
    g = AccGrid.from_desc(d)
    lib.orc_grid_destroy(h)
    return 0, g


def load_ply(path=None, text=None, translate=None):
    """Oracle's Mesh::load_ply (+ optional bake_transform) -> (rc, raymond_amd.scene.Mesh or None, (bbox_min, bbox_max)).
    rc = 1 where the reference would panic."""
    from raymond_amd.scene import Mesh

    lib = load()
    h = C.c_void_p()
    if text is not None:
        raw = text if isinstance(text, bytes) else text.encode()
        rc = lib.orc_mesh_load_ply_text(raw, len(raw), C.byref(h))
    else:
        rc = lib.orc_mesh_load_ply(str(path).encode(), C.byref(h))
    if rc != 0:
        return rc, None, None
    try:
        if translate is not None:
            t = f64(translate)
            lib.orc_mesh_bake_transform(h, ptr(t))
        n = int(lib.orc_mesh_size(h))
        pp, pn = C.c_void_p(), C.c_void_p()
        lib.orc_mesh_arrays(h, C.byref(pp), C.byref(pn))
        if n:
            pos = np.ctypeslib.as_array(C.cast(pp, C.POINTER(C.c_double)), shape=(n * 9,)).copy()
            nrm = np.ctypeslib.as_array(C.cast(pn, C.POINTER(C.c_double)), shape=(n * 9,)).copy()
        else:
            pos = nrm = np.zeros(0)
        mn, mx = np.zeros(3), np.zeros(3)
        lib.orc_mesh_bounds(h, ptr(mn), ptr(mx))
        return 0, Mesh(pos.reshape(-1, 9), nrm.reshape(-1, 9)), (mn, mx)
    finally:
        lib.orc_mesh_destroy(h)


def resolve_tonemap(accum, sample_count, exposure=1.0, gamma=2.2):
    """await's division + cli_old's tone-map / gamma / u8 cast -> uint8 array of accum's shape."""
    a = f64(accum)
    out = np.zeros(a.shape, dtype=np.uint8)
    load().orc_resolve_tonemap(ptr(a), a.size // 3, float(sample_count), float(exposure), float(gamma), ptr(out))
    return out


def block_uniforms(seed, pixel, sample, block):
    """-> (n, 3): first and second 53-bit uniform and the 22-bit uniform of each (pixel, sample, block)"""
    lib = load()
    out = np.zeros((len(pixel), 3))
    row = np.zeros(3)
    for i, (p, s, b) in enumerate(zip(pixel, sample, block)):
        lib.orc_block_uniforms(seed, int(p), int(s), int(b), ptr(row))
        out[i] = row
    return out


def counters():
    out = np.zeros(12, dtype=np.uint64)
    load().orc_counters_get(ptr(out))
    names = ["samples", "segments", "cells", "tri_tests", "mesh_hits", "bounces", "draws", "walks", "occupied_cells", "zero_weight_diffuse", "zero_weight_specular", "retests"]
    return dict(zip(names, (int(v) for v in out)))


def counters_reset():
    load().orc_counters_reset()
