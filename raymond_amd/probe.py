"""Python wrappers of the diagnostic probes (include/raymond_hip_probe.h) for the parity tests."""
import ctypes as C
import os

import numpy as np

from . import abi
from . import lib as _lib

_vp = C.c_void_p
_sz = C.c_size_t
_P = C.POINTER
_SIGS = {
    "rmd_probe_philox4x32_10": [_vp, _sz, _vp, _vp, _vp],
    "rmd_probe_block_uniforms": [_vp, C.c_uint64, _sz, _vp, _vp, _vp, _vp],
    "rmd_probe_sphere_intersect": [_vp, _sz, _vp, _vp, _vp, _vp],
    "rmd_probe_sphere_normal": [_vp, _sz, _vp, _vp, _vp, _vp],
    "rmd_probe_plane_intersect": [_vp, _sz, _vp, _vp, _vp, _vp],
    "rmd_probe_aabb_intersect": [_vp, _sz, _vp, _vp, _vp, _vp],
    "rmd_probe_triangle_intersect": [_vp, _sz, _vp, _vp, _vp, _vp],
    "rmd_probe_triangle_normal": [_vp, _sz, _vp, _vp, _vp, _vp, _vp],
    "rmd_probe_onb": [_vp, _sz, _vp, _vp, _vp],
    "rmd_probe_cosine_hemisphere": [_vp, _sz, _vp, _vp, _vp, _vp],
    "rmd_probe_importance_sample_ggx": [_vp, _sz, _vp, _vp, _vp, _vp, _vp],
    "rmd_probe_ggx_distribution": [_vp, _sz, _vp, _vp, _vp, _vp],
    "rmd_probe_geometry_smith": [_vp, _sz, _vp, _vp, _vp, _vp, _vp],
    "rmd_probe_fresnel_schlick": [_vp, _sz, _vp, _vp, _vp],
    "rmd_probe_primary_ray": [_vp, _sz, _P(abi.Camera), _vp, _vp, _vp],
    "rmd_probe_elementary": [_vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp],
    "rmd_probe_scene_intersect": [_vp, _vp, _sz, _vp, _vp, _vp, _vp],
    "rmd_probe_grid_intersect": [_vp, _vp, C.c_uint32, _sz, _vp, _vp, _vp, _vp],
    "rmd_probe_trace_samples": [_vp, _vp, _P(abi.Camera), _P(abi.Settings), _sz, _vp, _vp, _vp, _vp, _vp],
    "rmd_probe_triangle_sphere": [_sz, _vp, _vp],
    "rmd_probe_pretest_pairs": [_vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp],
}
PATH_STRIDE = 17
_ready = False


PROBE_LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libraymond_hip_probe.so")  # beside the product library it links against
_probe_lib = None


def _L():
    """libraymond_hip_probe.so: test infrastructure, a library of its own (the product library exports no probe)."""
    global _ready, _probe_lib
    if not _ready:
        _lib.load()  # the product library first: the probes call into it
        if not os.path.exists(PROBE_LIB_PATH):
            raise ImportError("raymond_amd.probe: %s is missing — build it with `python -c 'import __graft_entry__ as g; g.build()'`" % PROBE_LIB_PATH)
        _probe_lib = C.CDLL(PROBE_LIB_PATH)
        for name, args in _SIGS.items():
            fn = getattr(_probe_lib, name)
            fn.restype, fn.argtypes = C.c_int32, args
        _ready = True
    return _probe_lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f(a, w=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a if w is None else a.reshape(-1, w)


def hit_t(ctx, name, shape, rays):
    """sphere/plane/aabb/triangle intersect -> (hit int32[n], t float64[n])"""
    L = _L()
    rays = _f(rays, 6)
    shape = _f(shape).reshape(rays.shape[0], -1)
    n = rays.shape[0]
    hit = np.zeros(n, dtype=np.int32)
    t = np.zeros(n)
    ctx.check(getattr(L, "rmd_probe_%s_intersect" % name)(ctx.handle, n, _p(shape), _p(rays), _p(hit), _p(t)))
    return hit, t


def call(ctx, name, n, ins, outs):
    """generic: ins = list of float64 arrays, outs = list of output widths"""
    L = _L()
    ins = [_f(a) for a in ins]
    res = [np.zeros((n, w)) for w in outs]
    ctx.check(getattr(L, "rmd_probe_" + name)(ctx.handle, n, *[_p(a) for a in ins], *[_p(a) for a in res]))
    return res


def philox(ctx, ctr, key):
    L = _L()
    ctr = np.ascontiguousarray(ctr, dtype=np.uint32).reshape(-1, 4)
    key = np.ascontiguousarray(key, dtype=np.uint32).reshape(-1, 2)
    out = np.zeros_like(ctr)
    ctx.check(L.rmd_probe_philox4x32_10(ctx.handle, ctr.shape[0], _p(ctr), _p(key), _p(out)))
    return out


def block_uniforms(ctx, seed, pixel, sample, block):
    """-> (n, 5): u53_0, u53_1 of the block as next2() returns them, then r (22-bit), r1, r2 as next3() returns them"""
    L = _L()
    pixel, sample, block = (np.ascontiguousarray(a, dtype=np.uint32) for a in (pixel, sample, block))
    out = np.zeros((pixel.shape[0], 5))
    ctx.check(L.rmd_probe_block_uniforms(ctx.handle, seed, pixel.shape[0], _p(pixel), _p(sample), _p(block), _p(out)))
    return out


def elementary(ctx, x):
    """-> (sqrt64(x), sin(x), cos(x), root, 1/root) as the device computes them (root, 1/root: normalize()'s pair)"""
    L = _L()
    x = _f(x).ravel()
    s, si, co, ro, ir = (np.zeros_like(x) for _ in range(5))
    ctx.check(L.rmd_probe_elementary(ctx.handle, x.shape[0], _p(x), _p(s), _p(si), _p(co), _p(ro), _p(ir)))
    return s, si, co, ro, ir


def primary_ray(ctx, cam, xy, u):
    L = _L()
    xy = np.ascontiguousarray(xy, dtype=np.uint32).reshape(-1, 2)
    u = _f(u, 2)
    out = np.zeros((xy.shape[0], 6))
    c = cam.pod()
    ctx.check(L.rmd_probe_primary_ray(ctx.handle, xy.shape[0], C.byref(c), _p(xy), _p(u), _p(out)))
    return out


def scene_intersect(ctx, dscene, rays):
    L = _L()
    rays = _f(rays, 6)
    n = rays.shape[0]
    obj = np.zeros(n, dtype=np.int32)
    t = np.zeros(n)
    sub = np.zeros(n, dtype=np.uint32)
    ctx.check(L.rmd_probe_scene_intersect(ctx.handle, dscene.handle, n, _p(rays), _p(obj), _p(t), _p(sub)))
    return obj, t, sub


def grid_intersect(ctx, dscene, g, rays):
    L = _L()
    rays = _f(rays, 6)
    n = rays.shape[0]
    hit = np.zeros(n, dtype=np.int32)
    t = np.zeros(n)
    tri = np.zeros(n, dtype=np.uint32)
    ctx.check(L.rmd_probe_grid_intersect(ctx.handle, dscene.handle, g, n, _p(rays), _p(hit), _p(t), _p(tri)))
    return hit, t, tri


def trace_samples(ctx, dscene, cam, settings, xy, samples, paths=False):
    L = _L()
    xy = np.ascontiguousarray(xy, dtype=np.uint32).reshape(-1, 2)
    samples = np.ascontiguousarray(samples, dtype=np.uint32)
    n = xy.shape[0]
    rgb = np.zeros((n, 3))
    po = np.full((n, PATH_STRIDE), -2, dtype=np.int32) if paths else None
    ps = np.zeros((n, PATH_STRIDE), dtype=np.uint32) if paths else None
    c, s = cam.pod(), settings.pod()
    ctx.check(
        L.rmd_probe_trace_samples(ctx.handle, dscene.handle, C.byref(c), C.byref(s), n, _p(xy), _p(samples), _p(rgb),
                                  _p(po) if paths else None, _p(ps) if paths else None)
    )
    return (rgb, po, ps) if paths else rgb


def triangle_sphere(pos9):
    """Host only (no GPU): the pre-test sphere of each triangle -> (centre float64[n, 3], r2a float64[n], kb float64[n]); api: rmd_probe_triangle_sphere."""
    from . import abi as _abi

    L = _L()
    pos9 = _f(pos9, 9)
    out = np.zeros((pos9.shape[0], 5))
    st = L.rmd_probe_triangle_sphere(pos9.shape[0], _p(pos9), _p(out))
    if st != _abi.RMD_OK:
        raise RuntimeError("rmd_probe_triangle_sphere: status %d" % st)
    return out[:, :3], out[:, 3], out[:, 4]


def pretest_pairs(ctx, sphere5, pos9, ray6):
    """The walk's sphere pre-test in the device's own arithmetic and the device's triangle test on explicit pairs -> (passed bool[n], hit bool[n],
    t float64[n]); sphere5 = centre, r2a, kb per pair.  api: rmd_probe_pretest_pairs."""
    L = _L()
    sphere5, pos9, ray6 = _f(sphere5, 5), _f(pos9, 9), _f(ray6, 6)
    n = pos9.shape[0]
    passed, hit, t = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32), np.zeros(n)
    ctx.check(L.rmd_probe_pretest_pairs(ctx.handle, n, _p(sphere5), _p(pos9), _p(ray6), _p(passed), _p(hit), _p(t)))
    return passed.astype(bool), hit.astype(bool), t
