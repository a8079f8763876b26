"""Loader for the product C-ABI library (raymond_amd/csrc/libraymond_hip.so).

There is deliberately NO fallback: if the HIP library is missing or fails to load, every
entry point raises.  The CPU oracle under oracle/ is test infrastructure and is never
imported from here.
"""
import ctypes as C
import os

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# RAYMOND_HIP_LIB selects another build of the SAME library (tuning experiments); there is still no fallback.
LIB_PATH = os.environ.get("RAYMOND_HIP_LIB") or os.path.join(_HERE, "csrc", "libraymond_hip.so")

_lib = None

# name -> (restype, argtypes); mirrors include/raymond_hip.h
_P = C.POINTER
_vp = C.c_void_p
SIGNATURES = {
    "rmd_abi_version": (C.c_uint32, []),
    "rmd_context_create": (C.c_int32, [C.c_int32, _P(_vp)]),
    "rmd_context_create_on_stream": (C.c_int32, [C.c_int32, _vp, _P(_vp)]),
    "rmd_context_destroy": (None, [_vp]),
    "rmd_last_error": (C.c_char_p, [_vp]),
    "rmd_context_memory_info": (C.c_int32, [_vp, _P(C.c_uint64), _P(C.c_uint64)]),
    "rmd_context_set_tunable": (C.c_int32, [_vp, C.c_uint32, C.c_int64]),
    "rmd_context_get_tunable": (C.c_int32, [_vp, C.c_uint32, _P(C.c_int64)]),
    "rmd_scene_create": (C.c_int32, [_vp, _P(abi.Object), C.c_uint32, _P(abi.GridDesc), C.c_uint32, _P(_vp)]),
    "rmd_scene_destroy": (None, [_vp]),
    "rmd_framebuffer_alloc": (C.c_int32, [_vp, C.c_uint32, C.c_uint32, _P(_vp)]),
    "rmd_framebuffer_free": (C.c_int32, [_vp, _vp]),
    "rmd_framebuffer_zero": (C.c_int32, [_vp, _vp, C.c_size_t]),
    "rmd_framebuffer_download": (C.c_int32, [_vp, _vp, _vp, C.c_size_t]),
    "rmd_framebuffer_upload": (C.c_int32, [_vp, _vp, _vp, C.c_size_t]),
    "rmd_framebuffer_download_tiles": (C.c_int32, [_vp, _vp, C.c_uint32, C.c_uint32, _P(abi.TileRect), C.c_uint32, _vp]),
    "rmd_framebuffer_download_tiles_async": (C.c_int32, [_vp, _vp, C.c_uint32, C.c_uint32, _P(abi.TileRect), C.c_uint32, _vp]),
    "rmd_context_wait_transfers": (C.c_int32, [_vp]),
    "rmd_framebuffer_upload_tiles": (C.c_int32, [_vp, _vp, _vp, C.c_uint32, C.c_uint32, _P(abi.TileRect), C.c_uint32]),
    "rmd_host_alloc": (C.c_int32, [_vp, C.c_size_t, _P(_vp)]),
    "rmd_host_free": (C.c_int32, [_vp, _vp]),
    "rmd_render_tiles": (
        C.c_int32,
        [_vp, _vp, _P(abi.Camera), _P(abi.Settings), _P(abi.TileRect), C.c_uint32, _vp],
    ),
    "rmd_render_tiles_async": (
        C.c_int32,
        [_vp, _vp, _P(abi.Camera), _P(abi.Settings), _P(abi.TileRect), C.c_uint32, _vp],
    ),
    "rmd_context_synchronize": (C.c_int32, [_vp]),
    "rmd_render_tiles_host": (
        C.c_int32,
        [_vp, _vp, _P(abi.Camera), _P(abi.Settings), _P(abi.TileRect), C.c_uint32, _vp],
    ),
    "rmd_last_kernel_ms": (C.c_int32, [_vp, _P(C.c_float)]),
    "rmd_last_launch_info": (C.c_int32, [_vp, _P(abi.LaunchInfo)]),
    "rmd_resolve_tonemap": (
        C.c_int32,
        [_vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.c_double, _vp],
    ),
    "rmd_comm_prepare_process": (C.c_int32, []),
    "rmd_comm_unique_id": (C.c_int32, [_vp]),
    "rmd_comm_create": (C.c_int32, [_vp, _vp, C.c_int32, C.c_int32, _P(_vp)]),
    "rmd_comm_destroy": (None, [_vp]),
    "rmd_reduce_framebuffer": (C.c_int32, [_vp, _vp, C.c_size_t, C.c_int32]),
    "rmd_reduce_framebuffer_async": (C.c_int32, [_vp, _vp, C.c_size_t, C.c_int32]),
    "rmd_grid_build_from_mesh": (C.c_int32, [_vp, _vp, C.c_uint64, _P(_vp)]),
    "rmd_grid_build_from_mesh_gpu": (C.c_int32, [_vp, _vp, _vp, C.c_uint64, _P(_vp)]),
    "rmd_grid_build_describe": (C.c_int32, [_vp, _P(abi.GridDesc)]),
    "rmd_grid_build_destroy": (None, [_vp]),
}


class RaymondError(RuntimeError):
    def __init__(self, status, text):
        super().__init__("%s: %s" % (abi.STATUS_NAMES.get(status, status), text))
        self.status = status


def load():
    """Returns the ctypes handle of libraymond_hip.so with argtypes set; raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "raymond_amd: %s is missing — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback)" % LIB_PATH
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    if lib.rmd_abi_version() != abi.RMD_ABI_VERSION:
        raise ImportError("raymond_amd: ABI version mismatch between abi.py and libraymond_hip.so")
    _lib = lib
    return lib


def check(status, ctx=None):
    if status != abi.RMD_OK:
        text = load().rmd_last_error(ctx)
        raise RaymondError(status, text.decode() if text else "")
