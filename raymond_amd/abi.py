"""ctypes mirror of include/raymond_hip.h (POD structs and enums only).

Field order, names and widths follow the header one-to-one; the header in turn cites the
reference types it stands for (core/src/scene.rs, core/src/lib.rs, src/trace.rs:32-55,
core/src/tile.rs:7-14).  tests/test_abi_layout.py checks sizes/offsets against the
compiled library.
"""
import ctypes as C

RMD_ABI_VERSION = 5

RMD_OK = 0
RMD_ERR_INVALID_ARGUMENT = 1
RMD_ERR_NO_DEVICE = 2
RMD_ERR_HIP = 3
RMD_ERR_OUT_OF_MEMORY = 4
RMD_ERR_GRID_INDEX = 5
RMD_ERR_UNSUPPORTED = 6
RMD_ERR_RCCL = 7
RMD_ERR_DEVICE_FAULT = 8

STATUS_NAMES = {
    0: "RMD_OK",
    1: "RMD_ERR_INVALID_ARGUMENT",
    2: "RMD_ERR_NO_DEVICE",
    3: "RMD_ERR_HIP",
    4: "RMD_ERR_OUT_OF_MEMORY",
    5: "RMD_ERR_GRID_INDEX",
    6: "RMD_ERR_UNSUPPORTED",
    7: "RMD_ERR_RCCL",
    8: "RMD_ERR_DEVICE_FAULT",
}

# enum Geometry { Plane, Sphere, Grid } — core/src/scene.rs:9-13
RMD_GEOM_PLANE, RMD_GEOM_SPHERE, RMD_GEOM_GRID = 0, 1, 2
# enum Material { Diffuse, Metal, Emission } — core/src/lib.rs:21-26
RMD_MAT_DIFFUSE, RMD_MAT_METAL, RMD_MAT_EMISSION = 0, 1, 2

RMD_MAX_BOUNCE_LIMIT = 16
RMD_RENDER_DOF = 1  # rmd_settings.flags
RMD_RENDER_TRACE_BLACK_PATHS = 2  # never end a zero-throughput path early
RMD_RENDER_END_BLACK_PATHS = 4  # end them in scenes with grids too (flags 0: only where provably exact, i.e. scenes without grids)
(RMD_TUNE_SAMPLE_SPLIT, RMD_TUNE_WALK_BATCH, RMD_TUNE_MASK_BUDGET, RMD_TUNE_LAUNCH_FORM, RMD_TUNE_SCRATCH_CAP_MB, RMD_TUNE_WALK_CUT,
 RMD_TUNE_SPLIT_MIN_SAMPLES, RMD_TUNE_CHAIN_ITEMS, RMD_TUNE_AXIS_PAIRS, RMD_TUNE_PATH_QUEUES) = range(10)
RMD_LAUNCH_AUTO, RMD_LAUNCH_PER_ITEM, RMD_LAUNCH_PERSISTENT = range(3)
RMD_COMM_ID_BYTES = 128


class Material(C.Structure):
    _fields_ = [
        ("kind", C.c_uint32),
        ("_pad", C.c_uint32),
        ("color", C.c_double * 3),
        ("roughness", C.c_double),
        ("emission_aux", C.c_double * 5),
    ]


class Object(C.Structure):
    _fields_ = [
        ("geometry_kind", C.c_uint32),
        ("grid_index", C.c_uint32),
        ("origin", C.c_double * 3),
        ("normal", C.c_double * 3),
        ("radius", C.c_double),
        ("material", Material),
    ]


class GridDesc(C.Structure):
    _fields_ = [
        ("bbox_min", C.c_double * 3),
        ("bbox_max", C.c_double * 3),
        ("resolution", C.c_uint32 * 3),
        ("_pad", C.c_uint32),
        ("cell_size", C.c_double * 3),
        ("cells", C.POINTER(C.c_uint32)),
        ("n_cells", C.c_uint64),
        ("mapping_table", C.POINTER(C.c_uint32)),
        ("n_mapping", C.c_uint64),
        ("tri_pos", C.POINTER(C.c_double)),
        ("tri_nrm", C.POINTER(C.c_double)),
        ("n_tris", C.c_uint64),
        ("built", C.c_void_p),
    ]


class Camera(C.Structure):
    _fields_ = [
        ("backbuffer_width", C.c_uint32),
        ("backbuffer_height", C.c_uint32),
        ("fov_vert", C.c_double),
        ("position", C.c_double * 3),
        ("focal_length", C.c_double),
        ("aperture_radius", C.c_double),
    ]


class Settings(C.Structure):
    _fields_ = [
        ("bounce_limit", C.c_uint32),
        ("sample_begin", C.c_uint32),
        ("sample_count", C.c_uint32),
        ("flags", C.c_uint32),
        ("seed", C.c_uint64),
    ]


class TileRect(C.Structure):
    _fields_ = [
        ("left", C.c_uint32),
        ("top", C.c_uint32),
        ("width", C.c_uint32),
        ("height", C.c_uint32),
    ]


class LaunchInfo(C.Structure):  # rmd_launch_info
    _fields_ = [
        ("passes", C.c_uint32),
        ("split_k", C.c_uint32),
        ("persistent", C.c_uint32),
        ("end_black_paths", C.c_uint32),
        ("has_grid", C.c_uint32),
        ("waves_per_workgroup", C.c_uint32),
        ("buffered", C.c_uint32),
        ("chained", C.c_uint32),
        ("queued", C.c_uint32),
    ]
