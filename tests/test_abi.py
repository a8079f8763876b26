"""The C-ABI boundary without a GPU: the library loads, exports every symbol the headers declare, the ctypes
mirror has the C layout, and a call that needs a device fails loudly with a status and a message."""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

import pytest

from raymond_amd import abi, lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def declared_functions(header):
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", header)).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(rmd_[a-z0-9_]+)\s*\(", text)))


def exported_functions(path):
    out = subprocess.run(["nm", "-D", "--defined-only", path], check=True, capture_output=True, text=True).stdout
    return set(re.findall(r" T (rmd_[a-z0-9_]+)", out)), out


def test_every_declared_symbol_is_exported(product_lib):
    """include/raymond_hip.h <-> libraymond_hip.so (the product), include/raymond_hip_probe.h <-> libraymond_hip_probe.so (test
    infrastructure): each library exports every function its header declares, and the product exports no probe."""
    from raymond_amd import probe

    names = declared_functions("raymond_hip.h")
    assert len(names) >= 28
    exported, out = exported_functions(lib.LIB_PATH)
    missing = [n for n in names if n not in exported]
    assert not missing, "declared in include/raymond_hip.h but not exported: %s" % missing
    assert set(lib.SIGNATURES) <= exported
    assert not [n for n in exported if n.startswith("rmd_probe_")]
    probes = declared_functions("raymond_hip_probe.h")
    assert len(probes) >= 18 and all(n.startswith("rmd_probe_") for n in probes)
    probe_exported, _ = exported_functions(probe.PROBE_LIB_PATH)
    assert not [n for n in probes if n not in probe_exported] and set(probe._SIGS) <= probe_exported
    # the boundary is plain C: no C++-mangled rmd entry points, no torch types anywhere near it
    assert not re.search(r" T _Z\w*rmd_render", out)
    assert product_lib.rmd_abi_version() == abi.RMD_ABI_VERSION


def test_library_does_not_link_the_oracle_or_torch():
    out = subprocess.run(["ldd", lib.LIB_PATH], check=True, capture_output=True, text=True).stdout
    assert "liboracle" not in out and "torch" not in out and "libamdhip64" in out
    needed = subprocess.run(["readelf", "-d", lib.LIB_PATH], check=True, capture_output=True, text=True).stdout
    assert "rccl" not in needed  # resolved lazily with dlopen, only by rmd_comm_*


def test_struct_layout_matches_the_c_header():
    src = r"""
#include <stdio.h>
#include <stddef.h>
#include "raymond_hip.h"
#define S(T) printf(#T " %zu\n", sizeof(T))
#define O(T, f) printf(#T "." #f " %zu\n", offsetof(T, f))
int main(void) {
  S(rmd_material); O(rmd_material, color); O(rmd_material, roughness); O(rmd_material, emission_aux);
  S(rmd_object); O(rmd_object, grid_index); O(rmd_object, origin); O(rmd_object, normal); O(rmd_object, radius); O(rmd_object, material);
  S(rmd_grid_desc); O(rmd_grid_desc, bbox_max); O(rmd_grid_desc, resolution); O(rmd_grid_desc, cell_size); O(rmd_grid_desc, cells);
  O(rmd_grid_desc, n_cells); O(rmd_grid_desc, mapping_table); O(rmd_grid_desc, n_mapping); O(rmd_grid_desc, tri_pos); O(rmd_grid_desc, tri_nrm); O(rmd_grid_desc, n_tris);
  S(rmd_camera); O(rmd_camera, fov_vert); O(rmd_camera, position); O(rmd_camera, focal_length); O(rmd_camera, aperture_radius);
  S(rmd_settings); O(rmd_settings, sample_begin); O(rmd_settings, sample_count); O(rmd_settings, flags); O(rmd_settings, seed);
  S(rmd_tile_rect); O(rmd_tile_rect, height);
  S(rmd_launch_info); O(rmd_launch_info, split_k); O(rmd_launch_info, persistent); O(rmd_launch_info, end_black_paths); O(rmd_launch_info, has_grid); O(rmd_launch_info, waves_per_workgroup); O(rmd_launch_info, buffered); O(rmd_launch_info, chained); O(rmd_launch_info, queued);
  return 0; }
"""
    with tempfile.TemporaryDirectory() as d:
        c, exe = os.path.join(d, "layout.c"), os.path.join(d, "layout")
        open(c, "w").write(src)
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)
        lines = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split("\n")
    c_layout = dict((l.split()[0], int(l.split()[1])) for l in lines if l.strip())
    py = {"rmd_material": abi.Material, "rmd_object": abi.Object, "rmd_grid_desc": abi.GridDesc, "rmd_camera": abi.Camera,
          "rmd_settings": abi.Settings, "rmd_tile_rect": abi.TileRect, "rmd_launch_info": abi.LaunchInfo}
    for key, val in c_layout.items():
        if "." in key:
            t, f = key.split(".")
            assert getattr(py[t], f).offset == val, key
        else:
            assert C.sizeof(py[key]) == val, key


@pytest.mark.skipif(os.path.exists("/dev/kfd") and os.access("/dev/kfd", os.R_OK | os.W_OK), reason="a GPU is present")
def test_no_device_is_a_loud_error_not_a_fallback(product_lib):
    h = C.c_void_p()
    status = product_lib.rmd_context_create(0, C.byref(h))
    assert status == abi.RMD_ERR_NO_DEVICE and not h
    assert b"no HIP device" in product_lib.rmd_last_error(None)
    from raymond_amd import render

    with pytest.raises(lib.RaymondError):
        render.Context(0)


def test_null_arguments_are_rejected(product_lib):
    assert product_lib.rmd_context_create(0, None) == abi.RMD_ERR_INVALID_ARGUMENT
    assert product_lib.rmd_render_tiles(None, None, None, None, None, 0, None) == abi.RMD_ERR_INVALID_ARGUMENT
    assert product_lib.rmd_comm_unique_id(None) == abi.RMD_ERR_INVALID_ARGUMENT
    product_lib.rmd_context_destroy(None)  # no-ops, like free(NULL)
    product_lib.rmd_scene_destroy(None)
    product_lib.rmd_grid_build_destroy(None)


def test_product_package_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under raymond_amd/ (or bench.py outside its cpu_baseline leg) may
    import, link or name it."""
    pkg = os.path.join(ROOT, "raymond_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")) or f == "Makefile":
                text = open(os.path.join(base, f), errors="ignore").read()
                for needle in ("oracle_lib", "liboracle", "orc_", "oracle.h", "import oracle"):
                    assert needle not in text, "%s mentions %s" % (os.path.join(base, f), needle)
    bench = open(os.path.join(ROOT, "bench.py")).read()
    assert bench.count("import oracle_lib") == 1 and bench.index("import oracle_lib") > bench.index("not args.no_cpu_baseline")


def test_header_constants_match_the_python_mirror():
    """Flags, tunable keys and the ABI version of include/raymond_hip.h as the C compiler sees them == raymond_amd/abi.py; and Settings
    builds rmd_settings.flags from them (the thin lens and both black-path switches are opt-in: flags 0, the reference-identical mode, is the default)."""
    from raymond_amd import scenes
    from raymond_amd.scene import Settings

    src = r"""
#include <stdio.h>
#include "raymond_hip.h"
int main(void) {
  printf("%u %u %u %u %u %u %u %u %u %u %u %u %u %u %u\n", RMD_ABI_VERSION, RMD_RENDER_DOF, RMD_RENDER_TRACE_BLACK_PATHS, RMD_RENDER_END_BLACK_PATHS, (unsigned)RMD_TUNE_SAMPLE_SPLIT, (unsigned)RMD_TUNE_WALK_BATCH,
         (unsigned)RMD_TUNE_MASK_BUDGET, (unsigned)RMD_TUNE_LAUNCH_FORM, (unsigned)RMD_TUNE_SCRATCH_CAP_MB, (unsigned)RMD_TUNE_WALK_CUT, (unsigned)RMD_TUNE_SPLIT_MIN_SAMPLES, (unsigned)RMD_TUNE_CHAIN_ITEMS, (unsigned)RMD_TUNE_AXIS_PAIRS, (unsigned)RMD_TUNE_PATH_QUEUES, (unsigned)RMD_TUNE_COUNT);
  return 0; }
"""
    with tempfile.TemporaryDirectory() as d:
        c, exe = os.path.join(d, "consts.c"), os.path.join(d, "consts")
        open(c, "w").write(src)
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)
        got = [int(v) for v in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()]
    assert got == [abi.RMD_ABI_VERSION, abi.RMD_RENDER_DOF, abi.RMD_RENDER_TRACE_BLACK_PATHS, abi.RMD_RENDER_END_BLACK_PATHS, abi.RMD_TUNE_SAMPLE_SPLIT, abi.RMD_TUNE_WALK_BATCH,
                   abi.RMD_TUNE_MASK_BUDGET, abi.RMD_TUNE_LAUNCH_FORM, abi.RMD_TUNE_SCRATCH_CAP_MB, abi.RMD_TUNE_WALK_CUT, abi.RMD_TUNE_SPLIT_MIN_SAMPLES, abi.RMD_TUNE_CHAIN_ITEMS, abi.RMD_TUNE_AXIS_PAIRS, abi.RMD_TUNE_PATH_QUEUES, 10]
    cam = scenes.camera(64, 48, aperture_radius=0.5)
    assert Settings(cam, 4).pod().flags == 0
    assert Settings(cam, 4, use_dof=True).pod().flags == abi.RMD_RENDER_DOF
    assert Settings(cam, 4, use_dof=True, trace_black_paths=True).pod().flags == (abi.RMD_RENDER_DOF | abi.RMD_RENDER_TRACE_BLACK_PATHS)
    assert Settings(cam, 4, end_black_paths=True).pod().flags == abi.RMD_RENDER_END_BLACK_PATHS


def test_nothing_throws_across_the_boundary_when_the_host_runs_out_of_memory():
    """include/raymond_hip.h: "nothing throws or aborts across the boundary".  rmd_grid_build_from_mesh keeps a std::vector of std::vectors over every
    cell; under an address-space limit that leaves no room for them its std::bad_alloc must come back as RMD_ERR_OUT_OF_MEMORY — with a message,
    with the caller's process alive and able to build a small grid afterwards — and not terminate the caller (a Rust or ctypes host) through an
    exception that unwinds into C.  Run in a child process: the limit is the child's own."""
    child = r"""
import ctypes as C, os, resource, sys
import numpy as np
sys.path.insert(0, %r)
from raymond_amd import abi, lib
L = lib.load()
n = 1_500_000                                     # ~3 n cells, each a std::vector: far more than the limit below leaves
rng = np.random.default_rng(1)
base = rng.random((n, 1, 3))
pos = np.ascontiguousarray((base + 1e-3 * rng.random((n, 3, 3))).reshape(n, 9))
nrm = np.ascontiguousarray(np.tile(np.array([0.0, 0.0, 1.0] * 3), (n, 1)))
small_pos, small_nrm = np.ascontiguousarray(pos[:500]), np.ascontiguousarray(nrm[:500])
vm = int([l for l in open("/proc/self/status") if l.startswith("VmSize")][0].split()[1]) * 1024
resource.setrlimit(resource.RLIMIT_AS, (vm + (48 << 20), vm + (48 << 20)))
h = C.c_void_p()
s = L.rmd_grid_build_from_mesh(pos.ctypes.data_as(C.c_void_p), nrm.ctypes.data_as(C.c_void_p), n, C.byref(h))
msg = L.rmd_last_error(None).decode()
print("status", s, "handle", h.value, "text", msg)
assert s == abi.RMD_ERR_OUT_OF_MEMORY and not h.value and "memory" in msg
s2 = L.rmd_grid_build_from_mesh(small_pos.ctypes.data_as(C.c_void_p), small_nrm.ctypes.data_as(C.c_void_p), 500, C.byref(h))
assert s2 == abi.RMD_OK and h.value
L.rmd_grid_build_destroy(h)
print("survived")
""" % ROOT
    r = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "survived" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
