"""The render kernel's loops fail loudly.  Every loop of the kernel has a bound that no input reaches (render_kernel.hpp, grid_walk.hpp); a wave
that runs into one sets the context's fault word, poisons the launch's work counter so that the launch drains, and leaves — and the first call
that waits for the launch returns RMD_ERR_DEVICE_FAULT with the loop's name instead of a frame (include/raymond_hip.h; the reference's own
failure mode there is a hang: `TaskHandle::await` polls a counter that a panicked worker never decrements, src/trace.rs:82-92).

The bounds cannot be reached with the product library, so this test loads the DIAG build of the same sources (`make -C raymond_amd/csrc diag`:
diag/libraymond_hip.so, built by __graft_entry__.build()) whose RMD_DEBUG bits FORCE each bound: 32 = the trip loops' (stall watch of
render_wave, trip counts of render_wave_sorted and render_wave_queued), 128 = a persistent wave's work loop — the outer one and the draws the
chained / queued wave bodies make themselves.  (A grid walk's round and stepping loops carry no counter of
their own: they are bounded by the rays' exit counters — grid_walk.hpp says why.)  One child process, run once.

The same DIAG build also checks the walk's sphere pre-test (bit 64): see the second test."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIAG_LIB = os.path.join(ROOT, "raymond_amd", "csrc", "diag", "libraymond_hip.so")

CHILD = r"""
import os, sys, time
sys.path.insert(0, %(root)r)
from raymond_amd import abi, lib, render, scenes
from raymond_amd.scene import Settings, generate_tiles

assert lib.LIB_PATH.endswith("diag/libraymond_hip.so"), lib.LIB_PATH
spheres, mesh = scenes.reflective_spheres(), scenes.gold_dragon_standin(n=24)

def attempt(debug, scene, spp, want_words, W=256, H=256, queues=0, want_queued=None):
    os.environ["RMD_DEBUG"] = str(debug)  # read once, when the context is created (DIAG builds only)
    with render.Context(0) as ctx:
        ctx.set_tunable(abi.RMD_TUNE_PATH_QUEUES, queues)  # 1: the lane-per-path form of the mesh kernel (render_wave)
        cam = scenes.camera(W, H)
        st = Settings(cam, sample_count=spp, bounce_limit=5, seed=scenes.SEED)
        ds, fb = render.DeviceScene(ctx, scene), render.Framebuffer(ctx, W, H)
        tiles = generate_tiles(W, H, st.tile_size)
        t0 = time.perf_counter()
        try:
            render.render_tiles(ctx, ds, cam, st, tiles, fb)
            status, text = abi.RMD_OK, ""
        except lib.RaymondError as e:
            status, text = e.status, str(e)
        dt = time.perf_counter() - t0
        info = ctx.last_launch_info()
        assert want_queued is None or info.queued == want_queued, (debug, info.queued)
        if want_words is None:
            assert status == abi.RMD_OK, text
        else:
            assert status == abi.RMD_ERR_DEVICE_FAULT, (debug, status, text)
            assert "device fault" in text and "not valid" in text and any(w in text for w in want_words), text
            assert dt < 20.0, dt  # the poisoned work counter drains the launch: nobody spins
            # the fault has been reported and cleared: a wait with nothing new behind it is clean again
            ctx.synchronize()
            # an asynchronous launch reports at the wait, not at the enqueue
            render.render_tiles(ctx, ds, cam, st, tiles, fb, sync=False)
            try:
                ctx.synchronize()
                raise SystemExit("no fault reported by rmd_context_synchronize")
            except lib.RaymondError as e:
                assert e.status == abi.RMD_ERR_DEVICE_FAULT
            # ... and a download of the frame that launch wrote is refused as well
            render.render_tiles(ctx, ds, cam, st, tiles, fb, sync=False)
            try:
                fb.download()
                raise SystemExit("a frame cut short by a fault was handed out")
            except lib.RaymondError as e:
                assert e.status == abi.RMD_ERR_DEVICE_FAULT
        fb.close(), ds.close()
        print("debug %%3d spp %%3d grid %%d split %%d persistent %%d -> status %%d in %%.2f s  %%s" %% (debug, spp, info.has_grid, info.split_k, info.persistent, status, dt, text[:160]), flush=True)

attempt(0, spheres, 128, None)                                   # the DIAG build renders normally without a forced bound
attempt(32, spheres, 128, ["render_wave_sorted"])                # role-sorted spheres kernel: its trip count
attempt(32, spheres, 4, ["trip loop of render_wave;"])           # lane-per-path form, direct mode: the stall watch
attempt(32, mesh, 16, ["trip loop of render_wave_queued;"], want_queued=1)          # mesh kernel, split launch: the paths in queues
attempt(32, mesh, 16, ["trip loop of render_wave;"], queues=1, want_queued=0)       # ... and its lane-per-path form (RMD_TUNE_PATH_QUEUES = 1)
attempt(32, mesh, 2, ["trip loop of render_wave;"])              # mesh kernel, direct mode
attempt(128, mesh, 16, ["work loop of a persistent workgroup"], 512, 512, want_queued=1)  # a persistent wave's second draw (a launch with several items per wave slot)
attempt(128, mesh, 16, ["work loop of a persistent workgroup"], 512, 512, queues=1, want_queued=0)
attempt(128, spheres, 128, ["work loop of a persistent workgroup"], 512, 512)
attempt(0, mesh, 16, None, want_queued=1)
attempt(0, mesh, 16, None, queues=1, want_queued=0)
print("faults ok")
"""


def test_every_render_loop_reports_a_forced_bound_as_a_device_fault(product_lib):
    assert os.path.exists(DIAG_LIB), "build the DIAG library: python -c 'import __graft_entry__ as g; g.build()'"
    env = dict(os.environ, RAYMOND_HIP_LIB=DIAG_LIB)
    env.pop("RMD_DEBUG", None)
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    sys.stdout.write(r.stdout)
    assert r.returncode == 0 and "faults ok" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])


PRETEST_CHILD = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
from raymond_amd import lib, render, scenes
from raymond_amd.scene import Settings, generate_tiles

assert lib.LIB_PATH.endswith("diag/libraymond_hip.so"), lib.LIB_PATH
os.environ["RMD_DEBUG"] = "72"  # 8: event counters, 64: every pair the sphere pre-test drops is ALSO run through the reference's test and counted if it passes

def run(name, scene, W, H, spp, end_black_paths=False, dof=False):
    with render.Context(0) as ctx:
        cam = scenes.camera(W, H, aperture_radius=0.05 if dof else 0.0)
        st = Settings(cam, sample_count=spp, bounce_limit=5, seed=scenes.SEED)
        st.end_black_paths, st.use_dof = end_black_paths, dof
        ds, fb = render.DeviceScene(ctx, scene), render.Framebuffer(ctx, W, H)
        sys.stderr.write("== %%s\n" %% name), sys.stderr.flush()
        render.render_tiles(ctx, ds, cam, st, generate_tiles(W, H, st.tile_size), fb)
        info = ctx.last_launch_info()
        assert info.buffered == 1 and info.has_grid == 1, name  # the split launch: the walk with the pre-test
        fb.close(), ds.close()

run("benchmark mesh", scenes.gold_dragon_standin(), 384, 256, 16)
run("benchmark mesh, black paths ended, thin lens", scenes.gold_dragon_standin(), 256, 256, 16, True, True)
# the same shape in triangles a hundred times smaller and eight times larger (coarse meshes: few, large triangles seen at grazing angles too)
run("fine mesh", scenes.mesh_scene(scenes.lumpy_sphere_mesh(n=160, extent=(0.23, 0.17, 0.1), centre=(0.0, -0.5, 0.0))), 256, 256, 16)
run("coarse mesh", scenes.mesh_scene(scenes.lumpy_sphere_mesh(n=4)), 256, 256, 32)
run("coarse flat mesh", scenes.mesh_scene(scenes.lumpy_sphere_mesh(n=8, extent=(3.0, 0.5, 0.3))), 256, 256, 32)
print("pretest ok")
"""


def test_the_sphere_pre_test_drops_no_pair_that_passes_the_reference_test(product_lib):
    """grid_walk.hpp: a (ray, triangle) pair whose line passes the triangle's sphere by is not run through triangle.rs:11-44.  In the DIAG build
    (RMD_DEBUG bit 64) every dropped pair IS run through it as well and counted if it passes: the count stays 0, and the pre-test does drop pairs."""
    import re

    assert os.path.exists(DIAG_LIB), "build the DIAG library: python -c 'import __graft_entry__ as g; g.build()'"
    env = dict(os.environ, RAYMOND_HIP_LIB=DIAG_LIB)
    env.pop("RMD_DEBUG", None)
    r = subprocess.run([sys.executable, "-c", PRETEST_CHILD % {"root": ROOT}], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0 and "pretest ok" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])
    scenes_seen, name = {}, None
    for line in r.stderr.splitlines():
        if line.startswith("== "):
            name = line[3:]
        m = re.search(r"tests=(\d+)", line)
        if m and name:
            scenes_seen.setdefault(name, {})["tests"] = scenes_seen.get(name, {}).get("tests", 0) + int(m.group(1))
        m = re.search(r"pairs passed=(\d+) full chunks=(\d+) .*must be 0\)=(\d+)", line)
        if m and name:
            d = scenes_seen.setdefault(name, {})
            d["passed"] = d.get("passed", 0) + int(m.group(1))
            d["wrongly_dropped"] = d.get("wrongly_dropped", 0) + int(m.group(3))
    print(scenes_seen)
    assert len(scenes_seen) == 5, r.stderr[-3000:]
    for name, d in scenes_seen.items():
        assert d["wrongly_dropped"] == 0, (name, d)
        assert 0 < d["passed"] < d["tests"], (name, d)  # it ran, and it dropped pairs


def test_the_product_build_ignores_the_diag_switches(gpu_ctx):
    """RMD_DEBUG is honoured by DIAG builds only: the product library renders the same frame with it set."""
    import numpy as np

    from raymond_amd import render, scenes
    from raymond_amd.scene import Settings, generate_tiles

    sc = scenes.reflective_spheres()
    cam = scenes.camera(96, 64)
    st = Settings(cam, sample_count=4, bounce_limit=5, seed=scenes.SEED)
    tiles = generate_tiles(96, 64, st.tile_size)
    frames = []
    for dbg in (None, "32", "224"):
        if dbg is None:
            os.environ.pop("RMD_DEBUG", None)
        else:
            os.environ["RMD_DEBUG"] = dbg
        try:
            with render.Context(0) as ctx:
                ds, fb = render.DeviceScene(ctx, sc), render.Framebuffer(ctx, 96, 64)
                render.render_tiles(ctx, ds, cam, st, tiles, fb)
                frames.append(fb.download())
                fb.close(), ds.close()
        finally:
            os.environ.pop("RMD_DEBUG", None)
    assert np.array_equal(frames[0], frames[1]) and np.array_equal(frames[0], frames[2])
