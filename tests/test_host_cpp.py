"""The C++ host mirror (raymond_amd/host): reference-shaped API over the C-ABI, driven through raymond_cli.

CPU part: inputs it builds (procedural mesh, tile order, PLY loader + bake_transform) equal the Python mirror's.
GPU part: a render through render_tiled/TaskHandle::await in C++ equals the Python path bit for bit.
"""
import os
import subprocess

import numpy as np
import pytest

from raymond_amd import render, scenes
from raymond_amd.scene import Settings, generate_tiles

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "raymond_amd", "host", "raymond_cli")


@pytest.fixture(scope="module")
def cli(product_lib):
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "raymond_amd", "host")], check=True)
    return CLI


def run(cli, *args):
    return subprocess.run([cli, *[str(a) for a in args]], capture_output=True, text=True)


@pytest.mark.parametrize("n", [5, 24])
def test_procedural_mesh_is_bit_identical_to_the_python_generator(cli, tmp_path, n):
    out = tmp_path / "mesh.bin"
    assert run(cli, "mesh", n, out).returncode == 0
    m = scenes.lumpy_sphere_mesh(n)
    raw = np.fromfile(out)
    k = len(m) * 9
    assert raw[:k].tobytes() == m.tri_pos.tobytes() and raw[k:].tobytes() == m.tri_nrm.tobytes()


def test_tile_order(cli):
    r = run(cli, "tiles", 1920, 1080, 32, 32)
    tiles = [tuple(int(v) for v in line.split()) for line in r.stdout.split("\n") if line]
    assert tiles == generate_tiles(1920, 1080, (32, 32))


def test_ply_loader_and_bake_transform(cli, tmp_path):
    """mesh.rs:58-121: ASCII PLY in Blender's layout, with and without `s t`, quads dropped, then :48-56."""
    rng = np.random.default_rng(3)
    verts = rng.normal(size=(10, 8))
    faces = [(0, 1, 2), (2, 3, 4), (4, 5, 6), (7, 8, 9)]
    for with_uv in (True, False):
        ply = tmp_path / ("uv.ply" if with_uv else "nouv.ply")
        with open(ply, "w") as f:
            f.write("ply\nformat ascii 1.0\ncomment test\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n" % len(verts))
            f.write("property float nx\nproperty float ny\nproperty float nz\n")
            if with_uv:
                f.write("property float s\nproperty float t\n")
            f.write("element face %d\nproperty list uchar uint vertex_indices\nend_header\n" % (len(faces) + 1))
            for v in verts:
                f.write(" ".join(repr(float(x)) for x in (v if with_uv else v[:6])) + "\n")
            f.write("4 0 1 2 3\n")  # a quad: dropped (mesh.rs:116)
            for a, b, c in faces:
                f.write("3 %d %d %d\n" % (a, b, c))
        out = tmp_path / "ply.bin"
        r = run(cli, "ply", ply, out)
        assert r.returncode == 0 and "4 triangles" in r.stdout
        raw = np.fromfile(out)
        pos = np.array([[verts[i][:3] + np.array([0.0, -0.3, 2.9]) for i in f] for f in faces]).reshape(-1)
        nrm = np.array([[verts[i][3:6] for i in f] for f in faces]).reshape(-1)
        assert raw[:36].tobytes() == pos.tobytes() and raw[36:].tobytes() == nrm.tobytes()
    assert run(cli, "ply", tmp_path / "missing.ply", tmp_path / "x.bin").returncode == 1  # the reference panics


@pytest.mark.skipif(os.path.exists("/dev/kfd") and os.access("/dev/kfd", os.R_OK | os.W_OK), reason="a GPU is present")
def test_render_without_a_gpu_fails_loudly(cli, tmp_path):
    r = run(cli, "render", "spheres", 32, 32, 1, 2, tmp_path / "x.ppm")
    assert r.returncode == 1 and "no HIP device" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("scene,spi,gpus", [("spheres", 0, 1), ("dragon:12", 0, 1), ("spheres", 3, 1), ("project", 0, 1), ("dragon:12", 2, 1), ("dragon:12", 2, 2), ("spheres", 0, 3)])
def test_cpp_render_tiled_equals_python_path(cli, gpu_ctx, tmp_path, scene, spi, gpus):
    """The C++ host mirror (tile queue, progressive passes, tile sums resident on the GPU between passes, only the tiles of a message downloaded —
    on a copy stream while the next pass renders) gives the frame of one rmd_render_tiles call bit for bit: one pass, progressive passes
    (7 spp in passes of 3 or 2: the last pass is shorter), and several workers (rehearsed on this box's one GPU: every worker its own context),
    where a tile's sums travel through RAM from one GPU's framebuffer to another's."""
    W, H, spp, bounces = 96, 64, 7, 4
    ppm, raw = tmp_path / "o.ppm", tmp_path / "o.f64"
    if scene == "project":  # the spheres scene through its serde-JSON project file (core/src/project.rs)
        from raymond_amd.project import Project

        (tmp_path / "spheres.json").write_text(Project.from_scene(scenes.reflective_spheres()).dumps())
        scene = "project:%s" % (tmp_path / "spheres.json")
    os.environ["RAYMOND_REHEARSE_ON_DEVICE0"] = "1"
    try:
        r = run(cli, "render", scene, W, H, spp, bounces, ppm, "--raw", raw, "--spi", spi, "--gpus", gpus)
    finally:
        os.environ.pop("RAYMOND_REHEARSE_ON_DEVICE0", None)
    assert r.returncode == 0, r.stderr
    img_cpp = np.fromfile(raw).reshape(H, W, 3)
    sc = scenes.gold_dragon_standin(n=12) if scene.startswith("dragon") else scenes.reflective_spheres()
    st = Settings(scenes.camera(W, H), sample_count=spp, tile_size=(32, 32), bounce_limit=bounces, seed=scenes.SEED)
    ds = render.DeviceScene(gpu_ctx, sc)
    fb = render.Framebuffer(gpu_ctx, W, H)
    render.render_tiles(gpu_ctx, ds, st.camera_settings, st, generate_tiles(W, H, (32, 32)), fb)
    img_py = fb.download() / float(spp)
    assert img_cpp.tobytes() == img_py.tobytes()
    # tone-mapped PPM (host libm) == the library's resolve stage, byte for byte (pixels an ulp of the device's exp / pow could decide are
    # recomputed on the host: rmd_resolve_tonemap)
    with open(ppm, "rb") as f:
        assert f.readline() == b"P6\n" and f.readline() == b"%d %d\n" % (W, H) and f.readline() == b"255\n"
        host8 = np.frombuffer(f.read(), dtype=np.uint8).reshape(H, W, 3)
    dev8 = render.resolve_tonemap(gpu_ctx, fb, spp)
    assert np.array_equal(host8, dev8)
    fb.close(), ds.close()
