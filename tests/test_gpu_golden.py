"""GPU vs the COMMITTED golden fixtures (tests/golden/, produced by tools/gen_golden.py from the oracle).

Unlike test_gpu_parity.py this needs no oracle at run time: it pins the HIP path to vectors that were generated
once in the build container, so a silent change of the oracle cannot mask a change of the kernel.
"""
import json
import os

import numpy as np
import pytest

from raymond_amd import probe, render, scenes
from raymond_amd.scene import Settings, generate_tiles

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def close(a, b, rtol):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return (np.abs(a - b) <= rtol * np.maximum(np.maximum(np.abs(a), np.abs(b)), 1e-300)) | (a == b) | (np.isnan(a) & np.isnan(b))


def test_function_known_answers(gpu_ctx):
    kat = np.load(os.path.join(GOLD, "kat_functions.npz"))
    n = 256
    for name in ("sphere", "plane", "aabb", "triangle"):
        hit, t = probe.hit_t(gpu_ctx, name, kat[name + "_in"], kat[name + "_rays"])
        assert np.array_equal(hit, kat[name + "_hit"])
        m = hit == 1
        assert close(t[m], kat[name + "_t"][m], 1e-15).all()
    (nn,) = probe.call(gpu_ctx, "triangle_normal", n, [kat["trinrm_pos"], kat["trinrm_nrm"], kat["trinrm_rays"], kat["trinrm_t"]], [3])
    assert close(nn, kat["trinrm_out"], 1e-15).all()
    t3, b3 = probe.call(gpu_ctx, "onb", n, [kat["onb_n"]], [3, 3])
    assert np.array_equal(t3, kat["onb_t"]) and np.array_equal(b3, kat["onb_b"])
    d, pdf = probe.call(gpu_ctx, "cosine_hemisphere", n, [kat["cos_r1"], kat["cos_r2"]], [3, 1])
    assert np.abs(d - kat["cos_dir"]).max() <= 1e-15 and np.array_equal(pdf[:, 0], kat["cos_pdf"])
    (g,) = probe.call(gpu_ctx, "importance_sample_ggx", n, [kat["ggx_reflect"], kat["ggx_rough"], kat["cos_r1"], kat["cos_r2"]], [3])
    assert np.abs(g - kat["ggx_dir"]).max() <= 1e-14
    (D,) = probe.call(gpu_ctx, "ggx_distribution", n, [kat["brdf_n"], kat["brdf_h"], kat["ggx_rough"]], [1])
    (G,) = probe.call(gpu_ctx, "geometry_smith", n, [kat["brdf_n"], kat["brdf_v"], kat["brdf_l"], kat["ggx_rough"]], [1])
    assert close(D[:, 0], kat["brdf_D"], 1e-15).all() and close(G[:, 0], kat["brdf_G"], 1e-15).all()
    (F,) = probe.call(gpu_ctx, "fresnel_schlick", n, [kat["fresnel_cos"], kat["fresnel_f0"]], [3])
    assert close(F, kat["fresnel_out"], 1e-13).all()
    u = probe.block_uniforms(gpu_ctx, scenes.SEED, kat["rng_pixel"], kat["rng_sample"], kat["rng_block"])
    assert np.array_equal(u[:, :3], kat["rng_u"]) and np.array_equal(u[:, [3, 4, 2]], kat["rng_u"])  # next2 and next3 read the same block
    pr = probe.primary_ray(gpu_ctx, scenes.camera(1920, 1080), kat["pray_xy"], kat["pray_u"])
    assert close(pr, kat["pray_out"], 1e-15).all()


def _check_paths(gpu_ctx, scene, settings, fname):
    f = np.load(os.path.join(GOLD, fname))
    ds = render.DeviceScene(gpu_ctx, scene)
    rgb, po, ps = probe.trace_samples(gpu_ctx, ds, settings.camera_settings, settings, f["xy"], f["sample"], paths=True)
    ds.close()
    same = (po == f["path_obj"]).all(axis=1) & (ps == f["path_sub"]).all(axis=1)
    assert same.mean() >= 0.998
    assert close(rgb[same], f["rgb"][same], 1e-9).all()
    return same


def test_path_fixtures_spheres(gpu_ctx):
    _check_paths(gpu_ctx, scenes.reflective_spheres(), scenes.config_settings("C1"), "paths_spheres.npz")


def test_path_fixtures_mesh(gpu_ctx):
    st = Settings(scenes.camera(480, 270), sample_count=1, bounce_limit=5, seed=scenes.SEED)
    _check_paths(gpu_ctx, scenes.gold_dragon_standin(n=24), st, "paths_mesh.npz")


def test_config1_image_fixture(gpu_ctx):
    f = np.load(os.path.join(GOLD, "image_c1.npz"))
    sc, st = scenes.reflective_spheres(), scenes.config_settings("C1")
    ds = render.DeviceScene(gpu_ctx, sc)
    fb = render.Framebuffer(gpu_ctx, 256, 256)
    render.render_tiles(gpu_ctx, ds, st.camera_settings, st, generate_tiles(256, 256, st.tile_size), fb)
    img = fb.download()
    fb.close(), ds.close()
    assert np.allclose(img.reshape(8, 32, 8, 32, 3).sum(axis=(1, 3)), f["tile_sums"], rtol=2e-3, atol=1e-9)
    ok = close(img[96:160, 96:160], f["crop"], 1e-9).all(axis=2)
    assert ok.mean() >= 0.995
    assert np.allclose(img.mean(axis=(0, 1)), f["mean"], rtol=1e-3)


def test_full_size_grid_digest(gpu_ctx, product_lib):
    """The 99,372-triangle stand-in of C3-C5: product host builder == committed oracle digests, and it uploads."""
    import hashlib

    with open(os.path.join(GOLD, "grid_digests.json")) as fh:
        d = json.load(fh)["91"]
    sc = scenes.gold_dragon_standin()
    g = sc.objects[1].geometry.grid
    assert hashlib.sha256(g.cells.tobytes()).hexdigest() == d["cells_sha256"]
    assert hashlib.sha256(g.mapping_table.tobytes()).hexdigest() == d["mapping_sha256"]
    ds = render.DeviceScene(gpu_ctx, sc)
    ds.close()
