#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/ from the CPU oracle (and, for png_blocks, from the
reference's own example render).  Run in the build container:  python tools/gen_golden.py

  kat_functions.npz      256 seeded inputs + oracle outputs per device function (level-1 known answers)
  paths_spheres.npz      1,024 (pixel, sample) -> hit sequence + radiance, ReflectiveSpheres 256x256, 3 bounces
  paths_mesh.npz         1,024 (pixel, sample) -> hit sequence + radiance, GoldDragon-standin (n=24), 480x270, 5 bounces
  image_c1.npz           config-1 image (256x256, 16 spp, 3 bounces): 32x32-tile sums + 64x64 centre crop + sha256
  grid_digests.json      sha256 of cells / mapping_table for the procedural meshes (n = 13, 40, 91)
  work_counters.json     oracle work counters per config at 1920x1080 (cells, triangle tests, mesh hits per sample)
  png_blocks.npy         8x8 block means (42x74x3) of /root/reference/examples/ReflectiveSpheres.png — the one piece
                         of ground truth the reference itself provides for this path (data, not source)
"""
import ctypes as C
import hashlib
import json
import os
import struct
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from raymond_amd import scenes  # noqa: E402
from raymond_amd.scene import Settings, generate_tiles  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
os.makedirs(GOLD, exist_ok=True)
L = O.load()
P = O.ptr
N = 256


def unit(rng, n):
    v = rng.normal(size=(n, 3))
    return v / np.sqrt((v * v).sum(axis=1))[:, None]


def rays_toward(rng, n, target, spread):
    o = rng.uniform(-2, 2, size=(n, 3))
    d = np.asarray(target) + rng.normal(scale=spread, size=(n, 3)) - o
    return np.concatenate([o, d / np.sqrt((d * d).sum(axis=1))[:, None]], axis=1)


def hit_t(name, shape, rays):
    hit = np.zeros(len(rays), dtype=np.int32)
    t = np.zeros(len(rays))
    getattr(L, "orc_%s_intersect" % name)(len(rays), P(shape), P(rays), P(hit), P(t))
    return hit, t


def kat_functions():
    rng = np.random.default_rng(20261003)
    out = {}
    sph = np.concatenate([rng.uniform(-1, 1, (N, 3)), rng.uniform(0.1, 1.0, (N, 1))], axis=1)
    rays = rays_toward(rng, N, (0, 0, 0), 0.8)
    rays[:8, :3] = sph[:8, :3]
    out["sphere_in"], out["sphere_rays"] = sph, rays
    out["sphere_hit"], out["sphere_t"] = hit_t("sphere", sph, rays)
    pl = np.concatenate([rng.uniform(-1, 1, (N, 3)), unit(rng, N)], axis=1)
    rays = rays_toward(rng, N, (0, 0, 0), 1.0)
    rays[:8, 3:] = np.cross(pl[:8, 3:], unit(rng, 8))
    out["plane_in"], out["plane_rays"] = pl, rays
    out["plane_hit"], out["plane_t"] = hit_t("plane", pl, rays)
    lo = rng.uniform(-1, 0, (N, 3))
    bb = np.concatenate([lo, lo + rng.uniform(0.1, 1.5, (N, 3))], axis=1)
    rays = rays_toward(rng, N, (0, 0, 0), 1.0)
    rays[:8, 3] = 0.0
    rays[8:12, 4] = -0.0
    out["aabb_in"], out["aabb_rays"] = bb, rays
    out["aabb_hit"], out["aabb_t"] = hit_t("aabb", bb, rays)
    tri = (rng.uniform(-0.5, 0.5, (N, 1, 3)) + rng.normal(scale=0.4, size=(N, 3, 3))).reshape(N, 9)
    tri[:4, 3:6] = tri[:4, 0:3]
    rays = rays_toward(rng, N, (0, 0, 0), 0.5)
    out["triangle_in"], out["triangle_rays"] = tri, rays
    out["triangle_hit"], out["triangle_t"] = hit_t("triangle", tri, rays)
    # Heron normal
    pos = rng.normal(size=(N, 9))
    nrm = np.concatenate([unit(rng, N), unit(rng, N), unit(rng, N)], axis=1)
    bary = rng.dirichlet((1, 1, 1), N)
    p = bary[:, :1] * pos[:, 0:3] + bary[:, 1:2] * pos[:, 3:6] + bary[:, 2:3] * pos[:, 6:9]
    o = p + unit(rng, N)
    d = p - o
    tt = np.sqrt((d * d).sum(axis=1))
    rays = np.ascontiguousarray(np.concatenate([o, d / tt[:, None]], axis=1))
    n3 = np.zeros((N, 3))
    L.orc_triangle_normal(N, P(pos), P(nrm), P(rays), P(tt), P(n3))
    out["trinrm_pos"], out["trinrm_nrm"], out["trinrm_rays"], out["trinrm_t"], out["trinrm_out"] = pos, nrm, rays, tt, n3
    # ONB
    nn = unit(rng, N)
    nn[0], nn[1], nn[2], nn[3] = (0, 0, 1), (0, 0, -1), (1, 0, 0), (0, 1, -0.0)
    t3, b3 = np.zeros((N, 3)), np.zeros((N, 3))
    L.orc_onb(N, P(nn), P(t3), P(b3))
    out["onb_n"], out["onb_t"], out["onb_b"] = nn, t3, b3
    # samplers
    r1, r2 = rng.uniform(0, 1, N), rng.uniform(0, 1, N)
    r1[:4] = [0.0, 1.0 - 2**-53, 2**-53, 0.5]
    r2[:4] = [0.0, 1.0 - 2**-53, 0.25, 0.75]
    d3, pdf = np.zeros((N, 3)), np.zeros(N)
    L.orc_cosine_hemisphere(N, P(r1), P(r2), P(d3), P(pdf))
    out["cos_r1"], out["cos_r2"], out["cos_dir"], out["cos_pdf"] = r1, r2, d3, pdf
    refl, rough = unit(rng, N), rng.uniform(0.01, 0.9, N)
    g3 = np.zeros((N, 3))
    L.orc_importance_sample_ggx(N, P(refl), P(rough), P(r1), P(r2), P(g3))
    out["ggx_reflect"], out["ggx_rough"], out["ggx_dir"] = refl, rough, g3
    a, b, c, dd = unit(rng, N), unit(rng, N), unit(rng, N), unit(rng, N)
    D, G = np.zeros(N), np.zeros(N)
    L.orc_ggx_distribution(N, P(a), P(b), P(rough), P(D))
    L.orc_geometry_smith(N, P(a), P(c), P(dd), P(rough), P(G))
    out["brdf_n"], out["brdf_h"], out["brdf_v"], out["brdf_l"], out["brdf_D"], out["brdf_G"] = a, b, c, dd, D, G
    ct, f0 = rng.uniform(-1, 1, N), rng.uniform(0, 1, (N, 3))
    F = np.zeros((N, 3))
    L.orc_fresnel_schlick(N, P(ct), P(f0), P(F))
    out["fresnel_cos"], out["fresnel_f0"], out["fresnel_out"] = ct, f0, F
    # RNG stream + primary rays
    pix = rng.integers(0, 1920 * 1080, N).astype(np.uint32)
    smp = rng.integers(0, 4000, N).astype(np.uint32)
    blk = rng.integers(0, 32, N).astype(np.uint32)
    out["rng_pixel"], out["rng_sample"], out["rng_block"] = pix, smp, blk
    out["rng_u"] = O.block_uniforms(scenes.SEED, pix, smp, blk)
    cam = scenes.camera(1920, 1080)
    xy = np.stack([rng.integers(0, 1920, N), rng.integers(0, 1080, N)], axis=1).astype(np.uint32)
    u = rng.uniform(0, 1, (N, 2))
    pr = np.zeros((N, 6))
    cp = cam.pod()
    L.orc_primary_ray(N, C.byref(cp), P(xy), P(u), P(pr))
    out["pray_xy"], out["pray_u"], out["pray_out"] = xy, u, pr
    np.savez_compressed(os.path.join(GOLD, "kat_functions.npz"), **out)


def paths(scene, settings, n, seed, fname):
    cam = settings.camera_settings
    rng = np.random.default_rng(seed)
    xy = np.stack([rng.integers(0, cam.backbuffer_width, n), rng.integers(0, cam.backbuffer_height, n)], axis=1).astype(np.uint32)
    smp = rng.integers(0, 4000, n).astype(np.uint32)
    osc = O.OracleScene(scene)
    rgb = np.zeros((n, 3))
    po = np.full((n, 17), -2, dtype=np.int32)
    ps = np.zeros((n, 17), dtype=np.uint32)
    for i in range(n):
        c, o, s = osc.trace_sample_path(cam, settings, int(xy[i, 0]), int(xy[i, 1]), int(smp[i]))
        rgb[i] = c
        po[i, : len(o)] = o
        ps[i, : len(s)] = s
    np.savez_compressed(os.path.join(GOLD, fname), xy=xy, sample=smp, rgb=rgb, path_obj=po, path_sub=ps)


def image_c1():
    sc, st = scenes.reflective_spheres(), scenes.config_settings("C1")
    img = O.OracleScene(sc).render_tiles(st.camera_settings, st, generate_tiles(256, 256, st.tile_size))
    tiles = img.reshape(8, 32, 8, 32, 3).sum(axis=(1, 3))
    np.savez_compressed(os.path.join(GOLD, "image_c1.npz"), tile_sums=tiles, crop=img[96:160, 96:160].copy(),
                        sha256=np.frombuffer(hashlib.sha256(img.tobytes()).digest(), dtype=np.uint8), mean=img.mean(axis=(0, 1)))


def grid_digests():
    out = {}
    for n in (13, 40, 91):
        mesh = scenes.lumpy_sphere_mesh(n)
        mesh.bake_transform((0.0, -0.3, 2.9))
        rc, g = O.grid_build(mesh)
        assert rc == 0
        out[str(n)] = {
            "triangles": len(mesh),
            "resolution": [int(v) for v in g.resolution],
            "n_mapping": int(g.mapping_table.size),
            "tri_pos_sha256": hashlib.sha256(mesh.tri_pos.tobytes()).hexdigest(),
            "cells_sha256": hashlib.sha256(g.cells.tobytes()).hexdigest(),
            "mapping_sha256": hashlib.sha256(g.mapping_table.tobytes()).hexdigest(),
        }
    with open(os.path.join(GOLD, "grid_digests.json"), "w") as f:
        json.dump(out, f, indent=1)


def work_counters(only=()):
    path = os.path.join(GOLD, "work_counters.json")
    out = json.load(open(path)) if only and os.path.exists(path) else {}
    for name, spp in (("C1", 16), ("C2", 2), ("C3", 2), ("C4", 1), ("C5", 2)):
        if only and name not in only:
            continue
        st = scenes.config_settings(name, spp=spp)
        cam = st.camera_settings
        sc = getattr(scenes, scenes.CONFIGS[name][0])(grid_builder=lambda m: O.grid_build(m)[1]) if name in ("C3", "C4", "C5") else scenes.reflective_spheres()
        # <name>: the work of the product's default, which ends a path whose bounce weights multiply to exactly zero (its sample is zero in
        # the reference too); <name>_reference: everything the reference executes (RMD_RENDER_TRACE_BLACK_PATHS)
        for key, count_black in ((name, 0), (name + "_reference", 1)):
            O.load().orc_count_black_paths(count_black)
            O.counters_reset()
            O.OracleScene(sc).render_tiles(cam, st, generate_tiles(cam.backbuffer_width, cam.backbuffer_height, st.tile_size))
            c = O.counters()
            c["config"] = "%s %dx%d %d spp %d bounces%s" % (name, cam.backbuffer_width, cam.backbuffer_height, spp, st.bounce_limit,
                                                          ", path segments behind a zero bounce weight included" if count_black else "")
            out[key] = c
            print(key, c, flush=True)
        O.load().orc_count_black_paths(0)
    with open(os.path.join(GOLD, "work_counters.json"), "w") as f:
        json.dump(out, f, indent=1)


def read_png_rgb(path):
    d = open(path, "rb").read()
    assert d[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat = 8, b""
    while pos < len(d):
        n, typ = struct.unpack(">I4s", d[pos : pos + 8])
        body = d[pos + 8 : pos + 8 + n]
        pos += 12 + n
        if typ == b"IHDR":
            w, h, bd, ct, _, _, il = struct.unpack(">IIBBBBB", body)
        elif typ == b"IDAT":
            idat += body
    raw = zlib.decompress(idat)
    ch = {2: 3, 6: 4}[ct]
    assert bd == 8 and il == 0
    stride = w * ch
    out = np.zeros((h, stride), dtype=np.uint8)
    prev = np.zeros(stride, dtype=np.int32)
    p = 0
    for y in range(h):
        f = raw[p]
        line = np.frombuffer(raw[p + 1 : p + 1 + stride], dtype=np.uint8).astype(np.int32)
        p += 1 + stride
        cur = np.zeros(stride, dtype=np.int32)
        if f == 0:
            cur = line
        elif f == 2:
            cur = (line + prev) & 255
        else:
            for i in range(stride):
                a = cur[i - ch] if i >= ch else 0
                b = prev[i]
                c = prev[i - ch] if i >= ch else 0
                if f == 1:
                    pr = a
                elif f == 3:
                    pr = (a + b) // 2
                else:
                    pp = a + b - c
                    pa, pb, pc = abs(pp - a), abs(pp - b), abs(pp - c)
                    pr = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[i] = (line[i] + pr) & 255
        out[y] = cur
        prev = cur
    return out.reshape(h, w, ch)[:, :, :3]


def png_blocks():
    src = "/root/reference/examples/ReflectiveSpheres.png"
    if not os.path.exists(src):
        print("reference PNG not present; keeping the committed png_blocks.npy")
        return
    img = read_png_rgb(src).astype(np.float64)
    assert img.shape == (340, 592, 3)
    blocks = img[:336].reshape(42, 8, 74, 8, 3).mean(axis=(1, 3))
    np.save(os.path.join(GOLD, "png_blocks.npy"), blocks.astype(np.float32))
    # the reference's render of cli_old's scene (examples/GoldDragon.png, README.md:27): its dragon-free regions pin the room, the emitter,
    # the camera and the red sphere of cli_old/src/main.rs:45-141 (tests/png_pin.py: run_room_checks)
    img = read_png_rgb("/root/reference/examples/GoldDragon.png").astype(np.float64)
    assert img.shape == (340, 592, 3)
    np.save(os.path.join(GOLD, "png_blocks_dragon.npy"), img[:336].reshape(42, 8, 74, 8, 3).mean(axis=(1, 3)).astype(np.float32))


if __name__ == "__main__":
    which = sys.argv[1:] or ["kat", "paths", "image", "grids", "counters", "png"]
    if "kat" in which:
        kat_functions()
    if "paths" in which:
        paths(scenes.reflective_spheres(), scenes.config_settings("C1"), 1024, 101, "paths_spheres.npz")
        mesh_scene = scenes.gold_dragon_standin(n=24, grid_builder=lambda m: O.grid_build(m)[1])
        paths(mesh_scene, Settings(scenes.camera(480, 270), sample_count=1, bounce_limit=5, seed=scenes.SEED), 1024, 103, "paths_mesh.npz")
    if "image" in which:
        image_c1()
    if "grids" in which:
        grid_digests()
    if "png" in which:
        png_blocks()
    if "counters" in which:
        work_counters([w for w in which if w in scenes.CONFIGS])  # e.g. `gen_golden.py counters C4` recounts C4 only
    print("golden fixtures written to", GOLD)
