"""Which samples put a pixel of a production frame outside 1e-9 of the oracle's: python tools/off_pixel_report.py [C2|C3|C4|C5][-end] SPP
Renders the configuration's full frame through rmd_render_tiles and through the oracle's render_tiles, re-traces every sample of every off pixel
on both sides (list probe / orc_trace_sample) and prints, per off sample: pixel, sample, whether the vertex sequence is the same, both values and
their relative distance."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root), sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import oracle_lib
from raymond_amd import probe, render, scenes
from raymond_amd.scene import generate_tiles

arg, spp = sys.argv[1], int(sys.argv[2])
name, _, mode = arg.partition("-")
st = scenes.config_settings(name, spp=spp)
st.end_black_paths = mode == "end"
cam = st.camera_settings
W, H = cam.backbuffer_width, cam.backbuffer_height
sc = getattr(scenes, scenes.CONFIGS[name][0])()
tiles = generate_tiles(W, H, st.tile_size)
with render.Context(0) as ctx:
    ds, fb = render.DeviceScene(ctx, sc), render.Framebuffer(ctx, W, H)
    render.render_tiles(ctx, ds, cam, st, tiles, fb)
    dev = fb.download()
    osc = oracle_lib.OracleScene(sc, fast=True)
    ref = osc.render_tiles(cam, st, tiles, threads=16)
    def close(a, b, tol):
        return (np.abs(a - b) <= tol * np.maximum(np.maximum(np.abs(a), np.abs(b)), 1e-300)) | (a == b) | (np.isnan(a) & np.isnan(b))
    ok = close(dev, ref, 1e-9).all(axis=2)
    ys, xs = np.nonzero(~ok)
    print("%s %d spp: %d of %d pixels outside 1e-9" % (arg, spp, len(ys), ok.size))
    xy = np.repeat(np.stack([xs, ys], axis=1), spp, axis=0).astype(np.uint32)
    smp = np.tile(np.arange(spp, dtype=np.uint32), len(ys))
    if len(ys):
        drgb, dpo, dps = probe.trace_samples(ctx, ds, cam, st, xy, smp, paths=True)
        osl = oracle_lib.OracleScene(sc)
        for i in range(len(smp)):
            rgb, po, ps = osl.trace_sample_path(cam, st, int(xy[i, 0]), int(xy[i, 1]), int(smp[i]))
            k = len(po)
            same = bool((dpo[i, :k] == po).all() and (dps[i, :k] == ps).all() and (dpo[i, k:] == -2).all())
            if not close(drgb[i], rgb, 1e-9).all():
                rel = np.max(np.abs(drgb[i] - rgb) / np.maximum(np.maximum(np.abs(drgb[i]), np.abs(rgb)), 1e-300))
                print("pixel (%d, %d) sample %d same_path %s path %s rel %.3e\n   dev %s\n   orc %s" % (xy[i, 0], xy[i, 1], smp[i], same, list(po), rel, drgb[i], rgb))
    ds.close(), fb.close()
