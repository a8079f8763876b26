"""Finds non-finite pixels of a configuration's frame on the GPU, then the samples that cause them, and shows what the oracle
returns for the same (pixel, sample):  python tools/find_nan.py [C3] [spp]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib
from raymond_amd import probe, render, scenes
from raymond_amd.scene import generate_tiles

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 500
st = scenes.config_settings(name, spp=spp)
cam = st.camera_settings
sc = getattr(scenes, scenes.CONFIGS[name][0])()
W, H = cam.backbuffer_width, cam.backbuffer_height
with render.Context(0) as ctx:
    ds = render.DeviceScene(ctx, sc)
    fb = render.Framebuffer(ctx, W, H)
    render.render_tiles(ctx, ds, cam, st, generate_tiles(W, H, st.tile_size), fb)
    img = fb.download()
    bad = np.argwhere(~np.isfinite(img).all(axis=2))
    print("non-finite pixels:", len(bad))
    osc = oracle_lib.OracleScene(sc)
    for y, x in bad[:8]:
        xy = np.tile(np.array([[x, y]], dtype=np.uint32), (spp, 1))
        smp = np.arange(spp, dtype=np.uint32)
        d, po, ps = probe.trace_samples(ctx, ds, cam, st, xy, smp, paths=True)
        o = osc.trace_samples(cam, st, xy, smp)
        for s in np.nonzero(~np.isfinite(d).all(axis=1) | ~np.isfinite(o).all(axis=1))[0]:
            rgb, opo, ops = osc.trace_sample_path(cam, st, int(x), int(y), int(s))
            print("pixel", (int(x), int(y)), "sample", int(s), "gpu", d[s], "oracle", o[s], "gpu path", po[s][:len(opo)+1], ps[s][:len(opo)+1], "oracle path", opo, ops)
