"""Scratch timing of the render kernel (not the bench): python tools/quick_time.py [C2|C3|C4|C5][-end|-trace] [spp]
The suffix selects rmd_settings.flags: none = 0 (reference-identical), -end = RMD_RENDER_END_BLACK_PATHS, -trace = RMD_RENDER_TRACE_BLACK_PATHS."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raymond_amd import render, scenes
from raymond_amd.scene import generate_tiles

arg = sys.argv[1] if len(sys.argv) > 1 else "C2"
name, _, mode = arg.partition("-")
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 50
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
st = scenes.config_settings(name, spp=spp)
st.end_black_paths, st.trace_black_paths = mode == "end", mode == "trace"
cam = st.camera_settings
sc = getattr(scenes, scenes.CONFIGS[name][0])()
tiles = generate_tiles(cam.backbuffer_width, cam.backbuffer_height, st.tile_size)
with render.Context(0) as ctx:
    ds = render.DeviceScene(ctx, sc)
    fb = render.Framebuffer(ctx, cam.backbuffer_width, cam.backbuffer_height)
    for it in range(reps):
        fb.zero()
        t = time.time()
        render.render_tiles(ctx, ds, cam, st, tiles, fb)
        dt = time.time() - t
        ms = ctx.last_kernel_ms()
        n = cam.backbuffer_width * cam.backbuffer_height * spp
        print("%s spp=%d wall %.3fs kernel %.1f ms -> %.1f Msamples/s" % (arg, spp, dt, ms, n / ms / 1e3), flush=True)
    img = fb.download()
    import numpy as np
    print("mean radiance %.9g  (pixels with a non-finite channel: %d)" % (np.nanmean(img) / spp, int((~np.isfinite(img)).any(axis=2).sum())))
