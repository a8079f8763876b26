"""Path queues on / off on one workload: python tools/queue_ab.py [C3|C4|C5][-end|-trace] [spp] [reps]
Renders the frame with RMD_TUNE_PATH_QUEUES = 1 (a lane keeps its path: render_wave) and 0 (the library's choice: render_wave_queued),
compares the two frames bit for bit (NaN = NaN) and prints each form's kernel time and launch info."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raymond_amd import abi, render, scenes
from raymond_amd.scene import generate_tiles

arg = sys.argv[1] if len(sys.argv) > 1 else "C3"
name, _, mode = arg.partition("-")
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 50
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
st = scenes.config_settings(name, spp=spp)
st.end_black_paths, st.trace_black_paths = mode == "end", mode == "trace"
cam = st.camera_settings
sc = getattr(scenes, scenes.CONFIGS[name][0])()
tiles = generate_tiles(cam.backbuffer_width, cam.backbuffer_height, st.tile_size)
frames = {}
with render.Context(0) as ctx:
    ds = render.DeviceScene(ctx, sc)
    fb = render.Framebuffer(ctx, cam.backbuffer_width, cam.backbuffer_height)
    for form, tun in (("lane-per-path", 1), ("queued", 0)):
        ctx.set_tunable(abi.RMD_TUNE_PATH_QUEUES, tun)
        best = None
        for it in range(reps):
            fb.zero()
            render.render_tiles(ctx, ds, cam, st, tiles, fb)
            ms = ctx.last_kernel_ms()
            best = ms if best is None or ms < best else best
        info = ctx.last_launch_info()
        n = cam.backbuffer_width * cam.backbuffer_height * spp
        print("%s spp=%d %-14s %.2f ms -> %.1f Msamples/s (queued=%d chained=%d split_k=%d waves/wg=%d passes=%d)" % (
            arg, spp, form, best, n / best / 1e3, info.queued, info.chained, info.split_k, info.waves_per_workgroup, info.passes), flush=True)
        frames[form] = fb.download()
    a, b = frames["lane-per-path"], frames["queued"]
    same = (a == b) | (np.isnan(a) & np.isnan(b))
    print("frames bit-identical: %s (%d of %d values differ; max |diff| %.3g; non-finite pixels %d / %d)" % (
        bool(same.all()), int((~same).sum()), same.size, float(np.nanmax(np.abs(a - b))) if not same.all() else 0.0,
        int((~np.isfinite(a)).any(axis=2).sum()), int((~np.isfinite(b)).any(axis=2).sum())))
    sys.exit(0 if same.all() else 1)
