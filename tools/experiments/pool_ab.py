"""Walk pool on / off on the full C3 frame (python tools/experiments/pool_ab.py [spp]): kernel time of each, and the frames must be bit-identical."""
# NEEDS tools/experiments/walk_pool.patch applied (git apply) and the library rebuilt: the tunable it switches (abi.RMD_TUNE_*) exists only in that patch.
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))  # tools/experiments/ -> repo root
from raymond_amd import abi, render, scenes
from raymond_amd.scene import generate_tiles

name = sys.argv[2] if len(sys.argv) > 2 else "C3"
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 100
st = scenes.config_settings(name, spp=spp)
cam = st.camera_settings
sc = getattr(scenes, scenes.CONFIGS[name][0])()
tiles = generate_tiles(cam.backbuffer_width, cam.backbuffer_height, st.tile_size)
with render.Context(0) as ctx:
    ds = render.DeviceScene(ctx, sc)
    fb = render.Framebuffer(ctx, cam.backbuffer_width, cam.backbuffer_height)
    frames = {}
    for pool in (1, 0, 1, 0):
        ctx.set_tunable(abi.RMD_TUNE_WALK_POOL, pool)
        best = 1e9
        for it in range(3):
            fb.zero()
            render.render_tiles(ctx, ds, cam, st, tiles, fb)
            best = min(best, ctx.last_kernel_ms())
        frames.setdefault(pool, fb.download().tobytes())
        print("%s spp=%d walk pool %s: best kernel %.1f ms" % (name, spp, "off" if pool else "on", best), flush=True)
    print("bit-identical:", frames[0] == frames[1])
    sys.exit(0 if frames[0] == frames[1] else 1)
