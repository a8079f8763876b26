import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raymond_amd import abi, render, scenes, shard
from raymond_amd.scene import generate_tiles
name = sys.argv[1]; n = int(sys.argv[2])
st = scenes.config_settings(name, spp=500); cam = st.camera_settings
sc = getattr(scenes, scenes.CONFIGS[name][0])()
tiles = generate_tiles(cam.backbuffer_width, cam.backbuffer_height, st.tile_size)
share = shard.shard_tiles(tiles, 0, n)
with render.Context(0) as ctx:
    ds = render.DeviceScene(ctx, sc); fb = render.Framebuffer(ctx, cam.backbuffer_width, cam.backbuffer_height)
    for k in (0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16):
        ctx.set_tunable(abi.RMD_TUNE_SAMPLE_SPLIT, k)
        best = 1e9
        for _ in range(3):
            fb.zero(); render.render_tiles(ctx, ds, cam, st, share, fb); best = min(best, ctx.last_kernel_ms())
        print(name, "N=%d share, split %d: %.2f ms" % (n, k, best), flush=True)
