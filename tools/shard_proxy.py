"""Strong-scaling proxy on ONE GPU: time the share of rank r of an N-way run (tiles[r::N]) against the full frame.
python tools/shard_proxy.py [C2|C3] [spp]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raymond_amd import render, scenes, shard
from raymond_amd.scene import generate_tiles

name = sys.argv[1] if len(sys.argv) > 1 else "C2"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 500
st = scenes.config_settings(name, spp=spp)
cam = st.camera_settings
sc = getattr(scenes, scenes.CONFIGS[name][0])()
tiles = generate_tiles(cam.backbuffer_width, cam.backbuffer_height, st.tile_size)
with render.Context(0) as ctx:
    ds = render.DeviceScene(ctx, sc)
    fb = render.Framebuffer(ctx, cam.backbuffer_width, cam.backbuffer_height)
    def t(share):
        best = 1e9
        for _ in range(3):
            fb.zero()
            render.render_tiles(ctx, ds, cam, st, share, fb)
            best = min(best, ctx.last_kernel_ms())
        return best
    full = t(tiles)
    print("%s spp=%d full frame %.2f ms" % (name, spp, full), flush=True)
    for n in (2, 4, 8):
        ms = [t(shard.shard_tiles(tiles, r, n)) for r in range(n)]
        worst = max(ms)
        print("N=%d: slowest rank %.2f ms (ranks %s) -> speedup %.2fx, efficiency %.0f%%" % (n, worst, " ".join("%.1f" % m for m in ms), full / worst, 100 * full / worst / n), flush=True)
