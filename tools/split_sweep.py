"""Sample-split sweep on ONE GPU: kernel time of rank 0's share of an N-way run for forced split factors (0 = the library's choice).
python tools/split_sweep.py [C2|C3] [spp] [N,N,...] [k,k,...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raymond_amd import abi, render, scenes, shard
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
    ns = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 4, 8]
    ks = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0, 2, 4, 6, 8, 12, 16, 24]
    for n in ns:
        share = shard.shard_tiles(tiles, 0, n)
        for k in ks:
            ctx.set_tunable(abi.RMD_TUNE_SAMPLE_SPLIT, k)
            best = 1e9
            for _ in range(3):
                fb.zero()
                render.render_tiles(ctx, ds, cam, st, share, fb)
                best = min(best, ctx.last_kernel_ms())
            print("%s N=%d split=%d: %.2f ms" % (name, n, k, best), flush=True)
