"""Stress of the spheres kernel's in-kernel ordered sum (cross-XCD release / acquire): the full C2 frame, rendered N times as a split
launch of persistent workgroups, must equal the direct-mode frame (one wave per tile, no sample buffer) bit for bit every time.
python tools/stress_sum.py [frames] [spp]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raymond_amd import abi, render, scenes
from raymond_amd.scene import generate_tiles

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 20
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
st = scenes.config_settings("C2", spp=spp)
cam = st.camera_settings
sc = scenes.reflective_spheres()
tiles = generate_tiles(cam.backbuffer_width, cam.backbuffer_height, st.tile_size)
with render.Context(0) as ctx:
    ds = render.DeviceScene(ctx, sc)
    fb = render.Framebuffer(ctx, cam.backbuffer_width, cam.backbuffer_height)
    ctx.set_tunable(abi.RMD_TUNE_SAMPLE_SPLIT, 1)
    fb.zero()
    render.render_tiles(ctx, ds, cam, st, tiles, fb)
    want = fb.download().tobytes()
    bad = 0
    for split in (0, 2, 4, 8, 16):
        ctx.set_tunable(abi.RMD_TUNE_SAMPLE_SPLIT, split)
        for i in range(frames):
            fb.zero()
            render.render_tiles(ctx, ds, cam, st, tiles, fb)
            if fb.download().tobytes() != want:
                bad += 1
                print("MISMATCH split", split, "frame", i, flush=True)
        print("split %d: %d frames done, %d mismatches so far" % (split, frames, bad), flush=True)
    print("stress: %s" % ("FAILED" if bad else "ok"))
    sys.exit(1 if bad else 0)
