"""The full C3 frame in every launch form (path queues or a lane per path; 0 = persistent workgroups, 1 = one wave per work item), several splits and several values of the walk cut
(RMD_TUNE_WALK_CUT: 0 = the library's K, 1 = every walk call finishes its walks, 9 = K 8), N frames each: every frame must equal the first one bit
for bit (work items are drawn in a different order on every run, and with them which walks are put aside when).  python tools/stress_modes.py [frames] [spp]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raymond_amd import abi, render, scenes
from raymond_amd.scene import generate_tiles

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 5
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 40
st = scenes.config_settings("C3", spp=spp)
cam = st.camera_settings
sc = getattr(scenes, scenes.CONFIGS["C3"][0])()
tiles = generate_tiles(cam.backbuffer_width, cam.backbuffer_height, st.tile_size)
with render.Context(0) as ctx:
    ds = render.DeviceScene(ctx, sc)
    fb = render.Framebuffer(ctx, cam.backbuffer_width, cam.backbuffer_height)
    want, bad = None, 0
    for queues, mode in ((0, 0), (1, 0), (1, 1)):   # the path queues need the persistent form, so two lane-per-path forms and the queued one
        ctx.set_tunable(abi.RMD_TUNE_PATH_QUEUES, queues)
        for split in (0, 2, 7):
          for cut in (0, 1, 9):
            ctx.set_tunable(abi.RMD_TUNE_LAUNCH_FORM, mode), ctx.set_tunable(abi.RMD_TUNE_SAMPLE_SPLIT, split), ctx.set_tunable(abi.RMD_TUNE_WALK_CUT, cut)
            for i in range(frames):
                fb.zero()
                render.render_tiles(ctx, ds, cam, st, tiles, fb)
                got = fb.download().tobytes()
                if want is None:
                    want = got
                elif got != want:
                    bad += 1
                    print("MISMATCH mode", mode, "split", split, "cut", cut, "frame", i, flush=True)
            print("%s mode %d split %d cut %d: %d frames, %d mismatches so far" % (
                "queued" if ctx.last_launch_info().queued else "lane-per-path", mode, split, cut, frames, bad), flush=True)
    print("stress: %s" % ("FAILED" if bad else "ok"))
    sys.exit(1 if bad else 0)
