"""Pixels that differ between rmd_settings.flags 0 (reference-identical: every path traced to its end on a mesh scene) and
RMD_RENDER_END_BLACK_PATHS, at the FULL sample counts of BASELINE.json's mesh configurations:  python tools/count_mode_diffs.py OUT.json [C3 C4 C5]
Two GPU renders per configuration.  A pixel may differ only where the traced-on frame is non-finite (a NaN met behind a zero weight);
every other pixel must be bit-identical — the script fails otherwise."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raymond_amd import render, scenes
from raymond_amd.scene import generate_tiles

out_path = sys.argv[1]
names = sys.argv[2:] or ["C3", "C4", "C5"]
result = {}
with render.Context(0) as ctx:
    for name in names:
        st = scenes.config_settings(name)
        cam = st.camera_settings
        W, H = cam.backbuffer_width, cam.backbuffer_height
        sc = getattr(scenes, scenes.CONFIGS[name][0])()
        tiles = generate_tiles(W, H, st.tile_size)
        ds = render.DeviceScene(ctx, sc)
        fb = render.Framebuffer(ctx, W, H)
        frames, ms = {}, {}
        for mode in ("default", "end"):
            st.end_black_paths = mode == "end"
            fb.zero()
            render.render_tiles(ctx, ds, cam, st, tiles, fb)
            ms[mode] = ctx.last_kernel_ms()
            frames[mode] = fb.download()
            print(name, mode, "%.1f ms" % ms[mode], "passes", ctx.last_launch_info().passes, flush=True)
        fb.close(), ds.close()
        a, b = frames["default"], frames["end"]
        same = ((a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))).all(axis=2)
        nonfinite_default = (~np.isfinite(a)).any(axis=2)
        nonfinite_end = (~np.isfinite(b)).any(axis=2)
        differ = ~same
        ys, xs = np.nonzero(differ)
        entry = {
            "frame": "%dx%d, %d spp, %d bounces%s" % (W, H, st.sample_count, st.bounce_limit, ", thin lens" if st.use_dof else ""),
            "samples": int(W) * int(H) * int(st.sample_count),
            "pixels_that_differ": int(differ.sum()),
            "of_those_finite_in_the_reference_identical_frame": int((differ & ~nonfinite_default).sum()),
            "non_finite_pixels_flags_0": int(nonfinite_default.sum()),
            "non_finite_pixels_end_black_paths": int(nonfinite_end.sum()),
            "differing_pixels_xy": [[int(x), int(y)] for x, y in zip(xs[:32], ys[:32])],
            "kernel_ms": {k: round(v, 1) for k, v in ms.items()},
            "seed": st.seed,
        }
        result[name] = entry
        print(json.dumps(entry), flush=True)
        assert entry["of_those_finite_in_the_reference_identical_frame"] == 0, "a pixel that is finite in the reference-identical frame changed"
with open(out_path, "w") as f:
    json.dump(result, f, indent=1)
