"""Large per-sample parity campaign (not a test: takes a minute on 16 cores):  python tools/parity_campaign.py [n_per_config]
For every configuration, n random (pixel, sample) pairs are traced on the GPU and by the oracle; prints the share of samples
that keep the oracle's hit sequence and, among those, the worst relative radiance difference."""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT), sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib
from raymond_amd import probe, render, scenes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
oracle_lib.load()
with render.Context(0) as ctx:
    for config in ("C1", "C2", "C3", "C4", "C5"):
        st = scenes.config_settings(config)
        cam = st.camera_settings
        sc = getattr(scenes, scenes.CONFIGS[config][0])()
        rng = np.random.default_rng(1234)
        W, H = cam.backbuffer_width, cam.backbuffer_height
        xy = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], axis=1).astype(np.uint32)
        smp = rng.integers(0, st.sample_count, n).astype(np.uint32)
        ds, osc = render.DeviceScene(ctx, sc), oracle_lib.OracleScene(sc)
        t0 = time.time()
        drgb, dpo, dps = probe.trace_samples(ctx, ds, cam, st, xy, smp, paths=True)
        ds.close()
        def one(i):
            rgb, po, ps = osc.trace_sample_path(cam, st, int(xy[i, 0]), int(xy[i, 1]), int(smp[i]))
            k = len(po)
            return rgb, bool((dpo[i, :k] == po).all() and (dps[i, :k] == ps).all() and (dpo[i, k:] == -2).all())
        with ThreadPoolExecutor(16) as ex:
            res = list(ex.map(one, range(n), chunksize=2048))
        orgb = np.array([r[0] for r in res]); same = np.array([r[1] for r in res])
        a, b = drgb[same], orgb[same]
        rel = np.abs(a - b) / np.maximum(np.maximum(np.abs(a), np.abs(b)), 1e-300)
        rel[a == b] = 0
        exact = (a == b).all(axis=1).mean()
        print("%s: %d samples, same hit sequence %.4f %%, of those bit-identical radiance %.2f %%, worst relative difference %.3g  (%.0f s)" %
              (config, n, 100 * same.mean(), 100 * exact, rel.max(), time.time() - t0), flush=True)
