"""Which of the reference's quirks does its own render actually pin?

Renders the scene of examples/ReflectiveSpheres.png (592x340, 500 spp, 5 bounces) with the faithful oracle and with each single-quirk
"repair" compiled into oracle.cpp (MUT_*, selected by orc_set_mutation — test infrastructure, never set by a test or the product), runs
tests/png_pin.py's checks on each and prints a markdown table: mutation -> rejected (by which checks) / not detected.  Also the one
scene mutation SURVEY.md section 8d names: the sphere placement of server/src/main.rs:79-84.

    python tools/mutation_pins.py [--out profiles/r03_mutation_pins.md]        (CPU only; ~15 s per row on 8 cores)

Round 6: `--batches B` renders each row's EXPECTATION instead — B independent 500-spp renders (sample ranges [500 b, 500 b + 500)), each tone-mapped
and truncated to 8 bits the way the PNG was, block means averaged over the batches: the expectation of what a 500-spp render LOOKS like (the tone map
is concave: the expectation of the tone-mapped noisy image, not the tone map of the expectation), known to 1/sqrt(B) of the PNG's own noise, and that
noise itself from the batch-to-batch scatter.  The statistic's variance is then the PNG's alone (x (1 + 1/B)), where the two-halves form above
carries as much noise of our own as of the PNG's.  Checks: the pooled squared distance against the pooled variance, region biases at 5 sigma, the
sphere masks.  For every mutation the table also gives its EFFECT — the region bias it causes against the faithful restatement's expectation — and the
sample count at which the PNG's 5 sigma would equal it: what the PNG would have needed to see a survivor.

    python tools/mutation_pins.py --batches 10 --out profiles/r06_mutation_pins.md      (~1 min per row on 8 cores)
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib  # noqa: E402
import png_pin  # noqa: E402
from raymond_amd import scenes  # noqa: E402

MUTATIONS = [
    (0, "none (faithful restatement)", ""),
    (1, "Q1: view vector = -ray direction at every depth", "src/trace.rs:256"),
    (2, "Q2: diffuse pdf = sqrt(r1)/PI (BRDF still without 1/PI)", "src/trace.rs:399-401"),
    (3, "Q2: truly uniform hemisphere sampler, pdf 1/2 (an unbiased alternative: same expectation)", "src/trace.rs:396-406"),
    (4, "Q3: GGX polar angle through atan", "src/trace.rs:289"),
    (5, "Q4: a2 = roughness^4 in the GGX distribution", "src/trace.rs:363"),
    (6, "Q4: k = (r+1)^2/8 in Schlick-GGX", "src/trace.rs:374"),
    (7, "Q4: without the + 0.001 / + 0.0001 guards", "src/trace.rs:312,316"),
    (8, "Q4: specular cos_theta clamped to >= 0", "src/trace.rs:306"),
    (9, "prob_d = 1 - metalness (no specular lobe on Diffuse)", "src/trace.rs:263"),
    (10, "F0 = 0 for dielectrics instead of 0.04", "src/trace.rs:257"),
    (11, "ray offsets 1e-3 instead of 1e-5 / 1e-4", "src/trace.rs:269,300"),
    (12, "Q10: far root when the origin is inside a sphere", "sphere.rs:21-24"),
    (13, "Q11: two-sided planes", "plane.rs:14"),
    (14, "Q14: one more path segment (depth starts at 0)", "src/trace.rs:200,235"),
    (15, "Q14: emission only towards the front side", "src/trace.rs:250-252"),
]


def server_scene():
    """The sphere placement of server/src/main.rs:79-84 (x = -1.5 / 1.25), which SURVEY.md section 8d says does NOT match the PNG."""
    sc = scenes.reflective_spheres()
    for obj, x in ((sc.objects[0], -1.5), (sc.objects[1], 1.25)):
        obj.geometry.origin = (x,) + tuple(obj.geometry.origin[1:])
    return sc


REGIONS = None


def regions():
    import numpy as np

    yy, xx = np.mgrid[0:42, 0:74]
    return {
        "frame": np.ones((42, 74), dtype=bool),
        "red diffuse sphere": (xx * 8 + 4 - 202.2) ** 2 + (yy * 8 + 4 - 216.2) ** 2 < 38.0**2,
        "blue metal sphere with its reflections": (xx * 8 + 4 - 364.5) ** 2 + (yy * 8 + 4 - 192.8) ** 2 < 60.0**2,
        "floor": yy >= 36,
        "back wall above the spheres": (yy >= 10) & (yy < 14) & (xx > 20) & (xx < 55),
    }


def expectation(L, k, batches, threads):
    """-> (mean, std) over `batches` independent 500-spp renders of the 8x8 block means of the tone-mapped 8-bit frame, float64[42, 74, 3] each."""
    import numpy as np
    from raymond_amd.scene import Settings, generate_tiles

    L.orc_set_mutation(max(k, 0))
    try:
        osc = oracle_lib.OracleScene(server_scene() if k < 0 else scenes.reflective_spheres())
        st = Settings(scenes.camera(592, 340), sample_count=500, tile_size=(32, 32), bounce_limit=5, seed=scenes.SEED)
        tiles = generate_tiles(592, 340, (32, 32))
        blk = []
        for b in range(batches):
            img = osc.render_tiles(st.camera_settings, st, tiles, sample_begin=500 * b, sample_count=500, threads=threads)
            blk.append(png_pin.blocks(oracle_lib, img, 500))
    finally:
        L.orc_set_mutation(0)
    blk = np.stack(blk)
    return blk.mean(axis=0), blk.std(axis=0, ddof=1)


def main_expectation(args, L):
    import numpy as np

    ref = np.load(os.path.join(png_pin.GOLD, "png_blocks.npy")).astype(np.float64)
    B = args.batches
    reg = regions()
    rows, base = [], None
    for k, label, where in MUTATIONS + [(-1, "scene: spheres at x = -1.5 / 1.25 (server/src/main.rs:79-84)", "cli_old/src/main.rs:48-58")]:
        if args.only is not None and k not in args.only and k != 0:
            continue
        E, S = expectation(L, k, B, args.threads)
        if k == 0:
            base = (E, S)
        var = S**2 * (1.0 + 1.0 / B)  # variance of PNG block - expectation estimate, under "this row is what the reference computes"
        d = ref - E
        live = S > 0.02  # (saturated blocks — the emitter — have no noise: compared exactly below)
        ratio = float((d[live] ** 2).sum() / var[live].sum())
        n_live = int(live.sum())
        # pooled-variance statistic: E = 1; its own scatter ~ sqrt(2 / n) x (1 + kurtosis terms) — and the variance estimate from B batches adds
        # sqrt(2 / (B - 1) / n): both ~ 0.02 here; 8-bit truncation of the PNG adds < 0.01 of a block's variance.  Rejected above 1.15.
        failed = []
        if not ratio <= 1.15:
            failed.append("pooled squared distance / pooled variance = %.3f > 1.15" % ratio)
        if not ratio >= 0.87:
            # two-sided: the PNG scatters LESS about this row's expectation than this row's own noise allows — the row's renders are noisier than the
            # reference's (a sampler with the same expectation and another variance shows here and nowhere else)
            failed.append("pooled squared distance / pooled variance = %.3f < 0.87: the PNG is less noisy than this restatement's renders" % ratio)
        worst = (None, 0.0)
        for name, m in reg.items():
            mm = m[:, :, None] & live
            bias = np.array([d[:, :, c][mm[:, :, c]].mean() if mm[:, :, c].any() else 0.0 for c in range(3)])
            sig = np.array([np.sqrt(var[:, :, c][mm[:, :, c]].sum()) / max(1, int(mm[:, :, c].sum())) for c in range(3)])
            z = np.abs(bias) / np.maximum(5.0 * sig + 0.05, 1e-9)
            if z.max() > 1.0:
                failed.append("bias: %s (%.2f of 255 where 5 sigma + 0.05 = %.2f)" % (name, np.abs(bias)[z.argmax()], (5.0 * sig + 0.05)[z.argmax()]))
        sat = S == 0.0  # every batch gave the same block: deterministic (the emitter)
        if not (np.abs(d[sat]) <= 0.51).all():  # a block without noise (the emitter: 227 exactly) must equal the PNG's
            failed.append("a noise-free block differs from the PNG by %.2f" % np.abs(d[sat]).max())
        for ch, other, lab in ((0, 2, "red"), (2, 0, "blue")):
            def centroid(blk):
                m = (blk[:, :, ch] > 1.6 * blk[:, :, other] + 20) & (blk[:, :, ch] > 1.6 * blk[:, :, 1])
                ys, xs = np.nonzero(m)
                return (np.array([xs.mean(), ys.mean()]) if m.any() else np.array([np.inf, np.inf])), int(m.sum())
            (c_ref, n_ref), (c_our, n_our) = centroid(ref), centroid(E)
            if not (n_ref > 10 and abs(n_our - n_ref) <= 0.1 * n_ref and np.abs(c_ref - c_our).max() < 0.5):
                failed.append("%s sphere: mask size and centroid" % lab)
        # the mutation's effect against the faithful restatement, region by region, in units of what the PNG resolves at its 500 spp (5 sigma + 0.05)
        effect = "-"
        if k != 0 and base is not None:
            E0, S0 = base
            best = (0.0, "", 0.0, 0.0)
            for name, m in reg.items():
                mm = m[:, :, None] & live & (S0 > 0.02)
                for c in range(3):
                    if not mm[:, :, c].any():
                        continue
                    eff = float((E - E0)[:, :, c][mm[:, :, c]].mean())
                    thr = float(5.0 * np.sqrt((S0[:, :, c][mm[:, :, c]] ** 2).sum()) / int(mm[:, :, c].sum()) + 0.05)
                    if abs(eff) / thr > best[0]:
                        best = (abs(eff) / thr, name + " (%s)" % "rgb"[c], eff, thr)
            need = 500.0 / max(best[0], 1e-9) ** 2 if best[0] > 0 else float("inf")
            effect = "%+.3f of 255 in %s = %.2f of the PNG's threshold %.3f; a render of ~%s spp would resolve it" % (
                best[2], best[1], best[0], best[3], "%.0f" % need if need < 1e9 else "> 10^9")
        verdict = "rejected" if failed else ("passes" if k == 0 else "NOT detected")
        rows.append((label, where, verdict, "; ".join(failed), "%.3f (%d block values)" % (ratio, n_live), effect))
        print("%-90s %-12s %s | %s | %s" % (label, verdict, rows[-1][4], rows[-1][3], effect), flush=True)
    lines = ["| mutation (one at a time) | reference lines | verdict of the PNG pin | failed checks | squared distance / variance (1 expected) | largest effect against the faithful restatement |",
             "|---|---|---|---|---|---|"]
    lines += ["| %s | `%s` | **%s** | %s | %s | %s |" % (a, b, c, d or "-", e, f) for a, b, c, d, e, f in rows]
    n_rej = sum(1 for r in rows[1:-1] if r[2] == "rejected")
    head = ("# What examples/ReflectiveSpheres.png pins, with a noise-free expectation (tools/mutation_pins.py --batches %d; oracle at 592x340, %d x 500 spp)\n\n"
            "Each row's expectation is the mean of %d independent 500-spp renders, tone-mapped and truncated like the PNG; the yardstick is the batch-to-batch scatter.\n"
            "Rejected single-quirk mutations: **%d of %d** (round 3's two-halves form: 9 of 16).\n\n" % (B, B, B, n_rej, len(rows) - 2))
    text = "\n".join(lines) + "\n"
    if args.out:
        with open(os.path.join(ROOT, args.out), "w") as f:
            f.write(head + text)
    print(head + text)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=0, help="> 0: the noise-free expectation form (round 6)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--only", type=int, nargs="*", default=None, help="mutation numbers to run (-1 = the scene mutation); default: all")
    args = ap.parse_args()
    L = oracle_lib.load()
    n = L.orc_set_mutation(0)
    assert n == len(MUTATIONS), "oracle.cpp's MUT_COUNT and this table disagree"
    if args.batches > 0:
        return main_expectation(args, L)
    rows = []
    for k, label, where in MUTATIONS + [(-1, "scene: spheres at x = -1.5 / 1.25 (server/src/main.rs:79-84)", "cli_old/src/main.rs:48-58")]:
        if args.only is not None and k not in args.only:
            continue
        L.orc_set_mutation(max(k, 0))
        try:
            halves = png_pin.render_halves(oracle_lib, scene=server_scene() if k < 0 else None, threads=args.threads)
        finally:
            L.orc_set_mutation(0)
        res = png_pin.run_checks(oracle_lib, halves)
        failed = [name for name, ok, _ in res if not ok]
        ratio = [d for name, _, d in res if name.startswith("mean distance")][0]
        rows.append((label, where, "rejected" if failed else ("passes" if k == 0 else "NOT detected"), "; ".join(failed), ratio))
        print("%-90s %-12s %s | %s" % (label, rows[-1][2], ratio, rows[-1][3]), flush=True)
    lines = ["| mutation (one at a time) | reference lines | verdict of the PNG pin | failed checks | block distance: render-PNG vs half-to-half |", "|---|---|---|---|---|"]
    lines += ["| %s | `%s` | **%s** | %s | %s |" % (a, b, c, d or "-", e) for a, b, c, d, e in rows]
    text = "\n".join(lines) + "\n"
    if args.out:
        with open(os.path.join(ROOT, args.out), "w") as f:
            f.write("# What examples/ReflectiveSpheres.png pins (tools/mutation_pins.py; oracle at 592x340, 500 spp, seed 0x5EED0001)\n\n" + text)
    print(text)


if __name__ == "__main__":
    main()
