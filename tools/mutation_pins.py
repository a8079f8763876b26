"""Which of the reference's quirks does its own render actually pin?

Renders the scene of examples/ReflectiveSpheres.png (592x340, 500 spp, 5 bounces) with the faithful oracle and with each single-quirk
"repair" compiled into oracle.cpp (MUT_*, selected by orc_set_mutation — test infrastructure, never set by a test or the product), runs
tests/png_pin.py's checks on each and prints a markdown table: mutation -> rejected (by which checks) / not detected.  Also the one
scene mutation SURVEY.md section 8d names: the sphere placement of server/src/main.rs:79-84.

    python tools/mutation_pins.py [--out profiles/r03_mutation_pins.md]        (CPU only; ~15 s per row on 8 cores)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--only", type=int, nargs="*", default=None, help="mutation numbers to run (-1 = the scene mutation); default: all")
    args = ap.parse_args()
    L = oracle_lib.load()
    n = L.orc_set_mutation(0)
    assert n == len(MUTATIONS), "oracle.cpp's MUT_COUNT and this table disagree"
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
