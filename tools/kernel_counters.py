"""What the mesh kernel itself counts on a launch (DIAG build, RMD_DEBUG = 8), per sample -> tests/golden/kernel_counters.json.

    python tools/kernel_counters.py [C3] [spp] [--out tests/golden/kernel_counters.json]        (GPU box; needs raymond_amd/csrc/diag/libraymond_hip.so)

The oracle's work counters (tests/golden/work_counters.json) are the REFERENCE's work: cells visited, triangle tests, shaded mesh hits.  These are the
kernel's own: the (ray, triangle) pairs its rounds number (each a 4-byte index and a 32-byte sphere), the pairs that pass the sphere pre-test (each a
72-byte record), and the paths its queues move (a ray 96 bytes each way, a hit 88) — what bench.py's roofline.bytes_requested prices.  Deterministic
for a given build (the same pairs whatever the scheduling? no: speculative candidates depend on which rays share a round — the figures move in the
fourth digit from run to run)."""
import json, os, re, subprocess, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
name = args[0] if args else "C3"
spp = int(args[1]) if len(args) > 1 else 50
out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else os.path.join(root, "tests", "golden", "kernel_counters.json")
res = {}
for mode in ("", "-end"):
    env = dict(os.environ, RAYMOND_HIP_LIB=os.path.join(root, "raymond_amd", "csrc", "diag", "libraymond_hip.so"), RMD_DEBUG="8")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "quick_time.py"), name + mode, str(spp), "1"], capture_output=True, text=True, env=env, timeout=600)
    text = r.stderr + r.stdout
    vals = {k: int(v) for k, v in re.findall(r"([a-z_ ]+[a-z])=(\d+)", text)}
    g = lambda k: [v for kk, v in vals.items() if kk.strip().endswith(k)][0]
    W, H = (3840, 2160) if name == "C4" else (1920, 1080)
    n = float(W * H * spp)
    res[name + ("_end" if mode else "_reference")] = {
        "spp": spp, "samples": int(n),
        "pairs_numbered_per_sample": g("tests") / n, "pairs_passing_the_pre_test_per_sample": g("pairs passed") / n,
        "walk_calls_per_sample": g("walk_calls") / n, "rays_per_walk_call": g("walkers") / max(1, g("walk_calls")), "rounds_per_sample": g("rounds") / n,
        "stepping_iterations_per_sample": g("wave_steps") / n, "lanes_per_stepping_iteration": g("lane_steps") / max(1, g("wave_steps")),
        "chunks_per_sample": g("chunks") / n, "trips_per_sample": g("main_iterations") / n, "lanes_per_trip": g("live_lanes") / max(1, g("main_iterations")),
        "rays_pushed_per_sample": g("rays pushed") / n, "walks_put_aside_per_sample": g("of them walks put aside") / n,
        "hits_pushed_per_sample": g("hits pushed") / n, "hits_held_per_sample": g("hits held in their lanes") / n,
    }
    print(name + mode, json.dumps(res[name + ("_end" if mode else "_reference")]))
old = json.load(open(out)) if os.path.exists(out) else {}
old.update(res)
json.dump(old, open(out, "w"), indent=1, sort_keys=True)
