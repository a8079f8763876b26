#!/usr/bin/env python3
"""Benchmark of the hot path: Msamples/s of the ReflectiveSpheres scene, 1920x1080 @ 500 spp, 5 bounces (config C2).

    python bench.py --gpus N --steps K --warmup W

A "step" is one full pass of the hot path over the workload: every pixel of the 1920x1080 frame receives all 500
samples (rmd_render_tiles over this rank's share of the 32x32 host tiles, sample_begin 0, sample_count 500) and, for
N > 1, rank 0 assembles the frame with ONE RCCL collective: a gather of the tiles each rank owns (default; 1/N of a
frame per rank) or --assemble reduce, a sum of the full f64 framebuffers.  The frame is a fixed amount of
work split over the ranks (tile i -> rank i mod N), so scaling is "strong".  Scene, camera and tile list are resident
in HBM before the timed region; the timed region is bracketed by barrier + synchronize and the maximum over ranks
is reported.

The JSON line also carries
  roofline      HBM roofline of the traversal (same kernel, GoldDragon-standin mesh, config C3 at its 500 spp, one launch): the
                spheres workload touches ~0.05 B/sample and is FP64-VALU bound, so its HBM fraction is reported
                separately as roofline_c2 and is not the figure to optimise;
  cpu_baseline  the CPU oracle (reference-equivalent C++ restatement, kind "port") timed on this box's host cores
                on a bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_ISSUE_NS = 1.667  # one wave64 FP64 / INT32 vector instruction per SIMD every 4 cycles at the 2.4 GHz peak clock


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="C2", choices=["C2", "C3", "C4", "C5"])
    ap.add_argument("--spp", type=int, default=None, help="override samples per pixel (default: the config's)")
    ap.add_argument("--roofline-spp", type=int, default=None, help="spp of the C3 roofline leg (default: the config's 500)")
    ap.add_argument("--roofline-steps", type=int, default=3)
    ap.add_argument("--cpu-spp", type=int, default=0, help="spp of the bounded CPU-baseline sample (0 = calibrate to ~15 s)")
    ap.add_argument("--assemble", default="gather", choices=["gather", "reduce", "abi"],
                    help="N > 1: how rank 0 gets the frame - gather of the tiles each rank owns (1/N of a frame per rank), "
                         "reduce(sum) of the full frames through torch.distributed, or abi: the same reduce through the C-ABI's own "
                         "rmd_comm_* / rmd_reduce_framebuffer (RCCL, one rank per GPU); all bit-identical to the 1-GPU frame")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline-leg", action="store_true")
    ap.add_argument("--lean", action="store_true", help="only the timed workload and the roofline leg in its default mode (tools/profile_round.sh: the rocprofv3 "
                                                         "kernel-trace then averages each kernel over one workload, as the bench line does)")
    return ap.parse_args()


def load_counters():
    """Per-sample work counts of the oracle (cells visited, triangle tests, mesh hits), committed fixtures."""
    path = os.path.join(ROOT, "tests", "golden", "work_counters.json")
    with open(path) as f:
        return json.load(f)


def algorithmic_bytes_per_sample(name, spp, counters, all_cells=False, work_done=False):
    """8*C + 76*T + 72*H + 24/spp bytes per sample (T triangle tests, H shaded mesh hits) with C = the visited cells that hold a
    triangle: only those gather their 8-byte entry from memory — an empty cell is answered by one bit of the occupancy mask in LDS, which
    is part of the data layout (round 2's advisor: "count 8 B only for occupied cells").  all_cells=True prices every
    visited cell at 8 bytes, §8d to the letter (what rounds 1 and 2 reported)."""
    c = counters.get(name)
    if c is None:
        return None
    n = float(c["samples"])
    cells = c["cells"] if all_cells or "occupied_cells" not in c else c["occupied_cells"]
    tests = c["tri_tests"]
    if work_done:
        # the tests the kernel RUNS: the reference's minus the re-tests of triangles the previous cell of the same walk listed too (they missed there,
        # they miss here: the kernel's per-entry-direction lists leave them out, device_types.hpp) — oracle counter `retests`
        if "retests" not in c:
            return None
        tests = c["tri_tests"] - c["retests"]
    return 8.0 * cells / n + 76.0 * tests / n + 72.0 * c["mesh_hits"] / n + 24.0 / spp


def load_kernel_counters():
    """What the mesh kernel itself counts per sample (tools/kernel_counters.py: DIAG build, RMD_DEBUG = 8), committed fixture."""
    path = os.path.join(ROOT, "tests", "golden", "kernel_counters.json")
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        return json.load(f)


def requested_bytes_per_sample(name, spp, counters, kernel_counters):
    """The bytes the kernel REQUESTS per sample, by its own counts: per (ray, triangle) pair its rounds number a 4-byte index and a 32-byte sphere; per
    pair that passes the pre-test a 72-byte record; per visited cell that holds a triangle an 8-byte entry; per shaded mesh hit 176 bytes (positions,
    normals, Heron constants); per path through a queue its entry written and read (ray 96 B, hit 88 B each way); per sample a 24-byte store and, once,
    the sum's 24-byte load; the pixel once per launch.  (SURVEY.md section 8d's formula prices the REFERENCE's work instead: `achieved` / `frac`.)"""
    k, c = kernel_counters.get(name), counters.get(name)
    if not k or not c:
        return None
    n = float(c["samples"])
    return (36.0 * k["pairs_numbered_per_sample"] + 72.0 * k["pairs_passing_the_pre_test_per_sample"] + 8.0 * c.get("occupied_cells", c["cells"]) / n
            + 176.0 * c["mesh_hits"] / n + 2.0 * 96.0 * k["rays_pushed_per_sample"] + 2.0 * 88.0 * k["hits_pushed_per_sample"] + 48.0 + 48.0 / spp)


def gpu_state():
    """Clocks and power of GPU 0 as rocm-smi reports them NOW (the box-to-box spread of a VALU-bound kernel has to have a cause beside it)."""
    import subprocess

    try:
        r = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower", "--showmaxpower", "--showperflevel", "--json"], capture_output=True, text=True, timeout=20)
        d = json.loads(r.stdout)
        card = d[sorted(d)[0]]
        keep = {k: v for k, v in card.items() if any(w in k.lower() for w in ("sclk", "mclk", "fclk", "socclk", "power", "performance level"))}
        return keep or {"raw": r.stdout[:400]}
    except Exception as e:  # (no rocm-smi, no permission: say so instead of failing the bench)
        return {"error": "%s: %s" % (type(e).__name__, e)}


def usable_cpus():
    """Threads worth starting: the affinity mask, capped by the cgroup CPU quota when there is one (a container may list
    256 CPUs but be granted 16 cores' worth of time; more runnable threads than that only add contention)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:  # cgroup v2: "<quota> <period>" or "max <period>"
            q, period = f.read().split()
            if q != "max":
                quota = float(q) / float(period)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:  # cgroup v1
                q, period = float(f.read()), float(g.read())
                if q > 0:
                    quota = q / period
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.999)))
    return n, quota


def source_hash():
    """sha256 (16 hex digits) of the product's device and host sources: the committed PMC figures (profiles/pmc_latest.json,
    profiles/hbm_traffic.json; tools/pmc_summary.py stores the hash they were collected at) describe the kernels of THAT source."""
    import glob
    import hashlib

    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "raymond_amd", "csrc", "*.h*")) + glob.glob(os.path.join(ROOT, "raymond_amd", "csrc", "*.cpp")))
    for f in files + [os.path.join(ROOT, "include", "raymond_hip.h")]:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_entry(fname, name, spp):
    """Entry `name` of a committed PMC summary, with "stale": true when the sources it was measured at are not the ones built now."""
    path = os.path.join(ROOT, "profiles", fname)
    if not os.path.exists(path):
        return None
    with open(path) as f:
        entry = json.load(f).get(name)
    if not entry or entry.get("spp") != spp:
        return None
    entry = dict(entry)
    entry["stale"] = entry.get("source_hash") != source_hash()
    return entry


def load_traffic(name, spp):
    """Measured L2<->fabric bytes of one launch of workload `name` at `spp` samples per pixel, from the committed rocprofv3
    --pmc passes (profiles/hbm_traffic.json: FETCH_SIZE and WRITE_SIZE, separate passes, gfx950 corrections applied), or
    None when no pass was collected at that spp."""
    entry = pmc_entry("hbm_traffic.json", name, spp)
    return entry["bytes_per_launch"] if entry else None


def valu_block(name, avg_ms):
    """The SQ-counter figures of the committed profile plus, from THIS run's launch duration, the time one SIMD spends per
    vector instruction: 1,024 SIMDs x duration / wave instructions.  A wave64 FP64/INT32 vector instruction occupies its SIMD for
    4 cycles (1.67 ns at the 2.4 GHz peak clock; ~1.9 ns at the ~2.1 GHz the part sustains under this load), so a value near
    that means the vector ALUs issue back to back — the launch is VALU-issue bound."""
    v = pmc_entry("pmc_latest.json", name, 500)
    if not v:
        return None
    v["simd_ns_per_valu_instr"] = round(avg_ms * 1e6 * 1024.0 / v["wave_instr_valu_per_launch"], 3)
    v["valu_issue_floor_ns"] = {"at_2.4GHz": VALU_ISSUE_NS, "note": "4 cycles per wave64 instruction"}
    # the fraction of the vector ALUs' issue slots this launch fills, and the fraction that does useful work (active lanes)
    v["valu_issue_frac"] = round(VALU_ISSUE_NS / v["simd_ns_per_valu_instr"], 4)
    v["useful_frac"] = round(v["valu_issue_frac"] * v["lane_utilisation"], 4)
    # (v["wait"], when the profile holds it: SQ_WAIT_ANY / SQ_WAIT_INST_ANY per wave cycle and the LDS bank-conflict share, tools/profile_round.sh)
    if "valu_stream_ms" in v:
        # the launch's instruction stream priced class by class (f64 arithmetic 2.05 ns, rcp / rsq 6.7 ns, the rest 0.95 .. 1.28 ns per wave instruction
        # per SIMD: tools/microbench/valu_rate.hip) against this run's duration: how busy the vector ALUs are — valu_issue_frac prices every
        # instruction at the 1.667 ns of a 4-cycle issue and so underrates a stream of f64 operations
        v["valu_busy_frac"] = {"low": round(v["valu_stream_ms"]["low"] / avg_ms, 4), "high": round(v["valu_stream_ms"]["high"] / avg_ms, 4)}
    return v


def bound_by_counters(rl):
    """Which roof the counters put the launch under: VALU issue when the vector ALUs' issue slots are fuller than the memory pipe
    (measured L2<->fabric bytes / HBM peak); the text is derived from the numbers of this line."""
    v = rl.get("valu")
    if not v:
        return None, None
    mem = rl.get("measured_frac")
    bound = "valu" if mem is None or v["valu_issue_frac"] >= mem else "hbm"
    busy = v.get("valu_busy_frac")
    text = "%s: %.2f of the VALU issue slots busy at %.1f %% lanes (useful %.2f)%s, measured L2<->fabric traffic %s of the HBM peak%s" % (
        "VALU issue" if bound == "valu" else "HBM", v["valu_issue_frac"], 100.0 * v["lane_utilisation"], v["useful_frac"],
        "" if not busy else "; the instruction stream at its classes' measured issue costs keeps the vector ALUs busy %.2f - %.2f of the time" % (busy["low"], busy["high"]),
        "n/a" if mem is None else "%.3f" % mem, " (PMC figures are STALE: collected at other sources)" if v.get("stale") else "")
    return bound, text


def host_api_block(ctx):
    """raymond_amd/host/raymond_cli hostapi ... (cli.cpp): one JSON line per run; a child process (its own HIP runtime)."""
    import subprocess

    cli = os.path.join(ROOT, "raymond_amd", "host", "raymond_cli")
    if not os.path.exists(cli):
        return {"error": "raymond_amd/host/raymond_cli is not built"}
    block = {"how": "raymond_cli hostapi: scene built first, a tiny untimed render initialises the HIP runtime, then render_tiled(scene, settings) is timed from the "
                    "call to the last TileFinished message (best of 3); wall_ms includes the worker's context creation, the scene upload (setup_ms) and every "
                    "transfer; tile sums stay resident on the GPU between passes, the tiles of a pass's messages are downloaded on a copy stream while the next pass renders"}
    for key, scene in (("C2", "spheres"), ("C3", "dragon")):
        for spi in (0, 8):
            r = subprocess.run([cli, "hostapi", scene, "1920", "1080", "500", "5", str(spi)], capture_output=True, text=True, timeout=300)
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            name = "%s_%s" % (key, "one_pass" if spi == 0 else "progressive_%d_spp_a_pass" % spi)
            block[name] = json.loads(lines[-1]) if r.returncode == 0 and lines else {"error": (r.stderr or r.stdout)[-300:]}
    for key in ("C2", "C3"):
        a, b = block.get(key + "_one_pass", {}), block.get(key + "_progressive_8_spp_a_pass", {})
        if "msamples_per_s" in a and "msamples_per_s" in b:
            block[key + "_progressive_over_one_pass"] = round(b["msamples_per_s"] / a["msamples_per_s"], 4)
    return block


def launch_ranks(n):
    """`python bench.py --gpus N` outside a launcher: start the N ranks as FRESH child processes, one per GPU, through
    torch.distributed.run (the form the driver uses), let rank 0's JSON line through on stdout, and return their exit status.
    Called before this process has imported torch or touched the GPU — a process that has initialised the GPU is never
    re-executed; this parent only waits."""
    import socket
    import subprocess

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:  # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this pool's host driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))
    # every rank, whoever launched it (this script's own parent or the driver's torch.distributed.run): dmabuf IPC for RCCL, set before torch
    # is imported and before the first HIP call of the process (the HSA runtime reads its environment once)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if os.environ.get("RMD_BENCH_ECHO_ENV"):  # tests/test_distributed_gloo.py: what a rank's environment holds, without touching a GPU
        # (one write of line + newline: the ranks share the launcher's stdout, and print()'s separate newline can land after the other rank's line)
        sys.stdout.write(json.dumps({"rank": os.environ.get("RANK", "0"), "HSA_ENABLE_IPC_MODE_LEGACY": os.environ["HSA_ENABLE_IPC_MODE_LEGACY"]}) + "\n")
        sys.stdout.flush()
        return
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (there is no CPU fallback)")
    # RMD_BENCH_BACKEND=gloo is a rehearsal mode for a one-GPU box: the ranks share device 0 and reduce over gloo
    # (RCCL refuses two ranks on one GPU).  The driver's runs use the default: one rank per GPU, backend nccl = RCCL.
    backend = os.environ.get("RMD_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from raymond_amd import render, scenes, shard
    from raymond_amd.scene import generate_tiles, tile_array

    def barrier():
        if dist is not None:
            if backend == "nccl":
                dist.barrier(device_ids=[local_rank])
            else:
                dist.barrier()

    def setup(name, spp, mode="default"):
        # mode: "default" = rmd_settings.flags 0 (+ RMD_RENDER_DOF for C5): the reference-identical mode; "trace" = RMD_RENDER_TRACE_BLACK_PATHS;
        # "end" = RMD_RENDER_END_BLACK_PATHS (include/raymond_hip.h)
        st = scenes.config_settings(name, spp=spp)
        st.trace_black_paths, st.end_black_paths = mode == "trace", mode == "end"
        cam = st.camera_settings
        sc = getattr(scenes, scenes.CONFIGS[name][0])()
        tiles = generate_tiles(cam.backbuffer_width, cam.backbuffer_height, st.tile_size)
        share = shard.shard_tiles(tiles, rank, world)  # tile i -> rank i mod N
        return st, cam, sc, tiles, share

    # everything launches on torch's current stream so that zeroing, the render kernel and the reduce are ordered
    stream = torch.cuda.current_stream(dev).cuda_stream
    ctx = render.Context(local_rank, stream=stream)

    # --assemble abi: the C-ABI's own collective (rmd_comm_* over RCCL, csrc/comm.cpp) — rank 0 makes the id, everyone joins
    abi_comm = None
    if args.assemble == "abi":
        from raymond_amd import abi as rabi

        if world > 1 and backend != "nccl":
            raise SystemExit("--assemble abi needs one rank per GPU (RCCL refuses two ranks on one device): not available in the gloo rehearsal")
        uid = (C.c_uint8 * rabi.RMD_COMM_ID_BYTES)()
        if rank == 0:
            ctx.check(ctx.L.rmd_comm_unique_id(uid))
        box = [bytes(uid)]
        if dist is not None:
            dist.broadcast_object_list(box, src=0, device=dev)
        uid = (C.c_uint8 * rabi.RMD_COMM_ID_BYTES).from_buffer_copy(box[0])
        abi_comm = C.c_void_p()
        ctx.check(ctx.L.rmd_comm_create(ctx.handle, uid, rank, world, C.byref(abi_comm)))

    def run_workload(name, spp, steps, warmup, reduce, mode="default"):
        st, cam, sc, tiles, share = setup(name, spp, mode)
        W, H = cam.backbuffer_width, cam.backbuffer_height
        ds = render.DeviceScene(ctx, sc)
        fb_t = torch.zeros(W * H * 3, dtype=torch.float64, device=dev)
        fb = render.Framebuffer(ctx, W, H, device_ptr=fb_t.data_ptr())
        arr = (tile_array(share), len(share))
        kernel_ms = []
        gather = None
        if reduce and dist is not None and args.assemble == "gather":
            gather = shard.OwnedTileGather(torch, W, H, tiles, rank, world, dev, root=0)

        def step():
            fb_t.zero_()
            render.render_tiles(ctx, ds, cam, st, arr, fb, 0, st.sample_count, sync=False)
            if reduce and abi_comm is not None:
                ctx.check(ctx.L.rmd_reduce_framebuffer_async(abi_comm, fb_t.data_ptr(), fb_t.numel(), 0))  # ncclReduce enqueued on the context's stream: steps queue back to back
            elif gather is not None:
                gather(dist, fb_t, stage_host=(backend != "nccl"))
            elif reduce and dist is not None:
                if backend == "nccl":
                    shard.reduce_framebuffer(dist, fb_t, root=0)
                else:  # rehearsal: stage through the host
                    torch.cuda.synchronize(dev)
                    host = fb_t.cpu()
                    shard.reduce_framebuffer(dist, host, root=0)
                    if rank == 0:
                        fb_t.copy_(host)

        for _ in range(warmup):
            step()
            torch.cuda.synchronize(dev)
        barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
            # HIP events recorded by the library around the kernel, on the stream it was launched on.  Reading them waits for the kernel: at N = 1,
            # where the roofline needs every launch's duration, the steps are host-synchronous; with several ranks the steps are queued back to back
            # (zeroing, kernel, collective — all on one stream) and only the last launch's duration is read, after the timed region
            if world == 1:
                kernel_ms.append(ctx.last_kernel_ms())
        torch.cuda.synchronize(dev)
        barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            kernel_ms.append(ctx.last_kernel_ms())
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        # the frame's checksum: the sum of its finite values (a reference-identical mesh frame holds the pixels the reference itself makes NaN) and,
        # beside it, how many pixels hold a non-finite channel
        checksum, nonfinite = 0.0, 0
        if rank == 0:
            finite = torch.isfinite(fb_t)
            checksum = float(torch.where(finite, fb_t, torch.zeros_like(fb_t)).sum().item())
            nonfinite = int((~finite.view(-1, 3)).any(dim=1).sum().item())
        info = ctx.last_launch_info()
        ds.close()
        samples_per_step = W * H * st.sample_count
        my_samples = shard.shard_samples(share) * st.sample_count
        return dict(st=st, W=W, H=H, elapsed=elapsed, kernel_ms=kernel_ms, samples_per_step=samples_per_step,
                    my_samples=my_samples, checksum=checksum, nonfinite_pixels=nonfinite, n_tiles=len(tiles), passes=int(info.passes), split_k=int(info.split_k))

    name = args.workload
    spp = args.spp if args.spp is not None else scenes.CONFIGS[name][3]
    gpu_before = gpu_state() if rank == 0 else None
    main_run = run_workload(name, spp, args.steps, args.warmup, reduce=True)
    ms_per_step = main_run["elapsed"] / args.steps * 1e3
    value = main_run["samples_per_step"] * args.steps / main_run["elapsed"] / 1e6

    out = {
        "metric": "Msamples/s (whole node) + achieved HBM GB/s, 1920x1080 @ 500spp 5-bounce",
        "value": round(value, 3),
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "%s: %s, %dx%d, %d spp, %d bounces, 32x32 host tiles round-robin over %d GPU(s)%s"
            % (name, scenes.CONFIGS[name][0], main_run["W"], main_run["H"], spp, main_run["st"].bounce_limit, world,
               (", RCCL gather of each rank's own tiles to rank 0" if args.assemble == "gather" else
                ", RCCL reduce(sum) of the f64 framebuffer to rank 0" + (" through rmd_reduce_framebuffer" if args.assemble == "abi" else "")) if world > 1 else ""),
            "rng": "philox4x32-10, key = seed, counter = (pixel, sample, block, 0); one block per consumer (jitter, lens round, shaded depth)",
            "seed": scenes.SEED,
            "paths": "rmd_settings.flags 0, the reference-identical mode: a path whose throughput has become exactly (0, 0, 0) is ended where that provably "
                     "changes no sample (scenes without grids, as this one); in scenes with a mesh it is traced on unless RMD_RENDER_END_BLACK_PATHS",
        },
    }

    if rank == 0:
        out["gpu"] = {"before_the_timed_steps": gpu_before, "after_the_timed_steps": gpu_state()}
    # who took part (N > 1): every rank's own view — its rank, the world size IT saw, its device and the samples of its tile share — gathered to rank 0
    if dist is not None:
        mine = {"rank": rank, "world_size_seen": int(dist.get_world_size()), "device": int(local_rank), "backend": backend,
                "samples_per_step": int(main_run["my_samples"])}
        seen = [None] * world
        dist.all_gather_object(seen, mine)
        out["ranks"] = seen

    # for the record, beside `value`: the same frame with every path traced to its end as the reference does (RMD_RENDER_TRACE_BLACK_PATHS) —
    # the same checksum, the reference's full number of path segments
    if world == 1 and not args.no_roofline_leg and not args.lean:
        full = run_workload(name, spp, 1, 1, reduce=False, mode="trace")
        out["tracing_black_paths"] = {
            "value": round(full["samples_per_step"] / full["elapsed"] / 1e6, 3), "unit": "Msamples/s", "ms_per_step": round(full["elapsed"] * 1e3, 3),
            "checksum": full["checksum"], "nonfinite_pixels": full["nonfinite_pixels"],
            "note": "rmd_settings.flags = RMD_RENDER_TRACE_BLACK_PATHS: paths whose throughput is exactly (0, 0, 0) are traced on (1.8x the path segments); the same frame "
                    "bit for bit (same checksum) — the segment count the CPU baseline executes; not what `value` measures",
        }
    if rank == 0:
        counters = load_counters()
        avg_ms = sum(main_run["kernel_ms"]) / len(main_run["kernel_ms"])
        bps = algorithmic_bytes_per_sample(name, spp, counters)
        if bps is None:
            bps = 24.0 / spp
        ach = bps * main_run["my_samples"] / (avg_ms * 1e-3) / 1e9
        # <MODE, GRID> as rocprofv3 prints it: launches split every tile's samples over several waves (MODE 1) + the ordered sum
        # mesh scenes: persistent render kernel + sum_kernel; spheres: the render kernel's waves add the samples themselves
        kname = ("rmd::render_kernel<1, true, true, false, true> + rmd::sum_kernel" if scenes.CONFIGS[name][0] != "reflective_spheres"
                 else "rmd::render_kernel<1, false, true> (persistent workgroups, ordered sample sum inside)")
        out["kernel"] = {"name": kname, "avg_ms": round(avg_ms, 3), "launches": len(main_run["kernel_ms"]), "checksum": main_run["checksum"],
                         "nonfinite_pixels": main_run["nonfinite_pixels"]}
        traffic = load_traffic(name, spp) if world == 1 else None
        rl = {
            "bound": "hbm", "achieved": round(ach, 4), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "traffic": traffic, "bytes_per_sample": bps, "avg_ms": round(avg_ms, 3),
            "definition": "achieved / frac = ALGORITHMIC bytes per launch (SURVEY.md section 8d) / launch time; measured_gbs / measured_frac = L2<->fabric "
                          "bytes by PMC (an upper bound on HBM bytes: Infinity-Cache hits included)",
        }
        if traffic is not None:
            rl["measured_gbs"] = round(traffic / (avg_ms * 1e-3) / 1e9, 3)
            rl["measured_frac"] = traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        if name == "C2":
            # the real bound of this workload: FP64 vector-ALU issue (the 8-object scene lives in LDS/SGPRs, HBM sees the framebuffer only)
            rl["note"] = "VALU-issue bound, not HBM bound: see `valu`"
            v = valu_block("C2", avg_ms) if spp == 500 else None
            if v:
                rl["valu"] = v
                rl["bound"], rl["bound_by_counters"] = bound_by_counters(rl)
        out["roofline_%s" % name.lower()] = rl

    # roofline leg: the traversal on the ~100k-triangle mesh — config C3 at its 500 spp, ONE launch per step — in the reference-identical
    # mode (flags 0: every path traced to its end on a mesh scene) and, beside it, with RMD_RENDER_END_BLACK_PATHS
    if not args.no_roofline_leg and world == 1:
        rspp = args.roofline_spp if args.roofline_spp is not None else scenes.CONFIGS["C3"][3]
        counters = load_counters()

        def mesh_leg(mode, steps, counter_key, pmc_key):
            rr = run_workload("C3", rspp, steps, 1, reduce=False, mode=mode)
            bps = algorithmic_bytes_per_sample(counter_key, rspp, counters)
            avg_ms = sum(rr["kernel_ms"]) / len(rr["kernel_ms"])
            ach = bps * rr["samples_per_step"] / (avg_ms * 1e-3) / 1e9
            traffic = load_traffic(pmc_key, rspp)
            leg = {
                "bound": "hbm", "achieved": round(ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                "avg_ms": round(avg_ms, 3), "launch_ms": [round(v, 3) for v in rr["kernel_ms"]],
                "msamples_per_s": round(rr["samples_per_step"] / (avg_ms * 1e-3) / 1e6, 2),
                "bytes_per_sample": round(bps, 2), "bytes_per_launch": bps * rr["samples_per_step"],
                "frac_with_every_visited_cell_at_8_bytes": algorithmic_bytes_per_sample(counter_key, rspp, counters, all_cells=True) * rr["samples_per_step"] / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "checksum": rr["checksum"], "nonfinite_pixels": rr["nonfinite_pixels"],
            }
            req = requested_bytes_per_sample(counter_key, rspp, counters, load_kernel_counters())
            if req is not None:
                # what the kernel asks the memory system for, by its own counters (tests/golden/kernel_counters.json)
                leg["bytes_requested"] = {"per_sample": round(req, 2), "per_launch": req * rr["samples_per_step"],
                                          "gbs": round(req * rr["samples_per_step"] / (avg_ms * 1e-3) / 1e9, 2),
                                          "frac": req * rr["samples_per_step"] / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
            done = algorithmic_bytes_per_sample(counter_key, rspp, counters, work_done=True)
            if done is not None:
                # the contractual `frac` prices the REFERENCE's triangle tests; this one prices the tests the kernel runs (22 % fewer)
                leg["bytes_per_sample_of_work_done"] = round(done, 2)
                leg["frac_of_work_done"] = done * rr["samples_per_step"] / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            if traffic is not None:
                # L2<->fabric bytes per launch by PMC (upper bound on HBM bytes: Infinity-Cache hits are included)
                leg["measured_gbs"] = round(traffic / (avg_ms * 1e-3) / 1e9, 3)
                leg["measured_frac"] = traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
                leg["traffic_stale"] = pmc_entry("hbm_traffic.json", pmc_key, rspp)["stale"]
            v = valu_block(pmc_key, avg_ms) if rspp == 500 else None
            if v:
                leg["valu"] = v
                leg["valu_issue_frac"], leg["useful_frac"] = v["valu_issue_frac"], v["useful_frac"]
                leg["bound"], leg["bound_by_counters"] = bound_by_counters(leg)
            # the roof that binds first: bound, the vector ALUs' issue and useful fractions — then the contractual HBM figures
            first = ("bound", "bound_by_counters", "valu_issue_frac", "useful_frac")
            return {**{k: leg[k] for k in first if k in leg}, **{k: v for k, v in leg.items() if k not in first}}

        rl = mesh_leg("default", args.roofline_steps, "C3_reference", "C3")
        rl.update({
            "workload": "C3: gold_dragon_standin (99,372 triangles, DDA grid), 1920x1080, %d spp, 5 bounces, one launch; rmd_settings.flags 0 = "
                        "reference-identical: every path traced to its end" % rspp,
            "kernel": "rmd::render_kernel<1, true, true, false, true> + rmd::sum_kernel (persistent workgroups; the paths in per-wave queues in device memory: render_wave_queued)",
            "definition": "achieved / frac = ALGORITHMIC bytes per launch / launch time — most of these bytes are answered by LDS (occupancy mask), L2 and the "
                          "Infinity Cache, so the figure saturates and does not rank kernels any more; measured_gbs / measured_frac = L2<->fabric bytes by PMC, "
                          "an upper bound on HBM bytes; valu_issue_frac = 1.667 ns / (ns per wave-level vector instruction per SIMD): the bound that binds; "
                          "useful_frac = valu_issue_frac x lane utilisation",
            "how": "achieved = algorithmic bytes per launch (8 B x visited cells that hold a triangle + 76 B x triangle tests + 72 B x shaded mesh hits per sample, "
                   "oracle counters in tests/golden/work_counters.json: C3_reference for flags 0, C3 for ending_black_paths, + 24 B/pixel) / mean launch duration "
                   "from HIP events on the launch stream; profiles/: the rocprofv3 kernel-trace mean of render_kernel<1, true, true, false, true> (persistent workgroups, path queues) + "
                   "sum_kernel over the same launches; valu / traffic: committed rocprofv3 --pmc passes (separate SQ / FETCH_SIZE / WRITE_SIZE passes), "
                   "\"stale\": true when the sources have changed since",
        })
        out["roofline"] = rl
        if not args.lean:
            end = mesh_leg("end", max(1, args.roofline_steps - 1), "C3", "C3_end")
            end["note"] = ("rmd_settings.flags = RMD_RENDER_END_BLACK_PATHS (opt-in on mesh scenes): zero-throughput paths ended; every sample that is finite in the "
                           "reference keeps its value bit for bit, a sample the reference makes NaN behind a zero weight comes out (0, 0, 0)")
            rl["ending_black_paths"] = end
            # the other mesh configurations of BASELINE.json on this GPU, short launches (throughput does not depend on the sample count): both modes
            cfg = {}
            for cname, cspp in (("C4", 20), ("C5", 50)):
                for mode in ("default", "end"):
                    r = run_workload(cname, cspp, 1, 1, reduce=False, mode=mode)
                    ms = sum(r["kernel_ms"]) / len(r["kernel_ms"])
                    cfg.setdefault(cname, {"workload": "%s, %dx%d, %d bounces%s; %d-spp launch of the config's %d" % (
                        scenes.CONFIGS[cname][0], r["W"], r["H"], r["st"].bounce_limit, ", thin lens (RMD_RENDER_DOF)" if r["st"].use_dof else "", cspp, scenes.CONFIGS[cname][3])})
                    cfg[cname]["reference_identical" if mode == "default" else "ending_black_paths"] = {
                        "kernel_ms": round(ms, 3), "msamples_per_s": round(r["samples_per_step"] / (ms * 1e-3) / 1e6, 2)}
            # ... and ONE launch each of the two configurations BASELINE.json names for 8 GPUs at their FULL sample counts, flags 0 (reference-identical):
            # the per-sample scratch does not hold all samples at once, so the library runs them as several passes (rmd_last_launch_info)
            for cname in ("C4", "C5"):
                r = run_workload(cname, scenes.CONFIGS[cname][3], 1, 0, reduce=False, mode="default")
                ms = sum(r["kernel_ms"]) / len(r["kernel_ms"])
                cfg[cname]["full_configuration"] = {"spp": scenes.CONFIGS[cname][3], "kernel_ms": round(ms, 1), "passes": r["passes"], "msamples_per_s": round(r["samples_per_step"] / (ms * 1e-3) / 1e6, 2),
                                                   "nonfinite_pixels": r["nonfinite_pixels"], "checksum": r["checksum"]}
            cfg["C3"] = {"workload": rl["workload"], "reference_identical": {"kernel_ms": rl["avg_ms"], "msamples_per_s": rl["msamples_per_s"]},
                         "ending_black_paths": {"kernel_ms": end["avg_ms"], "msamples_per_s": end["msamples_per_s"]}}
            cfg["C2"] = {"workload": out["config"]["workload"], "reference_identical": {"kernel_ms": out["kernel"]["avg_ms"], "msamples_per_s": round(value, 2)},
                         "tracing_black_paths": {"kernel_ms": out["tracing_black_paths"]["ms_per_step"], "msamples_per_s": out["tracing_black_paths"]["value"]}}
            out["configs"] = cfg
    elif rank == 0:
        out["roofline"] = out.get("roofline_%s" % name.lower())

    # What a drop-in caller gets: C2 and C3 end to end through the reference-shaped API — raymond_amd/host's render_tiled (the C++ stand-in for
    # integration/gpu.rs: tile queue, progressive passes, TileProgressed / TileFinished messages), wall time from the call to the last
    # TileFinished, once in one pass (samples_per_iteration 0) and once progressively (8 samples a pass: src/trace.rs:207-219).
    if rank == 0 and world == 1 and not args.no_roofline_leg and not args.lean:
        out["host_api"] = host_api_block(ctx)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib

        cores, quota = usable_cpus()
        # calibrate on 1 spp, then size the sample for ~15 s of CPU work (the rate is spp-independent)
        st, cam, sc, tiles, _ = setup(name, 1)
        osc = oracle_lib.OracleScene(sc, fast=True)  # liboracle_fast.so: the restatement without its work counters, -O3
        osc.render_tiles(cam, st, tiles, threads=cores)  # untimed: thread start-up, page faults
        t0 = time.perf_counter()
        osc.render_tiles(cam, st, tiles, threads=cores)
        t1 = time.perf_counter() - t0
        cpu_spp = args.cpu_spp if args.cpu_spp > 0 else max(2, min(256, int(15.0 / max(t1, 1e-3))))
        st = scenes.config_settings(name, spp=cpu_spp)
        t0 = time.perf_counter()
        osc.render_tiles(cam, st, tiles, threads=cores)
        dt = time.perf_counter() - t0
        n = cam.backbuffer_width * cam.backbuffer_height * cpu_spp
        out["cpu_baseline"] = {
            "value": round(n / dt / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": "%s at %dx%d, %d spp (%.1f M samples, %.1f s): reference-equivalent C++ restatement (oracle/), worker pool of %d threads (num_cpus::get() capped by the cgroup CPU quota%s), 32x32 tiles, g++ -O3, work counters compiled out (-DORC_NO_COUNTERS)"
            % (name, cam.backbuffer_width, cam.backbuffer_height, cpu_spp, n / 1e6, dt, cores, "" if quota is None else " of %.1f CPUs" % quota),
        }
        out["speedup_vs_cpu_baseline"] = round(value / (n / dt / 1e6), 1)
        if "tracing_black_paths" in out:
            # like for like: the port traces every path to its end (1.8x the segments of the default on this scene)
            out["speedup_vs_cpu_baseline_same_segments"] = round(out["tracing_black_paths"]["value"] / (n / dt / 1e6), 1)
            out["cpu_baseline"]["note"] = ("the port executes the reference's full segment count; `speedup_vs_cpu_baseline` divides the default rate (zero-throughput "
                                           "paths ended: exact on this scene, and a CPU port could do the same) by it, `speedup_vs_cpu_baseline_same_segments` the "
                                           "rate with every path traced")
        # the reference's README quotes its ReflectiveSpheres time on a "4 core i5" (0.268 Msamples/s at 592x340): same port on 4 threads
        spp4 = max(1, min(cpu_spp, int(8.0 / max(t1 * cores / 4.0, 1e-3))))
        st4 = scenes.config_settings(name, spp=spp4)
        t0 = time.perf_counter()
        osc.render_tiles(cam, st4, tiles, threads=4)
        dt4 = time.perf_counter() - t0
        out["cpu_baseline_4_threads"] = {"value": round(cam.backbuffer_width * cam.backbuffer_height * spp4 / dt4 / 1e6, 4), "unit": "Msamples/s", "cores": 4,
                                         "kind": "port", "sample": "%s at %dx%d, %d spp, %.1f s" % (name, cam.backbuffer_width, cam.backbuffer_height, spp4, dt4)}
        # the mesh workload of the roofline leg on the same cores (the reference's README quotes its GoldDragon render at 0.101 Msamples/s on
        # the same "4 core i5", README.md:27): ~8 s of CPU work
        if "roofline" in out and not args.no_roofline_leg and name != "C3":
            st3, cam3, sc3, tiles3, _ = setup("C3", 1)
            osc3 = oracle_lib.OracleScene(sc3, fast=True)
            t0 = time.perf_counter()
            osc3.render_tiles(cam3, st3, tiles3, threads=cores)
            t3 = time.perf_counter() - t0
            spp3 = max(1, min(64, int(8.0 / max(t3, 1e-3))))
            st3 = scenes.config_settings("C3", spp=spp3)
            t0 = time.perf_counter()
            osc3.render_tiles(cam3, st3, tiles3, threads=cores)
            dt3 = time.perf_counter() - t0
            n3 = cam3.backbuffer_width * cam3.backbuffer_height * spp3
            out["roofline"]["cpu_baseline"] = {"value": round(n3 / dt3 / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
                                               "sample": "C3 at %dx%d, %d spp (%.1f M samples, %.1f s), same build and worker pool" % (cam3.backbuffer_width, cam3.backbuffer_height, spp3, n3 / 1e6, dt3)}
            if out["roofline"].get("msamples_per_s"):
                out["roofline"]["speedup_vs_cpu_baseline"] = round(out["roofline"]["msamples_per_s"] / (n3 / dt3 / 1e6), 1)

    if abi_comm is not None:
        ctx.L.rmd_comm_destroy(abi_comm)
    ctx.close()
    if rank == 0:
        print(json.dumps(out, allow_nan=False), flush=True)
    if dist is not None:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
