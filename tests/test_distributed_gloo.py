"""N > 1 host logic on CPU: two gloo ranks shard the tiles (raymond_amd.shard), render their share into zeroed
full-size framebuffers and assemble them on rank 0 (reduce(sum) of the frames, or a gather of the tiles each rank
owns) — which must equal the single-process image bit for bit
(SURVEY.md §8e).  The renderer here is the CPU oracle (this test checks the sharding/reduce logic, which is what
bench.py --gpus N runs around rmd_render_tiles; the GPU equivalent is test_tile_shards_sum_to_the_full_image)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path, assemble="reduce"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    import oracle_lib
    from raymond_amd import scenes, shard
    from raymond_amd.scene import Settings, generate_tiles

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        st = Settings(scenes.camera(160, 96), sample_count=3, tile_size=(32, 32), bounce_limit=4, seed=21)
        cam = st.camera_settings
        tiles = generate_tiles(160, 96, st.tile_size)
        mine = shard.shard_tiles(tiles, rank, world)
        osc = oracle_lib.OracleScene(scenes.reflective_spheres())
        fb = torch.from_numpy(osc.render_tiles(cam, st, mine, threads=2))
        covered = torch.tensor([float(shard.shard_samples(mine))], dtype=torch.float64)
        dist.barrier()
        if assemble == "gather":
            g = shard.OwnedTileGather(torch, 160, 96, tiles, rank, world, "cpu", root=0)
            assert g.pack_rows == max(shard.shard_samples(shard.shard_tiles(tiles, r, world)) for r in range(world))
            g(dist, fb)
        else:
            shard.reduce_framebuffer(dist, fb, root=0)
        dist.all_reduce(covered)
        assert int(covered.item()) == 160 * 96  # the shards partition the frame
        if rank == 0:
            np.save(out_path, fb.numpy())
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world,assemble", [(2, "reduce"), (3, "reduce"), (2, "gather"), (3, "gather")])
def test_sharded_render_plus_reduce_equals_single_process(oracle, tmp_path, world, assemble):
    import multiprocessing

    from raymond_amd import scenes, shard
    from raymond_amd.scene import Settings, generate_tiles

    out_path = str(tmp_path / "reduced.npy")
    # the ranks are spawned with the standard library, so that this (parent) process never imports torch: a later GPU test
    # in the same session would otherwise run libraymond_hip.so's RCCL path next to torch's bundled ROCm libraries
    mp = multiprocessing.get_context("spawn")
    port = _free_port()
    procs = [mp.Process(target=_worker, args=(rank, world, port, out_path, assemble)) for rank in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    st = Settings(scenes.camera(160, 96), sample_count=3, tile_size=(32, 32), bounce_limit=4, seed=21)
    tiles = generate_tiles(160, 96, st.tile_size)
    full = oracle.OracleScene(scenes.reflective_spheres()).render_tiles(st.camera_settings, st, tiles, threads=2)
    assert np.load(out_path).tobytes() == full.tobytes()
    # the partition itself: disjoint, complete, balanced to within one tile
    shares = [shard.shard_tiles(tiles, r, world) for r in range(world)]
    assert sorted(sum(shares, [])) == sorted(tiles)
    assert max(len(s) for s in shares) - min(len(s) for s in shares) <= 1


def test_tile_generation_order_matches_render_tiled():
    """src/trace.rs:142-173: y advances first, then x (column-major); edge tiles are clamped."""
    from raymond_amd.scene import generate_tiles

    t = generate_tiles(70, 50, (32, 32))
    assert t == [(0, 0, 32, 32), (0, 32, 32, 18), (32, 0, 32, 32), (32, 32, 32, 18), (64, 0, 6, 32), (64, 32, 6, 18)]
    assert len(generate_tiles(1920, 1080, (32, 32))) == 60 * 34
    assert generate_tiles(32, 32, (32, 32)) == [(0, 0, 32, 32)]


def test_shard_rejects_bad_rank():
    from raymond_amd import shard

    with pytest.raises(ValueError):
        shard.shard_tiles([(0, 0, 1, 1)], 2, 2)


def test_owned_tile_gather_indices():
    """The packs partition the frame: every pixel row is owned by exactly one rank; the padding of smaller shares is
    never unpacked; a tile list that leaves pixels out leaves them untouched on the root."""
    import torch

    from raymond_amd import shard
    from raymond_amd.scene import generate_tiles

    W, H = 70, 50
    tiles = generate_tiles(W, H, (32, 32))  # 6 tiles of four different sizes
    for world in (1, 2, 4, 7):  # 7 > 6 tiles: one rank owns nothing
        rows = [shard.tile_pixel_rows(shard.shard_tiles(tiles, r, world), W) for r in range(world)]
        allrows = np.sort(np.concatenate(rows))
        assert np.array_equal(allrows, np.arange(W * H))
        # single-process emulation of the collective: rank r's frame holds r+1 on its own pixels
        g = [shard.OwnedTileGather(torch, W, H, tiles, r, world, "cpu", root=0) for r in range(world)]
        frames = []
        for r in range(world):
            f = torch.zeros(W * H * 3, dtype=torch.float64)
            f.view(-1, 3)[torch.from_numpy(rows[r])] = float(r + 1)
            frames.append(f)
        for r in range(world):
            g[0].recv[r].copy_(g[r].pack(frames[r]))
        out = g[0].unpack(frames[0]).view(-1, 3)
        for r in range(world):
            assert bool((out[torch.from_numpy(rows[r])] == float(r + 1)).all())
    # x, y of a row: x + y * W
    assert shard.tile_pixel_rows([(64, 32, 6, 18)], W)[:7].tolist() == [64 + 32 * W + i for i in range(6)] + [64 + 33 * W]


def test_bench_starts_its_own_ranks_without_touching_the_gpu_in_the_parent():
    """`python bench.py --gpus 2` outside a launcher: the parent starts two fresh ranks (torch.distributed.run) before it has
    imported torch or made a GPU call, and returns their status.  Without a GPU (this container) each RANK refuses loudly —
    the product has no CPU fallback — and the parent's exit status is non-zero."""
    import subprocess

    if __import__("torch").cuda.is_available():
        pytest.skip("CPU-side check of the launch route; the GPU form is test_bench_contract_and_two_rank_rehearsal")
    env = dict(os.environ, RMD_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None), env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0
    # a rank ran bench.py's main() and refused (the launcher ends the other rank as soon as the first one fails, so
    # the second refusal is printed or not depending on which came first), and it was the launcher that reported it
    assert r.stderr.count("bench.py needs an MI355X") >= 1, r.stderr[-2000:]
    assert "ChildFailedError" in r.stderr or "bench.py FAILED" in r.stderr, r.stderr[-2000:]
    assert "{" not in r.stdout  # no JSON line from a run that measured nothing


@pytest.mark.parametrize("form", ["self-launch", "under-a-launcher"])
def test_every_rank_gets_the_dmabuf_ipc_setting_in_both_launch_forms(form):
    """RCCL on this pool needs HSA_ENABLE_IPC_MODE_LEGACY=0 in every rank before its first HIP call (the driver only supports dmabuf IPC).
    bench.py sets it (setdefault) at the top of every rank's main() — whether the ranks were started by `python bench.py --gpus 2` itself or
    by the driver's `python -m torch.distributed.run ... bench.py --gpus 2` — and never overrides a value the environment already holds.
    RMD_BENCH_ECHO_ENV makes a rank print what it holds and return before it touches torch or a GPU."""
    import json
    import subprocess

    for preset in (None, "1"):
        env = dict(os.environ, RMD_BENCH_ECHO_ENV="1")
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "HSA_ENABLE_IPC_MODE_LEGACY"):
            env.pop(k, None)
        if preset is not None:
            env["HSA_ENABLE_IPC_MODE_LEGACY"] = preset
        if form == "self-launch":
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"]
        else:
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
                   str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2"]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        echoed = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
        assert sorted(e["rank"] for e in echoed) == ["0", "1"], r.stdout
        assert all(e["HSA_ENABLE_IPC_MODE_LEGACY"] == (preset or "0") for e in echoed), echoed


def test_the_library_leaves_the_environment_alone_until_a_multi_gpu_caller_opts_in():
    """csrc/comm.cpp: loading libraymond_hip.so does not touch the environment (round 4's load-time constructor did: it changed the HSA runtime's
    behaviour for single-GPU callers and every other HIP user of the process); rmd_comm_prepare_process() — what a multi-GPU C-ABI caller invokes
    before its first HIP call — sets HSA_ENABLE_IPC_MODE_LEGACY=0 and never overwrites a value the caller chose."""
    import subprocess

    from raymond_amd import lib

    # (os.environ is a snapshot taken at interpreter start: read the C environment)
    code = ("import ctypes; L = ctypes.CDLL(%r); libc = ctypes.CDLL(None); libc.getenv.restype = ctypes.c_char_p; "
            "show = lambda: (libc.getenv(b'HSA_ENABLE_IPC_MODE_LEGACY') or b'unset').decode(); a = show(); "
            "assert L.rmd_comm_prepare_process() == 0; print(a, show())" % lib.LIB_PATH)
    for preset, want in ((None, "unset 0"), ("1", "1 1")):
        env = dict(os.environ)
        env.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)
        if preset is not None:
            env["HSA_ENABLE_IPC_MODE_LEGACY"] = preset
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)
        assert r.returncode == 0 and r.stdout.strip() == want, (r.stdout, r.stderr[-500:])
