"""Multi-GPU partition of the hot path: tiles are independent, so rank r of N renders
host tiles r, r+N, r+2N, ... (round-robin over the reference's column-major tile order, which interleaves
cheap wall tiles and expensive mesh tiles evenly) into a zero-initialised full-size f64 framebuffer.  The root
assembles the frame either by ONE gather of the tiles each rank owns (`OwnedTileGather`: 1/N of a frame per rank) or
by ONE reduce(sum) of the full frames (`reduce_framebuffer`: every pixel is non-zero on exactly one rank, so the sum
is exact whatever the reduction order).  Both give the single-GPU image bit for bit."""


def shard_tiles(tiles, rank, world_size):
    """Tiles of `rank`: tile i -> rank i mod world_size."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    return tiles[rank::world_size]


def shard_samples(tiles):
    """Pixels covered by a tile list (for throughput accounting)."""
    return sum(w * h for (_, _, w, h) in tiles)


def reduce_framebuffer(dist, tensor, root=0):
    """One sum-reduce of the accumulated framebuffer to `root` (RCCL over xGMI on GPUs, gloo on CPU)."""
    dist.reduce(tensor, dst=root, op=dist.ReduceOp.SUM)
    return tensor


def tile_pixel_rows(tiles, width):
    """Row numbers (x + y*width, the framebuffer's pixel order, `src/trace.rs:97`) of the pixels of `tiles`, tile after
    tile, each tile row-major."""
    import numpy as np

    parts = [(np.arange(y, y + h, dtype=np.int64)[:, None] * width + np.arange(x, x + w, dtype=np.int64)[None, :]).ravel()
             for (x, y, w, h) in tiles]
    return np.concatenate(parts) if parts else np.zeros(0, dtype=np.int64)


class OwnedTileGather:
    """Assembles the frame on `root` from the tiles each rank OWNS: every rank packs the pixels of its tiles (one
    index_select), one gather brings the packs to the root (1/N of the frame per rank, all N-1 xGMI links of the root
    in parallel), the root scatters them into its framebuffer (one index_copy_; its own tiles are already in place).
    Moves (N-1)/N of a frame once instead of reducing N full frames, and adds nothing: bit-identical to the
    single-GPU frame by construction.  The index tensors are built once per (frame size, tile list, world)."""

    def __init__(self, torch, width, height, tiles, rank, world, device, root=0):
        rows = [tile_pixel_rows(shard_tiles(tiles, r, world), width) for r in range(world)]
        self.torch, self.rank, self.world, self.root = torch, rank, world, root
        self.pack_rows = max(1, max(len(r) for r in rows))  # every rank sends the same count: the largest share
        mine = rows[rank]
        pad = self.pack_rows - len(mine)
        self.mine = torch.from_numpy(_pad(mine, pad)).to(device)
        self.recv = None
        if rank == root:
            self.recv = torch.empty((world, self.pack_rows, 3), dtype=torch.float64, device=device)
            import numpy as np

            others = [r for r in range(world) if r != root]
            self.theirs = torch.from_numpy(np.concatenate([rows[r] for r in others]) if others else np.zeros(0, dtype=np.int64)).to(device)
            self.src = torch.from_numpy(np.concatenate([r * self.pack_rows + np.arange(len(rows[r]), dtype=np.int64) for r in others])
                                        if others else np.zeros(0, dtype=np.int64)).to(device)

    def pack(self, framebuffer):
        """This rank's pixels, [pack_rows, 3] f64 (rows past its share are padding)."""
        return framebuffer.view(-1, 3).index_select(0, self.mine)

    def unpack(self, framebuffer):
        """Root only: the other ranks' packs (in self.recv) into the framebuffer."""
        framebuffer.view(-1, 3).index_copy_(0, self.theirs, self.recv.view(-1, 3).index_select(0, self.src))
        return framebuffer

    def __call__(self, dist, framebuffer, stage_host=False):
        """stage_host: the collective runs on host copies (rehearsal of several ranks on one GPU over gloo)."""
        pack = self.pack(framebuffer)
        is_root = self.rank == self.root
        if stage_host:
            recv = self.torch.empty(self.recv.shape, dtype=self.recv.dtype) if is_root else None
            dist.gather(pack.cpu(), gather_list=list(recv.unbind(0)) if is_root else None, dst=self.root)
            if is_root:
                self.recv.copy_(recv)
        else:
            dist.gather(pack, gather_list=list(self.recv.unbind(0)) if is_root else None, dst=self.root)
        if is_root:
            self.unpack(framebuffer)
        return framebuffer


def _pad(rows, pad):
    import numpy as np

    return np.concatenate([rows, np.zeros(pad, dtype=np.int64)]) if pad else rows
