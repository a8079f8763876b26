"""Multi-GPU partition of the hot path: tiles are independent, so rank r of N renders
host tiles r, r+N, r+2N, ... (round-robin over the reference's column-major tile order, which interleaves
cheap wall tiles and expensive mesh tiles evenly) into a zero-initialised full-size f64 framebuffer, and ONE
reduce(sum) to the root assembles the frame.  Every pixel is non-zero on exactly one rank, so the reduced
image is bit-identical to the single-GPU image whatever the reduction order."""


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
