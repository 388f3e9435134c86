"""Multi-GPU sharding of the hot path: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" on CPU for the tests).

The path partitions: images are independent, so rank r hashes the contiguous range
[r*N/W, (r+1)*N/W) with no communication; the only exchange step is ONE all-gather of the 32-byte hashes
(N/W x 32 B per rank: 4 MB at N = 1M, W = 8), after which every rank sweeps its round-robin share of the
upper-triangular tile pairs (part = rank, nparts = world) and the few edges are gathered to rank 0 for the
serial union-find / greedy clustering the reference also runs serially.

Nothing here computes: hashing and the sweep are the C-ABI calls of `Engine`; this module only decides who
does what and moves hashes/edges.  `sweep_fn` / `hash_fn` are injectable so the world_size-2 gloo tests can
drive the same control flow on CPU with the oracle standing in for the kernels.
"""
import numpy as np

from .engine import EDGE_DTYPE

TILE = 1024  # files per tile of the sweep kernel (hamming_kernels.hip: T_FILES)


def shard_range(n, rank, world):
    """contiguous range of rank `rank`: [lo, hi)"""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def n_tile_pairs(n, tile=TILE):
    nt = (n + tile - 1) // tile
    return nt * (nt + 1) // 2


def tile_pair(p, nt):
    """linear index -> (I, J), I <= J, row-major in I: the enumeration of hamming_kernels.hip::tile_pair"""
    i = int(((2 * nt + 1) - ((2 * nt + 1) ** 2 - 8 * p) ** 0.5) // 2)
    i = max(0, min(i, nt - 1))

    def off(ii):
        return ii * nt - ii * (ii - 1) // 2

    while i > 0 and off(i) > p:
        i -= 1
    while i + 1 < nt and off(i + 1) <= p:
        i += 1
    return i, i + (p - off(i))


def tile_pairs_of_part(n, part, nparts, tile=TILE):
    """the tile pairs rank `part` of `nparts` sweeps: p = part, part + nparts, ... (round robin)"""
    nt = (n + tile - 1) // tile
    return [tile_pair(p, nt) for p in range(part, n_tile_pairs(n, tile), nparts)]


def seg_g(m, s):
    """blocks that cover the last m row tiles when a row tile is cut into segments of s column tiles"""
    q, r = divmod(m, s)
    return s * q * (q + 1) // 2 + (q + 1) * r


def seg_block(b, nt, s):
    """block index -> (I, J0): the enumeration of hamming_kernels.hip::seg_block (int8 MFMA kernel): row tile I against the
    column tiles [J0, min(J0 + s, nt)), J0 = I, I + s, ...; row-major in I"""
    total = seg_g(nt, s)

    def f(ii):
        return total - seg_g(nt - ii, s)

    big = nt + 0.5 * s
    i = int(big - max(big * big - 2.0 * s * b, 0.0) ** 0.5)
    i = max(0, min(i, nt - 1))
    while i > 0 and f(i) > b:
        i -= 1
    while i + 1 < nt and f(i + 1) <= b:
        i += 1
    return i, i + (b - f(i)) * s


def seg_blocks_of_part(n, part, nparts, s, tile=TILE):
    """the (I, J0) segment blocks rank `part` of `nparts` sweeps (round robin over the block index)"""
    nt = (n + tile - 1) // tile
    return [seg_block(b, nt, s) for b in range(part, seg_g(nt, s), nparts)]


def all_gather_hashes(local_hashes, n_total, dist=None, device_tensor=False):
    """One all-gather of the per-rank hash shards into the full (n_total, 32) array on every rank.

    Shards must follow shard_range().  With `dist is None` (single process) the input is returned.
    Unequal shards (n_total % world != 0) are padded to the largest shard for the collective."""
    import torch

    if dist is None or dist.get_world_size() == 1:
        return local_hashes
    world, rank = dist.get_world_size(), dist.get_rank()
    t = local_hashes if isinstance(local_hashes, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(local_hashes, np.uint8))
    t = t.reshape(-1, 32)
    sizes = [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]
    assert t.shape[0] == sizes[rank], (t.shape, sizes, rank)
    big = max(sizes)
    if big != t.shape[0]:
        pad = torch.zeros((big - t.shape[0], 32), dtype=torch.uint8, device=t.device)
        t = torch.cat([t, pad])
    out = torch.empty((world * big, 32), dtype=torch.uint8, device=t.device)
    dist.all_gather_into_tensor(out, t.contiguous())
    if any(s != big for s in sizes):
        out = torch.cat([out[r * big:r * big + sizes[r]] for r in range(world)])
    return out if isinstance(local_hashes, torch.Tensor) else out.cpu().numpy()


def gather_edges(local_edges, dist=None, dst=0):
    """variable-length edge lists -> rank `dst` (None elsewhere)"""
    import torch

    local_edges = np.ascontiguousarray(local_edges, EDGE_DTYPE)
    if dist is None or dist.get_world_size() == 1:
        return local_edges
    world, rank = dist.get_world_size(), dist.get_rank()
    counts = [None] * world
    dist.all_gather_object(counts, int(len(local_edges)))
    big = max(counts) if counts else 0
    buf = np.zeros(max(big, 1), EDGE_DTYPE)
    buf[: len(local_edges)] = local_edges
    t = torch.from_numpy(buf.view(np.uint8).reshape(-1, EDGE_DTYPE.itemsize).copy())
    gathered = [torch.empty_like(t) for _ in range(world)] if rank == dst else None
    dist.gather(t, gathered, dst=dst)
    if rank != dst:
        return None
    parts = [g.numpy().reshape(-1).view(EDGE_DTYPE)[: counts[r]] for r, g in enumerate(gathered)]
    return np.concatenate(parts) if parts else np.zeros(0, EDGE_DTYPE)


def grouped_all_pairs(local_hashes, n_total, threshold, sweep_fn, group_fn, dist=None):
    """End-to-end grouping of hashes that live sharded across ranks (BASELINE configs 4 / 5).

    sweep_fn(all_hashes, threshold, part, nparts) -> edges of this rank's share of the tile pairs
        (Engine.hamming_all_pairs on a GPU)
    group_fn(edges, n_total) -> groups   (Engine.union_find_groups or Engine.find_groups_from_edges)
    Returns the groups on rank 0, None elsewhere."""
    world = 1 if dist is None else dist.get_world_size()
    rank = 0 if dist is None else dist.get_rank()
    all_hashes = all_gather_hashes(local_hashes, n_total, dist)      # the one exchange step
    edges = sweep_fn(all_hashes, threshold, rank, world)             # no communication
    merged = gather_edges(edges, dist, dst=0)                        # tiny
    if rank != 0:
        return None
    return group_fn(merged, n_total)


def hash_and_group(first_image, n_images_total, make_images_fn, hash_fn, threshold, sweep_fn, group_fn, dist=None):
    """BASELINE config 4: every rank hashes its own contiguous image range, then grouped_all_pairs."""
    world = 1 if dist is None else dist.get_world_size()
    rank = 0 if dist is None else dist.get_rank()
    lo, hi = shard_range(n_images_total, rank, world)
    hashes = hash_fn(make_images_fn(first_image + lo, hi - lo))      # no communication
    return grouped_all_pairs(hashes, n_images_total, threshold, sweep_fn, group_fn, dist)
