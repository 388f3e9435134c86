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


def _backend_is_device_only(dist):
    """True when the process group can only move device tensors (backend "nccl" = RCCL alone, no CPU backend beside it)"""
    b = str(dist.get_backend()).lower()
    return "nccl" in b and "gloo" not in b


def _collective_tensor(t, dist, device):
    """the tensor in the memory space this process group's collectives accept: device memory under nccl (RCCL), host under gloo"""
    import torch

    if _backend_is_device_only(dist):
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        return t.to(device)
    return t.cpu()


def all_gather_rows(local_rows, n_total, dist=None, device=None):
    """One all-gather of per-rank row shards (rows of any byte width: 32-byte hashes, 256-byte dihedral blocks, 1-byte flags)
    into the full (n_total, width) array on every rank.

    Shards must follow shard_range().  numpy in -> numpy out; torch tensor in -> tensor out on the input's device.  The
    collective itself runs where the backend needs it (device memory under nccl = RCCL, host memory under gloo).
    Unequal shards (n_total % world != 0) are padded to the largest shard for the collective."""
    import torch

    if dist is None or dist.get_world_size() == 1:
        return local_rows
    world, rank = dist.get_world_size(), dist.get_rank()
    is_tensor = isinstance(local_rows, torch.Tensor)
    t = local_rows if is_tensor else torch.from_numpy(np.ascontiguousarray(local_rows, np.uint8))
    home = t.device
    width = int(np.prod(t.shape[1:])) if t.dim() > 1 else 1  # (an empty shard -- more ranks than rows -- has no elements to infer it from)
    t = t.reshape(t.shape[0], width)
    sizes = [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]
    assert t.shape[0] == sizes[rank], (t.shape, sizes, rank)
    t = _collective_tensor(t, dist, device)
    big = max(sizes)
    if big != t.shape[0]:
        pad = torch.zeros((big - t.shape[0], width), dtype=t.dtype, device=t.device)
        t = torch.cat([t, pad])
    out = torch.empty((world * big, width), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t.contiguous())
    if any(s != big for s in sizes):
        out = torch.cat([out[r * big:r * big + sizes[r]] for r in range(world)])
    if is_tensor:
        return out.to(home)
    return out.cpu().numpy()


def all_gather_hashes(local_hashes, n_total, dist=None, device=None):
    """The exchange step of the path: (n_local, 32) hash shards -> (n_total, 32) on every rank (see all_gather_rows)."""
    import torch

    if dist is None or dist.get_world_size() == 1:
        return local_hashes
    if isinstance(local_hashes, torch.Tensor):
        return all_gather_rows(local_hashes.reshape(-1, 32), n_total, dist, device)
    return all_gather_rows(np.ascontiguousarray(local_hashes, np.uint8).reshape(-1, 32), n_total, dist, device)


def gather_edges(local_edges, dist=None, dst=0, device=None):
    """variable-length edge lists -> rank `dst` (None elsewhere).

    Works on every backend: the counts travel in one all-gather of an int64, the edges in one all-gather of buffers padded
    to the largest count (nccl has no CPU path, and `gather` of unequal sizes is not portable; the lists are tiny)."""
    import torch

    local_edges = np.ascontiguousarray(local_edges, EDGE_DTYPE)
    if dist is None or dist.get_world_size() == 1:
        return local_edges
    world, rank = dist.get_world_size(), dist.get_rank()
    cnt = _collective_tensor(torch.tensor([len(local_edges)], dtype=torch.int64), dist, device)
    counts_t = torch.empty(world, dtype=torch.int64, device=cnt.device)
    dist.all_gather_into_tensor(counts_t, cnt)
    counts = [int(c) for c in counts_t.cpu().tolist()]
    big = max(max(counts), 1)
    buf = np.zeros(big, EDGE_DTYPE)
    buf[: len(local_edges)] = local_edges
    t = _collective_tensor(torch.from_numpy(buf.view(np.uint8).reshape(big, EDGE_DTYPE.itemsize).copy()), dist, device)
    out = torch.empty((world * big, EDGE_DTYPE.itemsize), dtype=torch.uint8, device=t.device)
    dist.all_gather_into_tensor(out, t.contiguous())
    if rank != dst:
        return None
    host = out.cpu().numpy().reshape(world, big, EDGE_DTYPE.itemsize)
    parts = [np.ascontiguousarray(host[r, : counts[r]]).reshape(-1).view(EDGE_DTYPE) for r in range(world)]
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


def engine_fns(eng, semantics="union_find"):
    """(hash_fn, sweep_fn, group_fn) of a real Engine for hash_and_group / grouped_all_pairs: the kernels behind the C ABI"""

    def hash_fn(imgs):
        return eng.pdq_hash_batch(imgs, want_quality=False)["hash"]

    def sweep_fn(all_hashes, thr, part, nparts):
        return eng.hamming_all_pairs(all_hashes, thr, part=part, nparts=nparts)

    group_fn = eng.union_find_groups if semantics == "union_find" else eng.find_groups_from_edges
    return hash_fn, sweep_fn, group_fn


# ---------------------------------------------------------------------------------------------------------------------
# Device-resident form of the whole path (BASELINE config 4), production grouping semantics
# ---------------------------------------------------------------------------------------------------------------------
_side_streams = {}


def stored_quality_lowconf(quality):
    """quality in [0,1] (device tensor) -> uint8 low-confidence flags: round(q * 100) clamped to 0..100 (scanner.rs:1416-1417,
    round half away from zero) < PDQ_MIN_QUALITY (scanner.rs:1588-1594)"""
    import torch

    stored = torch.clamp(torch.floor(quality * 100.0 + 0.5), 0.0, 100.0)
    return (stored < 50.0).to(torch.uint8)


def hash_and_group_device(eng, images, n_total, similarity, dist=None, variants=True, w=512, h=512, channels=3, edge_cap=1 << 20,
                          timings=None):
    """Images resident in HBM -> PDQ hash -> all-gather -> sweep -> groups, the reference's scan-then-group call
    (scanner.rs:1146-1551: hash every file, then group_files_generic) with the data never leaving the devices until the
    (tiny) edge lists go to rank 0.

    images: torch uint8 tensor on this rank's GPU holding ITS shard (shard_range(n_total, rank, world)) of the image sequence,
            (n_local, h * w * channels) or (n_local, h, w, channels), packed.
    variants=True: production semantics of group_files_generic + PdqStrategy (8 dihedral variants per file as rows, the
            low-quality rule, union-find; scanner.rs:1596-1823).  The exchange is then ONE all-gather of the per-file
            dihedral blocks (8 x 32 B; slot 0 is the hash itself) plus one of the 1-byte low-confidence flags.
    variants=False: plain all-pairs over the hashes (BASELINE config 5's sweep), exchange = the 32-byte hashes.
    Kernels, collectives and copies are enqueued on torch's current stream.
    Returns (groups or None, info): groups on rank 0 (connected components, members ascending, by first member)."""
    import time

    import torch

    dev = images.device
    caller = torch.cuda.current_stream(dev)
    if caller.cuda_stream == 0:
        # The library takes a null stream handle to mean "the context's own stream", which the legacy default stream does not
        # order with: run on an explicit side stream instead, entered behind the caller's work and joined again at the end.
        side = _side_streams.get(dev)
        if side is None:
            side = _side_streams[dev] = torch.cuda.Stream(device=dev)
        side.wait_stream(caller)
        with torch.cuda.stream(side):
            out = hash_and_group_device(eng, images, n_total, similarity, dist, variants, w, h, channels, edge_cap, timings)
        caller.wait_stream(side)
        return out
    world = 1 if dist is None else dist.get_world_size()
    rank = 0 if dist is None else dist.get_rank()
    stream = caller.cuda_stream
    lo, hi = shard_range(n_total, rank, world)
    n_local = hi - lo
    assert images.shape[0] == n_local, (images.shape, lo, hi)
    t0 = time.perf_counter()
    d_hash = torch.empty((n_local, 32), dtype=torch.uint8, device=dev)
    d_q = torch.empty(n_local, dtype=torch.float32, device=dev)
    d_dih = torch.empty((n_local, 8, 32), dtype=torch.uint8, device=dev) if variants else None
    eng.pdq_hash_batch_dev(images.data_ptr(), n_local, w, h, channels, d_hash.data_ptr(), d_quality=d_q.data_ptr(),
                           d_dihedral=d_dih.data_ptr() if variants else None, stream=stream)
    low = stored_quality_lowconf(d_q)
    if timings is not None:  # (per-stage figures cost a synchronisation each: only when asked for)
        torch.cuda.synchronize(dev)
        timings["hash_ms"] = (time.perf_counter() - t0) * 1e3
        t1 = time.perf_counter()
    # ---- the one exchange step (RCCL all-gather under nccl)
    if variants:
        all_dih = all_gather_rows(d_dih.reshape(n_local, 256), n_total, dist, dev).reshape(n_total, 8, 32)
        all_low = all_gather_rows(low.reshape(n_local, 1), n_total, dist, dev).reshape(n_total).contiguous()
        all_hash = all_dih[:, 0, :].contiguous()
    else:
        all_hash = all_gather_rows(d_hash, n_total, dist, dev)
        all_dih, all_low = None, None
    if timings is not None:
        torch.cuda.synchronize(dev)
        timings["hash_and_exchange_s"] = time.perf_counter() - t0
        timings["allgather_ms"] = (time.perf_counter() - t1) * 1e3
        t2 = time.perf_counter()
    # ---- this rank's share of the block pairs; no communication
    cap = int(edge_cap)
    while True:
        d_edges = torch.empty((cap, EDGE_DTYPE.itemsize), dtype=torch.uint8, device=dev)
        d_count = torch.zeros(1, dtype=torch.int64, device=dev)
        if variants:
            eng.hamming_variant_pairs_dev(all_dih.data_ptr(), 8, all_hash.data_ptr(), n_total, similarity, d_edges.data_ptr(), cap,
                                          d_count.data_ptr(), d_low_conf=all_low.data_ptr(), part=rank, nparts=world, stream=stream)
        else:
            eng.hamming_all_pairs_dev(all_hash.data_ptr(), n_total, similarity, d_edges.data_ptr(), cap, d_count.data_ptr(),
                                      part=rank, nparts=world, stream=stream)
        found = int(d_count.item())
        if found <= cap:
            break
        cap = found + found // 8 + 1024  # rare: more edges than expected, sweep again into a buffer that fits
    if timings is not None:
        timings["sweep_ms"] = (time.perf_counter() - t2) * 1e3  # (d_count.item() above waited for the sweep)
    local_edges = d_edges[:found].cpu().numpy().reshape(-1).view(EDGE_DTYPE) if found else np.zeros(0, EDGE_DTYPE)
    merged = gather_edges(local_edges, dist, dst=0, device=dev)  # tiny
    info = {"n_local": n_local, "edges_local": found, "ranks_in_collective": world}
    if rank != 0:
        return None, info
    info["edges_total"] = int(len(merged))
    groups = eng.union_find_groups(merged, n_total)  # serial in the reference too (scanner.rs:1781-1817)
    if timings is not None:
        timings["total_s"] = time.perf_counter() - t0
    return groups, info


def scan_jpeg_files_and_group(eng, files, n_total, similarity, dist=None, flavour=0, threads=0, edge_cap=1 << 20, timings=None):
    """JPEG files in host memory -> decode + PDQ hash on this rank's GPU -> all-gather -> sweep -> groups: the reference's scan
    (scanner.rs:1146-1551: load_image_fast + generate_pdq_features for every file on the rayon pool, then group_files_generic) with
    the files sharded over the ranks of a node, one process per GPU.

    files: THIS rank's shard (shard_range(n_total, rank, world)) of the file sequence, a list of JPEG byte strings.  Every rank
    decodes and hashes its own files with rph_jpeg_pdq_hash_batch (entropy decoding on the device when the shard is large enough,
    hash + quality + the 8 dihedral hashes per file); the one exchange step is an all-gather of the 256-byte dihedral blocks, the
    1-byte low-confidence flags (scanner.rs:1588-1594) and the 1-byte validity flags; every rank then sweeps its share of the block
    pairs (rph_hamming_variant_pairs_dev) and rank 0 unites the edges.  A file that cannot be decoded takes part in nothing, as in
    the reference, where it never gets a hash: its rows are replaced by a pattern of its own (a function of its index) that lies
    ~128 bits from everything, and it is listed in info["unreadable"].
    Returns (groups or None, info): groups on rank 0 (connected components of more than one file, members ascending, by first member)."""
    import time

    import torch

    world = 1 if dist is None else dist.get_world_size()
    rank = 0 if dist is None else dist.get_rank()
    lo, hi = shard_range(n_total, rank, world)
    n_local = hi - lo
    assert len(files) == n_local, (len(files), lo, hi)
    dev = torch.device("cuda", eng.device)
    t0 = time.perf_counter()
    if n_local:
        out = eng.jpeg_pdq_hash_batch(files, flavour=flavour, threads=threads, want_quality=True, want_dihedral=True)
        dih, quality, valid = out["dihedral"], out["quality"], out["valid"].astype(np.uint8)
    else:
        dih, quality, valid = np.zeros((0, 8, 32), np.uint8), np.zeros(0, np.float32), np.zeros(0, np.uint8)
    for k in np.nonzero(valid == 0)[0]:  # unreadable (or below 5 px): unmatchable rows of its own
        dih[k, :, :] = np.random.default_rng(0x9E3779B97F4A7C15 ^ int(lo + k)).integers(0, 256, 32, dtype=np.uint8)
    stored = np.clip(np.floor(quality.astype(np.float32) * np.float32(100.0) + np.float32(0.5)), 0, 100)
    low = ((stored < 50) | (valid == 0)).astype(np.uint8)
    if timings is not None:
        timings["decode_and_hash_ms"] = (time.perf_counter() - t0) * 1e3
        t1 = time.perf_counter()
    # (the library reads a null stream handle as "the context's own stream", which torch's default stream is not ordered with: a side
    # stream carries the copies, the collectives and the sweep)
    side = _side_streams.get(dev)
    if side is None:
        side = _side_streams[dev] = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        d_dih = torch.from_numpy(np.ascontiguousarray(dih.reshape(n_local, 256))).to(dev)
        d_low = torch.from_numpy(low.reshape(n_local, 1)).to(dev)
        d_valid = torch.from_numpy(valid.reshape(n_local, 1)).to(dev)
        all_dih = all_gather_rows(d_dih, n_total, dist, dev).reshape(n_total, 8, 32).contiguous()
        all_low = all_gather_rows(d_low, n_total, dist, dev).reshape(n_total).contiguous()
        all_valid = all_gather_rows(d_valid, n_total, dist, dev).reshape(n_total)
        all_hash = all_dih[:, 0, :].contiguous()
        if timings is not None:
            side.synchronize()
            timings["allgather_ms"] = (time.perf_counter() - t1) * 1e3
            t2 = time.perf_counter()
        cap = int(edge_cap)
        while True:
            d_edges = torch.empty((cap, EDGE_DTYPE.itemsize), dtype=torch.uint8, device=dev)
            d_count = torch.zeros(1, dtype=torch.int64, device=dev)
            eng.hamming_variant_pairs_dev(all_dih.data_ptr(), 8, all_hash.data_ptr(), n_total, similarity, d_edges.data_ptr(), cap, d_count.data_ptr(),
                                          d_low_conf=all_low.data_ptr(), part=rank, nparts=world, stream=side.cuda_stream)
            found = int(d_count.item())
            if found <= cap:
                break
            cap = found + found // 8 + 1024  # rare: more edges than expected, sweep again into a buffer that fits
        if timings is not None:
            timings["sweep_ms"] = (time.perf_counter() - t2) * 1e3
        local_edges = d_edges[:found].cpu().numpy().reshape(-1).view(EDGE_DTYPE) if found else np.zeros(0, EDGE_DTYPE)
        unreadable = [int(i) for i in np.nonzero(all_valid.cpu().numpy() == 0)[0]]
    torch.cuda.current_stream(dev).wait_stream(side)
    merged = gather_edges(local_edges, dist, dst=0, device=dev)
    info = {"n_local": n_local, "edges_local": found, "ranks_in_collective": world}
    if rank != 0:
        return None, info
    info["edges_total"] = int(len(merged))
    info["unreadable"] = unreadable
    groups = eng.union_find_groups(merged, n_total)
    if timings is not None:
        timings["total_s"] = time.perf_counter() - t0
    return groups, info
