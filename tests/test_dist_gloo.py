"""Multi-GPU control flow on CPU: world_size-2 gloo runs of rupphash_amd.dist with the oracle standing in
for the kernels (hash_fn / sweep_fn are injected), plus the tile-pair partition arithmetic."""
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("n,tile,nparts", [(1, 4, 1), (10, 4, 3), (4096, 1024, 8), (5000, 1024, 3), (1_000_448, 1024, 8), (777, 64, 5)])
def test_tile_pairs_partition_is_exact(n, tile, nparts):
    from rupphash_amd import dist as D

    nt = (n + tile - 1) // tile
    total = D.n_tile_pairs(n, tile)
    assert total == nt * (nt + 1) // 2
    if total > 200_000:  # spot-check the closed form on a huge triangle
        for p in [0, 1, nt - 1, nt, total // 2, total - 1]:
            i, j = D.tile_pair(p, nt)
            assert 0 <= i <= j < nt and i * nt - i * (i - 1) // 2 + (j - i) == p
        return
    seen = []
    for part in range(nparts):
        seen += D.tile_pairs_of_part(n, part, nparts, tile)
    assert len(seen) == total and len(set(seen)) == total
    assert set(seen) == {(i, j) for i in range(nt) for j in range(i, nt)}


@pytest.mark.parametrize("n,tile,nparts,s", [(1, 4, 1, 8), (10, 4, 3, 2), (5000, 64, 3, 8), (4096, 1024, 8, 1), (777, 64, 5, 3), (70000, 64, 4, 8)])
def test_segment_blocks_partition_is_exact(n, tile, nparts, s):
    """the int8 MFMA kernel's enumeration: every (row tile, column tile >= row tile) pair lies in exactly one segment block"""
    from rupphash_amd import dist as D

    nt = (n + tile - 1) // tile
    covered = []
    for part in range(nparts):
        for (i, j0) in D.seg_blocks_of_part(n, part, nparts, s, tile):
            assert 0 <= i <= j0 < nt and (j0 - i) % s == 0
            covered += [(i, j) for j in range(j0, min(j0 + s, nt))]
    assert len(covered) == nt * (nt + 1) // 2 and set(covered) == {(i, j) for i in range(nt) for j in range(i, nt)}


def test_shard_ranges_cover():
    from rupphash_amd import dist as D

    for n, w in [(10, 3), (8, 8), (1_000_000, 8), (5, 8)]:
        r = [D.shard_range(n, k, w) for k in range(w)]
        assert r[0][0] == 0 and r[-1][1] == n and all(r[k][1] == r[k + 1][0] for k in range(w - 1))


def _oracle_sweep_factory(tile):
    import oracle
    from rupphash_amd import EDGE_DTYPE
    from rupphash_amd import dist as D

    def sweep(all_hashes, thr, part, nparts):
        h = np.ascontiguousarray(all_hashes, np.uint8).reshape(-1, 32)
        n = len(h)
        out = []
        for (I, J) in D.tile_pairs_of_part(n, part, nparts, tile):
            rows = range(I * tile, min(n, (I + 1) * tile))
            cols = range(J * tile, min(n, (J + 1) * tile))
            for i in rows:
                for j in cols:
                    if j > i:
                        d = oracle.hamming256(h[i], h[j])
                        if d <= thr:
                            out.append((i, j, d, 0))
        return np.array(out, EDGE_DTYPE) if out else np.zeros(0, EDGE_DTYPE)

    return sweep


def _worker(rank, world, port, tmpdir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    import oracle
    from rupphash_amd import _lib
    from rupphash_amd import dist as D
    from rupphash_amd.engine import Engine

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # host-only entry points (union-find) need no GPU context
        eng = Engine.__new__(Engine)
        eng.L = _lib.load()
        eng.ctx = None
        tile = 16
        sweep = _oracle_sweep_factory(tile)
        # --- config-5 shape: precomputed hashes live sharded; uneven shards (n % world != 0)
        n, nc = 301, 12
        full = oracle.synth_hashes(0, n, n, n_clusters=nc)
        lo, hi = D.shard_range(n, rank, world)
        groups = D.grouped_all_pairs(full[lo:hi], n, 32, sweep, eng.union_find_groups, dist)
        if rank == 0:
            want = oracle.group_pdq(full, 32)[1]
            assert groups == want and len(groups) == nc + 1
        else:
            assert groups is None
        # --- config-4 shape: hash own image range, all-gather hashes, sweep, group
        n_img = 6

        def make(first, count):
            return oracle.synth_images(first, count, 64, 48)

        def hash_fn(imgs):
            return oracle.pdq_batch_rgb(imgs)[0]

        g2 = D.hash_and_group(997, n_img, make, hash_fn, 40, sweep, eng.union_find_groups, dist)
        if rank == 0:
            hashes = hash_fn(make(997, n_img))
            assert g2 == oracle.group_pdq(hashes, 40)[1]
            assert [1, 2] in g2 or any({1, 2} <= set(g) for g in g2)  # images 998 / 999 share their blocks
        with open(os.path.join(tmpdir, f"ok{rank}"), "w") as f:
            f.write("ok")
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo_grouping(tmp_path):
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def test_world_size_8_gloo_grouping(tmp_path):
    """the driver's N = 8 shape: eight ranks, uneven shards (301 hashes, 6 images over 8 ranks: some ranks hold one image, some none)"""
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_worker, args=(8, port, str(tmp_path)), nprocs=8, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(8))
