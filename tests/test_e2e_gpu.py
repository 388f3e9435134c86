"""BASELINE configs 4 and 5 on the GPU.

config 5: 10 000 000 precomputed 256-bit hashes, all pairs at threshold 32 (one GPU sweeps all of it here; on a node the same
          call runs with part = rank, nparts = world) -- the edge set is known from the construction.
config 4: images -> PDQ -> all-gather -> sweep -> groups as ONE call (rupphash_amd.dist.hash_and_group_device) with the real
          Engine: at world 1 in this process, and with 3 ranks started as fresh child processes that share the one GPU
          (gloo: RCCL refuses two ranks on one device; the collectives are the same calls), against oracle.group_pdq.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def eng():
    from rupphash_amd import Engine

    e = Engine(0)
    yield e
    e.close()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _expected_cluster_edges(oracle, n, nc):
    pops = [0, 1, 2, 8, 16]
    want = []
    for c in range(nc):
        idx = [oracle.synth_cluster_index(n, c, j) for j in range(5)]
        masks = []
        for j in range(5):
            m = 0
            for t in range(pops[j]):
                m |= 1 << ((c * 31 + j * 11 + t * 37) & 255)
            masks.append(m)
        for a in range(5):
            for b in range(a + 1, 5):
                i, j = sorted((idx[a], idx[b]))
                want.append((i, j, bin(masks[a] ^ masks[b]).count("1")))
    i, j = sorted((oracle.synth_cluster_index(n, nc, 0), oracle.synth_cluster_index(n, nc, 1)))
    want.append((i, j, 32))
    return sorted(want)


def test_ten_million_hashes_threshold_32(eng, oracle):
    """BASELINE config 5 (5e13 pairs, ~2 s on one MI355X): every edge of the construction and nothing else, and the same
    edge set when the sweep is cut into 8 parts as 8 ranks would run it."""
    from rupphash_amd import EDGE_DTYPE

    n, nc = 10_000_000, 1000
    d_h = eng.dev_alloc(n * 32)
    cap = 1 << 16
    d_e = eng.dev_alloc(cap * 12)
    d_c = eng.dev_alloc(8)
    try:
        eng.synth_hashes_dev(d_h, 0, n, n, n_clusters=nc)
        eng.dev_memset(d_c, 0, 8)
        eng.hamming_all_pairs_dev(d_h, n, 32, d_e, cap, d_c)
        eng.synchronize()
        cnt = np.zeros(1, np.uint64)
        eng.dev_download(cnt, d_c)
        edges = np.zeros(int(cnt[0]), EDGE_DTYPE)
        eng.dev_download(edges, d_e)
        # 8 ranks' shares, one after the other, into the same buffer: one shared cursor, as on one device
        eng.dev_memset(d_c, 0, 8)
        for part in (0, 3, 7):
            eng.hamming_all_pairs_dev(d_h, n, 32, d_e, cap, d_c, part=part, nparts=8)
        eng.synchronize()
        eng.dev_download(cnt, d_c)
        some = np.zeros(int(cnt[0]), EDGE_DTYPE)
        eng.dev_download(some, d_e)
    finally:
        for p in (d_h, d_e, d_c):
            eng.dev_free(p)
    got = sorted((int(e["i"]), int(e["j"]), int(e["d"])) for e in edges)
    want = _expected_cluster_edges(oracle, n, nc)
    assert got == want
    part_edges = sorted((int(e["i"]), int(e["j"]), int(e["d"])) for e in some)
    assert len(set(part_edges)) == len(part_edges) and set(part_edges) <= set(want) and 0 < len(part_edges) < len(want)


def _oracle_production_groups(oracle, first, n, sim):
    imgs = oracle.synth_images(first, n)
    secs, hashes, qual = oracle.bench_pdq(imgs, min(16, os.cpu_count() or 1))
    dih = np.zeros((n, 8, 32), np.uint8)
    step = 256
    for a in range(0, n, step):  # coefficients of a slice at a time (the full-coefficient oracle call is single-threaded)
        _, _, c = oracle.pdq_batch_rgb(imgs[a:a + step], want_coeffs=True)
        for k in range(len(c)):
            dih[a + k] = oracle.dihedral_hashes(c[k])
    stored = np.clip(np.floor(qual * np.float32(100.0) + np.float32(0.5)), 0, 100).astype(np.int32)
    edges, groups = oracle.group_pdq(hashes, sim, variants=dih, quality=stored)
    return hashes, groups, len(edges)


N_E2E, FIRST_E2E, SIM_E2E = 1500, 500, 32  # images 998/999 and 1998/1999 are the near-duplicate pairs


@pytest.fixture(scope="module")
def e2e_expected(oracle):
    return _oracle_production_groups(oracle, FIRST_E2E, N_E2E, SIM_E2E)


def test_hash_and_group_world_1_real_engine(eng, oracle, e2e_expected):
    import torch

    from rupphash_amd import dist as D

    hashes, want_groups, want_cmp = e2e_expected
    dev = torch.device("cuda", 0)
    imgs = torch.empty((N_E2E, 512 * 512 * 3), dtype=torch.uint8, device=dev)
    eng.synth_images_dev(imgs.data_ptr(), FIRST_E2E, N_E2E, 512, 512, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    groups, info = D.hash_and_group_device(eng, imgs, N_E2E, SIM_E2E, None, variants=True)
    assert groups == want_groups and info["edges_total"] == want_cmp
    near = [[998 - FIRST_E2E, 999 - FIRST_E2E], [1998 - FIRST_E2E, 1999 - FIRST_E2E]]
    assert all(g in groups for g in near), groups
    # plain all-pairs form of the same path (numpy in / out, Engine functions injected where the gloo tests inject the oracle)
    hash_fn, sweep_fn, group_fn = D.engine_fns(eng)
    host_imgs = imgs.cpu().numpy().reshape(N_E2E, 512, 512, 3)
    g2 = D.hash_and_group(0, N_E2E, lambda first, count: host_imgs[first:first + count], hash_fn, SIM_E2E, sweep_fn, group_fn, None)
    assert g2 == oracle.group_pdq(hashes, SIM_E2E)[1]


@pytest.mark.parametrize("world", [3])
def test_hash_and_group_three_ranks_share_one_gpu(oracle, e2e_expected, tmp_path, world):
    hashes, want_groups, want_cmp = e2e_expected
    port = _free_port()
    out = str(tmp_path / "groups.json")
    worker = os.path.join(HERE, "dist_gpu_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), str(port), str(N_E2E), str(FIRST_E2E), str(SIM_E2E), "1", "gloo", out],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    with open(out) as f:
        res = json.load(f)
    assert res["world"] == world and res["info"]["ranks_in_collective"] == world
    assert res["groups"] == want_groups and res["info"]["edges_total"] == want_cmp


N_JPEG, SIM_JPEG = 90, 40


def test_scan_of_jpeg_files_sharded_over_three_ranks(eng, oracle, tmp_path):
    """the reference's scan with JPEG inputs (scanner.rs:1146-1551) as rupphash_amd.dist.scan_jpeg_files_and_group: every rank decodes and
    hashes its shard of the files on the GPU, one all-gather of dihedral blocks and flags, every rank sweeps its share, rank 0 groups.
    90 files (pairs of one picture at two qualities, some progressive, one unreadable) on one rank in this process and on three ranks
    that share the GPU (gloo) must give the groups the single-process production path gives (rph_jpeg_pdq_hash_batch over all files +
    group_files_pdq), with the unreadable file in no group."""
    from dist_jpeg_worker import make_files
    from rupphash_amd import dist as D
    from rupphash_amd import scanner

    files = make_files(eng, 0, N_JPEG)
    out = eng.jpeg_pdq_hash_batch(files, threads=4, want_quality=True, want_coeffs=True)
    assert out["valid"].sum() == N_JPEG - 1 and out["valid"][7] == 0
    ok = [k for k in range(N_JPEG) if out["valid"][k]]
    want, _ = scanner.group_with_pdqhash(out["hash"][ok], SIM_JPEG, coefficients=out["coeffs"][ok], has_features=np.ones(len(ok), bool),
                                         quality=[scanner.stored_quality(q) for q in out["quality"][ok]], engine=eng)
    want = [[ok[i] for i in g] for g in want]
    assert len(want) >= N_JPEG // 2 - 2 and all(len(g) >= 2 for g in want)  # (the pairs: same picture at quality 90 and 60)
    groups, info = D.scan_jpeg_files_and_group(eng, files, N_JPEG, SIM_JPEG, None, threads=4)
    assert groups == want and info["unreadable"] == [7]
    world, port, res_file = 3, _free_port(), str(tmp_path / "jpeg_groups.json")
    worker = os.path.join(HERE, "dist_jpeg_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), str(port), str(N_JPEG), str(SIM_JPEG), "gloo", res_file], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    with open(res_file) as f:
        res = json.load(f)
    assert res["world"] == world and res["info"]["ranks_in_collective"] == world
    assert res["groups"] == want and res["info"]["unreadable"] == [7]
