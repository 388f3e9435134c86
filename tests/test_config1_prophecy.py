"""BASELINE config 1: PDQ-hash tests/Prophecy_*.jpg and report their Hamming distance.

The two JPEGs are the reference's own test data (tests/golden/README.md).  Decode is Pillow here (the reference uses
zune-jpeg) and the > 512 px pre-downsample is a restatement of third-party code, so the hashes are PARITY UNPINNED against
the Rust binary; the distance is reported, and GPU == CPU oracle is asserted on the same decoded pixels."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load():
    from PIL import Image

    return [np.asarray(Image.open(os.path.join(GOLDEN, f"Prophecy_Has_Been_Fulfilled_{k}.jpg")).convert("RGB")) for k in (1, 2)]


def test_prophecy_pair_on_cpu_oracle(oracle):
    imgs = load()
    assert imgs[0].shape == (768, 780, 3) and imgs[1].shape == (768, 780, 3)
    hashes, quals = [], []
    for im in imgs:
        rc, coeffs, q = oracle.pdq_features(im)
        assert rc == 0
        hashes.append(oracle.to_hash(coeffs))
        quals.append(q)
    d = oracle.hamming256(hashes[0], hashes[1])
    print(f"config 1 (CPU oracle, Pillow decode): quality {quals}, Hamming distance {d}")
    assert quals == [1.0, 1.0]
    assert d <= 63, "the two files are near duplicates: within the reference's MAX_SIMILARITY_256"


@pytest.mark.gpu
def test_prophecy_pair_gpu_equals_oracle(oracle):
    from rupphash_amd import Engine, hamminghash, pdqhash

    eng = Engine(0)
    imgs = load()
    got = [pdqhash.generate_pdq(im, eng) for im in imgs]
    for im, g in zip(imgs, got):
        rc, coeffs, q = oracle.pdq_features(im)
        assert g is not None and np.array_equal(g[0], oracle.to_hash(coeffs)) and g[1] == np.float32(q)
    d = hamminghash.hamming_distance(got[0][0], got[1][0])
    print(f"config 1 (GPU): Hamming distance {d}")
    groups, _ = eng.group_files_pdq(np.stack([got[0][0], got[1][0]]), 40)
    assert (groups == [[0, 1]]) == (d <= 40)
    eng.close()
