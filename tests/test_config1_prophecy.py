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


@pytest.mark.gpu
@pytest.mark.parametrize("flavour", [0, 1])
def test_prophecy_pair_from_the_file_bytes(oracle, flavour):
    """config 1 without a host decoder in the loop: the JPEG bytes go to rph_jpeg_pdq_hash_batch (row N3); decode arithmetic of
    zune-jpeg as recalled (0, parity unpinned) or of libjpeg-turbo (1, == the Pillow decode the tests above use)"""
    from rupphash_amd import Engine, hamminghash

    eng = Engine(0)
    files = [open(os.path.join(GOLDEN, f"Prophecy_Has_Been_Fulfilled_{k}.jpg"), "rb").read() for k in (1, 2)]
    out = eng.jpeg_pdq_hash_batch(files, flavour=flavour, threads=2)
    assert out["valid"].all()
    for k in range(2):
        rc, coeffs, q = oracle.pdq_features(oracle.jpeg_decode(files[k], flavour))
        assert rc == 0 and np.array_equal(out["hash"][k], oracle.to_hash(coeffs)) and out["quality"][k] == np.float32(q)
    d = hamminghash.hamming_distance(out["hash"][0], out["hash"][1])
    print(f"config 1 (GPU, from the file bytes, flavour {flavour}): Hamming distance {d}")
    if flavour == 1:
        pil = [oracle.to_hash(oracle.pdq_features(im)[1]) for im in load()]
        assert np.array_equal(out["hash"][0], pil[0]) and np.array_equal(out["hash"][1], pil[1])  # same pixels as Pillow, hence the same hashes
    assert d <= 63
    eng.close()
