"""GPU parity of the streaming single-pass PDQ kernel (csrc/pdq_stream.hip; generate_pdq_from_luma, pdqhash.rs:238-262, for Luma8 images of
128..512 x 128..512): coefficients, quality, hash and dihedral hashes bit for bit equal to the CPU oracle, and to the multi-pass kernels."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from rupphash_amd import Engine

    e = Engine(0)
    yield e
    e.close()


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def contents(rng, n, h, w):
    """noise, gradients, flat black / white, sparse points (tiny rounding residues run along whole lines), a checkerboard, blocks"""
    imgs = rng.integers(0, 256, (n, h, w), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    imgs[1] = ((xx * 200) // max(w - 1, 1) + (yy * 55) // max(h - 1, 1)).astype(np.uint8)
    imgs[2] = 0
    imgs[3] = 255
    imgs[4] = 0
    pts = rng.integers(0, h * w, 40)
    imgs[4].reshape(-1)[pts] = rng.integers(1, 256, 40, dtype=np.uint8)
    imgs[5] = (((xx + yy) & 1) * 255).astype(np.uint8)
    blocks = rng.integers(0, 256, ((h + 31) // 32, (w + 31) // 32), dtype=np.uint8)
    imgs[6] = np.kron(blocks, np.ones((32, 32), np.uint8))[:h, :w]
    imgs[7] = np.where(xx < w // 3, 250, 3).astype(np.uint8)
    return imgs


def hash_padded(eng, imgs, pad_value=0xA5):
    """Luma8 images through the C ABI with rows padded to whole dwords (what the streaming kernel takes: a packed row of odd width would go to
    the multi-pass kernels); the pad bytes are not zero"""
    from rupphash_amd._lib import check as rc_check

    n, h, w = imgs.shape
    pitch = (w + 3) & ~3
    buf = np.full((n, h, pitch), pad_value, np.uint8)
    buf[:, :, :w] = imgs
    out = {"hash": np.zeros((n, 32), np.uint8), "quality": np.zeros(n, np.float32), "coeffs": np.zeros((n, 256), np.float32),
           "dihedral": np.zeros((n, 8, 32), np.uint8), "valid": np.zeros(n, np.uint8)}
    rc_check(eng.L.rph_pdq_hash_batch(eng.ctx, buf.ctypes.data, n, w, h, 1, pitch, pitch * h, out["hash"].ctypes.data, out["quality"].ctypes.data,
                                      out["coeffs"].ctypes.data, out["dihedral"].ctypes.data, out["valid"].ctypes.data), "rph_pdq_hash_batch")
    return out


def check(eng, oracle, imgs, which=6):  # 6: the streaming kernel at every batch size (automatic mode keeps small calls on the multi-pass kernels)
    eng.set_pdq_kernel(which)
    out = hash_padded(eng, imgs) if imgs.ndim == 3 else eng.pdq_hash_batch(imgs, want_quality=True, want_coeffs=True, want_dihedral=True)
    eng.set_pdq_kernel(4)
    for k in range(len(imgs)):
        rc, coeffs, q = oracle.pdq_features(imgs[k])
        assert rc == 0 and out["valid"][k] == 1
        assert np.array_equal(bits(out["coeffs"][k]), bits(coeffs)), f"coefficients differ for image {k} of {imgs.shape}"
        assert bits(out["quality"][k:k + 1])[0] == bits(np.float32(q))[()], f"quality differs for image {k}"
        assert np.array_equal(out["hash"][k], oracle.to_hash(coeffs))
        assert np.array_equal(out["dihedral"][k], oracle.dihedral_hashes(coeffs))
    return out


# every window 2..8 on both axes, both parities of the strip and band remainders, the frame cases (w or h = 128, 512)
STREAM_GEOMS = [(512, 344), (344, 512), (512, 288), (128, 128), (129, 131), (200, 500), (511, 509), (512, 512), (448, 130), (320, 240), (256, 256), (384, 512),
                (512, 504), (191, 190), (255, 257), (130, 512), (192, 193), (193, 192), (257, 320), (321, 384), (385, 449), (449, 385), (512, 128), (500, 452),
                (452, 500), (512, 341), (341, 512), (512, 384), (400, 300)]


@pytest.mark.parametrize("w,h", STREAM_GEOMS)
def test_stream_kernel_matches_oracle(eng, oracle, w, h):
    rng = np.random.default_rng(w * 1009 + h)
    imgs = contents(rng, 8, h, w)
    out = check(eng, oracle, imgs)
    # the multi-pass kernels (rows through LDS tiles) say the same
    eng.set_pdq_kernel(5)
    ref = eng.pdq_hash_batch(imgs, want_quality=True, want_coeffs=True, want_dihedral=True)
    eng.set_pdq_kernel(4)
    assert np.array_equal(bits(out["coeffs"]), bits(ref["coeffs"])) and np.array_equal(out["hash"], ref["hash"])


def test_stream_kernel_random_geometries(eng, oracle):
    rng = np.random.default_rng(20261005)
    for _ in range(60):
        w, h = int(rng.integers(128, 513)), int(rng.integers(128, 513))
        imgs = rng.integers(0, 256, (2, h, w), dtype=np.uint8)
        imgs[1] = (imgs[1] >> int(rng.integers(0, 8))).astype(np.uint8)
        check(eng, oracle, imgs)


def test_stream_kernel_strided_rows_and_many_images(eng, oracle):
    """row pitch > w (dword aligned) and an image stride with slack; a launch of more images than the chip holds waves"""
    from rupphash_amd._lib import check as rc_check

    rng = np.random.default_rng(5)
    w, h, n = 300, 200, 3000
    pitch, stride = 304, 304 * 200 + 64
    buf = rng.integers(0, 256, (n, stride), dtype=np.uint8)
    hashes = np.zeros((n, 32), np.uint8)
    quality = np.zeros(n, np.float32)
    coeffs = np.zeros((n, 256), np.float32)
    valid = np.zeros(n, np.uint8)
    rc_check(eng.L.rph_pdq_hash_batch(eng.ctx, buf.ctypes.data, n, w, h, 1, pitch, stride, hashes.ctypes.data, quality.ctypes.data, coeffs.ctypes.data, None,
                                      valid.ctypes.data), "rph_pdq_hash_batch")
    assert valid.all()
    for k in list(range(0, n, 97)) + [n - 1]:
        img = np.ascontiguousarray(buf[k, :pitch * h].reshape(h, pitch)[:, :w])
        rc, c, q = oracle.pdq_features(img)
        assert rc == 0 and np.array_equal(bits(coeffs[k]), bits(c)), k
        assert bits(quality[k:k + 1])[0] == bits(np.float32(q))[()]
        assert np.array_equal(hashes[k], oracle.to_hash(c))


@pytest.mark.parametrize("w,h,ch", [(512, 344, 3), (344, 512, 4), (129, 131, 3), (130, 258, 4), (511, 509, 3), (300, 200, 3), (512, 128, 4), (203, 307, 3)])
def test_stream_kernel_colour_inputs(eng, oracle, w, h, ch):
    """Rgb8 / Rgba8 of the same geometries: to_luma601 (pdqhash.rs:268-284) into a Luma8 plane, then the streaming kernel; widths that are
    not multiples of four end in a partial quad"""
    rng = np.random.default_rng(w * 31 + h + ch)
    imgs = rng.integers(0, 256, (4, h, w, ch), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    imgs[0, ..., 0] = ((xx * 255) // (w - 1)).astype(np.uint8)
    imgs[1, ..., 1] = ((yy * 255) // (h - 1)).astype(np.uint8)
    imgs[2] = 255
    out = check(eng, oracle, imgs)
    eng.set_pdq_kernel(5)
    ref = eng.pdq_hash_batch(imgs, want_quality=True, want_coeffs=True, want_dihedral=True)
    eng.set_pdq_kernel(4)
    assert np.array_equal(bits(out["coeffs"]), bits(ref["coeffs"])) and np.array_equal(out["hash"], ref["hash"])


@pytest.mark.parametrize("w,h,ch", [(129, 131, 1), (341, 512, 1), (511, 509, 1), (333, 200, 3), (203, 307, 3), (255, 257, 4)])
def test_stream_kernel_rows_of_any_alignment(eng, oracle, w, h, ch):
    """packed rows whose length is not a multiple of four (Luma8 of odd width; Rgb8 rows of 3 w bytes): realigned / converted into a plane
    with 16-byte rows by aligned dwords + v_alignbyte, then the streaming kernel; also from a base address that is not dword aligned"""
    from rupphash_amd._lib import check as rc_check

    rng = np.random.default_rng(w + 7 * h + ch)
    n = 5
    shape = (n, h, w) if ch == 1 else (n, h, w, ch)
    imgs = rng.integers(0, 256, shape, dtype=np.uint8)
    eng.set_pdq_kernel(6)
    out = eng.pdq_hash_batch(imgs, want_quality=True, want_coeffs=True, want_dihedral=True)
    # the same pixels one byte into a larger buffer: every row misaligned differently
    per = h * w * ch
    buf = np.zeros(n * per + 8, np.uint8)
    buf[1:1 + n * per] = imgs.reshape(-1)
    d_px, d_h = eng.dev_alloc(buf.nbytes), eng.dev_alloc(n * 32)
    try:
        eng.dev_upload(d_px, buf)
        eng.pdq_hash_batch_dev(d_px + 1, n, w, h, ch, d_h)
        eng.synchronize()
        shifted = np.zeros((n, 32), np.uint8)
        eng.dev_download(shifted, d_h)
    finally:
        eng.dev_free(d_px)
        eng.dev_free(d_h)
        eng.set_pdq_kernel(4)
    for k in range(n):
        rc, coeffs, q = oracle.pdq_features(imgs[k])
        assert rc == 0 and out["valid"][k] == 1
        assert np.array_equal(bits(out["coeffs"][k]), bits(coeffs)), k
        assert bits(out["quality"][k:k + 1])[0] == bits(np.float32(q))[()]
        assert np.array_equal(out["hash"][k], oracle.to_hash(coeffs)) and np.array_equal(shifted[k], oracle.to_hash(coeffs))
