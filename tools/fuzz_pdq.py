"""Differential fuzz of the PDQ kernels: the two fused geometries and the generic multi-pass kernel must agree bit for bit
(hash, quality, coefficients, dihedral hashes) on random 512x512 RGB images of many content classes; and the streaming single-pass
kernel (pdq_stream.hip) with the multi-pass kernels on Luma8 / Rgb8 crops of the same content at random geometries 128..512."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rupphash_amd.engine import Engine

eng = Engine(0)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
rng = np.random.default_rng(seed)
yy, xx = np.mgrid[0:512, 0:512]


def make(kind):
    if kind == 0:
        return rng.integers(0, 256, (512, 512, 3), dtype=np.uint8)
    if kind == 1:  # low-amplitude noise around a random level (ties in the median, tiny coefficients)
        return (int(rng.integers(0, 250)) + rng.integers(0, 4, (512, 512, 3))).astype(np.uint8)
    if kind == 2:  # blocks
        b = int(rng.choice([1, 2, 3, 5, 8, 16, 31, 32, 64, 100]))
        v = rng.integers(0, 256, (512 // b + 1, 512 // b + 1, 3), dtype=np.uint8)
        return np.ascontiguousarray(v[yy // b, xx // b])
    if kind == 3:  # flat or saturated
        return np.full((512, 512, 3), int(rng.choice([0, 1, 127, 254, 255])), np.uint8)
    if kind == 4:  # smooth gradients
        a, b = rng.uniform(-0.6, 0.6, 2)
        g = np.clip(128 + a * (xx - 256) + b * (yy - 256), 0, 255).astype(np.uint8)
        return np.stack([g, np.roll(g, 7, 0), 255 - g], axis=2)
    if kind == 5:  # isolated bright pixels near the frame on black
        img = np.zeros((512, 512, 3), np.uint8)
        for _ in range(int(rng.integers(1, 30))):
            img[int(rng.choice([0, 1, 2, 3, 4, 507, 508, 509, 510, 511, int(rng.integers(0, 512))])),
                int(rng.choice([0, 1, 2, 3, 4, 507, 508, 509, 510, 511, int(rng.integers(0, 512))]))] = rng.integers(0, 256, 3)
        return img
    img = rng.integers(0, 256, (512, 512, 3), dtype=np.uint8)  # 6: noise with saturated frame rows / columns
    img[:, :8] = 255
    img[-8:, :] = 0
    return img


t0 = time.time()
total = 0
while time.time() - t0 < budget:
    batch = np.stack([make(int(rng.integers(0, 7))) for _ in range(48)])
    outs = []
    for kern in (0, 1, 2, 3, 5):
        eng.set_pdq_kernel(kern)
        outs.append(eng.pdq_hash_batch(batch, want_quality=True, want_coeffs=True, want_dihedral=True))
    for other in outs[1:]:
        for k in ("hash", "valid", "dihedral"):
            assert np.array_equal(outs[0][k], other[k]), k
        assert np.array_equal(outs[0]["coeffs"].view(np.uint32), other["coeffs"].view(np.uint32))
        assert np.array_equal(outs[0]["quality"].view(np.uint32), other["quality"].view(np.uint32))
    total += len(batch)
    # the streaming kernel against the multi-pass kernels: crops at a random geometry, gray (packed rows of a width that is a multiple of
    # four are read in place; every other width is realigned into a plane with 16-byte rows first) and colour (always through the plane)
    w, h = int(rng.integers(128, 513)), int(rng.integers(128, 513))
    if rng.random() < 0.6:
        w &= ~3
    crops = np.ascontiguousarray(batch[:16, :h, :w, :])
    for imgs in (np.ascontiguousarray(crops[..., 1]), crops):
        res = []
        for kern in (0, 5, 6):
            eng.set_pdq_kernel(kern)
            res.append(eng.pdq_hash_batch(imgs, want_quality=True, want_coeffs=True, want_dihedral=True))
        for other in res[1:]:
            for k in ("hash", "valid", "dihedral"):
                assert np.array_equal(res[0][k], other[k]), (k, w, h, imgs.shape)
            assert np.array_equal(res[0]["coeffs"].view(np.uint32), other["coeffs"].view(np.uint32)), (w, h, imgs.shape)
            assert np.array_equal(res[0]["quality"].view(np.uint32), other["quality"].view(np.uint32)), (w, h, imgs.shape)
        total += len(imgs)
eng.set_pdq_kernel(4)
print(f"seed {seed}: {total} images, fused strip64 == fused strip128 == fused low-latency == generic (plain and tiled); streaming == multi-pass at random geometries, bit for bit")
