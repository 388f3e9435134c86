"""Fixtures for the > 512 px pre-downsample (pdqhash.rs:181-220 -> fast_image_resize 6.1.0, Convolution(Box) on U8) that do NOT
come from the oracle: results derived by hand from the published algorithm (Pillow-style two-pass fixed-point convolution,
horizontal pass into a u8 intermediate, then vertical).  PARITY with the Rust binary stays UNPINNED (the crate's source is not in
the reference tree); what these tests pin is that oracle AND product implement exactly the hand-derived arithmetic:

  exact 2:1   taps (1/2, 1/2), precision 15, coefficient 16384:  out = (2^14 + 16384 (a + b)) >> 15 = floor((a + b) / 2 + 1/2)
  exact 4:1   taps 4 x 1/4,    precision 15, coefficient 8192:   out = floor((a + b + c + d) / 4 + 1/2)
  3:2         period 3 in -> 2 out: [mean(p0, p1) rounded half up, p2]   (window of output 2k+1: the tap at x = 3k+1 has box
              argument exactly -1/2, which the half-open box (-1/2, 1/2] excludes; precision 14, coefficients 8192/8192 and 16384)
each applied horizontally, rounded to u8, then vertically (double rounding).
"""
import numpy as np
import pytest


def rhu_mean(*xs):
    """mean of u8 arrays rounded half up, as the fixed-point kernel does: (2^(p-1) + c * sum) >> p with c = 2^p / n"""
    n = len(xs)
    s = sum(x.astype(np.int64) for x in xs)
    return ((2 * s + n) // (2 * n)).astype(np.uint8)


def hand_downscale(img, factor):
    """exact integer box downscale with the u8 intermediate: horizontal, then vertical"""
    h = rhu_mean(*[img[:, k::factor] for k in range(factor)])
    return rhu_mean(*[h[k::factor, :] for k in range(factor)])


def hand_three_to_two(img):
    def axis(a):  # along axis 1
        out = np.zeros((a.shape[0], a.shape[1] // 3 * 2), np.uint8)
        out[:, 0::2] = rhu_mean(a[:, 0::3], a[:, 1::3])
        out[:, 1::2] = a[:, 2::3]
        return out
    return axis(axis(img).T).T


def test_precision_and_coefficient_sums():
    import oracle

    assert oracle.resize_axis_info(1024, 512) == (15, 3, 32768, 32768)    # 2:1: taps 16384 + 16384
    assert oracle.resize_axis_info(2048, 512) == (15, 5, 32768, 32768)    # 4:1: 4 x 8192 (the precision loop ends at 15 without its break)
    assert oracle.resize_axis_info(768, 512) == (14, 3, 16384, 16384)     # 3:2: max weight 1.0 -> round(2^(p+1)) >= 2^15 at p = 14
    for in_size, out_size in [(780, 512), (1280, 512), (854, 341), (4000, 512), (513, 512), (1023, 512)]:
        p, window, lo, hi = oracle.resize_axis_info(in_size, out_size)
        scale = in_size / out_size
        assert window == int(np.ceil(0.5 * scale)) * 2 + 1
        # every output's taps sum to 2^p up to the rounding of each tap (<= window / 2)
        assert abs(lo - (1 << p)) <= window / 2 + 1 and abs(hi - (1 << p)) <= window / 2 + 1
        assert 8 <= p <= 15


@pytest.mark.parametrize("size,factor", [(64, 2), (64, 4), (1024, 2)])
def test_oracle_integer_downscale_equals_hand_block_means(size, factor):
    import oracle

    rng = np.random.default_rng(size + factor)
    img = rng.integers(0, 256, (size, size + factor * 4), dtype=np.uint8)
    out = oracle.resize_box_u8(img, img.shape[1] // factor, size // factor)
    assert np.array_equal(out, hand_downscale(img, factor))
    # half-up ties: (1, 2) -> 2, double rounding: [[0, 1], [1, 1]] -> rows round to (1, 1) -> 1 although the true mean is 0.75
    assert hand_downscale(np.array([[0, 1], [1, 1]], np.uint8), 2)[0, 0] == 1
    assert oracle.resize_box_u8(np.tile(np.array([[0, 1], [1, 1]], np.uint8), (8, 8)), 8, 8)[0, 0] == 1


def test_oracle_three_to_two_equals_hand_pattern():
    import oracle

    rng = np.random.default_rng(32)
    img = rng.integers(0, 256, (48, 96), dtype=np.uint8)
    assert np.array_equal(oracle.resize_box_u8(img, 64, 32), hand_three_to_two(img))


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,kind", [(1024, 1024, "2:1"), (2048, 2048, "4:1"), (768, 768, "3:2"), (1024, 512, "2:1, 512 x 256"), (768, 384, "3:2, 512 x 256"),
                                      (640, 2048, "4:1, 160 x 512")])
def test_product_predownsample_equals_hand_derived_thumbnail(w, h, kind):
    """Through the C ABI: hashing the large Luma8 image must give exactly the result of hashing the hand-derived thumbnail
    (coefficients and quality bit for bit), i.e. the product's resize is the hand-derived arithmetic."""
    from rupphash_amd import Engine

    def axis_down(a, in_size, out_size):  # along axis 1
        if in_size == 2 * out_size:
            return rhu_mean(a[:, 0::2], a[:, 1::2])
        if in_size == 4 * out_size:
            return rhu_mean(a[:, 0::4], a[:, 1::4], a[:, 2::4], a[:, 3::4])
        assert 2 * in_size == 3 * out_size
        out = np.zeros((a.shape[0], out_size), np.uint8)
        out[:, 0::2] = rhu_mean(a[:, 0::3], a[:, 1::3])
        out[:, 1::2] = a[:, 2::3]
        return out

    eng = Engine(0)
    try:
        rng = np.random.default_rng(w * 3 + h)
        base = rng.integers(0, 256, (h // 64, w // 64), dtype=np.uint8).repeat(64, axis=0).repeat(64, axis=1)
        img = np.clip(base.astype(np.int16) + rng.integers(-20, 21, (h, w)), 0, 255).astype(np.uint8)
        # calculate_target_dimensions (pdqhash.rs:224-235): the long side becomes 512, the other keeps the aspect
        tw, th = (512, max(h * 512 // w, 1)) if w > h else (max(w * 512 // h, 1), 512)
        hor = axis_down(img, w, tw)                                        # horizontal pass into the u8 intermediate
        thumb = axis_down(np.ascontiguousarray(hor.T), h, th).T            # then vertical
        assert thumb.shape == (th, tw)
        big = eng.pdq_hash_batch(img[None], want_quality=True, want_coeffs=True)
        small = eng.pdq_hash_batch(np.ascontiguousarray(thumb)[None], want_quality=True, want_coeffs=True)
        assert big["valid"][0] and small["valid"][0]
        assert np.array_equal(big["coeffs"].view(np.uint32), small["coeffs"].view(np.uint32)), kind
        assert np.array_equal(big["hash"], small["hash"]) and np.array_equal(big["quality"].view(np.uint32), small["quality"].view(np.uint32))
    finally:
        eng.close()
