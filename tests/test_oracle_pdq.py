"""Pins oracle/pdq_ref.c with the reference's own property tests (src/pdqhash.rs:464-648).

The reference holds no known-answer PDQ hash; these re-express every hot-path
test it has, with the same seeds and generators, against the C restatement.
"""
import numpy as np
import pytest


def lcg_features(seed):
    """pseudo_random_features, pdqhash.rs:537-545"""
    state = np.uint32(seed)
    out = np.zeros(256, np.float32)
    with np.errstate(over="ignore"):
        for i in range(256):
            state = np.uint32(state * np.uint32(1664525) + np.uint32(1013904223))
            out[i] = np.float32(np.float32(int(state) >> 8) / np.float32(65536.0)) - np.float32(128.0)
    return out


def lcg_buffer64(seed):
    """LCG 64x64 buffer, pdqhash.rs:606-614"""
    state = np.uint32(seed)
    buf = np.zeros((64, 64), np.float32)
    with np.errstate(over="ignore"):
        for r in range(64):
            for c in range(64):
                state = np.uint32(state * np.uint32(1664525) + np.uint32(1013904223))
                buf[r, c] = np.float32((int(state) >> 16) & 0xFF)
    return buf


def transform(buf, variant):
    """transform(), pdqhash.rs:587-604"""
    n = 64
    out = np.zeros_like(buf)
    for x in range(n):
        for y in range(n):
            out[x, y] = {
                0: buf[x, y],
                1: buf[n - 1 - y, x],
                2: buf[n - 1 - x, n - 1 - y],
                3: buf[y, n - 1 - x],
                4: buf[x, n - 1 - y],
                5: buf[n - 1 - x, y],
                6: buf[y, x],
                7: buf[n - 1 - y, n - 1 - x],
            }[variant]
    return out


@pytest.mark.parametrize("seed", [1, 42, 0x12345678, 0xDEADBEEF])
def test_fast_dihedral_matches_naive(oracle, seed):
    """pdqhash.rs:548-558"""
    f = lcg_features(seed)
    naive = oracle.naive_dihedral(f)
    assert np.array_equal(oracle.to_hash(f), naive[0])
    assert np.array_equal(oracle.dihedral_hashes(f), naive)


def test_dihedral_set_is_the_full_group(oracle):
    """pdqhash.rs:561-570"""
    h = oracle.dihedral_hashes(lcg_features(7))
    for i in range(8):
        for j in range(i + 1, 8):
            assert not np.array_equal(h[i], h[j]), (i, j)


@pytest.mark.parametrize("seed", [1, 42, 0xDEADBEEF])
def test_dihedral_hashes_match_physically_transformed_buffer(oracle, seed):
    """pdqhash.rs:583-628: independent ground truth for DCT + sign parity + slot order."""
    buf = lcg_buffer64(seed)
    predicted = oracle.dihedral_hashes(oracle.features_from_buffer64(buf))
    for variant in range(8):
        actual = oracle.to_hash(oracle.features_from_buffer64(transform(buf, variant)))
        dist = oracle.hamming256(actual, predicted[variant])
        assert dist == 0, f"variant {variant} (seed {seed}) is {dist} bits from the real transform"


def test_quality_metric_scaling(oracle):
    """pdqhash.rs:631-639"""
    flat = np.full((64, 64), 128.0, np.float32)
    assert oracle.quality(flat) == 0.0
    buf = np.array([[0.0, 10.0], [0.0, 10.0]], np.float32)
    assert abs(oracle.quality(buf) - 6.0 / 90.0) < 1e-6


def test_target_dimensions_never_collapse_to_zero(oracle):
    """pdqhash.rs:642-647"""
    assert oracle.target_dimensions(4000, 5) == (512, 1)
    assert oracle.target_dimensions(5, 4000) == (1, 512)
    assert oracle.target_dimensions(1024, 1024) == (512, 512)
    assert oracle.target_dimensions(1024, 512) == (512, 256)


def test_min_hashable_dim_and_resize_gate(oracle):
    """pdqhash.rs:17,167-169 (None below 5 px, checked once, BEFORE the resize) and :181 (resize gate)."""
    assert oracle.pdq_features(np.zeros((4, 64, 3), np.uint8))[0] == oracle.REF_TOO_SMALL
    assert oracle.pdq_features(np.zeros((64, 4, 3), np.uint8))[0] == oracle.REF_TOO_SMALL
    assert oracle.pdq_features(np.zeros((5, 5, 3), np.uint8))[0] == oracle.REF_OK
    assert oracle.pdq_features(np.zeros((513, 8, 3), np.uint8))[0] == oracle.REF_OK     # resized to 512 x 7
    assert oracle.pdq_features(np.zeros((5, 4000, 3), np.uint8))[0] == oracle.REF_OK    # 4000 x 5 -> 512 x 1, still hashed


def test_resize_restatement_properties(oracle):
    """resize_ref.c restates fast_image_resize 6.1.0 Convolution(Box)/U8 from its published algorithm (PARITY UNPINNED: the
    crate's source is absent).  What can be checked here: exact integer ratios are plain rounded box means, constants stay
    constant, and the result stays within one grey level of Pillow's BOX resample (same algorithm family; Pillow carries
    22-bit coefficients, the crate 16-bit ones at an adaptive precision, so single levels may differ)."""
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, (1024, 1536), dtype=np.uint8)
    r = oracle.resize_box_u8(a, 512, 512)                       # 3x horizontally, 2x vertically
    h = (a.reshape(1024, 512, 3).astype(np.int64).sum(axis=2) * 2 + 3) // 6          # round(mean of 3), half up
    v = (h.reshape(512, 2, 512).sum(axis=1) + 1) // 2                                  # round(mean of 2), half up
    assert np.abs(r.astype(int) - v).max() <= 1   # fixed-point coefficients: at most one level off the exact means
    assert (oracle.resize_box_u8(np.full((700, 900), 37, np.uint8), 512, 398) == 37).all()
    from PIL import Image

    # (geometries whose window edges never fall exactly on a pixel boundary: on exact ties, e.g. 517 -> 132 rows, the two
    #  implementations evaluate the box filter argument with differently ordered f64 expressions and may disagree on
    #  whether the edge pixel is inside -- one more reason this step is unpinned)
    for (w, h_) in [(780, 768), (1280, 854), (513, 600), (2000, 1111)]:
        nw, nh = oracle.target_dimensions(w, h_)
        img = rng.integers(0, 256, (h_, w), dtype=np.uint8)
        pil = np.asarray(Image.fromarray(img).resize((nw, nh), Image.BOX))
        assert np.abs(oracle.resize_box_u8(img, nw, nh).astype(int) - pil.astype(int)).max() <= 1, (w, h_)


def test_luma601_integer_formula(oracle):
    """pdqhash.rs:270-273; RGBA ignores alpha (:279)."""
    rng = np.random.default_rng(3)
    rgb = rng.integers(0, 256, (7, 9, 3), dtype=np.uint8)
    r32 = rgb.astype(np.uint32)
    want = ((299 * r32[..., 0] + 587 * r32[..., 1] + 114 * r32[..., 2] + 500) // 1000).astype(np.uint8)
    assert np.array_equal(oracle.luma601(rgb), want)
    rgba = np.concatenate([rgb, rng.integers(0, 256, (7, 9, 1), dtype=np.uint8)], axis=2)
    assert np.array_equal(oracle.luma601(rgba), want)
    # Luma8 input is borrowed as is (:173)
    rc, c1, q1 = oracle.pdq_features(want)
    rc2, c2, q2 = oracle.pdq_features(rgb)
    assert rc == rc2 == 0 and np.array_equal(c1.view(np.uint32), c2.view(np.uint32)) and q1 == q2


def box1d_closed_form_windows(n, win):
    """window [lo, hi] of output o for the 4-phase loop."""
    win = max(1, min(win, max(n, 1)))
    half = (win + 2) // 2
    return [(max(0, o - (win - half)), min(n - 1, o + half - 1)) for o in range(n)]


@pytest.mark.parametrize("n,win", [(512, 8), (64, 1), (100, 2), (37, 1), (5, 1), (300, 5), (511, 8)])
def test_box_filter_on_integers_matches_window_means(oracle, n, win):
    """pdqhash.rs:341-396 on exactly summable (integer) input: each output is
    fl(window sum / window size) with the closed-form window."""
    rng = np.random.default_rng(n * 31 + win)
    row = rng.integers(0, 256, (1, n)).astype(np.float32)
    got = oracle.jarosz(row, win, 1, nreps=1)[0]  # col pass with window 1 on a 1-row image is the identity
    wins = box1d_closed_form_windows(n, win)
    want = np.array([np.float32(row[0, lo:hi + 1].sum(dtype=np.float64)) / np.float32(hi - lo + 1) for lo, hi in wins],
                    np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_decimate_indices(oracle):
    """pdqhash.rs:428-443: rows/cols ((2i+1)*R)/128."""
    plane = np.arange(512 * 512, dtype=np.float32).reshape(512, 512)
    d = oracle.decimate(plane)
    idx = (np.arange(64) * 2 + 1) * 512 // 128
    assert np.array_equal(idx, np.arange(64) * 8 + 4)
    assert np.array_equal(d, plane[np.ix_(idx, idx)])
    plane = np.arange(100 * 37, dtype=np.float32).reshape(37, 100)
    d = oracle.decimate(plane)
    ri = (np.arange(64) * 2 + 1) * 37 // 128
    ci = (np.arange(64) * 2 + 1) * 100 // 128
    assert np.array_equal(d, plane[np.ix_(ri, ci)])


def test_dct_matrix_properties(oracle):
    """pdqhash.rs:287-304: D[i][j] = 0.125*sqrt(2)*cosf(((PI*freq)*(2j+1))/128), freq = i+1."""
    D = oracle.dct_matrix()
    assert D.shape == (16, 64)
    ref = (np.sqrt(2.0) / 8.0) * np.cos(np.pi * (np.arange(16)[:, None] + 1) * (2 * np.arange(64)[None, :] + 1) / 128.0)
    assert np.max(np.abs(D - ref)) < 2e-6  # f32 angle rounding (angles up to ~50 rad)
    # committed device table == what the oracle computes on this libm
    import rupphash_amd.dct_table as t

    assert np.array_equal(D.view(np.uint32).ravel(), np.asarray(t.DCT_TABLE_BITS, np.uint32))


def test_to_hash_packing_layout(oracle):
    """pdqhash.rs:155-162: row r -> hash[31-2r] (low byte), hash[30-2r] (high byte)."""
    coeffs = np.full(256, -1.0, np.float32)
    coeffs[3 * 16 + 9] = 5.0  # row 3, col 9 -> high byte bit 1 -> hash[30-6] = 0x02
    coeffs[0] = 5.0           # row 0, col 0 -> hash[31] bit 0
    h = oracle.to_hash(coeffs)
    want = np.zeros(32, np.uint8)
    want[24] = 0x02
    want[31] = 0x01
    assert np.array_equal(h, want)


def test_median_is_128th_smallest_and_total_order(oracle):
    """pdqhash.rs:116-124: index (256-1)/2 = 127 under total_cmp; compare is plain '>'."""
    c = np.arange(256, dtype=np.float32)  # median = 127 -> values 128..255 set
    bits = np.unpackbits(oracle.to_hash(c)).sum()
    assert bits == 128
    c2 = c.copy()
    c2[:128] = 0.0  # many ties at the median value 0.0: nothing below or equal is set
    assert np.unpackbits(oracle.to_hash(c2)).sum() == 128
    c3 = np.zeros(256, np.float32)
    c3[:100] = -0.0
    c3[200:] = 1.0  # median is +0.0 or -0.0; '>' ignores the sign of zero
    assert np.unpackbits(oracle.to_hash(c3)).sum() == 56
