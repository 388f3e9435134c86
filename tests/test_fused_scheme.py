"""CPU model of the arithmetic scheme of the fused 512x512 PDQ kernel (rupphash_amd/csrc/pdq_fused512.hip),
checked bit for bit against the oracle's sequential Jarosz filter.

The fused kernel never materialises the four full-image box passes.  It relies on these facts, each of
which this file proves numerically on random and adversarial images:

 (1) pass 1 (rows then columns) is exact wherever both window sizes are powers of two: the value is
     (8x8 integer box sum) / 64, computable in any order (V = vertical 8-sum, Hs = horizontal 8-sum of V);
 (2) the six columns {0,1,2,508,509,510} divide by 5,6,7 in the row pass, are inexact, and their column
     pass must be the reference's sequential running sum ("E chains");
 (3) the six rows {0,1,2,508,509,510} divide an exact sum by 5,6,7 once (order free);
 (4) pass 2 rows is a sequential chain per row over all 512 x; pass 2 columns is a sequential chain per
     sampled column (x = 8j+4); only rows 8i+4 are emitted;
 (5) scaling every pass-2 quantity by 64 (so interior inputs are the integers Hs) commutes with f32
     rounding, so the kernel can skip the 1/64 until the very end;
 (6) q = fma(fma(-q0, d, N), 1/d, q0) with q0 = N * fl(1/d) equals the IEEE quotient N/d for the
     numerators and divisors that occur (Markstein), so the per-step division costs 3 ops.
"""
import numpy as np
import pytest

f32 = np.float32


def nwin(o, n=512):
    """window size of output o of the 4-phase box (win 8): 5,6,7,8 | 8 ... 8 | 7,6,5,4"""
    lo, hi = max(0, o - 3), min(n - 1, o + 4)
    return hi - lo + 1


def box_sum_clipped(a, axis):
    """integer sum over the clipped window [o-3, o+4] along axis"""
    n = a.shape[axis]
    c = np.concatenate([np.zeros_like(np.take(a, [0], axis)), np.cumsum(a, axis=axis)], axis=axis)
    lo = np.maximum(0, np.arange(n) - 3)
    hi = np.minimum(n - 1, np.arange(n) + 4)
    return np.take(c, hi + 1, axis) - np.take(c, lo, axis)


def chain_1d(inp, axis_len=512):
    """the reference's box_one_d_float (win 8) applied along axis 0 of inp[len, k] in f32, vectorised over k"""
    n = axis_len
    out = np.zeros_like(inp)
    s = np.zeros(inp.shape[1], f32)
    cw = f32(0)
    for t in range(4):
        s = s + inp[t]
        cw = f32(cw + 1)
    for t in range(4):
        s = s + inp[4 + t]
        cw = f32(cw + 1)
        out[t] = s / cw
    for t in range(n - 8):
        s = s + inp[8 + t]
        s = s - inp[t]
        out[4 + t] = s / cw
    for t in range(4):
        s = s - inp[n - 8 + t]
        cw = f32(cw - 1)
        out[n - 4 + t] = s / cw
    return out


def fused_model(luma):
    """returns the decimated 64x64 buffer exactly as the fused kernel computes it"""
    L = luma.astype(np.int64)
    V = box_sum_clipped(L, 0)             # vertical clipped 8-sums (exact integers)
    Hs = box_sum_clipped(V, 1)            # then horizontal: the clipped 2-D box sums
    ny = np.array([nwin(o) for o in range(512)])
    nx = ny
    # ---- pass-2 input, scaled by 64: in2s[y][x] = 64 * out1[y][x]
    in2s = np.zeros((512, 512), f32)
    exact_c = np.isin(nx, (8, 4))
    exact_r = np.isin(ny, (8, 4))
    cs = np.where(nx == 4, 2, 1)          # 64 / (8*nx) for power-of-two windows
    rs = np.where(ny == 4, 2, 1)
    N = (Hs * cs[None, :]).astype(f32)    # exact
    # exact rows: multiply; inexact rows: one IEEE division of the exact numerator
    for y in range(512):
        if exact_r[y]:
            in2s[y] = N[y] * f32(rs[y])
        else:
            in2s[y] = (N[y] * f32(8.0)) / f32(ny[y])
    # ---- E chains for the 6 inexact columns: rowval scaled by 64 = fl(64 R / nx), then the sequential column chain
    R = box_sum_clipped(L, 1)             # horizontal clipped sums of luma (exact)
    for x in (0, 1, 2, 508, 509, 510):
        rv = (R[:, x].astype(f32) * f32(64.0)) / f32(nx[x])
        in2s[:, x] = chain_1d(rv[:, None])[:, 0]
    assert exact_c[3] and exact_c[511] and not exact_c[2]
    # ---- pass 2 rows: sequential chain along x for all rows; keep the sampled columns
    tmp2s = chain_1d(np.ascontiguousarray(in2s.T)).T
    samp = np.ascontiguousarray(tmp2s[:, 4::8])          # [512][64], x = 8j+4
    # ---- pass 2 columns on the sampled columns; keep rows 8i+4; unscale
    out2s = chain_1d(samp)
    return out2s[4::8, :] * f32(1.0 / 64.0)


def make_images():
    rng = np.random.default_rng(2026)
    imgs = [rng.integers(0, 256, (512, 512), dtype=np.uint8)]
    yy, xx = np.mgrid[0:512, 0:512]
    imgs.append(((xx * 255) // 511).astype(np.uint8))                                   # horizontal ramp
    imgs.append((((yy // 32 + xx // 32) % 2) * 255).astype(np.uint8))                   # blocks
    imgs.append(np.full((512, 512), 255, np.uint8))                                     # saturated (largest sums)
    img = rng.integers(0, 4, (512, 512), dtype=np.uint8)                                # tiny values (finest lattice)
    img[:, :8] = 255
    img[:8, :] = 251
    imgs.append(img)
    img = np.zeros((512, 512), np.uint8)
    img[3, 2] = 255
    img[509, 510] = 254
    img[255, 255] = 1
    imgs.append(img)                                                                    # isolated pixels near the frame
    return imgs


@pytest.mark.parametrize("k", range(6))
def test_fused_scheme_equals_sequential_reference(oracle, k):
    luma = make_images()[k]
    rc, coeffs, q, b64 = oracle.pdq_from_luma(luma, want_buf64=True)
    assert rc == 0
    got = fused_model(luma)
    assert np.array_equal(got.view(np.uint32), b64.view(np.uint32))


def test_scheme_on_synthetic_bench_image(oracle):
    img = oracle.synth_images(999, 1)[0]
    luma = oracle.luma601(img)
    _, _, _, b64 = oracle.pdq_from_luma(luma, want_buf64=True)
    assert np.array_equal(fused_model(luma).view(np.uint32), b64.view(np.uint32))


def test_luma_float_formula_equals_integer_division():
    """kernel luma: trunc(fma(299,r, fma(587,g, fma(114,b,500))) * 0.001f) == (299r+587g+114b+500)/1000 for all sums"""
    num = np.arange(500, 255 * 1000 + 501, dtype=np.int64)       # every reachable numerator
    got = np.trunc(num.astype(f32) * f32(0.001)).astype(np.int64)
    assert np.array_equal(got, num // 1000)


def test_luma_fma_bias_then_rtz_f16_is_floor():
    """kernel luma (current form): every byte enters the dot products as the f16 number 1024 + v, so the accumulated numerator is
    n = 1 024 000 + (299 r + 587 g + 114 b) (exact, < 2^24) and y = fma(n, 0.001f, 0.5f) = 1024 + (299r+587g+114b+500)/1000 lies in
    [1024.5, 1280), where f16 numbers are 1 apart, so the round-toward-zero f32 -> f16 conversion (v_cvt_pkrtz_f16_f32) is
    floor((299r+587g+114b+500) / 1000) + 1024 for every reachable numerator."""
    x = np.arange(0, 255 * 1000 + 1, dtype=np.int64)               # 299 r + 587 g + 114 b
    n = x + 1024 * 1000
    assert n.max() < 2 ** 24
    c = np.float64(f32(0.001))
    y = (n.astype(np.float64) * c + 0.5).astype(f32)               # product and sum exact in f64 -> one rounding = fma
    assert np.array_equal(np.floor(y.astype(np.float64)).astype(np.int64), (x + 500) // 1000 + 1024)
    assert y.min() >= 1024 and y.max() < 1280                      # inside the binade where f16 spacing is exactly 1


def test_markstein_division_is_ieee_for_all_numerators():
    """(6): every numerator the kernel divides is an integer multiple of 8 below 2^22; divisors 5, 6, 7 (and 8, 4 exact)."""
    N = (np.arange(0, 16320 * 4 + 1, dtype=np.int64) * 8).astype(f32)
    assert np.array_equal(N.astype(np.int64), np.arange(0, 16320 * 4 + 1) * 8)
    for d in (5.0, 6.0, 7.0, 8.0, 4.0):
        d = f32(d)
        y = f32(1.0) / d
        q0 = N * y
        r = np.float32(np.float64(N) - np.float64(q0) * np.float64(d))  # fma(-q0, d, N): exact in f64, then one rounding
        assert np.array_equal(np.float64(r), np.float64(N) - np.float64(q0) * np.float64(d))  # residual is exactly representable
        q = np.float32(np.float64(q0) + np.float64(r) * np.float64(y))   # fma(r, y, q0): product exact in f64 (24x24 bits), one rounding
        assert np.array_equal(q.view(np.uint32), (N / d).view(np.uint32)), float(d)


def test_f16_holds_vertical_sums_exactly():
    """V <= 8*255 = 2040 < 2048: every vertical window sum is an exact f16 integer, and so are its f16 partial sums."""
    v = np.arange(0, 2041, dtype=np.int64)
    assert np.array_equal(v.astype(np.float16).astype(np.int64), v)
