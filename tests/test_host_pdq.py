"""Host-scalar half of the C ABI (no GPU needed): PdqFeatures::to_hash / generate_dihedral_hashes for one vector
(pdqhash.rs:59-87) against the oracle, and the cache record codecs against byte strings written out by hand from
/root/reference/src/db.rs:1200-1231 (hash_db = [2 || 32 B], coeff_db = [2 || postcard(Vec<f32>)])."""
import struct

import numpy as np
import pytest

from rupphash_amd import db, pdqhash


def lcg_features(seed):
    """the generator of the reference's own tests (pdqhash.rs:521-533)"""
    s = seed & 0xFFFFFFFFFFFFFFFF
    out = np.zeros(256, np.float32)
    for i in range(256):
        s = (s * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        out[i] = np.float32(np.int32(np.uint32(s >> 33) % np.uint32(20001)) - 10000) / np.float32(100.0)
    return out


@pytest.mark.parametrize("seed", [1, 42, 0x12345678, 0xDEADBEEF, 7])  # pdqhash.rs:549, :563
def test_host_to_hash_and_dihedral_match_oracle_on_reference_seeds(oracle, seed):
    c = lcg_features(seed)
    f = pdqhash.PdqFeatures(c)
    assert np.array_equal(f.to_hash(), oracle.to_hash(c))
    d = f.generate_dihedral_hashes()
    assert np.array_equal(d, oracle.dihedral_hashes(c))
    assert np.array_equal(d, oracle.naive_dihedral(c))  # fast == naive (pdqhash.rs:548-558)
    assert np.array_equal(d[0], f.to_hash())
    assert len({bytes(x) for x in d}) == 8  # the full dihedral group (pdqhash.rs:561-570)


def test_host_to_hash_ties_signed_zeros_and_specials(oracle):
    rng = np.random.default_rng(3)
    cases = []
    for t in range(300):
        c = rng.normal(0, 20, 256).astype(np.float32)
        if t % 3 == 0:
            c[rng.integers(0, 256, 60)] = c[0]            # many ties around the median
        if t % 5 == 0:
            c[::7] = 0.0
            c[1::7] = -0.0                                # total_cmp orders -0 < +0; `>` does not
        if t % 11 == 0:
            c[:] = 1.5                                     # all equal: no bit set
        if t % 13 == 0:
            c[rng.integers(0, 256, 8)] = np.float32("inf")
            c[rng.integers(0, 256, 8)] = -np.float32("inf")
        cases.append(c)
    cases.append(np.zeros(256, np.float32))
    cases.append(np.arange(256, dtype=np.float32))
    cases.append(-np.arange(256, dtype=np.float32))
    for c in cases:
        f = pdqhash.PdqFeatures(c)
        assert np.array_equal(f.to_hash(), oracle.to_hash(c))
        assert np.array_equal(f.generate_dihedral_hashes(), oracle.dihedral_hashes(c))


def test_median_is_the_128th_smallest():
    # pdqhash.rs:116-124: mid = (256 - 1) / 2 = 127, compare with `>`: for 0..255 exactly the values 128..255 set their bit
    c = np.arange(256, dtype=np.float32)
    h = pdqhash.PdqFeatures(c).to_hash()
    bits = np.unpackbits(h).sum()
    assert bits == 128
    # row r = coefficients 16r..16r+15 -> bytes 31-2r (low), 30-2r (high) (pdqhash.rs:155-162): rows 0..7 are all zero
    assert not h[16:].any() and (h[:16] == 0xFF).all()


# ---------------------------------------------------------------- cache records (db.rs)
def test_hash_record_layout_and_match_arms():
    h = bytes(range(100, 132))
    rec = db.encode_pdqhash(np.frombuffer(h, np.uint8))
    assert rec == b"\x02" + h                                         # db.rs:1203-1206: push(PDQ_ALGO_VERSION); extend(pdqhash)
    assert bytes(db.decode_pdqhash(rec)) == h
    assert db.decode_pdqhash(b"\x01" + h) is None                     # another pipeline's version: a miss (db.rs:683-696)
    assert db.decode_pdqhash(b"\x02" + h[:31]) is None                # rest.len() != 32
    assert db.decode_pdqhash(b"\x02" + h + b"\x00") is None
    assert db.decode_pdqhash(b"") is None                             # split_first() on an empty value
    many = np.arange(5 * 32, dtype=np.uint8).reshape(5, 32)
    recs = db.encode_pdqhashes(many)
    assert recs.shape == (5, 33) and (recs[:, 0] == 2).all() and np.array_equal(recs[:, 1:], many)
    recs[3, 0] = 1
    back, present = db.decode_pdqhashes(recs)
    assert present.tolist() == [True, True, True, False, True]
    assert np.array_equal(back[[0, 1, 2, 4]], many[[0, 1, 2, 4]]) and not back[3].any()


def test_coeff_record_is_version_byte_plus_postcard_vec_f32():
    c = np.array([1.0, -2.5, 0.0, -0.0] + [float(i) * 0.25 for i in range(252)], np.float32)
    rec = db.encode_coefficients(c)
    # postcard: struct = its fields in order; Vec<f32> = varint(len) then each f32 as 4 little-endian bytes; 256 = 0x80 0x02
    want = b"\x02" + b"\x80\x02" + b"".join(struct.pack("<f", float(x)) for x in c)
    assert rec == want and len(rec) == db.COEFF_RECORD_BYTES
    back = db.decode_coefficients(rec)
    assert np.array_equal(back.view(np.uint32), c.view(np.uint32))    # -0.0 and every bit pattern survive
    # hand-written small records
    assert db.encode_coefficients(np.array([1.0], np.float32)) == bytes([2, 1, 0x00, 0x00, 0x80, 0x3F])
    assert db.encode_coefficients(np.zeros(0, np.float32)) == bytes([2, 0])
    assert len(db.decode_coefficients(bytes([2, 0]))) == 0
    assert db.decode_coefficients(bytes([2, 2, 0, 0, 0x80, 0x3F, 0, 0, 0, 0xC0])).tolist() == [1.0, -2.0]
    n300 = db.encode_coefficients(np.zeros(300, np.float32))
    assert n300[:3] == bytes([2, 0xAC, 0x02]) and len(n300) == 3 + 1200  # 300 = 0b10_0101100 -> 0xAC 0x02
    assert len(db.decode_coefficients(n300)) == 300                    # the scanner drops it: len != 256 (scanner.rs:1265-1267)


def test_coeff_record_match_arms():
    c = np.linspace(-50, 50, 256).astype(np.float32)
    rec = db.encode_coefficients(c)
    assert db.decode_coefficients(b"\x01" + rec[1:]) is None          # older algorithm version: absent, not corrupt (db.rs:752-753)
    assert db.decode_coefficients(b"") is None
    with pytest.raises(db.Corrupted):                                  # CachedCoefficients::from_bytes fails -> lmdb::Error::Corrupted
        db.decode_coefficients(rec[:-1])
    with pytest.raises(db.Corrupted):
        db.decode_coefficients(b"\x02")                                # no length at all
    with pytest.raises(db.Corrupted):
        db.decode_coefficients(b"\x02\x80")                            # truncated varint
    with pytest.raises(db.Corrupted):
        db.decode_coefficients(b"\x02" + b"\xff" * 10 + b"\x01")       # varint overflowing 64 bits
    # postcard::from_bytes ignores bytes after the value
    assert np.array_equal(db.decode_coefficients(rec + b"tail"), c)
    # a non-minimal varint still decodes (postcard does not insist on the canonical form)
    assert db.decode_coefficients(bytes([2, 0x81, 0x00, 0, 0, 0x80, 0x3F])).tolist() == [1.0]
