"""CPU-side checks of the C-ABI library: it loads, exports exactly what include/rupphash.h
declares, its host-scalar functions match the oracle, and it refuses to run without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "rupphash.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rph_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound():
    from rupphash_amd import _lib

    names = declared_functions()
    assert len(names) >= 40
    L = _lib.load()
    for n in names:
        assert hasattr(L, n), f"librupphash_hip.so does not export {n}"
    assert sorted(_lib.SIGNATURES) == names, set(_lib.SIGNATURES) ^ set(names)
    assert L.rph_abi_version() == 1


def test_no_gpu_means_loud_failure_not_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from rupphash_amd import Engine, RphError

    with pytest.raises(RphError) as e:
        Engine(0)
    assert e.value.status == -2 and "no CPU fallback" in str(e.value)


def test_product_does_not_import_oracle():
    """The product path may not import, link or call anything under oracle/."""
    pkg = os.path.join(ROOT, "rupphash_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in text and "oracle/" not in text, f
                assert not re.search(r"^\s*(import|from)\s+oracle\b", text, flags=re.M), f


def test_phash_bitops_match_oracle_and_known_answer(oracle):
    from rupphash_amd import phash

    h = 0xDEB1E20C136F983C
    assert phash.calculate_rotation_invariant_hash(h) == 0x8B1BB7A646C5CD96  # NOTES.txt:64-67
    rng = np.random.default_rng(1)
    for v in [0, 2**64 - 1, h] + [int(x) for x in rng.integers(0, 2**64, 200, dtype=np.uint64)]:
        assert phash.rotate_hash_90(v) == oracle.rotate_hash_90(v)
        assert phash.rotate_hash_180(v) == oracle.rotate_hash_180(v)
        assert phash.rotate_hash_270(v) == oracle.rotate_hash_270(v)
        assert phash.flip_hash_horizontal(v) == oracle.flip_hash_horizontal(v)
        assert phash.calculate_rotation_invariant_hash(v) == oracle.rotation_invariant_hash(v)
        assert phash.generate_dihedral_hashes(v) == oracle.phash_dihedral(v)


def test_scalar_hamming_helpers_match_oracle(oracle):
    from rupphash_amd import hamminghash as hh
    from rupphash_amd import pdqhash, scanner

    rng = np.random.default_rng(2)
    a = rng.integers(0, 256, (50, 32), dtype=np.uint8)
    b = rng.integers(0, 256, (50, 32), dtype=np.uint8)
    for x, y in zip(a, b):
        assert hh.hamming_distance(x, y) == oracle.hamming256(x, y)
        for k in range(16):
            assert hh.HammingHash256.get_chunk(x, k) == int(x[2 * k]) | int(x[2 * k + 1]) << 8
    assert hh.hamming_distance(0, 0xFFF) == 12 and hh.HammingHash64.get_chunk(0x0102030405060708, 7) == 1
    assert hh.MAX_SIMILARITY_64 == 15 and hh.MAX_SIMILARITY_256 == 63
    for w, h in [(4000, 5), (5, 4000), (1024, 1024), (1024, 512), (780, 768), (1280, 854), (0, 7)]:
        assert pdqhash.calculate_target_dimensions(w, h) == oracle.target_dimensions(w, h)
    assert [scanner.is_low_pdq_quality(q) for q in (None, 0, 49, 50, 100)] == [False, True, True, False, False]
    s = hh.SparseBitSet(1000)
    assert (s.set(5), s.set(5), s.set(999)) == (False, True, False)
    s.clear()
    assert s.set(5) is False


def _flags_for(a, b, tol):
    x = np.bitwise_xor(a, b)
    for k in range(16):
        c16 = int(x[2 * k]) | int(x[2 * k + 1]) << 8
        pc = bin(c16).count("1")
        if pc <= tol:
            slot = 0 if pc == 0 else 1 + (c16 & -c16).bit_length() - 1
            return 0x8000 | (k << 5) | slot
    return 0


@pytest.mark.parametrize("thr", [8, 20, 40])
def test_host_grouping_from_edges_matches_oracle(oracle, thr):
    """rph_find_groups_from_edges / rph_union_find_groups are host C++ (as in the reference); feed them
    the edge list a sweep would produce and compare with the oracle's find_groups / group_pdq."""
    from rupphash_amd import EDGE_DTYPE, Engine

    rng = np.random.default_rng(thr)
    n = 1200
    hashes = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    for c in range(40):
        base = rng.integers(0, 256, 32, dtype=np.uint8)
        for j in rng.choice(n, 5, replace=False):
            v = base.copy()
            for bit in rng.choice(256, rng.integers(0, 30), replace=False):
                v[bit // 8] ^= 1 << (bit % 8)
            hashes[j] = v
    brute = oracle.all_pairs256(hashes, thr)
    edges = np.zeros(len(brute), EDGE_DTYPE)
    tol = 1 if thr // 16 >= 1 else 0
    for t, (i, j, d) in enumerate(brute[rng.permutation(len(brute))]):
        edges[t] = (i, j, d, _flags_for(hashes[i], hashes[j], tol))
    eng = Engine.__new__(Engine)  # host-only entry points need no context
    from rupphash_amd import _lib

    eng.L = _lib.load()
    eng.ctx = None
    assert eng.find_groups_from_edges(edges, n) == oracle.find_groups(oracle.KIND_PDQ, hashes, thr)
    assert eng.union_find_groups(edges, n) == oracle.group_pdq(hashes, thr)[1]


def _build_c_consumer():
    import subprocess
    out = os.path.join(ROOT, "tests", "c", "abi_smoke")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "abi_smoke.c"), "-o", out, "-L", os.path.join(ROOT, "rupphash_amd"),
                           "-lrupphash_hip", "-Wl,-rpath," + os.path.join(ROOT, "rupphash_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    return out


def test_header_is_plain_c_and_links():
    """include/rupphash.h compiles as strict C99 and a C program links against the library; without a GPU rph_init fails loudly"""
    import subprocess
    exe = _build_c_consumer()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    import torch
    if not torch.cuda.is_available():
        assert r.returncode == 3, (r.returncode, r.stdout, r.stderr)
        assert "no device" in r.stdout


@pytest.mark.gpu
def test_c_consumer_runs_on_gpu():
    import subprocess
    exe = _build_c_consumer()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "distance between the two images" in r.stdout
    assert "multi (1 device)" in r.stdout  # rph_multi at n_devices = 1: the same groups as rph_group_files_pdq


def test_rust_ffi_is_generated_from_the_header_and_matches_it():
    """rust/rph_ffi.rs (the binding the reference's Rust modules link against; uncompiled here: no Rust toolchain) is the transliteration of
    include/rupphash.h: regenerating it gives the committed text, every declared function is in it once with the header's arity, and the
    pointer / integer types map as the C ABI says."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("gen_rust_ffi", os.path.join(ROOT, "tools", "gen_rust_ffi.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    header = open(os.path.join(ROOT, "include", "rupphash.h")).read()
    committed = open(os.path.join(ROOT, "rust", "rph_ffi.rs")).read()
    assert gen.generate(header) == committed, "rust/rph_ffi.rs is stale: run python tools/gen_rust_ffi.py"
    _, _, funcs = gen.parse_header(header)
    assert sorted(f[0] for f in funcs) == declared_functions()
    code = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    for name, params, ret in funcs:
        m = re.search(r"pub fn %s\((.*?)\)( -> ([^;]+))?;" % name, committed)
        assert m, name
        rust_params = [p for p in m.group(1).split(", ") if p]
        c_decl = re.search(r"\b%s\s*\((.*?)\)\s*;" % name, code, flags=re.S).group(1)
        c_params = [] if c_decl.strip() in ("", "void") else c_decl.split(",")
        assert len(rust_params) == len(c_params) == len(params), name
        for (pname, rtype), cp in zip(params, c_params):
            assert cp.count("*") + cp.count("[") == rtype.count("*"), (name, cp, rtype)   # pointer depth (an array parameter is a pointer)
            assert ("const" in cp.split("*")[0]) == rtype.startswith("*const") or "*" not in cp, (name, cp, rtype)
        assert (ret is None) == (m.group(3) is None)
    # the hand-written modules only call what the binding declares
    for mod in ("pdqhash.rs", "hamminghash.rs", "phash.rs"):
        text = open(os.path.join(ROOT, "rust", mod)).read()
        for called in set(re.findall(r"ffi::(rph_[a-z0-9_]+)", text)):
            assert re.search(r"pub fn %s\(" % called, committed), (mod, called)
        assert "/* unchanged" not in text and "/* ..." not in text  # complete files: no elided bodies
