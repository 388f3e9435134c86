"""Short runs of the differential fuzzers under tools/ (independent formulations of the same kernels must agree bit for bit)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("tool,seed", [("fuzz_hamming.py", 11), ("fuzz_grouping.py", 12), ("fuzz_pdq.py", 13), ("fuzz_jpeg.py", 14)])
def test_differential_fuzz(tool, seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), str(seed), "6"], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr[-2000:])
    assert r.returncode == 0, r.stdout + r.stderr[-2000:]
    assert "agree" in r.stdout or "bit for bit" in r.stdout


@pytest.mark.gpu
def test_jpeg_path_is_repeatable_under_load():
    """tools/soak_jpeg.py, short form: a mixed call of 3 000 files (segments, restart intervals, progressive files with their atomics) through the
    device walks several times, alternating thread counts and segment sizes: the same bytes every time, and the host decoder's"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak_jpeg.py"), "6"], capture_output=True, text=True, timeout=600)
    print(r.stdout, r.stderr[-2000:])
    assert r.returncode == 0 and "identical to the host decoder" in r.stdout, r.stdout + r.stderr[-2000:]
