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
