"""Runs tests/cpp/reference_tests (the reference's unit tests through include/rupphash.hpp) on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "reference_tests")


def build_cpp_tests():
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "reference_tests.cpp"), "-o", BIN,
                           "-L", os.path.join(ROOT, "rupphash_amd"), "-lrupphash_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "rupphash_amd"), "-Wl,-rpath,/opt/rocm/lib", "-pthread"])


def test_cpp_mirror_compiles():
    build_cpp_tests()
    assert os.path.exists(BIN)


@pytest.mark.gpu
def test_reference_unit_tests_through_cpp_mirror():
    if not os.path.exists(BIN):
        build_cpp_tests()
    env = dict(os.environ, RPH_TEST_GOLDEN=os.path.join(ROOT, "tests", "golden"))
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=600, env=env)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
