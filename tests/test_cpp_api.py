"""The C++ host mirror of the reference's proof API (include/capgpu_proof.hpp) and the reference's own proof tests
restated on it (tests/cpp/proof_api_test.cpp: test_transfer/mint/freeze_validity_proof, src/proof/transfer.rs:599-760).
CPU: the header compiles warning-free as C++17 and the program fails loudly (Err(FailedSnark), exit 2) without a GPU.
GPU: every assertion of the restated tests holds and the golden instance's proof bytes equal the oracle's."""
import os
import shutil
import subprocess

import pytest

from tests import helpers as H
from tests.test_c_harness import expected_proof_bytes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = [os.path.join(H.GOLDEN, "harness_log5.bin"), os.path.join(H.GOLDEN, "harness_b_log4.bin")]


@pytest.fixture(scope="module")
def api_test(tmp_path_factory):
    cxx = shutil.which("g++") or shutil.which("c++")
    if not cxx:
        pytest.skip("no C++ compiler")
    exe = str(tmp_path_factory.mktemp("cppapi") / "proof_api_test")
    lib_dir = os.path.join(ROOT, "cap_amd")
    subprocess.check_call([cxx, "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "proof_api_test.cpp"), "-L", lib_dir, "-lcapgpu",
                           "-Wl,-rpath," + lib_dir, "-o", exe])
    return exe


def test_cpp_api_builds_and_fails_loudly_without_a_gpu(api_test):
    if H.gpu_present():
        pytest.skip("a GPU is present: covered by the gpu test")
    r = subprocess.run([api_test] + ARGS, capture_output=True, text=True)
    assert r.returncode == 2
    assert "FailedSnark(Failed to generate universal SRS" in r.stderr and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_cpp_api_runs_the_reference_proof_tests(api_test):
    r = subprocess.run([api_test] + ARGS, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1500:]
    lines = r.stdout.split("\n")
    assert "OK" in lines
    proof = [ln for ln in lines if ln.startswith("PROOF ")][0].split()[1]
    assert bytes.fromhex(proof) == expected_proof_bytes()
