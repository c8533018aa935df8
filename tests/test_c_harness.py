"""The C harness of the drop-in boundary (tools/c_harness/harness.c): a plain C caller of include/capgpu.h.
CPU: it compiles against the header, links against the built library and - without a GPU - fails loudly with the
"no CPU fallback" message and exit code 2.  GPU: it proves the golden log-5 instance and its 769 proof bytes equal the
oracle's (tests/golden/proof_log5.json), after an NTT round trip, verifier accept / reject and a batch-vs-single check."""
import os
import shutil
import subprocess

import pytest

from oracle import bn254 as bn
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    cc = shutil.which("gcc") or shutil.which("cc")
    if not cc:
        pytest.skip("no C compiler")
    exe = str(tmp_path_factory.mktemp("harness") / "harness")
    lib_dir = os.path.join(ROOT, "cap_amd")
    subprocess.check_call([cc, "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tools", "c_harness", "harness.c"), "-L", lib_dir, "-lcapgpu",
                           "-Wl,-rpath," + lib_dir, "-o", exe])
    return exe


def expected_proof_bytes():
    g = H.load_golden("proof_log5.json")
    pts = [H.unhex_pt(p) for p in g["wires_poly_comms"]] + [H.unhex_pt(g["prod_perm_poly_comm"])] + \
        [H.unhex_pt(p) for p in g["split_quot_poly_comms"]] + [H.unhex_pt(g["opening_proof"]),
                                                              H.unhex_pt(g["shifted_opening_proof"])]
    ev = [int(x, 16) for x in g["wires_evals"] + g["wire_sigma_evals"] + [g["perm_next_eval"]]]
    out = (5).to_bytes(8, "little") + b"".join(bn.g1_serialize_compressed(p) for p in pts[0:5])
    out += bn.g1_serialize_compressed(pts[5])
    out += (5).to_bytes(8, "little") + b"".join(bn.g1_serialize_compressed(p) for p in pts[6:11])
    out += bn.g1_serialize_compressed(pts[11]) + bn.g1_serialize_compressed(pts[12])
    out += (5).to_bytes(8, "little") + b"".join(bn.fr_to_bytes_le(v) for v in ev[0:5])
    out += (4).to_bytes(8, "little") + b"".join(bn.fr_to_bytes_le(v) for v in ev[5:9])
    return out + bn.fr_to_bytes_le(ev[9]) + b"\x00"


def test_harness_builds_and_fails_loudly_without_a_gpu(harness):
    if H.gpu_present():
        pytest.skip("a GPU is present: covered by the gpu test")
    r = subprocess.run([harness, os.path.join(H.GOLDEN, "harness_log5.bin")], capture_output=True, text=True)
    assert r.returncode == 2 and "no CPU fallback" in r.stderr


def test_harness_input_matches_its_generator():
    """the committed binary fixture is what tests/golden/make_harness_input.py writes today"""
    import hashlib
    before = hashlib.sha256(open(os.path.join(H.GOLDEN, "harness_log5.bin"), "rb").read()).hexdigest()
    subprocess.check_call(["python3", os.path.join(H.GOLDEN, "make_harness_input.py")], stdout=subprocess.DEVNULL)
    after = hashlib.sha256(open(os.path.join(H.GOLDEN, "harness_log5.bin"), "rb").read()).hexdigest()
    assert before == after


@pytest.mark.gpu
def test_harness_proves_the_golden_instance(harness):
    r = subprocess.run([harness, os.path.join(H.GOLDEN, "harness_log5.bin")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-800:]
    lines = r.stdout.split("\n")
    assert "OK" in lines and "COEFFS same bytes" in lines   # both input forms of the ABI give the golden bytes
    proof = [ln for ln in lines if ln.startswith("PROOF ")][0].split()[1]
    assert bytes.fromhex(proof) == expected_proof_bytes()
