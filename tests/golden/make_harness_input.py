"""Writes tests/golden/harness_log5.bin: the inputs of the golden log-5 proof (tests/golden/proof_log5.json) in the
flat binary layout tools/c_harness/harness.c reads.  Run here:  python tests/golden/make_harness_input.py"""
import json
import os
import struct
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from cap_amd import bench_utils as bu  # noqa: E402

g = json.load(open(os.path.join(HERE, "proof_log5.json")))
sc = bu.synthetic_circuit(g["log_n"], g["num_inputs"], seed=g["circuit_seed"])
w, pubs = sc.witness(g["witness_seed"])
bl = bu.blinders(g["blinder_seed"])
tau = bu.SplitMix64(g["tau_seed"]).field()
msg = g["ext_msg"].encode()
with open(os.path.join(HERE, "harness_log5.bin"), "wb") as f:
    f.write(b"CAPH\0\0\0\0" + struct.pack("<QQQ", g["log_n"], g["num_inputs"], len(msg)))
    f.write(bu.to_canonical_array([tau]).tobytes())
    f.write(sc.selectors_mont().tobytes())
    f.write(sc.sigma_mont().tobytes())
    f.write(sc.wires_mont(w).tobytes())
    f.write(bu.to_mont_array(pubs).tobytes())
    f.write(bu.to_mont_array(bl).tobytes())
    f.write(msg)
print("wrote harness_log5.bin")
