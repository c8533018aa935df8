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
tau = bu.SplitMix64(g["tau_seed"]).field()


def write_instance(name, log_n, num_inputs, circuit_seed, witness_seed, blinder_seed, msg):
    sc = bu.synthetic_circuit(log_n, num_inputs, seed=circuit_seed)
    w, pubs = sc.witness(witness_seed)
    bl = bu.blinders(blinder_seed)
    with open(os.path.join(HERE, name), "wb") as f:
        f.write(b"CAPH\0\0\0\0" + struct.pack("<QQQ", log_n, num_inputs, len(msg)))
        f.write(bu.to_canonical_array([tau]).tobytes())
        f.write(sc.selectors_mont().tobytes())
        f.write(sc.sigma_mont().tobytes())
        f.write(sc.wires_mont(w).tobytes())
        f.write(bu.to_mont_array(pubs).tobytes())
        f.write(bu.to_mont_array(bl).tobytes())
        f.write(msg)
    print("wrote", name)


# the golden proof's instance
write_instance("harness_log5.bin", g["log_n"], g["num_inputs"], g["circuit_seed"], g["witness_seed"], g["blinder_seed"],
               g["ext_msg"].encode())
# a second, different circuit under the same SRS (tests/cpp/proof_api_test.cpp: the reference's tests prove under two
# keys and cross them for the negative cases, src/proof/transfer.rs:599-760); same number of public inputs so that a
# crossed call is a wrong proof, not a malformed one
write_instance("harness_b_log4.bin", 4, g["num_inputs"], 7, 101, 201, b"second-memo-key")
