"""Shared by the CPU (oracle) and GPU (parity) tests: golden fixtures and seeded inputs."""
import json
import os

import numpy as np

from oracle import bn254 as bn
from oracle import capref as cr

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gpu_present() -> bool:
    """Is there an AMD GPU this process can open?  (The kernel driver's device node - not torch.cuda.is_available(): the
    tests load libcapgpu.so before torch, so the process runs on /opt/rocm's HIP runtime, and torch's own view of the
    device is then beside the point - see tests/conftest.py.)"""
    return os.path.exists("/dev/kfd") and os.access("/dev/kfd", os.R_OK | os.W_OK)


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def unhex_pt(p):
    return None if p is None else (int(p[0], 16), int(p[1], 16))


def msm_inputs(vec):
    """golden MSM vector -> (bases as python points, scalars as ints)."""
    n = vec["n"]
    ks = bn.SplitMix64(vec["base_seed"])
    kvals = [ks.field(bn.R) for _ in range(n)]
    bases_arr = cr.g1_fixed_base_batch(cr.ints_to_array(kvals))
    bases = [cr.affine_to_ints(b) for b in bases_arr]
    sc = bn.SplitMix64(vec["scalar_seed"])
    scalars = [sc.field(bn.R) for _ in range(n)]
    if vec["edge"]:
        bases[7] = bases[6]
        bases[5] = None
        scalars = [int(s, 16) for s in vec["scalars"]]
    return bases, scalars


def seeded_fr(seed, n, mont=True) -> np.ndarray:
    return cr.random_field(seed, 1, n, mont)


def fr_to_ints(arr):
    return [bn.from_mont(v, bn.R) for v in cr.array_to_ints(arr)]


def srs_powers(tau, n) -> np.ndarray:
    """[tau^i] G for i < n as (n, 8) Montgomery affine (CPU, oracle)."""
    vals, x = [], 1
    for _ in range(n):
        vals.append(x)
        x = x * tau % bn.R
    return cr.g1_fixed_base_batch(cr.ints_to_array(vals))


def proof_points(pr):
    """ctypes capgpu Proof -> (13 affine points as canonical ints, 10 evals as canonical ints)."""
    pts = [cr.affine_to_ints(np.ctypeslib.as_array(pr.wires_poly_comms[i])) for i in range(5)]
    pts.append(cr.affine_to_ints(np.ctypeslib.as_array(pr.prod_perm_poly_comm)))
    pts += [cr.affine_to_ints(np.ctypeslib.as_array(pr.split_quot_poly_comms[i])) for i in range(5)]
    pts.append(cr.affine_to_ints(np.ctypeslib.as_array(pr.opening_proof)))
    pts.append(cr.affine_to_ints(np.ctypeslib.as_array(pr.shifted_opening_proof)))
    ev = [fr_to_ints(np.ctypeslib.as_array(pr.wires_evals[i]))[0] for i in range(5)]
    ev += [fr_to_ints(np.ctypeslib.as_array(pr.wire_sigma_evals[i]))[0] for i in range(4)]
    ev.append(fr_to_ints(np.ctypeslib.as_array(pr.perm_next_eval))[0])
    return pts, ev


def oracle_proof_points(pr):
    pts = pr.wires_poly_comms + [pr.prod_perm_poly_comm] + pr.split_quot_poly_comms + \
        [pr.opening_proof, pr.shifted_opening_proof]
    ev = pr.wires_evals + pr.wire_sigma_evals + [pr.perm_next_eval]
    return pts, ev


def cref_proof_points(comms, evals):
    return [cr.affine_to_ints(c) for c in comms], fr_to_ints(evals)
