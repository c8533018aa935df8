"""Host-side mirror of the reference's `parameters` module on top of the capgpu C ABI.

Same names, default paths and error behaviour as `/root/reference/src/parameters.rs`:
  store_universal_parameter_for_demo :47-65     load_universal_parameter :97-109
  store_/load_transfer_proving_key   :113-188   store_/load_transfer_verifying_key :190-241
  store_/load_mint_proving_key       :244-312   store_/load_mint_verifying_key     :314-362
  store_/load_freeze_proving_key     :364-436   store_/load_freeze_verifying_key   :438-478
  default paths                      :484-557   store_data / load_data             :560-577
Files hold ark-serialize `CanonicalSerialize` bytes of the jf-plonk key followed by the note-shape trailer of the
wrapper struct (src/proof/transfer.rs:59-64, mint.rs:49-52, freeze.rs:49-53).  The byte layout of the jf-plonk /
ark-poly-commit types is restated from the crates (not in the reference tree): parity with a file written by the
reference is unpinned.

Differences forced by scope: the reference's `store_*_proving_key` builds the circuit and preprocesses it first
(CPU, out of scope - SURVEY 8a A9); here the caller passes the key `cap_amd.proof.preprocess` produced.  I/O errors
raise `TxnApiError.IoError`, malformed blobs `TxnApiError.DeserializationError` (src/errors.rs:49-53, 81-90).
All point decompression happens in libcapgpu.so - there is no CPU fallback.
"""
from __future__ import annotations

import os
import struct
from dataclasses import dataclass
from pathlib import Path

from . import lib as _lib
from .proof import ProvingKey, TxnApiError, UniversalSrs, VerifyingKey

DEFAULT_UNIVERSAL_SRS_FILENAME = "universal_srs"


def _io_error(e: Exception) -> TxnApiError:
    return TxnApiError(f"IoError: {e}")


def _de_error(e: Exception) -> TxnApiError:
    return TxnApiError(f"DeserializationError: {e}")


# ---- default paths (src/parameters.rs:484-557): $CAP_UNIV_PARAM_DIR/data/<name>.bin -----------------------------
def default_path(filename: str, extension: str) -> Path:
    d = Path(os.environ.get("CAP_UNIV_PARAM_DIR", "."))
    return (d / "data" / filename).with_suffix("." + extension)


def default_transfer_proving_key_path(num_input: int, num_output: int, tree_depth: int) -> Path:
    return default_path(f"transfer_prover_{num_input}_input_{num_output}_output_{tree_depth}_depth", "bin")


def default_mint_proving_key_path(tree_depth: int) -> Path:
    return default_path(f"mint_prover_1_input_2_output_{tree_depth}_depth", "bin")


def default_freeze_proving_key_path(num_input: int, tree_depth: int) -> Path:
    return default_path(f"freeze_prover_{num_input}_input_{num_input}_output_{tree_depth}_depth", "bin")


def default_transfer_verifying_key_path(num_input: int, num_output: int, tree_depth: int) -> Path:
    return default_path(f"transfer_verifier_{num_input}_input_{num_output}_output_{tree_depth}_depth", "bin")


def default_mint_verifying_key_path(tree_depth: int) -> Path:
    return default_path(f"mint_verifier_1_input_2_output_{tree_depth}_depth", "bin")


def default_freeze_verifying_key_path(num_input: int, tree_depth: int) -> Path:
    return default_path(f"freeze_verifier_{num_input}_input_{num_input}_output_{tree_depth}_depth", "bin")


def store_bytes(data: bytes, dest: Path) -> None:
    try:
        with open(dest, "wb") as f:
            f.write(data)
    except OSError as e:
        raise _io_error(e) from e


def load_bytes(src: Path) -> bytes:
    try:
        with open(src, "rb") as f:
            return f.read()
    except OSError as e:
        raise _io_error(e) from e


# ---- note-shape wrappers (the fields the reference's tests assert after a load) ---------------------------------
@dataclass
class TransferProvingKey:
    proving_key: ProvingKey
    n_inputs: int
    n_outputs: int
    tree_depth: int


@dataclass
class TransferVerifyingKey:
    verifying_key: VerifyingKey
    n_inputs: int
    n_outputs: int
    tree_depth: int


@dataclass
class MintProvingKey:
    proving_key: ProvingKey
    tree_depth: int


@dataclass
class MintVerifyingKey:
    verifying_key: VerifyingKey
    tree_depth: int


@dataclass
class FreezeProvingKey:
    proving_key: ProvingKey
    tree_depth: int
    num_input: int


@dataclass
class FreezeVerifyingKey:
    verifying_key: VerifyingKey
    tree_depth: int
    num_input: int


def _transfer_trailer(n_in, n_out, depth) -> bytes:
    return struct.pack("<QQB", n_in, n_out, depth)


def _mint_trailer(depth) -> bytes:
    return struct.pack("<B", depth)


def _freeze_trailer(depth, n_in) -> bytes:
    return struct.pack("<BQ", depth, n_in)


def _unpack_trailer(fmt: str, data: bytes, used: int):
    size = struct.calcsize(fmt)
    if len(data) - used < size:
        raise _de_error(ValueError("unexpected end of input in the key trailer"))
    return struct.unpack(fmt, data[used:used + size])


# ---- universal parameter ----------------------------------------------------------------------------------------
def serialize_universal_parameter(srs: UniversalSrs) -> bytes:
    try:
        return _lib.srs_serialize(srs.handle, srs.h, srs.beta_h)
    except _lib.CapGpuError as e:
        raise TxnApiError.FailedSnark(str(e)) from e


def deserialize_universal_parameter(data: bytes, max_degree: int = 0) -> UniversalSrs:
    try:
        _lib.init()
        handle, h, beta_h, _ = _lib.srs_deserialize(data, max_degree)
    except _lib.CapGpuError as e:
        raise _de_error(e) from e
    return UniversalSrs(handle, _lib.srs_size(handle) - 1, h, beta_h)


def store_universal_parameter_for_demo(srs: UniversalSrs, dest: Path | None = None) -> None:
    """src/parameters.rs:47-65 (the SRS comes from `proof.universal_setup`; the reference draws it from test_rng)."""
    store_bytes(serialize_universal_parameter(srs), dest or default_path(DEFAULT_UNIVERSAL_SRS_FILENAME, "bin"))


def load_universal_parameter(src: Path | None = None, max_degree: int = 0) -> UniversalSrs:
    """src/parameters.rs:97-109.  With src = None the reference's bn254 build reads the embedded Aztec CRS
    (src/proof/mod.rs:90-93), which is not in the tree; here None means the default path."""
    return deserialize_universal_parameter(load_bytes(src or default_path(DEFAULT_UNIVERSAL_SRS_FILENAME, "bin")),
                                           max_degree)


# ---- proving / verifying keys -----------------------------------------------------------------------------------
def serialize_proving_key(pk: ProvingKey) -> bytes:
    try:
        return _lib.plonk_key_serialize(pk.handle, pk.srs.h, pk.srs.beta_h)
    except _lib.CapGpuError as e:
        raise TxnApiError.FailedSnark(str(e)) from e


def deserialize_proving_key(data: bytes):
    """-> (ProvingKey, bytes consumed).  The commit key inside the blob becomes the key's own device-resident SRS."""
    try:
        _lib.init()
        srs_h, pk_h, vk, h, beta_h, used = _lib.plonk_key_deserialize(data)
    except _lib.CapGpuError as e:
        raise _de_error(e) from e
    srs = UniversalSrs(srs_h, _lib.srs_size(srs_h) - 1, h, beta_h)
    return ProvingKey(pk_h, int(vk.domain_size), int(vk.num_inputs), srs), used


def serialize_verifying_key(vk: VerifyingKey, srs: UniversalSrs) -> bytes:
    try:
        g = _lib.srs_download(srs.handle, 0, 1).reshape(8)
        return _lib.plonk_vk_serialize(vk.raw, g, vk.h, vk.beta_h)
    except _lib.CapGpuError as e:
        raise TxnApiError.FailedSnark(str(e)) from e


def deserialize_verifying_key(data: bytes):
    """-> (VerifyingKey, bytes consumed).  Host only: needs no GPU."""
    try:
        vk, _g, _gg, h, beta_h, used = _lib.plonk_vk_deserialize(data)
    except _lib.CapGpuError as e:
        raise _de_error(e) from e
    return VerifyingKey(vk, int(vk.domain_size), int(vk.num_inputs), h, beta_h), used


def store_transfer_proving_key(key: TransferProvingKey, dest: Path | None = None) -> None:
    dest = dest or default_transfer_proving_key_path(key.n_inputs, key.n_outputs, key.tree_depth)
    store_bytes(serialize_proving_key(key.proving_key)
                + _transfer_trailer(key.n_inputs, key.n_outputs, key.tree_depth), dest)


def load_transfer_proving_key(num_input: int, num_output: int, tree_depth: int,
                              src: Path | None = None) -> TransferProvingKey:
    data = load_bytes(src or default_transfer_proving_key_path(num_input, num_output, tree_depth))
    pk, used = deserialize_proving_key(data)
    return TransferProvingKey(pk, *_unpack_trailer("<QQB", data, used))


def store_transfer_verifying_key(key: TransferVerifyingKey, srs: UniversalSrs, dest: Path | None = None) -> None:
    dest = dest or default_transfer_verifying_key_path(key.n_inputs, key.n_outputs, key.tree_depth)
    store_bytes(serialize_verifying_key(key.verifying_key, srs)
                + _transfer_trailer(key.n_inputs, key.n_outputs, key.tree_depth), dest)


def load_transfer_verifying_key(num_input: int, num_output: int, tree_depth: int,
                                src: Path | None = None) -> TransferVerifyingKey:
    data = load_bytes(src or default_transfer_verifying_key_path(num_input, num_output, tree_depth))
    vk, used = deserialize_verifying_key(data)
    return TransferVerifyingKey(vk, *_unpack_trailer("<QQB", data, used))


def store_mint_proving_key(key: MintProvingKey, dest: Path | None = None) -> None:
    store_bytes(serialize_proving_key(key.proving_key) + _mint_trailer(key.tree_depth),
                dest or default_mint_proving_key_path(key.tree_depth))


def load_mint_proving_key(tree_depth: int, src: Path | None = None) -> MintProvingKey:
    data = load_bytes(src or default_mint_proving_key_path(tree_depth))
    pk, used = deserialize_proving_key(data)
    return MintProvingKey(pk, *_unpack_trailer("<B", data, used))


def store_mint_verifying_key(key: MintVerifyingKey, srs: UniversalSrs, dest: Path | None = None) -> None:
    store_bytes(serialize_verifying_key(key.verifying_key, srs) + _mint_trailer(key.tree_depth),
                dest or default_mint_verifying_key_path(key.tree_depth))


def load_mint_verifying_key(tree_depth: int, src: Path | None = None) -> MintVerifyingKey:
    data = load_bytes(src or default_mint_verifying_key_path(tree_depth))
    vk, used = deserialize_verifying_key(data)
    return MintVerifyingKey(vk, *_unpack_trailer("<B", data, used))


def store_freeze_proving_key(key: FreezeProvingKey, dest: Path | None = None) -> None:
    store_bytes(serialize_proving_key(key.proving_key) + _freeze_trailer(key.tree_depth, key.num_input),
                dest or default_freeze_proving_key_path(key.num_input, key.tree_depth))


def load_freeze_proving_key(num_input: int, tree_depth: int, src: Path | None = None) -> FreezeProvingKey:
    data = load_bytes(src or default_freeze_proving_key_path(num_input, tree_depth))
    pk, used = deserialize_proving_key(data)
    return FreezeProvingKey(pk, *_unpack_trailer("<BQ", data, used))


def store_freeze_verifying_key(key: FreezeVerifyingKey, srs: UniversalSrs, dest: Path | None = None) -> None:
    store_bytes(serialize_verifying_key(key.verifying_key, srs) + _freeze_trailer(key.tree_depth, key.num_input),
                dest or default_freeze_verifying_key_path(key.num_input, key.tree_depth))


def load_freeze_verifying_key(num_input: int, tree_depth: int, src: Path | None = None) -> FreezeVerifyingKey:
    data = load_bytes(src or default_freeze_verifying_key_path(num_input, tree_depth))
    vk, used = deserialize_verifying_key(data)
    return FreezeVerifyingKey(vk, *_unpack_trailer("<BQ", data, used))
