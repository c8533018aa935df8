// capgpu_proof.hpp - C++17 host-side mirror of the reference's proof API above the C ABI of capgpu.h.
//
// The reference is compiled code (Rust); where its toolchain is missing this header is what a caller links instead of
// src/proof/: the same module layout, names, argument meaning and error behaviour -
//
//   capgpu::proof::universal_setup            src/proof/mod.rs:59-69
//   capgpu::proof::load_srs                   src/proof/mod.rs:74-109   (the bytes of data/aztec-crs-131072.bin - an
//                                             ark-serialized UniversalSrs, src/proof/mod.rs:106 - handed in by the caller:
//                                             the reference embeds them with include_bytes!, its tree does not ship them)
//   capgpu::proof::universal_setup_for_staging src/proof/mod.rs:121-141  (bn254: an alias of load_srs, rng ignored)
//   capgpu::proof::{transfer,mint,freeze}::preprocess   src/proof/transfer.rs:124-155, mint.rs:69-93, freeze.rs:93-121
//   capgpu::proof::{transfer,mint,freeze}::prove        src/proof/transfer.rs:159-188, mint.rs:97-120, freeze.rs:125-158
//   capgpu::proof::{transfer,mint,freeze}::verify       src/proof/transfer.rs:192-212, mint.rs:124-140, freeze.rs:162-178
//   capgpu::TransferProvingKey / ...VerifyingKey         src/proof/transfer.rs:55-108 (key + note shape)
//   capgpu::TxnApiError::FailedSnark                     src/errors.rs:25-63: what every SNARK failure maps to
//   capgpu::Result<T>                                    Rust's Result<T, TxnApiError>: is_ok / is_err / unwrap / error
//
// What differs, and why: the circuit builders (src/circuit/*.rs) stay on the CPU and are out of scope,
// so `preprocess` takes the finalised circuit they produce (selector columns + extended permutation) next to the note
// shape, and `prove` takes the finalised wire assignment + public-input scalars instead of Witness / PublicInput
// structs.  The random number generator argument of `prove` is any callable returning one Fr (Montgomery words): it is
// asked for the 13 blinders in jf-plonk's order.  Everything heavy runs in libcapgpu.so on the GPU; there is no CPU path
// and a missing device surfaces as Err(FailedSnark("... no CPU fallback ...")).
//
// Header only; link with -lcapgpu.  tests/cpp/proof_api_test.cpp is the reference's test_*_validity_proof restated on it.
#ifndef CAPGPU_PROOF_HPP
#define CAPGPU_PROOF_HPP

#include <array>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <variant>
#include <vector>

#include "capgpu.h"

namespace capgpu {

using Fr = std::array<uint64_t, 4>;  // arkworks' in-memory form: 4 x u64, little-endian limbs, Montgomery (R = 2^256)

// src/errors.rs:25-63 - only the variants this path produces
struct TxnApiError {
  enum Kind { FailedSnark, InvalidParameter, FailedSerialization } kind;
  std::string msg;
  static TxnApiError failed_snark(std::string m) { return {FailedSnark, std::move(m)}; }
  std::string to_string() const {
    const char* k = kind == FailedSnark ? "FailedSnark" : (kind == InvalidParameter ? "InvalidParameter" : "FailedSerialization");
    return std::string(k) + "(" + msg + ")";
  }
};

struct Unit {};  // Rust's ()

template <class T>
class Result {
 public:
  Result(T v) : v_(std::move(v)) {}
  Result(TxnApiError e) : v_(std::move(e)) {}
  bool is_ok() const { return v_.index() == 0; }
  bool is_err() const { return !is_ok(); }
  T& unwrap() {
    if (is_err()) throw std::runtime_error("called unwrap() on Err: " + error().to_string());
    return std::get<0>(v_);
  }
  const TxnApiError& error() const { return std::get<1>(v_); }

 private:
  std::variant<T, TxnApiError> v_;
};

namespace detail {
inline std::string lib_error(int rc) {
  const char* m = capgpu_last_error();
  return "capgpu error " + std::to_string(rc) + (m && *m ? std::string(": ") + m : std::string());
}
inline TxnApiError map_error(int rc, const std::string& what) {
  // ark_serialize::SerializationError -> TxnApiError::FailedSerialization (src/errors.rs), everything else in the SNARK
  // calls -> FailedSnark (src/proof/transfer.rs:187)
  if (rc == CAPGPU_ERR_SERIALIZATION) return {TxnApiError::FailedSerialization, what + ": " + lib_error(rc)};
  return TxnApiError::failed_snark(what + ": " + lib_error(rc));
}
struct SrsHandle {
  uint64_t h = 0;
  explicit SrsHandle(uint64_t v) : h(v) {}
  SrsHandle(const SrsHandle&) = delete;
  SrsHandle& operator=(const SrsHandle&) = delete;
  ~SrsHandle() {
    if (h) capgpu_srs_free(h);
  }
};
struct KeyHandle {
  uint64_t h = 0;
  explicit KeyHandle(uint64_t v) : h(v) {}
  KeyHandle(const KeyHandle&) = delete;
  KeyHandle& operator=(const KeyHandle&) = delete;
  ~KeyHandle() {
    if (h) capgpu_plonk_free_key(h);
  }
};
}  // namespace detail

// jf-plonk's UniversalSrs: powers of tau in G1 (device resident) + the G2 part of the open key
struct UniversalSrs {
  std::shared_ptr<detail::SrsHandle> powers_of_g;
  size_t max_degree = 0;
  std::array<uint64_t, 16> h{}, beta_h{};
  uint64_t handle() const { return powers_of_g ? powers_of_g->h : 0; }
};

// jf-plonk's VerifyingKey (+ open key); plain data, host only
struct VerifyingKey {
  capgpu_verifying_key raw{};
  std::array<uint64_t, 16> h{}, beta_h{};
};

// jf-plonk's ProvingKey: selector / sigma polynomials and the commit key, resident on the device; `vk` as in the crate
struct ProvingKey {
  std::shared_ptr<detail::KeyHandle> key;
  UniversalSrs srs;  // keeps the commit key alive
  VerifyingKey vk;
  uint64_t handle() const { return key ? key->h : 0; }
  size_t domain_size() const { return (size_t)vk.raw.domain_size; }
  size_t num_inputs() const { return (size_t)vk.raw.num_inputs; }
};

// jf-plonk's Proof<Bn254> (13 G1 + 10 Fr), in the C ABI's field order
using Proof = capgpu_proof;

// The finalised circuit the reference's builders hand to jf-plonk (PlonkCircuit after finalize_for_arithmetization):
// column-major Montgomery field elements.  Out of scope to build here; tests and benches use a synthetic satisfiable one.
struct FinalisedCircuit {
  size_t domain_size = 0;            // n, a power of two (Arithmetization::eval_domain_size)
  size_t num_inputs = 0;             // public inputs (the first gates)
  const uint64_t* selectors = nullptr;  // 13 x n x 4 words, gate order of capgpu.h
  const uint64_t* sigma = nullptr;      // 5 x n x 4 words: the extended permutation as field elements
  size_t num_gates = 0;              // constraints before padding (preprocess returns it, transfer.rs:155)
  // CAPGPU_INPUT_EVALS: the columns are tables of values on the domain; CAPGPU_INPUT_COEFFS: they are the polynomials
  // jf-relation's Arithmetization trait returns (compute_selector_polynomials / compute_extended_permutation_polynomials),
  // n coefficients each, passed through untransformed
  int input_form = CAPGPU_INPUT_EVALS;
};
struct Assignment {
  const uint64_t* wires = nullptr;       // 5 x n x 4 words
  const uint64_t* pub_inputs = nullptr;  // num_inputs x 4 words  (PublicInput::to_scalars())
  int input_form = CAPGPU_INPUT_EVALS;   // CAPGPU_INPUT_COEFFS: wires = compute_wire_polynomials(), unblinded
};

namespace proof {

// src/proof/mod.rs:59-69.  The reference samples tau from its rng (benches: test_rng); here the caller passes it
// (canonical words) - the SRS is generated on the device.
// `gamma`, when given, also produces the hiding powers [gamma tau^i] G, i <= max_degree + 1, that KZG10::setup samples
// next to tau: they travel with stored parameter files and keys; the prover does not use them.
inline Result<UniversalSrs> universal_setup(size_t max_degree, const Fr& tau, const Fr* gamma = nullptr) {
  int rc = capgpu_init(nullptr, 0);
  if (rc != CAPGPU_OK) return detail::map_error(rc, "Failed to generate universal SRS");
  uint64_t h = 0;
  rc = gamma ? capgpu_srs_generate_hiding(tau.data(), gamma->data(), max_degree + 1, &h)
             : capgpu_srs_generate(tau.data(), max_degree + 1, &h);
  if (rc != CAPGPU_OK) return detail::map_error(rc, "Failed to generate universal SRS");
  UniversalSrs s;
  s.powers_of_g = std::make_shared<detail::SrsHandle>(h);
  s.max_degree = max_degree;
  capgpu_g2_generator(s.h.data());
  capgpu_g2_mul(s.h.data(), tau.data(), s.beta_h.data());
  return s;
}

// SHA-256 (FIPS 180-4) of a byte string: load_srs checks the integrity of the CRS file like the reference does
// (sha2::Sha256, src/proof/mod.rs:95-102).  Host only, ~1 GB/s: the 4 MB file is 4 ms.
inline std::array<uint8_t, 32> sha256(const uint8_t* data, size_t len) {
  static const uint32_t K[64] = {
      0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
      0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
      0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
      0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
      0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
      0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
      0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
  uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  auto rotr = [](uint32_t x, int n) { return (x >> n) | (x << (32 - n)); };
  auto block = [&](const uint8_t* p) {
    uint32_t w[64];
    for (int i = 0; i < 16; i++)
      w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
    for (int i = 16; i < 64; i++) {
      const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
      const uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
      w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; i++) {
      const uint32_t t1 = hh + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
      const uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
      hh = g, g = f, f = e, e = d + t1, d = c, c = b, b = a, a = t1 + t2;
    }
    h[0] += a, h[1] += b, h[2] += c, h[3] += d, h[4] += e, h[5] += f, h[6] += g, h[7] += hh;
  };
  size_t i = 0;
  for (; i + 64 <= len; i += 64) block(data + i);
  uint8_t tail[128] = {0};
  const size_t rem = len - i;
  std::memcpy(tail, data + i, rem);
  tail[rem] = 0x80;
  const size_t tl = rem < 56 ? 64 : 128;
  const uint64_t bits = (uint64_t)len * 8;
  for (int k = 0; k < 8; k++) tail[tl - 1 - k] = (uint8_t)(bits >> (8 * k));
  block(tail);
  if (tl == 128) block(tail + 64);
  std::array<uint8_t, 32> out;
  for (int k = 0; k < 8; k++)
    for (int b2 = 0; b2 < 4; b2++) out[4 * k + b2] = (uint8_t)(h[k] >> (24 - 8 * b2));
  return out;
}

// SHA-256 of data/aztec-crs-131072.bin as pinned by the reference (src/proof/mod.rs:100)
inline const std::array<uint8_t, 32>& aztec_crs_sha256() {
  static const std::array<uint8_t, 32> d = {0x6b, 0x81, 0xe7, 0x5f, 0xb9, 0xc1, 0x4f, 0xd0, 0xe5, 0x8f, 0xb2,
                                            0xb2, 0x9e, 0x48, 0x97, 0x8c, 0xda, 0xd5, 0x51, 0x15, 0x03, 0x68,
                                            0x5a, 0x61, 0xf1, 0x39, 0x1d, 0xc4, 0xa4, 0xfc, 0x7c, 0xbf};
  return d;
}

// src/proof/mod.rs:74-109.  `bytes`: the CRS file the reference embeds (an ark-serialized UniversalSrs).  As there:
//   - max_degree > 2^17 is refused with the reference's message (mod.rs:83-88);
//   - the SHA-256 of the bytes must equal `expected_sha256` - the reference's pinned digest unless the caller states
//     another one (a test SRS) - or the call PANICS like the reference's assert_eq! (mod.rs:96-102): std::runtime_error;
//   - the WHOLE file is deserialized and returned; max_degree plays no other role (mod.rs:106).
inline Result<UniversalSrs> load_srs(size_t max_degree, const std::vector<uint8_t>& bytes,
                                     const std::array<uint8_t, 32>& expected_sha256 = aztec_crs_sha256()) {
  if (max_degree > ((size_t)1 << 17))
    return TxnApiError::failed_snark("Currently only supports 2^17. Please update Aztec's CRS data file if needed.");
  if (sha256(bytes.data(), bytes.size()) != expected_sha256)
    throw std::runtime_error("Mismatched sha256sum digest, file might be corrupted!");
  int rc = capgpu_init(nullptr, 0);
  if (rc != CAPGPU_OK) return detail::map_error(rc, "Failed to load SRS");
  UniversalSrs s;
  uint64_t h = 0;
  size_t used = 0, n = 0;
  rc = capgpu_srs_deserialize(bytes.data(), bytes.size(), 0, &h, s.h.data(), s.beta_h.data(), &used);
  if (rc != CAPGPU_OK) return detail::map_error(rc, "Failed to load SRS");
  s.powers_of_g = std::make_shared<detail::SrsHandle>(h);
  capgpu_srs_size(h, &n);
  s.max_degree = n ? n - 1 : 0;
  return s;
}

// src/proof/mod.rs:121-141: "a unified API for SRS generation for testing/staging" - under the reference's default
// feature (bn254, the one this library implements) it IGNORES its rng and loads Aztec's CRS: a one-line alias of
// load_srs.  The rng parameter is kept (any type, unused) so that call sites read like the reference's.
template <class Rng>
inline Result<UniversalSrs> universal_setup_for_staging(size_t max_degree, Rng& /*rng*/, const std::vector<uint8_t>& crs_bytes,
                                                        const std::array<uint8_t, 32>& expected_sha256 = aztec_crs_sha256()) {
  return load_srs(max_degree, crs_bytes, expected_sha256);
}

namespace detail_snark {
inline Result<std::pair<ProvingKey, VerifyingKey>> preprocess(const UniversalSrs& srs, const FinalisedCircuit& c,
                                                              const std::string& what) {
  if (!c.selectors || !c.sigma) return TxnApiError::failed_snark(what + ": circuit not finalised");
  ProvingKey pk;
  uint64_t h = 0;
  int rc = capgpu_plonk_preprocess_ex(srs.handle(), c.domain_size, c.num_inputs, c.selectors, c.sigma, c.input_form, &h,
                                      &pk.vk.raw);
  if (rc != CAPGPU_OK) return detail::map_error(rc, what);
  pk.key = std::make_shared<detail::KeyHandle>(h);
  pk.srs = srs;
  pk.vk.h = srs.h;
  pk.vk.beta_h = srs.beta_h;
  VerifyingKey vk = pk.vk;
  return std::make_pair(std::move(pk), vk);
}
template <class Rng>
Result<Proof> prove(Rng& rng, const ProvingKey& pk, const Assignment& a, const std::vector<uint8_t>* ext_msg,
                    const std::string& what) {
  if (!a.wires || (!a.pub_inputs && pk.num_inputs())) return TxnApiError::failed_snark(what + ": empty assignment");
  std::array<uint64_t, 13 * 4> blinders;  // 2 per wire polynomial, 3 for the permutation product (jf-plonk's order)
  for (int i = 0; i < 13; i++) {
    Fr b = rng();
    std::memcpy(&blinders[4 * i], b.data(), 32);
  }
  Proof p;
  int rc = capgpu_plonk_prove_ex(pk.handle(), a.wires, a.pub_inputs, pk.num_inputs(),
                                 ext_msg ? ext_msg->data() : nullptr, ext_msg ? ext_msg->size() : 0, blinders.data(),
                                 a.input_form, &p);
  if (rc != CAPGPU_OK) return detail::map_error(rc, what);
  return p;
}
inline Result<Unit> verify(const VerifyingKey& vk, const std::vector<Fr>& pub_inputs, const Proof& proof,
                           const std::vector<uint8_t>* ext_msg, const std::string& what) {
  int ok = 0;
  int rc = capgpu_plonk_verify(&vk.raw, vk.h.data(), vk.beta_h.data(),
                               pub_inputs.empty() ? nullptr : pub_inputs[0].data(), pub_inputs.size(), &proof,
                               ext_msg ? ext_msg->data() : nullptr, ext_msg ? ext_msg->size() : 0, &ok);
  if (rc != CAPGPU_OK) return detail::map_error(rc, what);
  if (!ok) return TxnApiError::failed_snark(what + ": WrongProof");
  return Unit{};
}
// ext_msg = CanonicalSerialize(ver_key) || extra_proof_bound_data (src/proof/transfer.rs:177-180); the caller passes
// the 32 serialised bytes of the Schnorr verification key (the embedded-curve point, out of scope to compute here)
inline std::vector<uint8_t> bound_message(const std::vector<uint8_t>& ver_key_bytes,
                                          const std::vector<uint8_t>& extra_proof_bound_data) {
  std::vector<uint8_t> m(ver_key_bytes);
  m.insert(m.end(), extra_proof_bound_data.begin(), extra_proof_bound_data.end());
  return m;
}
}  // namespace detail_snark

}  // namespace proof

// ---- Transfer (src/proof/transfer.rs) ----------------------------------------------------------------------------
struct TransferProvingKey {
  ProvingKey proving_key;
  size_t n_inputs = 0, n_outputs = 0;
  uint8_t tree_depth = 0;
  size_t num_input() const { return n_inputs; }    // transfer.rs:100-102
  size_t num_output() const { return n_outputs; }  // transfer.rs:105-107
};
struct TransferVerifyingKey {
  VerifyingKey verifying_key;
  size_t n_inputs = 0, n_outputs = 0;
  uint8_t tree_depth = 0;
  TransferVerifyingKey() = default;
  explicit TransferVerifyingKey(const TransferProvingKey& pk)  // impl From<&TransferProvingKey>, transfer.rs:110-119
      : verifying_key(pk.proving_key.vk), n_inputs(pk.n_inputs), n_outputs(pk.n_outputs), tree_depth(pk.tree_depth) {}
};
// ---- Mint (src/proof/mint.rs:45-66) ------------------------------------------------------------------------------
struct MintProvingKey {
  ProvingKey proving_key;
  uint8_t tree_depth = 0;
};
struct MintVerifyingKey {
  VerifyingKey verifying_key;
  uint8_t tree_depth = 0;
  MintVerifyingKey() = default;
  explicit MintVerifyingKey(const MintProvingKey& pk) : verifying_key(pk.proving_key.vk), tree_depth(pk.tree_depth) {}
};
// ---- Freeze (src/proof/freeze.rs:44-90) --------------------------------------------------------------------------
struct FreezeProvingKey {
  ProvingKey proving_key;
  uint8_t tree_depth = 0;
  size_t num_input = 0;
};
struct FreezeVerifyingKey {
  VerifyingKey verifying_key;
  uint8_t tree_depth = 0;
  size_t num_input = 0;
  FreezeVerifyingKey() = default;
  explicit FreezeVerifyingKey(const FreezeProvingKey& pk)
      : verifying_key(pk.proving_key.vk), tree_depth(pk.tree_depth), num_input(pk.num_input) {}
};

namespace proof {
namespace transfer {
struct Preprocessed {
  TransferProvingKey proving_key;
  TransferVerifyingKey verifying_key;
  size_t n_constraints;
};
// transfer.rs:124-155.  `circuit` stands for TransferCircuit::build_for_preprocessing(n_inputs, n_outputs, tree_depth).
inline Result<Preprocessed> preprocess(const UniversalSrs& srs, size_t n_inputs, size_t n_outputs, uint8_t tree_depth,
                                       const FinalisedCircuit& circuit) {
  auto r = detail_snark::preprocess(srs, circuit,
                                    "Preprocessing Transfer circuit of " + std::to_string(n_inputs) + "-in-" +
                                        std::to_string(n_outputs) + "-out-" + std::to_string(tree_depth) + "-depth failed");
  if (r.is_err()) return r.error();
  TransferProvingKey pk{std::move(r.unwrap().first), n_inputs, n_outputs, tree_depth};
  TransferVerifyingKey vk(pk);
  return Preprocessed{std::move(pk), vk, circuit.num_gates};
}
// transfer.rs:159-188.  The satisfiability check of the reference (check_circuit_satisfiability, :169-176) is the
// circuit builder's; an unsatisfied assignment is still caught here: the quotient's degree check fails and the call
// returns Err(FailedSnark) like the reference does.
template <class Rng>
Result<Proof> prove(Rng& rng, const TransferProvingKey& pk, const Assignment& witness,
                    const std::vector<uint8_t>& txn_memo_ver_key, const std::vector<uint8_t>& extra_proof_bound_data) {
  const std::vector<uint8_t> ext_msg = detail_snark::bound_message(txn_memo_ver_key, extra_proof_bound_data);
  return detail_snark::prove(rng, pk.proving_key, witness, &ext_msg, "Transfer Proof Creation failure");
}
// transfer.rs:192-212
inline Result<Unit> verify(const TransferVerifyingKey& vk, const std::vector<Fr>& public_inputs, const Proof& proof,
                           const std::vector<uint8_t>& recv_memos_ver_key,
                           const std::vector<uint8_t>& extra_proof_bound_data) {
  const std::vector<uint8_t> ext_msg = detail_snark::bound_message(recv_memos_ver_key, extra_proof_bound_data);
  return detail_snark::verify(vk.verifying_key, public_inputs, proof, &ext_msg, "Transfer Proof Verification failure");
}
}  // namespace transfer

namespace mint {
struct Preprocessed {
  MintProvingKey proving_key;
  MintVerifyingKey verifying_key;
  size_t n_constraints;
};
// mint.rs:69-93
inline Result<Preprocessed> preprocess(const UniversalSrs& srs, uint8_t tree_depth, const FinalisedCircuit& circuit) {
  auto r = detail_snark::preprocess(srs, circuit, "Preprocessing Mint circuit of depth " + std::to_string(tree_depth) + " failed");
  if (r.is_err()) return r.error();
  MintProvingKey pk{std::move(r.unwrap().first), tree_depth};
  MintVerifyingKey vk(pk);
  return Preprocessed{std::move(pk), vk, circuit.num_gates};
}
// mint.rs:97-120: ext_msg = serialised txn memo verification key, no extra bound data
template <class Rng>
Result<Proof> prove(Rng& rng, const MintProvingKey& pk, const Assignment& witness,
                    const std::vector<uint8_t>& txn_memo_ver_key) {
  return detail_snark::prove(rng, pk.proving_key, witness, &txn_memo_ver_key, "Mint Proof creation failure");
}
// mint.rs:124-140
inline Result<Unit> verify(const MintVerifyingKey& vk, const std::vector<Fr>& public_inputs, const Proof& proof,
                           const std::vector<uint8_t>& recv_memos_ver_key) {
  return detail_snark::verify(vk.verifying_key, public_inputs, proof, &recv_memos_ver_key,
                              "Mint Proof verification failure");
}
}  // namespace mint

namespace freeze {
struct Preprocessed {
  FreezeProvingKey proving_key;
  FreezeVerifyingKey verifying_key;
  size_t n_constraints;
};
// freeze.rs:93-121
inline Result<Preprocessed> preprocess(const UniversalSrs& srs, size_t num_input, uint8_t tree_depth,
                                       const FinalisedCircuit& circuit) {
  auto r = detail_snark::preprocess(srs, circuit,
                                    "Preprocessing Freeze circuit of " + std::to_string(num_input) + "-inputs-" +
                                        std::to_string(tree_depth) + "-depth failed");
  if (r.is_err()) return r.error();
  FreezeProvingKey pk{std::move(r.unwrap().first), tree_depth, num_input};
  FreezeVerifyingKey vk(pk);
  return Preprocessed{std::move(pk), vk, circuit.num_gates};
}
// freeze.rs:125-158: ext_msg = serialised txn memo verification key only
template <class Rng>
Result<Proof> prove(Rng& rng, const FreezeProvingKey& pk, const Assignment& witness,
                    const std::vector<uint8_t>& txn_memo_ver_key) {
  return detail_snark::prove(rng, pk.proving_key, witness, &txn_memo_ver_key, "Freeze Proof creation failure");
}
// freeze.rs:162-178
inline Result<Unit> verify(const FreezeVerifyingKey& vk, const std::vector<Fr>& public_inputs, const Proof& proof,
                           const std::vector<uint8_t>& recv_memos_ver_key) {
  return detail_snark::verify(vk.verifying_key, public_inputs, proof, &recv_memos_ver_key,
                              "Freeze Proof Verification failure");
}
}  // namespace freeze
}  // namespace proof

// txn_batch_verify's SNARK part (src/lib.rs:455-529, PlonkKzgSnark::batch_verify at :517-522): one pairing product for
// proofs of different keys under one SRS.  on_device: the group arithmetic runs on the GPU (capgpu_plonk_batch_verify_dev).
struct BatchItem {
  const VerifyingKey* verifying_key;
  const std::vector<Fr>* public_inputs;
  const Proof* proof;
  const std::vector<uint8_t>* ext_msg;  // the bound message of that note (may be null)
};
inline Result<Unit> batch_verify(const std::vector<BatchItem>& items, bool on_device = false) {
  if (items.empty()) return Unit{};
  std::vector<const capgpu_verifying_key*> vks;
  std::vector<const uint64_t*> pubs;
  std::vector<size_t> nin, lens;
  std::vector<const capgpu_proof*> proofs;
  std::vector<const uint8_t*> msgs;
  for (const BatchItem& it : items) {
    vks.push_back(&it.verifying_key->raw);
    pubs.push_back(it.public_inputs->empty() ? nullptr : (*it.public_inputs)[0].data());
    nin.push_back(it.public_inputs->size());
    proofs.push_back(it.proof);
    msgs.push_back(it.ext_msg && !it.ext_msg->empty() ? it.ext_msg->data() : nullptr);
    lens.push_back(it.ext_msg ? it.ext_msg->size() : 0);
  }
  const VerifyingKey& k0 = *items[0].verifying_key;
  int ok = 0;
  auto fn = on_device ? capgpu_plonk_batch_verify_dev : capgpu_plonk_batch_verify;
  int rc = fn(vks.data(), k0.h.data(), k0.beta_h.data(), pubs.data(), nin.data(), proofs.data(), msgs.data(), lens.data(),
              items.size(), &ok);
  if (rc != CAPGPU_OK) return detail::map_error(rc, "Batch Proof Verification failure");
  if (!ok) return TxnApiError::failed_snark("Batch Proof Verification failure: WrongProof");
  return Unit{};
}

// `Proof` inside a note is ark-serialize bytes (src/transfer.rs:60): CanonicalSerialize / CanonicalDeserialize
inline Result<std::vector<uint8_t>> serialize(const Proof& p) {
  std::vector<uint8_t> out(1024);
  size_t len = 0;
  int rc = capgpu_proof_serialize(&p, out.data(), out.size(), &len);
  if (rc != CAPGPU_OK) return detail::map_error(rc, "Proof serialization");
  out.resize(len);
  return out;
}
inline Result<Proof> deserialize_proof(const std::vector<uint8_t>& bytes) {
  Proof p;
  size_t used = 0;
  int rc = capgpu_proof_deserialize(bytes.data(), bytes.size(), &p, &used);
  if (rc != CAPGPU_OK) return detail::map_error(rc, "Proof deserialization");
  return p;
}

}  // namespace capgpu

#endif  // CAPGPU_PROOF_HPP
