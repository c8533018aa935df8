// The reference's own proof tests restated on include/capgpu_proof.hpp (the C++ host mirror of src/proof/):
//   test_transfer_validity_proof   src/proof/transfer.rs:599-760
//   test_mint_validity_proof       src/proof/mint.rs:344-471
//   test_freeze_validity_proof     src/proof/freeze.rs:429-534
// Same shape: one universal setup, two preprocessed keys, prove + verify under each, then the bad paths - wrong public
// inputs, wrong verifying key, wrong proof, wrong bound data / memo key.  The circuits come from two fixture files
// (tests/golden/make_harness_input.py) because the circuit builders are out of scope; the first is the golden log-5
// instance, so its proof bytes are also printed ("PROOF <hex>") and compared with the oracle's by tests/test_cpp_api.py.
//
//   g++ -std=c++17 -I include tests/cpp/proof_api_test.cpp -L cap_amd -lcapgpu -Wl,-rpath,$PWD/cap_amd -o proof_api_test
//   ./proof_api_test tests/golden/harness_log5.bin tests/golden/harness_b_log4.bin
// Exit codes: 0 ok, 2 no usable GPU (universal_setup returned Err: no CPU fallback), 1 a failed assertion.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "capgpu_proof.hpp"

using namespace capgpu;

struct Instance {
  unsigned log_n = 0;
  size_t n = 0, num_inputs = 0;
  Fr tau{};
  std::vector<uint64_t> selectors, sigma, wires, pubs, blinders;
  std::vector<uint8_t> msg;
  FinalisedCircuit circuit() const { return {n, num_inputs, selectors.data(), sigma.data(), n}; }
  Assignment assignment() const { return {wires.data(), pubs.data()}; }
  std::vector<Fr> public_inputs() const {
    std::vector<Fr> v(num_inputs);
    for (size_t i = 0; i < num_inputs; i++) std::memcpy(v[i].data(), &pubs[4 * i], 32);
    return v;
  }
};

static std::vector<uint64_t> words(std::ifstream& f, size_t count) {
  std::vector<uint64_t> v(count);
  f.read(reinterpret_cast<char*>(v.data()), 8 * count);
  if (!f) {
    std::fprintf(stderr, "short input file\n");
    std::exit(1);
  }
  return v;
}
static Instance load(const char* path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) {
    std::perror(path);
    std::exit(1);
  }
  auto hdr = words(f, 4);
  if (std::memcmp(hdr.data(), "CAPH\0\0\0\0", 8) != 0) {
    std::fprintf(stderr, "bad magic\n");
    std::exit(1);
  }
  Instance in;
  in.log_n = (unsigned)hdr[1];
  in.n = (size_t)1 << in.log_n;
  in.num_inputs = (size_t)hdr[2];
  auto t = words(f, 4);
  std::memcpy(in.tau.data(), t.data(), 32);
  in.selectors = words(f, 13 * in.n * 4);
  in.sigma = words(f, 5 * in.n * 4);
  in.wires = words(f, 5 * in.n * 4);
  in.pubs = words(f, in.num_inputs * 4);
  in.blinders = words(f, 13 * 4);
  in.msg.resize((size_t)hdr[3]);
  f.read(reinterpret_cast<char*>(in.msg.data()), (std::streamsize)in.msg.size());
  return in;
}

// the `rng` of prove(): hands out the fixture's 13 blinders in order (the reference draws them from its RngCore)
struct FixtureRng {
  const std::vector<uint64_t>& b;
  size_t next = 0;
  Fr operator()() {
    Fr v;
    std::memcpy(v.data(), &b[4 * (next++ % 13)], 32);
    return v;
  }
};

static int failures = 0;
#define ASSERT(cond)                                                        \
  do {                                                                      \
    if (!(cond)) {                                                          \
      std::fprintf(stderr, "%s:%d: assertion failed: %s\n", __FILE__, __LINE__, #cond); \
      failures++;                                                           \
    }                                                                       \
  } while (0)

int main(int argc, char** argv) {
  if (argc < 3) {
    std::fprintf(stderr, "usage: %s instance_1.bin instance_2.bin\n", argv[0]);
    return 1;
  }
  const Instance in1 = load(argv[1]), in2 = load(argv[2]);

  // ---- test_transfer_validity_proof ------------------------------------------------------------------------------
  const size_t max_degree = in1.n + 2;  // compute_universal_param_size: domain + 2 (src/utils/mod.rs:108-112)
  auto setup = proof::universal_setup(max_degree, in1.tau);
  if (setup.is_err()) {
    std::fprintf(stderr, "universal_setup: %s\n", setup.error().to_string().c_str());
    return 2;
  }
  const UniversalSrs universal_param = setup.unwrap();

  // the memo verification key is the first bytes of the fixture's message, the rest is the extra bound data: together
  // they are the golden instance's transcript init message
  const size_t key_len = in1.msg.size() < 4 ? in1.msg.size() : 4;
  const std::vector<uint8_t> recv_memos_ver_key(in1.msg.begin(), in1.msg.begin() + key_len);
  const std::vector<uint8_t> extra_proof_bound_data(in1.msg.begin() + key_len, in1.msg.end());

  auto pre1 = proof::transfer::preprocess(universal_param, 2, 2, 26, in1.circuit());
  ASSERT(pre1.is_ok());
  if (pre1.is_err()) return 1;
  const TransferProvingKey& proving_key_1 = pre1.unwrap().proving_key;
  const TransferVerifyingKey& verifying_key_1 = pre1.unwrap().verifying_key;
  ASSERT(pre1.unwrap().n_constraints == in1.n);
  ASSERT(proving_key_1.num_input() == 2 && proving_key_1.num_output() == 2);
  ASSERT(proving_key_1.proving_key.domain_size() == in1.n);
  {
    TransferVerifyingKey from_pk(proving_key_1);  // impl From<&TransferProvingKey>
    ASSERT(std::memcmp(&from_pk.verifying_key.raw, &verifying_key_1.verifying_key.raw, sizeof(capgpu_verifying_key)) == 0);
  }

  FixtureRng rng1{in1.blinders};
  const std::vector<Fr> pub_input_1 = in1.public_inputs();
  auto validity_proof_1 =
      proof::transfer::prove(rng1, proving_key_1, in1.assignment(), recv_memos_ver_key, extra_proof_bound_data);
  ASSERT(validity_proof_1.is_ok());
  if (validity_proof_1.is_err()) {
    std::fprintf(stderr, "%s\n", validity_proof_1.error().to_string().c_str());
    return 1;
  }
  ASSERT(proof::transfer::verify(verifying_key_1, pub_input_1, validity_proof_1.unwrap(), recv_memos_ver_key,
                                 extra_proof_bound_data)
             .is_ok());
  {
    auto bytes = serialize(validity_proof_1.unwrap());
    ASSERT(bytes.is_ok());
    std::printf("PROOF ");
    for (uint8_t b : bytes.unwrap()) std::printf("%02x", b);
    std::printf("\n");
    auto back = deserialize_proof(bytes.unwrap());
    ASSERT(back.is_ok() && std::memcmp(&back.unwrap(), &validity_proof_1.unwrap(), sizeof(Proof)) == 0);
    std::vector<uint8_t> cut(bytes.unwrap().begin(), bytes.unwrap().end() - 5);
    auto bad = deserialize_proof(cut);
    ASSERT(bad.is_err() && bad.error().kind == TxnApiError::FailedSerialization);
  }

  // second key under the same universal parameters
  auto pre2 = proof::transfer::preprocess(universal_param, 1, 2, 26, in2.circuit());
  ASSERT(pre2.is_ok());
  if (pre2.is_err()) return 1;
  const TransferProvingKey& proving_key_2 = pre2.unwrap().proving_key;
  const TransferVerifyingKey& verifying_key_2 = pre2.unwrap().verifying_key;
  FixtureRng rng2{in2.blinders};
  const std::vector<Fr> pub_input_2 = in2.public_inputs();
  auto validity_proof_2 =
      proof::transfer::prove(rng2, proving_key_2, in2.assignment(), recv_memos_ver_key, extra_proof_bound_data);
  ASSERT(validity_proof_2.is_ok());
  ASSERT(proof::transfer::verify(verifying_key_2, pub_input_2, validity_proof_2.unwrap(), recv_memos_ver_key,
                                 extra_proof_bound_data)
             .is_ok());

  // bad paths (transfer.rs:683-758)
  // wrong pub inputs
  ASSERT(proof::transfer::verify(verifying_key_1, pub_input_2, validity_proof_1.unwrap(), recv_memos_ver_key,
                                 extra_proof_bound_data)
             .is_err());
  ASSERT(proof::transfer::verify(verifying_key_2, pub_input_1, validity_proof_2.unwrap(), recv_memos_ver_key,
                                 extra_proof_bound_data)
             .is_err());
  // wrong verifying key
  ASSERT(proof::transfer::verify(verifying_key_2, pub_input_1, validity_proof_1.unwrap(), recv_memos_ver_key,
                                 extra_proof_bound_data)
             .is_err());
  ASSERT(proof::transfer::verify(verifying_key_1, pub_input_2, validity_proof_2.unwrap(), recv_memos_ver_key,
                                 extra_proof_bound_data)
             .is_err());
  // wrong proof
  ASSERT(proof::transfer::verify(verifying_key_2, pub_input_2, validity_proof_1.unwrap(), recv_memos_ver_key,
                                 extra_proof_bound_data)
             .is_err());
  ASSERT(proof::transfer::verify(verifying_key_1, pub_input_1, validity_proof_2.unwrap(), recv_memos_ver_key,
                                 extra_proof_bound_data)
             .is_err());
  // wrong memo verification key, wrong extra bound data
  {
    std::vector<uint8_t> bad_key(recv_memos_ver_key);
    bad_key.push_back(1);
    auto r = proof::transfer::verify(verifying_key_1, pub_input_1, validity_proof_1.unwrap(), bad_key, extra_proof_bound_data);
    ASSERT(r.is_err() && r.error().kind == TxnApiError::FailedSnark);
    const std::vector<uint8_t> bad_extra = {'w', 'r', 'o', 'n', 'g'};
    ASSERT(proof::transfer::verify(verifying_key_1, pub_input_1, validity_proof_1.unwrap(), recv_memos_ver_key, bad_extra)
               .is_err());
  }
  // an unsatisfied assignment: the reference refuses it before proving (check_circuit_satisfiability, transfer.rs:169-176);
  // here the prover's quotient degree check refuses it - Err(FailedSnark) either way
  {
    std::vector<uint64_t> bad_wires(in1.wires);
    bad_wires[4 * (in1.n / 2)] ^= 1;
    FixtureRng rng{in1.blinders};
    auto r = proof::transfer::prove(rng, proving_key_1, Assignment{bad_wires.data(), in1.pubs.data()}, recv_memos_ver_key,
                                    extra_proof_bound_data);
    ASSERT(r.is_err() && r.error().kind == TxnApiError::FailedSnark);
  }

  // ---- test_mint_validity_proof / test_freeze_validity_proof: same flow, their own key wrappers and bound message ----
  {
    auto pre = proof::mint::preprocess(universal_param, 26, in2.circuit());
    ASSERT(pre.is_ok());
    FixtureRng rng{in2.blinders};
    auto p = proof::mint::prove(rng, pre.unwrap().proving_key, in2.assignment(), recv_memos_ver_key);
    ASSERT(p.is_ok());
    ASSERT(proof::mint::verify(pre.unwrap().verifying_key, pub_input_2, p.unwrap(), recv_memos_ver_key).is_ok());
    ASSERT(proof::mint::verify(pre.unwrap().verifying_key, pub_input_1, p.unwrap(), recv_memos_ver_key).is_err());
    std::vector<uint8_t> other_key = {9, 9, 9};
    ASSERT(proof::mint::verify(pre.unwrap().verifying_key, pub_input_2, p.unwrap(), other_key).is_err());
    ASSERT(MintVerifyingKey(pre.unwrap().proving_key).tree_depth == 26);
  }
  {
    auto pre = proof::freeze::preprocess(universal_param, 2, 5, in2.circuit());
    ASSERT(pre.is_ok());
    FixtureRng rng{in2.blinders};
    auto p = proof::freeze::prove(rng, pre.unwrap().proving_key, in2.assignment(), recv_memos_ver_key);
    ASSERT(p.is_ok());
    ASSERT(proof::freeze::verify(pre.unwrap().verifying_key, pub_input_2, p.unwrap(), recv_memos_ver_key).is_ok());
    ASSERT(proof::freeze::verify(pre.unwrap().verifying_key, pub_input_1, p.unwrap(), recv_memos_ver_key).is_err());
    // same circuit as proving_key_2, but the transfer proof is bound to memo key || extra data: not a freeze proof
    ASSERT(proof::freeze::verify(FreezeVerifyingKey(pre.unwrap().proving_key), pub_input_2, validity_proof_2.unwrap(),
                                 recv_memos_ver_key)
               .is_err());
    ASSERT(pre.unwrap().proving_key.num_input == 2);
  }
  // txn_batch_verify's SNARK part: both proofs (two keys, one SRS) under one pairing product, host and device forms
  {
    const std::vector<uint8_t> msg = proof::detail_snark::bound_message(recv_memos_ver_key, extra_proof_bound_data);
    std::vector<BatchItem> items = {{&verifying_key_1.verifying_key, &pub_input_1, &validity_proof_1.unwrap(), &msg},
                                    {&verifying_key_2.verifying_key, &pub_input_2, &validity_proof_2.unwrap(), &msg}};
    ASSERT(batch_verify(items).is_ok());
    ASSERT(batch_verify(items, true).is_ok());
    std::swap(items[0].proof, items[1].proof);
    ASSERT(batch_verify(items).is_err());
    ASSERT(batch_verify(items, true).is_err());
  }
  // a circuit larger than the universal parameters is refused at preprocessing, as Err
  {
    auto small = proof::universal_setup(in2.n / 2, in1.tau);
    ASSERT(small.is_ok());
    auto r = proof::transfer::preprocess(small.unwrap(), 2, 2, 26, in1.circuit());
    ASSERT(r.is_err() && r.error().kind == TxnApiError::FailedSnark);
  }

  // load_srs (src/proof/mod.rs:74-109): degree bound, SHA-256 integrity check, the whole file is deserialized.  The
  // CRS bytes here are this SRS's own ark-serialize form (the Aztec file is not shipped with the reference)
  {
    // FIPS 180-4 known answers pin the digest the integrity check rests on
    const std::array<uint8_t, 32> abc = proof::sha256(reinterpret_cast<const uint8_t*>("abc"), 3);
    ASSERT(abc[0] == 0xba && abc[1] == 0x78 && abc[2] == 0x16 && abc[31] == 0xad);
    const std::string two_blocks = "abcdbcdecdefdefgefghfghighijhijkijkljklmklmnlmnomnopnopq";
    const std::array<uint8_t, 32> d2 = proof::sha256(reinterpret_cast<const uint8_t*>(two_blocks.data()), two_blocks.size());
    ASSERT(d2[0] == 0x24 && d2[1] == 0x8d && d2[30] == 0x06 && d2[31] == 0xc1);
    size_t need = 0;
    ASSERT(capgpu_srs_serialize(universal_param.handle(), universal_param.h.data(), universal_param.beta_h.data(), nullptr,
                                0, &need) == CAPGPU_OK);
    std::vector<uint8_t> crs(need);
    ASSERT(capgpu_srs_serialize(universal_param.handle(), universal_param.h.data(), universal_param.beta_h.data(),
                                crs.data(), crs.size(), &need) == CAPGPU_OK);
    const std::array<uint8_t, 32> digest = proof::sha256(crs.data(), crs.size());
    auto too_big = proof::load_srs(((size_t)1 << 17) + 1, crs, digest);
    ASSERT(too_big.is_err() && too_big.error().kind == TxnApiError::FailedSnark &&
           too_big.error().msg.find("only supports 2^17") != std::string::npos);
    bool panicked = false;
    try {
      (void)proof::load_srs(max_degree, crs);  // the reference's pinned digest: these are not the Aztec bytes
    } catch (const std::runtime_error& e) {
      panicked = std::string(e.what()).find("Mismatched sha256sum digest") != std::string::npos;
    }
    ASSERT(panicked);
    std::vector<uint8_t> corrupted(crs);
    corrupted[crs.size() / 2] ^= 1;
    panicked = false;
    try {
      (void)proof::load_srs(max_degree, corrupted, digest);
    } catch (const std::runtime_error&) {
      panicked = true;
    }
    ASSERT(panicked);
    // universal_setup_for_staging (src/proof/mod.rs:121-141): the bn254 branch is load_srs, its rng is not touched
    {
      int untouched_rng = 7;
      auto staged = proof::universal_setup_for_staging(max_degree, untouched_rng, crs, digest);
      ASSERT(staged.is_ok() && untouched_rng == 7);
      if (staged.is_ok()) ASSERT(staged.unwrap().max_degree == max_degree && staged.unwrap().h == universal_param.h);
      auto staged_big = proof::universal_setup_for_staging(((size_t)1 << 17) + 1, untouched_rng, crs, digest);
      ASSERT(staged_big.is_err());
    }
    auto loaded = proof::load_srs(max_degree, crs, digest);
    ASSERT(loaded.is_ok());
    if (loaded.is_ok()) {
      ASSERT(loaded.unwrap().max_degree == max_degree);
      ASSERT(loaded.unwrap().h == universal_param.h && loaded.unwrap().beta_h == universal_param.beta_h);
      // a key preprocessed under the loaded parameters makes the same proof
      auto pre = proof::transfer::preprocess(loaded.unwrap(), 2, 2, 26, in1.circuit());
      ASSERT(pre.is_ok());
      if (pre.is_ok()) {
        FixtureRng rng{in1.blinders};
        auto p = proof::transfer::prove(rng, pre.unwrap().proving_key, in1.assignment(), recv_memos_ver_key,
                                        extra_proof_bound_data);
        ASSERT(p.is_ok() && std::memcmp(&p.unwrap(), &validity_proof_1.unwrap(), sizeof(Proof)) == 0);
      }
    }
  }

  // universal_setup with hiding powers (KZG10::setup samples gamma beside tau): the stored parameters carry
  // powers_of_gamma_g for degrees 0 ..= max_degree + 1, the powers of g are those of the plain setup
  {
    capgpu::Fr gamma = in1.tau;
    gamma[0] ^= 0x5a5a5a5a;  // any other canonical scalar
    gamma[3] &= 0x0fffffffffffffffull;
    auto hiding = proof::universal_setup(max_degree, in1.tau, &gamma);
    ASSERT(hiding.is_ok());
    if (hiding.is_ok()) {
      size_t plain_len = 0, hiding_len = 0;
      ASSERT(capgpu_srs_serialize(universal_param.handle(), universal_param.h.data(), universal_param.beta_h.data(), nullptr,
                                  0, &plain_len) == CAPGPU_OK);
      ASSERT(capgpu_srs_serialize(hiding.unwrap().handle(), hiding.unwrap().h.data(), hiding.unwrap().beta_h.data(), nullptr,
                                  0, &hiding_len) == CAPGPU_OK);
      ASSERT(hiding_len == plain_len + (max_degree + 2) * (8 + 32));  // BTreeMap<usize, G1>: key + compressed point
      std::vector<uint8_t> a(plain_len), b(hiding_len);
      ASSERT(capgpu_srs_serialize(universal_param.handle(), universal_param.h.data(), universal_param.beta_h.data(), a.data(),
                                  a.size(), &plain_len) == CAPGPU_OK);
      ASSERT(capgpu_srs_serialize(hiding.unwrap().handle(), hiding.unwrap().h.data(), hiding.unwrap().beta_h.data(), b.data(),
                                  b.size(), &hiding_len) == CAPGPU_OK);
      const size_t powers = 8 + 32 * (max_degree + 1);
      ASSERT(std::memcmp(a.data(), b.data(), powers) == 0);  // Vec<G1> powers_of_g
      uint64_t map_len = 0;
      std::memcpy(&map_len, b.data() + powers, 8);
      ASSERT(map_len == max_degree + 2);
    }
  }

  if (failures) return 1;
  std::printf("OK\n");
  return 0;
}
