/* C harness of the drop-in boundary (SURVEY.md 8b asks for one beside the Python one): a plain C caller of
 * include/capgpu.h, the way the Rust shim of INTEGRATION.md calls it - no Python, no ctypes.
 *
 *   cc -O1 -I include tools/c_harness/harness.c -L cap_amd -lcapgpu -Wl,-rpath,$PWD/cap_amd -o harness
 *   ./harness tests/golden/harness_log5.bin
 *
 * Input file (written by tests/golden/make_harness_input.py, little-endian u64 words, field elements in arkworks'
 * Montgomery form): magic "CAPH", log_n, num_inputs, ext_msg_len, tau[4] (canonical), selectors[13][n][4],
 * sigma[5][n][4], wires[5][n][4], pub_inputs[l][4], blinders[13][4], ext_msg bytes.
 *
 * It runs: init -> SRS of powers of tau -> NTT round trip (bit-exact) -> preprocess -> prove -> serialize -> verify
 * (accept) -> verify with a flipped public input (reject) -> a batch of two proofs equals two single proofs -> the same
 * key and proofs from COEFFICIENT-form columns (CAPGPU_INPUT_COEFFS, what a jf-relation caller holds) are the same bytes.
 * Output: "PROOF <hex of the 769 ark-serialize bytes>" and "OK"; tests/test_c_harness.py compares the bytes with the
 * golden proof.  Exit codes: 0 ok, 2 no usable GPU (capgpu_init failed: the library has no CPU fallback), 1 anything else. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "capgpu.h"

#define CHECK(call)                                                                      \
  do {                                                                                   \
    int rc_ = (call);                                                                    \
    if (rc_ != CAPGPU_OK) {                                                              \
      fprintf(stderr, "%s failed: %d (%s)\n", #call, rc_, capgpu_last_error());          \
      return 1;                                                                          \
    }                                                                                    \
  } while (0)

static uint64_t* read_words(FILE* f, size_t count) {
  uint64_t* p = (uint64_t*)malloc(8 * (count ? count : 1));
  if (!p || fread(p, 8, count, f) != count) {
    fprintf(stderr, "short input file\n");
    exit(1);
  }
  return p;
}

int main(int argc, char** argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s input.bin\n", argv[0]);
    return 1;
  }
  printf("%s\n", capgpu_version());
  int rc = capgpu_init(NULL, 0);
  if (rc != CAPGPU_OK) {
    fprintf(stderr, "capgpu_init failed: %d (%s)\n", rc, capgpu_last_error());
    return 2;
  }
  FILE* f = fopen(argv[1], "rb");
  if (!f) {
    perror(argv[1]);
    return 1;
  }
  uint64_t* hdr = read_words(f, 4);
  if (memcmp(hdr, "CAPH\0\0\0\0", 8) != 0) {
    fprintf(stderr, "bad magic\n");
    return 1;
  }
  const unsigned log_n = (unsigned)hdr[1];
  const size_t n = (size_t)1 << log_n, num_inputs = (size_t)hdr[2], msg_len = (size_t)hdr[3];
  uint64_t* tau = read_words(f, 4);
  uint64_t* selectors = read_words(f, 13 * n * 4);
  uint64_t* sigma = read_words(f, 5 * n * 4);
  uint64_t* wires = read_words(f, 5 * n * 4);
  uint64_t* pubs = read_words(f, num_inputs * 4);
  uint64_t* blinders = read_words(f, 13 * 4);
  uint8_t* msg = (uint8_t*)malloc(msg_len ? msg_len : 1);
  if (fread(msg, 1, msg_len, f) != msg_len) return 1;
  fclose(f);

  /* NTT round trip on the first wire column: forward coset transform, then its inverse, must give the bytes back */
  uint64_t* col = (uint64_t*)malloc(32 * n);
  memcpy(col, wires, 32 * n);
  CHECK(capgpu_ntt_fr(col, log_n, 0, 1));
  if (memcmp(col, wires, 32 * n) == 0) {
    fprintf(stderr, "NTT left the data unchanged\n");
    return 1;
  }
  CHECK(capgpu_ntt_fr(col, log_n, 1, 1));
  if (memcmp(col, wires, 32 * n) != 0) {
    fprintf(stderr, "NTT round trip differs\n");
    return 1;
  }

  uint64_t srs = 0, pk = 0;
  capgpu_verifying_key vk;
  CHECK(capgpu_srs_generate(tau, n + 3, &srs));
  CHECK(capgpu_plonk_preprocess(srs, n, num_inputs, selectors, sigma, &pk, &vk));
  size_t kn = 0, kin = 0;
  uint64_t ksrs = 0;
  CHECK(capgpu_plonk_key_info(pk, &kn, &kin, &ksrs));
  if (kn != n || kin != num_inputs || ksrs != srs) {
    fprintf(stderr, "key info mismatch\n");
    return 1;
  }
  capgpu_proof proof;
  CHECK(capgpu_plonk_prove(pk, wires, pubs, num_inputs, msg, msg_len, blinders, &proof));
  uint8_t bytes[1024];
  size_t len = 0;
  CHECK(capgpu_proof_serialize(&proof, bytes, sizeof(bytes), &len));
  printf("PROOF ");
  for (size_t i = 0; i < len; i++) printf("%02x", bytes[i]);
  printf("\n");

  /* verifier: open key of the synthetic SRS = (H, [tau] H) */
  uint64_t h[16], beta_h[16];
  CHECK(capgpu_g2_generator(h));
  CHECK(capgpu_g2_mul(h, tau, beta_h));
  int ok = 0;
  CHECK(capgpu_plonk_verify(&vk, h, beta_h, pubs, num_inputs, &proof, msg, msg_len, &ok));
  if (!ok) {
    fprintf(stderr, "verifier rejected a good proof\n");
    return 1;
  }
  if (num_inputs) {
    pubs[0] ^= 1;
    CHECK(capgpu_plonk_verify(&vk, h, beta_h, pubs, num_inputs, &proof, msg, msg_len, &ok));
    pubs[0] ^= 1;
    if (ok) {
      fprintf(stderr, "verifier accepted a wrong public input\n");
      return 1;
    }
  }
  /* wrong number of public inputs is an argument error, reported through the error string */
  if (capgpu_plonk_prove(pk, wires, pubs, num_inputs + 1, msg, msg_len, blinders, &proof) != CAPGPU_ERR_INVALID_ARG) {
    fprintf(stderr, "wrong public-input count was not refused\n");
    return 1;
  }

  /* batch of two identical inputs = the single proof twice */
  uint64_t* w2 = (uint64_t*)malloc(2 * 5 * n * 32);
  uint64_t* p2 = (uint64_t*)malloc(2 * (num_inputs ? num_inputs : 1) * 32);
  uint64_t b2[2 * 13 * 4];
  for (int k = 0; k < 2; k++) {
    memcpy(w2 + (size_t)k * 5 * n * 4, wires, 5 * n * 32);
    memcpy(p2 + (size_t)k * num_inputs * 4, pubs, num_inputs * 32);
    memcpy(b2 + k * 52, blinders, 13 * 32);
  }
  capgpu_proof two[2];
  CHECK(capgpu_plonk_prove_batch(pk, 2, w2, p2, num_inputs, msg, msg_len, b2, two));
  CHECK(capgpu_plonk_prove(pk, wires, pubs, num_inputs, msg, msg_len, blinders, &proof));
  if (memcmp(&two[0], &proof, sizeof(proof)) != 0 || memcmp(&two[1], &proof, sizeof(proof)) != 0) {
    fprintf(stderr, "batched proofs differ from the single proof\n");
    return 1;
  }

  /* The other input form.  A jf-relation caller holds polynomials in coefficient form; here they are made from the
   * columns with the library's own inverse transform.  Key and proofs must be the same bytes as from the columns. */
  {
    uint64_t* selc = (uint64_t*)malloc(13 * n * 32);
    uint64_t* sigc = (uint64_t*)malloc(5 * n * 32);
    uint64_t* wc = (uint64_t*)malloc(2 * 5 * n * 32);
    memcpy(selc, selectors, 13 * n * 32);
    memcpy(sigc, sigma, 5 * n * 32);
    memcpy(wc, wires, 5 * n * 32);
    for (int c = 0; c < 13; c++) CHECK(capgpu_ntt_fr(selc + (size_t)c * n * 4, log_n, 1, 0));
    for (int c = 0; c < 5; c++) CHECK(capgpu_ntt_fr(sigc + (size_t)c * n * 4, log_n, 1, 0));
    for (int c = 0; c < 5; c++) CHECK(capgpu_ntt_fr(wc + (size_t)c * n * 4, log_n, 1, 0));
    memcpy(wc + 5 * n * 4, wc, 5 * n * 32);
    uint64_t pk2 = 0;
    capgpu_verifying_key vk2;
    CHECK(capgpu_plonk_preprocess_ex(srs, n, num_inputs, selc, sigc, CAPGPU_INPUT_COEFFS, &pk2, &vk2));
    if (memcmp(&vk, &vk2, sizeof(vk)) != 0) {
      fprintf(stderr, "verifying key from coefficient-form columns differs\n");
      return 1;
    }
    capgpu_proof pc, pc2[2];
    CHECK(capgpu_plonk_prove_ex(pk2, wc, pubs, num_inputs, msg, msg_len, blinders, CAPGPU_INPUT_COEFFS, &pc));
    CHECK(capgpu_plonk_prove_batch_ex(pk, 2, wc, p2, num_inputs, msg, msg_len, b2, CAPGPU_INPUT_COEFFS, pc2));
    if (memcmp(&pc, &proof, sizeof(proof)) != 0 || memcmp(&pc2[0], &proof, sizeof(proof)) != 0 ||
        memcmp(&pc2[1], &proof, sizeof(proof)) != 0) {
      fprintf(stderr, "proofs from coefficient-form wires differ from the evaluation-form proof\n");
      return 1;
    }
    if (capgpu_plonk_prove_ex(pk, wires, pubs, num_inputs, msg, msg_len, blinders, 2, &pc) != CAPGPU_ERR_INVALID_ARG) {
      fprintf(stderr, "an unknown input form was not refused\n");
      return 1;
    }
    printf("COEFFS same bytes\n");
    CHECK(capgpu_plonk_free_key(pk2));
    free(selc);
    free(sigc);
    free(wc);
  }
  CHECK(capgpu_plonk_free_key(pk));
  CHECK(capgpu_srs_free(srs));
  capgpu_shutdown();
  printf("OK\n");
  return 0;
}
