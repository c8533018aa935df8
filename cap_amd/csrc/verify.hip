// Host-side TurboPlonk verifier: the acceptance predicate of the hot path's output.
//
// Replaces `jf_plonk::PlonkKzgSnark::verify::<SolidityTranscript>` as called by
// `proof::transfer::verify` (src/proof/transfer.rs:192-212), `proof::mint::verify` (src/proof/mint.rs:124-140)
// and `proof::freeze::verify` (src/proof/freeze.rs:162-178).  The reference verifies on the CPU in milliseconds;
// so does this (no device needed, no capgpu_init needed): challenges from the Keccak transcript, ~25 G1 scalar
// multiplications for the linearisation commitment, and one product of two pairings (pairing.hpp).
#include <string.h>

#include <algorithm>
#include <atomic>
#include <functional>
#include <thread>
#include <vector>

#include "../../include/capgpu.h"
#include "host_util.hpp"
#include "pairing.hpp"

namespace cap {
void set_error(const char* fmt, ...);

namespace {

using pairing::fq2;
using pairing::g2_affine;

g1_xyzz g1_smul(const g1_affine& p, const fe& k_mont) {
  fe k = Fr::from_mont(k_mont);
  g1_xyzz acc = G1::inf();
  bool started = false;
  for (int i = 7; i >= 0; i--)
    for (int b = 31; b >= 0; b--) {
      if (started) acc = G1::dbl(acc);
      if ((k.v[i] >> b) & 1) {
        acc = G1::add_mixed(acc, p);
        started = true;
      }
    }
  return acc;
}
g1_affine g1_from_words(const uint64_t w[8]) {
  g1_affine p;
  p.x = fe_from_words(w);
  p.y = fe_from_words(w + 4);
  return p;
}
g2_affine g2_from_words(const uint64_t w[16]) {
  g2_affine q;
  q.x = {fe_from_words(w), fe_from_words(w + 4)};
  q.y = {fe_from_words(w + 8), fe_from_words(w + 12)};
  q.inf = pairing::f2_is_zero(q.x) && pairing::f2_is_zero(q.y);
  return q;
}
void g2_to_words(const g2_affine& q, uint64_t w[16]) {
  if (q.inf) {
    memset(w, 0, 16 * sizeof(uint64_t));
    return;
  }
  fe_to_words(q.x.c0, w);
  fe_to_words(q.x.c1, w + 4);
  fe_to_words(q.y.c0, w + 8);
  fe_to_words(q.y.c1, w + 12);
}
// arkworks' Fp256 values are canonical by construction (deserialisation rejects anything >= the modulus), so v and
// v + r can never both be a valid public input or evaluation there.  The raw Montgomery words of this ABI could carry
// such a second encoding - Fq::mul / Fr::mul reduce silently - which would make proofs malleable: every field word is
// range-checked before any arithmetic touches it.
bool g1_canonical(const g1_affine& p) { return !Fq::geq_mod(p.x) && !Fq::geq_mod(p.y); }
bool g1_on_curve(const g1_affine& p) {
  if (!g1_canonical(p)) return false;
  if (G1::is_inf(p)) return true;
  fe b3 = Fq::add(Fq::add(Fq::one(), Fq::one()), Fq::one());
  return Fq::eq(Fq::sqr(p.y), Fq::add(Fq::mul(Fq::sqr(p.x), p.x), b3));
}
fe fr_pow_u64(const fe& a, uint64_t e) { return Fr::pow_u64(a, e); }

}  // namespace
}  // namespace cap

using namespace cap;

extern "C" {

int capgpu_g2_generator(uint64_t out[16]) {
  if (!out) return CAPGPU_ERR_INVALID_ARG;
  g2_to_words(pairing::g2_generator(), out);
  return CAPGPU_OK;
}

int capgpu_g2_mul(const uint64_t q[16], const uint64_t scalar[4], uint64_t out[16]) {
  if (!q || !scalar || !out) return CAPGPU_ERR_INVALID_ARG;
  g2_affine p = g2_from_words(q);
  if (!pairing::g2_on_curve(p)) {
    set_error("capgpu_g2_mul: point is not on the twist curve");
    return CAPGPU_ERR_INVALID_ARG;
  }
  g2_to_words(pairing::g2_mul(p, fe_from_words(scalar)), out);
  return CAPGPU_OK;
}

int capgpu_pairing_check(const uint64_t* g1_points, const uint64_t* g2_points, size_t n, int* ok_out) {
  if ((n && (!g1_points || !g2_points)) || !ok_out) return CAPGPU_ERR_INVALID_ARG;
  std::vector<std::pair<g1_affine, g2_affine>> pairs;
  for (size_t i = 0; i < n; i++) {
    g1_affine p = g1_from_words(g1_points + 8 * i);
    g2_affine q = g2_from_words(g2_points + 16 * i);
    if (!g1_on_curve(p) || !pairing::g2_on_curve(q)) {
      set_error("capgpu_pairing_check: input %zu is not on its curve", i);
      return CAPGPU_ERR_INVALID_ARG;
    }
    pairs.emplace_back(p, q);
  }
  *ok_out = pairing::pairing_product_is_one(pairs) ? 1 : 0;
  return CAPGPU_OK;
}

// The pairing inputs of one proof as linear combinations of group elements: A = sum a[k].s * a[k].p, likewise B, and the
// proof holds iff e(A, [tau]H) == e(B, H).  Kept as (point, scalar) terms so that the caller chooses how to evaluate
// them: the single verifier and the host batch verifier multiply on the CPU, capgpu_plonk_batch_verify_dev hands the
// terms of the whole batch to the device MSM (K3-K6).  `u` is the last transcript challenge: it depends on every byte
// of the statement and the proof.
struct Term {
  g1_affine p;
  fe s;  // Montgomery
};
struct ProofTerms {
  std::vector<Term> a, b;
  fe u;
};

// Everything of the verifier up to the group arithmetic: on success *valid = 1 and `out` holds the terms; *valid = 0
// means the proof is already known to be invalid.
static int verifier_terms(const capgpu_verifying_key* vk, const uint64_t* pub_inputs, size_t num_inputs,
                          const capgpu_proof* proof, const uint8_t* ext_msg, size_t ext_msg_len, ProofTerms* out,
                          int* valid) {
  const uint64_t n = vk->domain_size;
  if (n < 4 || (n & (n - 1)) || num_inputs != vk->num_inputs) {
    set_error("capgpu_plonk_verify: %zu public inputs given, key expects %llu (domain %llu)", num_inputs,
              (unsigned long long)vk->num_inputs, (unsigned long long)n);
    return CAPGPU_ERR_INVALID_ARG;
  }
  uint32_t log_n = 0;
  while ((1ull << log_n) < n) log_n++;

  // ---- gather the group elements; malformed points make the proof invalid, not the call ------------------
  g1_affine sel[kNumSelectors], sig[kNumWires], wc[kNumWires], tq[kNumWires];
  for (int i = 0; i < kNumSelectors; i++) sel[i] = g1_from_words(vk->selector_comms[i]);
  for (int i = 0; i < kNumWires; i++) {
    sig[i] = g1_from_words(vk->sigma_comms[i]);
    wc[i] = g1_from_words(proof->wires_poly_comms[i]);
    tq[i] = g1_from_words(proof->split_quot_poly_comms[i]);
  }
  g1_affine zc = g1_from_words(proof->prod_perm_poly_comm);
  g1_affine w_zeta = g1_from_words(proof->opening_proof), w_zeta_w = g1_from_words(proof->shifted_opening_proof);
  {
    // proof points: on the curve and canonical (x, y < p); key points likewise (a key is trusted input, but a
    // non-canonical coordinate in it would hash differently from the same key written canonically)
    bool on = g1_on_curve(zc) && g1_on_curve(w_zeta) && g1_on_curve(w_zeta_w);
    for (int i = 0; i < kNumWires; i++) on = on && g1_on_curve(wc[i]) && g1_on_curve(tq[i]) && g1_on_curve(sig[i]);
    for (int i = 0; i < kNumSelectors; i++) on = on && g1_on_curve(sel[i]);
    if (!on) return CAPGPU_OK;
  }
  fe we[kNumWires], se[kNumWires - 1];
  for (int i = 0; i < kNumWires; i++) we[i] = fe_from_words(proof->wires_evals[i]);
  for (int i = 0; i < kNumWires - 1; i++) se[i] = fe_from_words(proof->wire_sigma_evals[i]);
  const fe znext = fe_from_words(proof->perm_next_eval);
  fe kk[kNumWires];
  for (int i = 0; i < kNumWires; i++) kk[i] = fe_from_words(vk->k[i]);
  {
    // every Fr word canonical: the 10 evaluations, the public inputs (nullifiers, Merkle root, ... - a ledger that
    // de-duplicates on raw words must not see two accepted encodings of one value) and the key's coset constants
    bool canon = !Fr::geq_mod(znext);
    for (int i = 0; i < kNumWires; i++) canon = canon && !Fr::geq_mod(we[i]) && !Fr::geq_mod(kk[i]);
    for (int i = 0; i < kNumWires - 1; i++) canon = canon && !Fr::geq_mod(se[i]);
    for (size_t i = 0; i < num_inputs; i++) canon = canon && !Fr::geq_mod(fe_from_words(pub_inputs + 4 * i));
    if (!canon) return CAPGPU_OK;
  }

  // ---- challenges (same transcript as the prover) ------------------------------------------------------------
  SolidityTranscript t;
  if (ext_msg && ext_msg_len) t.append(ext_msg, ext_msg_len);
  t.append_u64_le(254);
  t.append_u64_le(n);
  t.append_u64_le((uint64_t)num_inputs);
  for (int i = 0; i < kNumWires; i++) append_fr(t, kk[i]);
  for (int i = 0; i < kNumSelectors; i++) append_g1(t, sel[i]);
  for (int i = 0; i < kNumWires; i++) append_g1(t, sig[i]);
  std::vector<fe> pub(num_inputs);
  for (size_t i = 0; i < num_inputs; i++) {
    pub[i] = fe_from_words(pub_inputs + 4 * i);
    append_fr(t, pub[i]);
  }
  for (int i = 0; i < kNumWires; i++) append_g1(t, wc[i]);
  (void)get_challenge(t);  // plookup tau
  const fe beta = get_challenge(t), gamma = get_challenge(t);
  append_g1(t, zc);
  const fe alpha = get_challenge(t), alpha2 = Fr::sqr(alpha);
  for (int i = 0; i < kNumWires; i++) append_g1(t, tq[i]);
  const fe zeta = get_challenge(t);
  for (int i = 0; i < kNumWires; i++) append_fr(t, we[i]);
  for (int i = 0; i < kNumWires - 1; i++) append_fr(t, se[i]);
  append_fr(t, znext);
  const fe v = get_challenge(t);
  append_g1(t, w_zeta);
  append_g1(t, w_zeta_w);
  const fe u = get_challenge(t);

  // ---- scalars ---------------------------------------------------------------------------------------------------
  const fe one = Fr::one();
  const fe zh = Fr::sub(fr_pow_u64(zeta, n), one);
  if (Fr::is_zero(zh) || Fr::eq(zeta, one)) return CAPGPU_OK;  // zeta in the domain: reject
  const fe n_m = fr_from_u64(n);
  const fe l1 = Fr::mul(zh, Fr::inv(Fr::mul(n_m, Fr::sub(zeta, one))));
  // omega_n = omega_28^(2^(28-log n))
  fe omega;
  {
    const uint32_t root28[8] = {0x725b19f0u, 0x9bd61b6eu, 0x41112ed4u, 0x402d111eu,
                                0x8ef62abcu, 0x00e0a7ebu, 0xa58a7e85u, 0x2a3c09f0u};
    fe w;
    for (int i = 0; i < 8; i++) w.v[i] = root28[i];
    w = Fr::to_mont(w);
    for (uint32_t i = log_n; i < 28; i++) w = Fr::sqr(w);
    omega = w;
  }
  fe pi = Fr::zero();
  {
    fe x = one;
    for (size_t i = 0; i < num_inputs; i++) {
      fe li = Fr::mul(Fr::mul(zh, x), Fr::inv(Fr::mul(n_m, Fr::sub(zeta, x))));
      pi = Fr::add(pi, Fr::mul(pub[i], li));
      x = Fr::mul(x, omega);
    }
  }
  // r0 = PI(zeta) - alpha^2 L1(zeta) - alpha z(zeta w) (w4 + gamma) prod_{i<4} (w_i + beta sigma_i + gamma)
  fe tt = Fr::mul(Fr::mul(alpha, znext), Fr::add(we[4], gamma));
  for (int j = 0; j < kNumWires - 1; j++) tt = Fr::mul(tt, Fr::add(Fr::add(we[j], gamma), Fr::mul(beta, se[j])));
  const fe r0 = Fr::sub(Fr::sub(pi, Fr::mul(alpha2, l1)), tt);

  // ---- D: commitment of the linearisation polynomial -----------------------------------------------------------
  out->a.clear();
  out->b.clear();
  out->b.reserve(34);
  auto add_term = [&](const g1_affine& c, const fe& s) { out->b.push_back(Term{c, s}); };
  for (int j = 0; j < 4; j++) add_term(sel[j], we[j]);
  const fe w01 = Fr::mul(we[0], we[1]), w23 = Fr::mul(we[2], we[3]);
  add_term(sel[4], w01);
  add_term(sel[5], w23);
  for (int j = 0; j < 4; j++) {
    fe w2 = Fr::sqr(we[j]);
    add_term(sel[6 + j], Fr::mul(Fr::sqr(w2), we[j]));
  }
  add_term(sel[10], Fr::neg(we[4]));
  add_term(sel[11], one);
  add_term(sel[12], Fr::mul(Fr::mul(w01, w23), we[4]));
  fe bz = Fr::mul(beta, zeta), cz = alpha;
  for (int j = 0; j < kNumWires; j++) cz = Fr::mul(cz, Fr::add(Fr::add(we[j], gamma), Fr::mul(kk[j], bz)));
  cz = Fr::add(cz, Fr::mul(alpha2, l1));
  add_term(zc, cz);
  fe cs = Fr::mul(Fr::mul(alpha, beta), znext);
  for (int j = 0; j < kNumWires - 1; j++) cs = Fr::mul(cs, Fr::add(Fr::add(we[j], gamma), Fr::mul(beta, se[j])));
  add_term(sig[kNumWires - 1], Fr::neg(cs));
  fe zp = fr_pow_u64(zeta, n + 2), cq = Fr::neg(zh);
  for (int j = 0; j < kNumWires; j++) {
    add_term(tq[j], cq);
    cq = Fr::mul(cq, zp);
  }
  // ---- batch the openings: F = D + sum v^j [p_j] + u [z],  E = -r0 + sum v^j p_j(zeta) + u z(zeta w) -------------
  fe e_acc = Fr::neg(r0), cf = v;
  for (int j = 0; j < kNumWires; j++) {
    add_term(wc[j], cf);
    e_acc = Fr::add(e_acc, Fr::mul(cf, we[j]));
    cf = Fr::mul(cf, v);
  }
  for (int j = 0; j < kNumWires - 1; j++) {
    add_term(sig[j], cf);
    e_acc = Fr::add(e_acc, Fr::mul(cf, se[j]));
    cf = Fr::mul(cf, v);
  }
  add_term(zc, u);
  e_acc = Fr::add(e_acc, Fr::mul(u, znext));
  // A = W_zeta + u W_zetaw ;  B = zeta W_zeta + u zeta omega W_zetaw + F - E G
  out->a.push_back(Term{w_zeta, one});
  out->a.push_back(Term{w_zeta_w, u});
  g1_affine gen;
  gen.x = Fq::one();
  gen.y = Fq::dbl(Fq::one());
  add_term(w_zeta, zeta);
  add_term(w_zeta_w, Fr::mul(Fr::mul(u, zeta), omega));
  add_term(gen, Fr::neg(e_acc));
  out->u = u;
  *valid = 1;
  return CAPGPU_OK;
}

// sum weight * t.s * t.p on the host
static g1_xyzz eval_terms_host(const std::vector<Term>& terms, const fe* weight) {
  g1_xyzz acc = G1::inf();
  const fe one = Fr::one();
  for (const Term& t : terms) {
    const fe sc = weight ? Fr::mul(*weight, t.s) : t.s;
    if (Fr::eq(sc, one)) acc = G1::add_mixed(acc, t.p);
    else acc = G1::add(acc, g1_smul(t.p, sc));
  }
  return acc;
}

// Weights of the random linear combination of a batch: r_0 = 1, r_i = challenges of a transcript over every proof's
// last challenge u_i (each of which already binds its statement and proof).
static std::vector<fe> batch_weights(const std::vector<ProofTerms>& pt) {
  SolidityTranscript seed;
  for (const ProofTerms& t : pt) append_fr(seed, t.u);
  std::vector<fe> rs(pt.size());
  for (size_t i = 0; i < pt.size(); i++) rs[i] = i == 0 ? Fr::one() : get_challenge(seed);
  return rs;
}

// terms of every proof of a batch, in parallel on host threads; returns a negative code for malformed arguments,
// *all_valid = 0 when some proof is already known to be invalid
static int batch_terms(const capgpu_verifying_key* const* vks, const uint64_t* const* pub_inputs,
                       const size_t* num_inputs, const capgpu_proof* const* proofs, const uint8_t* const* ext_msgs,
                       const size_t* ext_msg_lens, size_t count, std::vector<ProofTerms>* out, int* all_valid) {
  out->assign(count, ProofTerms{});
  std::vector<int> rcs(count, CAPGPU_OK), valids(count, 0);
  for (size_t i = 0; i < count; i++)
    if (!vks[i] || !proofs[i] || (num_inputs[i] && !pub_inputs[i])) return CAPGPU_ERR_INVALID_ARG;
  auto prepare = [&](size_t i) {
    rcs[i] = verifier_terms(vks[i], pub_inputs[i], num_inputs[i], proofs[i], ext_msgs ? ext_msgs[i] : nullptr,
                            (ext_msgs && ext_msg_lens) ? ext_msg_lens[i] : 0, &(*out)[i], &valids[i]);
  };
  const unsigned nt = (unsigned)std::min<size_t>(count, std::max(1u, std::min(std::thread::hardware_concurrency(), 32u)));
  if (nt <= 1) {
    for (size_t i = 0; i < count; i++) prepare(i);
  } else {
    std::atomic<size_t> next{0};
    std::vector<std::thread> th;
    auto work = [&] {
      for (;;) {
        size_t i = next.fetch_add(1);
        if (i >= count) break;
        prepare(i);
      }
    };
    for (unsigned t = 1; t < nt; t++) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
  }
  *all_valid = 1;
  for (size_t i = 0; i < count; i++) {
    if (rcs[i]) {  // the message was recorded on a worker thread: redo the failing one here for this thread's string
      int v = 0;
      ProofTerms scratch;
      return verifier_terms(vks[i], pub_inputs[i], num_inputs[i], proofs[i], ext_msgs ? ext_msgs[i] : nullptr,
                            (ext_msgs && ext_msg_lens) ? ext_msg_lens[i] : 0, &scratch, &v);
    }
    if (!valids[i]) *all_valid = 0;
  }
  return CAPGPU_OK;
}

static int load_open_key(const uint64_t g2_h[16], const uint64_t g2_beta_h[16], g2_affine* h, g2_affine* beta_h) {
  *h = g2_from_words(g2_h);
  *beta_h = g2_from_words(g2_beta_h);
  if (!pairing::g2_on_curve(*h) || !pairing::g2_on_curve(*beta_h) || h->inf || beta_h->inf) {
    set_error("capgpu_plonk_verify: open key G2 elements are not on the twist curve");
    return CAPGPU_ERR_INVALID_ARG;
  }
  return CAPGPU_OK;
}

int capgpu_plonk_verify(const capgpu_verifying_key* vk, const uint64_t g2_h[16], const uint64_t g2_beta_h[16],
                        const uint64_t* pub_inputs, size_t num_inputs, const capgpu_proof* proof,
                        const uint8_t* ext_msg, size_t ext_msg_len, int* ok_out) {
  if (!vk || !g2_h || !g2_beta_h || !proof || !ok_out || (num_inputs && !pub_inputs)) {
    set_error("capgpu_plonk_verify: bad argument");
    return CAPGPU_ERR_INVALID_ARG;
  }
  *ok_out = 0;
  g2_affine h, beta_h;
  int rc = load_open_key(g2_h, g2_beta_h, &h, &beta_h);
  if (rc) return rc;
  ProofTerms pt;
  int valid = 0;
  rc = verifier_terms(vk, pub_inputs, num_inputs, proof, ext_msg, ext_msg_len, &pt, &valid);
  if (rc || !valid) return rc;
  g1_affine a = G1::to_affine(eval_terms_host(pt.a, nullptr)), b = G1::to_affine(eval_terms_host(pt.b, nullptr));
  // e(A, [tau]H) == e(B, H)   <=>   e(A, [tau]H) e(-B, H) == 1
  b.y = Fq::neg(b.y);
  std::vector<std::pair<g1_affine, g2_affine>> pairs = {{a, beta_h}, {b, h}};
  *ok_out = pairing::pairing_product_is_one(pairs) ? 1 : 0;
  return CAPGPU_OK;
}

// Replaces PlonkKzgSnark::batch_verify as used by txn_batch_verify (src/lib.rs:455-529, call at :517-522): the
// per-proof pairing inputs (A_i, B_i) are folded with pseudo-random weights r_i and one pairing product decides the
// whole batch.  Proofs may belong to different circuits / keys of one SRS.  Host only: the ~35 scalar multiplications of
// each proof go to a pool of host threads.
int capgpu_plonk_batch_verify(const capgpu_verifying_key* const* vks, const uint64_t g2_h[16],
                              const uint64_t g2_beta_h[16], const uint64_t* const* pub_inputs,
                              const size_t* num_inputs, const capgpu_proof* const* proofs,
                              const uint8_t* const* ext_msgs, const size_t* ext_msg_lens, size_t count, int* ok_out) {
  if (!ok_out || !g2_h || !g2_beta_h || (count && (!vks || !pub_inputs || !num_inputs || !proofs))) {
    set_error("capgpu_plonk_batch_verify: bad argument");
    return CAPGPU_ERR_INVALID_ARG;
  }
  *ok_out = 0;
  g2_affine h, beta_h;
  int rc = load_open_key(g2_h, g2_beta_h, &h, &beta_h);
  if (rc) return rc;
  if (count == 0) {
    *ok_out = 1;
    return CAPGPU_OK;
  }
  std::vector<ProofTerms> pt;
  int all_valid = 0;
  rc = batch_terms(vks, pub_inputs, num_inputs, proofs, ext_msgs, ext_msg_lens, count, &pt, &all_valid);
  if (rc || !all_valid) return rc;
  const std::vector<fe> rs = batch_weights(pt);
  std::vector<g1_xyzz> ra(count), rb(count);
  {
    const unsigned nt = (unsigned)std::min<size_t>(count, std::max(1u, std::min(std::thread::hardware_concurrency(), 32u)));
    std::atomic<size_t> next{0};
    auto work = [&] {
      for (;;) {
        size_t i = next.fetch_add(1);
        if (i >= count) break;
        ra[i] = eval_terms_host(pt[i].a, &rs[i]);
        rb[i] = eval_terms_host(pt[i].b, &rs[i]);
      }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; t++) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
  }
  g1_xyzz a_sum = G1::inf(), b_sum = G1::inf();
  for (size_t i = 0; i < count; i++) {
    a_sum = G1::add(a_sum, ra[i]);
    b_sum = G1::add(b_sum, rb[i]);
  }
  g1_affine a = G1::to_affine(a_sum), b = G1::to_affine(b_sum);
  b.y = Fq::neg(b.y);
  std::vector<std::pair<g1_affine, g2_affine>> pairs = {{a, beta_h}, {b, h}};
  *ok_out = pairing::pairing_product_is_one(pairs) ? 1 : 0;
  return CAPGPU_OK;
}

// The same predicate with the group arithmetic on the device: the (point, scalar) terms of ALL proofs - 35 per proof,
// the weight r_i folded into the scalars - are two multi-scalar multiplications, run by the MSM kernels of the prover
// (K3-K6; the bases are uploaded like an SRS, their window tables built on the device).  SURVEY 8f row 4.  Needs an
// initialised library (CAPGPU_ERR_NOT_INITIALISED otherwise - there is no silent host path behind this entry point);
// the transcript work per proof and the final pairing product stay on the host.  Accepts and rejects exactly what
// capgpu_plonk_batch_verify does: both evaluate the same terms with the same weights.
int capgpu_plonk_batch_verify_dev(const capgpu_verifying_key* const* vks, const uint64_t g2_h[16],
                                  const uint64_t g2_beta_h[16], const uint64_t* const* pub_inputs,
                                  const size_t* num_inputs, const capgpu_proof* const* proofs,
                                  const uint8_t* const* ext_msgs, const size_t* ext_msg_lens, size_t count,
                                  int* ok_out) {
  if (!ok_out || !g2_h || !g2_beta_h || (count && (!vks || !pub_inputs || !num_inputs || !proofs))) {
    set_error("capgpu_plonk_batch_verify_dev: bad argument");
    return CAPGPU_ERR_INVALID_ARG;
  }
  *ok_out = 0;
  int rc = capgpu_device_info(nullptr, nullptr, nullptr);  // fails loudly without an initialised device
  if (rc) return rc;
  g2_affine h, beta_h;
  rc = load_open_key(g2_h, g2_beta_h, &h, &beta_h);
  if (rc) return rc;
  if (count == 0) {
    *ok_out = 1;
    return CAPGPU_OK;
  }
  std::vector<ProofTerms> pt;
  int all_valid = 0;
  rc = batch_terms(vks, pub_inputs, num_inputs, proofs, ext_msgs, ext_msg_lens, count, &pt, &all_valid);
  if (rc || !all_valid) return rc;
  const std::vector<fe> rs = batch_weights(pt);
  // bases: all A terms, then all B terms; scalars canonical (what the MSM entry point takes)
  size_t na = 0, nb = 0;
  for (const ProofTerms& t : pt) {
    na += t.a.size();
    nb += t.b.size();
  }
  std::vector<uint64_t> bases(8 * (na + nb)), scalars(4 * (na + nb));
  size_t ia = 0, ib = na;
  for (size_t i = 0; i < count; i++) {
    auto put = [&](size_t at, const Term& t) {
      fe_to_words(t.p.x, &bases[8 * at]);
      fe_to_words(t.p.y, &bases[8 * at + 4]);
      fe_to_words(Fr::from_mont(Fr::mul(rs[i], t.s)), &scalars[4 * at]);
    };
    for (const Term& t : pt[i].a) put(ia++, t);
    for (const Term& t : pt[i].b) put(ib++, t);
  }
  uint64_t hbases = 0;
  rc = capgpu_srs_upload(bases.data(), na + nb, 64, 1, &hbases);
  if (rc) return rc;
  uint64_t a_xyz[12], b_xyz[12];
  rc = capgpu_msm_g1(hbases, 0, scalars.data(), na, a_xyz);
  if (rc == CAPGPU_OK) rc = capgpu_msm_g1(hbases, na, scalars.data() + 4 * na, nb, b_xyz);
  capgpu_srs_free(hbases);
  if (rc) return rc;
  auto to_affine = [](const uint64_t w[12]) {
    g1_jac j;
    j.x = fe_from_words(w);
    j.y = fe_from_words(w + 4);
    j.z = fe_from_words(w + 8);
    return G1::jac_to_affine(j);
  };
  g1_affine a = to_affine(a_xyz), b = to_affine(b_xyz);
  b.y = Fq::neg(b.y);
  std::vector<std::pair<g1_affine, g2_affine>> pairs = {{a, beta_h}, {b, h}};
  *ok_out = pairing::pairing_product_is_one(pairs) ? 1 : 0;
  return CAPGPU_OK;
}

// ark-serialize 0.3 CanonicalSerialize of jf_plonk's Proof (what sits inside TransferNote.proof, src/transfer.rs:60):
// 13 compressed G1 (5 wires, prod_perm, 5 split quotient, opening, shifted opening), each Vec with a u64 LE length
// prefix, then the 10 evaluations (wires_evals Vec, wire_sigma_evals Vec, perm_next_eval), then plookup_proof: None.
// Layout as recalled: unverified against jellyfish.  Returns the number of bytes written.
int capgpu_proof_serialize(const capgpu_proof* proof, uint8_t* out, size_t cap, size_t* len_out) {
  if (!proof || !out || !len_out) return CAPGPU_ERR_INVALID_ARG;
  std::vector<uint8_t> buf;
  auto put_u64 = [&](uint64_t v) { buf.insert(buf.end(), (uint8_t*)&v, (uint8_t*)&v + 8); };
  auto put_g1 = [&](const uint64_t w[8]) {
    uint8_t b[32];
    serialize_g1(g1_from_words(w), b);
    buf.insert(buf.end(), b, b + 32);
  };
  auto put_fr = [&](const uint64_t w[4]) {
    uint8_t b[32];
    serialize_fr(fe_from_words(w), b);
    buf.insert(buf.end(), b, b + 32);
  };
  put_u64(kNumWires);
  for (int i = 0; i < kNumWires; i++) put_g1(proof->wires_poly_comms[i]);
  put_g1(proof->prod_perm_poly_comm);
  put_u64(kNumWires);
  for (int i = 0; i < kNumWires; i++) put_g1(proof->split_quot_poly_comms[i]);
  put_g1(proof->opening_proof);
  put_g1(proof->shifted_opening_proof);
  put_u64(kNumWires);
  for (int i = 0; i < kNumWires; i++) put_fr(proof->wires_evals[i]);
  put_u64(kNumWires - 1);
  for (int i = 0; i < kNumWires - 1; i++) put_fr(proof->wire_sigma_evals[i]);
  put_fr(proof->perm_next_eval);
  buf.push_back(0);  // Option::None for plookup_proof
  *len_out = buf.size();
  if (buf.size() > cap) {
    set_error("capgpu_proof_serialize: buffer too small (%zu needed)", buf.size());
    return CAPGPU_ERR_INVALID_ARG;
  }
  memcpy(out, buf.data(), buf.size());
  return CAPGPU_OK;
}

}  // extern "C"
