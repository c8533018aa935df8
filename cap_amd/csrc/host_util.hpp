// Host-side helpers shared by the prover (plonk.hip) and the verifier (verify.hip): word <-> field
// conversions, Jacobian -> affine with one batched inversion, ark-serialize 0.3 encodings and the
// challenge derivation of jf-plonk's SolidityTranscript.  O(1) work per commitment.
#pragma once
#include <string.h>

#include <vector>

#include "curve.hpp"
#include "keccak.hpp"

namespace cap {

constexpr int kNumWires = 5;
constexpr int kNumSelectors = 13;

// coset representatives k_i, canonical little-endian words
static const uint64_t K_CANON[kNumWires][4] = {
    {1, 0, 0, 0},
    {0x5da4fb7bb1301d4aULL, 0x73ca6c94813f8583ULL, 0xc4e12a44e110404cULL, 0x2f8dd1f1a7583c42ULL},
    {0x77f010424afeb025ULL, 0xa828a3703b311d0fULL, 0xeaa8fe837060498bULL, 0x1ee678a0470a75a6ULL},
    {0x66905a6895790c0aULL, 0x950b1db26d5c82d6ULL, 0x0a087c03e29c968bULL, 0x2042a587a90c187bULL},
    {0xeb9222db7c81e881ULL, 0x8f739da5d8d40dd3ULL, 0xdf57b799969dea1cULL, 0x2e2b91456103698aULL}};

inline fe fe_from_words(const uint64_t v[4]) {
  fe r;
  for (int i = 0; i < 4; i++) {
    r.v[2 * i] = (uint32_t)v[i];
    r.v[2 * i + 1] = (uint32_t)(v[i] >> 32);
  }
  return r;
}
inline void fe_to_words(const fe& a, uint64_t v[4]) {
  for (int i = 0; i < 4; i++) v[i] = (uint64_t)a.v[2 * i] | ((uint64_t)a.v[2 * i + 1] << 32);
}
inline fe fr_from_u64(uint64_t v) {
  fe t = Fr::zero();
  t.v[0] = (uint32_t)v;
  t.v[1] = (uint32_t)(v >> 32);
  return Fr::to_mont(t);
}

// ---- host-side group / serialisation helpers (O(1) per commitment) ------------------------------------
inline void batch_to_affine(const std::vector<g1_jac>& in, std::vector<g1_affine>& out) {
  size_t n = in.size();
  out.resize(n);
  std::vector<fe> pre(n);
  fe acc = Fq::one();
  for (size_t i = 0; i < n; i++) {
    pre[i] = acc;
    if (!Fq::is_zero(in[i].z)) acc = Fq::mul(acc, in[i].z);
  }
  fe inv = Fq::inv(acc);
  for (size_t i = n; i-- > 0;) {
    if (Fq::is_zero(in[i].z)) {
      out[i].x = Fq::zero();
      out[i].y = Fq::zero();
      continue;
    }
    fe zi = Fq::mul(inv, pre[i]);
    inv = Fq::mul(inv, in[i].z);
    fe zi2 = Fq::sqr(zi);
    out[i].x = Fq::mul(in[i].x, zi2);
    out[i].y = Fq::mul(in[i].y, Fq::mul(zi2, zi));
  }
}

// ark-serialize 0.3 compressed G1: x little-endian, bit 7 of the last byte = y is the larger root,
// bit 6 = infinity
inline void serialize_g1(const g1_affine& p, uint8_t out[32]) {
  if (G1::is_inf(p)) {
    memset(out, 0, 32);
    out[31] |= 0x40;
    return;
  }
  fe x = Fq::from_mont(p.x), y = Fq::from_mont(p.y), ny = Fq::from_mont(Fq::neg(p.y));
  memcpy(out, x.v, 32);
  bool larger = false;  // y > -y ?
  for (int i = 7; i >= 0; i--) {
    if (y.v[i] != ny.v[i]) {
      larger = y.v[i] > ny.v[i];
      break;
    }
  }
  if (larger) out[31] |= 0x80;
}
inline void serialize_fr(const fe& a_mont, uint8_t out[32]) {
  fe c = Fr::from_mont(a_mont);
  memcpy(out, c.v, 32);
}
// from_le_bytes_mod_order over the first 48 bytes
inline fe challenge_to_fr(const uint8_t h[64]) {
  fe lo, hi = Fr::zero();
  memcpy(lo.v, h, 32);
  memcpy(hi.v, h + 32, 16);
  fe lo_m = Fr::to_mont(lo);                      // works for any 256-bit input
  fe hi_m = Fr::mul(Fr::to_mont(hi), Fr::r2());   // * 2^256
  return Fr::add(lo_m, hi_m);
}
inline fe get_challenge(SolidityTranscript& t) {
  uint8_t h[64];
  t.challenge_bytes(h);
  return challenge_to_fr(h);
}
inline void append_g1(SolidityTranscript& t, const g1_affine& p) {
  uint8_t b[32];
  serialize_g1(p, b);
  t.append(b, 32);
}
inline void append_fr(SolidityTranscript& t, const fe& a) {
  uint8_t b[32];
  serialize_fr(a, b);
  t.append(b, 32);
}
inline void affine_to_words(const g1_affine& p, uint64_t out[8]) {
  fe_to_words(p.x, out);
  fe_to_words(p.y, out + 4);
}


}  // namespace cap
