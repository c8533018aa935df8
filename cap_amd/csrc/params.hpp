// ark-serialize 0.3 encodings of the reference's on-disk parameter types:
// byte reader/writer, compressed G1/G2 points, and the bulk G1 (de)compression that runs on the device.
// Reference: src/parameters.rs:560-577 (store_data / load_data write `CanonicalSerialize` bytes),
// src/proof/mod.rs:74-109 (load_srs deserialises a UniversalSrs blob).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/capgpu.h"
#include "host_util.hpp"
#include "pairing.hpp"

namespace cap {
namespace params {

// ---- byte cursor ---------------------------------------------------------------------------------------
struct Reader {
  const uint8_t* p;
  size_t len, pos = 0;
  bool ok = true;
  Reader(const uint8_t* data, size_t n) : p(data), len(n) {}
  const uint8_t* take(size_t n) {
    if (!ok || n > len - pos) {
      ok = false;
      return nullptr;
    }
    const uint8_t* r = p + pos;
    pos += n;
    return r;
  }
  uint64_t u64() {
    const uint8_t* b = take(8);
    uint64_t v = 0;
    if (b) memcpy(&v, b, 8);
    return v;
  }
  // a length prefix whose items (item_bytes each) must still fit in the input
  bool count(size_t item_bytes, uint64_t* out) {
    uint64_t c = u64();
    if (!ok || c > (len - pos) / item_bytes) return ok = false;
    *out = c;
    return true;
  }
};

struct Writer {
  std::vector<uint8_t> buf;
  void put(const void* src, size_t n) {
    const uint8_t* s = (const uint8_t*)src;
    buf.insert(buf.end(), s, s + n);
  }
  void u64(uint64_t v) { put(&v, 8); }
  void u8(uint8_t v) { buf.push_back(v); }
};

// ---- canonical comparisons -----------------------------------------------------------------------------
inline int cmp_words(const fe& a, const fe& b) {
  for (int i = 7; i >= 0; i--)
    if (a.v[i] != b.v[i]) return a.v[i] > b.v[i] ? 1 : -1;
  return 0;
}
inline fe fq_modulus() {
  fe m;
  for (int i = 0; i < 8; i++) m.v[i] = FqP::MOD[i];
  return m;
}
inline fe fr_modulus() {
  fe m;
  for (int i = 0; i < 8; i++) m.v[i] = FrP::MOD[i];
  return m;
}
// (p + 1) / 4: the square-root exponent for p = 3 mod 4
inline void fq_sqrt_exponent(uint32_t e[8]) {
  fe m = fq_modulus();
  uint64_t c = 1;
  for (int i = 0; i < 8; i++) {
    c += m.v[i];
    m.v[i] = (uint32_t)c;
    c >>= 32;
  }
  for (int i = 0; i < 8; i++) e[i] = (m.v[i] >> 2) | (i < 7 ? m.v[i + 1] << 30 : 0);
}
// candidate square root of a Montgomery Fq element; false if `a` is a non-residue
inline bool fq_sqrt(const fe& a, fe* out) {
  uint32_t e[8];
  fq_sqrt_exponent(e);
  fe r = Fq::pow(a, e);
  *out = r;
  return Fq::eq(Fq::sqr(r), a);
}

// ---- G1, one point on the host (verifying keys: 20 points) ----------------------------------------------
// Same predicate as the device kernel in params.hip; returns false on any encoding ark-serialize rejects.
inline bool g1_decompress_host(const uint8_t in[32], g1_affine* out) {
  fe x;
  memcpy(x.v, in, 32);
  uint32_t flags = x.v[7] >> 30;
  x.v[7] &= 0x3fffffffu;
  out->x = out->y = Fq::zero();
  if (flags == 3) return false;
  if (flags == 1) {
    for (int i = 0; i < 8; i++)
      if (x.v[i]) return false;
    return true;
  }
  if (cmp_words(x, fq_modulus()) >= 0) return false;
  fe xm = Fq::to_mont(x);
  fe three = Fq::zero();
  three.v[0] = 3;
  fe rhs = Fq::add(Fq::mul(Fq::sqr(xm), xm), Fq::to_mont(three));
  fe y;
  if (!fq_sqrt(rhs, &y)) return false;
  fe yc = Fq::from_mont(y), nyc = Fq::from_mont(Fq::neg(y));
  bool larger = cmp_words(yc, nyc) > 0;
  out->x = xm;
  out->y = (larger == (flags == 2)) ? y : Fq::neg(y);
  return true;
}

// ---- G2 (host only: a UniversalSrs holds two of them) -----------------------------------------------------
using pairing::fq2;
using pairing::g2_affine;

// ark-ff orders Fq2 by c1 first, then c0 (canonical integers)
inline bool f2_greater(const fq2& a, const fq2& b) {
  int c = cmp_words(Fq::from_mont(a.c1), Fq::from_mont(b.c1));
  if (c) return c > 0;
  return cmp_words(Fq::from_mont(a.c0), Fq::from_mont(b.c0)) > 0;
}
inline void g2_compress(const g2_affine& q, uint8_t out[64]) {
  memset(out, 0, 64);
  if (q.inf) {
    out[63] |= 0x40;
    return;
  }
  fe c0 = Fq::from_mont(q.x.c0), c1 = Fq::from_mont(q.x.c1);
  memcpy(out, c0.v, 32);
  memcpy(out + 32, c1.v, 32);
  if (f2_greater(q.y, pairing::f2_neg(q.y))) out[63] |= 0x80;
}
// square root in Fq[u]/(u^2 + 1) by the complex method
inline bool f2_sqrt(const fq2& a, fq2* out) {
  if (Fq::is_zero(a.c1)) {
    fe r;
    if (fq_sqrt(a.c0, &r)) {
      *out = {r, Fq::zero()};
      return true;
    }
    if (!fq_sqrt(Fq::neg(a.c0), &r)) return false;
    *out = {Fq::zero(), r};
    return true;
  }
  fe s;
  if (!fq_sqrt(Fq::add(Fq::sqr(a.c0), Fq::sqr(a.c1)), &s)) return false;
  fe two = Fq::add(Fq::one(), Fq::one());
  fe inv2 = Fq::inv(two);
  fe x0;
  if (!fq_sqrt(Fq::mul(Fq::add(a.c0, s), inv2), &x0) && !fq_sqrt(Fq::mul(Fq::sub(a.c0, s), inv2), &x0)) return false;
  fe x1 = Fq::mul(a.c1, Fq::inv(Fq::add(x0, x0)));
  fq2 r = {x0, x1};
  *out = r;
  return pairing::f2_eq(pairing::f2_mul(r, r), a);
}
// on-curve and prime-order-subgroup checked, like ark-ec's deserialisation
inline bool g2_decompress(const uint8_t in[64], g2_affine* out) {
  fe c0, c1;
  memcpy(c0.v, in, 32);
  memcpy(c1.v, in + 32, 32);
  uint32_t flags = c1.v[7] >> 30;
  c1.v[7] &= 0x3fffffffu;
  out->inf = false;
  out->x = out->y = {Fq::zero(), Fq::zero()};
  if (flags == 3) return false;
  if (flags == 1) {
    for (int i = 0; i < 8; i++)
      if (c0.v[i] | c1.v[i]) return false;
    out->inf = true;
    return true;
  }
  if (cmp_words(c0, fq_modulus()) >= 0 || cmp_words(c1, fq_modulus()) >= 0) return false;
  fq2 x = {Fq::to_mont(c0), Fq::to_mont(c1)};
  fq2 rhs = pairing::f2_add(pairing::f2_mul(pairing::f2_mul(x, x), x), pairing::f2_const(pairing::B2));
  fq2 y;
  if (!f2_sqrt(rhs, &y)) return false;
  fq2 ny = pairing::f2_neg(y);
  bool larger = f2_greater(y, ny);
  out->x = x;
  out->y = (larger == (flags == 2)) ? y : ny;
  g2_affine t = pairing::g2_mul(*out, fr_modulus());
  return t.inf;
}
inline g2_affine g2_from_words(const uint64_t w[16]) {
  g2_affine q;
  q.x = {fe_from_words(w), fe_from_words(w + 4)};
  q.y = {fe_from_words(w + 8), fe_from_words(w + 12)};
  q.inf = pairing::f2_is_zero(q.x) && pairing::f2_is_zero(q.y);
  return q;
}
inline void g2_to_words(const g2_affine& q, uint64_t w[16]) {
  if (q.inf) {
    memset(w, 0, 16 * sizeof(uint64_t));
    return;
  }
  fe_to_words(q.x.c0, w);
  fe_to_words(q.x.c1, w + 4);
  fe_to_words(q.y.c0, w + 8);
  fe_to_words(q.y.c1, w + 12);
}

// ---- VerifyingKey <-> bytes (host only; shared by the vk and the proving-key blob) ------------------------------
struct OpenKey {
  g1_affine g, gamma_g;  // (0, 0) = infinity
  g2_affine h, beta_h;
};
inline g1_affine g1_from_words(const uint64_t w[8]) {
  g1_affine p;
  p.x = fe_from_words(w);
  p.y = fe_from_words(w + 4);
  return p;
}
inline void write_vk(Writer& w, const capgpu_verifying_key& vk, const OpenKey& ok) {
  uint8_t b[64];
  w.u64(vk.domain_size);
  w.u64(vk.num_inputs);
  w.u64(kNumWires);
  for (int i = 0; i < kNumWires; i++) {
    serialize_g1(g1_from_words(vk.sigma_comms[i]), b);
    w.put(b, 32);
  }
  w.u64(kNumSelectors);
  for (int i = 0; i < kNumSelectors; i++) {
    serialize_g1(g1_from_words(vk.selector_comms[i]), b);
    w.put(b, 32);
  }
  w.u64(kNumWires);
  for (int i = 0; i < kNumWires; i++) {
    serialize_fr(fe_from_words(vk.k[i]), b);
    w.put(b, 32);
  }
  serialize_g1(ok.g, b);
  w.put(b, 32);
  serialize_g1(ok.gamma_g, b);
  w.put(b, 32);
  g2_compress(ok.h, b);
  w.put(b, 64);
  g2_compress(ok.beta_h, b);
  w.put(b, 64);
  w.u8(0);  // is_merged = false
  w.u8(0);  // plookup_vk = None
}
// nullptr on success, else a static description of what is wrong
inline const char* read_vk(Reader& rd, capgpu_verifying_key* vk, OpenKey* ok) {
  memset(vk, 0, sizeof(*vk));
  vk->domain_size = rd.u64();
  vk->num_inputs = rd.u64();
  auto g1 = [&](uint64_t out[8]) {
    const uint8_t* b = rd.take(32);
    g1_affine p;
    if (!b || !g1_decompress_host(b, &p)) return false;
    affine_to_words(p, out);
    return true;
  };
  uint64_t cnt = 0;
  if (!rd.count(32, &cnt)) return "unexpected end of input";
  if (cnt != kNumWires) return "sigma_comms: a TurboPlonk key has 5 of them";
  for (int i = 0; i < kNumWires; i++)
    if (!g1(vk->sigma_comms[i])) return "sigma_comms: invalid compressed G1 point";
  if (!rd.count(32, &cnt)) return "unexpected end of input";
  if (cnt != kNumSelectors) return "selector_comms: a TurboPlonk key has 13 of them";
  for (int i = 0; i < kNumSelectors; i++)
    if (!g1(vk->selector_comms[i])) return "selector_comms: invalid compressed G1 point";
  if (!rd.count(32, &cnt)) return "unexpected end of input";
  if (cnt != kNumWires) return "k: a TurboPlonk key has 5 coset representatives";
  for (int i = 0; i < kNumWires; i++) {
    const uint8_t* b = rd.take(32);
    if (!b) return "unexpected end of input";
    fe v;
    memcpy(v.v, b, 32);
    if (cmp_words(v, fr_modulus()) >= 0) return "k: scalar not canonical";
    fe_to_words(Fr::to_mont(v), vk->k[i]);
  }
  uint64_t gw[8];
  if (!g1(gw)) return "open_key.g: invalid compressed G1 point";
  ok->g = g1_from_words(gw);
  if (!g1(gw)) return "open_key.gamma_g: invalid compressed G1 point";
  ok->gamma_g = g1_from_words(gw);
  const uint8_t* b = rd.take(64);
  if (!b) return "unexpected end of input";
  if (!g2_decompress(b, &ok->h)) return "open_key.h: invalid compressed G2 point";
  b = rd.take(64);
  if (!b) return "unexpected end of input";
  if (!g2_decompress(b, &ok->beta_h)) return "open_key.beta_h: invalid compressed G2 point";
  b = rd.take(2);
  if (!b) return "unexpected end of input";
  if (b[0] > 1 || b[1] > 1) return "invalid bool / Option tag";
  if (b[0] || b[1]) return "merged and plookup verifying keys are not supported";
  return nullptr;
}

// ---- bulk G1 on the device (params.hip) ------------------------------------------------------------------
// `n` compressed points (host, 32 B each, any alignment) -> device affine points in arkworks' Montgomery form,
// (0, 0) = infinity.  CAPGPU_ERR_SERIALIZATION (with the index of the first bad point in the message) if any
// encoding is invalid.
int decompress_g1(const uint8_t* host_bytes, size_t n, g1_affine* d_out, hipStream_t s);
// device affine points -> compressed bytes on the host.  internal_form != 0: coordinates are x * 2^261 (the
// resident SRS table), otherwise arkworks' x * 2^256.
int compress_g1(const g1_affine* d_pts, int internal_form, size_t n, uint8_t* host_out, hipStream_t s);
// Fr vectors between canonical little-endian bytes (host) and arkworks' Montgomery form (device, `stride` elements
// between consecutive vectors)
int fr_bytes_to_mont(const uint8_t* host_bytes, size_t n, fe* d_out, hipStream_t s);  // rejects values >= r
int fr_mont_to_bytes(const fe* d_in, size_t n, uint8_t* host_out, hipStream_t s);

}  // namespace params
}  // namespace cap
