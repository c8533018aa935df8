// BN254 optimal-ate pairing on the host, for the verifier (the acceptance predicate
// `proof::{transfer,mint,freeze}::verify` -> `PlonkKzgSnark::verify`, src/proof/transfer.rs:192-212).
// Verification is CPU-cheap in the reference and stays on the CPU here (milliseconds, once per proof).
//
// Construction (ark-bn254 0.3.0 / EIP-197): Fq2 = Fq[u]/(u^2+1); twist E': y^2 = x^3 + 3/(9+u);
// Fq12 = Fq[w]/(w^12 - 18 w^6 + 82), i.e. u = w^6 - 9; twist map (x, y) -> (x w^2, y w^3).
// A line through twist points with slope m (in Fq2), evaluated at P = (xp, yp) in G1, is
//     l = -yp + (m xp) w + (yr - m xr) w^3
// (divide the untwisted line by w^3... every factor living in a proper subfield dies in the final exponentiation).
// Miller loop over 6x+2 = 29793968203157093288 with affine steps (one Fq2 inversion each), two Frobenius steps
// with pi(x, y) = (conj(x) xi^((p-1)/3), conj(y) xi^((p-1)/2)), then f^((p^12-1)/r) by plain square-and-multiply.
// Simple and dense on purpose; cross-checked against oracle/pairing.py in tests/test_verify.py.
#pragma once
#include <vector>

#include "curve.hpp"

namespace cap {
namespace pairing {

struct fq2 {
  fe c0, c1;
};
struct fq12 {
  fe c[12];
};
struct g2_affine {  // twist point, Montgomery coordinates; inf flag explicit
  fq2 x, y;
  bool inf = false;
};

inline fe fq_from_words32(const uint32_t w[8]) {
  fe r;
  for (int i = 0; i < 8; i++) r.v[i] = w[i];
  return r;
}
inline fe fq_small(uint32_t v) {
  fe t = Fq::zero();
  t.v[0] = v;
  return Fq::to_mont(t);
}

// ---- Fq2 --------------------------------------------------------------------------------------------------
inline fq2 f2_add(const fq2& a, const fq2& b) { return {Fq::add(a.c0, b.c0), Fq::add(a.c1, b.c1)}; }
inline fq2 f2_sub(const fq2& a, const fq2& b) { return {Fq::sub(a.c0, b.c0), Fq::sub(a.c1, b.c1)}; }
inline fq2 f2_neg(const fq2& a) { return {Fq::neg(a.c0), Fq::neg(a.c1)}; }
inline fq2 f2_conj(const fq2& a) { return {a.c0, Fq::neg(a.c1)}; }
inline fq2 f2_mul(const fq2& a, const fq2& b) {
  fe t0 = Fq::mul(a.c0, b.c0), t1 = Fq::mul(a.c1, b.c1);
  fe s = Fq::mul(Fq::add(a.c0, a.c1), Fq::add(b.c0, b.c1));
  return {Fq::sub(t0, t1), Fq::sub(Fq::sub(s, t0), t1)};
}
inline fq2 f2_scalar(const fq2& a, const fe& k) { return {Fq::mul(a.c0, k), Fq::mul(a.c1, k)}; }
inline fq2 f2_inv(const fq2& a) {
  fe d = Fq::inv(Fq::add(Fq::sqr(a.c0), Fq::sqr(a.c1)));
  return {Fq::mul(a.c0, d), Fq::neg(Fq::mul(a.c1, d))};
}
inline bool f2_is_zero(const fq2& a) { return Fq::is_zero(a.c0) && Fq::is_zero(a.c1); }
inline bool f2_eq(const fq2& a, const fq2& b) { return Fq::eq(a.c0, b.c0) && Fq::eq(a.c1, b.c1); }

// constants (Montgomery words), generated from oracle/pairing.py
static const uint32_t GAMMA_X1[2][8] = {{0x4563ab30u, 0xb5773b10u, 0xa9aa6454u, 0x347f91c8u, 0x242e0991u, 0x7a007127u, 0x118214ecu, 0x1956bcd8u}, {0xa0aa4757u, 0x6e849f1eu, 0x89f89141u, 0xaa1c7b6du, 0xfae0ca3au, 0xb6e713cdu, 0x4e82ebc3u, 0x26694fbbu}};
static const uint32_t GAMMA_Y1[2][8] = {{0x2936b629u, 0xe4bbdd0cu, 0xe133bacbu, 0xbb30f162u, 0xf9645366u, 0x31a9d1b6u, 0xa500f8ddu, 0x253570beu}, {0x5ffe77c7u, 0xa1d77ce4u, 0x7826d1dbu, 0x07affd11u, 0xbb7edc6bu, 0x6d16bd27u, 0x85defeccu, 0x2c872002u}};
static const uint32_t GAMMA_X2[2][8] = {{0x13e80b9cu, 0x3350c88eu, 0xdb5e56b9u, 0x7dce557cu, 0xb615564au, 0x6001b4b8u, 0x020217e0u, 0x2682e617u}, {0, 0, 0, 0, 0, 0, 0, 0}};
static const uint32_t GAMMA_Y2[2][8] = {{0x12edefaau, 0x68c34889u, 0x72aabf4fu, 0x8d087f68u, 0x09081231u, 0x51e1a247u, 0x4729c0fau, 0x2259d6b1u}, {0, 0, 0, 0, 0, 0, 0, 0}};
static const uint32_t B2[2][8] = {{0x77b802a8u, 0x3bf938e3u, 0x3633535du, 0x020b1b27u, 0x49755260u, 0x26b7edf0u, 0x4384a86du, 0x2514c632u}, {0xd1dcff67u, 0x38e7ecccu, 0x93ce0d3eu, 0x65f0b37du, 0x22ac00aau, 0xd749d0ddu, 0x4a688d4du, 0x0141b9ceu}};
static const uint32_t G2X[2][8] = {{0x02bc2026u, 0x8e83b5d1u, 0x497b0172u, 0xdceb1935u, 0x97811adfu, 0xfbb82647u, 0xaf96503bu, 0x19573841u}, {0xa84c6140u, 0xafb4737du, 0x5802d8c4u, 0x6043dd5au, 0x52a02f86u, 0x09e950fcu, 0x3aea7b6bu, 0x14fef083u}};
static const uint32_t G2Y[2][8] = {{0x886be9f6u, 0x619dfa9du, 0xf59e9b78u, 0xfe7fd297u, 0x231b7dfeu, 0xff9e1a62u, 0xae9e4206u, 0x28fd7eebu}, {0xc71856eeu, 0x64095b56u, 0x327d3cbbu, 0xdc57f922u, 0x33351076u, 0x55f935beu, 0x93fd6482u, 0x0da4a0e6u}};
// (p^12 - 1) / r, 2790 bits, little-endian 32-bit words
static const uint32_t FINAL_EXP[88] = {0xca86f120u, 0x86964b64u, 0xe54523a4u, 0x40a4efb7u, 0x96e84abbu, 0x837fa978u, 0xb9b2b918u, 0x361102b6u, 0xf35692dau, 0xc0de81deu, 0xa6c3c760u, 0xbe04c7e8u, 0xd570bb7fu, 0xd766f9c9u, 0x83561841u, 0xc230974du, 0xc3be69a3u, 0x5bba1668u, 0x10526294u, 0x7f3811c4u, 0xdadda71cu, 0x29baee7du, 0x145da900u, 0xbf813b8du, 0x423f9a2cu, 0x641bbadfu, 0x44eacc5eu, 0xa80bb4eau, 0x14fde37cu, 0xcd656648u, 0x580291d2u, 0x4a0364b9u, 0x0826f0ddu, 0xee93dfb1u, 0xc5514724u, 0x6b42db8du, 0x0b0f3785u, 0xbb10cf43u, 0x6f804216u, 0x40494e40u, 0xacf3aafbu, 0x55cfe107u, 0xe0ebae87u, 0x2088ec80u, 0x11a337a0u, 0x846a3ed0u, 0x1e3a5195u, 0x48a45a4au, 0xdfc50e16u, 0xe5664568u, 0x4c0cc4ebu, 0xab6a4129u, 0xd268c7dau, 0x82d0d602u, 0xed3cc48au, 0x6668449au, 0xb2015dfcu, 0x5062cd0fu, 0xb1ddb3d1u, 0x7f2940a8u, 0x2a226448u, 0x77f5b63au, 0x61e443aeu, 0xfef07813u, 0x88d5c6c8u, 0xf977870eu, 0x1f676baau, 0x790364a6u, 0xceaddea3u, 0x5887e72eu, 0xa09a1b70u, 0x1377e563u, 0x1bd8c3b2u, 0x0c54efeeu, 0xd524d8f7u, 0x3ec3d15au, 0xb2383a5du, 0xdaf15466u, 0xbb94fec0u, 0xe1e30a73u, 0x5f3f7be2u, 0x6a1c7101u, 0x6369b1ffu, 0x842d43bfu, 0x107d20bcu, 0x20fddadfu, 0x4b6dc970u, 0x0000002fu};
constexpr uint64_t ATE_LOOP_HI = 0x1ULL;                   // 6x + 2 = 0x1_9d797039be763ba8 (65 bits)
constexpr uint64_t ATE_LOOP_LO = 0x9d797039be763ba8ULL;

inline fq2 f2_const(const uint32_t w[2][8]) { return {fq_from_words32(w[0]), fq_from_words32(w[1])}; }

// ---- G2 (twist, affine) ---------------------------------------------------------------------------------
inline g2_affine g2_generator() {
  g2_affine g;
  g.x = f2_const(G2X);
  g.y = f2_const(G2Y);
  return g;
}
inline bool g2_on_curve(const g2_affine& q) {
  if (q.inf) return true;
  // canonical coordinates only (arkworks never holds a value >= p; Fq::mul would reduce one silently)
  if (Fq::geq_mod(q.x.c0) || Fq::geq_mod(q.x.c1) || Fq::geq_mod(q.y.c0) || Fq::geq_mod(q.y.c1)) return false;
  fq2 lhs = f2_mul(q.y, q.y);
  fq2 rhs = f2_add(f2_mul(f2_mul(q.x, q.x), q.x), f2_const(B2));
  return f2_eq(lhs, rhs);
}
// r + q with the slope returned (tangent when the points coincide); callers guarantee no vertical line
inline g2_affine g2_step(const g2_affine& r, const g2_affine& q, fq2* slope) {
  fq2 m;
  if (f2_eq(r.x, q.x) && f2_eq(r.y, q.y)) {
    fq2 xx = f2_mul(r.x, r.x);
    m = f2_mul(f2_add(f2_add(xx, xx), xx), f2_inv(f2_add(r.y, r.y)));
  } else {
    m = f2_mul(f2_sub(q.y, r.y), f2_inv(f2_sub(q.x, r.x)));
  }
  g2_affine o;
  o.x = f2_sub(f2_sub(f2_mul(m, m), r.x), q.x);
  o.y = f2_sub(f2_mul(m, f2_sub(r.x, o.x)), r.y);
  if (slope) *slope = m;
  return o;
}
inline g2_affine g2_add(const g2_affine& a, const g2_affine& b) {
  if (a.inf) return b;
  if (b.inf) return a;
  if (f2_eq(a.x, b.x) && !f2_eq(a.y, b.y)) {
    g2_affine o;
    o.inf = true;
    o.x = o.y = {Fq::zero(), Fq::zero()};
    return o;
  }
  return g2_step(a, b, nullptr);
}
// scalar: canonical little-endian 32-bit words
inline g2_affine g2_mul(const g2_affine& q, const fe& k) {
  g2_affine acc;
  acc.inf = true;
  acc.x = acc.y = {Fq::zero(), Fq::zero()};
  for (int i = 7; i >= 0; i--)
    for (int b = 31; b >= 0; b--) {
      acc = g2_add(acc, acc);
      if ((k.v[i] >> b) & 1) acc = g2_add(acc, q);
    }
  return acc;
}

// ---- Fq12 ---------------------------------------------------------------------------------------------------
inline fq12 f12_one() {
  fq12 r;
  for (int i = 0; i < 12; i++) r.c[i] = Fq::zero();
  r.c[0] = Fq::one();
  return r;
}
inline bool f12_eq(const fq12& a, const fq12& b) {
  for (int i = 0; i < 12; i++)
    if (!Fq::eq(a.c[i], b.c[i])) return false;
  return true;
}
inline fq12 f12_mul(const fq12& a, const fq12& b) {
  static const fe k18 = fq_small(18), k82 = fq_small(82);
  fe t[23];
  for (int i = 0; i < 23; i++) t[i] = Fq::zero();
  for (int i = 0; i < 12; i++) {
    if (Fq::is_zero(a.c[i])) continue;
    for (int j = 0; j < 12; j++) t[i + j] = Fq::add(t[i + j], Fq::mul(a.c[i], b.c[j]));
  }
  for (int k = 22; k >= 12; k--) {  // w^k = 18 w^(k-6) - 82 w^(k-12)
    if (Fq::is_zero(t[k])) continue;
    t[k - 6] = Fq::add(t[k - 6], Fq::mul(t[k], k18));
    t[k - 12] = Fq::sub(t[k - 12], Fq::mul(t[k], k82));
  }
  fq12 r;
  for (int i = 0; i < 12; i++) r.c[i] = t[i];
  return r;
}
inline fq12 f12_pow(const fq12& a, const uint32_t* e, int nwords) {
  fq12 r = f12_one();
  bool started = false;
  for (int i = nwords - 1; i >= 0; i--)
    for (int b = 31; b >= 0; b--) {
      if (started) r = f12_mul(r, r);
      if ((e[i] >> b) & 1) {
        r = started ? f12_mul(r, a) : a;
        started = true;
      }
    }
  return r;
}
// (c0 + c1 u) w^pos with u = w^6 - 9, added into l
inline void f12_add_f2(fq12& l, const fq2& c, int pos) {
  static const fe k9 = fq_small(9);
  l.c[pos] = Fq::add(l.c[pos], Fq::sub(c.c0, Fq::mul(c.c1, k9)));
  l.c[pos + 6] = Fq::add(l.c[pos + 6], c.c1);
}
inline fq12 line_value(const fq2& m, const g2_affine& r, const g1_affine& p) {
  fq12 l;
  for (int i = 0; i < 12; i++) l.c[i] = Fq::zero();
  l.c[0] = Fq::neg(p.y);
  f12_add_f2(l, f2_scalar(m, p.x), 1);
  f12_add_f2(l, f2_sub(r.y, f2_mul(m, r.x)), 3);
  return l;
}

inline fq12 miller_loop(const g2_affine& q, const g1_affine& p) {
  if (q.inf || G1::is_inf(p)) return f12_one();
  fq12 f = f12_one();
  g2_affine r = q;
  fq2 m;
  for (int i = 63; i >= 0; i--) {  // bit 64 of 6x+2 is the leading one
    g2_affine r2 = g2_step(r, r, &m);
    f = f12_mul(f12_mul(f, f), line_value(m, r, p));
    r = r2;
    if ((ATE_LOOP_LO >> i) & 1) {
      g2_affine r3 = g2_step(r, q, &m);
      f = f12_mul(f, line_value(m, r, p));
      r = r3;
    }
  }
  g2_affine q1, nq2;
  q1.x = f2_mul(f2_conj(q.x), f2_const(GAMMA_X1));
  q1.y = f2_mul(f2_conj(q.y), f2_const(GAMMA_Y1));
  nq2.x = f2_mul(q.x, f2_const(GAMMA_X2));
  nq2.y = f2_neg(f2_mul(q.y, f2_const(GAMMA_Y2)));
  g2_affine r3 = g2_step(r, q1, &m);
  f = f12_mul(f, line_value(m, r, p));
  r = r3;
  (void)g2_step(r, nq2, &m);
  f = f12_mul(f, line_value(m, r, p));
  return f;
}
inline fq12 final_exponentiation(const fq12& f) { return f12_pow(f, FINAL_EXP, 88); }

// prod_i e(P_i, Q_i) == 1 ?
inline bool pairing_product_is_one(const std::vector<std::pair<g1_affine, g2_affine>>& pairs) {
  fq12 f = f12_one();
  for (const auto& pq : pairs) f = f12_mul(f, miller_loop(pq.second, pq.first));
  return f12_eq(final_exponentiation(f), f12_one());
}

}  // namespace pairing
}  // namespace cap
