// BN254 G1 in XYZZ coordinates on the lazy 29-bit field (field29.hpp).  Same formulas as curve.hpp
// (madd-2008-s, add-2008-s, dbl-2008-s-1); what changes is bound bookkeeping instead of conditional
// subtractions.  Invariants of a point held in registers (`g1x`):
//     x, y   normalized, < 2p      (weak_reduce'd: they are differences)
//     zz,zzz normalized, < 1.1p    (products)
//     infinity  <=>  zz is the literal 0
// Affine table entries (`g1_affine` in memory) hold canonical values (< p) in internal Montgomery form
// (x * 2^261 mod p); (0, 0) encodes infinity.  Bucket points in memory (`g1_xyzz`) hold any value < 2^256.
#pragma once
#include "curve.hpp"
#include "field29.hpp"

namespace cap {

struct g1a {  // affine, registers
  fl x, y;
};
struct g1x {  // XYZZ, registers
  fl x, y, zz, zzz;
};

template <int SCHED>
struct G1LT {
  using F = Fl<FqP29, SCHED>;

  static CAP_HD bool all_zero(const fl& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) o |= a.v[i];
    return o == 0;
  }
  static CAP_HD bool is_inf(const g1x& p) { return all_zero(p.zz); }
  static CAP_HD bool is_inf(const g1a& p) { return all_zero(p.x) && all_zero(p.y); }
  static CAP_HD g1x inf() {
    g1x r;
    r.x = F::zero();
    r.y = F::zero();
    r.zz = F::zero();
    r.zzz = F::zero();
    return r;
  }
  static CAP_HD g1x from_affine(const g1a& p) {
    if (is_inf(p)) return inf();
    g1x r;
    r.x = p.x;
    r.y = p.y;
    r.zz = F::one();
    r.zzz = F::one();
    return r;
  }

  // ---- memory images ------------------------------------------------------------------------------------
  static CAP_HD g1a load(const g1_affine& m) {
    g1a r;
    r.x = F::load(m.x);
    r.y = F::load(m.y);
    return r;
  }
  static CAP_HD g1x load(const g1_xyzz& m) {
    g1x r;
    r.x = F::load(m.x);
    r.y = F::load(m.y);
    r.zz = F::load(m.zz);
    r.zzz = F::load(m.zzz);
    return r;
  }
  static CAP_HD g1_xyzz store(const g1x& p) {  // coordinates are already < 2p: pack only
    g1_xyzz m;
    m.x = F::pack(p.x);
    m.y = F::pack(p.y);
    m.zz = F::pack(p.zz);
    m.zzz = F::pack(p.zzz);
    return m;
  }
  static CAP_HD g1_affine store_affine(const g1a& p) {  // canonical
    g1_affine m;
    m.x = F::pack(F::canonical(p.x));
    m.y = F::pack(F::canonical(p.y));
    return m;
  }

  // ---- group law --------------------------------------------------------------------------------------------
  // doubling of an affine point (mdbl-2008-s-1); q canonical / < 2p
  static CAP_HD g1x dbl_affine(const g1a& q) {
    if (is_inf(q)) return inf();
    fl u = F::add(q.y, q.y);
    fl v = F::sqr(u);
    fl w = F::mul(u, v);
    fl s = F::mul(q.x, v);
    fl xx = F::sqr(q.x);
    fl m = F::normalize(F::add(F::add(xx, xx), xx));
    g1x r;
    r.x = F::weak_reduce(F::sub(F::sqr(m), F::add(s, s)));
    r.y = F::weak_reduce(F::mul_add_mul(m, F::sub(s, r.x), F::neg(w), q.y));
    r.zz = v;
    r.zzz = w;
    return r;
  }
  static CAP_HD g1x dbl(const g1x& p) {
    if (is_inf(p)) return p;
    fl u = F::add(p.y, p.y);
    fl v = F::sqr(u);
    fl w = F::mul(u, v);
    fl s = F::mul(p.x, v);
    fl xx = F::sqr(p.x);
    fl m = F::normalize(F::add(F::add(xx, xx), xx));
    g1x r;
    r.x = F::weak_reduce(F::sub(F::sqr(m), F::add(s, s)));
    r.y = F::weak_reduce(F::mul_add_mul(m, F::sub(s, r.x), F::neg(w), p.y));
    r.zz = F::mul(v, p.zz);
    r.zzz = F::mul(w, p.zzz);
    return r;
  }
  // acc + q, q affine; `negate` adds -q.  Handles infinity on either side, acc == q and acc == -q.
  static CAP_HD g1x add_mixed(const g1x& a, const g1a& q_in, bool negate = false) {
    if (is_inf(q_in)) return a;
    g1a q = q_in;
    if (negate) q.y = F::neg(q.y);  // 16p - y: normalized, fine as a multiplicand
    if (is_inf(a)) {
      g1x r;
      r.x = q.x;
      r.y = negate ? F::weak_reduce(q.y) : q.y;
      r.zz = F::one();
      r.zzz = F::one();
      return r;
    }
    fl u2 = F::mul(q.x, a.zz);
    fl s2 = F::mul(q.y, a.zzz);
    fl p = F::sub(u2, a.x);
    fl r = F::sub(s2, a.y);
    if (F::is_zero(p)) {
      if (F::is_zero(r)) {
        g1a qq;
        qq.x = q.x;
        qq.y = F::weak_reduce(q.y);
        return dbl_affine(qq);
      }
      return inf();
    }
    fl pp = F::sqr(p);
    fl ppp = F::mul(p, pp);
    fl qq = F::mul(a.x, pp);
    g1x o;
    o.x = F::weak_reduce(F::sub(F::sub(F::sqr(r), ppp), F::add(qq, qq)));
    o.y = F::weak_reduce(F::mul_add_mul(r, F::sub(qq, o.x), F::neg(a.y), ppp));  // one reduction for both products
    o.zz = F::mul(a.zz, pp);
    o.zzz = F::mul(a.zzz, ppp);
    return o;
  }
  // The mixed addition of the bucket-accumulation loop (msm_accumulate): acc += q (or -q), common case only.
  // Returns false - acc untouched - when the x-difference vanishes, which is how EVERY special case shows up: the
  // accumulator at infinity (zz = 0 and x = 0 make both sides 0), acc == q (doubling), acc == -q (cancellation); the
  // caller then takes the general add_mixed.  q must not be the point at infinity (the caller tests that).
  // Compared with add_mixed the carries are propagated only where the next operation needs them:
  //   * -q.y is 16p - y with un-carried limbs (< 2^31): it is one operand of one multiplication;
  //   * r^2 - ppp - 2qq is carried once, not twice;  qq - x3 (+ 2p, limbs < 1.5 * 2^30) feeds the fused product directly;
  //   * -acc.y is 4p - y with un-carried limbs (< 2^30);  y3 stays a product sum (< 1.5 p) without the weak reduction.
  // Invariants of acc here: x < 2p, y < 3p, zz, zzz < 1.2p, all with normalized limbs.
  static CAP_HD bool madd_acc(g1x& a, const g1a& q, bool negate) {
    const fl qy = negate ? F::neg_lazy(q.y) : q.y;
    const fl u2 = F::mul(q.x, a.zz);
    const fl s2 = F::mul(qy, a.zzz);
    const fl p = F::sub(u2, a.x);
    if (F::is_zero(p)) return false;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("; madd_acc: common path");  // no instruction: tools/isa_mix.py finds the hot block by this line
#endif
    const fl r = F::sub(s2, a.y);
    const fl pp = F::sqr(p);
    const fl ppp = F::mul(p, pp);
    const fl qq = F::mul(a.x, pp);
    g1x o;
    o.x = F::weak_reduce(F::sub_from_lazy(F::sub2p_lazy(F::sqr(r), ppp), F::add(qq, qq)));
    // one reduction for both products; both second operands un-carried: 9 x (2^29 * 1.5 * 2^30 + 2^30 * 2^29) plus the
    // reduction terms stays below 2^64 per column
    o.y = F::mul_add_mul(r, F::sub2p_lazy(qq, o.x), F::neg4p_lazy(a.y), ppp);
    o.zz = F::mul(a.zz, pp);
    o.zzz = F::mul(a.zzz, ppp);
    a = o;
    return true;
  }
  // a += b for two XYZZ points, common case only - the running sums of the bucket reduction (msm_reduce_segments).
  // Returns false - a untouched - when the x-difference vanishes: either side at infinity (zz = 0 and x = 0 make both
  // cross products 0), a == b, a == -b; the caller falls back to add().  Carries as in madd_acc.
  // Invariants in and out: x < 2p, y < 3p, zz, zzz < 1.2p, normalized limbs.
  static CAP_HD bool add_acc(g1x& a, const g1x& b) {
    const fl u1 = F::mul(a.x, b.zz);
    const fl u2 = F::mul(b.x, a.zz);
    const fl p = F::sub(u2, u1);
    if (F::is_zero(p)) return false;
    const fl s1 = F::mul(a.y, b.zzz);
    const fl s2 = F::mul(b.y, a.zzz);
    const fl r = F::sub(s2, s1);
    const fl pp = F::sqr(p);
    const fl ppp = F::mul(p, pp);
    const fl qq = F::mul(u1, pp);
    g1x o;
    o.x = F::weak_reduce(F::sub_from_lazy(F::sub2p_lazy(F::sqr(r), ppp), F::add(qq, qq)));
    o.y = F::mul_add_mul(r, F::sub2p_lazy(qq, o.x), F::neg2p_lazy(s1), ppp);
    o.zz = F::mul(F::mul(a.zz, b.zz), pp);
    o.zzz = F::mul(F::mul(a.zzz, b.zzz), ppp);
    a = o;
    return true;
  }
  static CAP_HD g1x add(const g1x& a, const g1x& b) {
    if (is_inf(a)) return b;
    if (is_inf(b)) return a;
    fl u1 = F::mul(a.x, b.zz);
    fl u2 = F::mul(b.x, a.zz);
    fl s1 = F::mul(a.y, b.zzz);
    fl s2 = F::mul(b.y, a.zzz);
    fl p = F::sub(u2, u1);
    fl r = F::sub(s2, s1);
    if (F::is_zero(p)) {
      if (F::is_zero(r)) return dbl(a);
      return inf();
    }
    fl pp = F::sqr(p);
    fl ppp = F::mul(p, pp);
    fl qq = F::mul(u1, pp);
    g1x o;
    o.x = F::weak_reduce(F::sub(F::sub(F::sqr(r), ppp), F::add(qq, qq)));
    o.y = F::weak_reduce(F::mul_add_mul(r, F::sub(qq, o.x), F::neg(s1), ppp));
    o.zz = F::mul(F::mul(a.zz, b.zz), pp);
    o.zzz = F::mul(F::mul(a.zzz, b.zzz), ppp);
    return o;
  }

  // ---- inversion (a^(p-2)) and normalisation to affine ------------------------------------------------
  static CAP_HD fl inv(const fl& a) {
    // exponent p - 2, 32-bit words of the modulus
    fl r = F::one();
    bool started = false;
    for (int i = 7; i >= 0; i--) {
      uint32_t e = FqP::MOD[i] - (i == 0 ? 2u : 0u);  // low word of p is ...fd47: no borrow
      for (int b = 31; b >= 0; b--) {
        if (started) r = F::sqr(r);
        if ((e >> b) & 1) {
          r = started ? F::mul(r, a) : a;
          started = true;
        }
      }
    }
    return r;
  }
  static CAP_HD g1a to_affine(const g1x& p) {
    g1a r;
    if (is_inf(p)) {
      r.x = F::zero();
      r.y = F::zero();
      return r;
    }
    fl zi = inv(p.zzz);              // 1/zzz
    fl t = F::mul(zi, p.zz);         // zz/zzz
    fl zz_inv = F::sqr(t);           // zz^3 = zzz^2  =>  (zz/zzz)^2 = 1/zz
    r.x = F::mul(p.x, zz_inv);
    r.y = F::mul(p.y, zi);
    return r;
  }
  // the Jacobian triple of the C ABI, external Montgomery form (x * 2^256), canonical
  static CAP_HD g1_jac to_jac_ext(const g1x& p) {
    g1_jac r;
    if (is_inf(p)) {
      r.x = Fq::one();
      r.y = Fq::one();
      r.z = Fq::zero();
      return r;
    }
    r.x = F::to_ext(F::mul(p.x, p.zz));
    r.y = F::to_ext(F::mul(p.y, p.zzz));
    r.z = F::to_ext(p.zz);
    return r;
  }
};
using G1L = G1LT<CAP_FL_SCHED>;  // the translation unit's default multiplication schedule (field29.hpp)

}  // namespace cap
