// BN254 G1 arithmetic for the MSM kernels (K5/K6 of SURVEY.md §8a).
//
// Replaces ark-ec 0.3.0 `GroupProjective::{add_assign_mixed, double_in_place,
// add_assign}` (Cargo.lock:103-105) as used by VariableBaseMSM under
// src/proof/transfer.rs:181-186.  The reference keeps buckets in Jacobian
// coordinates (madd-2007-bl, 7M+4S); on gfx950 every multiplication is ~450
// VALU instructions, so buckets live in extended Jacobian "XYZZ" coordinates
// (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2): mixed add 8M+2S, no doubling-by-adding
// tricks, cheap conversion to the Jacobian triple the C ABI returns.
// A G1 element has one affine value, so the coordinate system never shows in
// results (parity is checked on affine coordinates).
#pragma once
#include "field.hpp"

namespace cap {

struct alignas(16) g1_affine {  // Montgomery coordinates; (0,0) encodes infinity
  fe x, y;
};
struct alignas(16) g1_xyzz {  // infinity: zz == 0
  fe x, y, zz, zzz;
};
struct alignas(16) g1_jac {  // what capgpu_msm_g1 hands back (arkworks GroupProjective order)
  fe x, y, z;
};

struct G1 {
  static CAP_HD bool is_inf(const g1_affine& p) { return Fq::is_zero(p.x) && Fq::is_zero(p.y); }
  static CAP_HD bool is_inf(const g1_xyzz& p) { return Fq::is_zero(p.zz); }
  static CAP_HD g1_xyzz inf() {
    g1_xyzz r;
    r.x = Fq::zero();
    r.y = Fq::zero();
    r.zz = Fq::zero();
    r.zzz = Fq::zero();
    return r;
  }
  static CAP_HD g1_xyzz from_affine(const g1_affine& p) {
    g1_xyzz r;
    if (is_inf(p)) return inf();
    r.x = p.x;
    r.y = p.y;
    r.zz = Fq::one();
    r.zzz = Fq::one();
    return r;
  }
  static CAP_HD g1_affine neg(const g1_affine& p) {
    g1_affine r;
    r.x = p.x;
    r.y = Fq::neg(p.y);
    return r;
  }

  // dbl-2008-s-1 (a = 0)
  static CAP_HD g1_xyzz dbl(const g1_xyzz& p) {
    if (is_inf(p)) return p;
    fe u = Fq::dbl(p.y);
    fe v = Fq::sqr(u);
    fe w = Fq::mul(u, v);
    fe s = Fq::mul(p.x, v);
    fe xx = Fq::sqr(p.x);
    fe m = Fq::add(Fq::dbl(xx), xx);
    g1_xyzz r;
    r.x = Fq::sub(Fq::sqr(m), Fq::dbl(s));
    r.y = Fq::sub(Fq::mul(m, Fq::sub(s, r.x)), Fq::mul(w, p.y));
    r.zz = Fq::mul(v, p.zz);
    r.zzz = Fq::mul(w, p.zzz);
    return r;
  }
  // mdbl-2008-s-1: doubling of an affine point
  static CAP_HD g1_xyzz dbl_affine(const g1_affine& p) {
    if (is_inf(p)) return inf();
    fe u = Fq::dbl(p.y);
    fe v = Fq::sqr(u);
    fe w = Fq::mul(u, v);
    fe s = Fq::mul(p.x, v);
    fe xx = Fq::sqr(p.x);
    fe m = Fq::add(Fq::dbl(xx), xx);
    g1_xyzz r;
    r.x = Fq::sub(Fq::sqr(m), Fq::dbl(s));
    r.y = Fq::sub(Fq::mul(m, Fq::sub(s, r.x)), Fq::mul(w, p.y));
    r.zz = v;
    r.zzz = w;
    return r;
  }
  // madd-2008-s: acc += q (affine).  Handles acc == inf, q == inf, acc == q, acc == -q.
  static CAP_HD g1_xyzz add_mixed(const g1_xyzz& a, const g1_affine& q) {
    if (is_inf(q)) return a;
    if (is_inf(a)) return from_affine(q);
    fe u2 = Fq::mul(q.x, a.zz);
    fe s2 = Fq::mul(q.y, a.zzz);
    fe p = Fq::sub(u2, a.x);
    fe r = Fq::sub(s2, a.y);
    if (Fq::is_zero(p)) {
      if (Fq::is_zero(r)) return dbl_affine(q);
      return inf();
    }
    fe pp = Fq::sqr(p);
    fe ppp = Fq::mul(p, pp);
    fe qq = Fq::mul(a.x, pp);
    g1_xyzz o;
    o.x = Fq::sub(Fq::sub(Fq::sqr(r), ppp), Fq::dbl(qq));
    o.y = Fq::sub(Fq::mul(r, Fq::sub(qq, o.x)), Fq::mul(a.y, ppp));
    o.zz = Fq::mul(a.zz, pp);
    o.zzz = Fq::mul(a.zzz, ppp);
    return o;
  }
  // add-2008-s
  static CAP_HD g1_xyzz add(const g1_xyzz& a, const g1_xyzz& b) {
    if (is_inf(a)) return b;
    if (is_inf(b)) return a;
    fe u1 = Fq::mul(a.x, b.zz);
    fe u2 = Fq::mul(b.x, a.zz);
    fe s1 = Fq::mul(a.y, b.zzz);
    fe s2 = Fq::mul(b.y, a.zzz);
    fe p = Fq::sub(u2, u1);
    fe r = Fq::sub(s2, s1);
    if (Fq::is_zero(p)) {
      if (Fq::is_zero(r)) return dbl(a);
      return inf();
    }
    fe pp = Fq::sqr(p);
    fe ppp = Fq::mul(p, pp);
    fe qq = Fq::mul(u1, pp);
    g1_xyzz o;
    o.x = Fq::sub(Fq::sub(Fq::sqr(r), ppp), Fq::dbl(qq));
    o.y = Fq::sub(Fq::mul(r, Fq::sub(qq, o.x)), Fq::mul(s1, ppp));
    o.zz = Fq::mul(Fq::mul(a.zz, b.zz), pp);
    o.zzz = Fq::mul(Fq::mul(a.zzz, b.zzz), ppp);
    return o;
  }
  // XYZZ -> Jacobian triple (X*ZZ, Y*ZZZ, ZZ): x = X/ZZ = X*ZZ/ZZ^2, y = Y/ZZZ = Y*ZZZ/ZZ^3.
  static CAP_HD g1_jac to_jac(const g1_xyzz& p) {
    g1_jac r;
    if (is_inf(p)) {
      r.x = Fq::one();
      r.y = Fq::one();
      r.z = Fq::zero();
      return r;
    }
    r.x = Fq::mul(p.x, p.zz);
    r.y = Fq::mul(p.y, p.zzz);
    r.z = p.zz;
    return r;
  }
  // one field inversion; used by the SRS precompute and by the host for O(1) work per commitment
  static CAP_HD g1_affine to_affine(const g1_xyzz& p) {
    g1_affine r;
    if (is_inf(p)) {
      r.x = Fq::zero();
      r.y = Fq::zero();
      return r;
    }
    fe zi = Fq::inv(p.zzz);    // 1/zzz
    fe t = Fq::mul(zi, p.zz);  // zz/zzz
    fe zz_inv = Fq::sqr(t);    // zz^3 = zzz^2  =>  (zz/zzz)^2 = zz^2/zz^3 = 1/zz
    r.x = Fq::mul(p.x, zz_inv);
    r.y = Fq::mul(p.y, zi);
    return r;
  }
  static CAP_HD g1_affine jac_to_affine(const g1_jac& p) {
    g1_affine r;
    if (Fq::is_zero(p.z)) {
      r.x = Fq::zero();
      r.y = Fq::zero();
      return r;
    }
    fe zi = Fq::inv(p.z);
    fe zi2 = Fq::sqr(zi);
    r.x = Fq::mul(p.x, zi2);
    r.y = Fq::mul(p.y, Fq::mul(zi2, zi));
    return r;
  }
};

}  // namespace cap
