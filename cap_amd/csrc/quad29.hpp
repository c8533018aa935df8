// One XYZZ point spread over FOUR adjacent lanes (a "quad"): lane q of the quad holds coordinate q of (x, y, zz, zzz).
//
// Why: the finishing kernels of a small MSM launch (a single proof's four launches) are chains of dependent point
// additions on a chip that is mostly idle.  One lane executes the 12 field multiplications of an addition one after the
// other (6.5 us, tools/ubench_lonewave.hip); the formulas are only FOUR multiplications deep.  With the coordinates in
// the lanes of a quad every lane executes one multiplication per level - the same instruction stream, different
// operands, fetched from the neighbours with quad-permute DPP moves - and an addition costs 4 multiplications plus the
// exchange instead of 12.  Same formulas (add-2008-s, dbl-2008-s-1), same results; throughput per lane is lower, so
// this is for launches with lanes to spare only.
//
//   addition  level 1   u1 = x1 zz2      s1 = y1 zzz2     u2 = zz1 x2      s2 = zzz1 y2        (own A, B of lane q ^ 2)
//             level 2   pp = p p         rr = r r         zz12 = zz1 zz2   zzz12 = zzz1 zzz2   (p = u2 - u1, r = s2 - s1)
//             level 3   ppp = p pp       qq = u1 pp       zz3 = zz12 pp    -
//             level 4   t2 = s1 ppp      t1 = r (qq - x3) -                zzz3 = zzz12 ppp    (x3 = rr - ppp - 2 qq)
//             result    x3               y3 = t1 - t2     zz3              zzz3
//   doubling  level 1   xx = x x         v = u u          -                -                   (u = 2 y)
//             level 2   s = x v          w = u v          zz3 = zz v       -
//             level 3   mm = m m         wy = w y         -                zzz3 = w zzz        (m = 3 xx)
//             level 4   -                t = m (s - x3)   -                -                   (x3 = mm - 2 s)
//             result    x3               y3 = t - wy      zz3              zzz3
//
// Invariants of a quad point: x, y < 2p; zz, zzz < 1.2p; limbs normalized; infinity <=> zz is the literal 0.  Only
// operations with checked contracts (field29.hpp) are used; tests/cpp/quad_check.cpp runs the same code on the host -
// four simulated lanes, bound assertions on - against G1LT::add / dbl.
#pragma once
#include "curve29.hpp"

namespace cap {

// ---- lane policies ---------------------------------------------------------------------------------------------------
// V: what one "instruction" operates on; perm<a, b, c, d>(v): lane q of every quad gets lane (a, b, c, d)[q]'s value;
// sel(v0, v1, v2, v3): lane q keeps v_q; flag<k>(pred): the predicate's value in lane k of the quad, in all four lanes.
#if defined(__HIPCC__)
struct QuadDev {
  using V = fl;
  template <class Fn>
  static __device__ __forceinline__ V map1(Fn f, const V& a) { return f(a); }
  template <class Fn>
  static __device__ __forceinline__ V map2(Fn f, const V& a, const V& b) { return f(a, b); }
  template <int A, int B, int C, int D>
  static __device__ __forceinline__ V perm(const V& a) {
    V r;
#pragma unroll
    for (int i = 0; i < 9; i++)
      r.v[i] = (uint32_t)__builtin_amdgcn_update_dpp((int)a.v[i], (int)a.v[i], A | (B << 2) | (C << 4) | (D << 6), 0xf, 0xf, false);
    return r;
  }
  static __device__ __forceinline__ V sel(const V& v0, const V& v1, const V& v2, const V& v3) {
    const uint32_t q = threadIdx.x & 3;
    V r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = q & 2 ? (q & 1 ? v3.v[i] : v2.v[i]) : (q & 1 ? v1.v[i] : v0.v[i]);
    return r;
  }
  template <int K, class Pred>
  static __device__ __forceinline__ bool flag(Pred pr, const V& a) {
    const int mine = pr(a) ? 1 : 0;
    return __builtin_amdgcn_update_dpp(mine, mine, K | (K << 2) | (K << 4) | (K << 6), 0xf, 0xf, false) != 0;
  }
  template <int K>
  static __device__ __forceinline__ fl lane(const V& a) { return perm<K, K, K, K>(a); }
  static __device__ __forceinline__ V spread(const fl& x) { return x; }
};
#endif

struct fl4 {  // host simulation: the four lanes of one quad
  fl l[4];
};
struct QuadSim {
  using V = fl4;
  template <class Fn>
  static V map1(Fn f, const V& a) {
    V r;
    for (int q = 0; q < 4; q++) r.l[q] = f(a.l[q]);
    return r;
  }
  template <class Fn>
  static V map2(Fn f, const V& a, const V& b) {
    V r;
    for (int q = 0; q < 4; q++) r.l[q] = f(a.l[q], b.l[q]);
    return r;
  }
  template <int A, int B, int C, int D>
  static V perm(const V& a) {
    V r;
    r.l[0] = a.l[A];
    r.l[1] = a.l[B];
    r.l[2] = a.l[C];
    r.l[3] = a.l[D];
    return r;
  }
  static V sel(const V& v0, const V& v1, const V& v2, const V& v3) {
    V r;
    r.l[0] = v0.l[0];
    r.l[1] = v1.l[1];
    r.l[2] = v2.l[2];
    r.l[3] = v3.l[3];
    return r;
  }
  template <int K, class Pred>
  static bool flag(Pred pr, const V& a) { return pr(a.l[K]); }
  template <int K>
  static fl lane(const V& a) { return a.l[K]; }
  static V spread(const fl& x) {
    V r;
    for (int q = 0; q < 4; q++) r.l[q] = x;
    return r;
  }
};

// ---- the group law on quads --------------------------------------------------------------------------------------------
template <class G, class P>
struct QuadG1 {
  using F = typename G::F;
  using V = typename P::V;

  static CAP_HD V mul(const V& a, const V& b) {
    return P::map2([](const fl& x, const fl& y) { return F::mul(x, y); }, a, b);
  }
  static CAP_HD V sub(const V& a, const V& b) {
    return P::map2([](const fl& x, const fl& y) { return F::sub(x, y); }, a, b);
  }
  static CAP_HD V add_lazy(const V& a, const V& b) {
    return P::map2([](const fl& x, const fl& y) { return F::add(x, y); }, a, b);
  }
  static CAP_HD V weak(const V& a) {
    return P::map1([](const fl& x) { return F::weak_reduce(x); }, a);
  }
  static CAP_HD bool is_inf(const V& a) {
    return P::template flag<2>([](const fl& x) { return G::all_zero(x); }, a);
  }
  // every lane's copy of the whole point (the rare paths, and the hand-over to single-lane code)
  static CAP_HD g1x gather(const V& a) {
    g1x r;
    r.x = P::template lane<0>(a);
    r.y = P::template lane<1>(a);
    r.zz = P::template lane<2>(a);
    r.zzz = P::template lane<3>(a);
    return r;
  }
  static CAP_HD V scatter(const g1x& p) {
    return P::sel(P::spread(p.x), P::spread(p.y), P::spread(p.zz), P::spread(p.zzz));
  }

  // a += b.  Operands at infinity are a skip or a copy; equal or opposite operands - never with real data - take the
  // general single-lane addition (`slow`, supplied by the caller so that a kernel holds one out-of-line copy of it).
  template <class Slow>
  static CAP_HD void add(V& a, const V& b, Slow slow) {
    if (is_inf(b)) return;
    if (is_inf(a)) {
      a = b;
      return;
    }
    const V m1 = mul(a, P::template perm<2, 3, 0, 1>(b));                                   // u1, s1, u2, s2
    const V d = sub(P::template perm<2, 3, 2, 3>(m1), P::template perm<0, 1, 0, 1>(m1));    // p, r, p, r  (< 17.1p)
    if (P::template flag<0>([](const fl& x) { return F::is_zero(x); }, d)) {
      a = scatter(slow(gather(a), gather(b)));
      return;
    }
    const V m2 = mul(P::sel(d, d, a, a), P::sel(d, d, b, b));                               // pp, rr, zz12, zzz12
    const V pp = P::template perm<0, 0, 0, 0>(m2);
    const V u1 = P::template perm<0, 0, 0, 0>(m1);
    const V m3 = mul(P::sel(d, u1, m2, m2), pp);                                            // ppp, qq, zz3, -
    const V rr = P::template perm<1, 1, 1, 1>(m2), ppp = P::template perm<0, 0, 0, 0>(m3);
    const V qq = P::template perm<1, 1, 1, 1>(m3);
    // r^2 - ppp - 2 qq, carried once (as G1LT::add_acc): every lane computes it, lane 1 needs it for its operand
    const V x3 = weak(P::map2([](const fl& x, const fl& y) { return F::sub_from_lazy(x, y); },
                              P::map2([](const fl& x, const fl& y) { return F::sub2p_lazy(x, y); }, rr, ppp),
                              add_lazy(qq, qq)));
    const V s1 = P::template perm<1, 1, 1, 1>(m1);
    const V m4 = mul(P::sel(s1, d, m2, m2), P::sel(ppp, sub(qq, x3), ppp, ppp));            // t2, t1, -, zzz3
    const V y3 = weak(sub(m4, P::template perm<0, 0, 0, 0>(m4)));                           // lane 1: t1 - t2
    a = P::sel(x3, y3, m3, m4);
  }
  static CAP_HD void dbl(V& a) {
    if (is_inf(a)) return;
    const V o1 = P::sel(a, add_lazy(a, a), a, a);                                           // x, u = 2y, zz, zzz
    const V m1 = mul(o1, o1);                                                               // xx, v, -, -
    const V m2 = mul(o1, P::template perm<1, 1, 1, 1>(m1));                                 // s, w, zz3, -
    const V xx = P::template perm<0, 0, 0, 0>(m1);
    const V m = P::map1([](const fl& x) { return F::normalize(x); }, add_lazy(add_lazy(xx, xx), xx));
    const V w = P::template perm<1, 1, 1, 1>(m2);
    const V m3 = mul(P::sel(m, w, w, w), P::sel(m, a, a, a));                               // mm, w y, -, zzz3
    const V s = P::template perm<0, 0, 0, 0>(m2);
    const V x3 = weak(sub(P::template perm<0, 0, 0, 0>(m3), add_lazy(s, s)));
    const V t = mul(m, sub(s, x3));
    const V y3 = weak(sub(t, P::template perm<1, 1, 1, 1>(m3)));
    a = P::sel(x3, y3, m2, m3);
  }
};

}  // namespace cap
