// Host-side check of cap_amd/csrc/quad29.hpp: the four-lane point addition / doubling, run on four simulated lanes
// with the field's bound assertions on (CAP_FL_CHECK), against the single-lane G1LT::add / dbl on the same operands:
// random points, chains (results fed back, so the invariants of a quad point are exercised as they evolve), infinity
// on either side, P + P and P - P (the out-of-line path), and the 32-byte memory image in between.
#define CAP_FL_CHECK 1
#include "../../cap_amd/csrc/quad29.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace cap;
using Q = QuadG1<G1L, QuadSim>;

static bool same(const g1x& a, const g1x& b) {
  const bool ia = G1L::is_inf(a), ib = G1L::is_inf(b);
  if (ia || ib) return ia == ib;
  g1a x = G1L::to_affine(a), y = G1L::to_affine(b);
  return Fq29::eq(x.x, y.x) && Fq29::eq(x.y, y.y);
}
static g1a conv(const g1_affine& p) {
  g1a r;
  r.x = Fq29::canonical(Fq29::from_ext(p.x));
  r.y = Fq29::canonical(Fq29::from_ext(p.y));
  return r;
}
static int slow_calls = 0;
static g1x slow(const g1x& a, const g1x& b) {
  slow_calls++;
  return G1L::add(a, b);
}
// through memory, as the kernels hand quad points on: lane q packs / unpacks coordinate q
static fl4 reload(const fl4& v) {
  fl4 r;
  for (int q = 0; q < 4; q++) r.l[q] = Fq29::load(Fq29::pack(v.l[q]));
  return r;
}

int main() {
  const int N = 120;
  std::vector<g1x> pts(N);
  g1_affine g;
  g.x = Fq::one();
  g.y = Fq::dbl(Fq::one());
  g1_xyzz acc = G1::from_affine(g);
  unsigned long long s = 99991;
  for (int i = 0; i < N; i++) {
    // projective representatives with non-trivial zz: sums of a few multiples of the generator
    g1x p = G1L::inf();
    for (int k = 0; k < 3; k++) {
      int reps = 1 + (int)(s % 11);
      s = (s * 1103515245ULL + 12345ULL) % 2147483648ULL;
      for (int j = 0; j < reps; j++) acc = G1::add_mixed(acc, g);
      acc = G1::dbl(acc);
      p = G1L::add_mixed(p, conv(G1::to_affine(acc)));
    }
    pts[i] = p;
  }
  int bad = 0;
  // 1. single additions and doublings
  for (int i = 0; i + 1 < N; i++) {
    fl4 a = Q::scatter(pts[i]);
    Q::add(a, Q::scatter(pts[i + 1]), slow);
    if (!same(Q::gather(a), G1L::add(pts[i], pts[i + 1]))) { bad++; printf("add mismatch %d\n", i); }
    fl4 d = Q::scatter(pts[i]);
    Q::dbl(d);
    if (!same(Q::gather(d), G1L::dbl(pts[i]))) { bad++; printf("dbl mismatch %d\n", i); }
  }
  if (slow_calls) { bad++; printf("the common path fell back %d times\n", slow_calls); }
  // 2. chains: a running sum with doublings thrown in, quad result fed back (and through memory every few steps)
  {
    fl4 a = Q::scatter(G1L::inf());
    g1x ref = G1L::inf();
    for (int i = 0; i < N; i++) {
      Q::add(a, Q::scatter(pts[i]), slow);
      ref = G1L::add(ref, pts[i]);
      if (i % 7 == 3) {
        for (int k = 0; k < 6; k++) { Q::dbl(a); ref = G1L::dbl(ref); }
      }
      if (i % 5 == 2) a = reload(a);
      if (!same(Q::gather(a), ref)) { bad++; if (bad < 6) printf("chain mismatch at %d\n", i); }
    }
    // a tree over the quad sums themselves (both operands products of earlier quad additions)
    std::vector<fl4> t(16);
    std::vector<g1x> r(16);
    for (int i = 0; i < 16; i++) {
      t[i] = Q::scatter(G1L::inf());
      r[i] = G1L::inf();
      for (int k = 0; k < 5; k++) { Q::add(t[i], Q::scatter(pts[i * 5 + k]), slow); r[i] = G1L::add(r[i], pts[i * 5 + k]); }
    }
    for (int d = 8; d >= 1; d >>= 1)
      for (int i = 0; i < d; i++) { Q::add(t[i], t[i + d], slow); r[i] = G1L::add(r[i], r[i + d]); }
    if (!same(Q::gather(t[0]), r[0])) { bad++; printf("tree mismatch\n"); }
  }
  // 3. special cases
  {
    const int before = slow_calls;
    fl4 inf = Q::scatter(G1L::inf());
    fl4 a = Q::scatter(pts[3]);
    Q::add(a, inf, slow);
    if (!same(Q::gather(a), pts[3])) { bad++; printf("X + inf\n"); }
    a = inf;
    Q::add(a, Q::scatter(pts[3]), slow);
    if (!same(Q::gather(a), pts[3])) { bad++; printf("inf + X\n"); }
    a = inf;
    Q::add(a, inf, slow);
    if (!G1L::is_inf(Q::gather(a))) { bad++; printf("inf + inf\n"); }
    a = inf;
    Q::dbl(a);
    if (!G1L::is_inf(Q::gather(a))) { bad++; printf("dbl(inf)\n"); }
    if (slow_calls != before) { bad++; printf("infinity took the slow path\n"); }
    // P + P: a different representative of the same point
    g1x twice = G1L::add(pts[5], pts[6]);
    g1x other = G1L::add(pts[6], pts[5]);
    other = G1L::add(G1L::add(other, pts[7]), G1L::from_affine([&] { g1a n = G1L::to_affine(pts[7]); n.y = Fq29::weak_reduce(Fq29::neg(n.y)); return n; }()));
    a = Q::scatter(twice);
    Q::add(a, Q::scatter(other), slow);
    if (!same(Q::gather(a), G1L::dbl(twice))) { bad++; printf("P + P\n"); }
    // P - P
    g1x neg = other;
    neg.y = Fq29::weak_reduce(Fq29::neg(neg.y));
    a = Q::scatter(twice);
    Q::add(a, Q::scatter(neg), slow);
    if (!G1L::is_inf(Q::gather(a))) { bad++; printf("P - P\n"); }
    if (slow_calls != before + 2) { bad++; printf("expected two slow-path calls, saw %d\n", slow_calls - before); }
  }
  printf("bad=%d\n", bad);
  return bad != 0;
}
