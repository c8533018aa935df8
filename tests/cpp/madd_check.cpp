// Host-side check of the accumulation-loop mixed addition (G1L::madd_acc, cap_amd/csrc/curve29.hpp) and of the
// limb-bound contracts its un-carried operands rely on.  Built twice by tests/test_field29_host.py: with g++ (the
// CAP_FL_CHECK assertions) and with clang++ -fsanitize=unsigned-integer-overflow (every 64-bit column sum and every
// 32-bit limb sum trapped on wrap-around; the saturated reference code in field.hpp / curve.hpp is on the ignore list).
#define CAP_FL_CHECK 1
#include "../../cap_amd/csrc/curve29.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace cap;

static bool same_affine(const g1_affine& a32, const g1a& a29) {
  bool inf32 = G1::is_inf(a32), inf29 = G1L::is_inf(a29);
  if (inf32 || inf29) return inf32 == inf29;
  fe x32 = Fq::from_mont(a32.x), y32 = Fq::from_mont(a32.y);
  fe x29 = Fq29::from_mont(a29.x), y29 = Fq29::from_mont(a29.y);
  return Fq::eq(x32, x29) && Fq::eq(y32, y29);
}
static g1a conv(const g1_affine& p) {
  g1a r;
  if (G1::is_inf(p)) { r.x = Fq29::zero(); r.y = Fq29::zero(); return r; }
  r.x = Fq29::canonical(Fq29::from_ext(p.x));
  r.y = Fq29::canonical(Fq29::from_ext(p.y));
  return r;
}
// the loop body of msm_accumulate
template <class G>
static void loop_add(g1x& acc, const g1a& q, bool negate, int* rare) {
  if (G::is_inf(q) || !G::madd_acc(acc, q, negate)) { acc = G::add_mixed(acc, q, negate); (*rare)++; }
}

template <int SCHED>
static int directed_bounds() {
  using F = Fl<FqP29, SCHED>;
  // extreme limb patterns the contracts allow (the values are far outside [0, 2p); only the integer arithmetic is
  // exercised: nothing may wrap)
  fl norm_max, lazy30, lazy15, lazy31;
  for (int i = 0; i < 9; i++) {
    norm_max.v[i] = i < 8 ? (1u << 29) - 1 : 0x30644e * 3;     // normalized limbs, a few p in the top limb
    lazy30.v[i] = i < 8 ? (1u << 30) - 1 : 0x30644e * 3;       // one lazy sum of two
    lazy15.v[i] = i < 8 ? (3u << 29) - 1 : 0x30644e * 3;       // sub2p_lazy output: < 1.5 * 2^30
    lazy31.v[i] = i < 8 ? (1u << 31) - 1 : 0x30644e * 16;      // neg_lazy output: < 2^31
  }
  volatile uint32_t sink = 0;
  fl r = F::mul(lazy30, lazy30); sink += r.v[0];               // the documented contract of mul
  r = F::sqr(lazy30); sink += r.v[0];
  r = F::mul(lazy31, norm_max); sink += r.v[0];                // neg_lazy(q.y) * zzz
  r = F::mul(norm_max, lazy31); sink += r.v[0];
  r = F::mul_add_mul(norm_max, lazy15, lazy30, norm_max); sink += r.v[0];     // r * (qq - x3 + 2p) + (4p - y) * ppp
  r = F::mul_add_mul(norm_max, norm_max, norm_max, norm_max); sink += r.v[0];
  r = F::mul_add_mul(norm_max, lazy15, lazy30, norm_max); sink += r.v[0];     // ... + (2p - s1) * ppp in add_acc
  fl t = F::sub2p_lazy(norm_max, F::weak_reduce(norm_max));
  r = F::sub_from_lazy(t, lazy30); sink += r.v[0];             // (r^2 - ppp + 2p) - 2qq + 16p
  (void)sink;
  return 0;
}

template <int SCHED>
static int chains() {
  using G = G1LT<SCHED>;
  const int N = 400;
  std::vector<g1_affine> pts(N);
  g1_affine g; g.x = Fq::one(); g.y = Fq::dbl(Fq::one());
  g1_xyzz acc = G1::from_affine(g);
  uint32_t s = 12345;
  for (int i = 0; i < N; i++) {
    pts[i] = G1::to_affine(acc);
    s = (uint32_t)(((uint64_t)s * 1103515245ull + 12345ull) % 2147483648ull);   // small LCG, no wrap
    int reps = 1 + (int)(s % 7);
    for (int k = 0; k < reps; k++) acc = G1::add_mixed(acc, g);
    acc = G1::dbl(acc);
  }
  pts[17] = pts[16];                          // duplicate: P + P when they meet
  memset(&pts[33], 0, sizeof(g1_affine));     // a base at infinity
  int bad = 0, rare = 0;
  g1_xyzz A = G1::inf();
  g1x B = G::inf();
  for (int round = 0; round < 3; round++) {
    for (int i = 0; i < N; i++) {
      bool neg = ((i * 7 + round) % 5) == 3;
      g1_affine p = pts[i];
      if (neg) p.y = Fq::neg(p.y);
      A = G1::add_mixed(A, p);
      loop_add<G>(B, conv(pts[i]), neg, &rare);
      if (i == 20 || i == 150) {
        // acc == q (doubling) and then acc == -q (cancellation to infinity), then on from infinity
        g1_affine cur = G1::to_affine(A);
        A = G1::add_mixed(A, cur); loop_add<G>(B, conv(cur), false, &rare);
        cur = G1::to_affine(A);
        g1_affine ncur = cur; ncur.y = Fq::neg(cur.y);
        A = G1::add_mixed(A, ncur); loop_add<G>(B, conv(cur), true, &rare);
        if (!G1::is_inf(A) || !G::is_inf(B)) { bad++; printf("cancellation failed\n"); }
      }
      if ((i % 16) == 0 && !same_affine(G1::to_affine(A), G::to_affine(B))) { bad++; if (bad < 5) printf("mismatch at %d/%d\n", round, i); }
      if ((i % 37) == 0) B = G::load(G::store(B));             // the 32-byte images hold the relaxed y as well
    }
  }
  if (!same_affine(G1::to_affine(A), G::to_affine(B))) { bad++; printf("final mismatch\n"); }
  // the running sums of the bucket reduction: S += B_j, T += S through G::add_acc with the general add() as fallback,
  // over "buckets" that include empty ones, repeated ones (S == B: doubling) and opposite ones
  {
    std::vector<g1_xyzz> bk32;
    std::vector<g1x> bk29;
    for (int i = 0; i < 60; i++) {
      g1_xyzz x = G1::inf();
      g1x y = G::inf();
      int cnt = (i % 9 == 4) ? 0 : 1 + i % 4;                     // every ninth bucket is empty
      for (int k = 0; k < cnt; k++) { x = G1::add_mixed(x, pts[(i * 5 + k) % N]); int r0 = 0; loop_add<G>(y, conv(pts[(i * 5 + k) % N]), false, &r0); }
      bk32.push_back(x);
      bk29.push_back(G::load(G::store(y)));
    }
    bk32[7] = bk32[6]; bk29[7] = bk29[6];                          // S == B at some point is unlikely; equal neighbours at least
    g1_xyzz S32 = G1::inf(), T32 = G1::inf();
    g1x S = G::inf(), T = G::inf();
    int fallbacks = 0;
    for (int j = 59; j >= 0; j--) {
      S32 = G1::add(S32, bk32[j]); T32 = G1::add(T32, S32);
      if (!G::add_acc(S, bk29[j])) { S = G::add(S, bk29[j]); fallbacks++; }
      if (!G::add_acc(T, S)) { T = G::add(T, S); fallbacks++; }
      if (!same_affine(G1::to_affine(S32), G::to_affine(S)) || !same_affine(G1::to_affine(T32), G::to_affine(T))) { bad++; if (bad < 5) printf("running sums differ at bucket %d\n", j); }
    }
    // S + S and S + (-S) through the fallback
    g1x D = S; if (G::add_acc(D, S)) { bad++; printf("add_acc did not refuse a doubling\n"); }
    g1x M = S; M.y = G::F::weak_reduce(G::F::neg(S.y)); D = S; if (G::add_acc(D, M)) { bad++; printf("add_acc did not refuse a cancellation\n"); }
    if (fallbacks < 2) { bad++; printf("fallbacks %d\n", fallbacks); }
  }
  if (rare < 8 || rare > 40) { bad++; printf("unexpected number of special cases: %d\n", rare); }
  return bad;
}

int main() {
  int bad = chains<0>() + chains<1>() + directed_bounds<0>() + directed_bounds<1>();
  printf("bad=%d\n", bad);
  return bad != 0;
}
