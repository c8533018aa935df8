// Host-side check of cap_amd/csrc/field.hpp: the host's 64-bit-limb multiplication and its Euclidean inversion against the
// 32-bit-limb CIOS and the Fermat inversion the device runs - the SAME bits for any 256-bit inputs, both fields.
#include "../../cap_amd/csrc/field.hpp"
#include <cstdio>
#include <cstring>
using namespace cap;
static unsigned long long s = 88172645463325252ULL;
static uint32_t rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); }
template <class F>
static int run(const char* name) {
  int bad = 0;
  fe edge[8];
  edge[0] = F::zero();
  edge[1] = F::one();
  edge[2] = F::modulus();                     // p itself (non-canonical 0)
  edge[3] = F::sub(F::zero(), F::one());      // p - 1 in Montgomery form
  memset(&edge[4], 0xff, sizeof(fe));         // 2^256 - 1
  edge[5] = F::zero(); edge[5].v[0] = 1;      // plain 1
  edge[6] = F::modulus(); edge[6].v[0] -= 1;  // p - 1 (plain)
  edge[7] = F::modulus(); edge[7].v[0] += 5;  // p + 5
  for (int it = 0; it < 20000; it++) {
    fe a, b;
    for (int i = 0; i < 8; i++) { a.v[i] = rnd(); b.v[i] = rnd(); }
    if (it % 3 == 0) { a.v[7] &= 0x1fffffffu; b.v[7] &= 0x1fffffffu; }   // mostly below p
    if (it < 64) { a = edge[it / 8]; b = edge[it % 8]; }
    fe m1 = F::mul(a, b), m2 = F::mul_inline(a, b);
    if (!F::eq(m1, m2)) { bad++; if (bad < 5) printf("%s mul mismatch at %d\n", name, it); }
    if (it % 10 == 0) {
      fe i1 = F::inv(a), i2 = F::inv_fermat(a);
      // (inv_fermat is defined on canonical inputs; reduce a first for the comparison)
      fe ar = a;
      for (int k = 0; k < 6 && F::geq_mod(ar); k++) (void)F::sub_mod_raw(ar, ar);
      i2 = F::inv_fermat(ar);
      if (!F::eq(i1, i2)) { bad++; if (bad < 5) printf("%s inv mismatch at %d\n", name, it); }
      if (!F::is_zero(ar) && !F::eq(F::mul(i1, ar), F::one())) { bad++; if (bad < 5) printf("%s inv * a != 1 at %d\n", name, it); }
    }
  }
  return bad;
}
int main() {
  int bad = run<Fq>("Fq") + run<Fr>("Fr");
  printf("bad=%d\n", bad);
  return bad != 0;
}
