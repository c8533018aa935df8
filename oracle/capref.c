/* capref.c - single-thread C restatement of the arkworks CPU algorithms on the
 * CAP prove() hot path: Fp256 Montgomery arithmetic, G1 Jacobian formulas,
 * Pippenger VariableBaseMSM and Radix2EvaluationDomain (i)FFT.
 *
 * TEST INFRASTRUCTURE ONLY.  Built into oracle/_build/libcapref.so by
 * oracle/Makefile; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg load it, and only as the checker / the timed CPU baseline.
 * The product (cap_amd/, libcapgpu.so) never links or calls it.
 *
 * PARITY UNPINNED: the reference's arithmetic lives in crates that are absent
 * from /root/reference (ark-ec 0.3.0 Cargo.lock:103-105, ark-poly 0.3.0
 * Cargo.lock:194-196, ark-ff 0.3.0 Cargo.lock:153-155, ark-bn254 0.3.0
 * Cargo.lock:81-83) and its tests hold no vectors for this path.
 * This file restates the published algorithms and is
 * pinned against oracle/bn254.py (Python big integers) and public known
 * answers in tests/test_oracle.py.  It is a restatement, not the arkworks
 * binary, and is labelled "port" wherever it is timed.
 *
 * Call sites in the reference that reach these algorithms:
 *   src/proof/transfer.rs:181-186, src/proof/mint.rs:113, src/proof/freeze.rs:151
 *   (PlonkKzgSnark::prove -> KZG10::commit -> VariableBaseMSM::multi_scalar_mul,
 *    Radix2EvaluationDomain::{ifft,coset_fft,coset_ifft}).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fp;            /* 4 x u64 LE, Montgomery R = 2^256 */
typedef struct { const uint64_t m[4]; uint64_t ninv; const uint64_t r1[4]; const uint64_t r2[4]; } fparams;

static const fparams FQ = {
  {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
  0x87d20782e4866389ULL,
  {0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL},
  {0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL}};
static const fparams FR = {
  {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
  0xc2e1f593efffffffULL,
  {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL},
  {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL}};

/* ---------------- Fp256 (ark-ff 0.3.0 fields/models/mod.rs, "no-carry" CIOS) ------------- */
static inline int fp_is_zero(const fp *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int fp_eq(const fp *a, const fp *b) {
  return ((a->l[0] ^ b->l[0]) | (a->l[1] ^ b->l[1]) | (a->l[2] ^ b->l[2]) | (a->l[3] ^ b->l[3])) == 0;
}
static inline int geq_mod(const fp *a, const fparams *P) {
  for (int i = 3; i >= 0; i--) {
    if (a->l[i] > P->m[i]) return 1;
    if (a->l[i] < P->m[i]) return 0;
  }
  return 1;
}
static inline void sub_mod_inplace(fp *a, const fparams *P) {
  u128 br = 0;
  for (int i = 0; i < 4; i++) {
    u128 d = (u128)a->l[i] - P->m[i] - br;
    a->l[i] = (uint64_t)d;
    br = (d >> 64) & 1;
  }
}
static inline void fp_add(fp *r, const fp *a, const fp *b, const fparams *P) {
  u128 c = 0;
  for (int i = 0; i < 4; i++) { c += (u128)a->l[i] + b->l[i]; r->l[i] = (uint64_t)c; c >>= 64; }
  if (geq_mod(r, P)) sub_mod_inplace(r, P);
}
static inline void fp_sub(fp *r, const fp *a, const fp *b, const fparams *P) {
  u128 br = 0; fp t;
  for (int i = 0; i < 4; i++) { u128 d = (u128)a->l[i] - b->l[i] - br; t.l[i] = (uint64_t)d; br = (d >> 64) & 1; }
  if (br) { u128 c = 0; for (int i = 0; i < 4; i++) { c += (u128)t.l[i] + P->m[i]; t.l[i] = (uint64_t)c; c >>= 64; } }
  *r = t;
}
static inline void fp_neg(fp *r, const fp *a, const fparams *P) {
  if (fp_is_zero(a)) { *r = *a; return; }
  u128 br = 0;
  for (int i = 0; i < 4; i++) { u128 d = (u128)P->m[i] - a->l[i] - br; r->l[i] = (uint64_t)d; br = (d >> 64) & 1; }
}
static inline void fp_dbl(fp *r, const fp *a, const fparams *P) { fp_add(r, a, a, P); }
static inline void fp_mul(fp *r, const fp *a, const fp *b, const fparams *P) {
  uint64_t t[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) { c += (u128)a->l[j] * b->l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[4] = (uint64_t)c;
    uint64_t m = t[0] * P->ninv;
    c = (u128)m * P->m[0] + t[0]; c >>= 64;
    for (int j = 1; j < 4; j++) { c += (u128)m * P->m[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[3] = (uint64_t)c; t[4] = (uint64_t)(c >> 64);
  }
  fp o = {{t[0], t[1], t[2], t[3]}};
  if (geq_mod(&o, P)) sub_mod_inplace(&o, P);
  *r = o;
}
static inline void fp_sqr(fp *r, const fp *a, const fparams *P) { fp_mul(r, a, a, P); }
static void fp_pow(fp *r, const fp *a, const uint64_t e[4], const fparams *P) {
  fp acc; memcpy(acc.l, P->r1, 32);
  for (int i = 3; i >= 0; i--)
    for (int b = 63; b >= 0; b--) {
      fp_sqr(&acc, &acc, P);
      if ((e[i] >> b) & 1) fp_mul(&acc, &acc, a, P);
    }
  *r = acc;
}
static void fp_inv(fp *r, const fp *a, const fparams *P) {
  uint64_t e[4]; u128 br = 2;
  for (int i = 0; i < 4; i++) { u128 d = (u128)P->m[i] - br; e[i] = (uint64_t)d; br = (d >> 64) & 1; }
  fp_pow(r, a, e, P);
}
static inline void fp_one(fp *r, const fparams *P) { memcpy(r->l, P->r1, 32); }
static inline void fp_to_mont(fp *r, const fp *a, const fparams *P) { fp r2; memcpy(r2.l, P->r2, 32); fp_mul(r, a, &r2, P); }
static inline void fp_from_mont(fp *r, const fp *a, const fparams *P) { fp o = {{1, 0, 0, 0}}; fp_mul(r, a, &o, P); }
static void fp_from_u64(fp *r, uint64_t v, const fparams *P) { fp t = {{v, 0, 0, 0}}; fp_to_mont(r, &t, P); }

/* exported scalar ops (ctypes): which = 0 Fq, 1 Fr; op = 'm','a','s','n','i','t'(to_mont),'f'(from_mont) */
void capref_fp_op(int which, int op, const uint64_t *a, const uint64_t *b, uint64_t *out) {
  const fparams *P = which ? &FR : &FQ;
  fp x, y, r; memcpy(x.l, a, 32); memcpy(y.l, b, 32);
  switch (op) {
    case 'm': fp_mul(&r, &x, &y, P); break;
    case 'a': fp_add(&r, &x, &y, P); break;
    case 's': fp_sub(&r, &x, &y, P); break;
    case 'n': fp_neg(&r, &x, P); break;
    case 'i': fp_inv(&r, &x, P); break;
    case 't': fp_to_mont(&r, &x, P); break;
    default:  fp_from_mont(&r, &x, P); break;
  }
  memcpy(out, r.l, 32);
}
void capref_fp_vec_to_mont(int which, uint64_t *a, size_t n) {
  const fparams *P = which ? &FR : &FQ;
  for (size_t i = 0; i < n; i++) { fp x; memcpy(x.l, a + 4 * i, 32); fp_to_mont(&x, &x, P); memcpy(a + 4 * i, x.l, 32); }
}
void capref_fp_vec_from_mont(int which, uint64_t *a, size_t n) {
  const fparams *P = which ? &FR : &FQ;
  for (size_t i = 0; i < n; i++) { fp x; memcpy(x.l, a + 4 * i, 32); fp_from_mont(&x, &x, P); memcpy(a + 4 * i, x.l, 32); }
}

/* ---------------- G1 (ark-ec 0.3.0 short_weierstrass_jacobian.rs) ------------------------ */
typedef struct { fp x, y; } g1a;        /* affine, Montgomery; infinity encoded as (0,0) */
typedef struct { fp x, y, z; } g1j;     /* Jacobian; infinity = z == 0 */

static inline int g1a_is_inf(const g1a *p) { return fp_is_zero(&p->x) && fp_is_zero(&p->y); }
static inline void g1j_set_inf(g1j *p) { fp_one(&p->x, &FQ); fp_one(&p->y, &FQ); memset(&p->z, 0, 32); }
static inline int g1j_is_inf(const g1j *p) { return fp_is_zero(&p->z); }

/* dbl-2009-l (a = 0) */
static void g1j_double(g1j *r, const g1j *p) {
  if (g1j_is_inf(p)) { *r = *p; return; }
  fp a, b, c, d, e, f, t;
  fp_sqr(&a, &p->x, &FQ);
  fp_sqr(&b, &p->y, &FQ);
  fp_sqr(&c, &b, &FQ);
  fp_add(&d, &p->x, &b, &FQ); fp_sqr(&d, &d, &FQ); fp_sub(&d, &d, &a, &FQ); fp_sub(&d, &d, &c, &FQ); fp_dbl(&d, &d, &FQ);
  fp_dbl(&e, &a, &FQ); fp_add(&e, &e, &a, &FQ);
  fp_sqr(&f, &e, &FQ);
  fp z3; fp_mul(&z3, &p->z, &p->y, &FQ); fp_dbl(&z3, &z3, &FQ);
  fp x3; fp_sub(&x3, &f, &d, &FQ); fp_sub(&x3, &x3, &d, &FQ);
  fp y3; fp_sub(&t, &d, &x3, &FQ); fp_mul(&y3, &t, &e, &FQ);
  fp_dbl(&c, &c, &FQ); fp_dbl(&c, &c, &FQ); fp_dbl(&c, &c, &FQ);
  fp_sub(&y3, &y3, &c, &FQ);
  r->x = x3; r->y = y3; r->z = z3;
}
/* madd-2007-bl */
static void g1j_add_mixed(g1j *r, const g1j *p, const g1a *q) {
  if (g1a_is_inf(q)) { *r = *p; return; }
  if (g1j_is_inf(p)) { r->x = q->x; r->y = q->y; fp_one(&r->z, &FQ); return; }
  fp z1z1, u2, s2;
  fp_sqr(&z1z1, &p->z, &FQ);
  fp_mul(&u2, &q->x, &z1z1, &FQ);
  fp_mul(&s2, &p->z, &q->y, &FQ); fp_mul(&s2, &s2, &z1z1, &FQ);
  if (fp_eq(&p->x, &u2) && fp_eq(&p->y, &s2)) { g1j_double(r, p); return; }
  fp h, hh, i, j, rr, v, t;
  fp_sub(&h, &u2, &p->x, &FQ);
  fp_sqr(&hh, &h, &FQ);
  fp_dbl(&i, &hh, &FQ); fp_dbl(&i, &i, &FQ);
  fp_mul(&j, &h, &i, &FQ);
  fp_sub(&rr, &s2, &p->y, &FQ); fp_dbl(&rr, &rr, &FQ);
  fp_mul(&v, &p->x, &i, &FQ);
  fp x3; fp_sqr(&x3, &rr, &FQ); fp_sub(&x3, &x3, &j, &FQ); fp_sub(&x3, &x3, &v, &FQ); fp_sub(&x3, &x3, &v, &FQ);
  fp y3; fp_mul(&j, &p->y, &j, &FQ); fp_dbl(&j, &j, &FQ);
  fp_sub(&t, &v, &x3, &FQ); fp_mul(&y3, &t, &rr, &FQ); fp_sub(&y3, &y3, &j, &FQ);
  fp z3; fp_add(&z3, &p->z, &h, &FQ); fp_sqr(&z3, &z3, &FQ); fp_sub(&z3, &z3, &z1z1, &FQ); fp_sub(&z3, &z3, &hh, &FQ);
  r->x = x3; r->y = y3; r->z = z3;
}
/* add-2007-bl */
static void g1j_add(g1j *r, const g1j *p, const g1j *q) {
  if (g1j_is_inf(p)) { *r = *q; return; }
  if (g1j_is_inf(q)) { *r = *p; return; }
  fp z1z1, z2z2, u1, u2, s1, s2;
  fp_sqr(&z1z1, &p->z, &FQ); fp_sqr(&z2z2, &q->z, &FQ);
  fp_mul(&u1, &p->x, &z2z2, &FQ); fp_mul(&u2, &q->x, &z1z1, &FQ);
  fp_mul(&s1, &p->y, &q->z, &FQ); fp_mul(&s1, &s1, &z2z2, &FQ);
  fp_mul(&s2, &q->y, &p->z, &FQ); fp_mul(&s2, &s2, &z1z1, &FQ);
  if (fp_eq(&u1, &u2) && fp_eq(&s1, &s2)) { g1j_double(r, p); return; }
  fp h, i, j, rr, v, t;
  fp_sub(&h, &u2, &u1, &FQ);
  fp_dbl(&i, &h, &FQ); fp_sqr(&i, &i, &FQ);
  fp_mul(&j, &h, &i, &FQ);
  fp_sub(&rr, &s2, &s1, &FQ); fp_dbl(&rr, &rr, &FQ);
  fp_mul(&v, &u1, &i, &FQ);
  fp x3; fp_sqr(&x3, &rr, &FQ); fp_sub(&x3, &x3, &j, &FQ); fp_sub(&x3, &x3, &v, &FQ); fp_sub(&x3, &x3, &v, &FQ);
  fp y3; fp_sub(&t, &v, &x3, &FQ); fp_mul(&y3, &t, &rr, &FQ);
  fp_mul(&s1, &s1, &j, &FQ); fp_dbl(&s1, &s1, &FQ); fp_sub(&y3, &y3, &s1, &FQ);
  fp z3; fp_add(&z3, &p->z, &q->z, &FQ); fp_sqr(&z3, &z3, &FQ); fp_sub(&z3, &z3, &z1z1, &FQ); fp_sub(&z3, &z3, &z2z2, &FQ);
  fp_mul(&z3, &z3, &h, &FQ);
  r->x = x3; r->y = y3; r->z = z3;
}
static void g1j_to_affine(g1a *r, const g1j *p) {
  if (g1j_is_inf(p)) { memset(r, 0, sizeof(*r)); return; }
  fp zi, zi2, zi3;
  fp_inv(&zi, &p->z, &FQ); fp_sqr(&zi2, &zi, &FQ); fp_mul(&zi3, &zi2, &zi, &FQ);
  fp_mul(&r->x, &p->x, &zi2, &FQ); fp_mul(&r->y, &p->y, &zi3, &FQ);
}
/* scalar (canonical 4xu64) * affine point */
static void g1_mul_scalar(g1j *r, const g1a *p, const uint64_t k[4]) {
  g1j acc; g1j_set_inf(&acc);
  for (int i = 3; i >= 0; i--)
    for (int b = 63; b >= 0; b--) {
      g1j_double(&acc, &acc);
      if ((k[i] >> b) & 1) g1j_add_mixed(&acc, &acc, p);
    }
  *r = acc;
}

void capref_g1_to_affine(const uint64_t *jac, uint64_t *aff) {
  g1j p; memcpy(&p, jac, 96); g1a a; g1j_to_affine(&a, &p); memcpy(aff, &a, 64);
}
void capref_g1_mul(const uint64_t *aff, const uint64_t *k, uint64_t *out_jac) {
  g1a p; memcpy(&p, aff, 64); g1j r; g1_mul_scalar(&r, &p, k); memcpy(out_jac, &r, 96);
}
void capref_g1_add(const uint64_t *jac_a, const uint64_t *jac_b, uint64_t *out_jac) {
  g1j a, b, r; memcpy(&a, jac_a, 96); memcpy(&b, jac_b, 96); g1j_add(&r, &a, &b); memcpy(out_jac, &r, 96);
}

/* ---------------- VariableBaseMSM::multi_scalar_mul (ark-ec 0.3.0 msm/variable_base.rs) --- */
static unsigned ceil_log2(size_t n) { unsigned l = 0; while (((size_t)1 << l) < n) l++; return l; }
unsigned capref_msm_window(size_t n) { return n < 32 ? 3 : (ceil_log2(n) * 69) / 100 + 2; }

/* bases: n x 64 B affine (Montgomery; (0,0) = infinity); scalars: n x 32 B canonical; out: Jacobian 96 B.
 * c_override = 0 -> arkworks' window rule. */
void capref_msm_g1(const uint64_t *bases, const uint64_t *scalars, size_t n, unsigned c_override, uint64_t *out_jac) {
  unsigned c = c_override ? c_override : capref_msm_window(n);
  const unsigned num_bits = 254;
  size_t nbuckets = ((size_t)1 << c) - 1;
  g1j *buckets = (g1j *)malloc(sizeof(g1j) * nbuckets);
  unsigned nwin = (num_bits + c - 1) / c;
  g1j *wsum = (g1j *)malloc(sizeof(g1j) * nwin);
  const g1a *B = (const g1a *)bases;
  unsigned wi = 0;
  for (unsigned w = 0; w < num_bits; w += c, wi++) {
    g1j res; g1j_set_inf(&res);
    for (size_t b = 0; b < nbuckets; b++) g1j_set_inf(&buckets[b]);
    for (size_t i = 0; i < n; i++) {
      const uint64_t *k = scalars + 4 * i;
      if ((k[0] | k[1] | k[2] | k[3]) == 0) continue;
      if (k[0] == 1 && (k[1] | k[2] | k[3]) == 0) {
        if (w == 0) g1j_add_mixed(&res, &res, &B[i]);
        continue;
      }
      /* (k >> w) mod 2^c */
      unsigned limb = w / 64, off = w % 64;
      uint64_t d = k[limb] >> off;
      if (off + c > 64 && limb + 1 < 4) d |= k[limb + 1] << (64 - off);
      d &= ((uint64_t)1 << c) - 1;
      if (d) g1j_add_mixed(&buckets[d - 1], &buckets[d - 1], &B[i]);
    }
    g1j running; g1j_set_inf(&running);
    for (size_t b = nbuckets; b-- > 0;) {
      g1j_add(&running, &running, &buckets[b]);
      g1j_add(&res, &res, &running);
    }
    wsum[wi] = res;
  }
  g1j total; g1j_set_inf(&total);
  for (unsigned i = nwin; i-- > 1;) {
    g1j_add(&total, &total, &wsum[i]);
    for (unsigned k = 0; k < c; k++) g1j_double(&total, &total);
  }
  g1j_add(&total, &total, &wsum[0]);
  memcpy(out_jac, &total, 96);
  free(buckets); free(wsum);
}

/* bases[i] = [s_i] G for given canonical scalars (test / bench SRS generation) */
void capref_g1_fixed_base_batch(const uint64_t *scalars, size_t n, uint64_t *out_aff) {
  g1a G; fp_one(&G.x, &FQ); fp_from_u64(&G.y, 2, &FQ);
  /* 8-bit windowed table of G: T[w][d] = d * 256^w * G, as Jacobian -> affine lazily */
  static g1a *T = NULL;
  if (!T) {
    T = (g1a *)malloc(sizeof(g1a) * 32 * 256);
    g1j base; base.x = G.x; base.y = G.y; fp_one(&base.z, &FQ);
    for (int w = 0; w < 32; w++) {
      g1j acc; g1j_set_inf(&acc);
      memset(&T[w * 256], 0, sizeof(g1a));
      for (int d = 1; d < 256; d++) { g1j_add(&acc, &acc, &base); g1j_to_affine(&T[w * 256 + d], &acc); }
      g1j_add(&acc, &acc, &base); base = acc;
    }
  }
  for (size_t i = 0; i < n; i++) {
    g1j acc; g1j_set_inf(&acc);
    const uint8_t *kb = (const uint8_t *)(scalars + 4 * i);
    for (int w = 0; w < 32; w++) if (kb[w]) g1j_add_mixed(&acc, &acc, &T[w * 256 + kb[w]]);
    g1a a; g1j_to_affine(&a, &acc); memcpy(out_aff + 8 * i, &a, 64);
  }
}

/* ---------------- Radix2EvaluationDomain (ark-poly 0.3.0 domain/radix2/fft.rs) ----------- */
static const uint64_t ROOT28[4] = {0x9bd61b6e725b19f0ULL, 0x402d111e41112ed4ULL, 0x00e0a7eb8ef62abcULL, 0x2a3c09f0a58a7e85ULL};

static void fr_root_of_unity(fp *w, unsigned log_n) {
  fp r; fp t = {{ROOT28[0], ROOT28[1], ROOT28[2], ROOT28[3]}};
  fp_to_mont(&r, &t, &FR);
  for (unsigned i = log_n; i < 28; i++) fp_sqr(&r, &r, &FR);
  *w = r;
}
static inline size_t bitrev(size_t x, unsigned bits) {
  size_t r = 0;
  for (unsigned i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
  return r;
}
/* in-place, natural order in/out; data = 2^log_n Montgomery Fr.
 * dir 0: forward; 1: inverse (includes n^-1). coset 1: generator 5 (scale before fft / after ifft). */
void capref_ntt_fr(uint64_t *data, unsigned log_n, int dir, int coset) {
  size_t n = (size_t)1 << log_n;
  fp *a = (fp *)data;
  fp g, x;
  if (coset && !dir) {
    fp_from_u64(&g, 5, &FR); fp_one(&x, &FR);
    for (size_t i = 0; i < n; i++) { fp_mul(&a[i], &a[i], &x, &FR); fp_mul(&x, &x, &g, &FR); }
  }
  fp omega; fr_root_of_unity(&omega, log_n);
  if (dir) fp_inv(&omega, &omega, &FR);
  /* bit-reverse then DIT butterflies */
  for (size_t i = 0; i < n; i++) { size_t j = bitrev(i, log_n); if (i < j) { fp t = a[i]; a[i] = a[j]; a[j] = t; } }
  fp *tw = (fp *)malloc(sizeof(fp) * (n / 2 + 1));
  for (unsigned s = 1; s <= log_n; s++) {
    size_t m = (size_t)1 << s, h = m >> 1;
    fp wm = omega;
    for (unsigned k = s; k < log_n; k++) fp_sqr(&wm, &wm, &FR);
    fp_one(&tw[0], &FR);
    for (size_t k = 1; k < h; k++) fp_mul(&tw[k], &tw[k - 1], &wm, &FR);
    for (size_t base = 0; base < n; base += m)
      for (size_t k = 0; k < h; k++) {
        fp t; fp_mul(&t, &a[base + k + h], &tw[k], &FR);
        fp u = a[base + k];
        fp_add(&a[base + k], &u, &t, &FR);
        fp_sub(&a[base + k + h], &u, &t, &FR);
      }
  }
  free(tw);
  if (dir) {
    fp ninv; fp_from_u64(&ninv, (uint64_t)n, &FR); fp_inv(&ninv, &ninv, &FR);
    if (coset) {
      fp gi; fp_from_u64(&gi, 5, &FR); fp_inv(&gi, &gi, &FR);
      x = ninv;
      for (size_t i = 0; i < n; i++) { fp_mul(&a[i], &a[i], &x, &FR); fp_mul(&x, &x, &gi, &FR); }
    } else {
      for (size_t i = 0; i < n; i++) fp_mul(&a[i], &a[i], &ninv, &FR);
    }
  }
}

/* polynomial evaluation (Horner), Montgomery in/out */
void capref_poly_eval_fr(const uint64_t *coeffs, size_t n, const uint64_t *x, uint64_t *out) {
  fp acc; memset(&acc, 0, 32); fp xx; memcpy(&xx, x, 32);
  const fp *c = (const fp *)coeffs;
  for (size_t i = n; i-- > 0;) { fp_mul(&acc, &acc, &xx, &FR); fp_add(&acc, &acc, &c[i], &FR); }
  memcpy(out, &acc, 32);
}

/* ---------------- SplitMix64 inputs shared with oracle/bn254.py ------------------------- */
static uint64_t sm64_next(uint64_t *s) {
  uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
/* 256-bit value mod m by conditional subtraction of m << k (value < 2^256 < 8m) */
static void reduce256(uint64_t v[4], const fparams *P) {
  for (int k = 2; k >= 0; k--) {
    /* t = m << k (fits: m < 2^254) */
    uint64_t t[4];
    for (int i = 3; i >= 0; i--) t[i] = (P->m[i] << k) | ((k && i) ? (P->m[i - 1] >> (64 - k)) : 0);
    int ge = 1;
    for (int i = 3; i >= 0; i--) { if (v[i] > t[i]) break; if (v[i] < t[i]) { ge = 0; break; } }
    if (ge) { u128 br = 0; for (int i = 0; i < 4; i++) { u128 d = (u128)v[i] - t[i] - br; v[i] = (uint64_t)d; br = (d >> 64) & 1; } }
  }
}
/* out: n canonical field elements (which: 0 Fq, 1 Fr); mont != 0 -> converted to Montgomery form */
void capref_random_field(uint64_t seed, int which, int mont, uint64_t *out, size_t n) {
  const fparams *P = which ? &FR : &FQ;
  uint64_t s = seed;
  for (size_t i = 0; i < n; i++) {
    uint64_t v[4];
    for (int j = 0; j < 4; j++) v[j] = sm64_next(&s);
    reduce256(v, P);
    fp x = {{v[0], v[1], v[2], v[3]}};
    if (mont) fp_to_mont(&x, &x, P);
    memcpy(out + 4 * i, x.l, 32);
  }
}
