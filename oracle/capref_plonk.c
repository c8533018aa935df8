/* capref_plonk.c - single-thread C restatement of jf-plonk's TurboPlonk prover on top of
 * capref.c's arkworks-style primitives (Pippenger MSM, radix-2 FFT).
 *
 * TEST INFRASTRUCTURE ONLY (checker at full size + the timed CPU "port" baseline of bench.py).
 * PARITY UNPINNED: restates `PlonkKzgSnark::{preprocess, prove}` of jf-plonk 0.1.2 @ bcd92b2
 * (/root/reference/Cargo.lock:992-994; call sites src/proof/transfer.rs:133, :181-186) from
 * SURVEY.md Appendix A; the crate source is not under /root/reference.  It is pinned against
 * oracle/plonk.py (Python integers, verifier-checked) in tests/test_oracle.py.
 *
 * Follows the reference schedule literally: per proof 7 iFFT(n), 25 coset FFT(8n) (the 18
 * selector / sigma polynomials are re-transformed for every proof), 1 coset iFFT(8n), 13 MSMs.
 */
/* capref.c is compiled into this translation unit so that the field arithmetic inlines (a fair
 * single-thread baseline must not pay a function call per multiplication). */
#include "capref.c"

typedef fp fr;
#define NW 5
#define NS 13

static fr F_ONE, F_ZERO;
static inline fr fmul(fr a, fr b) { fr r; fp_mul(&r, &a, &b, &FR); return r; }
static inline fr fadd(fr a, fr b) { fr r; fp_add(&r, &a, &b, &FR); return r; }
static inline fr fsub(fr a, fr b) { fr r; fp_sub(&r, &a, &b, &FR); return r; }
static inline fr fneg(fr a) { fr r; fp_neg(&r, &a, &FR); return r; }
static inline fr finv(fr a) { fr r; fp_inv(&r, &a, &FR); return r; }
static inline fr fto_mont(fr a) { fr r; fp_to_mont(&r, &a, &FR); return r; }
static inline fr ffrom_mont(fr a) { fr r; fp_from_mont(&r, &a, &FR); return r; }
static inline fr fsqr(fr a) { return fmul(a, a); }
static fr ffrom_u64(uint64_t v) { fr t = {{v, 0, 0, 0}}; return fto_mont(t); }
static fr fpow_u64(fr a, uint64_t e) {
  fr r = F_ONE;
  for (int b = 63; b >= 0; b--) { r = fsqr(r); if ((e >> b) & 1) r = fmul(r, a); }
  return r;
}
static int fis_zero(fr a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }

static const uint64_t K_CANON[NW][4] = {
    {1, 0, 0, 0},
    {0x5da4fb7bb1301d4aULL, 0x73ca6c94813f8583ULL, 0xc4e12a44e110404cULL, 0x2f8dd1f1a7583c42ULL},
    {0x77f010424afeb025ULL, 0xa828a3703b311d0fULL, 0xeaa8fe837060498bULL, 0x1ee678a0470a75a6ULL},
    {0x66905a6895790c0aULL, 0x950b1db26d5c82d6ULL, 0x0a087c03e29c968bULL, 0x2042a587a90c187bULL},
    {0xeb9222db7c81e881ULL, 0x8f739da5d8d40dd3ULL, 0xdf57b799969dea1cULL, 0x2e2b91456103698aULL}};
static fr root_of_unity(unsigned log_n) { fr w; fr_root_of_unity(&w, log_n); return w; }

/* ---- Keccak-256 + SolidityTranscript ------------------------------------------------------------------ */
static void keccak_f(uint64_t st[25]) {
  static const uint64_t RC[24] = {
      0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL,
      0x000000000000808BULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
      0x000000000000008AULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000AULL,
      0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
      0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800AULL, 0x800000008000000AULL,
      0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
  static const int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
  for (int rnd = 0; rnd < 24; rnd++) {
    uint64_t C[5], D[5], B[25];
    for (int x = 0; x < 5; x++) C[x] = st[x] ^ st[x + 5] ^ st[x + 10] ^ st[x + 15] ^ st[x + 20];
    for (int x = 0; x < 5; x++) { uint64_t c1 = C[(x + 1) % 5]; D[x] = C[(x + 4) % 5] ^ ((c1 << 1) | (c1 >> 63)); }
    for (int i = 0; i < 25; i++) st[i] ^= D[i % 5];
    for (int x = 0; x < 5; x++)
      for (int y = 0; y < 5; y++) {
        int r = ROT[x + 5 * y]; uint64_t v = st[x + 5 * y];
        B[y + 5 * ((2 * x + 3 * y) % 5)] = r ? ((v << r) | (v >> (64 - r))) : v;
      }
    for (int y = 0; y < 5; y++)
      for (int x = 0; x < 5; x++) st[x + 5 * y] = B[x + 5 * y] ^ ((~B[(x + 1) % 5 + 5 * y]) & B[(x + 2) % 5 + 5 * y]);
    st[0] ^= RC[rnd];
  }
}
void capref_keccak256(const uint8_t *data, size_t len, uint8_t out[32]) {
  uint64_t st[25]; memset(st, 0, sizeof(st));
  size_t off = 0;
  while (len - off >= 136) {
    for (int i = 0; i < 17; i++) { uint64_t w; memcpy(&w, data + off + 8 * i, 8); st[i] ^= w; }
    keccak_f(st); off += 136;
  }
  uint8_t last[136]; memset(last, 0, 136); memcpy(last, data + off, len - off);
  last[len - off] ^= 0x01; last[135] ^= 0x80;
  for (int i = 0; i < 17; i++) { uint64_t w; memcpy(&w, last + 8 * i, 8); st[i] ^= w; }
  keccak_f(st);
  memcpy(out, st, 32);
}
typedef struct { uint8_t state[64]; uint8_t *buf; size_t len, cap; } transcript;
static void tr_init(transcript *t) { memset(t->state, 0, 64); t->cap = 4096; t->len = 0; t->buf = (uint8_t *)malloc(t->cap); }
static void tr_append(transcript *t, const void *p, size_t n) {
  if (t->len + n > t->cap) { while (t->len + n > t->cap) t->cap *= 2; t->buf = (uint8_t *)realloc(t->buf, t->cap); }
  memcpy(t->buf + t->len, p, n); t->len += n;
}
static void tr_append_fr(transcript *t, fr a_mont) { fr c = ffrom_mont(a_mont); tr_append(t, c.l, 32); }
/* ark-serialize 0.3 compressed G1 from an affine Montgomery point (8 words), (0,0) = infinity */
static void tr_append_g1(transcript *t, const uint64_t aff[8]) {
  uint8_t b[32];
  int inf = 1; for (int i = 0; i < 8; i++) if (aff[i]) inf = 0;
  if (inf) { memset(b, 0, 32); b[31] |= 0x40; tr_append(t, b, 32); return; }
  uint64_t x[4], y[4], ny[4], nym[4];
  capref_fp_op(0, 'f', aff, aff, x);
  capref_fp_op(0, 'f', aff + 4, aff + 4, y);
  capref_fp_op(0, 'n', aff + 4, aff + 4, nym);
  capref_fp_op(0, 'f', nym, nym, ny);
  memcpy(b, x, 32);
  int larger = 0;
  for (int i = 3; i >= 0; i--) if (y[i] != ny[i]) { larger = y[i] > ny[i]; break; }
  if (larger) b[31] |= 0x80;
  tr_append(t, b, 32);
}
static fr tr_challenge(transcript *t) {
  size_t n = 64 + t->len + 1;
  uint8_t *in = (uint8_t *)malloc(n);
  memcpy(in, t->state, 64); memcpy(in + 64, t->buf, t->len);
  uint8_t h[64];
  in[n - 1] = 0; capref_keccak256(in, n, h);
  in[n - 1] = 1; capref_keccak256(in, n, h + 32);
  free(in);
  memcpy(t->state, h, 64);
  /* from_le_bytes_mod_order(h[0..48]) */
  fr lo, hi = {{0, 0, 0, 0}};
  memcpy(lo.l, h, 32); memcpy(hi.l, h + 32, 16);
  /* reduce lo (any 256-bit value) by Montgomery-multiplying with R^2: lo * R */
  fr lo_m = fto_mont(lo);
  fr two256 = F_ONE;                       /* Montgomery form of 2^256 mod r is R^2 mod r = to_mont(one_mont) */
  two256 = fto_mont(F_ONE);
  fr hi_m = fmul(fto_mont(hi), two256);
  return fadd(lo_m, hi_m);
}

/* ---- proving key ------------------------------------------------------------------------------------------- */
typedef struct {
  size_t n, m, num_inputs; unsigned log_n, log_m;
  uint64_t *srs;            /* (n+3) x 8 words */
  fr *coef;                 /* [18][n] */
  fr *sig_eval;             /* [5][n] */
  uint64_t vk_comms[18][8];
  fr k[NW];
} pkey;

static void commit(const pkey *K, const fr *coeffs, size_t len, uint64_t out_aff[8]) {
  /* KZG10::commit: skip leading (high-degree) zeros, convert to canonical integers, MSM */
  while (len > 0 && fis_zero(coeffs[len - 1])) len--;
  uint64_t *sc = (uint64_t *)malloc(32 * (len ? len : 1));
  memcpy(sc, coeffs, 32 * len);
  capref_fp_vec_from_mont(1, sc, len);
  uint64_t jac[12];
  capref_msm_g1(K->srs, sc, len, 0, jac);
  capref_g1_to_affine(jac, out_aff);
  free(sc);
}

void *capref_plonk_preprocess(const uint64_t *srs_bases, size_t n, size_t num_inputs, const uint64_t *selectors,
                              const uint64_t *sigma_evals, uint64_t *vk_comms_out /* 18*8 */) {
  fr one = {{1, 0, 0, 0}};
  F_ONE = fto_mont(one); memset(&F_ZERO, 0, sizeof(F_ZERO));
  pkey *K = (pkey *)calloc(1, sizeof(pkey));
  K->n = n; K->m = 8 * n; K->num_inputs = num_inputs;
  while (((size_t)1 << K->log_n) < n) K->log_n++;
  K->log_m = K->log_n + 3;
  K->srs = (uint64_t *)malloc(64 * (n + 3)); memcpy(K->srs, srs_bases, 64 * (n + 3));
  K->coef = (fr *)malloc(sizeof(fr) * 18 * n);
  K->sig_eval = (fr *)malloc(sizeof(fr) * NW * n);
  memcpy(K->coef, selectors, 32 * NS * n);
  memcpy(K->coef + (size_t)NS * n, sigma_evals, 32 * NW * n);
  memcpy(K->sig_eval, sigma_evals, 32 * NW * n);
  for (int i = 0; i < NW; i++) { fr t; memcpy(t.l, K_CANON[i], 32); K->k[i] = fto_mont(t); }
  for (int i = 0; i < 18; i++) {
    capref_ntt_fr((uint64_t *)(K->coef + (size_t)i * n), K->log_n, 1, 0);
    commit(K, K->coef + (size_t)i * n, n, K->vk_comms[i]);
  }
  if (vk_comms_out) memcpy(vk_comms_out, K->vk_comms, sizeof(K->vk_comms));
  return K;
}
void capref_plonk_free(void *pk) {
  pkey *K = (pkey *)pk;
  if (!K) return;
  free(K->srs); free(K->coef); free(K->sig_eval); free(K);
}

static fr *coset_fft_padded(const fr *coeffs, size_t len, size_t m, unsigned log_m) {
  fr *a = (fr *)calloc(m, sizeof(fr));
  memcpy(a, coeffs, sizeof(fr) * len);
  capref_ntt_fr((uint64_t *)a, log_m, 0, 1);
  return a;
}
static void poly_add_scaled(fr *acc, const fr *p, size_t len, fr s) {
  for (size_t i = 0; i < len; i++) acc[i] = fadd(acc[i], fmul(s, p[i]));
}
/* quotient of f / (X - a), remainder dropped; q has len-1 coefficients */
static void divide_linear(const fr *f, size_t len, fr a, fr *q) {
  fr carry = F_ZERO;
  for (size_t k = len - 1; k >= 1; k--) { carry = fadd(f[k], fmul(a, carry)); q[k - 1] = carry; }
}

/* proof_out: 13 x 8 words (wires[5], prod_perm, split_quot[5], opening, shifted opening) then 10 x 4 words
 * (wires_evals[5], wire_sigma_evals[4], perm_next_eval) - the layout of capgpu_proof.  Returns 0, or -7 when
 * the quotient has the wrong degree (unsatisfied circuit). */
int capref_plonk_prove(void *pk, const uint64_t *wires_in, const uint64_t *pub_inputs, const uint8_t *ext_msg,
                       size_t ext_len, const uint64_t *blinders_in, uint64_t *proof_out) {
  pkey *K = (pkey *)pk;
  const size_t n = K->n, m = K->m, L = K->num_inputs;
  const fr *wires = (const fr *)wires_in, *bl = (const fr *)blinders_in, *pub = (const fr *)pub_inputs;
  uint64_t(*comm_out)[8] = (uint64_t(*)[8])proof_out;
  fr *eval_out = (fr *)(proof_out + 13 * 8);
  const fr omega = root_of_unity(K->log_n);
  transcript tr; tr_init(&tr);
  if (ext_msg && ext_len) tr_append(&tr, ext_msg, ext_len);
  { uint64_t v = 254; tr_append(&tr, &v, 8); v = n; tr_append(&tr, &v, 8); v = L; tr_append(&tr, &v, 8); }
  for (int i = 0; i < NW; i++) tr_append_fr(&tr, K->k[i]);
  for (int i = 0; i < 18; i++) tr_append_g1(&tr, K->vk_comms[i]);
  for (size_t i = 0; i < L; i++) tr_append_fr(&tr, pub[i]);

  /* round 1 */
  fr *wpoly[NW];
  for (int i = 0; i < NW; i++) {
    wpoly[i] = (fr *)calloc(n + 3, sizeof(fr));
    memcpy(wpoly[i], wires + (size_t)i * n, sizeof(fr) * n);
    capref_ntt_fr((uint64_t *)wpoly[i], K->log_n, 1, 0);
    for (int t = 0; t < 2; t++) {
      wpoly[i][t] = fsub(wpoly[i][t], bl[2 * i + t]);
      wpoly[i][n + t] = fadd(wpoly[i][n + t], bl[2 * i + t]);
    }
    commit(K, wpoly[i], n + 2, comm_out[i]);
    tr_append_g1(&tr, comm_out[i]);
  }
  fr *pi = (fr *)calloc(n, sizeof(fr));
  memcpy(pi, pub, sizeof(fr) * L);
  capref_ntt_fr((uint64_t *)pi, K->log_n, 1, 0);
  (void)tr_challenge(&tr); /* plookup tau */
  /* round 2 */
  const fr beta = tr_challenge(&tr), gamma = tr_challenge(&tr);
  fr *zpoly = (fr *)calloc(n + 3, sizeof(fr));
  {
    fr *num = (fr *)malloc(sizeof(fr) * n), *den = (fr *)malloc(sizeof(fr) * n);
    fr x = F_ONE;
    for (size_t j = 0; j < n; j++) {
      fr a = F_ONE, b = F_ONE, bx = fmul(beta, x);
      for (int i = 0; i < NW; i++) {
        fr wg = fadd(wires[(size_t)i * n + j], gamma);
        a = fmul(a, fadd(wg, fmul(K->k[i], bx)));
        b = fmul(b, fadd(wg, fmul(beta, K->sig_eval[(size_t)i * n + j])));
      }
      num[j] = a; den[j] = b;
      x = fmul(x, omega);
    }
    /* batch inversion of the denominators */
    fr *pre = (fr *)malloc(sizeof(fr) * n);
    fr acc = F_ONE;
    for (size_t j = 0; j < n; j++) { pre[j] = acc; acc = fmul(acc, den[j]); }
    fr inv = finv(acc);
    for (size_t j = n; j-- > 0;) { fr d = den[j]; den[j] = fmul(inv, pre[j]); inv = fmul(inv, d); }
    zpoly[0] = F_ONE;
    for (size_t j = 0; j + 1 < n; j++) zpoly[j + 1] = fmul(zpoly[j], fmul(num[j], den[j]));
    free(num); free(den); free(pre);
  }
  capref_ntt_fr((uint64_t *)zpoly, K->log_n, 1, 0);
  for (int t = 0; t < 3; t++) { zpoly[t] = fsub(zpoly[t], bl[10 + t]); zpoly[n + t] = fadd(zpoly[n + t], bl[10 + t]); }
  commit(K, zpoly, n + 3, comm_out[5]);
  tr_append_g1(&tr, comm_out[5]);
  /* round 3 */
  const fr alpha = tr_challenge(&tr), alpha2 = fsqr(alpha);
  fr *pkc[18], *wc[NW];
  for (int i = 0; i < 18; i++) pkc[i] = coset_fft_padded(K->coef + (size_t)i * n, n, m, K->log_m);
  for (int i = 0; i < NW; i++) wc[i] = coset_fft_padded(wpoly[i], n + 2, m, K->log_m);
  fr *zc = coset_fft_padded(zpoly, n + 3, m, K->log_m);
  fr *pic = coset_fft_padded(pi, n, m, K->log_m);
  fr *t = (fr *)malloc(sizeof(fr) * m);
  {
    const fr g = ffrom_u64(5), wm = root_of_unity(K->log_m), nm = ffrom_u64((uint64_t)n);
    fr zh_inv[8];
    { fr x = fpow_u64(g, n), w8 = fpow_u64(wm, n);
      for (int i = 0; i < 8; i++) { zh_inv[i] = finv(fsub(x, F_ONE)); x = fmul(x, w8); } }
    /* 1 / (n (x - 1)) by batch inversion over the coset */
    fr *d = (fr *)malloc(sizeof(fr) * m), *pre = (fr *)malloc(sizeof(fr) * m);
    { fr x = g, acc = F_ONE;
      for (size_t i = 0; i < m; i++) { d[i] = fmul(nm, fsub(x, F_ONE)); pre[i] = acc; acc = fmul(acc, d[i]); x = fmul(x, wm); }
      fr inv = finv(acc);
      for (size_t i = m; i-- > 0;) { fr di = d[i]; d[i] = fmul(inv, pre[i]); inv = fmul(inv, di); } }
    fr x = g;
    for (size_t i = 0; i < m; i++) {
      fr w[NW]; for (int j = 0; j < NW; j++) w[j] = wc[j][i];
      fr acc = fadd(pkc[11][i], pic[i]);
      for (int j = 0; j < 4; j++) {
        acc = fadd(acc, fmul(pkc[j][i], w[j]));
        fr w2 = fsqr(w[j]);
        acc = fadd(acc, fmul(pkc[6 + j][i], fmul(fsqr(w2), w[j])));
      }
      fr w01 = fmul(w[0], w[1]), w23 = fmul(w[2], w[3]);
      acc = fadd(acc, fmul(pkc[4][i], w01));
      acc = fadd(acc, fmul(pkc[5][i], w23));
      acc = fadd(acc, fmul(pkc[12][i], fmul(fmul(w01, w23), w[4])));
      acc = fsub(acc, fmul(pkc[10][i], w[4]));
      fr a = zc[i], b = zc[(i + 8) & (m - 1)], bx = fmul(beta, x);
      for (int j = 0; j < NW; j++) {
        fr wg = fadd(w[j], gamma);
        a = fmul(a, fadd(wg, fmul(K->k[j], bx)));
        b = fmul(b, fadd(wg, fmul(beta, pkc[NS + j][i])));
      }
      acc = fadd(acc, fmul(alpha, fsub(a, b)));
      acc = fmul(acc, zh_inv[i & 7]);
      t[i] = fadd(acc, fmul(fmul(alpha2, fsub(zc[i], F_ONE)), d[i]));
      x = fmul(x, wm);
    }
    free(d); free(pre);
  }
  for (int i = 0; i < 18; i++) free(pkc[i]);
  for (int i = 0; i < NW; i++) free(wc[i]);
  free(zc); free(pic);
  capref_ntt_fr((uint64_t *)t, K->log_m, 1, 1);
  {
    size_t deg = m; while (deg > 0 && fis_zero(t[deg - 1])) deg--;
    if (deg != NW * (n + 1) + 3) { /* degree + 1 */
      for (int i = 0; i < NW; i++) free(wpoly[i]);
      free(pi); free(zpoly); free(t); free(tr.buf);
      return -7;
    }
  }
  for (int i = 0; i < NW; i++) {
    size_t len = i < NW - 1 ? n + 2 : (NW * (n + 1) + 3) - (size_t)(NW - 1) * (n + 2);
    commit(K, t + (size_t)i * (n + 2), len, comm_out[6 + i]);
    tr_append_g1(&tr, comm_out[6 + i]);
  }
  /* round 4 */
  const fr zeta = tr_challenge(&tr), zeta_w = fmul(zeta, omega);
  fr we[NW], se[NW - 1], znext;
  for (int i = 0; i < NW; i++) capref_poly_eval_fr((uint64_t *)wpoly[i], n + 2, zeta.l, we[i].l);
  for (int i = 0; i < NW - 1; i++) capref_poly_eval_fr((uint64_t *)(K->coef + (size_t)(NS + i) * n), n, zeta.l, se[i].l);
  capref_poly_eval_fr((uint64_t *)zpoly, n + 3, zeta_w.l, znext.l);
  for (int i = 0; i < NW; i++) { tr_append_fr(&tr, we[i]); eval_out[i] = we[i]; }
  for (int i = 0; i < NW - 1; i++) { tr_append_fr(&tr, se[i]); eval_out[NW + i] = se[i]; }
  tr_append_fr(&tr, znext); eval_out[9] = znext;
  /* linearisation polynomial */
  fr *lin = (fr *)calloc(n + 3, sizeof(fr));
  {
    const fr zh = fsub(fpow_u64(zeta, n), F_ONE);
    const fr l1 = fmul(zh, finv(fmul(ffrom_u64((uint64_t)n), fsub(zeta, F_ONE))));
    const fr *sel = K->coef;
    for (int j = 0; j < 4; j++) poly_add_scaled(lin, sel + (size_t)j * n, n, we[j]);
    fr w01 = fmul(we[0], we[1]), w23 = fmul(we[2], we[3]);
    poly_add_scaled(lin, sel + (size_t)4 * n, n, w01);
    poly_add_scaled(lin, sel + (size_t)5 * n, n, w23);
    for (int j = 0; j < 4; j++) { fr w2 = fsqr(we[j]); poly_add_scaled(lin, sel + (size_t)(6 + j) * n, n, fmul(fsqr(w2), we[j])); }
    poly_add_scaled(lin, sel + (size_t)10 * n, n, fneg(we[4]));
    poly_add_scaled(lin, sel + (size_t)11 * n, n, F_ONE);
    poly_add_scaled(lin, sel + (size_t)12 * n, n, fmul(fmul(w01, w23), we[4]));
    fr bz = fmul(beta, zeta), cz = alpha;
    for (int j = 0; j < NW; j++) cz = fmul(cz, fadd(fadd(we[j], gamma), fmul(K->k[j], bz)));
    cz = fadd(cz, fmul(alpha2, l1));
    poly_add_scaled(lin, zpoly, n + 3, cz);
    fr cs = fmul(fmul(alpha, beta), znext);
    for (int j = 0; j < NW - 1; j++) cs = fmul(cs, fadd(fadd(we[j], gamma), fmul(beta, se[j])));
    poly_add_scaled(lin, K->coef + (size_t)(NS + NW - 1) * n, n, fneg(cs));
    fr zp = fpow_u64(zeta, n + 2), cq = fneg(zh);
    for (int j = 0; j < NW; j++) {
      size_t len = j < NW - 1 ? n + 2 : (NW * (n + 1) + 3) - (size_t)(NW - 1) * (n + 2);
      poly_add_scaled(lin, t + (size_t)j * (n + 2), len, cq);
      cq = fmul(cq, zp);
    }
  }
  /* round 5 */
  const fr v = tr_challenge(&tr);
  {
    fr cf = v;
    for (int j = 0; j < NW; j++) { poly_add_scaled(lin, wpoly[j], n + 2, cf); cf = fmul(cf, v); }
    for (int j = 0; j < NW - 1; j++) { poly_add_scaled(lin, K->coef + (size_t)(NS + j) * n, n, cf); cf = fmul(cf, v); }
  }
  fr *q = (fr *)calloc(n + 3, sizeof(fr));
  divide_linear(lin, n + 3, zeta, q);
  commit(K, q, n + 2, comm_out[11]);
  divide_linear(zpoly, n + 3, zeta_w, q);
  commit(K, q, n + 2, comm_out[12]);
  for (int i = 0; i < NW; i++) free(wpoly[i]);
  free(pi); free(zpoly); free(t); free(lin); free(q); free(tr.buf);
  return 0;
}
