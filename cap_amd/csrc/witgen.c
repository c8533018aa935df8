/* witgen.c - satisfying assignments for the synthetic TurboPlonk circuits of cap_amd/bench_utils.py, at C speed.
 *
 * Workload synthesis for bench.py and the tests (setup, outside every timed region): the counterpart of the reference's
 * witness generation inside `TransferCircuit::build` (src/circuit/transfer.rs:53-193), which runs on the CPU there too and
 * is out of scope.  It is NOT on the proving path and NOT the oracle: libcapgpu.so never links or calls it.
 * `SyntheticCircuit.witness()` (pure Python) is the definition; this file computes the same values bit for bit
 * (tests/test_bench_utils.py) so that a batch of 256 DISTINCT witnesses costs a second instead of minutes.
 *
 * Gate (spec PDF §4.2.1 eq. 1; selector order q_lc[4], q_mul[2], q_hash[4], q_o, q_c, q_ecc):
 *   q_c + sum q_lc_i w_i + q_mul0 w0 w1 + q_mul1 w2 w3 + sum q_hash_i w_i^5 = (q_o - q_ecc w0 w1 w2 w3) w4
 * Rows are walked in order; a row whose output variable has no value yet defines it, any other row is a constraint
 * between variables that already hold values and must simply hold (checked when `verify` is set).
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t v[4]; } fr;

static const uint64_t R_MOD[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static const uint64_t R2[4] = {0x1bb8e645ae216da7ull, 0x53fe3ab1e35c59e3ull, 0x8c49833d53bb8085ull, 0x0216d0b17f4e44a5ull}; /* 2^512 mod r */
static const uint64_t ONE_M[4] = {0xac96341c4ffffffbull, 0x36fc76959f60cd29ull, 0x666ea36f7879462eull, 0x0e0a77c19a07df2full}; /* 2^256 mod r */
#define NINV 0xc2e1f593efffffffull /* -r^-1 mod 2^64 */

static inline int geq_r(const uint64_t* a) {
  for (int i = 3; i >= 0; i--) {
    if (a[i] > R_MOD[i]) return 1;
    if (a[i] < R_MOD[i]) return 0;
  }
  return 1;
}
static inline void sub_r(uint64_t* a) {
  u128 b = 0;
  for (int i = 0; i < 4; i++) {
    u128 t = (u128)a[i] - R_MOD[i] - (uint64_t)b;
    a[i] = (uint64_t)t;
    b = (t >> 64) & 1;
  }
}
static inline int is_zero(const fr* a) { return (a->v[0] | a->v[1] | a->v[2] | a->v[3]) == 0; }
static inline int is_one_m(const fr* a) { return memcmp(a->v, ONE_M, 32) == 0; }

static inline fr f_add(const fr* a, const fr* b) {
  fr r;
  u128 c = 0;
  for (int i = 0; i < 4; i++) {
    c += (u128)a->v[i] + b->v[i];
    r.v[i] = (uint64_t)c;
    c >>= 64;
  }
  if (c || geq_r(r.v)) sub_r(r.v);
  return r;
}
static inline fr f_sub(const fr* a, const fr* b) {
  fr r;
  u128 br = 0;
  for (int i = 0; i < 4; i++) {
    u128 t = (u128)a->v[i] - b->v[i] - (uint64_t)br;
    r.v[i] = (uint64_t)t;
    br = (t >> 64) & 1;
  }
  if (br) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
      c += (u128)r.v[i] + R_MOD[i];
      r.v[i] = (uint64_t)c;
      c >>= 64;
    }
  }
  return r;
}
/* Montgomery product a b 2^-256 mod r (CIOS, 4 x 64) */
static inline fr f_mul(const fr* a, const fr* b) {
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) {
      c += (u128)a->v[j] * b->v[i] + t[j];
      t[j] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[4] = (uint64_t)c;
    t[5] = (uint64_t)(c >> 64);
    uint64_t m = t[0] * NINV;
    c = ((u128)m * R_MOD[0] + t[0]) >> 64;
    for (int j = 1; j < 4; j++) {
      c += (u128)m * R_MOD[j] + t[j];
      t[j - 1] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[3] = (uint64_t)c;
    t[4] = t[5] + (uint64_t)(c >> 64);
  }
  fr r = {{t[0], t[1], t[2], t[3]}};
  if (t[4] || geq_r(r.v)) sub_r(r.v);
  return r;
}
static fr f_inv(const fr* a) { /* a^(r-2): 254 squarings */
  uint64_t e[4] = {R_MOD[0] - 2, R_MOD[1], R_MOD[2], R_MOD[3]};
  fr acc;
  memcpy(acc.v, ONE_M, 32);
  for (int i = 253; i >= 0; i--) {
    acc = f_mul(&acc, &acc);
    if ((e[i >> 6] >> (i & 63)) & 1) acc = f_mul(&acc, a);
  }
  return acc;
}
static inline fr f_pow5(const fr* a) {
  fr s = f_mul(a, a);
  s = f_mul(&s, &s);
  return f_mul(&s, a);
}

typedef struct { uint64_t s; } splitmix;
static inline uint64_t sm_next(splitmix* g) {
  g->s += 0x9E3779B97F4A7C15ull;
  uint64_t z = g->s;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
/* four words, little-endian, reduced mod r; returned in Montgomery form */
static inline fr sm_field_mont(splitmix* g) {
  fr x;
  for (int i = 0; i < 4; i++) x.v[i] = sm_next(g);
  while (geq_r(x.v)) sub_r(x.v);
  fr r2;
  memcpy(r2.v, R2, 32);
  return f_mul(&x, &r2);
}
static inline fr small_mont(uint64_t k) {
  fr x = {{k, 0, 0, 0}}, r2;
  memcpy(r2.v, R2, 32);
  return f_mul(&x, &r2);
}

enum { Q_LC = 0, Q_MUL = 4, Q_HASH = 6, Q_O = 10, Q_C = 11, Q_ECC = 12 };
/* classes of a free variable: how its value is drawn (bench_utils.VAR_*) */
enum { VAR_UNIFORM = 0, VAR_BOOL = 1, VAR_U64 = 2 };

/* One witness.  sel: [13][n][4] Montgomery; wire_vars: [5][n]; free_vars / free_class: [n_free] in drawing order;
 * variable 0 holds 0 and variable 1 holds 1.  wires_out: [5][n][4] Montgomery, pubs_out: [num_inputs][4].
 * returns 0, or -(row + 1) for a degenerate defining row (zero denominator), or +(row + 1) for a violated constraint row. */
int capwit_fill(uint32_t n, uint32_t num_inputs, uint32_t gate_rows, uint32_t num_vars, const uint64_t* sel,
                const int32_t* wire_vars, const int32_t* free_vars, const uint8_t* free_class, uint32_t n_free,
                uint64_t seed, int verify, uint64_t* wires_out, uint64_t* pubs_out) {
  fr* val = (fr*)calloc(num_vars, sizeof(fr));
  uint8_t* has = (uint8_t*)calloc(num_vars, 1);
  if (!val || !has) { free(val); free(has); return -0x7fffffff; }
  splitmix g = {seed};
  has[0] = 1;
  memcpy(val[1].v, ONE_M, 32);
  has[1] = 1;
  for (uint32_t k = 0; k < n_free; k++) {
    int32_t v = free_vars[k];
    uint8_t cls = free_class ? free_class[k] : VAR_UNIFORM;
    if (cls == VAR_BOOL) val[v] = small_mont(sm_next(&g) & 1);
    else if (cls == VAR_U64) val[v] = small_mont(sm_next(&g));
    else val[v] = sm_field_mont(&g);
    has[v] = 1;
  }
  const fr* S = (const fr*)sel;
  int rc = 0;
  for (uint32_t j = num_inputs; j < gate_rows && rc == 0; j++) {
    int32_t out = wire_vars[4 * (size_t)n + j];
    int defines = !has[out];
    if (!defines && !verify) continue;
    fr w[4];
    for (int i = 0; i < 4; i++) w[i] = val[wire_vars[i * (size_t)n + j]];
    fr rest = S[Q_C * (size_t)n + j];
    for (int i = 0; i < 4; i++) {
      const fr* q = &S[(Q_LC + i) * (size_t)n + j];
      if (!is_zero(q) && !is_zero(&w[i])) { fr t = f_mul(q, &w[i]); rest = f_add(&rest, &t); }
    }
    for (int m = 0; m < 2; m++) {
      const fr* q = &S[(Q_MUL + m) * (size_t)n + j];
      if (!is_zero(q)) { fr t = f_mul(&w[2 * m], &w[2 * m + 1]); t = f_mul(&t, q); rest = f_add(&rest, &t); }
    }
    for (int i = 0; i < 4; i++) {
      const fr* q = &S[(Q_HASH + i) * (size_t)n + j];
      if (!is_zero(q)) { fr t = f_pow5(&w[i]); t = f_mul(&t, q); rest = f_add(&rest, &t); }
    }
    fr d = S[Q_O * (size_t)n + j];
    const fr* qe = &S[Q_ECC * (size_t)n + j];
    if (!is_zero(qe)) {
      fr t = f_mul(&w[0], &w[1]);
      t = f_mul(&t, &w[2]);
      t = f_mul(&t, &w[3]);
      t = f_mul(&t, qe);
      d = f_sub(&d, &t);
    }
    if (defines) {
      if (is_zero(&d)) { rc = -(int)(j + 1); break; }
      if (is_one_m(&d)) val[out] = rest;
      else { fr di = f_inv(&d); val[out] = f_mul(&rest, &di); }
      has[out] = 1;
    } else {
      fr lhs = f_mul(&d, &val[out]);
      if (memcmp(lhs.v, rest.v, 32) != 0) rc = (int)(j + 1);
    }
  }
  if (rc == 0) {
    fr* W = (fr*)wires_out;
    for (int i = 0; i < 5; i++)
      for (uint32_t j = 0; j < n; j++) W[i * (size_t)n + j] = val[wire_vars[i * (size_t)n + j]];
    fr* Pb = (fr*)pubs_out;
    for (uint32_t j = 0; j < num_inputs; j++) Pb[j] = val[wire_vars[4 * (size_t)n + j]];
  }
  free(val);
  free(has);
  return rc;
}

typedef struct {
  uint32_t n, num_inputs, gate_rows, num_vars, n_free, count, next, threads;
  const uint64_t* sel;
  const int32_t *wire_vars, *free_vars;
  const uint8_t* free_class;
  const uint64_t* seeds;
  int verify;
  uint64_t *wires_out, *pubs_out;
  int* rcs;
  pthread_mutex_t mu;
} job_t;

static void* worker(void* p) {
  job_t* jb = (job_t*)p;
  for (;;) {
    pthread_mutex_lock(&jb->mu);
    uint32_t i = jb->next++;
    pthread_mutex_unlock(&jb->mu);
    if (i >= jb->count) return 0;
    jb->rcs[i] = capwit_fill(jb->n, jb->num_inputs, jb->gate_rows, jb->num_vars, jb->sel, jb->wire_vars, jb->free_vars,
                             jb->free_class, jb->n_free, jb->seeds[i], jb->verify,
                             jb->wires_out + (size_t)i * 5 * jb->n * 4, jb->pubs_out + (size_t)i * jb->num_inputs * 4);
  }
}

/* `count` witnesses of one circuit on up to `threads` host threads; rcs[i] as capwit_fill.  returns the number of failures */
int capwit_fill_many(uint32_t n, uint32_t num_inputs, uint32_t gate_rows, uint32_t num_vars, const uint64_t* sel,
                     const int32_t* wire_vars, const int32_t* free_vars, const uint8_t* free_class, uint32_t n_free,
                     const uint64_t* seeds, uint32_t count, uint32_t threads, int verify, uint64_t* wires_out,
                     uint64_t* pubs_out, int* rcs) {
  job_t jb = {n, num_inputs, gate_rows, num_vars, n_free, count, 0, threads, sel, wire_vars, free_vars, free_class, seeds,
              verify, wires_out, pubs_out, rcs};
  pthread_mutex_init(&jb.mu, 0);
  if (threads < 1) threads = 1;
  if (threads > count) threads = count;
  if (threads > 256) threads = 256;
  pthread_t th[256];
  uint32_t started = 0;
  for (uint32_t t = 1; t < threads; t++)
    if (pthread_create(&th[started], 0, worker, &jb) == 0) started++;
  worker(&jb);
  for (uint32_t t = 0; t < started; t++) pthread_join(th[t], 0);
  pthread_mutex_destroy(&jb.mu);
  int bad = 0;
  for (uint32_t i = 0; i < count; i++) bad += rcs[i] != 0;
  return bad;
}

/* value classes of a wire table (what decides the digits of a commitment taken from evaluations):
 * out[0] = zeros, out[1] = ones, out[2] = other values below 2^64, out[3] = the rest; input Montgomery [count][4] */
void capwit_value_classes(const uint64_t* vals, uint64_t count, uint64_t* out) {
  fr one = {{1, 0, 0, 0}};
  out[0] = out[1] = out[2] = out[3] = 0;
  for (uint64_t i = 0; i < count; i++) {
    fr c = f_mul((const fr*)(vals + 4 * i), &one); /* out of Montgomery form */
    if (is_zero(&c)) out[0]++;
    else if (c.v[0] == 1 && !(c.v[1] | c.v[2] | c.v[3])) out[1]++;
    else if (!(c.v[1] | c.v[2] | c.v[3])) out[2]++;
    else out[3]++;
  }
}

/* Work of one MSM on these scalars with signed base-2^c digits on a shared bucket set (msm.hip: msm_digit): the number of
 * non-zero digits (= bucket-list entries) and of non-empty buckets; mixed additions = entries - non-empty buckets.
 * vals: Montgomery [count][4]; c <= 24. */
void capwit_msm_work(const uint64_t* vals, uint64_t count, uint32_t c, uint64_t* entries_out, uint64_t* buckets_out) {
  const uint32_t half = 1u << (c - 1), mask = (1u << c) - 1, windows = (254 + c - 1) / c;
  uint8_t* hit = (uint8_t*)calloc((size_t)half + 1, 1);
  uint64_t entries = 0, buckets = 0;
  fr one = {{1, 0, 0, 0}};
  for (uint64_t i = 0; i < count; i++) {
    fr k = f_mul((const fr*)(vals + 4 * i), &one);
    uint32_t carry = 0;
    for (uint32_t w = 0; w < windows; w++) {
      const uint32_t bit = w * c, limb = bit >> 6, off = bit & 63;
      uint64_t v = limb < 4 ? k.v[limb] >> off : 0;
      if (off + c > 64 && limb + 1 < 4) v |= k.v[limb + 1] << (64 - off);
      uint32_t d = ((uint32_t)v & mask) + carry;
      carry = 0;
      if (d > half) {
        d = (1u << c) - d;
        carry = 1;
      }
      if (d) {
        entries++;
        if (hit && !hit[d]) {
          hit[d] = 1;
          buckets++;
        }
      }
    }
  }
  free(hit);
  *entries_out = entries;
  *buckets_out = buckets;
}

/* ---- closed-loop callers (bench harness) ---------------------------------------------------------------------------
 * The reference's callers are rayon workers: compiled threads that call prove() for one note and come straight back for
 * the next (src/utils/params_builder.rs:194-226).  Python threads cannot play them - forty threads released by one batch
 * queue for the interpreter lock before they can call again, and the stragglers miss the next batch (round 6,
 * profiles/phase_trace_r06.md) - so the bench starts `threads` native threads here, each making `calls` calls of `fn`
 * (capgpu_plonk_prove_ex, passed as a pointer: this helper does not link the product library).  Call k of thread t uses
 * entry t * calls + k of the argument arrays.  Returns the number of failed calls; *seconds_out = first call to last
 * return. */
typedef int (*capwit_prove_fn)(uint64_t pk, const uint64_t* wires, const uint64_t* pubs, size_t num_inputs,
                               const uint8_t* msg, size_t msg_len, const uint64_t* blinders, int form, void* proof_out);
struct loop_ctx {
  capwit_prove_fn fn;
  uint64_t pk;
  int calls;
  const uint64_t* const* wires;
  const uint64_t* const* pubs;
  size_t num_inputs;
  const uint8_t* msg;
  size_t msg_len;
  const uint64_t* const* blinders;
  void* const* proofs;
  pthread_barrier_t* bar;
  int t, failed;
};
static void* loop_worker(void* p) {
  struct loop_ctx* c = (struct loop_ctx*)p;
  pthread_barrier_wait(c->bar);
  for (int k = 0; k < c->calls; k++) {
    const size_t i = (size_t)c->t * (size_t)c->calls + (size_t)k;
    if (c->fn(c->pk, c->wires[i], c->pubs[i], c->num_inputs, c->msg, c->msg_len, c->blinders[i], 0, c->proofs[i])) c->failed++;
  }
  return 0;
}
int capwit_closed_loop_callers(capwit_prove_fn fn, uint64_t pk, int threads, int calls, const uint64_t* const* wires,
                               const uint64_t* const* pubs, size_t num_inputs, const uint8_t* msg, size_t msg_len,
                               const uint64_t* const* blinders, void* const* proofs, double* seconds_out) {
  if (threads < 1 || calls < 1 || threads > 4096) return -1;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
  struct loop_ctx* cx = (struct loop_ctx*)malloc(sizeof(struct loop_ctx) * (size_t)threads);
  pthread_barrier_t bar;
  if (!th || !cx || pthread_barrier_init(&bar, 0, (unsigned)threads + 1)) return -1;
  int started = 0;
  for (int t = 0; t < threads; t++) {
    struct loop_ctx c = {fn, pk, calls, wires, pubs, num_inputs, msg, msg_len, blinders, proofs, &bar, t, 0};
    cx[t] = c;
    if (pthread_create(&th[t], 0, loop_worker, &cx[t])) break;
    started++;
  }
  if (started != threads) return -1; /* (a box that cannot start the threads: the leg reports the failure) */
  struct timespec t0, t1;
  pthread_barrier_wait(&bar);
  clock_gettime(CLOCK_MONOTONIC, &t0);
  int failed = 0;
  for (int t = 0; t < threads; t++) {
    pthread_join(th[t], 0);
    failed += cx[t].failed;
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (seconds_out) *seconds_out = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
  pthread_barrier_destroy(&bar);
  free(th);
  free(cx);
  return failed;
}
