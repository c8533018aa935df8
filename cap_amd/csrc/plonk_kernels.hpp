// K7-K9 of SURVEY.md §8a: the O(n) element-wise / scan kernels of the TurboPlonk
// prover (permutation grand product, fused quotient evaluation, polynomial
// evaluation, linear combination, division by a linear factor).  Everything here
// is batched over P independent proofs that share one proving key.
//
// Restates on the device the per-round loops of jf-plonk's prover
// (`PlonkKzgSnark::prove`, call site src/proof/transfer.rs:181-186; algorithm as
// recalled in SURVEY.md Appendix A.2-A.6).
#pragma once
#include <hip/hip_runtime.h>

#include "field.hpp"
#include "field29.hpp"

namespace cap {
namespace pk {

constexpr int NW = 5;    // wire types
constexpr int NS = 13;   // selectors: q_lc x4, q_mul x2, q_hash x4, q_o, q_c, q_ecc
constexpr int kThreads = 256;
constexpr int kScanPerThread = 4;
constexpr int kScanBlock = kThreads * kScanPerThread;  // 1024 elements per workgroup

// per-proof challenges, device layout
struct Chal {
  fe beta, gamma, alpha, alpha2;
};

struct QuotConst {
  fe g;           // coset generator 5 (Montgomery)
  fe k[NW];       // wire-subset separators
  fe zh_inv[8];   // 1 / ((g * w_m^i)^n - 1), i mod 6 (m = 6n: six distinct values)
};

// dst[(q/inner)*dst_outer + (q%inner)*dst_inner + k] = k < len ? src[(q/inner)*src_outer + (q%inner)*src_inner + k] : 0
__global__ __launch_bounds__(kThreads) void k_pad_copy(fe* __restrict__ dst, size_t dst_outer, size_t dst_inner,
                                                       const fe* __restrict__ src, size_t src_outer, size_t src_inner,
                                                       uint32_t inner, size_t len, size_t total) {
  size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= total) return;
  uint32_t q = blockIdx.y;
  size_t d = (size_t)(q / inner) * dst_outer + (size_t)(q % inner) * dst_inner + k;
  fe v = Fr::zero();
  if (k < len) v = src[(size_t)(q / inner) * src_outer + (size_t)(q % inner) * src_inner + k];
  dst[d] = v;
}

// poly_q += (b_0 + b_1 X + ...)(X^n - 1); blinder index = (q/inner)*13 + bl_off + (q%inner)*nb + t.
// SET_TAIL: the coefficients n .. n + 7 hold nothing yet (the interpolation wrote n of the n + 8 slots): they are set -
// blinders, then zeros - instead of added to.
template <int SET_TAIL>
__global__ void k_blind(fe* __restrict__ polys, size_t stride, size_t n, const fe* __restrict__ blinders,
                        uint32_t inner, uint32_t bl_off, uint32_t nb, uint32_t count) {
  uint32_t t = threadIdx.x, q = blockIdx.x;
  if (q >= count) return;
  fe* p = polys + (size_t)q * stride;
  if (t >= nb) {
    if (SET_TAIL && t < 8 && n + t < stride) p[n + t] = Fr::zero();
    return;
  }
  fe b = blinders[(size_t)(q / inner) * 13 + bl_off + (size_t)(q % inner) * nb + t];
  p[t] = Fr::sub(p[t], b);
  p[n + t] = SET_TAIL ? b : Fr::add(p[n + t], b);
}

// commitments from evaluations (lagrange.hip): the MSM scalars of column q are its n values followed by its nb blinders -
// commit(w + (b0 + b1 X [+ b2 X^2])(X^n - 1)) = sum_j w_j [L_j] + sum_e b_e [tau^(n+e) - tau^e].  Column q's values start at
// evals + q * src_stride; its blinders at blinders[(q / inner) * 13 + bl_off + (q % inner) * nb ..] (k_blind's layout).
__global__ __launch_bounds__(kThreads) void k_stage_evals(const fe* __restrict__ evals, size_t src_stride, size_t n,
                                                          const fe* __restrict__ blinders, uint32_t inner,
                                                          uint32_t bl_off, uint32_t nb,
                                                          fe* __restrict__ dst /*[count][n + nb]*/) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t q = blockIdx.y;
  if (j >= n + nb) return;
  dst[(size_t)q * (n + nb) + j] =
      j < n ? evals[(size_t)q * src_stride + j]
            : blinders[(size_t)(q / inner) * 13 + bl_off + (size_t)(q % inner) * nb + (j - n)];
}

// round 2: per-row numerator / denominator of the permutation grand product
//   num_j = prod_i (w_i + beta k_i omega^j + gamma),   den_j = prod_i (w_i + beta sigma_i(omega^j) + gamma)
// On the lazy field.  All inputs are in arkworks' form (x * 2^256); a lazy product of two such values is x y 2^256 / 2^5,
// so the scalars that multiply data - beta and the k_i (qc29) - are taken in the internal form (x * 2^261: their products
// with arkworks-form data stay in arkworks' form), and the four products of each chain, which lose 2^5 apiece, are
// put right by one multiplication with 2^20 (internal form).  21 lazy multiplications of ~210 instructions instead of 18
// saturated ones of ~400.
// sig_of (optional): the sigma evaluations of proof p's own key - a batch may mix proofs of several keys on one domain
__global__ __launch_bounds__(kThreads) void k_perm_numden(const fe* __restrict__ wires /*[P][5][n]*/,
                                                          const fe* __restrict__ sig_eval /*[5][n]*/,
                                                          const fe* const* __restrict__ sig_of,
                                                          const fe* __restrict__ tw_n, const Chal* __restrict__ chal,
                                                          QuotConst qc29, size_t n, fe* __restrict__ num,
                                                          fe* __restrict__ den) {
  using F = Fr29;
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  uint32_t p = blockIdx.y;
  if (sig_of) sig_eval = sig_of[p];
  const fl beta = F::from_ext(chal[p].beta);  // internal form
  const fl gamma = F::load(chal[p].gamma);    // arkworks' form, only ever added
  const fl bx = F::mul(beta, F::load(tw_n[j]));
  fl a, b;
#pragma unroll 1
  for (int i = 0; i < NW; i++) {
    const fl wg = F::add(F::load(wires[((size_t)p * NW + i) * n + j]), gamma);
    const fl t1 = F::normalize(F::add(wg, i == 0 ? bx : F::mul(F::load(qc29.k[i]), bx)));
    const fl t2 = F::normalize(F::add(wg, F::mul(beta, F::load(sig_eval[(size_t)i * n + j]))));
    a = i == 0 ? t1 : F::mul(a, t1);
    b = i == 0 ? t2 : F::mul(b, t2);
  }
  // 2^20 in the internal form: 2^281 mod r = (2^261 mod r) * 2^20 mod r, by 20 doublings of ONE
  fl fix = F::one();
#pragma unroll 1
  for (int k = 0; k < 20; k++) fix = F::weak_reduce(F::add(fix, fix));
  num[(size_t)p * n + j] = F::pack(F::canonical(F::mul(a, fix)));
  den[(size_t)p * n + j] = F::pack(F::canonical(F::mul(b, fix)));
}

// ---- exclusive scans over [batch][len] arrays (stride elements apart) --------------------------------
// OP 0: field multiplication, 1: field addition.  REV: suffix scan (logical index len-1-k).
template <int OP>
__device__ __forceinline__ fe scan_op(const fe& a, const fe& b) {
  return OP == 0 ? Fr::mul(a, b) : Fr::add(a, b);
}
template <int OP>
__device__ __forceinline__ fe scan_identity() {
  return OP == 0 ? Fr::one() : Fr::zero();
}
__device__ __forceinline__ fe shfl_up_fe(const fe& a, int d) {
  fe r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = __shfl_up(a.v[i], d);
  return r;
}

// phase 1: block-local exclusive scan; block totals to tot[batch][nblocks]
template <int OP, int REV>
__global__ __launch_bounds__(kThreads) void k_scan_local(const fe* __restrict__ in, fe* __restrict__ out, size_t len,
                                                         size_t stride, fe* __restrict__ tot, uint32_t nblocks) {
  __shared__ fe wave_tot[kThreads / 64];
  const uint32_t b = blockIdx.y, blk = blockIdx.x;
  const fe* src = in + (size_t)b * stride;
  fe* dst = out + (size_t)b * stride;
  size_t k0 = (size_t)blk * kScanBlock + (size_t)threadIdx.x * kScanPerThread;
  fe v[kScanPerThread];
  fe run = scan_identity<OP>();
#pragma unroll
  for (int t = 0; t < kScanPerThread; t++) {
    size_t k = k0 + t;
    fe x = scan_identity<OP>();
    if (k < len) x = src[REV ? (len - 1 - k) : k];
    v[t] = run;                 // exclusive within the thread
    run = t == 0 ? x : scan_op<OP>(run, x);
  }
  // inclusive scan of thread totals across the wavefront
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  fe inc = run;
  for (int d = 1; d < 64; d <<= 1) {
    fe o = shfl_up_fe(inc, d);
    if ((int)lane >= d) inc = scan_op<OP>(o, inc);
  }
  if (lane == 63) wave_tot[wave] = inc;
  fe excl = shfl_up_fe(inc, 1);
  if (lane == 0) excl = scan_identity<OP>();
  __syncthreads();
  fe wpre = scan_identity<OP>();
  for (uint32_t w = 0; w < wave; w++) wpre = scan_op<OP>(wpre, wave_tot[w]);
  fe base = scan_op<OP>(wpre, excl);
#pragma unroll
  for (int t = 0; t < kScanPerThread; t++) {
    size_t k = k0 + t;
    if (k < len) dst[REV ? (len - 1 - k) : k] = t == 0 ? base : scan_op<OP>(base, v[t]);
  }
  if (threadIdx.x == kThreads - 1) tot[(size_t)b * nblocks + blk] = scan_op<OP>(wpre, inc);
}

// phase 2: exclusive scan of the block totals (one wavefront per batch entry, 64 blocks per step)
template <int OP>
__global__ __launch_bounds__(64) void k_scan_totals(fe* __restrict__ tot, uint32_t nblocks) {
  fe* t = tot + (size_t)blockIdx.x * nblocks;
  const uint32_t lane = threadIdx.x;
  fe carry = scan_identity<OP>();
  for (uint32_t base = 0; base < nblocks; base += 64) {
    uint32_t i = base + lane;
    fe x = i < nblocks ? t[i] : scan_identity<OP>();
    fe inc = x;
    for (int d = 1; d < 64; d <<= 1) {
      fe o = shfl_up_fe(inc, d);
      if ((int)lane >= d) inc = scan_op<OP>(o, inc);
    }
    fe excl = shfl_up_fe(inc, 1);
    if (lane == 0) excl = scan_identity<OP>();
    if (i < nblocks) t[i] = scan_op<OP>(carry, excl);
    fe last;
#pragma unroll
    for (int w = 0; w < 8; w++) last.v[w] = __shfl(inc.v[w], 63);
    carry = scan_op<OP>(carry, last);
  }
}

// phase 3: fold the block prefix in
template <int OP, int REV>
__global__ __launch_bounds__(kThreads) void k_scan_apply(fe* __restrict__ out, size_t len, size_t stride,
                                                         const fe* __restrict__ tot, uint32_t nblocks) {
  const uint32_t b = blockIdx.y, blk = blockIdx.x;
  if (blk == 0) return;
  fe pre = tot[(size_t)b * nblocks + blk];
  fe* dst = out + (size_t)b * stride;
  for (int t = 0; t < kScanPerThread; t++) {
    size_t k = (size_t)blk * kScanBlock + (size_t)t * kThreads + threadIdx.x;
    if (k < len) {
      size_t idx = REV ? (len - 1 - k) : k;
      dst[idx] = scan_op<OP>(pre, dst[idx]);
    }
  }
}

// total[p] = sfx[p][0] * den[p][0], the product of all denominators of proof p: the host inverts it (one batched
// inversion for the whole batch, microseconds) where a lone device thread spent 0.17 ms on a 254-squaring chain
__global__ void k_perm_total(const fe* __restrict__ sfx, const fe* __restrict__ den, size_t n, fe* __restrict__ total,
                             uint32_t count) {
  uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= count) return;
  total[p] = Fr::mul(sfx[(size_t)p * n], den[(size_t)p * n]);
}

// inv_total[p] = 1 / (sfx[p][0] * den[p][0])   (device form, CAPGPU_PERM_INV_ON_DEVICE=1)
__global__ void k_perm_inv_total(const fe* __restrict__ sfx, const fe* __restrict__ den, size_t n,
                                 fe* __restrict__ inv_total, uint32_t count) {
  uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= count) return;
  // one thread per proof and nothing else on the device: the lazy field's windowed inversion in its row-wise
  // (latency) schedule - 0.31 -> 0.1 ms on the path of a single proof
  using F = Fl<FrP29, 0>;
  inv_total[p] = F::to_ext(F::inv(F::mul(F::from_ext(sfx[(size_t)p * n]), F::from_ext(den[(size_t)p * n]))));
}

// z[j] = prefix_num[j] * (suffix_den_excl[j] * den[j]) / prod(den); zero padding up to stride
__global__ __launch_bounds__(kThreads) void k_perm_finish(const fe* __restrict__ pre, const fe* __restrict__ sfx,
                                                          const fe* __restrict__ den,
                                                          const fe* __restrict__ inv_total, size_t n,
                                                          fe* __restrict__ z, size_t zstride) {
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t p = blockIdx.y;
  if (j >= zstride) return;
  fe v = Fr::zero();
  if (j < n) {
    size_t o = (size_t)p * n + j;
    v = Fr::mul(Fr::mul(pre[o], sfx[o]), Fr::mul(den[o], inv_total[p]));
  }
  z[(size_t)p * zstride + j] = v;
}

// ---- round 3: fused quotient evaluation on the coset of size m = 6n ----------------------------------
// (jf-plonk uses 8n; the quotient has degree < 5n + 8, so 6n = 3 * 2^(log n + 1) points determine it - ntt.hpp)
// All inputs are in the internal Montgomery form of the lazy 29-bit field (x * 2^261): the forward coset NTTs emit
// it (ntt3_forward) and the inverse coset NTT that follows consumes it (ntt3_inverse).
// pkc: [18][m] coset evaluations of 13 selectors then 5 sigmas (shared by all proofs)
// cos: [P][7][m] coset evaluations of 5 wires, z, pi;   tw_m: omega_m^i;   inv_nx1: 1 / (n (x_i - 1))
// Sums of products share one Montgomery reduction (at most 6 products per 64-bit column accumulator).
struct ColAcc {
  uint64_t c[18];
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int k = 0; k < 18; k++) c[k] = 0;
  }
  __device__ __forceinline__ void mad(const fl& a, const fl& b) {
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
      for (int j = 0; j < 9; j++) c[i + j] += (uint64_t)a.v[i] * b.v[j];
  }
  __device__ __forceinline__ fl reduce() { return Fr29::reduce_cols(c); }
};

// pkc_of (optional): the key columns of proof p's own key (mixed-key batches), otherwise pkc for all
__global__ __launch_bounds__(kThreads) void k_quotient(const fe* __restrict__ pkc, const fe* const* __restrict__ pkc_of,
                                                       const fe* __restrict__ cos, const fe* __restrict__ xs,
                                                       const fe* __restrict__ inv_nx1,
                                                       const Chal* __restrict__ chal, QuotConst qc, size_t m,
                                                       fe* __restrict__ t_out) {
  using F = Fr29;
  // grid (proofs, tiles): the proofs of one tile run together, so the 18 shared selector / sigma tiles are read from
  // HBM once per tile and served from L2 to the other proofs of the batch
  size_t i = (size_t)blockIdx.y * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const uint32_t p = blockIdx.x;
  if (pkc_of) pkc = pkc_of[p];
  const uint32_t mm = (uint32_t)(m / 3);                                  // block length M (uniform: scalar unit)
  const uint32_t blk = (i >= mm ? 1u : 0u) + (i >= 2 * (size_t)mm ? 1u : 0u);  // which of the three cosets
  const fe* c = cos + (size_t)p * 7 * m;
  auto sel = [&](int s) { return F::load(pkc[(size_t)s * m + i]); };
  fl w0 = F::load(c[i]), w1 = F::load(c[m + i]), w2 = F::load(c[2 * m + i]), w3 = F::load(c[3 * m + i]),
     w4 = F::load(c[4 * m + i]);
  // gate constraint (spec eq. (1); selector order q_lc, q_mul, q_hash, q_o, q_c, q_ecc)
  fl w01 = F::mul(w0, w1), w23 = F::mul(w2, w3);
  ColAcc acc;
  acc.clear();
  acc.mad(sel(0), w0);
  acc.mad(sel(1), w1);
  acc.mad(sel(2), w2);
  acc.mad(sel(3), w3);
  acc.mad(sel(4), w01);
  acc.mad(sel(5), w23);
  fl gate = acc.reduce();
  acc.clear();
  {
    fl t = F::sqr(w0);
    acc.mad(sel(6), F::mul(F::sqr(t), w0));
    t = F::sqr(w1);
    acc.mad(sel(7), F::mul(F::sqr(t), w1));
    t = F::sqr(w2);
    acc.mad(sel(8), F::mul(F::sqr(t), w2));
    t = F::sqr(w3);
    acc.mad(sel(9), F::mul(F::sqr(t), w3));
  }
  acc.mad(sel(12), F::mul(F::mul(w01, w23), w4));
  acc.mad(F::neg(sel(10)), w4);
  fl gate2 = acc.reduce();
  // gate + q_c + pi: four values < 2p each, limbs < 2^31 after the lazy adds
  fl total = F::normalize(F::add(F::add(gate, gate2), F::add(sel(11), F::load(c[6 * m + i]))));
  // permutation part
  const fl beta = F::load(chal[p].beta), gamma = F::load(chal[p].gamma);
  const fl zx = F::load(c[5 * m + i]);
  {
    fl x = F::load(xs[i]);  // the point itself (internal form): s_a * omega_M^k at index a * M + k
    fl bx = F::mul(beta, x);
    // index a M + k (ntt.hpp): x * omega_n = x * omega_M^2 is two steps further inside the same block of M = m / 3
    const uint32_t k1 = (uint32_t)i - blk * mm;
    const size_t inext = (size_t)blk * mm + ((k1 + 2) & (mm - 1));
    fl a = zx, b = F::load(c[5 * m + inext]);
    fl wg = F::add(w0, gamma);
    a = F::mul(a, F::normalize(F::add(wg, bx)));
    b = F::mul(b, F::normalize(F::add(wg, F::mul(beta, sel(NS + 0)))));
    wg = F::add(w1, gamma);
    a = F::mul(a, F::normalize(F::add(wg, F::mul(F::load(qc.k[1]), bx))));
    b = F::mul(b, F::normalize(F::add(wg, F::mul(beta, sel(NS + 1)))));
    wg = F::add(w2, gamma);
    a = F::mul(a, F::normalize(F::add(wg, F::mul(F::load(qc.k[2]), bx))));
    b = F::mul(b, F::normalize(F::add(wg, F::mul(beta, sel(NS + 2)))));
    wg = F::add(w3, gamma);
    a = F::mul(a, F::normalize(F::add(wg, F::mul(F::load(qc.k[3]), bx))));
    b = F::mul(b, F::normalize(F::add(wg, F::mul(beta, sel(NS + 3)))));
    wg = F::add(w4, gamma);
    a = F::mul(a, F::normalize(F::add(wg, F::mul(F::load(qc.k[4]), bx))));
    b = F::mul(b, F::normalize(F::add(wg, F::mul(beta, sel(NS + 4)))));
    total = F::normalize(F::add(total, F::mul(F::load(chal[p].alpha), F::sub(a, b))));
  }
  // (gate + alpha * perm) / Z_H  +  alpha^2 (z - 1) / (n (x - 1)) : two products, one reduction
  fl l1a = F::mul(F::load(chal[p].alpha2), F::sub(zx, F::one()));
  // x^n depends on the block and on the parity of k only: (s_a omega_M^k)^n = s_a^n (-1)^k = (5 omega_N^(3k + a))^n
  const uint32_t zi = 3 * (((uint32_t)i - blk * mm) & 1) + blk;  // < 6
  fl r = F::mul_add_mul(total, F::load(qc.zh_inv[zi]), l1a, F::load(inv_nx1[i]));
  t_out[(size_t)p * m + i] = F::store(r);
}

// out[i] = 1 / (n * (x_i - 1))   (one-time table of the proving key; xs = the points of the quotient domain)
__global__ __launch_bounds__(kThreads) void k_inv_nx1(fe* __restrict__ out, const fe* __restrict__ xs, fe n_mont,
                                                      size_t m) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  out[i] = Fr::inv(Fr::mul(n_mont, Fr::sub(xs[i], Fr::one())));
}

// flags[p] |= 1 if any coefficient at index >= lo is non-zero; |= 2 if coefficient lo-1 is zero
__global__ __launch_bounds__(kThreads) void k_check_degree(const fe* __restrict__ t, size_t m, size_t lo,
                                                           uint32_t* __restrict__ flags) {
  size_t k = lo - 1 + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= m) return;
  uint32_t p = blockIdx.y;
  bool zero = Fr::is_zero(t[(size_t)p * m + k]);
  if (k == lo - 1) {
    if (zero) atomicOr(&flags[p], 2u);
  } else if (!zero) {
    atomicOr(&flags[p], 1u);
  }
}

// ---- round 4/5 helpers ---------------------------------------------------------------------------------
// tables[q][k] = base_q^k, k < len;  pw[q][b] = base_q^(2^b), b < 24.  Two steps: a small table per base with
// base^i (i < 256) followed by base^(256 j), each entry by the binary method, then one multiplication per output.
constexpr uint32_t kPowLow = 256;
__global__ __launch_bounds__(kThreads) void k_powers_small(fe* __restrict__ small, uint32_t small_len,
                                                           const fe* __restrict__ pw) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= small_len) return;
  uint32_t q = blockIdx.y;
  const fe* b = pw + (size_t)q * 24;
  const uint32_t e = i < kPowLow ? i : (i - kPowLow) * kPowLow;
  fe r = Fr::one();
  bool started = false;
  for (int k = 0; (e >> k) != 0; k++) {
    if ((e >> k) & 1) {
      r = started ? Fr::mul(r, b[k]) : b[k];
      started = true;
    }
  }
  small[(size_t)q * small_len + i] = r;
}
__global__ __launch_bounds__(kThreads) void k_powers(fe* __restrict__ tables, size_t stride, size_t len,
                                                     const fe* __restrict__ small, uint32_t small_len) {
  size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= len) return;
  uint32_t q = blockIdx.y;
  const fe* t = small + (size_t)q * small_len;
  // blockDim.x == kPowLow: the high factor is the same for the whole workgroup
  tables[(size_t)q * stride + e] = Fr::mul(t[e & (kPowLow - 1)], t[kPowLow + (e >> 8)]);
}

struct EvalDesc {
  const fe* poly;
  const fe* pows;
  uint32_t len;
  uint32_t pad;
};

__device__ __forceinline__ fe wave_sum(fe v) {
  for (int d = 32; d >= 1; d >>= 1) {
    fe o;
#pragma unroll
    for (int i = 0; i < 8; i++) o.v[i] = __shfl_down(v.v[i], d);
    v = Fr::add(v, o);
  }
  return v;
}

// partial[e][chunk] = sum over the chunk of poly[k] * pows[k]
__global__ __launch_bounds__(kThreads) void k_eval_partial(const EvalDesc* __restrict__ desc, fe* __restrict__ partial,
                                                           uint32_t chunks, uint32_t per_chunk) {
  __shared__ fe sh[kThreads / 64];
  const uint32_t e = blockIdx.y, chunk = blockIdx.x;
  const EvalDesc d = desc[e];
  fe acc = Fr::zero();
  uint32_t lo = chunk * per_chunk, hi = lo + per_chunk;
  if (hi > d.len) hi = d.len;
  for (uint32_t k = lo + threadIdx.x; k < hi; k += kThreads) acc = Fr::add(acc, Fr::mul(d.poly[k], d.pows[k]));
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    fe r = sh[0];
    for (int w = 1; w < kThreads / 64; w++) r = Fr::add(r, sh[w]);
    partial[(size_t)e * chunks + chunk] = r;
  }
}
__global__ __launch_bounds__(64) void k_eval_final(const fe* __restrict__ partial, uint32_t chunks,
                                                   fe* __restrict__ out) {
  const uint32_t e = blockIdx.x;
  fe acc = Fr::zero();
  for (uint32_t k = threadIdx.x; k < chunks; k += 64) acc = Fr::add(acc, partial[(size_t)e * chunks + k]);
  acc = wave_sum(acc);
  if (threadIdx.x == 0) out[e] = acc;
}

struct LinTerm {
  const fe* poly;
  fe scalar;
  uint32_t len;
  uint32_t pad[3];
};

// out[p][k] = sum_t terms[p][t].scalar * terms[p][t].poly[k]   (k < terms[p][t].len), k < out_len
// Scalars arrive in the internal form of the lazy field (x * 2^261), polynomials in arkworks' form, so every product
// is back in arkworks' form; six products share one Montgomery reduction.
__global__ __launch_bounds__(kThreads) void k_lincomb(const LinTerm* __restrict__ terms, uint32_t nterms,
                                                      fe* __restrict__ out, size_t out_stride, size_t out_len) {
  using F = Fr29;
  size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= out_len) return;
  uint32_t p = blockIdx.y;
  const LinTerm* T = terms + (size_t)p * nterms;
  fl total = F::zero();
  for (uint32_t t0 = 0; t0 < nterms; t0 += 6) {
    ColAcc acc;
    acc.clear();
#pragma unroll
    for (uint32_t u = 0; u < 6; u++) {
      const uint32_t t = t0 + u;
      if (t < nterms && k < T[t].len) acc.mad(F::load(T[t].scalar), F::load(T[t].poly[k]));
    }
    total = F::normalize(F::add(total, acc.reduce()));
  }
  out[(size_t)p * out_stride + k] = F::pack(F::canonical(total));
}

// h[q][k] = f[q][k] * pows[tab(q)][k];  tab(q) = (q/2)*4 + (q%2)      (q = proof*2 + {zeta, zeta*omega})
__global__ __launch_bounds__(kThreads) void k_div_prepare(const fe* __restrict__ f, const fe* __restrict__ pows,
                                                          size_t stride, size_t len, fe* __restrict__ h) {
  size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= len) return;
  uint32_t q = blockIdx.y;
  size_t tab = (size_t)(q / 2) * 4 + (q % 2);
  h[(size_t)q * stride + k] = Fr::mul(f[(size_t)q * stride + k], pows[tab * stride + k]);
}
// quot[q][i] = sfx[q][i] * ainv^(i+1), i < len-1 (sfx = exclusive suffix sum of h); quot[q][len-1..stride) = 0
__global__ __launch_bounds__(kThreads) void k_div_finish(const fe* __restrict__ sfx, const fe* __restrict__ pows,
                                                         size_t stride, size_t len, fe* __restrict__ quot) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= stride) return;
  uint32_t q = blockIdx.y;
  size_t tab = (size_t)(q / 2) * 4 + 2 + (q % 2);
  fe v = Fr::zero();
  if (i + 1 < len) v = Fr::mul(sfx[(size_t)q * stride + i], pows[tab * stride + i + 1]);
  quot[(size_t)q * stride + i] = v;
}

}  // namespace pk
}  // namespace cap
