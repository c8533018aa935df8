// K2: multi-pass NTT over BN254 Fr with LDS-staged sub-transforms (see ntt.hpp).
//
// Decomposition (decimation in frequency, mixed radix 2^a x 2^b [x 2^c]):
//   n = N1*N2*N3, input index i = (i1*N2 + i2)*N3 + i3, output index k = k1 + N1*k2 + N1*N2*k3.
//   "column" passes transform one digit at a stride, multiply by the inter-pass
//   twiddle omega_M^(col*k) and store in place (to scratch); the final "row"
//   pass transforms contiguous rows and stores digit-reversed, C adjacent k1 per
//   workgroup so the scattered stores are C*32 B contiguous.
// Each workgroup stages a tile of len x C elements in LDS (<= 64 KiB) and runs
// log2(len) radix-2 stages there; HBM sees one read + one write of the array per
// pass.  All arithmetic is 254-bit modular integer work (no MFMA).
// (The column-wise multiplication schedule was measured here too: column pass 19.6 -> 19.4 ms, row pass 18.1 -> 19.0 ms
// per step.  The butterflies keep several independent multiplications in flight; the row-wise schedule stays.)
#include "ntt.hpp"
#include "field29.hpp"
#include "launch.hpp"

#include <stdlib.h>

#include <vector>

namespace cap {

namespace {

constexpr int kThreads = 256;
// Elements per thread whose global loads are issued together.  Measured in round 4 (round-4 same-box A/B `ntt_io`, profiles/LOG.md, same box,
// col + row pass per step): 1 (the old element-by-element loop) 37.2-37.6 ms, 2: 36.7 ms, 4: 38.1-38.4 ms (135 VGPRs: three
// waves per SIMD instead of four; held to 128 VGPRs with 52 B of scratch: 37.9 ms).  The passes are not waiting for
// memory - they are issue-bound butterflies with a barrier per radix-4 round - so 2 stays, for what it is worth.
#ifndef CAP_NTT_IO_BATCH
#define CAP_NTT_IO_BATCH 2
#endif
constexpr int kIoBatch = CAP_NTT_IO_BATCH;
#ifdef CAP_NTT_WAVES4  // experiment: hold the passes to the 128 VGPRs of four waves per SIMD (the LDS tile allows no more)
#define CAP_NTT_ATTR __attribute__((amdgpu_waves_per_eu(4, 4)))
#else
#define CAP_NTT_ATTR
#endif  // elements per thread whose global loads are issued together (a 1024-element tile: all 4)
constexpr uint32_t kMaxTileLogDefault = 10;  // 1024 elements * 36 B = 36 KiB of LDS (4 workgroups per CU)
uint32_t max_tile_log() {
  static uint32_t v = [] {
    const char* e = getenv("CAPGPU_NTT_TILE_LOG");
    int x = e ? atoi(e) : (int)kMaxTileLogDefault;
    return (uint32_t)(x >= 8 && x <= 11 ? x : (int)kMaxTileLogDefault);
  }();
  return v;
}

// A launch of a few transforms (a single proof's: 1 .. 15 arrays of 2^15 .. 2^16 elements) makes fewer 1024-element tiles
// than the chip has CUs, and a pass then lasts as long as ONE workgroup takes for its tile: such launches get tiles of 512
// or 256 elements - twice / four times the workgroups, each half / a quarter as long.  (For launches that fill the chip the
// smaller tiles are slower - narrower column tiles, more twiddle traffic: round-4 log - and are never chosen.)
uint32_t tile_log_for(uint32_t log_n, uint32_t count) {
  static const bool adapt = [] {
    const char* e = getenv("CAPGPU_NTT_TILE_ADAPT");
    return !e || atoi(e) != 0;
  }();
  uint32_t tl = max_tile_log();
  if (!adapt) return tl;
  while (tl > 8 && (((uint64_t)count << log_n) >> tl) < 256) tl--;
  return tl;
}

struct PassParams {
  const fe* in;
  fe* out;
  // array q of the batch starts at (q / group) * outer + (q % group) * inner elements; input elements at or beyond
  // in_len read as zero (a polynomial shorter than the transform: no padded copy has to exist in memory)
  size_t in_outer, in_inner, in_len;
  uint32_t in_group;
  uint32_t in_es;            // element stride of the first-pass input (decimated transforms); 1 = contiguous
  // second grouping level of the input (decimated transforms of grouped polynomials): array q = (q2, a) with
  // a = q % in_group the decimation phase and q2 = q / in_group placed at (q2 / in_group2) * in_outer + (q2 % in_group2) * in_inner2
  size_t in_inner2;
  uint32_t in_group2;
  size_t out_outer, out_inner;
  uint32_t out_group;
  size_t out_inner2;         // second grouping level of the output, like in_group2 / in_inner2
  uint32_t out_group2;
  size_t pre_inner;          // the pre-scale table of array q starts at (q % in_group) * pre_inner
  const fl* tw_small;        // omega_len^i, i < len/2, unpacked limbs
  const fe* tw_full;         // omega_N^e, e < N (col pass twiddles); may be null for row pass
  const fe* pre_scale;       // indexed by global input index, or null
  const fe* post_scale;      // indexed by global output index, or null
  fe post_scalar;            // used when use_post_scalar
  uint32_t use_post_scalar;
  uint32_t lazy_out;         // row pass: leave results weakly reduced (< 2r) instead of canonical
  uint32_t log_n;
  uint32_t log_len;          // sub-transform size
  uint32_t log_c;            // tile width
  uint32_t log_m;            // col pass: segment size
  uint32_t log_n1, log_n2;   // row pass: digit sizes of the leading digits
  // persistent launch: tiles_x * tiles_y tiles are handed out by *tile_counter to a fixed number of workgroups (null:
  // one workgroup per tile, blockIdx = tile)
  uint32_t* tile_counter;
  uint32_t tiles_x, tiles_y;
};

// the tile a workgroup works on next: (blockIdx.x, blockIdx.y) once, or the next one from the counter; false: none left
__device__ __forceinline__ bool next_tile(const PassParams& p, uint32_t round, uint32_t& bx, uint32_t& by) {
  if (!p.tile_counter) {
    bx = blockIdx.x;
    by = blockIdx.y;
    return round == 0;
  }
  __shared__ uint32_t tile_s;
  __syncthreads();  // (everybody is done with the LDS tile and with tile_s of the round before)
  if (threadIdx.x == 0) tile_s = atomicAdd(p.tile_counter, 1u);
  __syncthreads();
  const uint32_t t = tile_s;
  if (t >= p.tiles_x * p.tiles_y) return false;
  bx = t % p.tiles_x;
  by = t / p.tiles_x;
  return true;
}

__device__ __forceinline__ uint32_t bitrev32(uint32_t x, uint32_t bits) { return __brev(x) >> (32 - bits); }

// DIT stages on sh[len][C] (rows were written bit-reversed): natural order out.
// Lazy 29-bit field: per butterfly  t = v * w (one Montgomery product, < 1.2p),  (u + t, u + 2p - t); values
// grow by at most 2p per stage (<= 2p * 11 stages + input), far below the 169p capacity of 9 x 29-bit limbs.
// Twiddles are stored in the internal Montgomery form (w * 2^261), so data keeps whatever form it came in.
//
// Stage plan: one radix-2 stage when the stage count is odd, then radix-4 rounds (stages s, s + 1 on four rows held in
// registers: the same multiplications as two radix-2 stages at half the LDS traffic, index arithmetic and barriers -
// 147 instead of 216 instructions per element and stage).  The first round of an even-length transform (s = 0) has
// trivial twiddles - 1, 1 and omega_4 - and, for the prover's zero-padded inputs (a polynomial of n + 2 coefficients on a
// 2n-point coset: rows len/2.. are zero), two zero operands: ONE multiplication for its four elements instead of four.
__device__ __forceinline__ uint32_t fl_any(const fl& a) {
  uint32_t o = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) o |= a.v[k];
  return o;
}
// Round 6 (round-5 VERDICT item 4): the butterflies' twiddle products are constant-multiplicand products (Fl::mul_shoup,
// independent column accumulators: 143 multiply-adds and no serial digit chain against the Montgomery product's 171).
// tw_small holds PAIRS - entry i at [2i] the plain canonical twiddle, at [2i + 1] its quotient floor(w 2^261 / p) - and a
// product is below 5p instead of 1.2p: the subtractions add 8p instead of 2p, values grow by at most 13p per radix-4
// round (5 rounds + the inputs stay far below the 169p the limbs hold).  Everything outside the radix-4 rounds - coset
// pre-scale, inter-pass twiddle, post-scale: factors from 32-byte tables - stays a Montgomery product.  Measured on one
// box (profiles/ntt_shoup_ab_r06.txt): 768 x 2^16 coset transforms 4.17 -> 4.03 ms, column + row pass per headline step
// 35.7 -> 34.3 ms, 1373 -> 1387 proofs/s; the microbenchmark behind the decision: profiles/ubench_shoup29_r06.txt.
// -DCAP_NTT_NO_SHOUP builds the Montgomery rounds of rounds 1-5.
#ifndef CAP_NTT_NO_SHOUP
constexpr int kTwStride = 2;
#define NTT_TW_MUL(x, idx) Fr29::mul_shoup((x), tw_small[2 * (idx)], tw_small[2 * (idx) + 1])
#define NTT_SUB(u, t) Fr29::sub8p((u), (t))
#define NTT_SUB_LAZY(u, t) Fr29::sub8p_lazy((u), (t))
#define NTT_SUB_FROM_LAZY(u, t) Fr29::sub8p((u), (t))
#else
constexpr int kTwStride = 1;
#define NTT_TW_MUL(x, idx) Fr29::mul((x), tw_small[(idx)])
#define NTT_SUB(u, t) Fr29::sub2p((u), (t))
#define NTT_SUB_LAZY(u, t) Fr29::sub2p_lazy((u), (t))
#define NTT_SUB_FROM_LAZY(u, t) Fr29::sub2p((u), (t))
#endif
__device__ __forceinline__ void lds_ntt(fl* sh, const fl* __restrict__ tw_small, uint32_t log_len, uint32_t log_c) {
  const uint32_t half_tile = 1u << (log_len + log_c - 1);
  const uint32_t cmask = (1u << log_c) - 1;
  const uint32_t s0 = log_len & 1;
  if (s0) {
    // stage 0 alone: every twiddle is omega^0 = 1
    for (uint32_t b = threadIdx.x; b < half_tile; b += kThreads) {
      const uint32_t c = b & cmask, bb = b >> log_c;
      const uint32_t i0 = ((bb << 1) << log_c) + c, i1 = i0 + (1u << log_c);
      const fl u = sh[i0], v = sh[i1];
      if (fl_any(v) == 0) {  // zero padding: the butterfly is a copy
        sh[i1] = u;
        continue;
      }
      const fl t = Fr29::weak_reduce(v);
      sh[i0] = Fr29::normalize(Fr29::add(u, t));
      sh[i1] = Fr29::sub2p(u, t);
    }
    __syncthreads();
  }
  const uint32_t quarter_tile = half_tile >> 1;
  for (uint32_t s = s0; s + 1 < log_len; s += 2) {
    const uint32_t h = 1u << s;
    const uint32_t step = h << log_c;
    for (uint32_t g = threadIdx.x; g < quarter_tile; g += kThreads) {
      const uint32_t c = g & cmask;
      const uint32_t gg = g >> log_c;
      const uint32_t pos = gg & (h - 1);
      const uint32_t j = ((gg >> s) << (s + 2)) + pos;
      const uint32_t i0 = (j << log_c) + c, i1 = i0 + step, i2 = i1 + step, i3 = i2 + step;
      fl a = sh[i0], b = sh[i1], cc = sh[i2], d = sh[i3];
      // carries are propagated only where a value becomes a multiplicand (limbs < 2^30 needed) or goes back to LDS:
      // a1, c1 = x + t have limbs < 2^30 as they are; b1 = a - t + 2p is only added to afterwards
      fl a1, b1, c1, d1, t;
      if (s == 0) {
        // w1 = 1: t = b, d themselves (weakly reduced for the subtraction); zero operands make the butterfly a copy
        if (fl_any(b)) {
          t = Fr29::weak_reduce(b);
          a1 = Fr29::add(a, t);
          b1 = Fr29::sub2p_lazy(a, t);
        } else {
          a1 = a;
          b1 = a;
        }
        if (fl_any(d)) {
          t = Fr29::weak_reduce(d);
          c1 = Fr29::normalize(Fr29::add(cc, t));
          d1 = Fr29::sub2p(cc, t);
        } else {
          c1 = cc;
          d1 = cc;
        }
        t = Fr29::weak_reduce(c1);  // second-level twiddle of the (a1, c1) pair: omega^0 = 1
      } else {
        const uint32_t i_w1 = pos << (log_len - 1 - s);
        t = NTT_TW_MUL(b, i_w1);
        a1 = Fr29::add(a, t);
        b1 = NTT_SUB_LAZY(a, t);
        t = NTT_TW_MUL(d, i_w1);
        c1 = Fr29::add(cc, t);
        d1 = NTT_SUB(cc, t);
        t = NTT_TW_MUL(c1, pos << (log_len - 2 - s));
      }
      sh[i0] = Fr29::normalize(Fr29::add(a1, t));
      sh[i2] = NTT_SUB(a1, t);
      t = NTT_TW_MUL(d1, (pos + h) << (log_len - 2 - s));
      sh[i1] = Fr29::normalize(Fr29::add(b1, t));
      sh[i3] = NTT_SUB_FROM_LAZY(b1, t);
    }
    __syncthreads();
  }
}

// column pass: len rows at stride S = M/len, C adjacent columns per tile
__device__ __forceinline__ void col_tile(const PassParams& p, fl* sh, const uint32_t bx, const uint32_t by) {
  {
  const uint32_t log_s = p.log_m - p.log_len;           // columns per segment (log)
  const uint32_t tiles_per_seg_log = log_s - p.log_c;
  const uint32_t t = bx;
  const size_t seg = t >> tiles_per_seg_log;
  const uint32_t col0 = (t & ((1u << tiles_per_seg_log) - 1)) << p.log_c;
  const uint32_t q2 = by / p.in_group;
  const fe* in = p.in + (size_t)(q2 / p.in_group2) * p.in_outer + (size_t)(q2 % p.in_group2) * p.in_inner2 +
                 (size_t)(by % p.in_group) * p.in_inner;
  const uint32_t o2 = by / p.out_group;
  fe* out = p.out + (size_t)(o2 / p.out_group2) * p.out_outer + (size_t)(o2 % p.out_group2) * p.out_inner2 +
            (size_t)(by % p.out_group) * p.out_inner;
  const size_t base = (seg << p.log_m) + col0;
  const uint32_t tile = 1u << (p.log_len + p.log_c);
  const uint32_t cmask = (1u << p.log_c) - 1;
  // kIoBatch elements per thread and round, their global loads (and the loads of their coset factors) issued together
  // before any of them is used (the loop used to be load - wait - multiply - store, one element at a time)
  for (uint32_t e0 = threadIdx.x; e0 < tile; e0 += kIoBatch * kThreads) {
    fe raw[kIoBatch], pre[kIoBatch];
    uint32_t any[kIoBatch];
    size_t sgs[kIoBatch];
#pragma unroll
    for (int u = 0; u < kIoBatch; u++) {
      const uint32_t e = e0 + u * kThreads;
      const uint32_t c = e & cmask, j = e >> p.log_c;
      const size_t g = base + ((size_t)j << log_s) + c;
      // position inside the source array (decimated input: every in_es-th element, starting at the group offset)
      const size_t sg = p.in_es == 1 ? g : g * p.in_es + (size_t)(by % p.in_group) * p.in_inner;
      sgs[u] = sg;
      any[u] = 0;
      if (e < tile && sg < p.in_len) {
        raw[u] = p.in_es == 1 ? in[g] : in[g * p.in_es];
        any[u] = 1;
      } else {
#pragma unroll
        for (int k = 0; k < 8; k++) raw[u].v[k] = 0;
      }
    }
    if (p.pre_scale) {
#pragma unroll
      for (int u = 0; u < kIoBatch; u++)
        if (any[u]) pre[u] = p.pre_scale[sgs[u] + (size_t)(by % p.in_group) * p.pre_inner];
    }
#pragma unroll
    for (int u = 0; u < kIoBatch; u++) {
      const uint32_t e = e0 + u * kThreads;
      if (e >= tile) break;
      const uint32_t c = e & cmask, j = e >> p.log_c;
      uint32_t nz = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) nz |= raw[u].v[k];
      fl v = Fr29::load(raw[u]);
      if (p.pre_scale && any[u] && nz) v = Fr29::mul(v, Fr29::load(pre[u]));  // zero padding needs no coset scaling
      sh[(bitrev32(j, p.log_len) << p.log_c) + c] = v;
    }
  }
  __syncthreads();
  lds_ntt(sh, p.tw_small, p.log_len, p.log_c);
  const uint32_t tw_shift = p.log_n - p.log_m;  // omega_M^x = omega_N^(x << tw_shift)
  for (uint32_t e0 = threadIdx.x; e0 < tile; e0 += kIoBatch * kThreads) {
    fe tw[kIoBatch];
    size_t exs[kIoBatch];
#pragma unroll
    for (int u = 0; u < kIoBatch; u++) {  // the inter-pass twiddles of the batch: gathers from a 2 MB table, in flight together
      const uint32_t e = e0 + u * kThreads;
      const uint32_t c = e & cmask, k = e >> p.log_c;
      exs[u] = e < tile ? ((size_t)(col0 + c) * k) << tw_shift : 0;
      if (exs[u]) tw[u] = p.tw_full[exs[u]];
    }
#pragma unroll
    for (int u = 0; u < kIoBatch; u++) {
      const uint32_t e = e0 + u * kThreads;
      if (e >= tile) break;
      const uint32_t c = e & cmask, k = e >> p.log_c;
      fl v = sh[e];
      v = exs[u] ? Fr29::mul(v, Fr29::load(tw[u])) : Fr29::weak_reduce(v);
      out[base + ((size_t)k << log_s) + c] = Fr29::pack(v);   // < 2p: fits the 32-byte image
    }
  }
  }
}
// PERSIST: a fixed number of workgroups take tile after tile from a counter (next_tile); otherwise one tile per workgroup.
// Two instantiations, because the loop around the tile costs the one-shot form 55 VGPRs (93 -> 148: three waves per SIMD).
template <bool PERSIST>
__global__ __launch_bounds__(kThreads) CAP_NTT_ATTR void ntt_col_pass(PassParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  fl* sh = reinterpret_cast<fl*>(smem);
  if constexpr (!PERSIST) {
    col_tile(p, sh, blockIdx.x, blockIdx.y);
  } else {
    uint32_t bx, by;
    for (uint32_t round = 0; next_tile(p, round, bx, by); round++) col_tile(p, sh, bx, by);
  }
}

// row pass: contiguous rows of len elements; C rows with adjacent k1 per tile; digit-reversed store
__device__ __forceinline__ void row_tile(const PassParams& p, fl* sh, const uint32_t bx, const uint32_t by) {
  {
  const uint32_t t = bx;
  const uint32_t k2 = t & ((1u << p.log_n2) - 1);
  const uint32_t r0 = (t >> p.log_n2) << p.log_c;
  const uint32_t q2 = by / p.in_group;
  const fe* in = p.in + (size_t)(q2 / p.in_group2) * p.in_outer + (size_t)(q2 % p.in_group2) * p.in_inner2 +
                 (size_t)(by % p.in_group) * p.in_inner;
  const uint32_t o2 = by / p.out_group;
  fe* out = p.out + (size_t)(o2 / p.out_group2) * p.out_outer + (size_t)(o2 % p.out_group2) * p.out_inner2 +
            (size_t)(by % p.out_group) * p.out_inner;
  const uint32_t tile = 1u << (p.log_len + p.log_c);
  const uint32_t lmask = (1u << p.log_len) - 1;
  const uint32_t cmask = (1u << p.log_c) - 1;
  for (uint32_t e0 = threadIdx.x; e0 < tile; e0 += kIoBatch * kThreads) {  // batched loads: see ntt_col_pass
    fe raw[kIoBatch], pre[kIoBatch];
    uint32_t any[kIoBatch];
    size_t sgs[kIoBatch];
#pragma unroll
    for (int u = 0; u < kIoBatch; u++) {
      const uint32_t e = e0 + u * kThreads;
      const uint32_t j = e & lmask, c = e >> p.log_len;
      const size_t g = ((((size_t)(r0 + c) << p.log_n2) + k2) << p.log_len) + j;
      const size_t sg = p.in_es == 1 ? g : g * p.in_es + (size_t)(by % p.in_group) * p.in_inner;
      sgs[u] = sg;
      any[u] = 0;
      if (e < tile && sg < p.in_len) {
        raw[u] = p.in_es == 1 ? in[g] : in[g * p.in_es];
        any[u] = 1;
      } else {
#pragma unroll
        for (int k = 0; k < 8; k++) raw[u].v[k] = 0;
      }
    }
    if (p.pre_scale) {
#pragma unroll
      for (int u = 0; u < kIoBatch; u++)
        if (any[u]) pre[u] = p.pre_scale[sgs[u] + (size_t)(by % p.in_group) * p.pre_inner];
    }
#pragma unroll
    for (int u = 0; u < kIoBatch; u++) {
      const uint32_t e = e0 + u * kThreads;
      if (e >= tile) break;
      const uint32_t j = e & lmask, c = e >> p.log_len;
      uint32_t nz = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) nz |= raw[u].v[k];
      fl v = Fr29::load(raw[u]);
      if (p.pre_scale && any[u] && nz) v = Fr29::mul(v, Fr29::load(pre[u]));  // zero padding needs no coset scaling
      sh[(bitrev32(j, p.log_len) << p.log_c) + c] = v;
    }
  }
  __syncthreads();
  lds_ntt(sh, p.tw_small, p.log_len, p.log_c);
  for (uint32_t e = threadIdx.x; e < tile; e += kThreads) {
    uint32_t c = e & cmask, k = e >> p.log_c;
    fl v = sh[e];
    size_t g = (size_t)(r0 + c) + ((size_t)k2 << p.log_n1) + ((size_t)k << (p.log_n1 + p.log_n2));
    if (p.post_scale) v = Fr29::mul(v, Fr29::load(p.post_scale[g]));
    else if (p.use_post_scalar) v = Fr29::mul(v, Fr29::load(p.post_scalar));
    // results leave the transform canonical (< r), as arkworks stores them - except the internal-form coset
    // evaluations, whose only reader (k_quotient) takes any representative below 2^256
    out[g] = p.lazy_out ? Fr29::store(v) : Fr29::pack(Fr29::canonical(v));
  }
  }
}
template <bool PERSIST>
__global__ __launch_bounds__(kThreads) CAP_NTT_ATTR void ntt_row_pass(PassParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  fl* sh = reinterpret_cast<fl*>(smem);
  if constexpr (!PERSIST) {
    row_tile(p, sh, blockIdx.x, blockIdx.y);
  } else {
    uint32_t bx, by;
    for (uint32_t round = 0; next_tile(p, round, bx, by); round++) row_tile(p, sh, bx, by);
  }
}

// tw_small: the internal-form table `in` (x * 2^261, canonical) as the limbs the butterflies multiply by - kTwStride = 1:
// the same values unpacked; kTwStride = 2 (CAP_NTT_SHOUP): pairs (plain canonical x, floor(x 2^261 / p))
__global__ void table_unpack(fl* __restrict__ out, const fe* __restrict__ in, size_t n) {
  size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  if constexpr (kTwStride == 1) {
    out[e] = Fr29::load(in[e]);
  } else {
    const fl t = Fr29::load(in[e]);
    out[2 * e] = Fr29::load(Fr29::from_mont(t));
    out[2 * e + 1] = Fr29::shoup_quotient(t);
  }
}

// internal-form table: out[e] = pack(canonical(in_ext[e] * 2^5))  (x * 2^256 -> x * 2^261)
__global__ void table_to_internal(fe* __restrict__ out, const fe* __restrict__ in, size_t n) {
  size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  out[e] = Fr29::pack(Fr29::canonical(Fr29::from_ext(in[e])));
}

// out[e] = base^e * scale for e < n; pw[b] = base^(2^b)
__global__ void powers_table(fe* out, size_t n, const fe* __restrict__ pw, fe scale, int has_scale) {
  size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  fe r = has_scale ? scale : Fr::one();
  bool started = has_scale;
  for (int b = 0; (e >> b) != 0; b++) {
    if ((e >> b) & 1) {
      r = started ? Fr::mul(r, pw[b]) : pw[b];
      started = true;
    }
  }
  out[e] = r;
}

// ---- radix-3 stage of the N = 3 M transforms -------------------------------------------------------------------
// y: [count][3][M] sub-transform outputs (lazy, internal form).  out: array q at (q / group) * outer + (q % group) * inner.
//   X[k + M b] = Y0 + w3^b T1 + w3^(2b) T2,  T1 = omega_N^k Y1[k], T2 = omega_N^(2k) Y2[k];  with w3^2 = -1 - w3:
//   X[k] = Y0 + T1 + T2,  X[k + M] = Y0 - T2 + D,  X[k + 2M] = Y0 - T1 - D,  D = w3 (T1 - T2)     (3 multiplications)
// post != null (inverse transform): every output is multiplied by post[index] (an arkworks-form integer table, which
// takes internal-form data to arkworks' form) and leaves canonical; otherwise outputs leave weakly reduced.
__global__ __launch_bounds__(kThreads) void ntt3_combine(const fe* __restrict__ y, fe* __restrict__ out, size_t out_outer,
                                                         size_t out_inner, uint32_t out_group, size_t m_len,
                                                         const fe* __restrict__ tw, fe w3, const fe* __restrict__ post) {
  using F = Fr29;
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= m_len) return;
  const uint32_t q = blockIdx.y;
  const fe* yq = y + (size_t)q * 3 * m_len;
  fe* o = out + (size_t)(q / out_group) * out_outer + (size_t)(q % out_group) * out_inner;
  const fl y0 = F::load(yq[k]);
  const fl t1 = F::mul(F::load(yq[m_len + k]), F::load(tw[k]));
  const fl t2 = F::mul(F::load(yq[2 * m_len + k]), F::load(tw[2 * k]));
  const fl d = F::mul(F::sub(t1, t2), F::load(w3));
  fl x0 = F::normalize(F::add(F::add(y0, t1), t2));
  fl x1 = F::normalize(F::add(F::sub(y0, t2), d));
  fl x2 = F::sub(F::sub(y0, t1), d);
  if (post) {
    o[k] = F::pack(F::canonical(F::mul(x0, F::load(post[k]))));
    o[m_len + k] = F::pack(F::canonical(F::mul(x1, F::load(post[m_len + k]))));
    o[2 * m_len + k] = F::pack(F::canonical(F::mul(x2, F::load(post[2 * m_len + k]))));
  } else {
    o[k] = F::store(x0);
    o[m_len + k] = F::store(x1);
    o[2 * m_len + k] = F::store(x2);
  }
}

fe host_root_of_unity(uint32_t log_n) {
  // omega_28 = 5^((r-1)/2^28), canonical little-endian limbs
  fe w;
  const uint32_t root28[8] = {0x725b19f0u, 0x9bd61b6eu, 0x41112ed4u, 0x402d111eu,
                              0x8ef62abcu, 0x00e0a7ebu, 0xa58a7e85u, 0x2a3c09f0u};
  for (int i = 0; i < 8; i++) w.v[i] = root28[i];
  w = Fr::to_mont(w);
  for (uint32_t i = log_n; i < 28; i++) w = Fr::sqr(w);
  return w;
}

fe host_from_u64(uint64_t v) {
  fe t = Fr::zero();
  t.v[0] = (uint32_t)v;
  t.v[1] = (uint32_t)(v >> 32);
  return Fr::to_mont(t);
}

int build_powers(fe* d_out, size_t n, fe base, const fe* scale, hipStream_t stream) {
  std::vector<fe> pw(40);
  fe x = base;
  for (int b = 0; b < 40; b++) {
    pw[b] = x;
    x = Fr::sqr(x);
  }
  fe* d_pw = nullptr;
  hipError_t e = hipMalloc(&d_pw, sizeof(fe) * pw.size());
  if (e != hipSuccess) return (int)e;
  e = hipMemcpyAsync(d_pw, pw.data(), sizeof(fe) * pw.size(), hipMemcpyHostToDevice, stream);
  if (e != hipSuccess) return (int)e;
  fe sc = scale ? *scale : Fr::one();
  size_t blocks = (n + 255) / 256;
  launch("powers_table", powers_table, dim3((unsigned)blocks), dim3(256), 0, stream, d_out, n, d_pw, sc, scale ? 1 : 0);
  e = hipStreamSynchronize(stream);
  hipFree(d_pw);
  return (int)e;
}

}  // namespace

fe ntt_root_of_unity(uint32_t log_n) { return host_root_of_unity(log_n); }

int ntt_build_small_tables(NttSmallTables* t, hipStream_t stream) {
  if (!t->pass_counters) {
    hipError_t ec = hipMalloc(&t->pass_counters, sizeof(uint32_t) * kNttPassCounters);
    if (ec != hipSuccess) return (int)ec;
  }
  for (int s = 1; s <= kMaxLogTile; s++) {
    size_t n = (size_t)1 << (s - 1);
    hipError_t e = hipMalloc(&t->fwd[s], sizeof(fe) * n);
    if (e != hipSuccess) return (int)e;
    e = hipMalloc(&t->inv[s], sizeof(fe) * n);
    if (e != hipSuccess) return (int)e;
    fe w = host_root_of_unity(s);
    int rc = build_powers(t->fwd[s], n, w, nullptr, stream);
    if (rc) return rc;
    rc = build_powers(t->inv[s], n, Fr::inv(w), nullptr, stream);
    if (rc) return rc;
    launch("table_to_internal", table_to_internal, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, t->fwd[s],
           (const fe*)t->fwd[s], n);
    launch("table_to_internal", table_to_internal, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, t->inv[s],
           (const fe*)t->inv[s], n);
    for (int d = 0; d < 2; d++) {
      uint32_t** dst = d ? &t->inv_u[s] : &t->fwd_u[s];
      e = hipMalloc(dst, sizeof(fl) * n * kTwStride);
      if (e != hipSuccess) return (int)e;
      launch("table_unpack", table_unpack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
             reinterpret_cast<fl*>(*dst), (const fe*)(d ? t->inv[s] : t->fwd[s]), n);
    }
    hipError_t es = hipStreamSynchronize(stream);
    if (es != hipSuccess) return (int)es;
  }
  return 0;
}

void ntt_free_small_tables(NttSmallTables* t) {
  if (t->pass_counters) hipFree(t->pass_counters);
  t->pass_counters = nullptr;
  for (int s = 0; s <= kMaxLogTile; s++) {
    if (t->fwd[s]) hipFree(t->fwd[s]);
    if (t->inv[s]) hipFree(t->inv[s]);
    if (t->fwd_u[s]) hipFree(t->fwd_u[s]);
    if (t->inv_u[s]) hipFree(t->inv_u[s]);
    t->fwd[s] = t->inv[s] = nullptr;
    t->fwd_u[s] = t->inv_u[s] = nullptr;
  }
}

int ntt_build_domain(NttDomain* d, uint32_t log_n, hipStream_t stream) {
  d->log_n = log_n;
  size_t n = (size_t)1 << log_n;
  hipError_t e;
  if ((e = hipMalloc(&d->tw_fwd, sizeof(fe) * n)) != hipSuccess) return (int)e;
  if ((e = hipMalloc(&d->tw_inv, sizeof(fe) * n)) != hipSuccess) return (int)e;
  if ((e = hipMalloc(&d->coset_fwd, sizeof(fe) * n)) != hipSuccess) return (int)e;
  if ((e = hipMalloc(&d->coset_inv, sizeof(fe) * n)) != hipSuccess) return (int)e;
  fe w = host_root_of_unity(log_n);
  fe g = host_from_u64(5);
  d->n_inv = Fr::inv(host_from_u64((uint64_t)n));
  int rc;
  if ((rc = build_powers(d->tw_fwd, n, w, nullptr, stream))) return rc;
  if ((rc = build_powers(d->tw_inv, n, Fr::inv(w), nullptr, stream))) return rc;
  if ((rc = build_powers(d->coset_fwd, n, g, nullptr, stream))) return rc;
  if ((rc = build_powers(d->coset_inv, n, Fr::inv(g), &d->n_inv, stream))) return rc;
  fe* src[4] = {d->tw_fwd, d->tw_inv, d->coset_fwd, d->coset_inv};
  fe** dst[4] = {&d->tw29_fwd, &d->tw29_inv, &d->coset29_fwd, &d->coset29_inv};
  for (int k = 0; k < 4; k++) {
    if ((e = hipMalloc(dst[k], sizeof(fe) * n)) != hipSuccess) return (int)e;
    launch("table_to_internal", table_to_internal, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, *dst[k],
           (const fe*)src[k], n);
  }
  d->n_inv29 = Fr29::pack(Fr29::canonical(Fr29::from_ext(d->n_inv)));
  if ((e = hipStreamSynchronize(stream)) != hipSuccess) return (int)e;
  // only omega^j (tw_fwd) is still read in arkworks' form (by the prover's elementwise kernels)
  hipFree(d->tw_inv);
  hipFree(d->coset_fwd);
  hipFree(d->coset_inv);
  d->tw_inv = d->coset_fwd = d->coset_inv = nullptr;
  return 0;
}

void ntt_free_domain(NttDomain* d) {
  if (d->tw_fwd) hipFree(d->tw_fwd);
  if (d->tw_inv) hipFree(d->tw_inv);
  if (d->coset_fwd) hipFree(d->coset_fwd);
  if (d->coset_inv) hipFree(d->coset_inv);
  for (fe* t : {d->tw29_fwd, d->tw29_inv, d->coset29_fwd, d->coset29_inv})
    if (t) hipFree(t);
  d->tw_fwd = d->tw_inv = d->coset_fwd = d->coset_inv = nullptr;
  d->tw29_fwd = d->tw29_inv = d->coset29_fwd = d->coset29_inv = nullptr;
}

void ntt_table_to_internal(fe* out, const fe* in, size_t n, hipStream_t stream) {
  if (n == 0) return;
  launch("table_to_internal", table_to_internal, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, out, in, n);
}

namespace {
// Workgroups per CU of a persistent pass launch (CAPGPU_NTT_PERSISTENT; 0 = one workgroup per tile)
unsigned ntt_persistent() {
  static const unsigned v = [] {
    const char* e = getenv("CAPGPU_NTT_PERSISTENT");
    const long x = e ? atol(e) : 0;
    return (unsigned)(x >= 1 && x <= 16 ? x : 0);
  }();
  return v;
}
template <class K>
void launch_pass(const char* name, K kernel, K kernel_persistent, PassParams& p, const NttSmallTables& small, size_t tiles,
                 uint32_t count, size_t lds, hipStream_t stream) {
  p.tile_counter = nullptr;
  p.tiles_x = (uint32_t)tiles;
  p.tiles_y = count;
  const unsigned per_cu = ntt_persistent();
  const uint64_t total = (uint64_t)tiles * count;
  if (per_cu && small.pass_counters && total > 4ull * 256 * per_cu && total < (1ull << 32)) {
    uint32_t* ctr = small.pass_counters + (small.next_counter++ % kNttPassCounters);
    if (hipMemsetAsync(ctr, 0, sizeof(uint32_t), stream) == hipSuccess) {
      p.tile_counter = ctr;
      launch(name, kernel_persistent, dim3(256u * per_cu), dim3(kThreads), lds, stream, p);
      return;
    }
    (void)hipGetLastError();
  }
  launch(name, kernel, dim3((unsigned)tiles, count), dim3(kThreads), lds, stream, p);
}
}  // namespace

int ntt_run(const NttDomain& dom, const NttSmallTables& small, fe* data, fe* scratch, size_t stride_elems,
            uint32_t count, int dir, int coset, hipStream_t stream, const NttIo* io) {
  const uint32_t log_n = dom.log_n;
  if (count == 0) return 0;
  if (log_n == 0) {
    // n = 1: forward is the identity (5^0 = 1); inverse multiplies by 1^-1 = 1.
    return 0;
  }
  const uint32_t tile_log = tile_log_for(log_n, count);
  // digit split
  uint32_t lg[3] = {0, 0, 0};
  int passes;
  if (log_n <= 10) {
    passes = 1;
    lg[2] = log_n;
  } else if (log_n <= 20) {
    passes = 2;
    lg[0] = (log_n + 1) / 2;
    lg[2] = log_n / 2;
  } else {
    passes = 3;
    lg[0] = (log_n + 2) / 3;
    lg[1] = (log_n - lg[0] + 1) / 2;
    lg[2] = log_n - lg[0] - lg[1];
  }
  if (lg[0] > 10 || lg[1] > 10 || lg[2] > 10) return (int)hipErrorInvalidValue;
  uint32_t* const* tws = dir ? small.inv_u : small.fwd_u;
  const fe* tw_full = dir ? dom.tw29_inv : dom.tw29_fwd;

  PassParams p{};
  // first-pass input and last-pass output addressing; the passes in between go through the scratch buffer
  const size_t scratch_stride = io ? ((size_t)1 << log_n) : stride_elems;
  struct Addr {
    size_t outer, inner;
    uint32_t group;
  };
  const Addr a_src = io ? Addr{io->src_outer, io->src_inner, io->src_group ? io->src_group : 1u} : Addr{stride_elems, 0, 1};
  const Addr a_dst = io ? Addr{io->dst_outer, io->dst_inner, io->dst_group ? io->dst_group : 1u} : Addr{stride_elems, 0, 1};
  const Addr a_tmp = Addr{scratch_stride, 0, 1};
  const size_t src_len = io ? io->src_len : ~(size_t)0;
  auto set_in = [&](const Addr& a, size_t len, uint32_t es) {
    p.in_outer = a.outer;
    p.in_inner = a.inner;
    p.in_group = a.group;
    p.in_len = len;
    p.in_es = es;
    p.in_group2 = 1;
    p.in_inner2 = 0;
    if (&a == &a_src && io && io->src_group2 > 1) {
      p.in_group2 = io->src_group2;
      p.in_inner2 = io->src_inner2;
    }
  };
  const uint32_t src_es = io && io->src_elem_stride ? io->src_elem_stride : 1;
  auto set_out = [&](const Addr& a) {
    p.out_outer = a.outer;
    p.out_inner = a.inner;
    p.out_group = a.group;
    p.out_group2 = 1;
    p.out_inner2 = 0;
    if (&a == &a_dst && io && io->dst_group2 > 1) {
      p.out_group2 = io->dst_group2;
      p.out_inner2 = io->dst_inner2;
    }
  };
  p.pre_inner = io ? io->pre_inner : 0;
  p.log_n = log_n;
  p.tw_full = tw_full;
  p.use_post_scalar = 0;

  const fe* pre = (!dir && coset) ? dom.coset29_fwd : nullptr;
  if (io && io->pre_scale) pre = io->pre_scale;
  bool first = true;
  const fe* cur_in = io ? io->src : data;
  // column passes
  uint32_t log_m = log_n;
  for (int d = 0; d < passes - 1; d++) {
    uint32_t log_len = lg[d];
    uint32_t log_s = log_m - log_len;
    uint32_t log_c = tile_log > log_len ? tile_log - log_len : 0;
    if (log_c > log_s) log_c = log_s;
    if (log_c > 4) log_c = 4;
    p.in = cur_in;
    p.out = scratch;
    set_in(first ? a_src : a_tmp, first ? src_len : ~(size_t)0, first ? src_es : 1);
    set_out(a_tmp);
    p.tw_small = reinterpret_cast<const fl*>(tws[log_len]);
    p.pre_scale = first ? pre : nullptr;
    p.post_scale = nullptr;
    p.log_len = log_len;
    p.log_c = log_c;
    p.log_m = log_m;
    size_t tiles = (size_t)1 << (log_n - log_len - log_c);
    size_t lds = sizeof(fl) << (log_len + log_c);
    launch_pass("ntt_col_pass", ntt_col_pass<false>, ntt_col_pass<true>, p, small, tiles, count, lds, stream);
    cur_in = scratch;
    first = false;
    log_m -= log_len;
  }
  // row pass
  {
    uint32_t log_len = lg[2];
    uint32_t log_n1 = passes >= 2 ? lg[0] : 0;
    uint32_t log_n2 = passes == 3 ? lg[1] : 0;
    uint32_t log_c = tile_log > log_len ? tile_log - log_len : 0;
    if (log_c > log_n1) log_c = log_n1;
    if (log_c > 4) log_c = 4;
    p.in = cur_in;
    p.out = data;
    set_in(first ? a_src : a_tmp, first ? src_len : ~(size_t)0, first ? src_es : 1);
    set_out(a_dst);
    p.tw_small = reinterpret_cast<const fl*>(tws[log_len]);
    p.pre_scale = first ? pre : nullptr;
    p.post_scale = (dir && coset) ? dom.coset29_inv : nullptr;
    p.use_post_scalar = (dir && !coset) ? 1 : 0;
    p.post_scalar = dom.n_inv29;
    p.lazy_out = (io && io->lazy_out) ? 1 : 0;
    p.log_len = log_len;
    p.log_c = log_c;
    p.log_n1 = log_n1;
    p.log_n2 = log_n2;
    size_t tiles = (size_t)1 << (log_n - log_len - log_c);
    size_t lds = sizeof(fl) << (log_len + log_c);
    launch_pass("ntt_row_pass", ntt_row_pass<false>, ntt_row_pass<true>, p, small, tiles, count, lds, stream);
  }
  return 0;  // launch failures are latched by launch() and reported by take_launch_error()
}

// ---- N = 3 * 2^k ------------------------------------------------------------------------------------------------
namespace {
// (r - 1) / 3, little-endian 32-bit words
void exponent_third(uint32_t e[8]) {
  uint32_t t[8];
  for (int i = 0; i < 8; i++) t[i] = FrP::MOD[i];
  t[0] -= 1;  // r is odd and r - 1 is divisible by 3 (r - 1 = 2^28 * 3^2 * ...)
  uint64_t rem = 0;
  for (int i = 7; i >= 0; i--) {
    uint64_t cur = (rem << 32) | t[i];
    e[i] = (uint32_t)(cur / 3);
    rem = cur % 3;
  }
}
}  // namespace

int ntt3_build_domain(Ntt3Domain* d, uint32_t log_m, hipStream_t stream) {
  d->log_m = log_m;
  const size_t M = (size_t)1 << log_m, N = 3 * M;
  // omega_N = omega_3 * omega_M^c, 3c = 1 mod M
  uint32_t e3[8];
  exponent_third(e3);
  const fe g = host_from_u64(5);
  const fe w3gen = Fr::pow(g, e3);  // a primitive cube root of unity
  size_t c = (M % 3 == 2) ? (M + 1) / 3 : (2 * M + 1) / 3;
  if (M == 1) c = 1;
  uint32_t ec[8] = {(uint32_t)c, (uint32_t)((uint64_t)c >> 32), 0, 0, 0, 0, 0, 0};
  const fe wM = host_root_of_unity(log_m);
  const fe wN = Fr::mul(w3gen, Fr::pow(wM, ec));
  d->omega = wN;
  uint32_t eM[8] = {(uint32_t)M, (uint32_t)((uint64_t)M >> 32), 0, 0, 0, 0, 0, 0};
  const fe w3 = Fr::pow(wN, eM);
  auto to_internal = [](const fe& a) { return Fr29::pack(Fr29::canonical(Fr29::from_ext(a))); };
  d->w3inv_29 = to_internal(Fr::inv(w3));
  hipError_t e;
  if ((e = hipMalloc(&d->xs_ext, sizeof(fe) * N)) != hipSuccess) return (int)e;
  if ((e = hipMalloc(&d->xs29, sizeof(fe) * N)) != hipSuccess) return (int)e;
  if ((e = hipMalloc(&d->pre3, sizeof(fe) * N)) != hipSuccess) return (int)e;
  if ((e = hipMalloc(&d->tw29_inv, sizeof(fe) * 2 * M)) != hipSuccess) return (int)e;
  if ((e = hipMalloc(&d->coset_inv_ext, sizeof(fe) * N)) != hipSuccess) return (int)e;
  int rc;
  const fe k32 = host_from_u64(32);
  fe s_a = g;  // 5 * omega_N^a
  for (int a = 0; a < 3; a++) {
    if ((rc = build_powers(d->xs_ext + (size_t)a * M, M, wM, &s_a, stream))) return rc;   // s_a * omega_M^k
    if ((rc = build_powers(d->pre3 + (size_t)a * M, M, s_a, &k32, stream))) return rc;     // 32 * s_a^i
    s_a = Fr::mul(s_a, wN);
  }
  if ((rc = build_powers(d->tw29_inv, 2 * M, Fr::inv(wN), nullptr, stream))) return rc;
  const fe third = Fr::inv(host_from_u64(3));
  if ((rc = build_powers(d->coset_inv_ext, N, Fr::inv(g), &third, stream))) return rc;
  const unsigned blocks = (unsigned)((N + 255) / 256);
  launch("table_to_internal", table_to_internal, dim3(blocks), dim3(256), 0, stream, d->xs29, (const fe*)d->xs_ext, N);
  launch("table_to_internal", table_to_internal, dim3(blocks), dim3(256), 0, stream, d->pre3, (const fe*)d->pre3, N);
  launch("table_to_internal", table_to_internal, dim3((unsigned)((2 * M + 255) / 256)), dim3(256), 0, stream,
         d->tw29_inv, (const fe*)d->tw29_inv, 2 * M);
  return (int)hipStreamSynchronize(stream);
}

void ntt3_free_domain(Ntt3Domain* d) {
  for (fe** t : {&d->xs_ext, &d->xs29, &d->pre3, &d->tw29_inv, &d->coset_inv_ext}) {
    if (*t) hipFree(*t);
    *t = nullptr;
  }
}

int ntt3_forward(const Ntt3Domain& d3, const NttDomain& dom_m, const NttSmallTables& small, fe* data, NttIo io,
                 uint32_t count, fe* scratch, hipStream_t stream) {
  if (dom_m.log_n != d3.log_m) return (int)hipErrorInvalidValue;
  const size_t M = (size_t)1 << d3.log_m;
  if (io.src_len > M) return (int)hipErrorInvalidValue;
  // array 3 q + a: the same polynomial q, pre-scaled by the powers of s_a, transformed, written to block a
  NttIo sub{};
  sub.src = io.src;
  sub.src_outer = io.src_outer;
  sub.src_inner = 0;
  sub.src_group = 3;
  sub.src_group2 = io.src_group ? io.src_group : 1;  // the caller's own grouping moves one level up
  sub.src_inner2 = io.src_inner;
  sub.src_len = io.src_len;
  sub.pre_scale = d3.pre3;
  sub.pre_inner = M;
  sub.lazy_out = 1;
  sub.dst_outer = io.dst_outer;
  sub.dst_inner = M;
  sub.dst_group = 3;
  sub.dst_group2 = io.dst_group ? io.dst_group : 1;
  sub.dst_inner2 = io.dst_inner;
  return ntt_run(dom_m, small, data, scratch, M, count * 3, 0, 0, stream, &sub);
}

int ntt3_inverse(const Ntt3Domain& d3, const NttDomain& dom_m, const NttSmallTables& small, fe* data, uint32_t count,
                 fe* scratch, hipStream_t stream) {
  if (dom_m.log_n != d3.log_m) return (int)hipErrorInvalidValue;
  const size_t M = (size_t)1 << d3.log_m, N = 3 * M;
  fe* y = scratch;
  fe* sub_scratch = scratch + (size_t)count * 3 * M;
  NttIo sub{};
  sub.src = data;
  sub.src_outer = N;
  sub.src_inner = M;  // block a of array q
  sub.src_group = 3;
  sub.src_len = M;
  sub.lazy_out = 1;
  sub.dst_outer = M;
  sub.dst_inner = 0;
  sub.dst_group = 1;
  // inverse sub-transforms of the three blocks (their 1 / M is applied inside), then the radix-3 stage with the
  // coset / 1/3 / form-changing table
  int rc = ntt_run(dom_m, small, y, sub_scratch, M, count * 3, 1, 0, stream, &sub);
  if (rc) return rc;
  launch("ntt3_combine", ntt3_combine, dim3((unsigned)((M + kThreads - 1) / kThreads), count), dim3(kThreads), 0, stream,
         (const fe*)y, data, N, (size_t)0, 1u, M, (const fe*)d3.tw29_inv, d3.w3inv_29, (const fe*)d3.coset_inv_ext);
  return 0;  // launch failures are latched by launch() and reported by take_launch_error()
}

}  // namespace cap
