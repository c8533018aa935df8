// K3-K6: Pippenger MSM over BN254 G1 with fixed-base window precomputation.
// Design notes in msm.hpp.  Kernel sequence for one launch (a batch of MSMs; a long MSM is itself cut into a batch of
// sub-MSMs over consecutive point ranges, see "parts" below):
//   msm_digits_local  K3  tile of 1024 scalars -> signed base-2^c digits, counting-sorted by bucket (or bin) in LDS
//   msm_scan*         K4  exclusive scan of the [key][tile] count table
//   msm_scatter_runs  K4  one-level sort (c = 13): run copies;  msm_sort_level2: per-bin LDS sort (c = 15)
//   msm_accumulate    K5  one thread per work item: mixed adds of gathered 64 B points
//   msm_combine       K5  bucket = sum of its work items
//   msm_reduce_*      K6  running sums over segments (large batches) or bit planes (log depth) -> one point per sub-MSM
//   msm_sum_parts     K6  sum of the sub-MSM results of one MSM
// the MSM kernels run long dependent chains of field multiplications per thread: column-wise schedule (field29.hpp)
#define CAP_FL_SCHED 1
#include "msm.hpp"
#include "curve29.hpp"
#include "quad29.hpp"
#include "launch.hpp"

#include <math.h>
#include <stdio.h>

#include <atomic>
#include <stdlib.h>

namespace cap {

namespace {

constexpr int kThreads = 256;
constexpr uint32_t kDigitTile = 1024;    // scalars per msm_digits_local workgroup (its entries are sorted in LDS)
constexpr int kDigitThreads = 1024;       // one scalar per thread: 16 waves hide the LDS-atomic latency
// A bucket's list is cut into work items of at most item_len entries (one msm_accumulate thread each).  item_len
// is chosen per launch: long items (one per bucket at the prover's sizes) when the batch alone fills the chip,
// short ones when a single MSM has to be spread over all CUs.
constexpr uint32_t kMinItemLen = 32, kMaxItemLen = 512;
constexpr uint32_t kBatchItemCap = 256;  // cap of the batched plans (the deep plan takes the full kMaxItemLen)

// Two schedules of the same field arithmetic (field29.hpp).  G1L (column-wise products, 204 instructions per
// multiplication) is for everything that runs more than a handful of waves.  G1S (row-wise, 220 instructions, 18
// independent accumulators) is for the one-wavefront-per-MSM finishing kernels, single dependent chains of ~30 point
// operations.  (Measured on a single proof's launches, which leave most SIMDs with one wave or none: the row-wise
// schedule made msm_accumulate and msm_reduce_bits 5 % slower there too - only the finishing kernels gain from it.  An
// out-of-line point addition for the tree kernels, one copy instead of three inlined ones: four times slower - the
// operands travel through scratch memory.)
using G1S = G1LT<0>;

__device__ __forceinline__ fl shfl_down_fl(const fl& a, int d) {
  fl r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = __shfl_down(a.v[i], d);
  return r;
}
__device__ __forceinline__ g1x shfl_down_pt(const g1x& a, int d) {
  g1x r;
  r.x = shfl_down_fl(a.x, d);
  r.y = shfl_down_fl(a.y, d);
  r.zz = shfl_down_fl(a.zz, d);
  r.zzz = shfl_down_fl(a.zzz, d);
  return r;
}

// ---- setup: window multiples of every base -----------------------------------------------
// top_shift: the last window's points are 2^(c (W - 1) - top_shift) P instead of 2^(c (W - 1)) P (deep table, see MsmBases)
__global__ __launch_bounds__(kThreads) void msm_precompute_kernel(g1_affine* __restrict__ ext,
                                                                  const g1_affine* __restrict__ bases, size_t n,
                                                                  uint32_t c, uint32_t windows, uint32_t top_shift) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // caller bases are in arkworks' Montgomery form (x * 2^256); the table is kept in the internal form
  // (x * 2^261, canonical) of the lazy 29-bit field
  g1_affine b = bases[i];
  g1a p;
  if (G1::is_inf(b)) {
    p.x = Fq29::zero();
    p.y = Fq29::zero();
  } else {
    p.x = Fq29::canonical(Fq29::from_ext(b.x));
    p.y = Fq29::canonical(Fq29::from_ext(b.y));
  }
  ext[i] = G1L::store_affine(p);
  g1x acc = G1L::from_affine(p);
  for (uint32_t w = 1; w < windows; w++) {
    const uint32_t steps = w + 1 == windows ? c - top_shift : c;
    for (uint32_t k = 0; k < steps; k++) acc = G1L::dbl(acc);
    ext[(size_t)w * n + i] = G1L::store_affine(G1L::to_affine(acc));
  }
}

// ---- K3: digits, sorted per tile in LDS --------------------------------------------------------------------------
// A workgroup owns a tile of kDigitTile scalars of one (sub-)MSM and sorts that tile's (window, scalar) entries by
// bucket entirely in LDS: pass A histograms the digits,
// a workgroup scan turns counts into tile-local offsets, pass B recomputes the digits and drops each table index
// at its slot.  The tile-sorted chunk goes to HBM with coalesced stores together with one row of the
// [bucket][tile] count table and of the tile-local offset table.  A scan of the count table (bucket-major) then
// gives every (bucket, tile) run its final position: no global atomics, no per-entry scattered store.
__device__ __forceinline__ uint32_t msm_digit(const fe& k, uint32_t w, uint32_t c, uint32_t& carry) {
  const uint32_t half = 1u << (c - 1), mask = (1u << c) - 1;
  uint32_t bit = w * c;
  uint32_t limb = bit >> 5, off = bit & 31;
  uint32_t v = 0;
  if (limb < 8) {
    uint64_t two = (uint64_t)k.v[limb] | (limb + 1 < 8 ? ((uint64_t)k.v[limb + 1] << 32) : 0);
    v = (uint32_t)(two >> off) & mask;
  }
  v += carry;
  if (v > half) {
    carry = 1;
    return ((1u << c) - v) | 0x80000000u;  // magnitude | sign
  }
  carry = 0;
  return v;
}

// Compile-time window size: with the loop over the windows unrolled every limb index is a constant and the scalar
// stays in registers (with a run-time c the compiler indexes it through scratch memory: two scratch loads per digit).
template <uint32_t C, uint32_t W>
__device__ __forceinline__ uint32_t msm_digit_ct(const fe& k, uint32_t& carry) {
  constexpr uint32_t half = 1u << (C - 1), mask = (1u << C) - 1;
  constexpr uint32_t bit = W * C, limb = bit >> 5, off = bit & 31;
  uint32_t v = 0;
  if constexpr (limb < 8) {
    uint64_t two = (uint64_t)k.v[limb];
    if constexpr (limb + 1 < 8) two |= (uint64_t)k.v[limb + 1] << 32;
    v = (uint32_t)(two >> off) & mask;
  }
  v += carry;
  if (v > half) {
    carry = 1;
    return ((1u << C) - v) | 0x80000000u;
  }
  carry = 0;
  return v;
}
template <uint32_t C, uint32_t W, uint32_t NW>
__device__ __forceinline__ void all_digits(const fe& k, uint32_t& carry, uint32_t (&d)[NW]) {
  if constexpr (W < NW) {
    d[W] = msm_digit_ct<C, W>(k, carry);
    all_digits<C, W + 1, NW>(k, carry, d);
  }
}

// A tile's entries do not carry table indices but (window, scalar-within-tile): 15 bits.  The consumer of a run knows
// which tile it reads and rebuilds the index  w * srs_n + base + tile * 1024 + t  (tile_entry_index), so the table may
// hold up to 2^31 points whatever the tile format.  Layout of a tile entry:
//   bits 0-9 t, 10-14 w, 15-21 tile (filled in by msm_sort_level2), 24-30 low bucket bits (two-level sort), 31 sign.
__device__ __forceinline__ uint32_t tile_entry_index(uint32_t v, uint32_t tile, size_t srs_n, size_t base) {
  return (uint32_t)((size_t)((v >> 10) & 31u) * srs_n + base + (size_t)tile * kDigitTile + (v & 1023u));
}

// Sub-MSM `sb` of a launch = (MSM b = sb / parts, part = sb % parts): scalars [part * n_sub, part * n_sub + len) of MSM
// b, where len = min(n_sub, n - part * n_sub).
// DEEP (the three-level sort of long single MSMs, below): up to 14 low key bits ride at bit 14 of the entry (the window
// field is then 4 bits: at most 16 windows).
template <uint32_t CT, bool DEEP = false>  // CT: window size known at compile time; 0: taken from the argument c
__global__ __launch_bounds__(kDigitThreads) void msm_digits_local(const fe* __restrict__ scalars, size_t outer_stride,
                                                             uint32_t inner, size_t inner_stride, size_t n_total,
                                                             size_t n_sub, uint32_t parts,
                                                             int montgomery, uint32_t c, uint32_t windows,
                                                             uint32_t nblk, uint32_t batch, uint32_t sub_bits,
                                                             uint32_t top_shift,
                                                             uint32_t* __restrict__ table,
                                                             uint32_t* __restrict__ tloc,
                                                             uint32_t* __restrict__ chunks,
                                                             uint32_t* __restrict__ sync_a, uint32_t n_a,
                                                             uint32_t* __restrict__ sync_b, uint32_t n_b) {
  // the flag words of this launch's chained kernels (msm_scan_chained, msm_items): zeroed here, by the first kernel of
  // the launch, so that no separate memset sits on the path of a small MSM
  if (blockIdx.x == 0) {
    for (uint32_t j = threadIdx.x; j < n_a; j += kDigitThreads) sync_a[j] = 0;
    for (uint32_t j = threadIdx.x; j < n_b; j += kDigitThreads) sync_b[j] = 0;
  }
  // sub_bits == 0: the tile is sorted by bucket.  sub_bits > 0 (two-level sort for wide windows): it is sorted by
  // bin = bucket >> sub_bits only, and the low bucket bits ride in bits 24..30 of the entry for msm_sort_level2.
  extern __shared__ uint32_t lds[];
  const uint32_t half = (1u << (c - 1)) >> sub_bits;  // number of sort keys of this level
  const uint32_t sub_mask = (1u << sub_bits) - 1;
  uint32_t* hist = lds;              // [half]   counts, then running cursors
  uint32_t* loff = lds + half;       // [half]   tile-local exclusive offsets
  uint32_t* buf = lds + 2 * half;    // [kDigitTile * windows] sorted entries
  __shared__ uint32_t wave_tot[kDigitThreads / 64];
  // XCD-aware block order: workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  The tiles of one
  // batch entry write interleaved 4-byte cells of the same table lines ([bucket][tile] layout), so they are all sent
  // to one XCD, where the partial writes merge in that L2 instead of leaving eight L2s as masked partial lines.
  const uint32_t xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  // (a launch of ONE entry has nothing to spread: its grid is just the tiles)
  const uint32_t b = batch == 1 ? 0 : (slot / nblk) * 8 + xcd, blk = batch == 1 ? blockIdx.x : slot % nblk;
  if (b >= batch) return;
  for (uint32_t j = threadIdx.x; j < half; j += kDigitThreads) hist[j] = 0;
  __syncthreads();
  const uint32_t msm = b / parts, part = b % parts;
  const size_t n = n_total - (size_t)part * n_sub < n_sub ? n_total - (size_t)part * n_sub : n_sub;
  const fe* sc = scalars + (size_t)(msm / inner) * outer_stride + (size_t)(msm % inner) * inner_stride +
                 (size_t)part * n_sub;
  static_assert(kDigitTile == kDigitThreads, "one scalar per thread: it stays in registers between the passes");
  const size_t i = (size_t)blk * kDigitTile + threadIdx.x;
  // CT != 0: the digits are extracted once, with compile-time limb indices, and live in registers across both passes
  constexpr uint32_t CTD = CT ? CT : 1;  // (keeps the CT == 0 instantiation free of divisions by zero)
  constexpr uint32_t NWIN = CT ? (256 + CTD - 1) / CTD + (256 % CTD == 0 ? 1 : 0) : 1;
  uint32_t dg[NWIN];
  fe k;
  if (i < n) {
    k = sc[i];
    if (montgomery) k = Fr::from_mont(k);
    if constexpr (DEEP) {
      // the shifted top window (MsmBases::top_shift3) needs scalars below 2^254: a larger one - never what arkworks hands
      // over, but the ABI takes any 256-bit integer - is folded by the group order first (k P = (k - r) P)
      if (top_shift)
        while (k.v[7] >> 30) (void)Fr::sub_mod_raw(k, k);
    }
    // pass A: histogram
    uint32_t carry = 0;
    if constexpr (CT != 0) {
      all_digits<CT, 0, NWIN>(k, carry, dg);
      if constexpr (DEEP) dg[NWIN - 1] <<= top_shift;  // never negative: the top digit is far below 2^(c-1)
#pragma unroll
      for (uint32_t w = 0; w < NWIN; w++) {
        const uint32_t d = dg[w] & 0x7FFFFFFFu;
        if (d) atomicAdd(&hist[(d - 1) >> sub_bits], 1u);
      }
    } else {
      for (uint32_t w = 0; w < windows; w++) {
        uint32_t d = msm_digit(k, w, c, carry) & 0x7FFFFFFFu;
        if (d) atomicAdd(&hist[(d - 1) >> sub_bits], 1u);
      }
    }
  }
  __syncthreads();
  // exclusive scan of hist -> loff (each thread owns half / kDigitThreads consecutive buckets)
  {
    const uint32_t per_t = half / kDigitThreads ? half / kDigitThreads : 1;
    uint32_t j0 = threadIdx.x * per_t;
    uint32_t run = 0;
    for (uint32_t t = 0; t < per_t && j0 + t < half; t++) run += hist[j0 + t];
    uint32_t inc = run;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int d = 1; d < 64; d <<= 1) {
      uint32_t o = __shfl_up(inc, d);
      if ((int)lane >= d) inc += o;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    uint32_t pre = inc - run;
    for (uint32_t w = 0; w < wave; w++) pre += wave_tot[w];
    for (uint32_t t = 0; t < per_t && j0 + t < half; t++) {
      loff[j0 + t] = pre;
      pre += hist[j0 + t];
    }
  }
  __syncthreads();
  // rows of the global tables; cursors start at the local offsets
  for (uint32_t j = threadIdx.x; j < half; j += kDigitThreads) {
    size_t row = ((size_t)b * half + j) * nblk + blk;
    table[row] = hist[j];
    tloc[row] = loff[j];
    hist[j] = loff[j];
  }
  __syncthreads();
  // pass B: place the table indices
  if (i < n) {
    uint32_t carry = 0;
    auto place = [&](uint32_t w, uint32_t d) {
      const uint32_t mag = d & 0x7FFFFFFFu;
      if (mag) {
        uint32_t pos = atomicAdd(&hist[(mag - 1) >> sub_bits], 1u);
        buf[pos] = (w << 10) | threadIdx.x | (((mag - 1) & sub_mask) << (DEEP ? 14 : 24)) | (d & 0x80000000u);
      }
    };
    if constexpr (CT != 0) {
#pragma unroll
      for (uint32_t w = 0; w < NWIN; w++) place(w, dg[w]);
    } else {
      for (uint32_t w = 0; w < windows; w++) place(w, msm_digit(k, w, c, carry));
    }
  }
  __syncthreads();
  // coalesced store of the tile-sorted chunk
  const uint32_t total = hist[half - 1];  // cursor of the last bucket after pass B = number of entries of the tile
  uint32_t* dst = chunks + ((size_t)b * nblk + blk) * ((size_t)kDigitTile * windows);
  for (uint32_t e = threadIdx.x; e < total; e += kDigitThreads) dst[e] = buf[e];
}

// K4: every (bucket, tile) run of a tile-sorted chunk is copied to its final place; runs of one bucket from
// consecutive tiles are adjacent in the destination, so consecutive threads write consecutive bytes.
__global__ __launch_bounds__(kThreads) void msm_scatter_runs(const uint32_t* __restrict__ chunks,
                                                             const uint32_t* __restrict__ table,
                                                             const uint32_t* __restrict__ tloc,
                                                             const uint32_t* __restrict__ off2, size_t per,
                                                             uint32_t half, uint32_t nblk, uint32_t windows,
                                                             size_t total_rows, size_t srs_n, size_t offset,
                                                             size_t n_sub, uint32_t parts,
                                                             uint32_t* __restrict__ sorted,
                                                             uint32_t* __restrict__ counts,
                                                             uint32_t* __restrict__ offsets) {
  size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // (b * half + bucket) * nblk + blk
  if (row >= total_rows) return;
  uint32_t cnt = table[row];
  uint32_t blk = (uint32_t)(row % nblk);
  if (blk == 0) {  // the bucket's list range (what msm_bucket_ranges was a launch of its own for)
    const uint32_t gb = (uint32_t)(row / nblk), o = off2[row];
    const uint32_t next = gb % half + 1 < half ? off2[row + nblk] : off2[row + nblk - 1] + table[row + nblk - 1];
    offsets[gb] = o;
    counts[gb] = next - o;
  }
  if (cnt == 0) return;
  uint32_t b = (uint32_t)(row / ((size_t)half * nblk));
  const uint32_t* src = chunks + ((size_t)b * nblk + blk) * ((size_t)kDigitTile * windows) + tloc[row];
  uint32_t* dst = sorted + (size_t)b * per + off2[row];
  const size_t base = offset + (size_t)(b % parts) * n_sub;
  for (uint32_t e = 0; e < cnt; e++) {
    const uint32_t v = src[e];
    dst[e] = tile_entry_index(v, blk, srs_n, base) | (v & 0x80000000u);  // tile entry -> table index | sign
  }
}

// ---- K4, second level of the two-level sort ---------------------------------------------------------------------
// Wide windows (c = 15: 16384 buckets, 17 instead of 20 digits per 254-bit scalar) cut the mixed additions by a sixth,
// but a tile of 1024 scalars then holds about one entry per bucket: a [bucket][tile] table would be six times larger
// than the entries themselves.  So the tile sort and the run copy above work on bins of 2^sub_bits buckets, and this
// kernel finishes the job: one workgroup per (batch entry, bin) histograms the low bucket bits of the bin's ~4000
// entries in LDS, writes the per-bucket counts / offsets, and places every entry (a few KB per bin: the scattered
// 4-byte stores stay in L2).  Up to kL2Stage entries are staged in LDS between the passes; the rest of an oversized
// bin (skewed scalars) is simply fetched again, so any bin size works.
// Entries of a bin staged in LDS between the two passes (18 KiB).  A bin of the prover's MSMs holds 4352 +- 66 entries;
// the size is chosen so that eight workgroups - the CU's 32 wave slots - fit in the 160 KiB of LDS: the kernel is a
// chain of three global round trips per workgroup, and with 6144 entries (six workgroups per CU) it ran at 7.1 ms per
// step against 5.0 ms now, which is the HBM rate (4.5 TB/s over the entries read and written).
#ifndef CAP_L2_STAGE
#define CAP_L2_STAGE 4608
#endif
constexpr uint32_t kL2Stage = CAP_L2_STAGE;
constexpr uint32_t kL2MaxTiles = 72;  // tiles per sub-MSM the run table in LDS holds (7-bit tile field; n = 2^16 + 3 fits)
constexpr size_t kMaxSubPoints = (size_t)kL2MaxTiles * kDigitTile;  // longer MSMs are cut into parts (choose_plan)

// (8 waves per SIMD = at most 64 VGPRs: without the attribute the compiler takes 78 and the CU holds six workgroups)
#ifndef CAP_L2_WAVES
#define CAP_L2_WAVES __attribute__((amdgpu_waves_per_eu(8, 8)))
#endif
__global__ __launch_bounds__(kThreads) CAP_L2_WAVES void msm_sort_level2(const uint32_t* __restrict__ chunks,
                                                            const uint32_t* __restrict__ table,
                                                            const uint32_t* __restrict__ tloc,
                                                            const uint32_t* __restrict__ off2, size_t per,
                                                            uint32_t bins, uint32_t nblk, uint32_t windows,
                                                            uint32_t sub_bits, size_t srs_n, size_t offset,
                                                            size_t n_sub, uint32_t parts,
                                                            uint32_t* __restrict__ counts,
                                                            uint32_t* __restrict__ offsets,
                                                            uint32_t* __restrict__ sorted) {
  // The bin's entries are read straight from the tile-sorted chunks (one run per tile, ~140 entries each): the
  // run-copy pass of the one-level sort is not needed here.
  __shared__ uint32_t hist[128];  // the low bucket bits travel in 7 bits of the entry: sub_bits <= 7
  __shared__ uint32_t start[128];
  __shared__ uint32_t run_pre[kL2MaxTiles + 1];  // exclusive prefix of the run lengths
  __shared__ uint32_t run_src[kL2MaxTiles];      // start of each run inside the chunk buffer of the batch entry
  __shared__ uint32_t stage[kL2Stage];
  const uint32_t b = blockIdx.y, bin = blockIdx.x;
  const uint32_t nsub = 1u << sub_bits;
  const size_t row0 = ((size_t)b * bins + bin) * nblk;
  const size_t tile_words = (size_t)kDigitTile * windows;
  const uint32_t* cbase = chunks + (size_t)b * nblk * tile_words;
  // run table: the loads go out in parallel (a serial walk over ~33 dependent global loads per workgroup was the
  // critical path of this kernel), the prefix over the LDS copy is cheap
  for (uint32_t t = threadIdx.x; t < nblk; t += kThreads) {
    run_src[t] = (uint32_t)(t * tile_words) + tloc[row0 + t];
    run_pre[t + 1] = table[row0 + t];
  }
  for (uint32_t j = threadIdx.x; j < nsub; j += kThreads) hist[j] = 0;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t acc = 0;
    run_pre[0] = 0;
    for (uint32_t t = 1; t <= nblk; t++) {
      acc += run_pre[t];
      run_pre[t] = acc;
    }
  }
  __syncthreads();
  const uint32_t cnt = run_pre[nblk], off = off2[row0];
  // p-th entry of the bin.  A thread asks for increasing p (stride 256 against runs of ~130 entries), so the run is
  // found by walking on from the previous one - about two steps - instead of a binary search over all runs per entry
  uint32_t walk = 0;
  auto fetch = [&](uint32_t p) {
    while (walk + 1 < nblk && run_pre[walk + 1] <= p) walk++;
    return cbase[run_src[walk] + (p - run_pre[walk])] | (walk << 15);  // the entry remembers the tile it came from
  };
  const size_t base = offset + (size_t)(b % parts) * n_sub;
  auto final_entry = [&](uint32_t v) {  // tile entry -> table index | sign
    return tile_entry_index(v, (v >> 15) & 127u, srs_n, base) | (v & 0x80000000u);
  };
  // pass A: the first kL2Stage entries stay in registers (24 per thread), the tail of an oversized bin is re-fetched
  constexpr int kPerThread = kL2Stage / kThreads;
  const uint32_t staged = cnt < kL2Stage ? cnt : kL2Stage;
  uint32_t ev[kPerThread];
#pragma unroll
  for (int i = 0; i < kPerThread; i++) {
    const uint32_t p = threadIdx.x + i * kThreads;
    if (p < staged) {
      ev[i] = fetch(p);
      atomicAdd(&hist[(ev[i] >> 24) & (nsub - 1)], 1u);
    }
  }
  for (uint32_t p = kL2Stage + threadIdx.x; p < cnt; p += kThreads) atomicAdd(&hist[(fetch(p) >> 24) & (nsub - 1)], 1u);
  __syncthreads();
  if (threadIdx.x < 64) {  // exclusive scan of up to 256 counters by one wave
    const uint32_t per_lane = (nsub + 63) / 64;
    uint32_t j0 = threadIdx.x * per_lane, run = 0;
    for (uint32_t t = 0; t < per_lane && j0 + t < nsub; t++) run += hist[j0 + t];
    uint32_t inc = run;
    for (int d = 1; d < 64; d <<= 1) {
      uint32_t o = __shfl_up(inc, d);
      if ((int)threadIdx.x >= d) inc += o;
    }
    uint32_t pre = inc - run;
    for (uint32_t t = 0; t < per_lane && j0 + t < nsub; t++) {
      start[j0 + t] = pre;
      pre += hist[j0 + t];
    }
  }
  __syncthreads();
  for (uint32_t j = threadIdx.x; j < nsub; j += kThreads) {
    const size_t gb = ((size_t)b * bins + bin) * nsub + j;
    counts[gb] = hist[j];
    offsets[gb] = off + start[j];
  }
  __syncthreads();  // start[] now serves as the placement cursor
  // pass B: entries are placed in LDS (positions beyond the stage go straight to HBM) and leave with coalesced stores
  uint32_t* dst = sorted + (size_t)b * per + off;
  auto place = [&](uint32_t v) {
    const uint32_t pos = atomicAdd(&start[(v >> 24) & (nsub - 1)], 1u);
    if (pos < kL2Stage) stage[pos] = final_entry(v);
    else dst[pos] = final_entry(v);
  };
#pragma unroll
  for (int i = 0; i < kPerThread; i++)
    if (threadIdx.x + i * kThreads < staged) place(ev[i]);
  walk = 0;
  for (uint32_t p = kL2Stage + threadIdx.x; p < cnt; p += kThreads) place(fetch(p));
  __syncthreads();
  for (uint32_t p = threadIdx.x; p < staged; p += kThreads) dst[p] = stage[p];
}

// ---- K4 for long single MSMs: three-level ("deep") sort ---------------------------------------------------------
// A single MSM of 2^20 .. 2^24 points wants ONE bucket set with windows of c = log2(n) - 2 bits: 12-15 digits per scalar
// instead of the 17 of the c = 15 table, and a bucket reduction that is noise against n * W / 2^(c-1) ~ 100 additions
// per bucket.  Its W n entries must then be sorted by a key of K = c - 1 = 17 .. 21 bits.  Three counting sorts in LDS:
//   L1  msm_digits_local<CT, DEEP>: a tile of 1024 scalars by the top `top` bits of the bucket (<= 128 super-bins; runs
//       of ~100-500 entries per tile and super-bin)
//   L2  msm_deep_sort<0>: workgroup (super-bin S, group g of G tiles) gathers the tiles' runs for S (~12-16 K entries)
//       and sorts them by the next `mid` bits; it writes them back to the range the runs occupy in the S-major order,
//       with one row of the [S][mid][group] count table
//   L3  msm_deep_sort<1>: workgroup (S, mid) gathers its run from every group - the ~12 K entries of 32-128 buckets -
//       sorts by the low 5-7 bits and emits the final bucket lists (table index | sign), bucket counts and offsets.
// Entries stay 4 bytes throughout because position carries the rest: a tile entry is (t, w, low key bits); L2 adds the
// tile's number within its group (7 bits), L3 knows the group from the run it reads.
//   after L1:  t 0-9 | w 10-13 | the key's mid + low bits 14-27 | sign 31
//   after L2:  t 0-9 | w 10-13 | the key's low bits (5-7 of them) from 14 | tile in group 21-27 | sign 31
// Everything downstream sees the 2^K buckets as 2^(K-14) batch entries of 16384 buckets (msm_accumulate, the running-sum
// or bit-plane reductions), whose (sum, weighted sum) pairs one more msm_reduce_final folds with the entry offsets.
constexpr uint32_t kDeepThreads = 1024;   // 16 waves share a 64 KiB stage: two workgroups per CU fill its wave slots
constexpr uint32_t kDeepStage = 16384;    // entries staged in LDS between the passes (a bin beyond it: see level 2 sort)
constexpr uint32_t kDeepMaxRuns = 512;    // runs a workgroup gathers (tiles per group <= 128; groups per MSM <= 512)
// buckets per downstream batch entry.  (4096 - a quarter of the msm_reduce_final chain - was measured: the reduction
// 0.67 -> 0.43 ms, but msm_accumulate 16.3 -> 17.4 ms at 2^24 points: its items are length-sorted per entry.)
constexpr uint32_t kDeepEntryBucketsDefault = 16384;
uint32_t deep_entry_buckets() {
  const char* e = getenv("CAPGPU_MSM_DEEP_ENTRY");
  const int x = e ? atoi(e) : 0;
  return x == 4096 || x == 8192 || x == 16384 ? (uint32_t)x : kDeepEntryBucketsDefault;
}
// Entries per work item of the deep plan: about four items per bucket, 64 or 128 entries.  Measured at 2^24 points
// (416 entries per bucket; msm_accumulate + msm_combine + msm_sort_items): items of 512 -> 16.4 ms, 256 -> 15.9, 128 -> 15.6,
// 64 -> 15.7; at 2^22 (104 per bucket): 512 -> 4.34, 128 -> 4.34, 64 -> 4.26.  Shorter items mean more, shorter waves:
// the launch's tail is shorter and the chip's 3 waves per SIMD are kept full for longer.
uint32_t deep_item_len(size_t avg_bucket) {
  const char* e = getenv("CAPGPU_MSM_DEEP_ITEM");
  const int x = e ? atoi(e) : 0;
  if (x >= 32 && x <= 512) return (uint32_t)x;
  return avg_bucket / 4 > 64 ? 128u : 64u;
}
#define kDeepEntryBuckets deep_entry_buckets()
// buckets per running-sum segment of the deep plan: its 2^16 .. 2^21 buckets must spread over the chip as short chains
// (16: 32 dependent additions per thread, 1024 segments per entry for the one-wave finish)
constexpr uint32_t kDeepSegLen = 16;
// buckets per reduction entry (256 segments: 4 per lane of the entry's msm_reduce_final wave; the last launch then folds
// 2^(K-12) <= 512 (sum, weighted sum) pairs)
constexpr uint32_t kDeepReduceBuckets = 4096;

template <int FINAL>
__global__ __launch_bounds__(kDeepThreads) void msm_deep_sort(const uint32_t* __restrict__ src,
                                                             const uint32_t* __restrict__ run_cnt,
                                                             const uint32_t* __restrict__ run_loc,
                                                             const uint32_t* __restrict__ off2,
                                                             const uint32_t* __restrict__ off3, uint32_t nblk,
                                                             uint32_t G, uint32_t groups, uint32_t M, uint32_t low,
                                                             uint32_t tile_words, size_t srs_n, size_t base_index,
                                                             uint32_t* __restrict__ dst, uint32_t* __restrict__ cnt_out,
                                                             uint32_t* __restrict__ loc_out) {
  // FINAL: the key is (bucket within the bin, window pair) - up to 128 x 8 counters - so that every bucket's list comes
  // out ordered by window: the lanes of msm_accumulate then walk the W n x 64 B table (14 GB at 2^24 points) two
  // windows, ~2 GB, at a time instead of gathering across all of it.  (8 KiB of counters + the 64 KiB stage: two
  // workgroups still fit a CU.)
  constexpr uint32_t kKeys = FINAL ? 1024 : 128;
  __shared__ uint32_t hist[kKeys];
  __shared__ uint32_t start[kKeys];
  __shared__ uint32_t run_pre[kDeepMaxRuns + 1];  // run lengths, then their inclusive prefix (run_pre[0] = 0)
  __shared__ uint32_t run_src[kDeepMaxRuns];
  __shared__ uint32_t coarse[kDeepStage / 256];  // the run that holds staged position b * 256
  __shared__ uint32_t stage[kDeepStage];
  const uint32_t S = blockIdx.y, tid = threadIdx.x;
  uint32_t R, nkeys, out_base;
  if (FINAL == 0) {
    const uint32_t t0 = blockIdx.x * G;
    R = nblk - t0 < G ? nblk - t0 : G;
    nkeys = M;
    for (uint32_t r = tid; r < R; r += kDeepThreads) {
      const size_t row = (size_t)S * nblk + t0 + r;
      run_src[r] = (t0 + r) * tile_words + run_loc[row];
      run_pre[r + 1] = run_cnt[row];
    }
    out_base = off2[(size_t)S * nblk + t0];
  } else {
    R = groups;
    nkeys = 8u << low;
    for (uint32_t g = tid; g < R; g += kDeepThreads) {
      const size_t row = ((size_t)S * M + blockIdx.x) * groups + g;
      run_src[g] = off2[(size_t)S * nblk + (size_t)g * G] + run_loc[row];
      run_pre[g + 1] = run_cnt[row];
    }
    out_base = off3[((size_t)S * M + blockIdx.x) * groups];
  }
  for (uint32_t j = tid; j < kKeys; j += kDeepThreads) hist[j] = 0;
  __syncthreads();
  if (tid < 64) {  // inclusive prefix of the run lengths by one wave
    const uint32_t per = (R + 63) / 64, j0 = tid * per;
    uint32_t run = 0;
    for (uint32_t t = 0; t < per && j0 + t < R; t++) run += run_pre[j0 + t + 1];
    uint32_t inc = run;
    for (int d = 1; d < 64; d <<= 1) {
      uint32_t o = __shfl_up(inc, d);
      if ((int)tid >= d) inc += o;
    }
    uint32_t pre = inc - run;
    for (uint32_t t = 0; t < per && j0 + t < R; t++) {
      pre += run_pre[j0 + t + 1];
      run_pre[j0 + t + 1] = pre;
    }
    if (tid == 0) run_pre[0] = 0;
  }
  __syncthreads();
  const uint32_t cnt = run_pre[R];
  // p-th entry of the gathered runs -> (entry for the next level, its sort key)
  auto find_run = [&](uint32_t p) -> uint32_t {
    uint32_t lo = 0, hi = R;
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (run_pre[mid] <= p) lo = mid;
      else hi = mid;
    }
    return lo;
  };
  auto fetch_in = [&](uint32_t p, uint32_t lo, uint32_t& key) -> uint32_t {  // entry p, known to lie in run lo
    const uint32_t v = src[run_src[lo] + (p - run_pre[lo])];
    if (FINAL == 0) {
      key = (v >> (14 + low)) & (M - 1);
      // the mid bits are spent; the entry keeps its low key bits and now names its tile within the group
      return (v & (0x80003FFFu | (((1u << low) - 1) << 14))) | (lo << 21);
    }
    key = (((v >> 14) & ((1u << low) - 1)) << 3) | ((v >> 11) & 7u);
    const size_t tile = (size_t)lo * G + ((v >> 21) & 127u);
    return (uint32_t)((size_t)((v >> 10) & 15u) * srs_n + base_index + tile * kDigitTile + (v & 1023u)) |
           (v & 0x80000000u);
  };
  auto fetch = [&](uint32_t p, uint32_t& key) -> uint32_t { return fetch_in(p, find_run(p), key); };
  if (tid < kDeepStage / 256) coarse[tid] = tid * 256 < cnt ? find_run(tid * 256) : 0;
  __syncthreads();
  constexpr int kPer = kDeepStage / kDeepThreads;
  const uint32_t staged = cnt < kDeepStage ? cnt : kDeepStage;
  uint32_t ev[kPer], keys[kPer / 2];  // 16-bit keys, two per register
#pragma unroll
  for (int i = 0; i < kPer / 2; i++) keys[i] = 0;
  // Entry p = tid + i * 1024 (a wave reads 256 contiguous bytes per load).  Its run is found from a coarse table - the
  // run that holds position b * 256, one binary search per 256 positions - and a short walk, instead of a binary search
  // (8-9 dependent LDS reads) per entry.
  {
#pragma unroll
    for (int i = 0; i < kPer; i++) {
      const uint32_t p = tid + i * kDeepThreads;
      if (p < staged) {
        uint32_t run = coarse[p >> 8];
        while (run + 1 < R && run_pre[run + 1] <= p) run++;
        uint32_t k;
        ev[i] = fetch_in(p, run, k);
        keys[i >> 1] |= k << (16 * (i & 1));
        atomicAdd(&hist[k], 1u);
      }
    }
  }
  for (uint32_t p = kDeepStage + tid; p < cnt; p += kDeepThreads) {  // an oversized bin (skewed scalars): counted now,
    uint32_t k;                                                       // fetched again for the placement
    (void)fetch(p, k);
    atomicAdd(&hist[k], 1u);
  }
  __syncthreads();
  if (tid < 64) {  // exclusive scan of the counters by one wave (kKeys / 64 consecutive ones per lane)
    constexpr uint32_t per = kKeys / 64;
    uint32_t run = 0;
#pragma unroll 4
    for (uint32_t t = 0; t < per; t++) run += hist[tid * per + t];
    uint32_t inc = run;
    for (int d = 1; d < 64; d <<= 1) {
      uint32_t o = __shfl_up(inc, d);
      if ((int)tid >= d) inc += o;
    }
    uint32_t pre = inc - run;
#pragma unroll 4
    for (uint32_t t = 0; t < per; t++) {
      start[tid * per + t] = pre;
      pre += hist[tid * per + t];
    }
  }
  __syncthreads();
  if (FINAL == 0) {
    if (tid < nkeys) {
      const size_t row = ((size_t)S * M + tid) * groups + blockIdx.x;
      cnt_out[row] = hist[tid];
      loc_out[row] = start[tid];
    }
  } else if (tid < (1u << low)) {  // a bucket = its 8 consecutive (bucket, window pair) keys
    const size_t gb = (((size_t)S * M + blockIdx.x) << low) + tid;
    const uint32_t k0 = tid << 3;
    cnt_out[gb] = (k0 + 8 < nkeys ? start[k0 + 8] : cnt) - start[k0];
    loc_out[gb] = out_base + start[k0];
  }
  __syncthreads();  // start[] now serves as the placement cursor
  uint32_t* out = dst + out_base;
#pragma unroll
  for (int i = 0; i < kPer; i++) {
    if (tid + i * kDeepThreads < staged) {
      const uint32_t pos = atomicAdd(&start[(keys[i >> 1] >> (16 * (i & 1))) & 0xFFFFu], 1u);
      if (pos < kDeepStage) stage[pos] = ev[i];
      else out[pos] = ev[i];
    }
  }
  for (uint32_t p = kDeepStage + tid; p < cnt; p += kDeepThreads) {
    uint32_t k;
    const uint32_t e = fetch(p, k);
    const uint32_t pos = atomicAdd(&start[k], 1u);
    if (pos < kDeepStage) stage[pos] = e;
    else out[pos] = e;
  }
  __syncthreads();
  for (uint32_t p = tid; p < staged; p += kDeepThreads) out[p] = stage[p];
}

// ---- K4: exclusive scan, one workgroup per batch entry ------------------------------------------
// ITEMS == 0: scans the bucket counts (-> list offsets).  ITEMS == 1: scans ceil(count / item_len), the number of
// work items of each bucket (-> item offsets inside the batch entry); the per-entry total goes to totals[].
template <int ITEMS>
__global__ __launch_bounds__(1024) void msm_scan(const uint32_t* __restrict__ counts, uint32_t* __restrict__ offsets,
                                                 uint32_t nb, uint32_t* __restrict__ totals, uint32_t item_len) {
  constexpr uint32_t E = 8;  // consecutive elements per thread
  __shared__ uint32_t sh[1024];
  __shared__ uint32_t carry_s;
  const uint32_t* cnt = counts + (size_t)blockIdx.x * nb;
  uint32_t* off = offsets + (size_t)blockIdx.x * nb;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (uint32_t base = 0; base < nb; base += 1024 * E) {
    uint32_t idx0 = base + threadIdx.x * E;
    uint32_t v[E];
    uint32_t run = 0;
#pragma unroll
    for (uint32_t t = 0; t < E; t++) {
      uint32_t x = idx0 + t < nb ? cnt[idx0 + t] : 0;
      if (ITEMS) x = (x + item_len - 1) / item_len;
      v[t] = run;
      run += x;
    }
    sh[threadIdx.x] = run;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
      uint32_t add = threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
      __syncthreads();
      sh[threadIdx.x] += add;
      __syncthreads();
    }
    uint32_t incl = sh[threadIdx.x];
    uint32_t carry = carry_s;
    uint32_t pre = carry + incl - run;
#pragma unroll
    for (uint32_t t = 0; t < E; t++)
      if (idx0 + t < nb) off[idx0 + t] = pre + v[t];
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = carry + incl;
    __syncthreads();
  }
  if (totals && threadIdx.x == 0) totals[blockIdx.x] = carry_s;
}

// ---- K4 for long tables (a single large MSM): the same exclusive scan in three parallel steps --------------------
// One workgroup per batch entry walks its table serially in 8192-element strides; that is fine when a launch holds
// hundreds of entries, but a single 2^17-point MSM has a 524 288-row table and spent 0.55 ms in that walk.  Here every
// 8192-element segment is scanned by its own workgroup, the segment totals are scanned, and the bases are added.
constexpr uint32_t kScanSeg = 8192;
__global__ __launch_bounds__(1024) void msm_scan_seg(const uint32_t* __restrict__ counts, uint32_t* __restrict__ offsets,
                                                     uint32_t nb, uint32_t nseg, uint32_t* __restrict__ seg_tot) {
  __shared__ uint32_t sh[1024];
  const uint32_t b = blockIdx.y, sg = blockIdx.x;
  const uint32_t* cnt = counts + (size_t)b * nb;
  uint32_t* off = offsets + (size_t)b * nb;
  const uint32_t idx0 = sg * kScanSeg + threadIdx.x * 8;
  uint32_t v[8], run = 0;
#pragma unroll
  for (uint32_t t = 0; t < 8; t++) {
    v[t] = run;
    run += idx0 + t < nb ? cnt[idx0 + t] : 0;
  }
  sh[threadIdx.x] = run;
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d <<= 1) {
    uint32_t add = threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
    __syncthreads();
    sh[threadIdx.x] += add;
    __syncthreads();
  }
  const uint32_t pre = sh[threadIdx.x] - run;
#pragma unroll
  for (uint32_t t = 0; t < 8; t++)
    if (idx0 + t < nb) off[idx0 + t] = pre + v[t];
  if (threadIdx.x == 1023) seg_tot[(size_t)b * nseg + sg] = sh[1023];
}
__global__ __launch_bounds__(1024) void msm_scan_tot(uint32_t* __restrict__ seg_tot, uint32_t nseg) {
  __shared__ uint32_t sh[1024];
  __shared__ uint32_t carry_s;
  uint32_t* t = seg_tot + (size_t)blockIdx.x * nseg;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (uint32_t base = 0; base < nseg; base += 1024) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t x = i < nseg ? t[i] : 0;
    sh[threadIdx.x] = x;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
      uint32_t add = threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
      __syncthreads();
      sh[threadIdx.x] += add;
      __syncthreads();
    }
    const uint32_t carry = carry_s;
    if (i < nseg) t[i] = carry + sh[threadIdx.x] - x;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = carry + sh[1023];
    __syncthreads();
  }
}
__global__ __launch_bounds__(1024) void msm_scan_add(uint32_t* __restrict__ offsets, uint32_t nb, uint32_t nseg,
                                                     const uint32_t* __restrict__ seg_tot) {
  const uint32_t b = blockIdx.y, sg = blockIdx.x;
  const uint32_t base = seg_tot[(size_t)b * nseg + sg];
  uint32_t* off = offsets + (size_t)b * nb;
  for (uint32_t i = sg * kScanSeg + threadIdx.x; i < (sg + 1) * kScanSeg && i < nb; i += 1024) off[i] += base;
}

// ---- chained forms for small launches (round 5) ------------------------------------------------------------------
// A single 2^17-point MSM spent 0.14 of its 0.49 ms in nine short launches in front of the accumulation - digits, three
// scan kernels, ranges, scatter, item scan, item bases, item sort - each ~6 us of launch and drain for a few microseconds
// of work.  The three scan kernels become ONE (every segment publishes its total behind a flag and adds up the totals
// of the segments before it: a chained scan), the ranges ride in the scatter kernel, and the three item kernels become
// ONE (msm_items).  Workgroups take their index from a ticket counter, so a workgroup only ever waits for workgroups
// that are already running; the flag words are zeroed by msm_digits_local, the first kernel of the launch.
constexpr uint32_t kFlag = 0x80000000u;  // totals stay below 2^31 (batch_slice keeps every per-launch counter in 31 bits)

// exclusive prefix of `v` over the 1024 threads of a workgroup (wave shuffles + one pass over the 16 wave totals);
// returns the prefix, *total = the sum.  sh: 17 words.
__device__ __forceinline__ uint32_t block_scan_1024(uint32_t v, uint32_t* sh, uint32_t* total) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(inc, d);
    if ((int)lane >= d) inc += o;
  }
  __syncthreads();  // (sh may still be read from a previous call)
  if (lane == 63) sh[wave] = inc;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (int w = 0; w < 16; w++) {
      const uint32_t x = sh[w];
      sh[w] = run;
      run += x;
    }
    sh[16] = run;
  }
  __syncthreads();
  *total = sh[16];
  return sh[wave] + inc - v;
}
// sum over j < count of the totals published in state[j] (spinning until each is there); all 1024 threads call it
__device__ __forceinline__ uint32_t chained_base(const uint32_t* state, uint32_t count, uint32_t* sh) {
  uint32_t acc = 0;
  for (uint32_t j = threadIdx.x; j < count; j += 1024) {
    uint32_t x;
    do {
      x = __hip_atomic_load(&state[j], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    } while (!(x & kFlag));
    acc += x & ~kFlag;
  }
  uint32_t tot;
  (void)block_scan_1024(acc, sh, &tot);
  return tot;
}
// exclusive scan of nb counters per batch entry in ONE launch: grid = nseg * batch workgroups of 8192 counters
__global__ __launch_bounds__(1024) void msm_scan_chained(const uint32_t* __restrict__ counts, uint32_t* __restrict__ offsets,
                                                         uint32_t nb, uint32_t nseg, uint32_t* state /*[batch][nseg]*/,
                                                         uint32_t* ticket) {
  __shared__ uint32_t sh[17];
  __shared__ uint32_t tk;
  if (threadIdx.x == 0) tk = atomicAdd(ticket, 1u);
  __syncthreads();
  const uint32_t b = tk / nseg, sg = tk % nseg;
  const uint32_t* cnt = counts + (size_t)b * nb;
  uint32_t* off = offsets + (size_t)b * nb;
  const uint32_t idx0 = sg * kScanSeg + threadIdx.x * 8;
  uint32_t v[8], run = 0;
#pragma unroll
  for (uint32_t t = 0; t < 8; t++) {
    v[t] = run;
    run += idx0 + t < nb ? cnt[idx0 + t] : 0;
  }
  uint32_t total;
  const uint32_t pre = block_scan_1024(run, sh, &total);
  if (threadIdx.x == 0) __hip_atomic_store(&state[(size_t)b * nseg + sg], total | kFlag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  const uint32_t base = chained_base(state + (size_t)b * nseg, sg, sh);
#pragma unroll
  for (uint32_t t = 0; t < 8; t++)
    if (idx0 + t < nb) off[idx0 + t] = base + pre + v[t];
}

// bucket = sum of its work items, G lanes per bucket: lanes take items, a shuffle tree adds them up.  For the
// small-batch path, where there are too few buckets for one thread each (one thread per bucket is then a serial chain
// on a handful of waves: 1.2 ms for 2^17 points).
// a += b in the reduction trees: an operand at infinity (an empty bucket, a padding lane) is a skip or a copy decided
// on zz, everything else goes through the lean addition; equal or opposite operands - never with real data - fall back
// to the general one.  (tools/ubench_lonewave.hip: 6.5 against 7.8 us per dependent addition; one wave per SIMD already
// runs at 91 % of what two or three reach together, so these chains only get faster by executing fewer instructions.)
template <class G>
__device__ __forceinline__ void add_tree(g1x& a, const g1x& b) {
  if (G::is_inf(b)) return;
  if (G::is_inf(a)) {
    a = b;
    return;
  }
  if (__builtin_expect(!G::add_acc(a, b), 0)) a = G::add(a, b);
}

template <int G>
__global__ __launch_bounds__(kThreads) void msm_combine_wave(const g1_xyzz* __restrict__ item_pts,
                                                             const uint32_t* __restrict__ counts,
                                                             const uint32_t* __restrict__ item_off,
                                                             const uint32_t* __restrict__ item_base, uint32_t half,
                                                             uint32_t total_buckets, uint32_t item_len,
                                                             g1_xyzz* __restrict__ buckets) {
  const uint32_t gb = (blockIdx.x * blockDim.x + threadIdx.x) / G, lane = threadIdx.x % G;
  // whole groups leave together (total_buckets * G threads are launched in whole groups), so the shuffles below
  // always see their partners
  if (gb >= total_buckets) return;
  const uint32_t items = (counts[gb] + item_len - 1) / item_len;
  const uint32_t first = item_base[gb / half] + item_off[gb];
  g1x acc = G1L::inf();
  if (items != 1) {  // single-item buckets were written by msm_accumulate
    if (lane < items) acc = G1L::load(item_pts[first + lane]);
    for (uint32_t j = lane + G; j < items; j += G) add_tree<G1L>(acc, G1L::load(item_pts[first + j]));
  }
  for (int d = G / 2; d >= 1; d >>= 1) {
    g1x o = shfl_down_pt(acc, d);
    // (lanes d .. G - d - 1 add too, uselessly; restricting the addition to lane < d was measured 20 % slower here)
    if (lane + d < G) add_tree<G1L>(acc, o);
  }
  if (lane == 0 && items != 1) buckets[gb] = G1L::store(acc);
}

// item_base[b] = sum of totals[0..b), item_base[batch] = number of work items of the whole launch.
// One workgroup of 1024 threads: block scans over slabs of 1024 entries (a long MSM run as thousands of sub-MSMs made
// the former single-thread loop a visible serial step).
__global__ __launch_bounds__(1024) void msm_item_bases(const uint32_t* __restrict__ totals, uint32_t batch,
                                                       uint32_t* __restrict__ item_base) {
  __shared__ uint32_t sh[1024];
  __shared__ uint32_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (uint32_t base = 0; base < batch; base += 1024) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t x = i < batch ? totals[i] : 0;
    sh[threadIdx.x] = x;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
      uint32_t add = threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
      __syncthreads();
      sh[threadIdx.x] += add;
      __syncthreads();
    }
    const uint32_t carry = carry_s;
    if (i < batch) item_base[i] = carry + sh[threadIdx.x] - x;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = carry + sh[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    item_base[batch] = carry_s;
    item_base[batch + 1] = 0;  // chunk counter of a dynamic msm_accumulate launch
    item_base[batch + 2] = 0;  // length of msm_combine's list of heavy buckets
  }
}

// Work items of one batch entry, ordered by length (longest first): slot -> (bucket, sub-item).  Bucket sizes are
// Poisson-spread (160 +- 13 entries at the prover's sizes), and a wavefront runs as long as its longest item; with
// the items of a wave drawn from a sorted list every lane runs (nearly) the same number of additions.  One
// workgroup per batch entry; counting sort over the item lengths in LDS.
__global__ __launch_bounds__(1024) void msm_sort_items(const uint32_t* __restrict__ counts,
                                                       const uint32_t* __restrict__ item_base, uint32_t half,
                                                       uint32_t item_len, uint32_t* __restrict__ item_bucket,
                                                       uint32_t* __restrict__ item_sub) {
  __shared__ uint32_t hist[kMaxItemLen + 1];
  __shared__ uint32_t cursor[kMaxItemLen + 1];
  const uint32_t b = blockIdx.x;
  for (uint32_t l = threadIdx.x; l <= kMaxItemLen; l += blockDim.x) hist[l] = 0;
  __syncthreads();
  for (uint32_t j = threadIdx.x; j < half; j += blockDim.x) {
    const uint32_t cnt = counts[(size_t)b * half + j];
    if (cnt == 0) continue;
    const uint32_t items = (cnt + item_len - 1) / item_len, q = cnt / items, r = cnt - q * items;
    if (r) atomicAdd(&hist[q + 1], r);
    atomicAdd(&hist[q], items - r);
  }
  __syncthreads();
  // Even batch entries list their items longest first, odd ones shortest first.  Workgroups go to the 8 XCDs
  // round-robin and a batch entry's item range is a whole number of workgroups at the prover's sizes, so with one
  // direction only, XCD 0 would receive the longest workgroup of every entry and XCD 7 the shortest.
  // (an exclusive scan of the 513 length classes by the whole workgroup - one thread walking them took 15 of the
  // kernel's 22-32 us, on the path of every launch)
  {
    __shared__ uint32_t scan[1024];
    const uint32_t t = threadIdx.x;
    const uint32_t l = (b & 1) ? t : kMaxItemLen - t;  // the class this thread holds, in listing order (t > 512: none)
    const uint32_t v = t <= kMaxItemLen ? hist[l] : 0;
    scan[t] = v;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
      const uint32_t add = t >= d ? scan[t - d] : 0;
      __syncthreads();
      scan[t] += add;
      __syncthreads();
    }
    if (t <= kMaxItemLen) cursor[l] = scan[t] - v;
  }
  __syncthreads();
  const uint32_t ib = item_base[b];
  for (uint32_t j = threadIdx.x; j < half; j += blockDim.x) {
    const uint32_t gb = b * half + j;
    const uint32_t cnt = counts[gb];
    if (cnt == 0) continue;
    const uint32_t items = (cnt + item_len - 1) / item_len, q = cnt / items, r = cnt - q * items;
    if (r) {
      uint32_t pos = ib + atomicAdd(&cursor[q + 1], r);
      for (uint32_t k = 0; k < r; k++) {
        item_bucket[pos + k] = gb;
        item_sub[pos + k] = k;
      }
    }
    uint32_t pos = ib + atomicAdd(&cursor[q], items - r);
    for (uint32_t k = r; k < items; k++) {
      item_bucket[pos + k - r] = gb;
      item_sub[pos + k - r] = k;
    }
  }
}

// msm_scan<1> + msm_item_bases + msm_sort_items in ONE launch (see "chained forms" above): workgroup b scans the item
// counts of its batch entry (-> item_off), publishes the entry's total behind a flag, adds up the totals of the entries
// before it (-> item_base[b]) and lays down its length-sorted item list.  state: [batch] flag words + the ticket counter
// at [batch], zeroed by msm_digits_local.
__global__ __launch_bounds__(1024) void msm_items(const uint32_t* __restrict__ counts, uint32_t half, uint32_t item_len,
                                                  uint32_t batch, uint32_t* __restrict__ item_off,
                                                  uint32_t* __restrict__ item_base, uint32_t* state,
                                                  uint32_t* __restrict__ item_bucket, uint32_t* __restrict__ item_sub) {
  __shared__ uint32_t hist[kMaxItemLen + 1];
  __shared__ uint32_t cursor[kMaxItemLen + 1];
  __shared__ uint32_t sh[17];
  __shared__ uint32_t tk;
  if (threadIdx.x == 0) tk = atomicAdd(state + batch, 1u);
  for (uint32_t l = threadIdx.x; l <= kMaxItemLen; l += 1024) hist[l] = 0;
  __syncthreads();
  const uint32_t b = tk;
  const uint32_t* cnt_b = counts + (size_t)b * half;
  // item counts per bucket: exclusive scan in bucket order (8 consecutive buckets per thread and slab) + length histogram
  uint32_t carry = 0;
  for (uint32_t base = 0; base < half; base += 1024 * 8) {
    const uint32_t j0 = base + threadIdx.x * 8;
    uint32_t v[8], run = 0;
#pragma unroll
    for (uint32_t t = 0; t < 8; t++) {
      const uint32_t cnt = j0 + t < half ? cnt_b[j0 + t] : 0;
      const uint32_t items = (cnt + item_len - 1) / item_len;
      v[t] = run;
      run += items;
      if (cnt) {
        const uint32_t q = cnt / items, r = cnt - q * items;
        if (r) atomicAdd(&hist[q + 1], r);
        atomicAdd(&hist[q], items - r);
      }
    }
    uint32_t tot;
    const uint32_t pre = block_scan_1024(run, sh, &tot);
#pragma unroll
    for (uint32_t t = 0; t < 8; t++)
      if (j0 + t < half) item_off[(size_t)b * half + j0 + t] = carry + pre + v[t];
    carry += tot;
  }
  if (threadIdx.x == 0) __hip_atomic_store(&state[b], carry | kFlag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  // cursors of the length classes: even entries list their items longest first, odd ones shortest first (msm_sort_items)
  {
    const uint32_t t = threadIdx.x;
    const uint32_t l = (b & 1) ? t : kMaxItemLen - t;
    const uint32_t v = t <= kMaxItemLen ? hist[l] : 0;
    uint32_t tot;
    const uint32_t pre = block_scan_1024(v, sh, &tot);
    if (t <= kMaxItemLen) cursor[l] = pre;
  }
  const uint32_t ib = chained_base(state, b, sh);  // (ends with barriers: the cursors are visible)
  if (threadIdx.x == 0) {
    item_base[b] = ib;
    if (b + 1 == batch) {
      item_base[batch] = ib + carry;
      item_base[batch + 1] = 0;  // chunk counter of a dynamic msm_accumulate launch
      item_base[batch + 2] = 0;  // length of msm_combine's list of heavy buckets
    }
  }
  for (uint32_t j = threadIdx.x; j < half; j += 1024) {
    const uint32_t gb = b * half + j;
    const uint32_t cnt = cnt_b[j];
    if (cnt == 0) continue;
    const uint32_t items = (cnt + item_len - 1) / item_len, q = cnt / items, r = cnt - q * items;
    if (r) {
      const uint32_t pos = ib + atomicAdd(&cursor[q + 1], r);
      for (uint32_t k = 0; k < r; k++) {
        item_bucket[pos + k] = gb;
        item_sub[pos + k] = k;
      }
    }
    const uint32_t pos = ib + atomicAdd(&cursor[q], items - r);
    for (uint32_t k = r; k < items; k++) {
      item_bucket[pos + k - r] = gb;
      item_sub[pos + k - r] = k;
    }
  }
}

// ---- K5: bucket accumulation --------------------------------------------------------------------
// One thread per work item (a slice of one bucket's list).  The slices of a bucket are of (almost) equal length
// and the items of a wavefront come from a length-sorted list (msm_sort_items), so every lane of a wavefront runs
// practically the same number of mixed additions whatever the bucket sizes are; a skewed scalar distribution (one
// giant bucket) is cut into many items instead of one serial chain.
// Occupancy does not move it (147 VGPRs: three waves per SIMD).  Measured per step: a fourth wave -
// amdgpu_waves_per_eu(4, 4), 128 VGPRs, 7 of them spilled - 115.9 against 116.3 ms (round 2: 130.73 vs 130.75); two waves
// (64 KiB of dynamic LDS per workgroup, CAPGPU_ACC_LDS) 116.7 against 113.9; four waves WITHOUT scratch spills - the
// accumulator's y, zz, zzz kept in LDS between their uses, 27 limbs read twice and written once per addition - 126.7
// against 114.9 (the extra instructions and LDS round trips cost more than the fourth wave gives).  tools/ubench_chain.hip
// shows what the issue rate of v_mad_u64_u32 does with the waves on a SIMD - 1: 17 T, 2: 34 T, 3: 28 T, 4+: 35-36 T
// lane-operations/s, whatever the number of independent chains - but this kernel sits at 0.79-0.80 of its per-class
// issue floor at two, three and four waves alike, and a two-stage software pipeline of its gathers (list entry two steps
// ahead, table point one ahead: 168 VGPRs) changes nothing at three waves nor at two (119.6 vs 119.4, 121.5 vs 122.1 ms):
// what it loses is lost per instruction, not to occupancy or memory latency.  (The same one-entry-ahead gather for the deep
// plan alone, whose 14 GB table at 2^24 points lives in HBM rather than in the Infinity Cache: 15.5 ms with and without.)
#ifdef CAP_ACC_WAVES  // experiments: occupancy the compiler budgets registers for
#define CAP_ACC_ATTR __attribute__((amdgpu_waves_per_eu(CAP_ACC_WAVES, CAP_ACC_WAVES)))
#else
#define CAP_ACC_ATTR
#endif
__global__ __launch_bounds__(kThreads) CAP_ACC_ATTR void msm_accumulate(const g1_affine* __restrict__ ext,
                                                           const uint32_t* __restrict__ sorted,
                                                           const uint32_t* __restrict__ counts,
                                                           const uint32_t* __restrict__ offsets,
                                                           const uint32_t* __restrict__ item_off,
                                                           const uint32_t* __restrict__ item_base,
                                                           const uint32_t* __restrict__ item_bucket,
                                                           const uint32_t* __restrict__ item_sub, size_t per,
                                                           uint32_t half, uint32_t batch, uint32_t item_len,
                                                           uint32_t* chunk_counter, g1_xyzz* __restrict__ item_pts,
                                                           g1_xyzz* __restrict__ buckets) {
  // Grid-stride over the items: with a grid of a few workgroups per CU (accumulate_persistent) a workgroup walks chunk
  // b, b + G, b + 2G, ... itself instead of being dispatched once per chunk.  G is a multiple of 8, so a chunk stays on the
  // XCD the one-shot launch would put it on.
  const uint32_t n_items = item_base[batch];
  __shared__ uint32_t chunk_s;
  for (uint32_t round = 0;; round++) {
    uint32_t it;
    if (chunk_counter) {  // the next chunk of 256 items from a counter: a workgroup that finishes early takes more
      // (the counter is the word behind the item bases - zeroed with them - passed as a pointer of its own: item_base
      // is const __restrict__ and must not be written through)
      if (threadIdx.x == 0) chunk_s = atomicAdd(chunk_counter, 1u);
      __syncthreads();
      const uint32_t chunk = chunk_s;
      __syncthreads();
      if ((uint64_t)chunk * blockDim.x >= n_items) break;
      it = chunk * blockDim.x + threadIdx.x;
    } else {
      it = (blockIdx.x + round * gridDim.x) * blockDim.x + threadIdx.x;
      if ((uint64_t)(blockIdx.x + round * gridDim.x) * blockDim.x >= n_items) break;
    }
    if (it >= n_items) continue;
  const uint32_t gb = item_bucket[it], j = item_sub[it];
  const uint32_t b = gb / half;
  const uint32_t cnt = counts[gb];
  const uint32_t items = (cnt + item_len - 1) / item_len;
  const uint32_t q = cnt / items, r = cnt - q * items;
  const uint32_t lo = j * q + (j < r ? j : r), hi = lo + q + (j < r ? 1u : 0u);
  const uint32_t* lst = sorted + (size_t)b * per + offsets[gb];
  g1x acc = G1L::inf();
  // Common path kept lean: the loop body is G1L::madd_acc - the XYZZ mixed addition with ONE test.  Everything else -
  // the accumulator still at infinity (first entry of the item, or after P + (-P)), P + P - makes the x-difference
  // vanish, a table point at infinity is caught by the OR over its limbs, and both go to the general G1L::add_mixed
  // in a branch of its own, so that its selects and copies never sit in the hot block (2650 -> 2260 instructions per
  // addition on the common path).
  for (uint32_t e = lo; e < hi; e++) {
    const uint32_t v = lst[e];
    const g1a pt = G1L::load(ext[v & 0x7FFFFFFFu]);
    const bool negate = (v >> 31) != 0;
    if (__builtin_expect(G1L::is_inf(pt) || !G1L::madd_acc(acc, pt, negate), 0)) acc = G1L::add_mixed(acc, pt, negate);
  }
  // a bucket made of a single item is final: it goes straight to the bucket array and msm_combine skips it
  if (items == 1) buckets[gb] = G1L::store(acc);
  else item_pts[item_base[b] + item_off[gb] + j] = G1L::store(acc);
  }
}

// ---- K6: bucket reduction by running sums ------------------------------------------------------------------
// sum_j (j + 1) B_j.  One thread per segment of seg_len consecutive buckets walks it from the top with S += B_j,
// T += S, so that after the walk  S = sum B_j,  T = sum (j - j_lo + 1) B_j : two additions per bucket, where the
// bit-plane reduction below spends (c - 1) / 2.
__global__ __launch_bounds__(kThreads) void msm_reduce_segments(const g1_xyzz* __restrict__ buckets, uint32_t half,
                                                                uint32_t seg_len, uint32_t nseg, uint32_t batch,
                                                                g1_xyzz* __restrict__ seg_pts) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nseg * batch) return;
  const uint32_t b = t / nseg, sg = t - b * nseg;
  const uint32_t j_lo = sg * seg_len;
  uint32_t j_hi = j_lo + seg_len;
  if (j_hi > half) j_hi = half;
  const g1_xyzz* bk = buckets + (size_t)b * half;
  // The walk from infinity starts with S = T = the top bucket.  Then the fast loop: both additions in their lean form
  // (G1L::add_acc: one test, no special cases in the hot block).  The two special cases that do occur - an empty bucket
  // and a running sum still at infinity: with 8 entries per bucket (a 2^20-point MSM cut into 128 parts) every other
  // wave has such a lane - are a skipped addition or a copy, decided on the zz coordinate before the addition; they
  // must not throw the lane out of the loop, or the wave walks the rest of that lane's segment a second time (measured:
  // the launch 2.3 times longer).  Only equal or opposite operands (add_acc fails) leave it, for the general loop below.
  g1x S = G1L::load(bk[j_hi - 1]);
  g1x T = S;
  g1_xyzz cur = bk[j_hi - 1 > j_lo ? j_hi - 2 : j_hi - 1];
  for (uint32_t j = j_hi - 1; j > j_lo; j--) {  // buckets j - 1, ..., j_lo are still to be walked
    const g1_xyzz nxt = bk[j - 1 > j_lo ? j - 2 : j - 1];  // in flight while the two additions below run
    const g1x c = G1L::load(cur);
    if (!G1L::is_inf(c)) {
      if (G1L::is_inf(S)) S = c;
      else if (__builtin_expect(!G1L::add_acc(S, c), 0)) S = G1L::add(S, c);
    }
    if (!G1L::is_inf(S)) {
      if (G1L::is_inf(T)) T = S;
      else if (__builtin_expect(!G1L::add_acc(T, S), 0)) T = G1L::add(T, S);  // T == S: an empty bucket right after the first
    }
    cur = nxt;
  }
  seg_pts[2 * (size_t)t] = G1L::store(S);
  seg_pts[2 * (size_t)t + 1] = G1L::store(T);
}

template <class G = G1L>
__device__ __forceinline__ g1x wave_sum(g1x v) {
  // Level d needs the sums of lanes < d only.  Letting every lane add is not harmless: a lane beyond the range gets its
  // own value back from the shuffle, P + P sends it down the doubling branch of add(), and the whole wave pays for that
  // branch at every level.
  const uint32_t lane = threadIdx.x & 63;
  for (int d = 32; d >= 1; d >>= 1) {
    g1x o = shfl_down_pt(v, d);
    if (lane < (uint32_t)d) add_tree<G>(v, o);
  }
  return v;  // lane 0 holds the sum
}
template <class G = G1L>
__device__ g1x mul_small(const g1x& p, uint32_t k) {
  g1x r = G::inf();
  for (int bit = 31; bit >= 0; bit--) {
    if (!G::is_inf(r)) r = G::dbl(r);
    if ((k >> bit) & 1) r = G::add(r, p);
  }
  return r;
}

// One wavefront per batch entry: total = sum_s T_s + seg_len * sum_s s S_s.  Lane l owns q consecutive segments and
// repeats the running-sum walk one level up; the lane weights come from a suffix scan across the wave:
// sum_l l S_l = sum_{m >= 1} (S_m + S_{m+1} + ... + S_63).  The whole kernel is one dependent chain of ~30 point
// operations per launch, so it is written for depth, not for work.
// out (one Jacobian point per MSM) or, when the MSMs of the launch are parts of longer ones, out_part (XYZZ, summed by
// msm_sum_parts)
// (312 VGPRs, one wave per SIMD; forcing two - 256 VGPRs, 343 spilled - was measured: 1.56 -> 1.46 ms per step, not adopted)
// out_pair (deep plan): the entry's buckets are a slice of one larger bucket set, so the plain sum of its buckets is
// wanted beside the weighted one: out_pair[2 b] = sum_j B_j, out_pair[2 b + 1] = sum_j (j + 1) B_j - the (S, T) pair of
// a segment one level up, folded by one more launch of this kernel.
__global__ __launch_bounds__(64) void msm_reduce_final(const g1_xyzz* __restrict__ seg_pts, uint32_t seg_len,
                                                       uint32_t nseg, g1_jac* __restrict__ out,
                                                       g1_xyzz* __restrict__ out_part, g1_xyzz* __restrict__ out_pair) {
  const uint32_t b = blockIdx.x, lane = threadIdx.x;
  const uint32_t q = (nseg + 63) / 64;
  const uint32_t s_lo = lane * q;
  uint32_t s_hi = s_lo + q;
  if (s_hi > nseg) s_hi = nseg;
  const g1_xyzz* sp = seg_pts + 2 * (size_t)b * nseg;
  g1x S = G1S::inf(), T = G1S::inf(), A = G1S::inf();
  for (uint32_t s = s_hi; s > s_lo;) {
    s--;
    add_tree<G1S>(T, S);                      // every segment above s gains one more unit of weight
    add_tree<G1S>(S, G1S::load(sp[2 * s]));
    add_tree<G1S>(A, G1S::load(sp[2 * s + 1]));
  }
  // S = sum S_s, T = sum (s - s_lo) S_s, A = sum T_s over the lane's segments
  g1x suf = S;  // inclusive suffix sum over lanes
  for (int d = 1; d < 64; d <<= 1) {
    g1x o = shfl_down_pt(suf, d);
    if (lane + d < 64) add_tree<G1S>(suf, o);
  }
  const g1x all = suf;  // lane 0: the sum of every segment
  if (lane == 0) suf = G1S::inf();
  // per lane: A + seg_len * (T + q * suf); their sum over the wave is the result
  g1x r = G1S::add(T, mul_small<G1S>(suf, q));
  r = G1S::add(mul_small<G1S>(r, seg_len), A);
  r = wave_sum<G1S>(r);
  if (lane == 0) {
    if (out_pair) {
      out_pair[2 * (size_t)b] = G1S::store(all);
      out_pair[2 * (size_t)b + 1] = G1S::store(r);
    } else if (out_part) {
      out_part[b] = G1S::store(r);
    } else {
      out[b] = G1S::to_jac_ext(r);
    }
  }
}

// ---- K6 for small batches --------------------------------------------------------------------------------------
// The running-sum walk needs thousands of independent segments to fill the chip; one large MSM (or a handful) does
// not have them.  There the buckets are combined per bucket and reduced through bit planes, both log-depth.
// msm_reduce_bits splits the buckets of a sub-MSM into chunks, one workgroup per (chunk, bit plane): up to 16 chunks of
// at least 512 buckets - 256 of them in a plane, one per thread - so that a workgroup's chain of dependent additions is
// 1-2 + 6 (wave) + 2 (workgroup) instead of the 8 + 6 + 2 of one workgroup per plane; msm_reduce_bits_final folds the
// chunks of a plane with a 16-lane shuffle tree.  (A single proof's four MSM launches: reduce_bits + final 1.16 -> 1.11 ms,
// a 2^17-point MSM 0.83 -> 0.80 ms: every tree level of the final kernel costs what a serial step of the first saved.)
constexpr uint32_t kReduceMaxChunks = 16;
// ... as many chunks as keep the launch at about one workgroup per CU (c * chunks * sub-MSMs <= 320): a second wave of
// this code on a SIMD buys little (measured: 8 chunks for a 5-MSM launch made every addition 35 % slower)
uint32_t reduce_chunks(uint32_t half, uint32_t c, uint32_t sb) {
  const uint32_t cap = std::min<uint32_t>(kReduceMaxChunks, std::max<uint32_t>(1, half / 512));
  uint32_t chunks = 1;
  while (chunks * 2 <= cap && (uint64_t)c * (chunks * 2) * sb <= 320) chunks *= 2;
  return chunks;
}
// bucket = sum of its work items (a handful of full additions per bucket)
__global__ __launch_bounds__(kThreads) void msm_combine(const g1_xyzz* __restrict__ item_pts,
                                                        const uint32_t* __restrict__ counts,
                                                        const uint32_t* __restrict__ item_off,
                                                        const uint32_t* __restrict__ item_base, uint32_t half,
                                                        uint32_t total_buckets, uint32_t item_len,
                                                        g1_xyzz* __restrict__ buckets, uint32_t heavy_min,
                                                        uint32_t* __restrict__ heavy_list,
                                                        uint32_t* __restrict__ heavy_count) {
  uint32_t gb = blockIdx.x * blockDim.x + threadIdx.x;
  if (gb >= total_buckets) return;
  uint32_t items = (counts[gb] + item_len - 1) / item_len;
  if (items == 1) return;  // written by msm_accumulate
  // A bucket of many items (a skewed scalar distribution; the deep table's top window with its digits spread over the
  // bucket range: 66 items in one bucket of every 512) is not walked here - one long lane per wave, 5.7 ms at 2^24 points -
  // but handed to msm_combine_heavy, 32 lanes per bucket.
  if (heavy_list && items >= heavy_min) {
    heavy_list[atomicAdd(heavy_count, 1u)] = gb;
    return;
  }
  uint32_t first = item_base[gb / half] + item_off[gb];
  g1x acc = G1L::inf();
  if (items) {
    acc = G1L::load(item_pts[first]);
    g1_xyzz cur = item_pts[first + (items > 1 ? 1 : 0)];
    for (uint32_t j = 1; j < items; j++) {
      const g1_xyzz nxt = item_pts[first + (j + 1 < items ? j + 1 : j)];  // in flight during the addition
      acc = G1L::add(acc, G1L::load(cur));
      cur = nxt;
    }
  }
  buckets[gb] = G1L::store(acc);
}

// ---- K6a: bit-plane sums ---------------------------------------------------------------------------
// grid (chunks, c, batch).  partial[(b*c + bit)*chunks + chunk] = sum of buckets j in the chunk with
// bit `bit` of (j+1) set.  The launch is a few dozen workgroups, each a dependent chain of point additions: 8 serial
// + 6 shuffle + 2 cross-wave.  (512 threads - a chain of 4 + 6 + 3 - were measured slower, 0.93 against 0.79 ms over a
// single proof's four launches: two waves of this code on a SIMD run at little more than the speed of one.)
constexpr int kReduceThreads = 256;
// planes = c, or c + 1: plane c is then the plain sum of the chunk's buckets (deep plan, see msm_reduce_final)
__global__ __launch_bounds__(kReduceThreads) void msm_reduce_bits(const g1_xyzz* __restrict__ buckets, uint32_t half,
                                                                  uint32_t c, uint32_t planes, uint32_t chunks,
                                                                  g1_xyzz* __restrict__ partial) {
  constexpr int kWaves = kReduceThreads / 64;
  __shared__ g1_xyzz sh[kWaves];
  const uint32_t chunk = blockIdx.x, bit = blockIdx.y, b = blockIdx.z;
  const g1_xyzz* bk = buckets + (size_t)b * half;
  g1x acc = G1L::inf();
  // enumerate only the weights v = j + 1 in [1, half] that have `bit` set: v = idx with a 1 inserted at `bit`
  // (every lane does useful work; a predicate on j would leave half the lanes idle for the low bits); the chunk owns
  // the indices [chunk * per_chunk, (chunk + 1) * per_chunk) of the half / 2 such weights
  if (bit >= c) {
    const uint32_t per_all = (half + chunks - 1) / chunks;
    for (uint32_t q = threadIdx.x; q < per_all; q += kReduceThreads) {
      const uint32_t j = chunk * per_all + q;
      if (j < half) add_tree<G1L>(acc, G1L::load(bk[j]));
    }
  } else {
    const uint32_t per_chunk = (half / 2 + chunks - 1) / chunks;
    for (uint32_t q = threadIdx.x; q < per_chunk; q += kReduceThreads) {
      const uint32_t idx = chunk * per_chunk + q;
      const uint32_t v = ((idx >> bit) << (bit + 1)) | (1u << bit) | (idx & ((1u << bit) - 1));
      if (idx < half / 2 && v <= half) add_tree<G1L>(acc, G1L::load(bk[v - 1]));  // (the top plane holds v = half only)
    }
  }
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int d = 32; d >= 1; d >>= 1) {
    g1x o = shfl_down_pt(acc, d);
    if (lane < (uint32_t)d) add_tree<G1L>(acc, o);  // see wave_sum
  }
  if (lane == 0) sh[wave] = G1L::store(acc);
  __syncthreads();
  if (wave == 0) {  // the wave sums: a shuffle tree over the first kWaves lanes
    g1x r = lane < kWaves ? G1L::load(sh[lane]) : G1L::inf();
    for (int d = kWaves / 2; d >= 1; d >>= 1) {
      g1x o = shfl_down_pt(r, d);
      if (lane < (uint32_t)d) add_tree<G1L>(r, o);
    }
    if (lane == 0) partial[((size_t)b * planes + bit) * chunks + chunk] = G1L::store(r);
  }
}

// ---- K6b: sum_b 2^b T_b ------------------------------------------------------------------------------
// One workgroup of 256 threads per batch entry.  Thread (bit, k) = (t / 16, t % 16) takes chunk k of plane `bit`; a
// 16-lane shuffle tree folds the chunks of each plane (planes never straddle a wave: four of them per wave), then the
// first wave has T_bit in lane `bit`, doubles it `bit` times and adds the planes up.  c <= 16, chunks <= 16.
__global__ __launch_bounds__(256) void msm_reduce_bits_final(const g1_xyzz* __restrict__ partial, uint32_t c,
                                                        uint32_t nplanes, uint32_t chunks, g1_jac* __restrict__ out,
                                                        g1_xyzz* __restrict__ out_part,
                                                        g1_xyzz* __restrict__ out_pair) {
  __shared__ g1_xyzz planes[16];
  const uint32_t b = blockIdx.x, t = threadIdx.x, bit = t >> 4, k = t & 15;
  g1x acc = G1S::inf();
  if (bit < nplanes && k < chunks) acc = G1S::load(partial[((size_t)b * nplanes + bit) * chunks + k]);
  for (int d = 8; d >= 1; d >>= 1) {
    g1x o = shfl_down_pt(acc, d);
    if (k < (uint32_t)d && k + d < chunks) add_tree<G1S>(acc, o);
  }
  if (k == 0 && bit < 16) planes[bit] = G1S::store(acc);
  __syncthreads();
  if (t >= 64) return;
  const uint32_t lane = t;
  acc = lane < c ? G1S::load(planes[lane]) : G1S::inf();
  for (uint32_t j = 0; j + 1 < c; j++) {
    g1x d = G1S::dbl(acc);
    if (lane > j && lane < c) acc = d;
  }
  for (int d = 8; d >= 1; d >>= 1) {
    g1x o = shfl_down_pt(acc, d);
    if (lane < (uint32_t)d) add_tree<G1S>(acc, o);  // see wave_sum
  }
  if (lane == 0) {
    if (out_pair) {
      out_pair[2 * (size_t)b] = planes[c];  // the plain-sum plane
      out_pair[2 * (size_t)b + 1] = G1S::store(acc);
    } else if (out_part) {
      out_part[b] = G1S::store(acc);
    } else {
      out[b] = G1S::to_jac_ext(acc);
    }
  }
}

// ---- K6 for small batches, second form: the bucket index as a 2-D grid -------------------------------------------------
// sum_j (j + 1) B_j with j = hi * 2^lo_bits + lo:
//     = 2^lo_bits * sum_hi hi * R_hi  +  sum_lo (lo + 1) * C_lo,      R_hi = sum_lo B[hi][lo],  C_lo = sum_hi B[hi][lo].
// The bit-plane form above adds every bucket into (c - 1) / 2 plane sums on average - 13 x 2048 additions for 4096
// buckets, through trees in which half of a wave's lanes idle at every level - and finishes with c - 1 dependent doublings.
// Here every bucket enters TWO sums (its row and its column: 2 x 4096 additions, 8 of them serial per lane, then a
// 3-level tree over 8 lanes), and the two weighted sums of 64 terms that remain take no doublings at all: with the terms
// in the lanes of a wave, sum_k k X_k = sum_{l >= 1} suffix_l - a suffix scan across the lanes and one wave sum; only
// the factor 2^lo_bits costs lo_bits = 6 doublings.  A single proof's four MSM launches (1, 2 and 5 MSMs of 4096 buckets):
// see DESIGN.md round-4 log for the measured times.
// grid: (ceil(sb * 2 * (R_hi + R_lo)... one thread per (entry, sum, slice)
// Lanes per row / column sum: each adds dim / slices buckets serially, then a log2(slices)-level tree folds them.  More
// slices for launches with lanes to spare - 64 for a single proof's launches: 6 tree levels instead of 8 + 3 additions -
// were measured in round 4 and are no gain (single proof 3.33 -> 3.36-3.41 ms, batches of 2 and 4 within noise): a tree
// level costs what a serial addition costs, plus its 36 shuffles.  CAPGPU_MSM_GRID_SLICES=64 brings them back.
uint32_t grid_slices(uint32_t sb, uint32_t nsum, uint32_t dim) {
  static const uint32_t cap = [] {
    const char* e = getenv("CAPGPU_MSM_GRID_SLICES");
    const int x = e ? atoi(e) : 8;
    return (uint32_t)(x == 16 || x == 32 || x == 64 ? x : 8);
  }();
  uint32_t sl = 8;
  while (sl < cap && sl * 2 <= dim && (uint64_t)sb * nsum * (sl * 2) <= 65536) sl *= 2;
  return sl;
}
template <uint32_t slices>
__global__ __launch_bounds__(kThreads) void msm_reduce_grid(const g1_xyzz* __restrict__ buckets, uint32_t half,
                                                            uint32_t lo_bits, uint32_t sb,
                                                            g1_xyzz* __restrict__ sums /* [sb][rows + cols] */) {
  const uint32_t cols = 1u << lo_bits, rows = half >> lo_bits, nsum = rows + cols;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t g = t / slices, q = t % slices;  // g: (entry, sum); whole groups leave together (slices divides 64)
  if (g >= sb * nsum) return;
  const uint32_t b = g / nsum, sidx = g - b * nsum;
  const g1_xyzz* bk = buckets + (size_t)b * half;
  g1x acc = G1L::inf();
  if (sidx < rows) {  // row sum R_hi: buckets hi * cols + lo, lo in this lane's slice (consecutive 128-byte entries)
    const uint32_t per = (cols + slices - 1) / slices, l0 = q * per;
    for (uint32_t l = l0; l < l0 + per && l < cols; l++) add_tree<G1L>(acc, G1L::load(bk[(size_t)sidx * cols + l]));
  } else {  // column sum C_lo: buckets hi * cols + lo, hi in this lane's slice
    const uint32_t lo = sidx - rows;
    const uint32_t per = (rows + slices - 1) / slices, h0 = q * per;
    for (uint32_t h = h0; h < h0 + per && h < rows; h++) add_tree<G1L>(acc, G1L::load(bk[(size_t)h * cols + lo]));
  }
  for (int d = (int)slices / 2; d >= 1; d >>= 1) {
    g1x o = shfl_down_pt(acc, d);
    if (q < (uint32_t)d) add_tree<G1L>(acc, o);
  }
  if (q == 0) sums[g] = G1L::store(acc);
}

// (the form for grids of up to 128 x 128: the wide table's 16384 buckets, launches of 24 .. 63 MSMs; grids of at most
// 64 x 64 - a single proof's launches, where every microsecond of this chain is a microsecond of the proof - take
// msm_reduce_grid_final_small below: one barrier instead of two, no pairing of waves: 155 against 180 us per launch)
constexpr bool BIG = true;
__global__ __launch_bounds__(256) void msm_reduce_grid_final(const g1_xyzz* __restrict__ sums, uint32_t half,
                                                             uint32_t lo_bits, g1_jac* __restrict__ out,
                                                             g1_xyzz* __restrict__ out_part) {
  __shared__ g1_xyzz part[4][2];  // [wave][weighted sum, plain sum]
  const uint32_t cols = 1u << lo_bits, rows = half >> lo_bits;
  const uint32_t b = blockIdx.x, lane = threadIdx.x & 63;
  const uint32_t wave = threadIdx.x >> 6;  // waves 0, 1: rows (low, high terms), 2, 3: columns
  const bool is_col = wave >= 2;
  const uint32_t cnt = is_col ? cols : rows;
  const uint32_t k = (wave & 1) * 64 + lane;  // the term this lane holds
  const g1_xyzz* sp = sums + (size_t)b * (rows + cols) + (is_col ? rows : 0);
  g1x suf = k < cnt ? G1S::load(sp[k]) : G1S::inf();
  // nothing to do for a wave whose terms do not exist (dimensions <= 64): its sums are infinity
  const bool live = (wave & 1) * 64 < cnt;
  g1x w = G1S::inf(), plain = G1S::inf();
  if (live) {
    for (int d = 1; d < 64; d <<= 1) {  // suf_l = X_l + X_{l+1} + ... within the wave (lanes beyond cnt hold infinity)
      g1x o = shfl_down_pt(suf, d);
      if (lane + d < 64) add_tree<G1S>(suf, o);
    }
    plain = suf;  // lane 0: the wave's plain sum
    if (lane == 0) suf = G1S::inf();  // sum_{l >= 1} suf_l = sum_l l X_l
    w = wave_sum<G1S>(suf);
  }
  if (lane == 0) {
    part[wave][0] = G1S::store(w);
    part[wave][1] = G1S::store(plain);
  }
  __syncthreads();
  const bool finisher = lane == 0 && !(wave & 1);  // lane 0 of waves 0 and 2 finish their dimension
  if (finisher) {
    g1x tot = G1S::load(part[wave][0]);
    if (BIG && cnt > 64) {
      g1x hi = G1S::load(part[wave + 1][1]);  // 64 * S_hi
      for (int d = 0; d < 6; d++)
        if (!G1S::is_inf(hi)) hi = G1S::dbl(hi);
      add_tree<G1S>(tot, hi);
      add_tree<G1S>(tot, G1S::load(part[wave + 1][0]));
    }
    if (is_col) {
      // columns weigh lo + 1: one more plain sum of every column
      add_tree<G1S>(tot, G1S::load(part[2][1]));
      if (BIG && cnt > 64) add_tree<G1S>(tot, G1S::load(part[3][1]));
    } else {
      for (uint32_t d = 0; d < lo_bits; d++)
        if (!G1S::is_inf(tot)) tot = G1S::dbl(tot);
    }
    part[wave][0] = G1S::store(tot);  // (read above by this lane only)
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    g1x r = G1S::load(part[0][0]);
    add_tree<G1S>(r, G1S::load(part[2][0]));
    if (out_part) out_part[b] = G1S::store(r);
    else out[b] = G1S::to_jac_ext(r);
  }
}
// Grids of at most 64 x 64: one workgroup of two waves per entry - wave 0 takes the row sums (weights hi = 0 .. rows - 1),
// wave 1 the column sums (weights lo + 1 = 1 .. cols).  Inclusive suffix scan over the lanes, then the wave sum of the
// suffixes - from lane 1 on for the rows, from lane 0 on for the columns.  A chain of 6 + 6 additions, lo_bits
// doublings and one addition.
__global__ __launch_bounds__(128) void msm_reduce_grid_final_small(const g1_xyzz* __restrict__ sums, uint32_t half,
                                                                   uint32_t lo_bits, g1_jac* __restrict__ out,
                                                                   g1_xyzz* __restrict__ out_part) {
  __shared__ g1_xyzz row_part;
  const uint32_t cols = 1u << lo_bits, rows = half >> lo_bits;
  const uint32_t b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const g1_xyzz* sp = sums + (size_t)b * (rows + cols) + (wave ? rows : 0);
  const uint32_t cnt = wave ? cols : rows;
  g1x suf = lane < cnt ? G1S::load(sp[lane]) : G1S::inf();
  for (int d = 1; d < 64; d <<= 1) {  // suf_l = X_l + X_{l+1} + ... (lanes beyond cnt hold infinity)
    g1x o = shfl_down_pt(suf, d);
    if (lane + d < 64) add_tree<G1S>(suf, o);
  }
  if (wave == 0 && lane == 0) suf = G1S::inf();  // rows: weight hi starts at 0, so the full sum (lane 0) is left out
  g1x r = wave_sum<G1S>(suf);
  if (wave == 0) {
    for (uint32_t k = 0; k < lo_bits; k++)
      if (!G1S::is_inf(r)) r = G1S::dbl(r);
    if (lane == 0) row_part = G1S::store(r);
  }
  __syncthreads();
  if (wave == 1 && lane == 0) {
    add_tree<G1S>(r, G1S::load(row_part));
    if (out_part) out_part[b] = G1S::store(r);
    else out[b] = G1S::to_jac_ext(r);
  }
}
// ---- K5 / K6 tails of SMALL launches on quads: one point per four lanes (quad29.hpp) ---------------------------------
// A single proof's launches (1, 2 and 5 MSMs of 4096 buckets) spend most of their time in three chains of dependent point
// additions - bucket = sum of its items, the row / column sums of the grid, the two weighted sums of 64 terms - on a chip
// with lanes to spare.  The quad form of the same kernels runs every addition four multiplications deep instead of twelve.
// Same sums, same order of the operands where it matters (none: the group is commutative and the results leave as affine
// coordinates); taken for launches of up to quad_max_batch() MSMs, the one-lane kernels above stay for everything else.
using QD = QuadG1<G1S, QuadDev>;
__device__ __noinline__ g1x quad_slow_add(const g1x& a, const g1x& b) { return G1S::add(a, b); }
struct QuadSlow {
  __device__ __forceinline__ g1x operator()(const g1x& a, const g1x& b) const { return quad_slow_add(a, b); }
};
__device__ __forceinline__ fl quad_zero() { return G1S::F::zero(); }
__device__ __forceinline__ fl quad_load(const g1_xyzz* p) {
  return G1S::F::load(reinterpret_cast<const fe*>(p)[threadIdx.x & 3]);
}
__device__ __forceinline__ void quad_store(g1_xyzz* p, const fl& v) {
  reinterpret_cast<fe*>(p)[threadIdx.x & 3] = G1S::F::pack(v);
}
// the value of the quad `d` quads up the wave
__device__ __forceinline__ fl quad_shfl_down(const fl& a, int d) { return shfl_down_fl(a, 4 * d); }
// the additions as calls (nine registers per operand: they travel in registers): one copy of the code for all the call
// sites of a kernel
__device__ __noinline__ fl quad_add_call(fl a, fl b) {
  QD::add(a, b, QuadSlow());
  return a;
}
__device__ __noinline__ fl quad_dbl_call(fl a) {
  QD::dbl(a);
  return a;
}

// bucket = sum of its work items, GQ quads per bucket (msm_combine_wave's job: there eight lanes hold eight items and a
// three-level tree adds them; here two quads - the same eight lanes - take the items in turn and one level joins them)
template <int GQ>
__global__ __launch_bounds__(kThreads) void msm_combine_quad(const g1_xyzz* __restrict__ item_pts,
                                                             const uint32_t* __restrict__ counts,
                                                             const uint32_t* __restrict__ item_off,
                                                             const uint32_t* __restrict__ item_base, uint32_t half,
                                                             uint32_t total_buckets, uint32_t item_len,
                                                             g1_xyzz* __restrict__ buckets) {
  const uint32_t quad = (blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  const uint32_t gb = quad / GQ, lane = quad % GQ;
  if (gb >= total_buckets) return;  // whole groups leave together (4 GQ lanes, a divisor of 64)
  const uint32_t items = (counts[gb] + item_len - 1) / item_len;
  const uint32_t first = item_base[gb / half] + item_off[gb];
  fl acc = quad_zero();
  if (items != 1) {  // single-item buckets were written by msm_accumulate
    if (lane < items) acc = quad_load(&item_pts[first + lane]);
    if (lane + GQ < items) {
      fl cur = quad_load(&item_pts[first + lane + GQ]);
#pragma unroll 1
      for (uint32_t j = lane + GQ; j < items; j += GQ) {
        // (the next item is in flight while this one is added)
        const fl nxt = quad_load(&item_pts[first + (j + GQ < items ? j + GQ : j)]);
        QD::add(acc, cur, QuadSlow());
        cur = nxt;
      }
    }
  }
#pragma unroll 1
  for (int d = GQ / 2; d >= 1; d >>= 1) {
    const fl o = quad_shfl_down(acc, d);
    if (lane < (uint32_t)d) QD::add(acc, o, QuadSlow());
  }
  if (lane == 0 && items != 1) quad_store(&buckets[gb], acc);
}

// the buckets msm_combine left on its list (those of kHeavyItems items and more): eight quads per bucket, a fixed grid that
// walks the list
constexpr uint32_t kHeavyItems = 12;
__global__ __launch_bounds__(kThreads) void msm_combine_heavy(const g1_xyzz* __restrict__ item_pts,
                                                              const uint32_t* __restrict__ counts,
                                                              const uint32_t* __restrict__ item_off,
                                                              const uint32_t* __restrict__ item_base, uint32_t half,
                                                              uint32_t item_len, const uint32_t* __restrict__ heavy_list,
                                                              const uint32_t* __restrict__ heavy_count,
                                                              g1_xyzz* __restrict__ buckets) {
  constexpr uint32_t GQ = 8;
  const uint32_t quad = (blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  const uint32_t group = quad / GQ, lane = quad % GQ, groups = (gridDim.x * blockDim.x) / (4 * GQ);
  const uint32_t n_heavy = *heavy_count;
  for (uint32_t w = group; w < n_heavy; w += groups) {  // (uniform over the 32 lanes of a group)
    const uint32_t gb = heavy_list[w];
    const uint32_t items = (counts[gb] + item_len - 1) / item_len;
    const uint32_t first = item_base[gb / half] + item_off[gb];
    fl acc = lane < items ? quad_load(&item_pts[first + lane]) : quad_zero();
    if (lane + GQ < items) {
      fl cur = quad_load(&item_pts[first + lane + GQ]);
#pragma unroll 1
      for (uint32_t j = lane + GQ; j < items; j += GQ) {
        const fl nxt = quad_load(&item_pts[first + (j + GQ < items ? j + GQ : j)]);
        acc = quad_add_call(acc, cur);
        cur = nxt;
      }
    }
#pragma unroll 1
    for (int d = GQ / 2; d >= 1; d >>= 1) {
      const fl o = quad_shfl_down(acc, d);
      if (lane < (uint32_t)d) acc = quad_add_call(acc, o);
    }
    if (lane == 0) quad_store(&buckets[gb], acc);
  }
}

// msm_reduce_grid on quads: `slices` quads per row / column sum
template <uint32_t slices>
__global__ __launch_bounds__(kThreads) void msm_reduce_grid_quad(const g1_xyzz* __restrict__ buckets, uint32_t half,
                                                                 uint32_t lo_bits, uint32_t sb,
                                                                 g1_xyzz* __restrict__ sums /* [sb][rows + cols] */,
                                                                 uint32_t* __restrict__ fin_count /* [sb] */) {
  const uint32_t cols = 1u << lo_bits, rows = half >> lo_bits, nsum = rows + cols;
  const uint32_t quad = (blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  const uint32_t g = quad / slices, q = quad % slices;
  if (g >= sb * nsum) return;
  const uint32_t b = g / nsum, sidx = g - b * nsum;
  const g1_xyzz* bk = buckets + (size_t)b * half;
  // one loop for both kinds of sum (a row walks along the buckets, a column strides over them): one copy of the addition
  const bool row = sidx < rows;
  const uint32_t dim = row ? cols : rows, per = (dim + slices - 1) / slices, i0 = q * per;
  const size_t origin = row ? (size_t)sidx * cols : (size_t)(sidx - rows), stride = row ? 1 : cols;
  const uint32_t i1 = i0 + per < dim ? i0 + per : dim;
  fl acc = quad_zero();
  if (i0 < i1) {
    acc = quad_load(&bk[origin + (size_t)i0 * stride]);
    if (i0 + 1 < i1) {
      fl cur = quad_load(&bk[origin + (size_t)(i0 + 1) * stride]);
#pragma unroll 1
      for (uint32_t i = i0 + 1; i < i1; i++) {
        const fl nxt = quad_load(&bk[origin + (size_t)(i + 1 < i1 ? i + 1 : i) * stride]);  // in flight during the addition
        QD::add(acc, cur, QuadSlow());
        cur = nxt;
      }
    }
  }
#pragma unroll 1
  for (int d = (int)slices / 2; d >= 1; d >>= 1) {
    const fl o = quad_shfl_down(acc, d);
    if (q < (uint32_t)d) QD::add(acc, o, QuadSlow());
  }
  if (q == 0) quad_store(&sums[g], acc);
  if (sidx == 0 && q == 0 && (threadIdx.x & 3) == 0) fin_count[b] = 0;  // for msm_reduce_grid_final_quad
}

// msm_reduce_grid_final_small on quads: TWO workgroups of 256 lanes per entry - 64 quads for the row sums in one, 64 for
// the column sums in the other, so that each runs one wave per SIMD on a CU of its own (eight waves on one CU: 90 us per
// launch, the two chains in each other's way).  The 64 terms of a dimension span four waves, so the suffix scan and the
// tree exchange them through LDS (two buffers in turn: one barrier per step).  Chain: 6 + 6 additions, lo_bits
// doublings, one addition - as before, each a third as deep.  The workgroup that finishes second (a counter per entry,
// zeroed by msm_reduce_grid_quad) adds the other one's result to its own and writes the entry's sum.
// fin: [sb][2] results of the two dimensions, then [sb] counters.
// TERMS: 64 (grids of up to 64 x 64: the narrow table's 4096 buckets) or 128 (the wide table's 16384: eight waves).
template <uint32_t TERMS>
__global__ __launch_bounds__(4 * TERMS) void msm_reduce_grid_final_quad(const g1_xyzz* __restrict__ sums, uint32_t half,
                                                                        uint32_t lo_bits, g1_xyzz* __restrict__ fin,
                                                                        uint32_t* __restrict__ fin_count,
                                                                        g1_jac* __restrict__ out,
                                                                        g1_xyzz* __restrict__ out_part) {
  constexpr uint32_t LOG = TERMS == 64 ? 6 : 7;
  __shared__ fl ex[2][TERMS][4];  // [buffer][term][coordinate]
  __shared__ uint32_t second_s;
  const uint32_t cols = 1u << lo_bits, rows = half >> lo_bits;
  const uint32_t b = blockIdx.x, dim = blockIdx.y, k = threadIdx.x >> 2, c = threadIdx.x & 3;
  const g1_xyzz* sp = sums + (size_t)b * (rows + cols) + (dim ? rows : 0);
  const uint32_t cnt = dim ? cols : rows;
  fl suf = k < cnt ? quad_load(&sp[k]) : quad_zero();
  uint32_t buf = 0;
#pragma unroll 1
  for (uint32_t step = 0; step < 2 * LOG; step++) {
    // steps 0 .. LOG - 1: suf_k += suf_{k + d}, d = 1, 2, .. TERMS / 2 (inclusive suffix sums); then LOG steps for the sum
    // of the suffixes - from term 1 on for the rows (weights hi = 0 .. rows - 1), from term 0 on for the columns (lo + 1)
    if (step == LOG && dim == 0 && k == 0) suf = quad_zero();
    const uint32_t d = step < LOG ? 1u << step : (TERMS / 2) >> (step - LOG);
    const bool take = step < LOG ? k + d < TERMS : k < d;
    ex[buf][k][c] = suf;
    __syncthreads();
    if (take) QD::add(suf, ex[buf][k + d][c], QuadSlow());
    buf ^= 1;
  }
  if (k == 0) {
    if (dim == 0)
      for (uint32_t i = 0; i < lo_bits; i++) QD::dbl(suf);
    quad_store(&fin[2 * (size_t)b + dim], suf);
    __threadfence();  // the result is out before the counter says so
  }
  __syncthreads();
  if (threadIdx.x == 0) second_s = atomicAdd(&fin_count[b], 1u);
  __syncthreads();
  if (second_s == 0 || k != 0) return;
  __threadfence();
  // (read past the caches: the other workgroup ran on another CU, likely on another XCD)
  const uint32_t* op = reinterpret_cast<const uint32_t*>(&fin[2 * (size_t)b + (1 - dim)]) + 8 * c;
  fe o;
#pragma unroll
  for (int i = 0; i < 8; i++) o.v[i] = __hip_atomic_load(op + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  QD::add(suf, G1S::F::load(o), QuadSlow());
  const g1x r = QD::gather(suf);
  if (c == 0) {
    if (out_part) out_part[b] = G1S::store(r);
    else out[b] = G1S::to_jac_ext(r);
  }
}
// msm_reduce_final on quads: 64 quads per entry (one workgroup of 256 lanes), the lane loop of the one-lane kernel per
// quad, the suffix sums and the closing sum through LDS.  The additions are calls here (quad_add_call), one copy of the
// code for the kernel's eight call sites.
__device__ fl quad_mul_small(const fl& p, uint32_t k) {
  fl r = quad_zero();
  bool started = false;
  for (int bit = 31; bit >= 0; bit--) {
    if (started) r = quad_dbl_call(r);
    if ((k >> bit) & 1) {
      r = quad_add_call(r, p);
      started = true;
    }
  }
  return r;
}
__global__ __launch_bounds__(256) void msm_reduce_final_quad(const g1_xyzz* __restrict__ seg_pts, uint32_t seg_len,
                                                            uint32_t nseg, g1_jac* __restrict__ out,
                                                            g1_xyzz* __restrict__ out_part,
                                                            g1_xyzz* __restrict__ out_pair) {
  __shared__ fl ex[2][64][4];
  const uint32_t b = blockIdx.x, lane = threadIdx.x >> 2, c = threadIdx.x & 3;
  const uint32_t q = (nseg + 63) / 64;
  const uint32_t s_lo = lane * q;
  uint32_t s_hi = s_lo + q;
  if (s_hi > nseg) s_hi = nseg;
  const g1_xyzz* sp = seg_pts + 2 * (size_t)b * nseg;
  fl S = quad_zero(), T = quad_zero(), A = quad_zero();
#pragma unroll 1
  for (uint32_t s = s_hi; s > s_lo;) {
    s--;
    const fl ls = quad_load(&sp[2 * s]), lt = quad_load(&sp[2 * s + 1]);
    T = quad_add_call(T, S);  // every segment above s gains one more unit of weight
    S = quad_add_call(S, ls);
    A = quad_add_call(A, lt);
  }
  // S = sum S_s, T = sum (s - s_lo) S_s, A = sum T_s over the quad's segments; inclusive suffix sums of S over the quads
  fl suf = S;
  uint32_t buf = 0;
#pragma unroll 1
  for (uint32_t d = 1; d < 64; d <<= 1) {
    ex[buf][lane][c] = suf;
    __syncthreads();
    if (lane + d < 64) suf = quad_add_call(suf, ex[buf][lane + d][c]);
    buf ^= 1;
  }
  const fl all = suf;  // quad 0: the sum of every segment
  if (lane == 0) suf = quad_zero();
  // per quad: A + seg_len * (T + q * suf); their sum over the workgroup is the result
  fl r = quad_add_call(T, quad_mul_small(suf, q));
  r = quad_add_call(quad_mul_small(r, seg_len), A);
#pragma unroll 1
  for (uint32_t d = 32; d >= 1; d >>= 1) {
    ex[buf][lane][c] = r;
    __syncthreads();
    if (lane < d) r = quad_add_call(r, ex[buf][lane + d][c]);
    buf ^= 1;
  }
  if (lane == 0) {
    if (out_pair) {
      quad_store(&out_pair[2 * (size_t)b], all);
      quad_store(&out_pair[2 * (size_t)b + 1], r);
    } else if (out_part) {
      quad_store(&out_part[b], r);
    } else {
      const g1x p = QD::gather(r);
      if (c == 0) out[b] = G1S::to_jac_ext(p);
    }
  }
}
bool quad_reduce_final() {  // CAPGPU_MSM_QUAD_FINAL=0: the one-lane msm_reduce_final
  static const bool on = [] {
    const char* e = getenv("CAPGPU_MSM_QUAD_FINAL");
    return !e || atoi(e) != 0;
  }();
  return on;
}
// launches the quad tails take: up to this many MSMs (CAPGPU_MSM_QUAD_MAX; 0: never)
uint32_t quad_max_batch() {
  static const uint32_t v = [] {
    const char* e = getenv("CAPGPU_MSM_QUAD_MAX");
    const int x = e ? atoi(e) : 23;  // (every launch of the narrow table: the wide one takes over at 24 MSMs)
    return (uint32_t)(x < 0 ? 0 : x);
  }();
  return v;
}

// ... and of the wide table (24 MSMs of 16384 buckets and up; CAPGPU_MSM_QUAD_MAX_WIDE, 0: never).  Beyond 63 MSMs the
// alternative is the running-sum reduction, whose 64-bucket segments fill the chip only from about 250 MSMs on: launches
// of 65 - 100 MSMs (batches of 13 - 20 proofs) are 7 - 9 % faster on the quad grid, from about 130 on there is no difference.
uint32_t quad_max_batch_wide() {
  static const uint32_t v = [] {
    const char* e = getenv("CAPGPU_MSM_QUAD_MAX_WIDE");
    const int x = e ? atoi(e) : 130;  // (measured: round-4 same-box A/B `quadwide2`, profiles/LOG.md)
    return (uint32_t)(x < 0 ? 0 : x);
  }();
  return v;
}
// bucket sets the grid form takes: 2^k buckets, 2 <= k <= 14, both grid dimensions <= 128 (CAPGPU_MSM_GRID_REDUCE=0: off)
bool use_grid_reduce(uint32_t half) {
  static const bool on = [] {
    const char* e = getenv("CAPGPU_MSM_GRID_REDUCE");
    return !e || atoi(e) != 0;
  }();
  return on && half >= 4 && half <= 16384 && (half & (half - 1)) == 0;
}

// out[b] = sum of the `parts` sub-MSM results of MSM b (one wavefront per MSM)
__global__ __launch_bounds__(64) void msm_sum_parts(const g1_xyzz* __restrict__ part_pts, uint32_t parts,
                                                    g1_jac* __restrict__ out) {
  const uint32_t b = blockIdx.x, lane = threadIdx.x;
  g1x acc = G1S::inf();
  for (uint32_t k = lane; k < parts; k += 64) acc = G1S::add(acc, G1S::load(part_pts[(size_t)b * parts + k]));
  acc = wave_sum<G1S>(acc);
  if (lane == 0) out[b] = G1S::to_jac_ext(acc);
}
// an MSM over no points
__global__ void msm_fill_inf(g1_jac* __restrict__ out, uint32_t batch) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < batch) out[b] = G1L::to_jac_ext(G1L::inf());
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct WsLayout {
  size_t counts, offsets, sorted, buckets, partial, item_off, item_base, totals, item_bucket, item_sub, item_pts, max_items,
      table, off2, tloc, chunks, nblk, seg_tot, part_pts, total;
};

// which table / sort a launch uses
// window bits of the small-launch table (CAPGPU_MSM_SMALL_C: 9 .. 12, 0 = no such table) and the largest launch, in MSMs,
// that takes it (CAPGPU_MSM_SMALL_MAX)
// OFF by default: measured in round 4 (tools/gpu_latency_ab.py, profiles/latency_ab_r04.jsonl) the c = 11 table makes a
// single proof SLOWER, 3.33 -> 4.15 ms, and a lone 2^17-point MSM 0.63 -> 0.82 ms.  The chain through the bucket reduction
// does shrink (msm_reduce_grid 93 -> 63 us per launch), but a bucket of 768 entries is cut into 64 work items instead of
// 16 and their per-bucket combine tree costs three times what the reduction saves (msm_combine_wave 60 -> 155-215 us per
// launch; the 24-entry runs of msm_scatter_runs and the item lists of msm_sort_items grow likewise).  The knob stays for
// experiments.
uint32_t small_c() {
  static const uint32_t v = [] {
    const char* e = getenv("CAPGPU_MSM_SMALL_C");
    const int x = e ? atoi(e) : 0;
    return (uint32_t)(x >= 9 && x <= 12 ? x : 0);
  }();
  return v;
}
uint32_t small_max_batch() {
  static const uint32_t v = [] {
    const char* e = getenv("CAPGPU_MSM_SMALL_MAX");
    const int x = e ? atoi(e) : 8;
    return (uint32_t)(x >= 1 && x <= 64 ? x : 8);
  }();
  return v;
}
// Smallest launch, in (sub-)MSMs, that takes the wide table (CAPGPU_MSM_WIDE_MIN).  32 until round 4, when a launch of
// 32 .. 63 MSMs reduced its 16384 buckets through bit planes (0.9 ms for 40 MSMs); with the grid form (0.65 ms) and one
// combine thread per single-item bucket the wide table pays from about two dozen MSMs: one-context rates at batch 6 (30
// MSMs per wire launch) 573 -> 627 proofs/s, batch 4 (20) 521 / 515, batch 2 (10) 358 -> 315 (round-4 same-box A/B `widemin`, profiles/LOG.md).
uint32_t wide_min_batch() {
  static const uint32_t v = [] {
    const char* e = getenv("CAPGPU_MSM_WIDE_MIN");
    const int x = e ? atoi(e) : 24;
    return (uint32_t)(x >= 1 && x <= 1024 ? x : 24);
  }();
  return v;
}
constexpr uint32_t kWideCDefault = 15;  // window bits of the large-batch table
uint32_t wide_c() {
  static uint32_t v = [] {
    const char* e = getenv("CAPGPU_MSM_WIDE_C");
    int x = e ? atoi(e) : (int)kWideCDefault;
    return (uint32_t)(x >= 14 && x <= 16 ? x : (int)kWideCDefault);
  }();
  return v;
}
// Buckets per level-1 bin = 2^sub_bits: 128, or 64 when a bin of 128 would outgrow msm_sort_level2's LDS stage (a
// sub-MSM of 2^16 points has 8704 entries per 128-bucket bin; the overflow is placed entry by entry straight to HBM).
uint32_t pick_sub_bits(size_t n_sub, uint32_t c, uint32_t windows) {
  static const int forced = [] {
    const char* e = getenv("CAPGPU_MSM_SUB_BITS");  // experiments / tests: 5..7
    const int x = e ? atoi(e) : 0;
    return x >= 5 && x <= 7 ? x : 0;
  }();
  if (forced) return (uint32_t)forced;
  const size_t bins7 = ((size_t)1 << (c - 1)) >> 7;
  return n_sub * windows / bins7 > kL2Stage ? 6u : 7u;  // by the mean: 4352 at n = 2^15 stays at 128 (measured better)
}
// A launch = `batch` MSMs of n points, each cut into `parts` sub-MSMs over consecutive ranges of n_sub points (the last
// one shorter).  Every kernel below sees batch * parts independent MSMs of at most n_sub points; msm_sum_parts adds
// the parts.  This is how one long MSM (2^19 .. 2^24 points) gets the chip-filling batched path: 2^24 points are 256
// sub-MSMs of 2^16, whose bucket reductions cost 4 % of the bucket accumulation.
struct Plan {
  uint32_t c, windows, sub_bits;
  const g1_affine* ext;
  uint32_t parts;
  size_t n_sub;
  // deep plan (one long MSM on one bucket set of 2^(c-1) buckets, three-level sort): key bits = top + mid + low,
  // tiles are gathered in groups of G
  bool deep = false;
  uint32_t top = 0, mid = 0, low = 0, G = 0, groups = 0;
};
// The third table's window size for an SRS of n points.  Not every c is usable: the top window holds only
// 254 - c (W - 1) bits, and when those are few, all n of its digits fall into a handful of buckets (c = 18: 2 bits - four
// buckets of n / 4 entries each; c = 19: 7 bits; c = 21: 2; c = 22: 12).  c = 17 (16 of 17 bits, W = 15) and c = 20 (14 of
// 20, W = 13) have nearly full top windows, like the c = 15 table (14 of 15).
// Scalars are < 2^254 (canonical; msm_digits_local folds larger inputs by r first), so the top window of the deep table
// holds top_bits = 254 - c (W - 1) of its c bits and its digit is at most 2^top_bits (with the carry from below).
// Shifted left by (c - 1) - top_bits it still fits the 2^(c-1) buckets and lands on every 2^shift-th of them.
// First measured in round 4 at 2^24 points (round-4 same-box A/B `deepshift`, profiles/LOG.md, profiles/deepshift_r04.jsonl): the shift does what
// it was meant to - msm_deep_sort<1> 2.21 -> 1.08 ms at c = 22, the crowded bins gone - and c = 22 does shorten the
// accumulation (12 digits instead of 13), but the 4096 buckets that hold the top window's n / 4096 entries each are now one
// in every 512 instead of 4096 in a row, and msm_combine - one thread per bucket, serial over the bucket's ~66 work items -
// ran one long lane per wave: 0.2 -> 5.7 ms.  With those buckets on a list of their own, 32 lanes each (msm_combine_heavy:
// 0.05 ms), c = 22 with the shift is the default from 2^23 points on: 2^24 points 18.17 -> 17.25 ms, 2^23 9.75 -> 9.43 ms,
// 2^22 5.34 -> 5.29 ms (round-4 same-box A/B `deepwide`, profiles/LOG.md; the 2^21 buckets cost +0.5 ms in sorts and reductions).  c = 20 with the
// shift: 18.87 against 18.33 ms - the shift is for the 22-bit windows only.
bool deep_wide_default() {  // CAPGPU_MSM_DEEP_WIDE=0: c = 20, unshifted, for every table (the plan until the end of round 4)
  static const bool on = [] {
    const char* e = getenv("CAPGPU_MSM_DEEP_WIDE");
    return !e || atoi(e) != 0;
  }();
  return on;
}
uint32_t deep_top_shift(uint32_t c, uint32_t windows) {
  static const int env = [] {
    const char* e = getenv("CAPGPU_MSM_DEEP_SHIFT");
    return e ? (atoi(e) != 0 ? 1 : 0) : -1;
  }();
  const bool on = env >= 0 ? env == 1 : (c == 22 && deep_wide_default());  // default: with the 22-bit windows only
  const int top_bits = 254 - (int)c * ((int)windows - 1);
  if (!on || top_bits <= 0 || top_bits >= (int)c - 1) return 0;
  return (uint32_t)((int)c - 1 - top_bits);
}
uint32_t deep_c(size_t n) {
  if (const char* e = getenv("CAPGPU_MSM_DEEP_C")) {
    const int x = atoi(e);
    if (x >= 17 && x <= 22) return (uint32_t)x;  // (c = 16 has 17 windows: one more than the entry's window field holds)
  }
  // c = 22 with the shifted top window for tables of 2^23 points and more (CAPGPU_MSM_DEEP_C / _DEEP_SHIFT override both):
  // 12 digits per scalar instead of 13; below that the sorts and reductions of 2^21 buckets cost what the shorter
  // accumulation saves (see deep_top_shift)
  if (n >= ((size_t)1 << 23) && deep_wide_default()) return 22u;
  return 20u;  // (c = 17 measured at 2^19 .. 2^21 points: 1.57 / 2.20 / 3.75 ms against 1.47 / 2.07 / 3.26 ms for c = 20)
}
bool deep_enabled() {
  const char* e = getenv("CAPGPU_MSM_DEEP");
  return !e || atoi(e) != 0;
}
size_t deep_min_points() {  // shortest MSM that takes the deep plan
  const char* e = getenv("CAPGPU_MSM_DEEP_MIN");
  // measured (tools/gpu_msm_deep_ab.py, c = 20): 2^19 points 1.47 ms deep against 1.85 ms as 64 sub-MSMs, 2^20 2.07
  // against 2.56, 2^21 3.26 against 4.03
  const long long x = e ? atoll(e) : (1ll << 19);
  return (size_t)(x >= 4096 ? x : 4096);
}
size_t deep_min_density() {  // average entries per bucket below which a launch leaves the deep plan (tests: 0)
  const char* e = getenv("CAPGPU_MSM_DEEP_DENSITY");
  const long long x = e ? atoll(e) : 8;
  return (size_t)(x >= 0 ? x : 8);
}
// fills the deep fields for an MSM of n points on the table (c, windows); false when the shape does not fit the sort
bool deep_shape(Plan& pl, size_t n) {
  const uint32_t K = pl.c - 1;
  if (K < 14 || pl.windows > 16) return false;
  // buckets per level-3 bin = 2^low: as many as keep an average bin inside the LDS stage
  const size_t avg_bucket = std::max<size_t>(((size_t)pl.windows * n) >> K, 1);
  pl.low = 7;
  while (pl.low > 4 && (avg_bucket << pl.low) > 14000) pl.low--;
  const uint32_t rest = K - pl.low;
  pl.top = (rest + 1) / 2;
  pl.mid = rest - pl.top;
  if (pl.top > 7 || pl.mid > 7 || pl.mid + pl.low > 14) return false;
  const size_t nblk = (n + kDigitTile - 1) / kDigitTile;
  const size_t run1 = std::max<size_t>(((size_t)kDigitTile * pl.windows) >> pl.top, 1);
  uint32_t G = 1;
  while (G < 128 && (size_t)(2 * G) * run1 <= 14336) G *= 2;  // a group's gathered runs must fit the 16384-entry stage
  pl.G = G;
  pl.groups = (uint32_t)((nblk + G - 1) / G);
  if (pl.groups > kDeepMaxRuns) return false;
  pl.sub_bits = K - pl.top;
  pl.deep = true;
  pl.parts = 1;
  pl.n_sub = n;
  return true;
}
size_t split_target() {  // sub-MSMs a long MSM is cut into at least (when it has the points for it)
  static const size_t v = [] {
    const char* e = getenv("CAPGPU_MSM_SPLIT");
    int x = e ? atoi(e) : 128;  // measured: 2^20 points 3.3 -> 3.0 ms, 2^22 8.7 -> 8.1 ms against 64
    return (size_t)(x >= 1 && x <= 4096 ? x : 128);
  }();
  return v;
}
Plan choose_plan(const MsmBases& bases, size_t n, uint32_t batch) {
  Plan pl{bases.c, bases.windows, 0, bases.ext, 1, n};
  // one long MSM: the deep plan, as long as its buckets hold >= 8 entries on average (a short range of a long table
  // would pay for 2^(c-1) bucket reductions it has no entries for)
  if (bases.ext3 && batch == 1 && deep_enabled() && n >= deep_min_points() &&
      n * bases.windows3 >= (deep_min_density() << (bases.c3 - 1))) {
    Plan dp{bases.c3, bases.windows3, 0, bases.ext3, 1, n};
    if (deep_shape(dp, n)) return dp;
  }
  const bool primary_wide = bases.c >= 14;  // tables of more than 2^18 points hold the wide windows only
  if (primary_wide) {
    pl.sub_bits = pick_sub_bits(n, pl.c, pl.windows);
    // enough sub-MSMs to fill the chip (the running-sum reduction wants >= 64; 128 measured best), none shorter than 8192 points (below
    // that the 16384-bucket reduction of a part costs more than a third of its accumulation) nor longer than the
    // level-2 sort's run table
    size_t want = (n * batch + split_target() - 1) / split_target();
    want = std::max<size_t>(want, 8192);
    want = std::min<size_t>(align_up(want, kDigitTile), (size_t)64 * kDigitTile);
    if (n > want) {
      pl.n_sub = want;
      pl.parts = (uint32_t)((n + want - 1) / want);
      pl.sub_bits = pick_sub_bits(want, pl.c, pl.windows);
    }
    return pl;
  }
  // wide windows pay off once (a) the batch alone gives every CU work, so that the 4x larger bucket set costs two
  // running-sum additions per bucket instead of a log-depth tree, and (b) buckets still hold several entries
  if (bases.ext2 && n >= 4096) {
    uint32_t parts = n > kMaxSubPoints ? (uint32_t)((n + 65535) / 65536) : 1;
    const size_t n_sub0 = parts > 1 ? (size_t)65536 : n;
    // (sub-)MSMs of >= 2^16 points put twice the entries into each of the 16384 buckets, so the wide table's bucket
    // reduction weighs half as much: it pays from 4 of them, not two dozen (round 6, same box, ms per call at 2^17 points:
    // 2 MSMs 0.76 -> 0.65, 5 MSMs 1.40 -> 1.21, 8 MSMs 1.99 -> 1.76; at 2^16: 5 MSMs 0.80 -> 0.77, 8 MSMs 1.10 -> 1.01; a
    // lone 2^17-point MSM stays on the narrow table: 0.48 against 0.53)
    const size_t wide_from = n_sub0 >= 65536 ? std::min<size_t>(wide_min_batch(), 4) : wide_min_batch();
    if ((size_t)batch * parts >= wide_from) {
      const size_t n_sub = n_sub0;
      pl = Plan{bases.c2, bases.windows2, pick_sub_bits(n_sub, bases.c2, bases.windows2), bases.ext2, parts, n_sub};
      return pl;
    }
  }
  // a handful of MSMs: the small-launch table (fewer buckets: a shorter chain through the bucket reduction)
  if (bases.ext0 && batch <= small_max_batch() && n >= 1024) {
    pl = Plan{bases.c0, bases.windows0, 0, bases.ext0, 1, n};
    return pl;
  }
  return pl;  // one-level sort on the narrow table: n <= 2^18 here, whole MSMs
}
// buckets per msm_reduce_segments thread
uint32_t reduce_seg_len(uint32_t half) {
  if (half >= 16384) return 64;  // wide windows: 256 segments per entry keep msm_reduce_final at 4 segments per lane
  return half >= 1024 ? 16u : (half >= 64 ? half / 64 : 1u);
}
// ... sized to the launch (round-4 experiment; see the default below).  Every msm_reduce_segments thread
// does the same work - 2 * seg_len dependent additions - and the kernel holds two waves per SIMD (226 VGPRs): 131072
// threads at a time; a launch of 1280 MSMs with 256 segments each is 2.5 such "rounds".  Choosing the segment length that
// minimises rounds * 2 * seg_len plus the one-wave finish (1280 MSMs: 82 buckets, 200 segments; 256 MSMs: 32) was
// measured and is no gain: msm_reduce_segments 9.33 -> 9.45 ms per step, msm_reduce_final 1.50 -> 1.65 ms
// (round-4 same-box A/B `segtune`, profiles/LOG.md) - the chip does not run this kernel in lock-step rounds.
constexpr uint32_t kSegLenMin = 32, kSegLenMax = 128;
uint32_t reduce_seg_len_for(uint32_t half, uint32_t sb) {
  // CAPGPU_MSM_SEG_TUNE: 1 = every launch, 0 = none; default: launches of up to 512 MSMs, whose segments all fit the chip at
  // once - there the shorter chains do pay (batch 64, 320-MSM launches: 52.1 -> 50.9 ms, round-4 same-box A/B `segtune2`, profiles/LOG.md; batches of
  // 27 - 51: no difference either way)
  static const int tune = [] {
    const char* e = getenv("CAPGPU_MSM_SEG_TUNE");
    return e ? (atoi(e) != 0 ? 1 : 0) : -1;
  }();
  const uint32_t base = reduce_seg_len(half);
  if (tune == 0 || (tune < 0 && sb > 512) || half < 16384 || sb == 0) return base;
  const uint64_t slots = 2ull * 1024 * 64;  // threads of msm_reduce_segments the chip holds at once
  uint32_t best = base;
  uint64_t best_cost = ~0ull;
  for (uint32_t sl = kSegLenMin; sl <= kSegLenMax; sl += 2) {
    const uint64_t nseg = (half + sl - 1) / sl;
    const uint64_t rounds = (nseg * sb + slots - 1) / slots;
    const uint64_t q = (nseg + 63) / 64;
    const uint64_t final_rounds = ((uint64_t)sb + 1023) / 1024;  // msm_reduce_final: one wave per entry, one per SIMD
    const uint64_t cost = rounds * 2 * sl + final_rounds * (3 * q + 25);
    if (cost < best_cost) {
      best_cost = cost;
      best = sl;
    }
  }
  return best;
}
// the running-sum reduction wants >= 16 Ki independent segments (64 waves per XCD); below that the log-depth path
bool use_segment_reduce(uint32_t half, uint32_t batch) {
  uint32_t seg_len = reduce_seg_len(half);
  return (size_t)((half + seg_len - 1) / seg_len) * batch >= 16384;
}
// about 8 waves per SIMD worth of work items (256 CUs x 4 SIMDs x 64 lanes x 8) before items grow beyond the minimum.
// (Shorter items for a single proof's 1- and 2-MSM launches - down to 8 entries below 2^20 entries - were measured:
// msm_accumulate 0.84 -> 0.78 ms per proof, msm_combine_wave 0.29 -> 0.45 ms.)
uint32_t choose_item_len(size_t entries, size_t buckets) {
  static const size_t cap = [] {
    const char* e = getenv("CAPGPU_MSM_ITEM_MAX");
    int x = e ? atoi(e) : (int)kBatchItemCap;
    return (size_t)(x >= 8 && x <= (int)kMaxItemLen ? x : (int)kBatchItemCap);
  }();
  static const size_t floor_len = [] {
    const char* e = getenv("CAPGPU_MSM_ITEM_MIN");
    int x = e ? atoi(e) : (int)kMinItemLen;
    return (size_t)(x >= 4 && x <= (int)kBatchItemCap ? x : (int)kMinItemLen);
  }();
  // Small launches (a single proof's MSMs, single MSMs up to 2^18 points): one wave per SIMD already saturates it
  // (tools/ubench_lonewave.hip), and a SIMD that gets two waves takes twice as long - so the launch should be ONE wave
  // of items per SIMD at most (65536 items), every item as short as that allows.  With `buckets` buckets of about
  // `entries / buckets` entries each that is m = 65536 / buckets items per bucket, and the item length is chosen so that
  // a bucket 4 sigma above the average still needs no more than m.
  static const bool tune_small = [] {
    const char* e = getenv("CAPGPU_MSM_ITEM_SMALL");
    return !e || atoi(e) != 0;
  }();
  static const size_t small_waves = [] {  // experiment: waves of items per SIMD the small launches aim at
    const char* e = getenv("CAPGPU_MSM_ITEM_WAVES");
    const int x = e ? atoi(e) : 1;
    return (size_t)(x >= 1 && x <= 8 ? x : 1);
  }();
  static const size_t small_buckets = [] {  // experiment: the largest bucket set (all MSMs of the launch) the rule takes
    const char* e = getenv("CAPGPU_MSM_ITEM_SMALL_BUCKETS");
    const int x = e ? atoi(e) : 8192;
    return (size_t)(x >= 0 ? x : 8192);
  }();
  if (tune_small && buckets && buckets <= small_buckets && entries >= buckets) {  // (1 or 2 MSMs; with 5 the items get long
                                                                   // and a lone wave cannot hide its gathers: measured worse)
    const double avg = (double)entries / (double)buckets;
    const double hi = avg + 4.0 * sqrt(avg);
    const size_t m = std::max<size_t>(1, 65536 * small_waves / buckets);
    const size_t l = (size_t)ceil(hi / (double)m);
    return (uint32_t)std::min<size_t>(std::max<size_t>(l, 8), cap);
  }
  size_t l = entries / ((size_t)1 << 19);
  return (uint32_t)std::min<size_t>(std::max<size_t>(l, std::min<size_t>(floor_len, cap)), cap);
}
// the chained forms of the small kernels in front of the accumulation (CAPGPU_MSM_CHAINED=0: the separate launches)
bool chained_enabled() {
  static const bool on = [] {
    const char* e = getenv("CAPGPU_MSM_CHAINED");
    return !e || atoi(e) != 0;
  }();
  return on;
}
// sb = sub-MSMs of the launch (batch * parts), n = points per sub-MSM
WsLayout ws_layout(uint32_t c, uint32_t windows, size_t n, uint32_t sb, uint32_t sub_bits, bool has_parts) {
  WsLayout L{};
  const size_t bins = ((size_t)1 << (c - 1)) >> sub_bits;  // sort keys of the tile-local level
  size_t half = (size_t)1 << (c - 1);
  size_t per = (size_t)windows * n;
  size_t seg_len = std::min<size_t>(reduce_seg_len((uint32_t)half), half >= 16384 ? kSegLenMin : ~(size_t)0);
  size_t nseg = (half + seg_len - 1) / seg_len;  // (the most segments reduce_seg_len_for may choose: room in `partial`)
  size_t o = 0;
  L.counts = o;  o = align_up(o + sizeof(uint32_t) * half * sb, 256);
  L.offsets = o; o = align_up(o + sizeof(uint32_t) * half * sb, 256);
  L.sorted = o;  o = align_up(o + sizeof(uint32_t) * per * sb, 256);
  {
    size_t chunks = kReduceMaxChunks;  // room for any choice of reduce_chunks
    size_t npart = use_segment_reduce((uint32_t)half, sb) ? 2 * nseg : c * chunks;
    {  // the grid reduction's row and column sums (msm_reduce_grid): 2^ceil(k/2) + 2^floor(k/2) per entry
      uint32_t k = 0;
      while (((size_t)1 << k) < half) k++;
      npart = std::max(npart, ((size_t)half >> (k / 2)) + ((size_t)1 << (k / 2)) + 3);  // (+ 3: msm_reduce_grid_final_quad)
    }
    L.buckets = o; o = align_up(o + sizeof(g1_xyzz) * half * sb, 256);
    L.partial = o; o = align_up(o + sizeof(g1_xyzz) * npart * sb, 256);
  }
  L.max_items = per * sb / choose_item_len(per * sb, half * sb) + half * sb;  // sum ceil(cnt/L) <= entries/L + buckets
  L.item_off = o;    o = align_up(o + sizeof(uint32_t) * half * sb, 256);
  L.item_base = o;   o = align_up(o + sizeof(uint32_t) * ((size_t)sb + 3), 256);
  L.totals = o;      o = align_up(o + sizeof(uint32_t) * ((size_t)sb + 1), 256);  // (+ 1: msm_items' ticket counter)
  L.item_bucket = o; o = align_up(o + sizeof(uint32_t) * L.max_items, 256);
  L.item_sub = o;    o = align_up(o + sizeof(uint32_t) * L.max_items, 256);
  L.item_pts = o;    o = align_up(o + sizeof(g1_xyzz) * L.max_items, 256);
  L.nblk = (n + kDigitTile - 1) / kDigitTile;
  L.table = o;       o = align_up(o + sizeof(uint32_t) * bins * L.nblk * sb, 256);
  L.off2 = o;        o = align_up(o + sizeof(uint32_t) * bins * L.nblk * sb, 256);
  L.tloc = o;        o = align_up(o + sizeof(uint32_t) * bins * L.nblk * sb, 256);
  L.chunks = o;      o = align_up(o + sizeof(uint32_t) * (size_t)kDigitTile * windows * L.nblk * sb, 256);
  {
    size_t rows = std::max<size_t>(bins * L.nblk, half);
    L.seg_tot = o;   o = align_up(o + sizeof(uint32_t) * (((rows + kScanSeg - 1) / kScanSeg) * sb + 1), 256);  // (+ 1: ticket)
  }
  L.part_pts = o;    o = align_up(o + sizeof(g1_xyzz) * (has_parts ? sb : 0), 256);
  L.total = o;
  return L;
}
// workspace of the deep plan (one MSM of n points)
struct DeepLayout {
  size_t counts, offsets, chunks /* = sorted, once level 2 has read the chunks */, mid_buf, table, off2, tloc, table2, off3,
      tloc2, seg_tot, buckets, partial, pairs, item_off, item_base, totals, item_bucket, item_sub, item_pts, max_items, nblk,
      entries, total;
  uint32_t sbp, item_len;
};
DeepLayout deep_layout(const Plan& pl, size_t n) {
  DeepLayout L{};
  const uint32_t K = pl.c - 1;
  const size_t nb = (size_t)1 << K;
  L.sbp = (uint32_t)(nb / kDeepEntryBuckets);
  L.nblk = (n + kDigitTile - 1) / kDigitTile;
  L.entries = (size_t)kDigitTile * pl.windows * L.nblk;  // capacity of the tile chunks >= W n
  L.item_len = deep_item_len(std::max<size_t>(((size_t)pl.windows * n) >> K, 1));
  L.max_items = (size_t)pl.windows * n / L.item_len + nb;
  const size_t rows1 = ((size_t)1 << pl.top) * L.nblk, rows2 = ((size_t)1 << (pl.top + pl.mid)) * pl.groups;
  size_t o = 0;
  auto take = [&](size_t bytes) {
    const size_t at = o;
    o = align_up(o + bytes, 256);
    return at;
  };
  L.counts = take(4 * nb);
  L.offsets = take(4 * nb);
  L.chunks = take(4 * L.entries);
  L.mid_buf = take(4 * L.entries);
  L.table = take(4 * rows1);
  L.off2 = take(4 * rows1);
  L.tloc = take(4 * rows1);
  L.table2 = take(4 * rows2);
  L.off3 = take(4 * rows2);
  L.tloc2 = take(4 * rows2);
  L.seg_tot = take(4 * ((std::max(rows1, rows2) + kScanSeg - 1) / kScanSeg + 1));
  L.buckets = take(sizeof(g1_xyzz) * nb);
  L.partial = take(sizeof(g1_xyzz) * (size_t)L.sbp * 2 * (kDeepEntryBuckets / kDeepSegLen));  // (S, T) per segment
  L.pairs = take(sizeof(g1_xyzz) * 2 * (nb / kDeepReduceBuckets));
  L.item_off = take(4 * nb);
  L.item_base = take(4 * ((size_t)L.sbp + 3));
  L.totals = take(4 * ((size_t)L.sbp + 1));
  L.item_bucket = take(4 * L.max_items);
  L.item_sub = take(4 * L.max_items);
  L.item_pts = take(sizeof(g1_xyzz) * L.max_items);
  L.total = o;
  return L;
}
// MSMs of a launch that go through the kernels together: the [key][tile] tables must stay below 2^28 rows and every
// per-launch counter within 32 bits; larger batches are run in slices
uint32_t batch_slice(const Plan& pl, uint32_t batch) {
  if (pl.deep) return 1;
  const size_t bins = ((size_t)1 << (pl.c - 1)) >> pl.sub_bits;
  const size_t nblk = (pl.n_sub + kDigitTile - 1) / kDigitTile;
  const size_t per_msm_rows = bins * std::max<size_t>(nblk, 1) * pl.parts;
  const size_t per_msm_entries = (size_t)pl.windows * pl.n_sub * pl.parts + ((size_t)1 << (pl.c - 1)) * pl.parts;
  size_t lim = std::min<size_t>(((size_t)1 << 28) / per_msm_rows, ((size_t)3 << 30) / per_msm_entries);
  lim = std::min<size_t>(lim, 65535 / pl.parts);  // grid y of msm_sort_level2
  if (const char* e = getenv("CAPGPU_MSM_SLICE")) {  // test hook: force slicing at small sizes
    const int x = atoi(e);
    if (x >= 1) lim = std::min<size_t>(lim, (size_t)x);
  }
  return (uint32_t)std::max<size_t>(1, std::min<size_t>(lim, batch));
}

}  // namespace

uint32_t msm_choose_window(size_t n) {
  const char* env = getenv("CAPGPU_MSM_C");
  if (env) {
    int v = atoi(env);
    if (v >= 2 && v <= 16) return (uint32_t)v;
  }
  if (n <= ((size_t)1 << 10)) return 9;
  if (n <= ((size_t)1 << 18)) return 13;
  return wide_c();  // long MSMs are run as batches of sub-MSMs on the wide windows (choose_plan)
}

uint32_t msm_num_windows(uint32_t c) {
  uint32_t w = (256 + c - 1) / c;
  if (256 % c == 0) w += 1;  // room for the final signed-digit carry
  return w;
}

int msm_precompute(MsmBases* out, const g1_affine* d_bases, size_t n, uint32_t c, hipStream_t stream) {
  out->n = n;
  out->c = c;
  out->windows = msm_num_windows(c);
  if ((size_t)out->windows * n >= ((size_t)1 << 31) || out->windows > 31) return (int)hipErrorInvalidValue;
  hipError_t e = hipMalloc(&out->ext, sizeof(g1_affine) * (n ? n : 1) * out->windows);
  if (e != hipSuccess) return (int)e;
  if (n == 0) return 0;
  size_t blocks = (n + kThreads - 1) / kThreads;
  launch("msm_precompute_kernel", msm_precompute_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, stream, out->ext, d_bases, n, c,
                     out->windows, 0u);
  // the wide-window table for large batches
  const char* env = getenv("CAPGPU_MSM_WIDE");
  const uint32_t kWideC = wide_c();
  const uint32_t w2 = msm_num_windows(kWideC);
  if (!getenv("CAPGPU_MSM_C") && !(env && atoi(env) == 0) && c < kWideC && n >= 4096) {
    e = hipMalloc(&out->ext2, sizeof(g1_affine) * n * w2);
    if (e != hipSuccess) {
      (void)hipStreamSynchronize(stream);
      msm_free_bases(out);  // do not leave the first table behind on the out-of-memory path
      return (int)e;
    }
    out->c2 = kWideC;
    out->windows2 = w2;
    launch("msm_precompute_kernel", msm_precompute_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, stream, out->ext2,
           d_bases, n, kWideC, w2, 0u);
  }
  // the small-launch table (the primary table is the c = 13 one: 1024 < n <= 2^18)
  if (!getenv("CAPGPU_MSM_C") && small_c() && c == 13 && n >= 1024) {
    const uint32_t c0 = small_c(), w0 = msm_num_windows(c0);
    if (w0 <= 31 && (size_t)w0 * n < ((size_t)1 << 31)) {
      e = hipMalloc(&out->ext0, sizeof(g1_affine) * n * w0);
      if (e != hipSuccess) {
        (void)hipStreamSynchronize(stream);
        msm_free_bases(out);
        return (int)e;
      }
      out->c0 = c0;
      out->windows0 = w0;
      launch("msm_precompute_kernel", msm_precompute_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, stream, out->ext0,
             d_bases, n, c0, w0, 0u);
    }
  }
  // the deep-window table for long single MSMs (tables of more than 2^18 points)
  if (!getenv("CAPGPU_MSM_C") && deep_enabled() && n > ((size_t)1 << 18) && n >= deep_min_points()) {
    const uint32_t c3 = deep_c(n), w3 = msm_num_windows(c3);
    Plan probe{c3, w3, 0, nullptr, 1, n};
    if ((size_t)w3 * n < ((size_t)1 << 31) && deep_shape(probe, n)) {
      e = hipMalloc(&out->ext3, sizeof(g1_affine) * n * w3);
      if (e != hipSuccess) {
        (void)hipStreamSynchronize(stream);
        msm_free_bases(out);
        return (int)e;
      }
      out->c3 = c3;
      out->windows3 = w3;
      out->top_shift3 = deep_top_shift(c3, w3);
      launch("msm_precompute_kernel", msm_precompute_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, stream, out->ext3,
             d_bases, n, c3, w3, out->top_shift3);
    }
  }
  return 0;  // launch failures are latched by launch() and reported by take_launch_error()
}

void msm_free_bases(MsmBases* b) {
  if (b->ext) hipFree(b->ext);
  if (b->ext2) hipFree(b->ext2);
  if (b->ext3) hipFree(b->ext3);
  if (b->ext0) hipFree(b->ext0);
  b->ext = b->ext2 = b->ext3 = b->ext0 = nullptr;
  b->n = 0;
}

size_t msm_workspace_bytes(const MsmBases& bases, size_t n, uint32_t batch) {
  if (n == 0 || batch == 0) return 256;
  Plan pl = choose_plan(bases, n, batch);
  if (pl.deep) return deep_layout(pl, n).total;
  const uint32_t slice = batch_slice(pl, batch);
  return ws_layout(pl.c, pl.windows, pl.n_sub, slice * pl.parts, pl.sub_bits, pl.parts > 1).total;
}

const char* msm_plan_describe(const MsmBases& bases, size_t n, uint32_t batch, char* buf, size_t cap) {
  Plan pl = choose_plan(bases, n, batch);
  if (pl.deep) {
    snprintf(buf, cap,
             "c=%u windows=%u sort=deep parts=1 n_sub=%zu slice=1 top=%u mid=%u low=%u group_tiles=%u groups=%u entries=%u",
             pl.c, pl.windows, n, pl.top, pl.mid, pl.low, pl.G, pl.groups, (1u << (pl.c - 1)) / kDeepEntryBuckets);
    return buf;
  }
  int len = snprintf(buf, cap, "c=%u windows=%u sort=%s parts=%u n_sub=%zu slice=%u", pl.c, pl.windows,
                     pl.sub_bits ? "two-level" : "one-level", pl.parts, pl.n_sub, batch_slice(pl, batch));
  if (pl.sub_bits && len > 0 && (size_t)len < cap) snprintf(buf + len, cap - len, " bin_buckets=%u", 1u << pl.sub_bits);
  return buf;
}

namespace {
// Dynamic LDS msm_accumulate is launched with - not to use it, but to set how many of its workgroups a CU holds
// (experiments: CAPGPU_ACC_LDS).
size_t accumulate_lds_bytes() {
  const char* e = getenv("CAPGPU_ACC_LDS");
  const long x = e ? atol(e) : 0;
  return (size_t)(x >= 0 && x <= 65536 ? x : 0);
}
// msm_accumulate as a PERSISTENT launch (round 4): k workgroups per CU (CAPGPU_ACC_PERSISTENT, default 4; 0 = one
// workgroup per chunk of 256 items) that take chunk after chunk from an atomic counter (CAPGPU_ACC_DYNAMIC, default 1;
// 0 = a fixed stride).  The one-shot launch of a batch is ~12 000 workgroups of equal duration - their items are
// length-sorted - so the three workgroups a CU holds start together, end together, and the CU idles while the next
// three are dispatched: SQ_BUSY_CU_CYCLES said a CU was busy 0.95 of the launch (round 3: "not occupancy, not the tail").
// Measured, same box, batch 256 (round-4 same-box A/B `accpersist`, profiles/LOG.md): one-shot 114.2 ms per step; persistent + counter, k = 3 / 4 /
// 6: 109.7 / 109.3 / 109.3 ms (-4.3 %, 1303 -> 1346 proofs/s on one context); persistent with a FIXED stride: 149.7 ms -
// without the counter a workgroup cannot make up for a slower CU, and a stride of whole MSMs (64 chunks each) would
// even hand it the same position of every MSM's length-sorted list (183 ms).
unsigned accumulate_persistent() {
  static const unsigned v = [] {
    const char* e = getenv("CAPGPU_ACC_PERSISTENT");
    const long x = e ? atol(e) : 4;
    return (unsigned)(x >= 1 && x <= 64 ? x : 0);
  }();
  return v;
}
bool accumulate_dynamic() {
  static const bool v = [] {
    const char* e = getenv("CAPGPU_ACC_DYNAMIC");
    return accumulate_persistent() != 0 && (!e || atoi(e) != 0);
  }();
  return v;
}
// Workgroup size of msm_accumulate (experiments: CAPGPU_ACC_THREADS = 64, 128 or 256)
unsigned accumulate_threads() {
  static const unsigned v = [] {
    const char* e = getenv("CAPGPU_ACC_THREADS");
    const long x = e ? atol(e) : 0;
    return (unsigned)(x == 64 || x == 128 ? x : kThreads);
  }();
  return v;
}

// K5 + K6 over `sb` batch entries of `half` buckets each: work items, accumulation, bucket reduction.  One result per
// entry goes to out (Jacobian), out_part (XYZZ: parts of a longer MSM) or out_pair (XYZZ (sum, weighted sum): entries
// that are slices of one bucket set).
struct Tail {
  const g1_affine* ext;
  uint32_t *counts, *offsets, *sorted, *item_off, *item_base, *totals, *item_bucket, *item_sub;
  g1_xyzz *item_pts, *buckets, *partial;
  size_t per, max_items, entries;
  uint32_t half, sb, item_len, planes /* bit planes of the log-depth reduction: log2(half) + 1 */;
  uint32_t seg_len = 0;  // != 0: running-sum reduction with segments of this many buckets, whatever the launch size
  // != 0 (with seg_len): the reduction sees the bucket array as entries of this many buckets - smaller than `half`, which
  // sizes the length-sorted item lists - so that the one-wave finish of an entry is a short chain
  uint32_t reduce_half = 0;
  bool chained = false;  // the item kernels as one chained launch (msm_items); the digits kernel zeroed t.totals[0 .. sb]
};
void run_tail(const Tail& t, g1_jac* out, g1_xyzz* out_part, g1_xyzz* out_pair, hipStream_t stream) {
  const uint32_t half = t.half, sb = t.sb, item_len = t.item_len;
  const uint32_t seg_len = t.seg_len ? t.seg_len : reduce_seg_len_for(half, sb);
  // small and middling launches: the tails run on quads (see msm_combine_quad) - also where the running sums would
  // otherwise take over with too few segments to fill the chip (CAPGPU_MSM_QUAD_MAX_WIDE beyond 63)
  const bool quad = !out_pair && t.seg_len == 0 && use_grid_reduce(half) &&
                    sb <= (half <= 4096 ? quad_max_batch() : quad_max_batch_wide());
  const bool segments = t.seg_len != 0 || (use_segment_reduce(half, sb) && !quad);
  const uint32_t total_buckets = half * sb;
  if (t.chained) {  // (t.totals: the flag words msm_digits_local zeroed)
    launch("msm_items", msm_items, dim3(sb), dim3(1024), 0, stream, (const uint32_t*)t.counts, half, item_len, sb, t.item_off,
           t.item_base, t.totals, t.item_bucket, t.item_sub);
  } else {
    launch("msm_scan_items", msm_scan<1>, dim3(sb), dim3(1024), 0, stream, (const uint32_t*)t.counts, t.item_off, half,
           t.totals, item_len);
    launch("msm_item_bases", msm_item_bases, dim3(1), dim3(1024), 0, stream, (const uint32_t*)t.totals, sb, t.item_base);
    launch("msm_sort_items", msm_sort_items, dim3(sb), dim3(1024), 0, stream, (const uint32_t*)t.counts,
           (const uint32_t*)t.item_base, half, item_len, t.item_bucket, t.item_sub);
  }
  if (t.max_items > 0) {
    const unsigned at = accumulate_threads();
    unsigned wgs = (unsigned)((t.max_items + at - 1) / at);
    // (256 CUs; + 8: a multiple of 8 that is NOT a whole number of MSMs - an MSM is 64 chunks of length-sorted items, and a
    // stride of whole MSMs would hand a workgroup the same position, the longest or the shortest items, every time)
    bool dynamic = false;
    if (const unsigned per_cu = accumulate_persistent()) {
      if (wgs > 256u * per_cu + 8u) {  // (a launch that fits the chip at once stays one-shot: nothing to re-dispatch)
        wgs = 256u * per_cu + 8u;
        dynamic = accumulate_dynamic();
      }
    }
    launch("msm_accumulate", msm_accumulate, dim3(wgs), dim3(at),
           accumulate_lds_bytes(), stream, t.ext, (const uint32_t*)t.sorted, (const uint32_t*)t.counts, (const uint32_t*)t.offsets,
           (const uint32_t*)t.item_off, (const uint32_t*)t.item_base, (const uint32_t*)t.item_bucket,
           (const uint32_t*)t.item_sub, t.per, half, sb, item_len, dynamic ? t.item_base + sb + 1 : (uint32_t*)nullptr,
           t.item_pts, t.buckets);
  }
  // bucket = sum of its items, one thread per bucket; the few buckets of many items go to a list (msm_accumulate's item
  // list, dead by now, holds it; its length sits behind the item bases) and get 32 lanes each
  auto combine_per_bucket = [&] {
    static const bool heavy = [] {
      const char* e = getenv("CAPGPU_MSM_HEAVY_COMBINE");
      return !e || atoi(e) != 0;
    }();
    uint32_t* heavy_list = heavy && t.max_items > 0 ? t.item_bucket : nullptr;
    uint32_t* heavy_count = t.item_base + sb + 2;
    launch("msm_combine", msm_combine, dim3((total_buckets + kThreads - 1) / kThreads), dim3(kThreads), 0, stream,
           (const g1_xyzz*)t.item_pts, (const uint32_t*)t.counts, (const uint32_t*)t.item_off,
           (const uint32_t*)t.item_base, half, total_buckets, item_len, t.buckets, kHeavyItems, heavy_list, heavy_count);
    if (heavy_list)
      launch("msm_combine_heavy", msm_combine_heavy, dim3(512), dim3(kThreads), 0, stream, (const g1_xyzz*)t.item_pts,
             (const uint32_t*)t.counts, (const uint32_t*)t.item_off, (const uint32_t*)t.item_base, half, item_len,
             (const uint32_t*)heavy_list, (const uint32_t*)heavy_count, t.buckets);
  };
  if (segments) {
    combine_per_bucket();
    const uint32_t rhalf = t.reduce_half ? t.reduce_half : half, rsb = total_buckets / rhalf;
    const uint32_t rnseg = (rhalf + seg_len - 1) / seg_len;
    launch("msm_reduce_segments", msm_reduce_segments, dim3((rnseg * rsb + kThreads - 1) / kThreads), dim3(kThreads), 0,
           stream, (const g1_xyzz*)t.buckets, rhalf, seg_len, rnseg, rsb, t.partial);
    if (quad_reduce_final())
      launch("msm_reduce_final", msm_reduce_final_quad, dim3(rsb), dim3(256), 0, stream, (const g1_xyzz*)t.partial, seg_len,
             rnseg, out, out_part, out_pair);
    else
      launch("msm_reduce_final", msm_reduce_final, dim3(rsb), dim3(64), 0, stream, (const g1_xyzz*)t.partial, seg_len, rnseg,
             out, out_part, out_pair);
  } else {
    // G lanes per bucket: the smallest power of two that holds a bucket's items with some room for the spread of the
    // bucket sizes (a bucket with more items than lanes loops), so that the shuffle tree is no deeper than needed
    const size_t avg_items = (t.entries / total_buckets + item_len - 1) / item_len;
    const size_t want = avg_items * 3 / 2;
    if (avg_items <= 1 && total_buckets >= 65536) {
      // buckets of (almost always) ONE item - a few dozen MSMs on the wide table: msm_accumulate wrote those buckets
      // itself, and eight lanes per bucket would be five million idle threads (a 40-MSM launch: 300 us of them); one
      // thread per bucket walks the rare second item
      combine_per_bucket();
    } else if (quad && want <= 32) {
      // (the same 8, 16 or 32 lanes per bucket as msm_combine_wave would take: two, four or eight quads)
      const uint32_t GQ = want <= 8 ? 2u : (want <= 16 ? 4u : 8u);
      auto kern = GQ == 2 ? msm_combine_quad<2> : (GQ == 4 ? msm_combine_quad<4> : msm_combine_quad<8>);
      launch("msm_combine", kern, dim3((unsigned)(((size_t)total_buckets * GQ * 4 + kThreads - 1) / kThreads)),
             dim3(kThreads), 0, stream, (const g1_xyzz*)t.item_pts, (const uint32_t*)t.counts,
             (const uint32_t*)t.item_off, (const uint32_t*)t.item_base, half, total_buckets, item_len, t.buckets);
    } else {
      const uint32_t G = want <= 8 ? 8u : (want <= 16 ? 16u : (want <= 32 ? 32u : 64u));
      auto kern = G == 8 ? msm_combine_wave<8>
                         : (G == 16 ? msm_combine_wave<16> : (G == 32 ? msm_combine_wave<32> : msm_combine_wave<64>));
      launch("msm_combine", kern, dim3((unsigned)(((size_t)total_buckets * G + kThreads - 1) / kThreads)),
             dim3(kThreads), 0, stream, (const g1_xyzz*)t.item_pts, (const uint32_t*)t.counts,
             (const uint32_t*)t.item_off, (const uint32_t*)t.item_base, half, total_buckets, item_len, t.buckets);
    }
    if (!out_pair && use_grid_reduce(half)) {
      uint32_t k = 0;
      while ((1u << k) < half) k++;
      const uint32_t lo_bits = k / 2, nsum = (half >> lo_bits) + (1u << lo_bits);
      if (quad) {
        // 16 quads per sum where the dimension allows it (4 + 4 additions deep), 8 otherwise
        const bool s16 = std::min(half >> lo_bits, 1u << lo_bits) >= 32;
        // behind the sums in `partial` (ws_layout keeps the room): the two dimensions' results and a counter per entry
        g1_xyzz* fin = t.partial + (size_t)sb * nsum;
        uint32_t* fin_count = reinterpret_cast<uint32_t*>(fin + 2 * (size_t)sb);
        const uint32_t qs = s16 ? 16u : 8u;
        launch("msm_reduce_grid", s16 ? msm_reduce_grid_quad<16> : msm_reduce_grid_quad<8>,
               dim3((unsigned)(((size_t)sb * nsum * qs * 4 + kThreads - 1) / kThreads)), dim3(kThreads), 0, stream,
               (const g1_xyzz*)t.buckets, half, lo_bits, sb, t.partial, fin_count);
        if (half <= 4096)
          launch("msm_reduce_grid_final", msm_reduce_grid_final_quad<64>, dim3(sb, 2), dim3(256), 0, stream,
                 (const g1_xyzz*)t.partial, half, lo_bits, fin, fin_count, out, out_part);
        else
          launch("msm_reduce_grid_final", msm_reduce_grid_final_quad<128>, dim3(sb, 2), dim3(512), 0, stream,
                 (const g1_xyzz*)t.partial, half, lo_bits, fin, fin_count, out, out_part);
        return;
      }
      const uint32_t slices = grid_slices(sb, nsum, std::min(half >> lo_bits, 1u << lo_bits));
      auto gk = slices == 64 ? msm_reduce_grid<64> : (slices == 32 ? msm_reduce_grid<32> : (slices == 16 ? msm_reduce_grid<16> : msm_reduce_grid<8>));
      launch("msm_reduce_grid", gk, dim3((unsigned)(((size_t)sb * nsum * slices + kThreads - 1) / kThreads)),
             dim3(kThreads), 0, stream, (const g1_xyzz*)t.buckets, half, lo_bits, sb, t.partial);
      if (half > 4096)
        launch("msm_reduce_grid_final", msm_reduce_grid_final, dim3(sb), dim3(256), 0, stream,
               (const g1_xyzz*)t.partial, half, lo_bits, out, out_part);
      else
        launch("msm_reduce_grid_final", msm_reduce_grid_final_small, dim3(sb), dim3(128), 0, stream,
               (const g1_xyzz*)t.partial, half, lo_bits, out, out_part);
      return;
    }
    const uint32_t nplanes = t.planes + (out_pair ? 1u : 0u);
    const uint32_t chunks = reduce_chunks(half, nplanes, sb);
    launch("msm_reduce_bits", msm_reduce_bits, dim3(chunks, nplanes, sb), dim3(kReduceThreads), 0, stream,
           (const g1_xyzz*)t.buckets, half, t.planes, nplanes, chunks, t.partial);
    launch("msm_reduce_bits_final", msm_reduce_bits_final, dim3(sb), dim3(256), 0, stream, (const g1_xyzz*)t.partial,
           t.planes, nplanes, chunks, out, out_part, out_pair);
  }
}

// one long MSM on the deep-window table (choose_plan): three-level sort, then the common tail over 2^(K-14) batch
// entries of 16384 buckets, then one more msm_reduce_final over the entries' (sum, weighted sum) pairs
int msm_run_deep(const MsmBases& bases, const Plan& pl, size_t offset, const fe* d_scalars, size_t n, int montgomery,
                 g1_jac* d_out, void* ws, size_t ws_bytes, hipStream_t stream) {
  const DeepLayout L = deep_layout(pl, n);
  if (ws_bytes < L.total) return (int)hipErrorInvalidValue;
  char* base = reinterpret_cast<char*>(ws);
  auto u32 = [&](size_t off) { return reinterpret_cast<uint32_t*>(base + off); };
  auto pts = [&](size_t off) { return reinterpret_cast<g1_xyzz*>(base + off); };
  const uint32_t c = pl.c, W = pl.windows, K = c - 1, nblk = (uint32_t)L.nblk;
  const uint32_t S1 = 1u << pl.top, M = 1u << pl.mid;
  const uint32_t tile_words = kDigitTile * W;
  uint32_t *table = u32(L.table), *off2 = u32(L.off2), *tloc = u32(L.tloc), *chunks = u32(L.chunks);
  uint32_t *table2 = u32(L.table2), *off3 = u32(L.off3), *tloc2 = u32(L.tloc2), *mid_buf = u32(L.mid_buf);
  uint32_t *counts = u32(L.counts), *offsets = u32(L.offsets), *seg_tot = u32(L.seg_tot);
  uint32_t* sorted = chunks;  // level 3 writes where level 1 wrote: the tile chunks are dead once level 2 has read them
  auto scan = [&](const uint32_t* in, uint32_t* out, size_t nbs) {
    const uint32_t nb = (uint32_t)nbs;
    if (nb <= 4 * kScanSeg) {
      launch("msm_scan", msm_scan<0>, dim3(1), dim3(1024), 0, stream, in, out, nb, (uint32_t*)nullptr, 0u);
      return;
    }
    const uint32_t ns = (nb + kScanSeg - 1) / kScanSeg;
    launch("msm_scan_seg", msm_scan_seg, dim3(ns, 1), dim3(1024), 0, stream, in, out, nb, ns, seg_tot);
    launch("msm_scan_tot", msm_scan_tot, dim3(1), dim3(1024), 0, stream, seg_tot, ns);
    launch("msm_scan_add", msm_scan_add, dim3(ns, 1), dim3(1024), 0, stream, out, nb, ns, (const uint32_t*)seg_tot);
  };
  // level 1: tiles sorted by the top key bits
  {
    static std::atomic<uint64_t> attr_set{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!(attr_set.load() >> (dev & 63) & 1)) {
      for (const void* f : {reinterpret_cast<const void*>(msm_digits_local<17, true>),
                            reinterpret_cast<const void*>(msm_digits_local<18, true>),
                            reinterpret_cast<const void*>(msm_digits_local<19, true>),
                            reinterpret_cast<const void*>(msm_digits_local<20, true>),
                            reinterpret_cast<const void*>(msm_digits_local<21, true>),
                            reinterpret_cast<const void*>(msm_digits_local<22, true>)})
        hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
      attr_set.fetch_or(1ull << (dev & 63));
    }
    auto kern = c == 17   ? msm_digits_local<17, true>
                : c == 18 ? msm_digits_local<18, true>
                : c == 19 ? msm_digits_local<19, true>
                : c == 20 ? msm_digits_local<20, true>
                : c == 21 ? msm_digits_local<21, true>
                          : msm_digits_local<22, true>;
    const size_t lds_bytes = sizeof(uint32_t) * (2 * (size_t)S1 + (size_t)kDigitTile * W);
    launch("msm_digits_local", kern, dim3(nblk), dim3(kDigitThreads), lds_bytes, stream, d_scalars, (size_t)0, 1u,
           (size_t)0, n, n, 1u, montgomery, c, W, nblk, 1u, pl.sub_bits, bases.top_shift3, table, tloc, chunks,
           (uint32_t*)nullptr, 0u, u32(L.totals), chained_enabled() ? L.sbp + 1 : 0u);
  }
  scan(table, off2, (size_t)S1 * nblk);
  // level 2: (super-bin, group of tiles) sorted by the middle key bits, written back in super-bin-major order
  launch("msm_deep_sort_mid", msm_deep_sort<0>, dim3(pl.groups, S1), dim3(kDeepThreads), 0, stream, (const uint32_t*)chunks,
         (const uint32_t*)table, (const uint32_t*)tloc, (const uint32_t*)off2, (const uint32_t*)nullptr, nblk, pl.G,
         pl.groups, M, pl.low, tile_words, bases.n, offset, mid_buf, table2, tloc2);
  scan(table2, off3, (size_t)S1 * M * pl.groups);
  // level 3: (super-bin, mid-bin) = 128 buckets sorted by the low key bits -> bucket lists, counts, offsets
  launch("msm_deep_sort_low", msm_deep_sort<1>, dim3(M, S1), dim3(kDeepThreads), 0, stream, (const uint32_t*)mid_buf,
         (const uint32_t*)table2, (const uint32_t*)tloc2, (const uint32_t*)off2, (const uint32_t*)off3, nblk, pl.G,
         pl.groups, M, pl.low, tile_words, bases.n, offset, sorted, counts, offsets);
  Tail t{};
  t.ext = pl.ext;
  t.counts = counts;
  t.offsets = offsets;
  t.sorted = sorted;
  t.item_off = u32(L.item_off);
  t.item_base = u32(L.item_base);
  t.totals = u32(L.totals);
  t.item_bucket = u32(L.item_bucket);
  t.item_sub = u32(L.item_sub);
  t.item_pts = pts(L.item_pts);
  t.buckets = pts(L.buckets);
  t.partial = pts(L.partial);
  t.per = 0;  // offsets[] are positions in the one sorted array
  t.max_items = L.max_items;
  t.entries = (size_t)W * n;
  t.half = kDeepEntryBuckets;
  t.sb = L.sbp;
  t.item_len = L.item_len;
  t.planes = 15;  // (unused: the deep plan reduces by running sums)
  t.seg_len = kDeepSegLen;
  t.reduce_half = kDeepReduceBuckets;
  t.chained = chained_enabled();
  g1_xyzz* pairs = pts(L.pairs);
  run_tail(t, nullptr, nullptr, pairs, stream);
  // reduction entry e holds buckets e * 4096 ..: total = sum_e T_e + 4096 * sum_e e * S_e
  if (quad_reduce_final())
    launch("msm_reduce_final", msm_reduce_final_quad, dim3(1), dim3(256), 0, stream, (const g1_xyzz*)pairs,
           kDeepReduceBuckets, (uint32_t)(((size_t)1 << K) / kDeepReduceBuckets), d_out, (g1_xyzz*)nullptr,
           (g1_xyzz*)nullptr);
  else
    launch("msm_reduce_final", msm_reduce_final, dim3(1), dim3(64), 0, stream, (const g1_xyzz*)pairs, kDeepReduceBuckets,
           (uint32_t)(((size_t)1 << K) / kDeepReduceBuckets), d_out, (g1_xyzz*)nullptr, (g1_xyzz*)nullptr);
  return 0;
}

// one slice of a launch: `batch` MSMs, all kernels
int msm_run_slice(const MsmBases& bases, const Plan& pl, size_t offset, const fe* d_scalars, size_t outer_stride,
                  uint32_t inner, size_t inner_stride, size_t n, uint32_t first, uint32_t batch, int montgomery,
                  g1_jac* d_out, void* ws, size_t ws_bytes, hipStream_t stream) {
  const uint32_t c = pl.c, W = pl.windows, parts = pl.parts;
  const size_t n_sub = pl.n_sub;
  const uint32_t half = 1u << (c - 1);
  const uint32_t sb = batch * parts;  // sub-MSMs of this slice
  WsLayout L = ws_layout(c, W, n_sub, sb, pl.sub_bits, parts > 1);
  if (ws_bytes < L.total) return (int)hipErrorInvalidValue;
  char* base = reinterpret_cast<char*>(ws);
  uint32_t* counts = reinterpret_cast<uint32_t*>(base + L.counts);
  uint32_t* offsets = reinterpret_cast<uint32_t*>(base + L.offsets);
  uint32_t* sorted = reinterpret_cast<uint32_t*>(base + L.sorted);
  g1_xyzz* partial = reinterpret_cast<g1_xyzz*>(base + L.partial);
  g1_xyzz* part_pts = parts > 1 ? reinterpret_cast<g1_xyzz*>(base + L.part_pts) : nullptr;
  const size_t per = (size_t)W * n_sub;
  // the scalars of MSM `first + b`: the addressing of msm_run with the slice's first MSM folded into the pointer
  // (slices start at multiples of `inner` or the launch has inner == 1 / a single slice)
  const fe* sc0 = d_scalars + (size_t)(first / inner) * outer_stride + (size_t)(first % inner) * inner_stride;
  // exclusive scan of `nb` counters per sub-MSM: one workgroup per entry, or - for a long table - segments in parallel
  uint32_t* seg_tot = reinterpret_cast<uint32_t*>(base + L.seg_tot);
  const bool chained = chained_enabled();
  // the long-table scan as one chained launch: its flag words (seg_tot) and ticket are zeroed by the digits kernel
  auto scan_is_chained = [&](uint32_t nb) {
    return chained && nb > 4 * kScanSeg && (size_t)((nb + kScanSeg - 1) / kScanSeg) * sb <= 65535;
  };
  auto scan_counts = [&](const uint32_t* in, uint32_t* out, uint32_t nb) {
    if (nb <= 4 * kScanSeg) {
      launch("msm_scan", msm_scan<0>, dim3(sb), dim3(1024), 0, stream, in, out, nb, (uint32_t*)nullptr, 0u);
      return;
    }
    const uint32_t ns = (nb + kScanSeg - 1) / kScanSeg;
    if (scan_is_chained(nb)) {
      launch("msm_scan_chained", msm_scan_chained, dim3(ns * sb), dim3(1024), 0, stream, in, out, nb, ns, seg_tot,
             seg_tot + (size_t)ns * sb);
      return;
    }
    launch("msm_scan_seg", msm_scan_seg, dim3(ns, sb), dim3(1024), 0, stream, in, out, nb, ns, seg_tot);
    launch("msm_scan_tot", msm_scan_tot, dim3(sb), dim3(1024), 0, stream, seg_tot, ns);
    launch("msm_scan_add", msm_scan_add, dim3(ns, sb), dim3(1024), 0, stream, out, nb, ns, (const uint32_t*)seg_tot);
  };

  {
    const uint32_t nblk = (uint32_t)L.nblk;
    uint32_t* table = reinterpret_cast<uint32_t*>(base + L.table);
    uint32_t* off2 = reinterpret_cast<uint32_t*>(base + L.off2);
    uint32_t* tloc = reinterpret_cast<uint32_t*>(base + L.tloc);
    uint32_t* chunk_buf = reinterpret_cast<uint32_t*>(base + L.chunks);
    const uint32_t bins = half >> pl.sub_bits;  // sort keys of the tile-local level (= buckets without a 2nd level)
    const uint32_t total_bins = bins * sb;
    size_t lds_bytes = sizeof(uint32_t) * (2 * (size_t)bins + (size_t)kDigitTile * W);
    // (function attributes are per device: a process may drive several)
    static std::atomic<uint64_t> attr_set{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!(attr_set.load() >> (dev & 63) & 1)) {
      for (const void* f : {reinterpret_cast<const void*>(msm_digits_local<0>),
                            reinterpret_cast<const void*>(msm_digits_local<11>),
                            reinterpret_cast<const void*>(msm_digits_local<13>),
                            reinterpret_cast<const void*>(msm_digits_local<15>)})
        hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
      attr_set.fetch_or(1ull << (dev & 63));
    }
    auto digits_kernel = c == 13   ? msm_digits_local<13>
                         : c == 15 ? msm_digits_local<15>
                         : c == 11 ? msm_digits_local<11>
                                   : msm_digits_local<0>;
    launch("msm_digits_local", digits_kernel, dim3(sb == 1 ? nblk : nblk * ((sb + 7) / 8) * 8), dim3(kDigitThreads),
           lds_bytes, stream,
           sc0, outer_stride, inner, inner_stride, n, n_sub, parts, montgomery, c, W, nblk, sb, pl.sub_bits, 0u, table,
           tloc, chunk_buf, seg_tot,
           scan_is_chained(bins * nblk) ? (uint32_t)(((bins * nblk + kScanSeg - 1) / kScanSeg) * sb + 1) : 0u,
           reinterpret_cast<uint32_t*>(base + L.totals), chained ? sb + 1 : 0u);
    scan_counts(table, off2, bins * nblk);
    if (pl.sub_bits) {
      // two-level sort: bins are finished per workgroup straight from the tile chunks
      launch("msm_sort_level2", msm_sort_level2, dim3(bins, sb), dim3(kThreads), 0, stream, (const uint32_t*)chunk_buf,
             (const uint32_t*)table, (const uint32_t*)tloc, (const uint32_t*)off2, per, bins, nblk, W, pl.sub_bits,
             bases.n, offset, n_sub, parts, counts, offsets, sorted);
    } else {
      // (the buckets' list ranges - once msm_bucket_ranges, a launch of its own - are written by the rows of tile 0)
      size_t rows = (size_t)total_bins * nblk;
      launch("msm_scatter", msm_scatter_runs, dim3((unsigned)((rows + kThreads - 1) / kThreads)), dim3(kThreads), 0,
             stream, (const uint32_t*)chunk_buf, (const uint32_t*)table, (const uint32_t*)tloc, (const uint32_t*)off2,
             per, bins, nblk, W, rows, bases.n, offset, n_sub, parts, sorted, counts, offsets);
    }
  }
  Tail t{};
  t.ext = pl.ext;
  t.counts = counts;
  t.offsets = offsets;
  t.sorted = sorted;
  t.item_off = reinterpret_cast<uint32_t*>(base + L.item_off);
  t.item_base = reinterpret_cast<uint32_t*>(base + L.item_base);
  t.totals = reinterpret_cast<uint32_t*>(base + L.totals);
  t.item_bucket = reinterpret_cast<uint32_t*>(base + L.item_bucket);
  t.item_sub = reinterpret_cast<uint32_t*>(base + L.item_sub);
  t.item_pts = reinterpret_cast<g1_xyzz*>(base + L.item_pts);
  t.buckets = reinterpret_cast<g1_xyzz*>(base + L.buckets);
  t.partial = partial;
  t.per = per;
  t.max_items = L.max_items;
  t.entries = per * sb;
  t.half = half;
  t.sb = sb;
  t.item_len = choose_item_len(per * sb, (size_t)half * sb);
  t.planes = c;
  t.chained = chained;
  run_tail(t, d_out + first, part_pts, nullptr, stream);
  if (parts > 1)
    launch("msm_sum_parts", msm_sum_parts, dim3(batch), dim3(64), 0, stream, (const g1_xyzz*)part_pts, parts, d_out + first);
  return 0;  // launch failures are latched by launch() and reported by take_launch_error()
}
}  // namespace

int msm_run(const MsmBases& bases, size_t offset, const fe* d_scalars, size_t outer_stride, uint32_t inner,
            size_t inner_stride, size_t n, uint32_t batch, int montgomery, g1_jac* d_out, void* ws, size_t ws_bytes,
            hipStream_t stream) {
  if (inner == 0) inner = 1;
  if (batch == 0) return 0;
  if (offset + n > bases.n) return (int)hipErrorInvalidValue;
  if (n == 0) {
    launch("msm_fill_inf", msm_fill_inf, dim3((batch + 255) / 256), dim3(256), 0, stream, d_out, batch);
    return 0;  // launch failures are latched by launch() and reported by take_launch_error()
  }
  const Plan pl = choose_plan(bases, n, batch);
  if (pl.deep) return msm_run_deep(bases, pl, offset, d_scalars, n, montgomery, d_out, ws, ws_bytes, stream);
  uint32_t slice = batch_slice(pl, batch);
  if (slice < batch && inner > 1) slice = slice >= inner ? slice / inner * inner : 0;  // slices start on an `inner` boundary
  if (slice == 0) return (int)hipErrorInvalidValue;
  for (uint32_t first = 0; first < batch; first += slice) {
    int rc = msm_run_slice(bases, pl, offset, d_scalars, outer_stride, inner, inner_stride, n, first,
                           std::min(slice, batch - first), montgomery, d_out, ws, ws_bytes, stream);
    if (rc) return rc;
  }
  return 0;
}

}  // namespace cap
