// Pippenger multi-scalar multiplication over BN254 G1 for gfx950 (K3-K6 of
// SURVEY.md §8a).
//
// Replaces ark-ec 0.3.0 `VariableBaseMSM::multi_scalar_mul(&[G1Affine],
// &[BigInteger256])` (Cargo.lock:103-105), reached from KZG10::commit under
// src/proof/transfer.rs:181-186 / src/proof/mint.rs:113 / src/proof/freeze.rs:151.
//
// MI355X-first design (not the reference's per-window rayon loop):
//  * the commit key is fixed for the life of a proving key, and 288 GB of HBM is
//    there to be used: at upload every base P_i is expanded to its W window
//    multiples 2^(c*w) * P_i (affine, 64 B each).  All W windows of a scalar then
//    fall into ONE bucket set, so the per-window running-sum reductions and the
//    254 sequential doublings of the window combine (ruinous on a 64-lane SIMT
//    machine where one field multiplication is ~1 us of latency) disappear;
//  * two such tables: c = 13 for a single MSM (few buckets, log-depth reduction)
//    and c = 15 for batches of >= 24 MSMs (17 instead of 20 mixed additions per
//    254-bit scalar; the 4x larger bucket set costs two running-sum additions per
//    bucket because the batch alone fills the chip);
//  * signed digits halve the bucket count (2^(c-1) buckets);
//  * bucket lists come from a counting sort done per 1024-scalar tile in LDS, a
//    table scan and run copies (c = 13), or - when a tile holds only about one
//    entry per bucket (c = 15) - a tile sort on 128-bucket bins followed by a
//    per-bin LDS sort that reads the tile chunks directly; tile entries are
//    15-bit (window, scalar-in-tile) pairs, the table index is rebuilt by the
//    consumer of a run, so a table may hold up to 2^31 points;
//  * an MSM of more than 2^18 points (and every MSM of more than 73 728 points
//    in a batch) is cut into sub-MSMs over consecutive point ranges that go
//    through the batched path together; one more kernel adds their results -
//    the same shape as sharding an MSM over several GPUs, one level down;
//  * the unit of parallelism is a work item (a slice of one bucket's list), the
//    items of a launch are ordered by length and alternate direction per MSM so
//    that lanes of a wave, and the eight XCDs, finish together.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "curve.hpp"

namespace cap {

struct MsmBases {
  size_t n = 0;              // number of caller bases
  uint32_t c = 0;            // window bits
  uint32_t windows = 0;      // W
  g1_affine* ext = nullptr;  // [W][n]: ext[w*n + i] = 2^(c*w) * P_i  (w = 0 is the input itself)
  // Second table with wider windows for large batches (see msm.hip, "two-level sort"): fewer, larger digits mean
  // fewer mixed additions per scalar; only worth it when the batch alone fills the chip.  Null when not built
  // (fewer than 4096 points, or c itself is already the wide window: tables of more than 2^18 points).
  uint32_t c2 = 0, windows2 = 0;
  g1_affine* ext2 = nullptr;
  // Small-launch table (experiment, CAPGPU_MSM_SMALL_C = 9 .. 12; not built by default): narrower windows - c = 11: 1024
  // buckets, 24 digits - for launches of a handful of MSMs.  Measured slower than the c = 13 table in round 4 (the
  // per-bucket combine of four times as many work items outweighs the shorter bucket reduction; msm.hip: small_c).
  uint32_t c0 = 0, windows0 = 0;
  g1_affine* ext0 = nullptr;
  // Third table for long single MSMs (>= 2^19 points, see msm.hip "deep sort"): windows of 20 bits (22 from 2^23 points on)
  // on ONE bucket set shared by all points - 13 (12) digits per scalar instead of 17.  Null when not built.
  uint32_t c3 = 0, windows3 = 0;
  g1_affine* ext3 = nullptr;
  // With 22-bit windows (tables of >= 2^23 points; CAPGPU_MSM_DEEP_SHIFT forces it on or off): the top window of the deep
  // table holds only 254 - c3 (W3 - 1) scalar bits, so its digits would all fall into the lowest 2^top_bits buckets - 32 of
  // the 16384 sort bins at c3 = 22.  With the shift the table stores the top window's points 2^top_shift3 times smaller and
  // the digit is multiplied by 2^top_shift3: the same multiples, spread over the whole bucket range (msm.hip:
  // deep_top_shift, with the measurement).
  uint32_t top_shift3 = 0;
};

struct MsmWorkspace {
  void* buf = nullptr;
  size_t bytes = 0;
};

uint32_t msm_choose_window(size_t n);
uint32_t msm_num_windows(uint32_t c);

// Expand `n` affine bases (device, Montgomery, (0,0) = infinity) into the window table.
int msm_precompute(MsmBases* out, const g1_affine* d_bases, size_t n, uint32_t c, hipStream_t stream);
void msm_free_bases(MsmBases* b);

size_t msm_workspace_bytes(const MsmBases& bases, size_t n, uint32_t batch);
// "c=15 windows=18 sort=two-level parts=256 n_sub=65536 slice=1": which table, sort and split a launch would use
const char* msm_plan_describe(const MsmBases& bases, size_t n, uint32_t batch, char* buf, size_t cap);

// out[b] = sum_{i<n} scalars_b[i] * bases[offset + i]   for b < batch, where
// scalars_b = d_scalars + (b / inner) * outer_stride + (b % inner) * inner_stride   (elements).
// d_scalars: device, 32 B each; montgomery != 0 -> converted to canonical integers first.
// d_out: device, batch x g1_jac (Montgomery).  ws must hold msm_workspace_bytes().
int msm_run(const MsmBases& bases, size_t offset, const fe* d_scalars, size_t outer_stride, uint32_t inner,
            size_t inner_stride, size_t n, uint32_t batch, int montgomery, g1_jac* d_out, void* ws, size_t ws_bytes,
            hipStream_t stream);

}  // namespace cap
