// Radix-2 NTT / iNTT over BN254 Fr for gfx950 (K2 of SURVEY.md §8a).
//
// Replaces ark-poly 0.3.0 `Radix2EvaluationDomain::{fft,ifft,coset_fft,
// coset_ifft}_in_place` (Cargo.lock:194-196) reached from
// src/proof/transfer.rs:181-186 (PlonkKzgSnark::prove).  Semantics kept:
// natural order in / out, omega_n = omega_28^(2^(28-log n)), coset generator 5,
// inverse includes n^-1, coset inverse scales by 5^-i afterwards.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "field.hpp"

namespace cap {

// Per-domain-size device tables (built once per log_n, kept resident).
struct NttDomain {
  uint32_t log_n = 0;
  fe* tw_fwd = nullptr;     // omega_n^e,  e in [0, n)
  fe* tw_inv = nullptr;     // omega_n^-e, e in [0, n)
  fe* coset_fwd = nullptr;  // 5^i,           i in [0, n)
  fe* coset_inv = nullptr;  // n^-1 * 5^-i,   i in [0, n)
  fe n_inv;                 // n^-1 (Montgomery)
  // the same tables in the internal Montgomery form of the lazy 29-bit field (x * 2^261, canonical): what the
  // NTT kernels multiply by.  (tw_fwd above stays in arkworks' form for the kernels that have not moved yet.)
  fe* tw29_fwd = nullptr;
  fe* tw29_inv = nullptr;
  fe* coset29_fwd = nullptr;
  fe* coset29_inv = nullptr;
  fe n_inv29;
};

// Small-size twiddles shared by every domain (internal form): fwd[s] -> omega_{2^s}^i, i < 2^(s-1), s <= kMaxLogTile
constexpr int kMaxLogTile = 11;
struct NttSmallTables {
  fe* fwd[kMaxLogTile + 1] = {nullptr};
  fe* inv[kMaxLogTile + 1] = {nullptr};
  // the same values unpacked to the 9 x 29-bit limbs the butterflies multiply by (36 B per entry): saves the
  // 8 -> 9 word unpacking in every butterfly
  uint32_t* fwd_u[kMaxLogTile + 1] = {nullptr};
  uint32_t* inv_u[kMaxLogTile + 1] = {nullptr};
  // tile counters of persistent pass launches (ntt.hip: a launch of many tiles is a few workgroups per CU that take tile
  // after tile from a counter); one slot per launch, reused in rotation on the context's one stream
  uint32_t* pass_counters = nullptr;
  mutable uint32_t next_counter = 0;
};
constexpr uint32_t kNttPassCounters = 256;

// Host-side table construction (runs tiny setup kernels).  Returns hipError_t as int.
int ntt_build_small_tables(NttSmallTables* t, hipStream_t stream);
int ntt_build_domain(NttDomain* d, uint32_t log_n, hipStream_t stream);
void ntt_free_domain(NttDomain* d);
void ntt_free_small_tables(NttSmallTables* t);

// omega_{2^log_n} in Montgomery form (host)
fe ntt_root_of_unity(uint32_t log_n);

// In-place (from the caller's view) batched transform of `count` arrays of 2^log_n
// elements, array b at data + b * stride_elems.  `scratch` must hold count * 2^log_n
// elements.  dir: 0 forward, 1 inverse.  coset: 0/1.
// Optional out-of-place addressing: the transform reads array q of the batch from
// src + (q / src_group) * src_outer + (q % src_group) * src_inner, treating elements at or beyond src_len as zero (a
// polynomial shorter than the domain needs no zero-padded copy), and writes it to
// data + (q / dst_group) * dst_outer + (q % dst_group) * dst_inner.  `scratch` then holds count * 2^log_n elements.
struct NttIo {
  const fe* src;
  size_t src_outer, src_inner, src_len;
  uint32_t src_group;
  size_t dst_outer, dst_inner;
  uint32_t dst_group;
  // decimated input (the 3 * 2^k transforms below): element g of array q is src[...][g * src_elem_stride]; src_len
  // and the pre-scale table are indexed by that source position plus (q % src_group) * src_inner
  uint32_t src_elem_stride = 1;
  uint32_t src_group2 = 1;        // second grouping level above src_group (see PassParams::in_group2)
  size_t src_inner2 = 0;
  uint32_t dst_group2 = 1;        // the same for the output
  size_t dst_inner2 = 0;
  size_t pre_inner = 0;           // pre-scale table of array q: pre_scale + (q % src_group) * pre_inner
  const fe* pre_scale = nullptr;  // overrides the coset table of the domain (internal form)
  int lazy_out = 0;               // leave results weakly reduced (< 2r) instead of canonical
};
int ntt_run(const NttDomain& dom, const NttSmallTables& small, fe* data, fe* scratch, size_t stride_elems,
            uint32_t count, int dir, int coset, hipStream_t stream, const NttIo* io = nullptr);

// ---- evaluation on N = 3 * 2^k points ------------------------------------------------------------------------
// The quotient polynomial of the prover has degree < 5n + 8, so 6n = 3 * 2^(log n + 1) points carry it; jf-plonk
// evaluates on 8n only because it wants a power of two.  With M = 2^k = 2n and omega_N = omega_3 * omega_M^c,
// 3c = 1 (mod M)  (so omega_N^3 = omega_M and omega_N^6 = omega_n), the coset 5 <omega_N> is the union of the three
// cosets s_a <omega_M>, s_a = 5 omega_N^a, a = 0, 1, 2, and the evaluations are kept in that order:
//   index a M + k  <->  the point s_a omega_M^k          ("next row", x omega_n, is index a M + (k + 2 mod M)).
// Forward: the evaluations of a polynomial with at most M coefficients on the three cosets are three ordinary
// M-point coset transforms of the SAME zero-extended input - no radix-3 stage, no strided access.
// Inverse (the quotient, up to 3M coefficients): three ordinary inverse transforms of the blocks, then one radix-3
// stage:  t[k + M b] = 5^-(k + M b) / 3 * (Y0[k] + w3^-b omega_N^-k Y1[k] + w3^-2b omega_N^-2k Y2[k]),  w3 = omega_N^M.
struct Ntt3Domain {
  uint32_t log_m = 0;           // M = 2^log_m, N = 3 M
  fe* xs_ext = nullptr;         // the N points in the order above, arkworks form
  fe* xs29 = nullptr;           // the same, internal form
  fe* pre3 = nullptr;           // [3][M]: 32 * s_a^i, internal form: pre-scale of the forward sub-transform a
                                //         (arkworks-form coefficients in, internal-form evaluations out)
  fe* tw29_inv = nullptr;       // omega_N^-i, i < 2M, internal form (inverse radix-3 stage)
  fe* coset_inv_ext = nullptr;  // 5^-i / 3, i < N, as an arkworks-form integer: internal-form data in, arkworks out
  fe w3inv_29;                  // omega_N^-M, internal form
  fe omega;                     // omega_N, arkworks form (host)
};
int ntt3_build_domain(Ntt3Domain* d, uint32_t log_m, hipStream_t stream);
void ntt3_free_domain(Ntt3Domain* d);
// Forward: `count` polynomials (arkworks-form coefficients read through `io`, zero-extended beyond io.src_len <= M) ->
// internal-form evaluations (weakly reduced) in the block order above, written through io's dst addressing.
// dom_m: the 2^log_m domain.  scratch: 3 M count elements.
int ntt3_forward(const Ntt3Domain& d3, const NttDomain& dom_m, const NttSmallTables& small, fe* data, NttIo io,
                 uint32_t count, fe* scratch, hipStream_t stream);
// Inverse in place: data[q * N .. +N) internal-form evaluations (block order) -> arkworks-form coefficients (natural
// order, canonical).  scratch: 6 M count elements.
int ntt3_inverse(const Ntt3Domain& d3, const NttDomain& dom_m, const NttSmallTables& small, fe* data, uint32_t count,
                 fe* scratch, hipStream_t stream);

// out[i] = internal Montgomery form (x * 2^261, canonical) of the arkworks-form value in[i]
void ntt_table_to_internal(fe* out, const fe* in, size_t n, hipStream_t stream);

}  // namespace cap
