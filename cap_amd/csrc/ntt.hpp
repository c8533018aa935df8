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
  // form-changing coset tables (kNttOutInternal / kNttInInternal below)
  fe* coset29_fwd_x32 = nullptr;  // 5^i * 2^266: forward coset transform of arkworks-form data -> internal-form output
                                  // (coset_inv, the arkworks-form table n^-1 5^-i 2^256, does the reverse on the way back)
};

// io_form flags of ntt_run: the quotient kernel works on internal-form (x * 2^261) data; the transforms feeding and
// draining it change the form for free through their coset-scaling multiplication.
constexpr int kNttOutInternal = 1;  // forward coset transform: arkworks-form input, internal-form output
constexpr int kNttInInternal = 2;   // inverse coset transform: internal-form input, arkworks-form (canonical) output

// Small-size twiddles shared by every domain (internal form): fwd[s] -> omega_{2^s}^i, i < 2^(s-1), s <= kMaxLogTile
constexpr int kMaxLogTile = 11;
struct NttSmallTables {
  fe* fwd[kMaxLogTile + 1] = {nullptr};
  fe* inv[kMaxLogTile + 1] = {nullptr};
  // the same values unpacked to the 9 x 29-bit limbs the butterflies multiply by (36 B per entry): saves the
  // 8 -> 9 word unpacking in every butterfly
  uint32_t* fwd_u[kMaxLogTile + 1] = {nullptr};
  uint32_t* inv_u[kMaxLogTile + 1] = {nullptr};
};

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
};
int ntt_run(const NttDomain& dom, const NttSmallTables& small, fe* data, fe* scratch, size_t stride_elems,
            uint32_t count, int dir, int coset, hipStream_t stream, int io_form = 0, const NttIo* io = nullptr);

// out[i] = internal Montgomery form (x * 2^261, canonical) of the arkworks-form value in[i]
void ntt_table_to_internal(fe* out, const fe* in, size_t n, hipStream_t stream);

}  // namespace cap
