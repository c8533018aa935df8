// Lazy-carry BN254 field arithmetic for gfx950: 9 limbs of 29 bits, Montgomery radix R' = 2^261.
//
// Why a second representation (measured on MI355X, profiles/ubench_r01.txt, tools/ubench_mul29.hip):
// with saturated 32-bit limbs (field.hpp) every row of a multiplication needs a v_addc_co_u32 carry chain,
// and v_addc costs nearly as much as v_mad_u64_u32.  With 29-bit limbs the 81 + 81 partial products of a
// Montgomery multiplication accumulate in 64-bit columns by v_mad_u64_u32 alone - no carry instruction in
// the product phase - which is 1.2x (saturated) to 1.7x (one wave per SIMD) faster, and because
// 2^261 >> p no conditional subtraction is ever needed: values stay "lazy" (a few multiples of p).
//
// Contract (all bounds are checked by tests/host tools with CAP_FL_CHECK):
//   * "normalized": limbs 0..7 < 2^29, limb 8 < 2^29  (value < 2^261).
//   * mul / sqr:  operands with limbs < 2^30 (a normalized value or one lazy sum of two); returns a
//     normalized value < p * (1 + A*B/169) for operands < A*p, B*p.
//   * add: limb-wise, no carry; sub(a, b) = a + 16p - b for b < 15.9 p with limbs < 2^30, normalized result.
//   * weak_reduce: any normalized value -> normalized value < 2p.   canonical: -> the unique value < p.
// Memory format: 8 x u32 (the same 32 bytes as `fe`), any value < 2^256.
#pragma once
#include "field.hpp"

namespace cap {

struct fl {
  uint32_t v[9];
};

struct FqP29 {
  using Base = FqP;
  static constexpr uint32_t MOD[9] = {0x187cfd47u, 0x010460b6u, 0x1c72a34fu, 0x02d522d0u, 0x1585d978u,
                                      0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
  static constexpr uint32_t NINV = 0x04866389u;  // -p^-1 mod 2^29
  static constexpr uint32_t ONE[9] = {0x157ccc21u, 0x141c2758u, 0x185230d3u, 0x014c0419u, 0x0aa36fb9u,
                                      0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};  // 2^261 mod p
  static constexpr uint32_t C256[9] = {0x058f0d9du, 0x1aea1c6eu, 0x11c2cf74u, 0x11d651ebu, 0x1462c0a7u,
                                       0x11b7bc3cu, 0x1cbd99bau, 0x183340fbu, 0x000e0a77u};  // 2^256 mod p (plain)
  static constexpr uint32_t R522[9] = {0x059bac10u, 0x0d1503a3u, 0x018016b8u, 0x10ab0ca8u, 0x02632639u,
                                       0x02c0169fu, 0x169bfd53u, 0x11869d4cu, 0x002a11a6u};  // 2^522 mod p
  // 16p with limbs 0..7 inflated into [2^30, 2^31)
  static constexpr uint32_t SUB16P[9] = {0x47cfd470u, 0x50460b6au, 0x472a34eeu, 0x4d522d0cu, 0x585d977fu,
                                         0x4db40c08u, 0x4a6e140fu, 0x45c2633eu, 0x030644e5u};
  static constexpr uint32_t MU = 0x54a47u;  // floor(2^272 / p)
  static constexpr uint32_t K266[9] = {0x13349ca1u, 0x1a5d84a8u, 0x0a3e5cacu, 0x100249e0u, 0x12b951e8u, 0x0e92d304u, 0x14cb95b3u, 0x041b9d3du, 0x00058003u};  // 2^266 mod p
  // 2p with limbs 0..7 inflated into [2^29, 2^30)
  static constexpr uint32_t SUB2P[9] = {0x30f9fa8eu, 0x2208c16cu, 0x38e5469du, 0x25aa45a0u, 0x2b0bb2efu,
                                        0x25b68180u, 0x214dc281u, 0x3cb84c67u, 0x0060c89bu};
  // 4p, same inflation
  static constexpr uint32_t SUB4P[9] = {0x21f3f51cu, 0x241182dau, 0x31ca8d3bu, 0x2b548b42u, 0x361765dfu,
                                        0x2b6d0301u, 0x229b8503u, 0x397098cfu, 0x00c19138u};
  // 8p with limbs 0..7 inflated into [2^29, 2^30)
  static constexpr uint32_t SUB8P[9] = {0x23e7ea38u, 0x282305b5u, 0x23951a77u, 0x36a91686u, 0x2c2ecbbfu,
                                        0x36da0604u, 0x25370a07u, 0x32e1319fu, 0x01832272u};
  // constant-multiplicand product (mul_shoup): 2^261 - p, and -p^-1 mod 2^261
  static constexpr uint32_t NEGP[9] = {0x078302b9u, 0x1efb9f49u, 0x038d5cb0u, 0x1d2add2fu, 0x0a7a2687u,
                                       0x1d24bf3fu, 0x1f591ebeu, 0x11a3d9cbu, 0x1fcf9bb1u};
  static constexpr uint32_t NPINV[9] = {0x04866389u, 0x1e903c17u, 0x129ab261u, 0x1cfaca3du, 0x1da809edu,
                                        0x05e80c19u, 0x11af62bfu, 0x16f23111u, 0x0ff57a22u};
};
struct FrP29 {
  using Base = FrP;
  static constexpr uint32_t MOD[9] = {0x10000001u, 0x1f0fac9fu, 0x0e5c2450u, 0x07d090f3u, 0x1585d283u,
                                      0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
  static constexpr uint32_t NINV = 0x0fffffffu;
  static constexpr uint32_t ONE[9] = {0x0fffff57u, 0x1ea70ab4u, 0x052c068bu, 0x17504f49u, 0x0aa8075bu,
                                      0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};
  static constexpr uint32_t C256[9] = {0x0ffffffbu, 0x04b1a0e2u, 0x18334a6bu, 0x18ed2b3eu, 0x1462e36fu,
                                       0x11b7bc3cu, 0x1cbd99bau, 0x183340fbu, 0x000e0a77u};
  static constexpr uint32_t R522[9] = {0x05b69bd4u, 0x06170a5au, 0x020cddceu, 0x1db6310bu, 0x0e54d0ffu,
                                       0x1cf855e3u, 0x1c15e103u, 0x07d09161u, 0x000a054au};
  static constexpr uint32_t SUB16P[9] = {0x40000010u, 0x50fac9f6u, 0x45c2450du, 0x5d090f35u, 0x585d2831u,
                                         0x4db40c08u, 0x4a6e140fu, 0x45c2633eu, 0x030644e5u};
  static constexpr uint32_t MU = 0x54a47u;
  static constexpr uint32_t K266[9] = {0x0fffead7u, 0x1d5444f4u, 0x04438aa5u, 0x03b4d096u, 0x134c84dau, 0x0e92d304u, 0x14cb95b3u, 0x041b9d3du, 0x00058003u};
  static constexpr uint32_t SUB2P[9] = {0x20000002u, 0x3e1f593eu, 0x3cb848a0u, 0x2fa121e5u, 0x2b0ba505u,
                                        0x25b68180u, 0x214dc281u, 0x3cb84c67u, 0x0060c89bu};
  static constexpr uint32_t SUB4P[9] = {0x20000004u, 0x3c3eb27du, 0x39709142u, 0x3f4243ccu, 0x36174a0bu,
                                        0x2b6d0301u, 0x229b8503u, 0x397098cfu, 0x00c19138u};
  static constexpr uint32_t SUB8P[9] = {0x20000008u, 0x387d64fbu, 0x32e12286u, 0x3e84879au, 0x2c2e9418u,
                                        0x36da0604u, 0x25370a07u, 0x32e1319fu, 0x01832272u};
  static constexpr uint32_t NEGP[9] = {0x0fffffffu, 0x00f05360u, 0x11a3dbafu, 0x182f6f0cu, 0x0a7a2d7cu,
                                       0x1d24bf3fu, 0x1f591ebeu, 0x11a3d9cbu, 0x1fcf9bb1u};
  static constexpr uint32_t NPINV[9] = {0x0fffffffu, 0x170fac9fu, 0x1a446cf0u, 0x0d0c9698u, 0x02391658u,
                                        0x0c144c83u, 0x06cb8e6au, 0x03a1b068u, 0x1273f82fu};
};

#ifdef CAP_FL_CHECK
#include <assert.h>
#define CAP_FL_ASSERT(x) assert(x)
#else
#define CAP_FL_ASSERT(x) ((void)0)
#endif

// Host builds of this header also run under clang's unsigned-integer-overflow sanitizer (tests/test_field29_host.py):
// every 64-bit column sum and every 32-bit limb sum is then checked for wrap-around on the tested inputs.  The few
// places that wrap on purpose carry this attribute.
#if defined(__clang__) && !defined(__HIP_DEVICE_COMPILE__)
#define CAP_WRAPS __attribute__((no_sanitize("unsigned-integer-overflow")))
#else
#define CAP_WRAPS
#endif

// Low 32 bits of a 32 x 32 product.  On gfx950 v_mul_lo_u32 issues at a quarter of the rate of v_mad_u64_u32
// (tools/ubench_mlo.hip: 145 -> 158 G Montgomery multiplications/s when the nine digit multiplications of a
// reduction go through the 64-bit multiply-add instead), so the device build asks for the latter explicitly.
CAP_WRAPS static CAP_HD uint32_t mul_lo32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CAP_NO_MADLO)
  uint64_t r, carry;
  asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(r), "=s"(carry) : "v"(a), "v"(b));
  return (uint32_t)r;
#else
  return a * b;
#endif
}

// acc + a * b as ONE 64-bit multiply-add even where only the low word of the sum is live: left to itself the compiler
// narrows such a product to v_mul_lo_u32 + v_add, and v_mul_lo_u32 issues at a quarter of the multiply-add's rate (the
// top column of Fl::mul_shoup: 18 of its 143 products - isa_mix counted 3.8 % v_mul_lo_u32 in the NTT passes).
CAP_WRAPS static CAP_HD uint64_t mad_wide(uint32_t a, uint32_t b, uint64_t acc) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CAP_NO_MADLO) && !defined(CAP_NO_MADWIDE)
  uint64_t r, carry;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(carry) : "v"(a), "v"(b), "v"(acc));
  return r;
#else
  return acc + (uint64_t)a * b;
#endif
}

// SCHED selects the schedule of the Montgomery multiplication (same arithmetic, same results, same bounds):
//   0  row-wise (operand scanning): 18 independent 64-bit column accumulators, every product lands in its column,
//      then the carries walk up with a 64-bit shift + add per column.  Best where a kernel keeps many independent
//      multiplications in flight (NTT butterflies, the quotient kernel).
//   1  column-wise (product scanning): one running accumulator per column that starts from the carry of the column
//      below, so the carry add is the addend of a v_mad_u64_u32 instead of its own instruction: 204 instead of 220
//      instructions per multiplication.  On gfx950 every one of these instructions issues at the same rate
//      (tools/ubench_mulcol.hip: v_mad_u64_u32 36 T lane-ops/s = one wave instruction per 4 cycles per SIMD), so the
//      count is the cost: 172 instead of 139-158 G multiplications/s in isolation.  Used by the MSM kernels.
// A translation unit picks its default with CAP_FL_SCHED (the two instantiations are distinct types).
#ifndef CAP_FL_SCHED
#ifdef CAP_FL_COLWISE
#define CAP_FL_SCHED 1
#else
#define CAP_FL_SCHED 0
#endif
#endif
// LLVM's reassociation pass would sort the addends of a column by rank and add the carry (the youngest value) last,
// as a separate 64-bit add - the row-wise code again.  A second, empty use of every partial sum stops it from
// merging the additions into one expression tree (it only walks through single-use values), so each
// "acc += x * y" stays a multiply-add whose addend is the running sum.  No instruction is emitted for it.
#if defined(__HIP_DEVICE_COMPILE__)
#define CAP_FL_KEEP(x) asm volatile("" ::"v"(x))
#else
#define CAP_FL_KEEP(x) ((void)0)
#endif

template <class PR, int SCHED = CAP_FL_SCHED>
struct Fl {
  static constexpr uint32_t M29 = 0x1fffffffu;

  // ---- constants / conversions ---------------------------------------------------------------------
  static CAP_HD fl zero() {
    fl r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = 0;
    return r;
  }
  static CAP_HD fl one() {  // Montgomery (R' = 2^261) form of 1
    fl r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = PR::ONE[i];
    return r;
  }
  // 8 x 32-bit words (any value < 2^256) -> 9 x 29-bit limbs, same integer
  static CAP_HD fl unpack(const fe& a) {
    fl r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      const int bit = 29 * i, w = bit >> 5, off = bit & 31;
      uint32_t lo = a.v[w] >> off;
      if (off > 3 && w + 1 < 8) lo |= a.v[w + 1] << (32 - off);
      r.v[i] = i < 8 ? (lo & M29) : lo;  // limb 8 = bits 232..255 (24 bits)
    }
    return r;
  }
  // normalized value < 2^256 -> 8 x 32-bit words
  static CAP_HD fe pack(const fl& a) {
    fe r;
#pragma unroll
    for (int w = 0; w < 8; w++) {
      const int bit = 32 * w, i = bit / 29, off = bit - 29 * i;
      uint32_t x = a.v[i] >> off;
      if (i + 1 < 9) x |= a.v[i + 1] << (29 - off);
      if (off > 26 && i + 2 < 9) x |= a.v[i + 2] << (58 - off);
      r.v[w] = x;
    }
    return r;
  }

  // ---- carries -----------------------------------------------------------------------------------------
  // limbs up to < 2^32 - 8 -> normalized (same integer; limb 8 absorbs the top carry)
  static CAP_HD fl normalize(const fl& a) {
    fl r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      uint32_t t = a.v[i] + c;
      r.v[i] = t & M29;
      c = t >> 29;
    }
    r.v[8] = a.v[8] + c;
    CAP_FL_ASSERT(r.v[8] < (1u << 29));
    return r;
  }

  // ---- additive ops ------------------------------------------------------------------------------------
  static CAP_HD fl add(const fl& a, const fl& b) {  // lazy: limbs add, no carry
    fl r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + b.v[i];
    return r;
  }
  static CAP_HD fl add_norm(const fl& a, const fl& b) { return normalize(add(a, b)); }
  // a - b + 16p, normalized.  Requires limbs(a) < 2^30, limbs(b) < 2^30 (i < 8), b < 15.9 p.
  static CAP_HD fl sub(const fl& a, const fl& b) {
    fl r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      CAP_FL_ASSERT(i == 8 ? b.v[i] <= PR::SUB16P[i] : (b.v[i] < (1u << 30) && a.v[i] < (1u << 30)));
      r.v[i] = a.v[i] + (PR::SUB16P[i] - b.v[i]);
    }
    return normalize(r);
  }
  static CAP_HD fl neg(const fl& b) { return sub(zero(), b); }
  // a - b + 2p, normalized, for a normalized b < 1.9 p (a product); limbs(a) < 2^31
  static CAP_HD fl sub2p(const fl& a, const fl& b) {
    fl r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      CAP_FL_ASSERT(i == 8 ? b.v[i] <= PR::SUB2P[i] : (b.v[i] < (1u << 29) && a.v[i] < (1u << 31)));
      r.v[i] = a.v[i] + (PR::SUB2P[i] - b.v[i]);
    }
    return normalize(r);
  }

  // 16p - b WITHOUT carrying, for a normalized b < 15.9 p: limbs in (2^29, 2^31).  Only as ONE operand of a
  // multiplication whose other operand is normalized (81 products of < 2^60 and the reduction stay below 2^64).
  static CAP_HD fl neg_lazy(const fl& b) {
    fl r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      CAP_FL_ASSERT(i == 8 ? b.v[i] <= PR::SUB16P[i] : b.v[i] < (1u << 29));
      r.v[i] = PR::SUB16P[i] - b.v[i];
    }
    return r;
  }
  // 2p - b WITHOUT carrying, for a normalized b < 1.9 p (a product): limbs < 2^30
  static CAP_HD fl neg2p_lazy(const fl& b) {
    fl r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      CAP_FL_ASSERT(i == 8 ? b.v[i] <= PR::SUB2P[i] : b.v[i] < (1u << 29));
      r.v[i] = PR::SUB2P[i] - b.v[i];
    }
    return r;
  }
  // 4p - b WITHOUT carrying, for a normalized b < 3.9 p: limbs < 2^30 (a lazy operand like a sum of two)
  static CAP_HD fl neg4p_lazy(const fl& b) {
    fl r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      CAP_FL_ASSERT(i == 8 ? b.v[i] <= PR::SUB4P[i] : b.v[i] < (1u << 29));
      r.v[i] = PR::SUB4P[i] - b.v[i];
    }
    return r;
  }
  // a - b + 16p, normalized, for a LAZY a (limbs < 2^31, e.g. the result of sub2p_lazy) and limbs(b) < 2^30
  static CAP_HD fl sub_from_lazy(const fl& a, const fl& b) {
    fl r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      CAP_FL_ASSERT(i == 8 ? b.v[i] <= PR::SUB16P[i] : (b.v[i] < (1u << 30) && a.v[i] < (1u << 31)));
      r.v[i] = a.v[i] + (PR::SUB16P[i] - b.v[i]);  // < 2^31 + 2^31
    }
    return normalize(r);
  }
  // a - b + 2p WITHOUT carrying: limbs(a) + 2^30 at most.  For a value that is only ever added to / subtracted from
  // before its next normalisation (the "u" side of the following butterfly), never a multiplicand.
  static CAP_HD fl sub2p_lazy(const fl& a, const fl& b) {
    fl r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      CAP_FL_ASSERT(i == 8 ? b.v[i] <= PR::SUB2P[i] : (b.v[i] < (1u << 29) && a.v[i] < (1u << 30)));
      r.v[i] = a.v[i] + (PR::SUB2P[i] - b.v[i]);
    }
    return r;
  }

  // ---- weak reduction: normalized x -> normalized value in [0, 2p) ------------------------------------
  static CAP_HD fl weak_reduce(const fl& x) {
    // q <= floor(x / p) <= q + 1 from the top limb: q = (x_8 * floor(2^272/p)) >> 40
    uint32_t q = (uint32_t)(((uint64_t)x.v[8] * PR::MU) >> 40);
    fl r;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      int64_t t = (int64_t)x.v[i] - (int64_t)((uint64_t)q * PR::MOD[i]) + c;
      r.v[i] = (uint32_t)t & M29;
      c = t >> 29;
    }
    int64_t t = (int64_t)x.v[8] - (int64_t)((uint64_t)q * PR::MOD[8]) + c;
    CAP_FL_ASSERT(t >= 0 && t < (1 << 26));
    r.v[8] = (uint32_t)t;
    return r;
  }
  // the unique representative < p (exact; used at the ABI boundary and in rare-path comparisons)
  static CAP_HD fl canonical(const fl& x) {
    fl r = weak_reduce(x);
#pragma unroll 1
    for (int k = 0; k < 2; k++) {  // r < 2p: at most one subtraction needed; the second pass is a no-op guard
      fl d;
      int64_t c = 0;
#pragma unroll
      for (int i = 0; i < 9; i++) {
        int64_t t = (int64_t)r.v[i] - (int64_t)PR::MOD[i] + c;
        d.v[i] = i < 8 ? ((uint32_t)t & M29) : (uint32_t)t;
        c = t >> 29;
      }
      if (c == 0) r = d;  // no borrow: r >= p
    }
    return r;
  }
  // x == 0 (mod p)?  x normalized.  Fast path: one multiplication by p^-1 mod 2^29 decides almost always.
  CAP_WRAPS static CAP_HD bool is_zero(const fl& x) {
    // x = k p  =>  k = x_0 * p^-1 mod 2^29 ; a multiple of p below 2^261 has k < 169
    uint32_t k = mul_lo32(x.v[0], 0u - PR::NINV) & M29;  // p^-1 = -NINV mod 2^29
    if (k >= 256) return false;
    fl c = canonical(x);
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) o |= c.v[i];
    return o == 0;
  }
  static CAP_HD bool eq(const fl& a, const fl& b) { return is_zero(sub(a, weak_reduce(b))); }

  // ---- Montgomery multiplication: a * b * 2^-261 mod p (lazy) --------------------------------------
  // The Montgomery digit m = c * (-p^-1) mod 2^29.  BN254's scalar field has r = 1 (mod 2^28), so -r^-1 = 2^28 - 1
  // (mod 2^29) and the digit is a shift and a subtraction (CAP_FR_DIGIT_SHIFT: +4 .. 6 % on the row-wise product at four
  // workgroups per CU, nothing at two - tools/ubench_shoup29.hip; the base field has no such luck).
  CAP_WRAPS static CAP_HD uint32_t mont_digit(uint32_t c) {
#ifdef CAP_FR_DIGIT_SHIFT
    if constexpr (PR::NINV == 0x0fffffffu) return ((c << 28) - c) & M29;
#endif
    return mul_lo32(c, PR::NINV) & M29;
  }
  static CAP_HD fl reduce_cols(uint64_t c[18]) {
#pragma unroll
    for (int k = 0; k < 9; k++) {
      uint32_t m = mont_digit((uint32_t)c[k]);
#pragma unroll
      for (int j = 0; j < 9; j++) c[k + j] += (uint64_t)m * PR::MOD[j];
      c[k + 1] += c[k] >> 29;
    }
    fl r;
#pragma unroll
    for (int k = 9; k < 17; k++) {
      r.v[k - 9] = (uint32_t)c[k] & M29;
      c[k + 1] += c[k] >> 29;
    }
    CAP_FL_ASSERT(c[17] < (1ull << 29));
    r.v[8] = (uint32_t)c[17];
    return r;
  }
  // ---- column-wise schedule ---------------------------------------------------------------------------
  template <class ProdFn>
  static CAP_HD fl montmul_cols(ProdFn prod_col) {
    uint32_t m[9];
    fl r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 17; k++) {
      acc = prod_col(k, acc);
      if (k < 9) {
#pragma unroll
        for (int i = 0; i < k; i++) {
          acc += (uint64_t)m[i] * PR::MOD[k - i];
          CAP_FL_KEEP(acc);
        }
        m[k] = mul_lo32((uint32_t)acc, PR::NINV) & M29;
        acc += (uint64_t)m[k] * PR::MOD[0];
      } else {
#pragma unroll
        for (int i = k - 8; i < 9; i++) {
          acc += (uint64_t)m[i] * PR::MOD[k - i];
          CAP_FL_KEEP(acc);
        }
        r.v[k - 9] = (uint32_t)acc & M29;
      }
      acc >>= 29;
    }
    CAP_FL_ASSERT(acc < (1ull << 29));
    r.v[8] = (uint32_t)acc;
    return r;
  }
  static CAP_HD fl mul_col(const fl& a, const fl& b) {
    return montmul_cols([&](int k, uint64_t acc) {
#pragma unroll
      for (int i = 0; i < 9; i++) {
        const int j = k - i;
        if (j >= 0 && j < 9) {
          acc += (uint64_t)a.v[i] * b.v[j];
          CAP_FL_KEEP(acc);
        }
      }
      return acc;
    });
  }
  static CAP_HD fl sqr_col(const fl& a) {
    uint32_t d[9];
#pragma unroll
    for (int i = 0; i < 9; i++) d[i] = a.v[i] << 1;  // limbs < 2^30 -> < 2^31
    return montmul_cols([&](int k, uint64_t acc) {
#pragma unroll
      for (int i = 0; i < 9; i++) {
        const int j = k - i;
        if (j > i && j < 9) {
          acc += (uint64_t)d[i] * a.v[j];
          CAP_FL_KEEP(acc);
        }
        if (j == i) {
          acc += (uint64_t)a.v[i] * a.v[i];
          CAP_FL_KEEP(acc);
        }
      }
      return acc;
    });
  }
  static CAP_HD fl mul_add_mul_col(const fl& a, const fl& b, const fl& c2, const fl& d) {
    return montmul_cols([&](int k, uint64_t acc) {
#pragma unroll
      for (int i = 0; i < 9; i++) {
        const int j = k - i;
        if (j >= 0 && j < 9) {
          acc += (uint64_t)a.v[i] * b.v[j];
          CAP_FL_KEEP(acc);
          acc += (uint64_t)c2.v[i] * d.v[j];
          CAP_FL_KEEP(acc);
        }
      }
      return acc;
    });
  }
  // ---- row-wise schedule ------------------------------------------------------------------------------
  static CAP_HD fl mul(const fl& a, const fl& b) {
    if constexpr (SCHED == 1) return mul_col(a, b);
    uint64_t c[18];
#pragma unroll
    for (int k = 0; k < 18; k++) c[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
      for (int j = 0; j < 9; j++) c[i + j] += (uint64_t)a.v[i] * b.v[j];
    return reduce_cols(c);
  }
  static CAP_HD fl sqr(const fl& a) {
    if constexpr (SCHED == 1) return sqr_col(a);
    uint64_t c[18];
#pragma unroll
    for (int k = 0; k < 18; k++) c[k] = 0;
    uint32_t d[9];
#pragma unroll
    for (int i = 0; i < 9; i++) d[i] = a.v[i] << 1;  // limbs < 2^30 -> < 2^31
#pragma unroll
    for (int i = 0; i < 9; i++) {
      c[2 * i] += (uint64_t)a.v[i] * a.v[i];
#pragma unroll
      for (int j = i + 1; j < 9; j++) c[i + j] += (uint64_t)d[i] * a.v[j];
    }
    return reduce_cols(c);
  }
  // a*b + c*d with one reduction (all four operands normalized: 18 products of < 2^58 per column)
  static CAP_HD fl mul_add_mul(const fl& a, const fl& b, const fl& c2, const fl& d) {
    if constexpr (SCHED == 1) return mul_add_mul_col(a, b, c2, d);
    uint64_t c[18];
#pragma unroll
    for (int k = 0; k < 18; k++) c[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
      for (int j = 0; j < 9; j++) c[i + j] += (uint64_t)a.v[i] * b.v[j] + (uint64_t)c2.v[i] * d.v[j];
    return reduce_cols(c);
  }

  // ---- constant-multiplicand product (round-5 VERDICT item 4) --------------------------------------------------
  // a * w mod p for a table constant w given as the pair (w, wq): w the PLAIN canonical value (< p), wq =
  // floor(w * 2^261 / p) (shoup_quotient).  No Montgomery factor: data in the internal form stays in it.
  //   q  = floor(a * wq / 2^261), from columns 7 .. 16 of the product (53 multiply-adds; what lies below column 7
  //        changes q by at most one),
  //   r  = low 261 bits of a * w + q * (2^261 - p)   (45 + 45 multiply-adds)  =  a * w - q * p  exactly,
  // 143 multiply-adds and no serial digit chain against the 171 of the Montgomery product.
  // Bounds: limbs(a) < 2^30 (a normalized value or one lazy sum of two), w and wq normalized.  With Q = floor(a w / p):
  // Q - q <= a / 2^261 + 2, so the normalized result is < 4p for a < 2^261 and < 5p for a lazy sum of two (< 2^262).
  // Two schedules, like the Montgomery product: SCHED 1 one running accumulator (the carry is the next column's addend;
  // 0.38 - 0.5 of the Montgomery product's rate in the NTT's setting: one serial chain of 143 multiply-adds), SCHED 0
  // independent 64-bit column accumulators whose carries walk up afterwards (1.11 - 1.15 of it at four workgroups per
  // CU, 1.04 at two: tools/ubench_shoup29.hip, profiles/ubench_shoup29_r06.txt).
  static CAP_HD fl mul_shoup(const fl& a, const fl& w, const fl& wq) {
    if constexpr (SCHED == 0) return mul_shoup_row(a, w, wq);
    uint32_t q[9];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 7; k < 17; k++) {
#pragma unroll
      for (int i = 0; i < 9; i++) {
        const int j = k - i;
        if (j >= 0 && j < 9) {
          acc += (uint64_t)a.v[i] * wq.v[j];
          CAP_FL_KEEP(acc);
        }
      }
      if (k >= 9) q[k - 9] = (uint32_t)acc & M29;
      acc >>= 29;
    }
    CAP_FL_ASSERT(acc < (1ull << 31));
    q[8] = (uint32_t)acc;
    fl r;
    acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
      for (int i = 0; i <= k; i++) {
        acc += (uint64_t)a.v[i] * w.v[k - i];
        CAP_FL_KEEP(acc);
        acc += (uint64_t)q[i] * PR::NEGP[k - i];
        CAP_FL_KEEP(acc);
      }
      r.v[k] = (uint32_t)acc & M29;
      acc >>= 29;
    }
    return r;  // (what is left in acc are bits 261.. of a w + q (2^261 - p): multiples of 2^261, dropped)
  }
  static CAP_HD fl mul_shoup_row(const fl& a, const fl& w, const fl& wq) {
    uint64_t c[10];  // columns 7 .. 16 of a * wq
#pragma unroll
    for (int k = 0; k < 10; k++) c[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
      for (int j = 0; j < 9; j++)
        if (i + j >= 7) c[i + j - 7] += (uint64_t)a.v[i] * wq.v[j];
    uint32_t q[9];
    c[1] += c[0] >> 29;
    c[2] += c[1] >> 29;
#pragma unroll
    for (int k = 2; k < 10; k++) {
      q[k - 2] = (uint32_t)c[k] & M29;
      if (k < 9) c[k + 1] += c[k] >> 29;
    }
    CAP_FL_ASSERT((c[9] >> 29) < (1ull << 31));
    q[8] = (uint32_t)(c[9] >> 29);
    uint64_t d[9];  // columns 0 .. 8 of a * w + q * (2^261 - p)
#pragma unroll
    for (int k = 0; k < 9; k++) d[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
      for (int j = 0; i + j < 9; j++) {
        if (i + j < 8) {
          d[i + j] += (uint64_t)a.v[i] * w.v[j] + (uint64_t)q[i] * PR::NEGP[j];
        } else {  // the top column: only its low 29 bits are kept (see mad_wide)
          d[8] = mad_wide(a.v[i], w.v[j], d[8]);
          d[8] = mad_wide(q[i], PR::NEGP[j], d[8]);
        }
      }
    fl r;
#pragma unroll
    for (int k = 0; k < 9; k++) {
      r.v[k] = (uint32_t)d[k] & M29;
      if (k < 8) d[k + 1] += d[k] >> 29;
    }
    return r;
  }
  // wq = floor(w * 2^261 / p) from t = w * 2^261 mod p, CANONICAL (the internal Montgomery form of w): the division
  // (w 2^261 - t) / p is exact and its quotient is below 2^261, so it is (-t) * p^-1 mod 2^261 - one low-half product.
  static CAP_HD fl shoup_quotient(const fl& t_canonical) {
    fl r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
      for (int i = 0; i <= k; i++) acc += (uint64_t)t_canonical.v[i] * PR::NPINV[k - i];
      r.v[k] = (uint32_t)acc & M29;
      acc >>= 29;
    }
    return r;
  }
  // a - b + 8p, normalized, for a normalized b < 7.99 p (a mul_shoup result); limbs(a) < 2^31
  static CAP_HD fl sub8p(const fl& a, const fl& b) {
    fl r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      CAP_FL_ASSERT(i == 8 ? b.v[i] <= PR::SUB8P[i] : (b.v[i] < (1u << 29) && a.v[i] < (1u << 31)));
      r.v[i] = a.v[i] + (PR::SUB8P[i] - b.v[i]);
    }
    return normalize(r);
  }
  // a - b + 8p WITHOUT carrying (see sub2p_lazy): limbs(a) + 2^30 at most, for limbs(a) < 2^30
  static CAP_HD fl sub8p_lazy(const fl& a, const fl& b) {
    fl r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      CAP_FL_ASSERT(i == 8 ? b.v[i] <= PR::SUB8P[i] : (b.v[i] < (1u << 29) && a.v[i] < (1u << 30)));
      r.v[i] = a.v[i] + (PR::SUB8P[i] - b.v[i]);
    }
    return r;
  }

  // 1 / a = a^(p-2) in the internal Montgomery form (0 for a = 0): 4-bit fixed windows, 252 squarings and at most
  // 63 + 14 multiplications.  For single-thread inversions that sit on a latency path (the saturated field's
  // square-and-multiply in field.hpp spends 380 multiplications of 300+ instructions each).
  static CAP_HD fl inv(const fl& a) {
    fl tab[16];
    tab[0] = one();
    tab[1] = a;
    for (int i = 2; i < 16; i++) tab[i] = mul(tab[i - 1], a);
    uint32_t e[9];
#pragma unroll
    for (int i = 0; i < 9; i++) e[i] = PR::MOD[i];
    e[0] -= 2;  // both moduli end in a limb >= 2
    auto window = [&](int w) {  // bits 4w .. 4w + 3 of p - 2
      const int bit = 4 * w, l = bit / 29, o = bit % 29;
      uint32_t v = e[l] >> o;
      if (o > 25 && l + 1 < 9) v |= e[l + 1] << (29 - o);
      return v & 15u;
    };
    fl r = tab[window(63)];  // p < 2^254: the top window is 3
#pragma unroll 1
    for (int w = 62; w >= 0; w--) {
      r = sqr(sqr(sqr(sqr(r))));
      const uint32_t d = window(w);
      if (d) r = mul(r, tab[d]);
    }
    return r;
  }

  // ---- forms ---------------------------------------------------------------------------------------------
  static CAP_HD fl konst(const uint32_t (&k)[9]) {
    fl r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = k[i];
    return r;
  }
  // external Montgomery form (x * 2^256 mod p, what arkworks keeps in memory) -> internal (x * 2^261)
  static CAP_HD fl from_ext(const fe& a) { return mul(unpack(a), konst(PR::K266)); }
  // internal -> external Montgomery form, canonical (< p)
  static CAP_HD fe to_ext(const fl& a) { return pack(canonical(mul(a, konst(PR::C256)))); }
  // plain integer (< 2^256) <-> internal Montgomery form
  static CAP_HD fl to_mont(const fe& a) { return mul(unpack(a), konst(PR::R522)); }
  static CAP_HD fe from_mont(const fl& a) {
    fl o = zero();
    o.v[0] = 1;
    return pack(canonical(mul(a, o)));
  }
  // lazy value (normalized, < 2^256 after the weak reduction) <-> its 32-byte memory image
  static CAP_HD fe store(const fl& a) { return pack(weak_reduce(a)); }
  static CAP_HD fl load(const fe& a) { return unpack(a); }
};

using Fq29 = Fl<FqP29>;
using Fr29 = Fl<FrP29>;

}  // namespace cap
