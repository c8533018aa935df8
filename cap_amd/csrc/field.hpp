// BN254 Fq / Fr Montgomery arithmetic for gfx950 (and for the host side of the
// same library).  K1 of SURVEY.md §8a: the device field library every kernel
// inlines.
//
// Replaces, on the reference's hot path, ark-ff 0.3.0 `Fp256<FqParameters>` /
// `Fp256<FrParameters>` (Cargo.lock:153-155; curve constants ark-bn254 0.3.0,
// Cargo.lock:81-83) reached from src/proof/transfer.rs:181-186.
//
// Representation: 8 x u32 little-endian limbs, Montgomery form with R = 2^256.
// In memory this is byte-identical to arkworks' 4 x u64 little-endian limbs, so
// buffers cross the C ABI without repacking.
//
// gfx950 notes: the inner products compile to v_mad_u64_u32 (32x32+64 -> 64);
// a row of 8 independent mads is followed by one v_addc carry chain.  No MFMA:
// this is 254-bit modular integer work, not a dense contraction.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#include <hip/hip_runtime.h>
#define CAP_HD __host__ __device__ __forceinline__
#else
#define CAP_HD inline
#endif

namespace cap {

struct alignas(16) fe {
  uint32_t v[8];
};

// ---- per-field parameters -------------------------------------------------
struct FqP {
  static constexpr uint32_t MOD[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u,
                                      0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  static constexpr uint32_t NINV = 0xe4866389u;  // -p^-1 mod 2^32
  static constexpr uint32_t R1[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u,
                                     0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
  static constexpr uint32_t R2[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u,
                                     0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
};
struct FrP {
  static constexpr uint32_t MOD[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u,
                                      0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  static constexpr uint32_t NINV = 0xefffffffu;  // -r^-1 mod 2^32
  static constexpr uint32_t R1[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u,
                                     0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
  static constexpr uint32_t R2[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u,
                                     0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
};


// 32-bit add/sub with carry; lowers to v_addc_co_u32 / v_subb_co_u32 chains on
// gfx950 (plain 64-bit C arithmetic makes hipcc emit v_lshl_add_u64 + v_mov
// pairs instead: 4x the instruction count of this form).
static CAP_HD uint32_t addc32(uint32_t a, uint32_t b, uint32_t& c) {
#if defined(__has_builtin) && __has_builtin(__builtin_addc)
  unsigned co;
  uint32_t r = __builtin_addc(a, b, c, &co);
  c = co;
  return r;
#else
  uint64_t s = (uint64_t)a + b + c;
  c = (uint32_t)(s >> 32);
  return (uint32_t)s;
#endif
}
static CAP_HD uint32_t subb32(uint32_t a, uint32_t b, uint32_t& br) {
#if defined(__has_builtin) && __has_builtin(__builtin_subc)
  unsigned bo;
  uint32_t r = __builtin_subc(a, b, br, &bo);
  br = bo;
  return r;
#else
  uint64_t d = (uint64_t)a - b - br;
  br = (uint32_t)(d >> 63);
  return (uint32_t)d;
#endif
}

template <class PR>
struct Fp {
  // -- constants ------------------------------------------------------------
  static CAP_HD fe zero() {
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = 0;
    return r;
  }
  static CAP_HD fe one() {  // Montgomery form of 1
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = PR::R1[i];
    return r;
  }
  static CAP_HD fe r2() {
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = PR::R2[i];
    return r;
  }
  static CAP_HD fe modulus() {
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = PR::MOD[i];
    return r;
  }

  // -- predicates -----------------------------------------------------------
  static CAP_HD bool is_zero(const fe& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.v[i];
    return o == 0;
  }
  static CAP_HD bool eq(const fe& a, const fe& b) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= (a.v[i] ^ b.v[i]);
    return o == 0;
  }
  // a >= modulus ?
  static CAP_HD bool geq_mod(const fe& a) {
    uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) (void)subb32(a.v[i], PR::MOD[i], br);
    return br == 0;
  }

  // -- raw 256-bit add / sub with carry -------------------------------------
  static CAP_HD uint32_t add_raw(fe& r, const fe& a, const fe& b) {
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = addc32(a.v[i], b.v[i], c);
    return c;
  }
  static CAP_HD uint32_t sub_raw(fe& r, const fe& a, const fe& b) {
    uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = subb32(a.v[i], b.v[i], br);
    return br;
  }
  static CAP_HD uint32_t sub_mod_raw(fe& r, const fe& a) {  // r = a - p, returns borrow
    uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = subb32(a.v[i], PR::MOD[i], br);
    return br;
  }

  // conditional final subtraction: a in [0, 2p) -> [0, p)
  static CAP_HD fe reduce_once(const fe& a) {
    fe t;
    uint32_t br = sub_mod_raw(t, a);
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = br ? a.v[i] : t.v[i];
    return r;
  }

  static CAP_HD fe add(const fe& a, const fe& b) {
    fe s;
    add_raw(s, a, b);  // p < 2^254: no carry out of 256 bits
    return reduce_once(s);
  }
  static CAP_HD fe sub(const fe& a, const fe& b) {
    fe d;
    uint32_t br = sub_raw(d, a, b);
    fe dp;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) dp.v[i] = addc32(d.v[i], PR::MOD[i], c);
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = br ? dp.v[i] : d.v[i];
    return r;
  }
  static CAP_HD fe neg(const fe& a) {
    if (is_zero(a)) return a;
    fe r;
    uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = subb32(PR::MOD[i], a.v[i], br);
    return r;
  }
  static CAP_HD fe dbl(const fe& a) { return add(a, a); }

  // -- Montgomery multiplication (CIOS, 32-bit limbs) -----------------------
  // r = a * b * 2^-256 mod p, inputs < p, output < p.
  static CAP_HD fe mul(const fe& a, const fe& b) {
#if defined(CAP_NOINLINE_MUL) && defined(__HIP_DEVICE_COMPILE__)
    return mul_call(a, b);
#elif !defined(__HIP_DEVICE_COMPILE__) && defined(__SIZEOF_INT128__) && !defined(CAP_HOST_MUL32)
    return mul_host64(a, b);
#else
    return mul_inline(a, b);
#endif
  }
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__SIZEOF_INT128__)
  // The host side of the library (transcripts, the prover's scalars between rounds, the verifier's pairing) multiplies on
  // 4 x 64-bit limbs: the same CIOS, the same integer (a b + m p) / 2^256 with the same m - bit for bit what mul_inline
  // returns for ANY pair of 256-bit inputs (tests/cpp/field_host_check.cpp) - at a third of its time (184 -> 53 ns).
  static inline uint64_t ninv64() {
    const uint64_t p0 = (uint64_t)PR::MOD[0] | ((uint64_t)PR::MOD[1] << 32);
    uint64_t x = (uint64_t)(0u - PR::NINV);  // p^-1 mod 2^32
    x *= 2 - p0 * x;                         // ... mod 2^64 (Newton)
    return 0 - x;
  }
  static inline fe mul_host64(const fe& a, const fe& b) {
    typedef unsigned __int128 u128;
    // (two operands far above p can carry out of 256 bits, which the two limb sizes drop at different points: such a
    // pair - never field elements - takes the 32-bit form)
    if ((a.v[7] >> 30) && (b.v[7] >> 30)) return mul_inline(a, b);  // both >= 2^254
    uint64_t x[4], y[4], p[4];
    for (int i = 0; i < 4; i++) {
      x[i] = (uint64_t)a.v[2 * i] | ((uint64_t)a.v[2 * i + 1] << 32);
      y[i] = (uint64_t)b.v[2 * i] | ((uint64_t)b.v[2 * i + 1] << 32);
      p[i] = (uint64_t)PR::MOD[2 * i] | ((uint64_t)PR::MOD[2 * i + 1] << 32);
    }
    const uint64_t ninv = ninv64();
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
      u128 c = 0;
      for (int j = 0; j < 4; j++) {
        c += (u128)x[j] * y[i] + t[j];
        t[j] = (uint64_t)c;
        c >>= 64;
      }
      c += t[4];
      t[4] = (uint64_t)c;
      t[5] = (uint64_t)(c >> 64);
      const uint64_t m = t[0] * ninv;
      c = (u128)m * p[0] + t[0];
      c >>= 64;
      for (int j = 1; j < 4; j++) {
        c += (u128)m * p[j] + t[j];
        t[j - 1] = (uint64_t)c;
        c >>= 64;
      }
      c += t[4];
      t[3] = (uint64_t)c;
      t[4] = t[5] + (uint64_t)(c >> 64);
    }
    fe res;  // (a carry into t[4] - inputs far above p - is dropped, as mul_inline drops its ninth word)
    for (int i = 0; i < 4; i++) {
      res.v[2 * i] = (uint32_t)t[i];
      res.v[2 * i + 1] = (uint32_t)(t[i] >> 32);
    }
    return reduce_once(res);
  }
#endif
#if defined(__HIPCC__) || defined(__HIP__)
  // One shared copy of the multiplication per kernel: keeps the hot loops of the curve kernels inside the
  // instruction cache (an inlined mixed addition is ~35 KB of code).
  static __device__ __noinline__ fe mul_call(const fe& a, const fe& b) { return mul_inline(a, b); }
#endif
  static CAP_HD fe mul_inline(const fe& a, const fe& b) {
    // Row i: 8 independent 32x32+32 products (v_mad_u64_u32), then one
    // add-with-carry chain (v_addc_co_u32) folds hi(r[j-1]) into lo(r[j]).
    uint32_t t[9];
#pragma unroll
    for (int i = 0; i < 9; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      uint64_t r[8];
      uint32_t c;
      // t += a * b[i]
#pragma unroll
      for (int j = 0; j < 8; j++) r[j] = (uint64_t)a.v[j] * b.v[i] + t[j];
      c = 0;
      t[0] = (uint32_t)r[0];
#pragma unroll
      for (int j = 1; j < 8; j++) t[j] = addc32((uint32_t)r[j], (uint32_t)(r[j - 1] >> 32), c);
      t[8] = addc32(t[8], (uint32_t)(r[7] >> 32), c);
      // t = (t + m * p) / 2^32   (p < 2^254: the sum fits 9 words, "no-carry" CIOS)
      uint32_t m = t[0] * PR::NINV;
#pragma unroll
      for (int j = 0; j < 8; j++) r[j] = (uint64_t)m * PR::MOD[j] + t[j];
      c = 0;  // low word of r[0] is zero by construction and is dropped
#pragma unroll
      for (int j = 1; j < 8; j++) t[j - 1] = addc32((uint32_t)r[j], (uint32_t)(r[j - 1] >> 32), c);
      t[7] = addc32(t[8], (uint32_t)(r[7] >> 32), c);
      t[8] = c;
    }
    fe res;
#pragma unroll
    for (int i = 0; i < 8; i++) res.v[i] = t[i];
    return reduce_once(res);
  }
  static CAP_HD fe sqr(const fe& a) { return mul(a, a); }

  static CAP_HD fe to_mont(const fe& a) { return mul(a, r2()); }
  static CAP_HD fe from_mont(const fe& a) {
    fe o = zero();
    o.v[0] = 1;
    return mul(a, o);
  }

  // a^e, e = 8 x u32 little endian (plain integer), a in Montgomery form
  static CAP_HD fe pow(const fe& a, const uint32_t e[8]) {
    fe r = one();
    bool started = false;
    for (int i = 7; i >= 0; i--) {
      for (int b = 31; b >= 0; b--) {
        if (started) r = sqr(r);
        if ((e[i] >> b) & 1) {
          r = started ? mul(r, a) : a;
          started = true;
        }
      }
    }
    return r;
  }
  static CAP_HD fe pow_u64(const fe& a, uint64_t e) {
    uint32_t ee[8] = {(uint32_t)e, (uint32_t)(e >> 32), 0, 0, 0, 0, 0, 0};
    return pow(a, ee);
  }
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__SIZEOF_INT128__)
  // Host inversion by the binary extended Euclidean algorithm on the plain integer (HAC 14.61 for an odd modulus): 8 us
  // against 60-70 us of the 254 squarings below.  A single proof has six of them on its critical path between the rounds
  // (the commitments' affine form, the grand product's total, zeta's powers): 2.64 -> 2.4 ms.  Same value as inv_fermat
  // (the inverse is unique, the result canonical); inv(0) = 0.
  static inline fe inv_host(const fe& a) {
    uint64_t u[4], w[4], x1[4] = {1, 0, 0, 0}, x2[4] = {0, 0, 0, 0}, p[4];
    fe v = a;
    for (int k = 0; k < 6 && geq_mod(v); k++) (void)sub_mod_raw(v, v);  // any 256-bit input: below p first
    if (is_zero(v)) return zero();
    for (int i = 0; i < 4; i++) {
      u[i] = (uint64_t)v.v[2 * i] | ((uint64_t)v.v[2 * i + 1] << 32);
      p[i] = (uint64_t)PR::MOD[2 * i] | ((uint64_t)PR::MOD[2 * i + 1] << 32);
      w[i] = p[i];
    }
    auto is_one = [](const uint64_t* z) { return z[0] == 1 && (z[1] | z[2] | z[3]) == 0; };
    auto geq = [](const uint64_t* z, const uint64_t* y) {
      for (int i = 3; i >= 0; i--)
        if (z[i] != y[i]) return z[i] > y[i];
      return true;
    };
    auto sub_in = [](uint64_t* z, const uint64_t* y) {  // z -= y, returns the borrow
      unsigned __int128 br = 0;
      for (int i = 0; i < 4; i++) {
        const unsigned __int128 d = (unsigned __int128)z[i] - y[i] - (uint64_t)br;
        z[i] = (uint64_t)d;
        br = (d >> 64) & 1;
      }
      return (uint64_t)br;
    };
    auto add_in = [](uint64_t* z, const uint64_t* y) {
      unsigned __int128 c = 0;
      for (int i = 0; i < 4; i++) {
        c += (unsigned __int128)z[i] + y[i];
        z[i] = (uint64_t)c;
        c >>= 64;
      }
    };
    auto shr1 = [](uint64_t* z) {
      for (int i = 0; i < 3; i++) z[i] = (z[i] >> 1) | (z[i + 1] << 63);
      z[3] >>= 1;
    };
    auto halve_pair = [&](uint64_t* z, uint64_t* x) {  // z even: z /= 2, x /= 2 mod p  (x < p < 2^254: x + p fits)
      while (!(z[0] & 1)) {
        shr1(z);
        if (x[0] & 1) add_in(x, p);
        shr1(x);
      }
    };
    while (!is_one(u) && !is_one(w)) {
      halve_pair(u, x1);
      halve_pair(w, x2);
      if (geq(u, w)) {
        sub_in(u, w);
        if (sub_in(x1, x2)) add_in(x1, p);
      } else {
        sub_in(w, u);
        if (sub_in(x2, x1)) add_in(x2, p);
      }
    }
    const uint64_t* r = is_one(u) ? x1 : x2;
    fe x;
    for (int i = 0; i < 4; i++) {
      x.v[2 * i] = (uint32_t)r[i];
      x.v[2 * i + 1] = (uint32_t)(r[i] >> 32);
    }
    // x = (a R)^-1 as a plain integer; the Montgomery form of a^-1 is x R^2
    return to_mont(to_mont(x));
  }
#endif
  static CAP_HD fe inv(const fe& a) {
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__SIZEOF_INT128__) && !defined(CAP_HOST_MUL32)
    return inv_host(a);
#else
    return inv_fermat(a);
#endif
  }
  // Fermat inversion a^(p-2); inv(0) = 0
  static CAP_HD fe inv_fermat(const fe& a) {
    uint32_t e[8];
    uint32_t br = 0;
    e[0] = subb32(PR::MOD[0], 2u, br);
#pragma unroll
    for (int i = 1; i < 8; i++) e[i] = subb32(PR::MOD[i], 0u, br);
    return pow(a, e);
  }
};

using Fq = Fp<FqP>;
using Fr = Fp<FrP>;

}  // namespace cap
