// Lagrange-form commit key: [L_j(tau)] G for the n-th roots of unity, from the monomial SRS [tau^i] G.
//
// jf-plonk commits to a wire polynomial through its COEFFICIENTS (KZG10::commit under src/proof/transfer.rs:181-186), so
// round 1 of the reference is five inverse FFTs followed by five MSMs on full-width scalars.  The commitment is the group
// element  sum_j w_j [L_j(tau)] G + b0 [tau^n - 1] G + b1 [tau^(n+1) - tau] G  whichever way it is computed, and the witness
// VALUES w_j of a CAP circuit are mostly zeros, booleans and small range-check limbs (src/circuit/transfer.rs:53-193):
// as MSM scalars they have one or no non-zero digit where a coefficient has seventeen.  With
//     L_j(tau) = (1/n) sum_i omega^(-ij) tau^i
// the Lagrange points are the inverse DFT of the SRS points taken in the GROUP: a radix-2 transform whose butterflies are
// (A + [w] B, A - [w] B) with a 254-bit scalar multiplication per butterfly.  That is ~n/2 (log n + 1) scalar
// multiplications of ~380 point operations each - 0.25 M of them at n = 2^15 - once per (SRS, domain size); the result is
// an ordinary window table (msm.hpp) of n + 3 points: L_0 .. L_(n-1), then the blinding points Z_e = [tau^(n+e) - tau^e] G,
// e = 0, 1, 2 (two blinders per wire polynomial, three for the permutation product).  Same group elements, same proof
// bytes (tests/test_gpu_lagrange.py).
//
// One thread per butterfly; the launches are one wave per CU (n / 2 threads), every lane a chain of dependent point
// operations: the row-wise multiplication schedule, like the MSM's one-wavefront finishing kernels.
#define CAP_FL_SCHED 0
#include "lagrange.hpp"

#include <vector>

#include "curve29.hpp"
#include "launch.hpp"
#include "ntt.hpp"

namespace cap {
namespace {

using F = Fq29;
constexpr uint32_t kLagBlind = 3;  // blinding points behind the n Lagrange points

__device__ __forceinline__ g1x neg_pt(const g1x& p) {
  g1x r = p;
  r.y = F::weak_reduce(F::neg(p.y));
  return r;
}

// [k] B, k a canonical 256-bit integer: two bits per step from the top (B, 2B, 3B held in registers)
__device__ __noinline__ g1x scalar_mul(const g1x& b, const fe& k) {
  int top = -1;
  for (int i = 7; i >= 0 && top < 0; i--)
    if (k.v[i]) top = 32 * i + (31 - __clz(k.v[i]));
  if (top < 0 || G1L::is_inf(b)) return G1L::inf();
  const g1x b2 = G1L::dbl(b);
  const g1x b3 = G1L::add(b2, b);
  g1x acc = G1L::inf();
  for (int pos = top | 1; pos >= 1; pos -= 2) {  // digit = bits pos, pos - 1
    acc = G1L::dbl(G1L::dbl(acc));
    const uint32_t d = (k.v[(pos - 1) >> 5] >> ((pos - 1) & 31)) & 3u;
    if (d) acc = G1L::add(acc, d == 1 ? b : (d == 2 ? b2 : b3));
  }
  return acc;
}

// a[bitrev(i)] = P_i (the window-0 rows of the SRS table: internal form, canonical)
__global__ __launch_bounds__(256) void lag_load(g1_xyzz* __restrict__ a, const g1_affine* __restrict__ srs, uint32_t n,
                                                uint32_t log_n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t r = log_n ? __brev(i) >> (32 - log_n) : 0;
  a[r] = G1L::store(G1L::from_affine(G1L::load(srs[i])));
}

// one decimation-in-time stage on blocks of 2 * half points: (A, B) -> (A + [w] B, A - [w] B), w = omega_n^-(j * step)
__global__ __launch_bounds__(64) void lag_stage(g1_xyzz* __restrict__ a, uint32_t n, uint32_t half,
                                                const fe* __restrict__ tw, uint32_t step) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n / 2) return;
  const uint32_t j = t & (half - 1), i0 = ((t - j) << 1) + j, i1 = i0 + half;
  const g1x A = G1L::load(a[i0]);
  g1x B = G1L::load(a[i1]);
  if (j) B = scalar_mul(B, tw[(size_t)j * step]);
  a[i0] = G1L::store(G1L::add(A, B));
  a[i1] = G1L::store(G1L::add(A, neg_pt(B)));
}

// out[j] = [1/n] a[j] in arkworks' affine form (what msm_precompute takes), j < n; out[n .. n + 2] = the blinding points
__global__ __launch_bounds__(64) void lag_finish(const g1_xyzz* __restrict__ a, uint32_t n, fe n_inv,
                                                 const g1_affine* __restrict__ srs, g1_affine* __restrict__ out) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n + kLagBlind) return;
  g1x p;
  if (j < n) {
    p = scalar_mul(G1L::load(a[j]), n_inv);
  } else {  // [tau^(n + e)] G - [tau^e] G, e = j - n
    p = G1L::add_mixed(G1L::from_affine(G1L::load(srs[j])), G1L::load(srs[j - n]), true);
  }
  g1_affine o;
  if (G1L::is_inf(p)) {
    o.x = Fq::zero();
    o.y = Fq::zero();
  } else {
    const g1a q = G1L::to_affine(p);
    o.x = F::to_ext(q.x);
    o.y = F::to_ext(q.y);
  }
  out[j] = o;
}

}  // namespace

int lagrange_build(const MsmBases& srs, uint32_t log_n, MsmBases* out, hipStream_t stream) {
  const size_t n = (size_t)1 << log_n;
  if (log_n > 26 || srs.n < n + kLagBlind || !srs.ext) return (int)hipErrorInvalidValue;
  // omega_n^-k, k < n / 2, and 1/n as plain integers (the scalars of the butterflies)
  std::vector<fe> tw(n / 2 ? n / 2 : 1);
  const fe w_inv = Fr::inv(ntt_root_of_unity(log_n));
  fe x = Fr::one();
  for (size_t k = 0; k < n / 2; k++) {
    tw[k] = Fr::from_mont(x);
    x = Fr::mul(x, w_inv);
  }
  fe nn = Fr::zero();
  nn.v[0] = (uint32_t)n;
  const fe n_inv = Fr::from_mont(Fr::inv(Fr::to_mont(nn)));
  g1_xyzz* d_a = nullptr;
  fe* d_tw = nullptr;
  g1_affine* d_aff = nullptr;
  auto cleanup = [&] {
    if (d_a) hipFree(d_a);
    if (d_tw) hipFree(d_tw);
    if (d_aff) hipFree(d_aff);
  };
  hipError_t e = hipMalloc(&d_a, sizeof(g1_xyzz) * n);
  if (e == hipSuccess) e = hipMalloc(&d_tw, sizeof(fe) * tw.size());
  if (e == hipSuccess) e = hipMalloc(&d_aff, sizeof(g1_affine) * (n + kLagBlind));
  if (e == hipSuccess) e = hipMemcpyAsync(d_tw, tw.data(), sizeof(fe) * tw.size(), hipMemcpyHostToDevice, stream);
  if (e != hipSuccess) {
    cleanup();
    return (int)e;
  }
  launch("lag_load", lag_load, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_a, (const g1_affine*)srs.ext,
         (uint32_t)n, log_n);
  for (uint32_t half = 1; half < n; half <<= 1)
    launch("lag_stage", lag_stage, dim3((unsigned)((n / 2 + 63) / 64)), dim3(64), 0, stream, d_a, (uint32_t)n, half,
           (const fe*)d_tw, (uint32_t)(n / (2 * half)));
  launch("lag_finish", lag_finish, dim3((unsigned)((n + kLagBlind + 63) / 64)), dim3(64), 0, stream, (const g1_xyzz*)d_a, (uint32_t)n,
         n_inv, (const g1_affine*)srs.ext, d_aff);
  int rc = msm_precompute(out, d_aff, n + kLagBlind, msm_choose_window(n + kLagBlind), stream);
  e = hipStreamSynchronize(stream);  // the table may be used from another context's stream next; the temporaries go
  cleanup();
  if (rc == 0 && e != hipSuccess) rc = (int)e;
  if (rc) msm_free_bases(out);
  return rc;
}

}  // namespace cap
