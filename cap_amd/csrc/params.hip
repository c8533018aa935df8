// On-disk parameter formats of the reference: bulk compressed-G1 (de)compression kernels and the
// UniversalSrs blob.  The reference notes that "deserializing these parameter files takes longer than reproducing
// them" (src/lib.rs:81-86): the cost is one square root in Fq per point (2^17 + 2 of them for the
// Aztec CRS, src/proof/mod.rs:79-93), which here is one kernel launch.
//
// Field arithmetic: the lazy 9 x 29-bit representation (field29.hpp).  HBM traffic is 32 B in / 64 B out per point;
// the kernel is bound by the ~380 multiplications of the fixed-exponent power (p + 1) / 4, not by memory.
#include <stdlib.h>

#include <vector>

#include "context.hpp"
#include "field29.hpp"
#include "launch.hpp"
#include "params.hpp"

namespace cap {
namespace params {
namespace {

constexpr int kThreads = 256;

struct Words8 {
  uint32_t w[8];
};

__device__ __forceinline__ int cmp8(const fe& a, const fe& b) {
  int r = 0;
#pragma unroll
  for (int i = 0; i < 8; i++)
    if (a.v[i] != b.v[i]) r = a.v[i] > b.v[i] ? 1 : -1;  // the most significant difference wins (written last)
  return r;
}
__device__ __forceinline__ fe sub8(const fe& a, const fe& b) {  // a - b, a >= b
  fe r;
  uint32_t br = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t t = (uint64_t)a.v[i] - b.v[i] - br;
    r.v[i] = (uint32_t)t;
    br = (uint32_t)(t >> 63);
  }
  return r;
}

// in: n x 8 little-endian words (x with the two flag bits on top); out: arkworks-form affine points
__global__ __launch_bounds__(kThreads) void g1_decompress_kernel(const uint32_t* __restrict__ in, size_t n,
                                                                 g1_affine* __restrict__ out, Words8 sqrt_exp,
                                                                 Words8 modulus, unsigned long long* first_bad) {
  using F = Fq29;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fe x, p;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    x.v[k] = in[8 * i + k];
    p.v[k] = modulus.w[k];
  }
  const uint32_t flags = x.v[7] >> 30;  // bit 31 = 0x80 of the last byte (larger root), bit 30 = 0x40 (infinity)
  x.v[7] &= 0x3fffffffu;
  g1_affine o;
#pragma unroll
  for (int k = 0; k < 8; k++) o.x.v[k] = o.y.v[k] = 0;
  bool ok = true;
  if (flags == 3) {
    ok = false;
  } else if (flags == 1) {
    uint32_t any = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) any |= x.v[k];
    ok = any == 0;
  } else if (cmp8(x, p) >= 0) {
    ok = false;
  } else {
    fl xi = F::to_mont(x);
    fe three_w;
#pragma unroll
    for (int k = 0; k < 8; k++) three_w.v[k] = k == 0 ? 3u : 0u;
    fl rhs = F::add_norm(F::mul(F::sqr(xi), xi), F::to_mont(three_w));
    fl y = F::one();
#pragma unroll 1
    for (int b = 253; b >= 0; b--) {  // the exponent is uniform across the wave: no divergence
      y = F::sqr(y);
      if ((sqrt_exp.w[b >> 5] >> (b & 31)) & 1) y = F::mul(y, rhs);
    }
    ok = F::eq(F::sqr(y), rhs);
    fe yc = F::from_mont(y);
    fe nyc = sub8(p, yc);  // y = 0 does not occur on y^2 = x^3 + 3 over this field
    const bool larger = cmp8(yc, nyc) > 0;
    if (larger != (flags == 2)) y = F::neg(y);
    o.x = F::to_ext(xi);
    o.y = F::to_ext(y);
  }
  if (!ok) atomicMin(first_bad, (unsigned long long)i);
  out[i] = o;
}

__global__ __launch_bounds__(kThreads) void g1_compress_kernel(const g1_affine* __restrict__ pts, int internal_form,
                                                               size_t n, uint32_t* __restrict__ out, Words8 modulus) {
  using F = Fq29;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  g1_affine a = pts[i];
  uint32_t any = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) any |= a.x.v[k] | a.y.v[k];
  fe xc;
  if (any == 0) {
#pragma unroll
    for (int k = 0; k < 8; k++) xc.v[k] = 0;
    xc.v[7] = 1u << 30;
  } else {
    fl xi = internal_form ? F::load(a.x) : F::from_ext(a.x);
    fl yi = internal_form ? F::load(a.y) : F::from_ext(a.y);
    xc = F::from_mont(xi);
    fe yc = F::from_mont(yi), p;
#pragma unroll
    for (int k = 0; k < 8; k++) p.v[k] = modulus.w[k];
    fe nyc = sub8(p, yc);
    if (cmp8(yc, nyc) > 0) xc.v[7] |= 1u << 31;
  }
#pragma unroll
  for (int k = 0; k < 8; k++) out[8 * i + k] = xc.v[k];
}

// canonical integers <-> arkworks' Montgomery form (32-bit saturated field: these run once per key load)
__global__ __launch_bounds__(kThreads) void fr_to_mont_kernel(fe* data, size_t n, unsigned long long* first_bad) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fe a = data[i];
  if (Fr::geq_mod(a)) atomicMin(first_bad, (unsigned long long)i);
  data[i] = Fr::to_mont(a);
}
__global__ __launch_bounds__(kThreads) void fr_from_mont_kernel(const fe* __restrict__ in, fe* __restrict__ out,
                                                                size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = Fr::from_mont(in[i]);
}

Words8 words_of(const fe& a) {
  Words8 w;
  for (int i = 0; i < 8; i++) w.w[i] = a.v[i];
  return w;
}

struct DevBuf {
  void* p = nullptr;
  ~DevBuf() {
    if (p) hipFree(p);
  }
};

}  // namespace

int decompress_g1(const uint8_t* host_bytes, size_t n, g1_affine* d_out, hipStream_t s) {
  if (n == 0) return CAPGPU_OK;
  DevBuf in, bad;
  CAP_HIP(hipMalloc(&in.p, 32 * n));
  CAP_HIP(hipMalloc(&bad.p, sizeof(unsigned long long)));
  CAP_HIP(hipMemcpyAsync(in.p, host_bytes, 32 * n, hipMemcpyHostToDevice, s));
  CAP_HIP(hipMemsetAsync(bad.p, 0xff, sizeof(unsigned long long), s));
  Words8 e;
  fq_sqrt_exponent(e.w);
  launch("g1_decompress", g1_decompress_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, s,
         (const uint32_t*)in.p, n, d_out, e, words_of(fq_modulus()), (unsigned long long*)bad.p);
  unsigned long long first = 0;
  CAP_HIP(hipMemcpyAsync(&first, bad.p, sizeof(first), hipMemcpyDeviceToHost, s));
  CAP_HIP(hipStreamSynchronize(s));
  if (first != ~0ull) {
    set_error("capgpu: compressed G1 point %llu is not a valid encoding (x >= p, x not on the curve, or bad flags)",
              first);
    return CAPGPU_ERR_SERIALIZATION;
  }
  return CAPGPU_OK;
}

int compress_g1(const g1_affine* d_pts, int internal_form, size_t n, uint8_t* host_out, hipStream_t s) {
  if (n == 0) return CAPGPU_OK;
  DevBuf out;
  CAP_HIP(hipMalloc(&out.p, 32 * n));
  launch("g1_compress", g1_compress_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, s,
         d_pts, internal_form, n, (uint32_t*)out.p, words_of(fq_modulus()));
  CAP_HIP(hipMemcpyAsync(host_out, out.p, 32 * n, hipMemcpyDeviceToHost, s));
  CAP_HIP(hipStreamSynchronize(s));
  return CAPGPU_OK;
}

int fr_bytes_to_mont(const uint8_t* host_bytes, size_t n, fe* d_out, hipStream_t s) {
  if (n == 0) return CAPGPU_OK;
  DevBuf bad;
  CAP_HIP(hipMalloc(&bad.p, sizeof(unsigned long long)));
  CAP_HIP(hipMemsetAsync(bad.p, 0xff, sizeof(unsigned long long), s));
  CAP_HIP(hipMemcpyAsync(d_out, host_bytes, 32 * n, hipMemcpyHostToDevice, s));
  launch("fr_to_mont", fr_to_mont_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, d_out,
         n, (unsigned long long*)bad.p);
  unsigned long long first = 0;
  CAP_HIP(hipMemcpyAsync(&first, bad.p, sizeof(first), hipMemcpyDeviceToHost, s));
  CAP_HIP(hipStreamSynchronize(s));
  if (first != ~0ull) {
    set_error("capgpu: scalar %llu of a serialized vector is not canonical (>= r)", first);
    return CAPGPU_ERR_SERIALIZATION;
  }
  return CAPGPU_OK;
}

int fr_mont_to_bytes(const fe* d_in, size_t n, uint8_t* host_out, hipStream_t s) {
  if (n == 0) return CAPGPU_OK;
  DevBuf out;
  CAP_HIP(hipMalloc(&out.p, 32 * n));
  launch("fr_from_mont", fr_from_mont_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, s,
         d_in, (fe*)out.p, n);
  CAP_HIP(hipMemcpyAsync(host_out, out.p, 32 * n, hipMemcpyDeviceToHost, s));
  CAP_HIP(hipStreamSynchronize(s));
  return CAPGPU_OK;
}

}  // namespace params
}  // namespace cap

using namespace cap;
using namespace cap::params;

extern "C" {

int capgpu_g1_decompress(const uint8_t* in, size_t n, uint64_t* out_xy) {
  CAP_CHECK_INIT();
  if (n && (!in || !out_xy)) {
    set_error("capgpu_g1_decompress: bad argument");
    return CAPGPU_ERR_INVALID_ARG;
  }
  if (n == 0) return CAPGPU_OK;
  Context& c = ctx();
  Entry lk(c);
  DevTmp<g1_affine> d;
  CAP_HIP(d.alloc(n));
  int rc = decompress_g1(in, n, d, c.stream);
  if (rc == CAPGPU_OK) {
    hipError_t e = hipMemcpyAsync(out_xy, d, sizeof(g1_affine) * n, hipMemcpyDeviceToHost, c.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c.stream);
    if (e != hipSuccess) rc = hip_fail(e, "copy of decompressed points");
  }
  return rc;
}

int capgpu_g1_compress(const uint64_t* xy, size_t n, uint8_t* out) {
  CAP_CHECK_INIT();
  if (n && (!xy || !out)) {
    set_error("capgpu_g1_compress: bad argument");
    return CAPGPU_ERR_INVALID_ARG;
  }
  if (n == 0) return CAPGPU_OK;
  Context& c = ctx();
  Entry lk(c);
  DevTmp<g1_affine> d;
  CAP_HIP(d.alloc(n));
  hipError_t e = hipMemcpyAsync(d, xy, sizeof(g1_affine) * n, hipMemcpyHostToDevice, c.stream);
  return e == hipSuccess ? compress_g1(d, 0, n, out, c.stream) : hip_fail(e, "upload of points");
}

// UniversalSrs<Bn254> = ark_poly_commit::kzg10::UniversalParams: powers_of_g: Vec<G1>, powers_of_gamma_g:
// BTreeMap<usize, G1>, h: G2, beta_h: G2, neg_powers_of_h: BTreeMap<usize, G2>
int capgpu_srs_deserialize(const uint8_t* bytes, size_t len, size_t max_degree, uint64_t* handle_out, uint64_t h_out[16],
                           uint64_t beta_h_out[16], size_t* consumed_out) {
  CAP_CHECK_INIT();
  if (!bytes || !handle_out) {
    set_error("capgpu_srs_deserialize: bad argument");
    return CAPGPU_ERR_INVALID_ARG;
  }
  Context& c = ctx();
  Entry lk(c);
  Reader rd(bytes, len);
  uint64_t n_g = 0, n_gamma = 0, n_neg = 0;
  if (!rd.count(32, &n_g)) goto truncated;
  {
    const uint8_t* g_bytes = rd.take(32 * n_g);
    if (!rd.count(40, &n_gamma)) goto truncated;
    std::vector<uint8_t> gamma(32 * n_gamma);
    std::vector<uint64_t> gamma_deg(n_gamma);
    for (uint64_t i = 0; i < n_gamma; i++) {
      gamma_deg[i] = rd.u64();  // the degree this power belongs to
      const uint8_t* v = rd.take(32);
      if (!v) goto truncated;
      memcpy(&gamma[32 * i], v, 32);
    }
    const uint8_t* hb = rd.take(64);
    const uint8_t* bhb = rd.take(64);
    if (!hb || !bhb) goto truncated;
    g2_affine h, beta_h;
    if (!g2_decompress(hb, &h) || !g2_decompress(bhb, &beta_h)) {
      set_error("capgpu_srs_deserialize: h / beta_h is not a valid compressed G2 point of the prime-order subgroup");
      return CAPGPU_ERR_SERIALIZATION;
    }
    if (!rd.count(72, &n_neg)) goto truncated;
    std::vector<uint8_t> neg_h;
    for (uint64_t i = 0; i < n_neg; i++) {
      const uint8_t* entry = bytes + rd.pos;
      rd.u64();
      const uint8_t* v = rd.take(64);
      g2_affine q;
      if (!v) goto truncated;
      neg_h.insert(neg_h.end(), entry, entry + 72);
      if (!g2_decompress(v, &q)) {
        set_error("capgpu_srs_deserialize: neg_powers_of_h[%llu] is not a valid compressed G2 point",
                  (unsigned long long)i);
        return CAPGPU_ERR_SERIALIZATION;
      }
    }
    if (n_g == 0) {
      set_error("capgpu_srs_deserialize: the blob holds no powers of g");
      return CAPGPU_ERR_SERIALIZATION;
    }
    // every point is validated like ark-serialize does; only the requested prefix stays resident
    size_t keep = max_degree ? std::min<size_t>(n_g, max_degree + 1) : (size_t)n_g;
    DevTmp<g1_affine> d;
    CAP_HIP(d.alloc(std::max<size_t>(n_g, n_gamma)));
    int rc = n_gamma ? decompress_g1(gamma.data(), n_gamma, d, c.stream) : CAPGPU_OK;
    if (rc == CAPGPU_OK) rc = decompress_g1(g_bytes, n_g, d, c.stream);
    if (rc == CAPGPU_OK) rc = register_srs(d, keep, handle_out);
    if (rc) return rc;
    // the hiding powers and the negative powers of h are not used by this (non-hiding) prover, but a reference-side
    // consumer of a re-stored file indexes them (jf-plonk `trim`): they stay with the handle, validated, verbatim
    if (SrsEntry* e = find_srs_entry(*handle_out)) {
      if (keep < n_g) {
        // a trimmed load keeps the hiding powers a setup of that degree would hold (degrees 0 .. max_degree + 1), so that
        // a re-stored file is consistent
        std::vector<uint64_t> deg2;
        std::vector<uint8_t> pts2;
        for (uint64_t i = 0; i < n_gamma; i++)
          if (gamma_deg[i] <= (uint64_t)keep) {
            deg2.push_back(gamma_deg[i]);
            pts2.insert(pts2.end(), gamma.begin() + 32 * i, gamma.begin() + 32 * i + 32);
          }
        gamma_deg.swap(deg2);
        gamma.swap(pts2);
      }
      e->gamma_deg = std::move(gamma_deg);
      e->gamma_pts = std::move(gamma);
      e->neg_h = std::move(neg_h);
    }
    if (h_out) g2_to_words(h, h_out);
    if (beta_h_out) g2_to_words(beta_h, beta_h_out);
    if (consumed_out) *consumed_out = rd.pos;
    return CAPGPU_OK;
  }
truncated:
  set_error("capgpu_srs_deserialize: unexpected end of input at byte %zu of %zu", rd.pos, len);
  return CAPGPU_ERR_SERIALIZATION;
}

int capgpu_srs_serialize(uint64_t handle, const uint64_t h[16], const uint64_t beta_h[16], uint8_t* out, size_t cap,
                         size_t* len_out) {
  CAP_CHECK_INIT();
  Context& c = ctx();
  Entry lk(c);
  const MsmBases* B = nullptr;
  int rc = find_srs(handle, &B);
  if (rc) return rc;
  if (!h || !beta_h || !len_out) {
    set_error("capgpu_srs_serialize: bad argument");
    return CAPGPU_ERR_INVALID_ARG;
  }
  const SrsEntry* E = find_srs_entry(handle);
  const size_t n_gamma = E ? E->gamma_pts.size() / 32 : 0, n_neg = E ? E->neg_h.size() / 72 : 0;
  const size_t need = 8 + 32 * B->n + 8 + 40 * n_gamma + 64 + 64 + 8 + 72 * n_neg;
  *len_out = need;
  if (!out) return CAPGPU_OK;  // size query
  if (cap < need) {
    set_error("capgpu_srs_serialize: buffer of %zu bytes, %zu needed", cap, need);
    return CAPGPU_ERR_INVALID_ARG;
  }
  uint64_t n64 = B->n;
  memcpy(out, &n64, 8);
  if ((rc = compress_g1(B->ext, 1, B->n, out + 8, c.stream))) return rc;  // window 0 of the table = the bases
  uint8_t* p = out + 8 + 32 * B->n;
  // powers_of_gamma_g / neg_powers_of_h: what the blob this handle was loaded from held (none for an SRS generated
  // here: commitments are non-hiding, see the limitation note in include/capgpu.h)
  n64 = n_gamma;
  memcpy(p, &n64, 8);
  p += 8;
  for (size_t i = 0; i < n_gamma; i++, p += 40) {
    memcpy(p, &E->gamma_deg[i], 8);
    memcpy(p + 8, &E->gamma_pts[32 * i], 32);
  }
  g2_compress(g2_from_words(h), p);
  g2_compress(g2_from_words(beta_h), p + 64);
  n64 = n_neg;
  memcpy(p + 128, &n64, 8);
  if (n_neg) memcpy(p + 136, E->neg_h.data(), 72 * n_neg);
  return CAPGPU_OK;
}

// ---- Proof bytes -> capgpu_proof: host only --------------------------------------------------------------------
// The inverse of capgpu_proof_serialize: what `Proof::deserialize` does when a TransferNote / MintNote / FreezeNote
// arrives as bytes (src/transfer.rs:54-66).  Every encoding ark-serialize rejects is rejected: wrong vector lengths,
// x >= p, x not on the curve, both flag bits, scalars >= r, a plookup proof.
int capgpu_proof_deserialize(const uint8_t* bytes, size_t len, capgpu_proof* proof_out, size_t* consumed_out) {
  if (!bytes || !proof_out) {
    set_error("capgpu_proof_deserialize: bad argument");
    return CAPGPU_ERR_INVALID_ARG;
  }
  Reader rd(bytes, len);
  auto fail = [&](const char* why) {
    set_error("capgpu_proof_deserialize: %s (byte %zu of %zu)", why, rd.pos, len);
    return CAPGPU_ERR_SERIALIZATION;
  };
  memset(proof_out, 0, sizeof(*proof_out));
  auto g1 = [&](uint64_t out[8]) {
    const uint8_t* b = rd.take(32);
    g1_affine p;
    if (!b || !g1_decompress_host(b, &p)) return false;
    affine_to_words(p, out);
    return true;
  };
  auto fr = [&](uint64_t out[4]) {
    const uint8_t* b = rd.take(32);
    if (!b) return false;
    fe v;
    memcpy(v.v, b, 32);
    if (cmp_words(v, fr_modulus()) >= 0) return false;
    fe_to_words(Fr::to_mont(v), out);
    return true;
  };
  uint64_t cnt = 0;
  if (!rd.count(32, &cnt)) return fail("unexpected end of input");
  if (cnt != kNumWires) return fail("wires_poly_comms: a TurboPlonk proof has 5 of them");
  for (int i = 0; i < kNumWires; i++)
    if (!g1(proof_out->wires_poly_comms[i])) return fail("wires_poly_comms: invalid compressed G1 point");
  if (!g1(proof_out->prod_perm_poly_comm)) return fail("prod_perm_poly_comm: invalid compressed G1 point");
  if (!rd.count(32, &cnt)) return fail("unexpected end of input");
  if (cnt != kNumWires) return fail("split_quot_poly_comms: a TurboPlonk proof has 5 of them");
  for (int i = 0; i < kNumWires; i++)
    if (!g1(proof_out->split_quot_poly_comms[i])) return fail("split_quot_poly_comms: invalid compressed G1 point");
  if (!g1(proof_out->opening_proof)) return fail("opening_proof: invalid compressed G1 point");
  if (!g1(proof_out->shifted_opening_proof)) return fail("shifted_opening_proof: invalid compressed G1 point");
  if (!rd.count(32, &cnt)) return fail("unexpected end of input");
  if (cnt != kNumWires) return fail("wires_evals: 5 expected");
  for (int i = 0; i < kNumWires; i++)
    if (!fr(proof_out->wires_evals[i])) return fail("wires_evals: scalar missing or not canonical");
  if (!rd.count(32, &cnt)) return fail("unexpected end of input");
  if (cnt != kNumWires - 1) return fail("wire_sigma_evals: 4 expected");
  for (int i = 0; i < kNumWires - 1; i++)
    if (!fr(proof_out->wire_sigma_evals[i])) return fail("wire_sigma_evals: scalar missing or not canonical");
  if (!fr(proof_out->perm_next_eval)) return fail("perm_next_eval: scalar missing or not canonical");
  const uint8_t* tag = rd.take(1);
  if (!tag) return fail("unexpected end of input");
  if (*tag > 1) return fail("invalid Option tag");
  if (*tag) return fail("plookup proofs are not supported");
  if (consumed_out) *consumed_out = rd.pos;
  return CAPGPU_OK;
}

// ---- VerifyingKey blob: host only ---------------------------------------------------------------------------
int capgpu_plonk_vk_serialize(const capgpu_verifying_key* vk, const uint64_t g[8], const uint64_t gamma_g[8],
                              const uint64_t h[16], const uint64_t beta_h[16], uint8_t* out, size_t cap,
                              size_t* len_out) {
  if (!vk || !g || !h || !beta_h || !len_out) {
    set_error("capgpu_plonk_vk_serialize: bad argument");
    return CAPGPU_ERR_INVALID_ARG;
  }
  OpenKey ok;
  ok.g = g1_from_words(g);
  ok.gamma_g.x = ok.gamma_g.y = Fq::zero();
  if (gamma_g) ok.gamma_g = g1_from_words(gamma_g);
  ok.h = g2_from_words(h);
  ok.beta_h = g2_from_words(beta_h);
  Writer w;
  write_vk(w, *vk, ok);
  *len_out = w.buf.size();
  if (!out) return CAPGPU_OK;
  if (cap < w.buf.size()) {
    set_error("capgpu_plonk_vk_serialize: buffer of %zu bytes, %zu needed", cap, w.buf.size());
    return CAPGPU_ERR_INVALID_ARG;
  }
  memcpy(out, w.buf.data(), w.buf.size());
  return CAPGPU_OK;
}

int capgpu_plonk_vk_deserialize(const uint8_t* bytes, size_t len, capgpu_verifying_key* vk_out, uint64_t g_out[8],
                                uint64_t gamma_g_out[8], uint64_t h_out[16], uint64_t beta_h_out[16],
                                size_t* consumed_out) {
  if (!bytes || !vk_out) {
    set_error("capgpu_plonk_vk_deserialize: bad argument");
    return CAPGPU_ERR_INVALID_ARG;
  }
  Reader rd(bytes, len);
  OpenKey ok;
  const char* why = read_vk(rd, vk_out, &ok);
  if (why) {
    set_error("capgpu_plonk_vk_deserialize: %s (byte %zu of %zu)", why, rd.pos, len);
    return CAPGPU_ERR_SERIALIZATION;
  }
  if (g_out) affine_to_words(ok.g, g_out);
  if (gamma_g_out) affine_to_words(ok.gamma_g, gamma_g_out);
  if (h_out) g2_to_words(ok.h, h_out);
  if (beta_h_out) g2_to_words(ok.beta_h, beta_h_out);
  if (consumed_out) *consumed_out = rd.pos;
  return CAPGPU_OK;
}

}  // extern "C"
