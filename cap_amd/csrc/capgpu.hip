// C-ABI entry points of libcapgpu.so (include/capgpu.h): lifecycle, device
// memory plumbing, SRS management, MSM and NTT.  The PLONK entry points live in
// plonk.hip.  There is no CPU fallback anywhere in this library.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "context.hpp"
#include "curve29.hpp"
#include "launch.hpp"

namespace cap {

static thread_local char g_err[512] = "";

Context& ctx() {
  static Context c;
  return c;
}
Profiler& profiler() {
  static Profiler p;
  return p;
}
LaunchError& launch_error() {
  static LaunchError e;
  return e;
}
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
int hip_fail(hipError_t e, const char* what) {
  set_error("capgpu: HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
  (void)hipGetLastError();
  return e == hipErrorOutOfMemory ? CAPGPU_ERR_OOM : CAPGPU_ERR_HIP;
}

int take_launch_error() {
  LaunchError& le = launch_error();
  if (le.code == hipSuccess) return CAPGPU_OK;
  set_error("capgpu: launch of kernel %s failed: HIP error %d (%s)", le.kernel ? le.kernel : "?", (int)le.code,
            hipGetErrorString(le.code));
  le = LaunchError{};
  return CAPGPU_ERR_HIP;
}

int scratch_reserve(Scratch& s, size_t bytes) {
  if (bytes <= s.cap) return CAPGPU_OK;
  Context& c = ctx();
  if (s.p) {
    CAP_HIP(hipStreamSynchronize(c.stream));
    CAP_HIP(hipFree(s.p));
    s.p = nullptr;
    s.cap = 0;
  }
  size_t want = bytes + bytes / 4;
  CAP_HIP(hipMalloc(&s.p, want));
  s.cap = want;
  return CAPGPU_OK;
}

int get_domain(uint32_t log_n, const NttDomain** out) {
  Context& c = ctx();
  auto it = c.domains.find(log_n);
  if (it == c.domains.end()) {
    NttDomain d;
    int rc = ntt_build_domain(&d, log_n, c.stream);
    if (rc) return hip_fail((hipError_t)rc, "ntt_build_domain");
    it = c.domains.emplace(log_n, d).first;
  }
  *out = &it->second;
  return CAPGPU_OK;
}

int get_domain3(uint32_t log_m, const Ntt3Domain** out) {
  Context& c = ctx();
  auto it = c.domains3.find(log_m);
  if (it == c.domains3.end()) {
    Ntt3Domain d;
    int rc = ntt3_build_domain(&d, log_m, c.stream);
    if (rc) return hip_fail((hipError_t)rc, "ntt3_build_domain");
    it = c.domains3.emplace(log_m, d).first;
  }
  *out = &it->second;
  return CAPGPU_OK;
}

namespace {

// ---- SRS generation kernels ---------------------------------------------------------------
// scalar_i = mode 0: table[i] (Montgomery Fr, e.g. tau^i); mode 1: a + i*b.  out[i] = [scalar_i] G.
__global__ __launch_bounds__(256) void srs_fixed_base_kernel(g1_affine* __restrict__ out, size_t n, int mode,
                                                             const fe* __restrict__ table, fe a_mont, fe b_mont) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fe s;
  if (mode == 0) {
    s = table[i];
  } else {
    fe idx = Fr::zero();
    idx.v[0] = (uint32_t)i;
    idx.v[1] = (uint32_t)((uint64_t)i >> 32);
    s = Fr::add(a_mont, Fr::mul(Fr::to_mont(idx), b_mont));
  }
  s = Fr::from_mont(s);
  g1_affine g;
  g.x = Fq::one();
  g.y = Fq::dbl(Fq::one());
  g1_xyzz acc = G1::inf();
  bool started = false;
  for (int l = 7; l >= 0; l--) {
    for (int b = 31; b >= 0; b--) {
      if (started) acc = G1::dbl(acc);
      if ((s.v[l] >> b) & 1) {
        acc = G1::add_mixed(acc, g);
        started = true;
      }
    }
  }
  out[i] = G1::to_affine(acc);
}

__global__ void fr_powers_kernel(fe* out, size_t n, const fe* __restrict__ pw) {
  size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  fe r = Fr::one();
  bool started = false;
  for (int b = 0; (e >> b) != 0; b++) {
    if ((e >> b) & 1) {
      r = started ? Fr::mul(r, pw[b]) : pw[b];
      started = true;
    }
  }
  out[e] = r;
}

__global__ void fq_to_mont_kernel(fe* data, size_t count) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) data[i] = Fq::to_mont(data[i]);
}

// sum of n Jacobian points (one wavefront; n is tiny: the G partial sums of a sharded MSM)
__global__ __launch_bounds__(64) void g1_sum_kernel(const g1_jac* __restrict__ in, size_t n, g1_jac* __restrict__ out) {
  g1x acc = G1L::inf();
  for (size_t i = threadIdx.x; i < n; i += 64) {
    g1_jac p = in[i];
    g1x q = G1L::inf();
    if (!Fq::is_zero(p.z)) {
      fl z = Fq29::from_ext(p.z);
      q.zz = Fq29::sqr(z);
      q.zzz = Fq29::mul(q.zz, z);
      q.x = Fq29::weak_reduce(Fq29::from_ext(p.x));
      q.y = Fq29::weak_reduce(Fq29::from_ext(p.y));
    }
    acc = G1L::add(acc, q);
  }
  for (int d = 32; d >= 1; d >>= 1) {
    g1x o;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      o.x.v[i] = __shfl_down(acc.x.v[i], d);
      o.y.v[i] = __shfl_down(acc.y.v[i], d);
      o.zz.v[i] = __shfl_down(acc.zz.v[i], d);
      o.zzz.v[i] = __shfl_down(acc.zzz.v[i], d);
    }
    acc = G1L::add(acc, o);
  }
  if (threadIdx.x == 0) *out = G1L::to_jac_ext(acc);
}

// Issue rate of v_mad_u64_u32, the instruction 84 % of a field multiplication consists of: 8 independent accumulate
// chains per lane, 8 waves per SIMD.  Prices the multiplication ceiling of the chip (bench.py: alu_roofline.peak).
__global__ __launch_bounds__(256) void ubench_mad_kernel(uint64_t* io, int iters) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t acc[8];
  const uint32_t a = (uint32_t)io[i] | 1u, b = (uint32_t)(io[i] >> 32) | 1u;
#pragma unroll
  for (int c = 0; c < 8; c++) acc[c] = io[i] + c;
  for (int k = 0; k < iters; k++) {
#pragma unroll
    for (int rep = 0; rep < 8; rep++)
#pragma unroll
      for (int c = 0; c < 8; c++) {
        uint64_t r, carry;
        asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(carry) : "v"(a), "v"(b + c), "v"(acc[c]));
        acc[c] = r;
      }
  }
  uint64_t x = 0;
#pragma unroll
  for (int c = 0; c < 8; c++) x ^= acc[c];
  io[i] = x;
}

fe fe_from_u64x4(const uint64_t v[4]) {
  fe r;
  for (int i = 0; i < 4; i++) {
    r.v[2 * i] = (uint32_t)v[i];
    r.v[2 * i + 1] = (uint32_t)(v[i] >> 32);
  }
  return r;
}

}  // namespace

int register_srs(g1_affine* d_bases, size_t n, uint64_t* handle_out) {
  Context& c = ctx();
  SrsEntry e;
  uint32_t cw = msm_choose_window(n);
  int rc = msm_precompute(&e.bases, d_bases, n, cw, c.stream);
  if (rc) return hip_fail((hipError_t)rc, "msm_precompute");
  CAP_HIP(hipStreamSynchronize(c.stream));
  uint64_t h = c.next_handle++;
  c.srs[h] = e;
  *handle_out = h;
  return CAPGPU_OK;
}

SrsEntry* find_srs_entry(uint64_t h) {
  Context& c = ctx();
  auto it = c.srs.find(h);
  return it == c.srs.end() ? nullptr : &it->second;
}

int find_srs(uint64_t h, const MsmBases** out) {
  Context& c = ctx();
  auto it = c.srs.find(h);
  if (it == c.srs.end()) {
    set_error("capgpu: unknown SRS handle %llu", (unsigned long long)h);
    return CAPGPU_ERR_BAD_HANDLE;
  }
  *out = &it->second.bases;
  return CAPGPU_OK;
}

}  // namespace cap

using namespace cap;

extern "C" {

const char* capgpu_last_error(void) { return g_err; }
const char* capgpu_version(void) { return "capgpu 0.1.0 (gfx950)"; }

int capgpu_init(const int* device_ids, int n_devices) {
  Context& c = ctx();
  std::lock_guard<std::recursive_mutex> lk(c.mu);
  if (c.initialised) return CAPGPU_OK;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0) {
    set_error("capgpu: no HIP device visible (%s); this library has no CPU fallback",
              e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    (void)hipGetLastError();
    return CAPGPU_ERR_NO_DEVICE;
  }
  int dev = (device_ids && n_devices > 0) ? device_ids[0] : 0;
  if (dev < 0 || dev >= count) {
    set_error("capgpu: device id %d out of range (0..%d)", dev, count - 1);
    return CAPGPU_ERR_INVALID_ARG;
  }
  CAP_HIP(hipSetDevice(dev));
  hipDeviceProp_t prop;
  CAP_HIP(hipGetDeviceProperties(&prop, dev));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    set_error("capgpu: device %d is %s; kernels are built for gfx950 (MI355X) only", dev, prop.gcnArchName);
    return CAPGPU_ERR_NO_DEVICE;
  }
  c.device = dev;
  CAP_HIP(hipStreamCreateWithFlags(&c.own_stream, hipStreamNonBlocking));
  c.stream = c.own_stream;
  int rc = ntt_build_small_tables(&c.small, c.stream);
  if (rc) return hip_fail((hipError_t)rc, "ntt_build_small_tables");
  c.initialised = true;
  return CAPGPU_OK;
}

void capgpu_shutdown(void) {
  Context& c = ctx();
  std::lock_guard<std::recursive_mutex> lk(c.mu);
  if (!c.initialised) return;
  (void)capgpu_comm_destroy();
  hipDeviceSynchronize();
  c.keys.clear();
  for (auto& kv : c.srs) msm_free_bases(&kv.second.bases);
  c.srs.clear();
  for (auto& kv : c.domains) ntt_free_domain(&kv.second);
  c.domains.clear();
  for (auto& kv : c.domains3) ntt3_free_domain(&kv.second);
  c.domains3.clear();
  ntt_free_small_tables(&c.small);
  for (Scratch* s : {&c.ntt_scratch, &c.msm_ws, &c.stage_a, &c.stage_b}) {
    if (s->p) hipFree(s->p);
    s->p = nullptr;
    s->cap = 0;
  }
  if (c.own_stream) hipStreamDestroy(c.own_stream);
  if (c.copy_stream) hipStreamDestroy(c.copy_stream);
  c.own_stream = c.stream = c.copy_stream = nullptr;
  c.initialised = false;
}

int capgpu_device_info(char* name_out, int* cu_count_out, uint64_t* hbm_bytes_out) {
  CAP_CHECK_INIT();
  hipDeviceProp_t prop;
  CAP_HIP(hipGetDeviceProperties(&prop, ctx().device));
  if (name_out) {
    snprintf(name_out, 256, "%s (%s)", prop.name, prop.gcnArchName);
  }
  if (cu_count_out) *cu_count_out = prop.multiProcessorCount;
  if (hbm_bytes_out) *hbm_bytes_out = (uint64_t)prop.totalGlobalMem;
  return CAPGPU_OK;
}

int capgpu_malloc(void** dev_ptr_out, size_t bytes) {
  CAP_CHECK_INIT();
  if (!dev_ptr_out) return CAPGPU_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  CAP_HIP(hipMalloc(dev_ptr_out, bytes ? bytes : 1));
  return CAPGPU_OK;
}
int capgpu_free(void* dev_ptr) {
  CAP_CHECK_INIT();
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  CAP_HIP(hipStreamSynchronize(ctx().stream));
  CAP_HIP(hipFree(dev_ptr));
  return CAPGPU_OK;
}
int capgpu_memcpy_h2d(void* dev_dst, const void* host_src, size_t bytes) {
  CAP_CHECK_INIT();
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  CAP_HIP(hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, ctx().stream));
  CAP_HIP(hipStreamSynchronize(ctx().stream));
  return CAPGPU_OK;
}
int capgpu_memcpy_d2h(void* host_dst, const void* dev_src, size_t bytes) {
  CAP_CHECK_INIT();
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  CAP_HIP(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, ctx().stream));
  CAP_HIP(hipStreamSynchronize(ctx().stream));
  return CAPGPU_OK;
}
int capgpu_sync(void) {
  CAP_CHECK_INIT();
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  CAP_HIP(hipStreamSynchronize(ctx().stream));
  return CAPGPU_OK;
}
int capgpu_set_stream(void* hip_stream) {
  CAP_CHECK_INIT();
  Context& c = ctx();
  std::lock_guard<std::recursive_mutex> lk(c.mu);
  CAP_HIP(hipStreamSynchronize(c.stream));
  c.stream = hip_stream ? (hipStream_t)hip_stream : c.own_stream;
  return CAPGPU_OK;
}

// ---- SRS ------------------------------------------------------------------------------------------
int capgpu_srs_upload(const void* bases, size_t n, size_t stride_bytes, int coords_montgomery,
                      uint64_t* handle_out) {
  CAP_CHECK_INIT();
  if (!handle_out || (!bases && n) || (stride_bytes != 64 && stride_bytes != 72)) {
    set_error("capgpu_srs_upload: bad argument (stride must be 64 or 72)");
    return CAPGPU_ERR_INVALID_ARG;
  }
  Context& c = ctx();
  std::lock_guard<std::recursive_mutex> lk(c.mu);
  std::vector<g1_affine> packed(n ? n : 1);
  const unsigned char* src = (const unsigned char*)bases;
  for (size_t i = 0; i < n; i++) {
    memcpy(&packed[i], src + i * stride_bytes, 64);
    if (stride_bytes == 72 && src[i * stride_bytes + 64]) memset(&packed[i], 0, 64);  // infinity flag
  }
  DevTmp<g1_affine> d;
  CAP_HIP(d.alloc(n));
  CAP_HIP(hipMemcpyAsync(d, packed.data(), sizeof(g1_affine) * n, hipMemcpyHostToDevice, c.stream));
  if (!coords_montgomery && n) {
    size_t cnt = 2 * n;
    launch("fq_to_mont_kernel", fq_to_mont_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, c.stream,
           reinterpret_cast<fe*>(d.p), cnt);
  }
  return register_srs(d, n, handle_out);
}

static int srs_generate_common(int mode, const uint64_t a[4], const uint64_t b[4], size_t n, uint64_t* handle_out) {
  CAP_CHECK_INIT();
  if (!handle_out || !a || n == 0) {
    set_error("capgpu_srs_generate: bad argument");
    return CAPGPU_ERR_INVALID_ARG;
  }
  Context& c = ctx();
  std::lock_guard<std::recursive_mutex> lk(c.mu);
  DevTmp<g1_affine> d;
  CAP_HIP(d.alloc(n));
  DevTmp<fe> d_tab, d_pw;
  fe am = Fr::to_mont(fe_from_u64x4(a));
  fe bm = b ? Fr::to_mont(fe_from_u64x4(b)) : Fr::zero();
  if (mode == 0) {
    std::vector<fe> pw(64);
    fe x = am;
    for (int i = 0; i < 64; i++) {
      pw[i] = x;
      x = Fr::sqr(x);
    }
    CAP_HIP(d_pw.alloc(64));
    CAP_HIP(d_tab.alloc(n));
    CAP_HIP(hipMemcpyAsync(d_pw, pw.data(), sizeof(fe) * 64, hipMemcpyHostToDevice, c.stream));
    launch("fr_powers_kernel", fr_powers_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c.stream, d_tab.p, n,
           (const fe*)d_pw.p);
    CAP_HIP(hipStreamSynchronize(c.stream));
  }
  launch("srs_fixed_base_kernel", srs_fixed_base_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c.stream,
         d.p, n, mode, (const fe*)d_tab.p, am, bm);
  CAP_HIP(hipStreamSynchronize(c.stream));
  return register_srs(d, n, handle_out);
}
int capgpu_srs_generate(const uint64_t tau[4], size_t n, uint64_t* handle_out) {
  return srs_generate_common(0, tau, nullptr, n, handle_out);
}
int capgpu_srs_generate_affine_seq(const uint64_t a[4], const uint64_t b[4], size_t n, uint64_t* handle_out) {
  if (!b) return CAPGPU_ERR_INVALID_ARG;
  return srs_generate_common(1, a, b, n, handle_out);
}

int capgpu_srs_size(uint64_t handle, size_t* n_out) {
  CAP_CHECK_INIT();
  Context& c = ctx();
  std::lock_guard<std::recursive_mutex> lk(c.mu);
  auto it = c.srs.find(handle);
  if (it == c.srs.end() || !n_out) {
    set_error("capgpu: unknown SRS handle %llu", (unsigned long long)handle);
    return CAPGPU_ERR_BAD_HANDLE;
  }
  *n_out = it->second.bases.n;
  return CAPGPU_OK;
}
int capgpu_srs_download(uint64_t handle, size_t offset, size_t n, void* out) {
  CAP_CHECK_INIT();
  Context& c = ctx();
  std::lock_guard<std::recursive_mutex> lk(c.mu);
  auto it = c.srs.find(handle);
  if (it == c.srs.end()) {
    set_error("capgpu: unknown SRS handle %llu", (unsigned long long)handle);
    return CAPGPU_ERR_BAD_HANDLE;
  }
  if (offset + n > it->second.bases.n || !out) return CAPGPU_ERR_INVALID_ARG;
  CAP_HIP(hipMemcpyAsync(out, it->second.bases.ext + offset, sizeof(g1_affine) * n, hipMemcpyDeviceToHost, c.stream));
  CAP_HIP(hipStreamSynchronize(c.stream));
  // the resident table is in the internal Montgomery form (x * 2^261); hand back arkworks' form (x * 2^256)
  g1_affine* pts = reinterpret_cast<g1_affine*>(out);
  for (size_t i = 0; i < n; i++) {
    if (G1::is_inf(pts[i])) continue;
    pts[i].x = Fq29::to_ext(Fq29::load(pts[i].x));
    pts[i].y = Fq29::to_ext(Fq29::load(pts[i].y));
  }
  return CAPGPU_OK;
}
int capgpu_srs_free(uint64_t handle) {
  CAP_CHECK_INIT();
  Context& c = ctx();
  std::lock_guard<std::recursive_mutex> lk(c.mu);
  auto it = c.srs.find(handle);
  if (it == c.srs.end()) {
    set_error("capgpu: unknown SRS handle %llu", (unsigned long long)handle);
    return CAPGPU_ERR_BAD_HANDLE;
  }
  hipStreamSynchronize(c.stream);
  msm_free_bases(&it->second.bases);
  c.srs.erase(it);
  return CAPGPU_OK;
}

// ---- MSM ------------------------------------------------------------------------------------------
int capgpu_msm_g1_dev(uint64_t srs_handle, size_t offset, const void* d_scalars, size_t scalar_stride, size_t n,
                      int count, int scalars_montgomery, void* d_out_xyz) {
  CAP_CHECK_INIT();
  Context& c = ctx();
  std::lock_guard<std::recursive_mutex> lk(c.mu);
  auto it = c.srs.find(srs_handle);
  if (it == c.srs.end()) {
    set_error("capgpu: unknown SRS handle %llu", (unsigned long long)srs_handle);
    return CAPGPU_ERR_BAD_HANDLE;
  }
  const MsmBases& B = it->second.bases;
  if (count < 0 || offset + n > B.n || (!d_scalars && n) || !d_out_xyz) {
    set_error("capgpu_msm_g1: bad argument (offset %zu + n %zu vs SRS size %zu)", offset, n, B.n);
    return CAPGPU_ERR_INVALID_ARG;
  }
  if (count == 0) return CAPGPU_OK;
  size_t need = msm_workspace_bytes(B, n, (uint32_t)count);
  int rc = scratch_reserve(c.msm_ws, need);
  if (rc) return rc;
  rc = msm_run(B, offset, (const fe*)d_scalars, scalar_stride, 1, 0, n, (uint32_t)count, scalars_montgomery,
               (g1_jac*)d_out_xyz, c.msm_ws.p, c.msm_ws.cap, c.stream);
  if (rc) return hip_fail((hipError_t)rc, "msm_run");
  return take_launch_error();
}

int capgpu_msm_plan(uint64_t srs_handle, size_t n, int count, char* buf, size_t cap) {
  CAP_CHECK_INIT();
  Context& c = ctx();
  std::lock_guard<std::recursive_mutex> lk(c.mu);
  const MsmBases* B = nullptr;
  int rc = find_srs(srs_handle, &B);
  if (rc) return rc;
  if (!buf || cap == 0 || count < 1 || n > B->n) return CAPGPU_ERR_INVALID_ARG;
  msm_plan_describe(*B, n, (uint32_t)count, buf, cap);
  return CAPGPU_OK;
}

int capgpu_msm_g1_batch(uint64_t srs_handle, const size_t* offsets, const uint64_t* const* scalars, const size_t* ns,
                        int count, uint64_t* out_xyz) {
  CAP_CHECK_INIT();
  if (count < 0 || (count && (!offsets || !scalars || !ns || !out_xyz))) return CAPGPU_ERR_INVALID_ARG;
  Context& c = ctx();
  std::lock_guard<std::recursive_mutex> lk(c.mu);
  // equal (offset, n) entries run as one batched launch; otherwise one launch each
  bool uniform = true;
  for (int i = 1; i < count; i++) uniform = uniform && offsets[i] == offsets[0] && ns[i] == ns[0];
  int groups = uniform ? 1 : count;
  for (int g = 0; g < groups; g++) {
    int first = uniform ? 0 : g, cnt = uniform ? count : 1;
    size_t n = ns[first];
    int rc = scratch_reserve(c.stage_a, sizeof(fe) * (n ? n : 1) * cnt + 96 * cnt);
    if (rc) return rc;
    fe* d_sc = (fe*)c.stage_a.p;
    void* d_out = (char*)c.stage_a.p + sizeof(fe) * (n ? n : 1) * cnt;
    for (int k = 0; k < cnt; k++)
      CAP_HIP(hipMemcpyAsync(d_sc + (size_t)k * n, scalars[first + k], sizeof(fe) * n, hipMemcpyHostToDevice,
                             c.stream));
    rc = capgpu_msm_g1_dev(srs_handle, offsets[first], d_sc, n, n, cnt, 0, d_out);
    if (rc) return rc;
    CAP_HIP(hipMemcpyAsync(out_xyz + 12 * (size_t)first, d_out, 96 * (size_t)cnt, hipMemcpyDeviceToHost, c.stream));
    CAP_HIP(hipStreamSynchronize(c.stream));
  }
  return CAPGPU_OK;
}

int capgpu_msm_g1(uint64_t srs_handle, size_t offset, const uint64_t* scalars, size_t n, uint64_t out_xyz[12]) {
  const uint64_t* sp[1] = {scalars};
  return capgpu_msm_g1_batch(srs_handle, &offset, sp, &n, 1, out_xyz);
}

int capgpu_g1_sum(const uint64_t* points_xyz, size_t n, uint64_t out_xyz[12]) {
  CAP_CHECK_INIT();
  if ((!points_xyz && n) || !out_xyz) return CAPGPU_ERR_INVALID_ARG;
  Context& c = ctx();
  std::lock_guard<std::recursive_mutex> lk(c.mu);
  int rc = scratch_reserve(c.stage_a, sizeof(g1_jac) * (n + 1));
  if (rc) return rc;
  g1_jac* d = (g1_jac*)c.stage_a.p;
  if (n) CAP_HIP(hipMemcpyAsync(d + 1, points_xyz, sizeof(g1_jac) * n, hipMemcpyHostToDevice, c.stream));
  launch("g1_sum_kernel", g1_sum_kernel, dim3(1), dim3(64), 0, c.stream, (const g1_jac*)(d + 1), n, d);
  CAP_HIP(hipMemcpyAsync(out_xyz, d, sizeof(g1_jac), hipMemcpyDeviceToHost, c.stream));
  CAP_HIP(hipStreamSynchronize(c.stream));
  return CAPGPU_OK;
}

// ---- NTT ------------------------------------------------------------------------------------------
int capgpu_ntt_fr_dev(void* d_data, size_t stride_elems, int count, uint32_t log_n, int dir, int coset) {
  CAP_CHECK_INIT();
  if (count < 0 || (count && !d_data) || log_n > 28 || (dir != 0 && dir != 1) || (coset != 0 && coset != 1) ||
      (count > 1 && stride_elems < ((size_t)1 << log_n))) {
    set_error("capgpu_ntt_fr: bad argument (log_n %u, dir %d, coset %d, count %d)", log_n, dir, coset, count);
    return CAPGPU_ERR_INVALID_ARG;
  }
  if (count == 0) return CAPGPU_OK;
  Context& c = ctx();
  std::lock_guard<std::recursive_mutex> lk(c.mu);
  const NttDomain* dom = nullptr;
  int rc = get_domain(log_n, &dom);
  if (rc) return rc;
  size_t n = (size_t)1 << log_n;
  size_t stride = count > 1 ? stride_elems : n;
  rc = scratch_reserve(c.ntt_scratch, sizeof(fe) * stride * count);
  if (rc) return rc;
  rc = ntt_run(*dom, c.small, (fe*)d_data, (fe*)c.ntt_scratch.p, stride, (uint32_t)count, dir, coset, c.stream);
  if (rc) return hip_fail((hipError_t)rc, "ntt_run");
  return take_launch_error();
}

int capgpu_ntt_fr_batch(uint64_t* const* data, int count, uint32_t log_n, int dir, int coset) {
  CAP_CHECK_INIT();
  if (count < 0 || (count && !data) || log_n > 28) return CAPGPU_ERR_INVALID_ARG;
  if (count == 0) return CAPGPU_OK;
  Context& c = ctx();
  std::lock_guard<std::recursive_mutex> lk(c.mu);
  size_t n = (size_t)1 << log_n;
  int rc = scratch_reserve(c.stage_b, sizeof(fe) * n * count);
  if (rc) return rc;
  fe* d = (fe*)c.stage_b.p;
  for (int k = 0; k < count; k++) {
    if (!data[k]) return CAPGPU_ERR_INVALID_ARG;
    CAP_HIP(hipMemcpyAsync(d + (size_t)k * n, data[k], sizeof(fe) * n, hipMemcpyHostToDevice, c.stream));
  }
  rc = capgpu_ntt_fr_dev(d, n, count, log_n, dir, coset);
  if (rc) return rc;
  for (int k = 0; k < count; k++)
    CAP_HIP(hipMemcpyAsync(data[k], d + (size_t)k * n, sizeof(fe) * n, hipMemcpyDeviceToHost, c.stream));
  CAP_HIP(hipStreamSynchronize(c.stream));
  return CAPGPU_OK;
}

int capgpu_ntt_fr(uint64_t* data, uint32_t log_n, int dir, int coset) {
  uint64_t* p[1] = {data};
  return capgpu_ntt_fr_batch(p, 1, log_n, dir, coset);
}

// ---- instrumentation ------------------------------------------------------------------------------
int capgpu_ubench_mad_rate(double* lane_ops_per_s_out) {
  CAP_CHECK_INIT();
  if (!lane_ops_per_s_out) return CAPGPU_ERR_INVALID_ARG;
  Context& c = ctx();
  std::lock_guard<std::recursive_mutex> lk(c.mu);
  hipDeviceProp_t prop;
  CAP_HIP(hipGetDeviceProperties(&prop, c.device));
  const int blocks = prop.multiProcessorCount * 8, iters = 2000;  // 8 workgroups of 4 waves per CU: 8 waves per SIMD
  DevTmp<uint64_t> d;
  CAP_HIP(d.alloc((size_t)blocks * 256));
  CAP_HIP(hipMemsetAsync(d, 0x5a, sizeof(uint64_t) * blocks * 256, c.stream));
  hipEvent_t e0, e1;
  CAP_HIP(hipEventCreate(&e0));
  CAP_HIP(hipEventCreate(&e1));
  hipLaunchKernelGGL(ubench_mad_kernel, dim3(blocks), dim3(256), 0, c.stream, d.p, 10);  // warm-up
  double best = 0;
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0, c.stream);
    hipLaunchKernelGGL(ubench_mad_kernel, dim3(blocks), dim3(256), 0, c.stream, d.p, iters);
    hipEventRecord(e1, c.stream);
    CAP_HIP(hipEventSynchronize(e1));
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double rate = (double)blocks * 256 * iters * 64 / (ms * 1e-3);
    if (rate > best) best = rate;
  }
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  *lane_ops_per_s_out = best;
  return take_launch_error();
}

int capgpu_profile_enable(int on) {
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  profiler().on = on != 0;
  return CAPGPU_OK;
}
int capgpu_profile_reset(void) {
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  profiler().reset();
  return CAPGPU_OK;
}
int capgpu_profile_get(const char* name, double* total_ms_out, uint64_t* launches_out) {
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  if (!name) return CAPGPU_ERR_INVALID_ARG;
  const auto& st = profiler().stats();
  auto it = st.find(name);
  double ms = 0;
  uint64_t cnt = 0;
  if (it != st.end()) {
    ms = it->second.ms;
    cnt = it->second.launches;
  }
  if (total_ms_out) *total_ms_out = ms;
  if (launches_out) *launches_out = cnt;
  return CAPGPU_OK;
}
int capgpu_profile_dump(char* buf, size_t cap) {
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  if (!buf || cap == 0) return CAPGPU_ERR_INVALID_ARG;
  size_t o = 0;
  buf[0] = 0;
  for (const auto& kv : profiler().stats()) {
    int w = snprintf(buf + o, cap - o, "%s %.6f %llu\n", kv.first.c_str(), kv.second.ms,
                     (unsigned long long)kv.second.launches);
    if (w < 0 || (size_t)w >= cap - o) break;
    o += (size_t)w;
  }
  return CAPGPU_OK;
}

}  // extern "C"
