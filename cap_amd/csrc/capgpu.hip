// C-ABI entry points of libcapgpu.so (include/capgpu.h): lifecycle, the device contexts and the table of logical
// handles (context.hpp), device memory plumbing, SRS management, MSM and NTT.  The PLONK entry points live in
// plonk.hip.  There is no CPU fallback anywhere in this library.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <unistd.h>

#include <algorithm>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "context.hpp"
#include "curve29.hpp"
#include "lagrange.hpp"
#include "launch.hpp"
#include "params.hpp"

namespace cap {

static thread_local char g_err[512] = "";
static thread_local Context* tl_ctx = nullptr;  // set by ScopedCtx
static thread_local int tl_bound = -1;          // capgpu_set_device

Runtime& rt() {
  static Runtime r;
  return r;
}
Context& ctx() {
  if (tl_ctx) return *tl_ctx;
  Runtime& R = rt();
  if (!R.initialised.load(std::memory_order_acquire) || R.ctxs.empty()) {
    static Context inert;  // initialised == false: every entry point refuses
    return inert;
  }
  const int s = tl_bound >= 0 && (size_t)tl_bound < R.ctxs.size() ? tl_bound : 0;
  return *R.ctxs[s];
}
// a binding made before a shutdown / re-init with fewer contexts is stale: treated as unbound
bool force_replicate() {
  const char* e = getenv("CAPGPU_FORCE_REPLICATE");
  return e && atoi(e) != 0;
}
ScalarSet::~ScalarSet() {
  Runtime& R = rt();
  for (ScalarSlice& sl : slices) {
    if (!sl.d) continue;
    if (sl.slot >= 0 && (size_t)sl.slot < R.ctxs.size()) {
      Context& home = *R.ctxs[(size_t)sl.slot];
      // capgpu_msm_g1_resident lets ANY context of the slice's device run MSMs on it, on that context's own stream, and
      // returns with the work enqueued: every such stream is drained - under its context's lock, in slot order, so that no
      // capture or enqueue of another thread is in flight on it - before the memory goes (round-4 ADVICE)
      for (auto& cp : R.ctxs) {
        if (cp->device != home.device || !cp->initialised) continue;
        ScopedCtx sc(*cp);
        Entry lk(*cp);
        (void)hipStreamSynchronize(cp->stream);
      }
      ScopedCtx sc(home);
      (void)hipFree(sl.d);
    }
    sl.d = nullptr;
  }
}
int thread_bound_slot() { return tl_bound >= 0 && (size_t)tl_bound < rt().ctxs.size() ? tl_bound : -1; }
int& thread_entry_depth() {
  static thread_local int d = 0;
  return d;
}

ScopedCtx::ScopedCtx(Context& c) : prev(tl_ctx) {
  // an unbound caller thread keeps ITS HIP device across the call (it may be torch's, or another library's)
  if (!prev && hipGetDevice(&prev_device) != hipSuccess) prev_device = -1;
  tl_ctx = &c;
  (void)hipSetDevice(c.device);
}
ScopedCtx::~ScopedCtx() {
  tl_ctx = prev;
  if (prev) (void)hipSetDevice(prev->device);
  else if (prev_device >= 0) (void)hipSetDevice(prev_device);
}

Profiler& profiler() { return ctx().prof; }
LaunchError& launch_error() { return ctx().lerr; }

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* last_error() { return g_err; }
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

// bytes of scratch (the Scratch buffers of every context) the library holds per HIP device, and the cap on it
// (capgpu_set_memory_limit; 0 = none).  Tables the caller created - SRS, keys, domains - are not scratch.
static std::atomic<size_t> g_scratch_bytes[64];
static std::atomic<size_t> g_scratch_limit{0};
static void scratch_count(int device, size_t add, size_t sub) {
  std::atomic<size_t>& a = g_scratch_bytes[(size_t)device & 63];
  if (add) a.fetch_add(add);
  if (sub) a.fetch_sub(sub);
}
void scratch_account(int device, size_t add, size_t sub) { scratch_count(device, add, sub); }
bool scratch_room_for(int device, size_t bytes) {
  const size_t limit = g_scratch_limit.load();
  return !limit || g_scratch_bytes[(size_t)device & 63].load() + bytes <= limit;
}

// every stream of the context that may still be running kernels on its scratch (the side stream of the
// small-batch overlap and the copy stream were not drained before a buffer was freed)
static hipError_t drain_context_streams(Context& c) {
  hipError_t e = hipStreamSynchronize(c.stream);
  if (c.own_stream && c.own_stream != c.stream && e == hipSuccess) e = hipStreamSynchronize(c.own_stream);
  if (c.side_stream && e == hipSuccess) e = hipStreamSynchronize(c.side_stream);
  if (c.copy_stream && e == hipSuccess) e = hipStreamSynchronize(c.copy_stream);
  return e;
}

// Releases what context c holds beyond its tables: scratch buffers, the pinned result area, captured prover graphs (they
// name scratch addresses).  The caller holds c.mu.  Returns the device bytes released.
static size_t trim_context(Context& c) {
  int prev = -1;
  (void)hipGetDevice(&prev);
  (void)hipSetDevice(c.device);
  (void)drain_context_streams(c);
  size_t released = 0;
  c.prove_graphs.reset();
  for (Scratch* s : {&c.ntt_scratch, &c.msm_ws, &c.stage_a, &c.stage_b, &c.prove_ws, &c.gather}) {
    if (!s->p) continue;
    (void)hipFree(s->p);
    released += s->cap;
    scratch_count(c.device, 0, s->cap);
    s->p = nullptr;
    s->cap = 0;
  }
  if (c.pin_host) {
    (void)hipHostFree(c.pin_host);
    c.pin_host = nullptr;
    c.pin_cap = 0;
  }
  (void)hipGetLastError();
  if (prev >= 0) (void)hipSetDevice(prev);
  return released;
}
// every context of `device` (all devices: -1) nobody is using right now, except `keep`; *busy_out counts the others
static size_t trim_idle_contexts(int device, Context* keep, int* busy_out) {
  size_t released = 0;
  for (auto& cp : rt().ctxs) {
    Context& o = *cp;
    if (&o == keep || (device >= 0 && o.device != device)) continue;
    if (!o.mu.try_lock()) {
      if (busy_out) (*busy_out)++;
      continue;
    }
    if (o.depth == 0) released += trim_context(o);  // (depth > 0: an entry point of THIS thread is running on it)
    else if (busy_out) (*busy_out)++;
    o.mu.unlock();
  }
  return released;
}

double& scratch_growth_scale() {
  static thread_local double f = 1.0;
  return f;
}

void context_trim_if_pending(Context& c) {
  c.trim_pending = false;
  (void)trim_context(c);
}

int scratch_reserve(Scratch& s, size_t bytes) {
  if (bytes <= s.cap) return CAPGPU_OK;
  Context& c = ctx();
  if (c.capturing) {  // (the capture is abandoned and the work enqueued directly: plonk.hip, GraphRun)
    set_error("capgpu: scratch growth during stream capture");
    return CAPGPU_ERR_HIP;
  }
  if (s.p) {
    CAP_HIP(drain_context_streams(c));
    CAP_HIP(hipFree(s.p));
    scratch_count(c.device, 0, s.cap);
    s.p = nullptr;
    s.cap = 0;
  }
  size_t want = std::max(bytes + bytes / 4, (size_t)((double)bytes * std::min(scratch_growth_scale(), 64.0)));
  if (const size_t limit = g_scratch_limit.load()) {
    // capgpu_set_memory_limit: first without the growth slack, then with what the device's idle contexts give back
    auto over = [&](size_t w) { return g_scratch_bytes[(size_t)c.device & 63].load() + w > limit; };
    if (over(want)) want = bytes;
    if (over(want)) (void)trim_idle_contexts(c.device, &c, nullptr);
    if (over(want)) {
      c.trim_pending = true;  // (what this failing call has already grown is released at the context's next entry)
      set_error("capgpu: this call needs %zu more bytes of device scratch; %zu are in use on device %d and "
                "capgpu_set_memory_limit allows %zu (prove in smaller batches, or raise the limit)",
                want, g_scratch_bytes[(size_t)c.device & 63].load(), c.device, limit);
      return CAPGPU_ERR_OOM;
    }
  }
  hipError_t e = hipMalloc(&s.p, want);
  if (e == hipErrorOutOfMemory) {  // the device is full: what idle contexts hold may be all that is missing
    (void)hipGetLastError();
    s.p = nullptr;
    if (trim_idle_contexts(c.device, &c, nullptr)) e = hipMalloc(&s.p, want);
  }
  if (e != hipSuccess) {
    s.p = nullptr;
    return hip_fail(e, "hipMalloc (scratch)");
  }
  s.cap = want;
  scratch_count(c.device, want, 0);
  return CAPGPU_OK;
}

int pinned_reserve(Context& c, size_t bytes) {
  if (bytes <= c.pin_cap) return CAPGPU_OK;
  if (c.pin_host) {
    CAP_HIP(drain_context_streams(c));
    CAP_HIP(hipHostFree(c.pin_host));
    c.pin_host = nullptr;
    c.pin_cap = 0;
  }
  const size_t want = std::max<size_t>(bytes + bytes / 4, (size_t)1 << 16);
  CAP_HIP(hipHostMalloc(&c.pin_host, want, hipHostMallocDefault));
  c.pin_cap = want;
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

hipError_t copy_between(void* dst, int dst_device, const void* src, int src_device, size_t bytes, hipStream_t s) {
  if (bytes == 0) return hipSuccess;
  if (dst_device == src_device) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s);
  return hipMemcpyPeerAsync(dst, dst_device, src, src_device, bytes, s);
}

Context* try_acquire_context() {
  Runtime& R = rt();
  const size_t n = R.ctxs.size();
  const uint32_t start = R.rr.load(std::memory_order_relaxed);
  for (size_t i = 0; i < n; i++) {
    Context* c = R.ctxs[(start + i) % n].get();
    if (c->mu.try_lock()) {
      R.rr.store((uint32_t)((start + i + 1) % n), std::memory_order_relaxed);
      return c;
    }
  }
  return nullptr;
}
Context& pick_context() {
  Runtime& R = rt();
  if (tl_ctx) return *tl_ctx;  // already placed by a dispatcher
  if (const int ss = comm_shard_slot(); ss >= 0 && (size_t)ss < R.ctxs.size()) return *R.ctxs[(size_t)ss];
  if (tl_bound >= 0 || R.ctxs.size() <= 1) return ctx();
  if (Context* c = try_acquire_context()) {
    c->mu.unlock();  // the caller takes the lock through Entry; losing the race to another thread only costs a wait
    return *c;
  }
  return *R.ctxs[R.rr.fetch_add(1, std::memory_order_relaxed) % R.ctxs.size()];
}

namespace {

// ---- SRS generation kernels ---------------------------------------------------------------
// scalar_i = mode 0: table[i] (Montgomery Fr, e.g. tau^i); mode 1: a + i*b.  out[i] = [scalar_i] G.
// `first`: logical index of out[0] (a shard of a sharded SRS generates its own point range)
__global__ __launch_bounds__(256) void srs_fixed_base_kernel(g1_affine* __restrict__ out, size_t n, int mode,
                                                             const fe* __restrict__ table, fe a_mont, fe b_mont,
                                                             size_t first) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fe s;
  if (mode == 0) {
    s = table[i];
  } else {
    fe idx = Fr::zero();
    idx.v[0] = (uint32_t)(first + i);
    idx.v[1] = (uint32_t)((uint64_t)(first + i) >> 32);
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

__global__ void fr_powers_kernel(fe* out, size_t n, const fe* __restrict__ pw, size_t first) {
  size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  fe* dst = out + e;
  e += first;
  fe r = Fr::one();
  bool started = false;
  for (int b = 0; (e >> b) != 0; b++) {
    if ((e >> b) & 1) {
      r = started ? Fr::mul(r, pw[b]) : pw[b];
      started = true;
    }
  }
  *dst = r;
}

__global__ void fr_scale_kernel(fe* data, size_t count, fe factor_mont) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) data[i] = Fr::mul(data[i], factor_mont);
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


// Issue rate of one VALU instruction class (capgpu_ubench_issue_rates): the same shape as ubench_mad_kernel - 8
// independent dependency chains per lane, 8 waves per SIMD - so that the classes are compared at equal occupancy.
//   0 v_mad_u64_u32   1 v_add_u32   2 v_and_b32   3 v_mov_b32   4 v_lshl_add_u64   5 v_lshrrev_b64   6 v_alignbit_b32
//   7 v_mul_lo_u32
//   8 the mixed stream of a column-wise Montgomery product: three multiply-adds, then one plain instruction (and / add
//     alternating) - msm_accumulate's common path is 1558 : 563.  Is a mix issued in the sum of its classes' own times?
//     (rates_out[9] is the same kernel held to three waves per SIMD, msm_accumulate's occupancy.)
template <int CLS>
__global__ __launch_bounds__(256) void ubench_issue_kernel(uint64_t* io, int iters) {
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
        if constexpr (CLS == 0) {
          uint64_t r, carry;
          asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(carry) : "v"(a), "v"(b + c), "v"(acc[c]));
          acc[c] = r;
        } else if constexpr (CLS == 1) {
          uint32_t r;
          asm volatile("v_add_u32 %0, %1, %2" : "=v"(r) : "v"(a), "v"((uint32_t)acc[c]));
          acc[c] = r;
        } else if constexpr (CLS == 2) {
          uint32_t r;
          asm volatile("v_and_b32 %0, %1, %2" : "=v"(r) : "v"(a), "v"((uint32_t)acc[c]));
          acc[c] = r;
        } else if constexpr (CLS == 3) {
          uint32_t r;
          asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"((uint32_t)acc[c]));
          acc[c] = r;
        } else if constexpr (CLS == 4) {
          uint64_t r;
          asm volatile("v_lshl_add_u64 %0, %1, 1, %2" : "=v"(r) : "v"(acc[c]), "v"((uint64_t)a));
          acc[c] = r;
        } else if constexpr (CLS == 5) {
          uint64_t r;
          asm volatile("v_lshrrev_b64 %0, %1, %2" : "=v"(r) : "v"(b & 3u), "v"(acc[c]));
          acc[c] = r;
        } else if constexpr (CLS == 6) {
          uint32_t r;
          asm volatile("v_alignbit_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"((uint32_t)acc[c]), "v"(b & 31u));
          acc[c] = r;
        } else if constexpr (CLS == 7) {
          uint32_t r;
          asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(r) : "v"(a), "v"((uint32_t)acc[c]));
          acc[c] = r;
        } else if ((c & 3) != 3) {  // chains 0-2, 4-6: multiply-adds
          uint64_t r, carry;
          asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(carry) : "v"(a), "v"(b + c), "v"(acc[c]));
          acc[c] = r;
        } else {  // chains 3 and 7: v_and_b32 / v_add_u32
          uint32_t r;
          if (c == 3) asm volatile("v_and_b32 %0, %1, %2" : "=v"(r) : "v"(a), "v"((uint32_t)acc[c]));
          else asm volatile("v_add_u32 %0, %1, %2" : "=v"(r) : "v"(a), "v"((uint32_t)acc[c]));
          acc[c] = r;
        }
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


// a new SRS entry on the current context: window tables of `n` device-resident bases
int make_srs_entry(g1_affine* d_bases, size_t n, size_t range_lo, std::shared_ptr<SrsEntry>* out) {
  Context& c = ctx();
  auto e = std::make_shared<SrsEntry>();
  e->device = c.device;
  e->range_lo = range_lo;
  uint32_t cw = msm_choose_window(n);
  int rc = msm_precompute(&e->bases, d_bases, n, cw, c.stream);
  if (rc) return hip_fail((hipError_t)rc, "msm_precompute");
  CAP_HIP(hipStreamSynchronize(c.stream));
  // msm_precompute latches a failed launch of its table kernels (launch.hpp); a handle whose window tables were never
  // written must not be published: the entry is dropped here (its destructor frees the tables)
  if ((rc = take_launch_error())) return rc;
  *out = e;
  return CAPGPU_OK;
}

// peer copy of a resident SRS onto the current context
int clone_srs_to_current(const SrsEntry& src, std::shared_ptr<SrsEntry>* out) {
  Context& c = ctx();
  auto e = std::make_shared<SrsEntry>();
  e->device = c.device;
  e->range_lo = src.range_lo;
  e->gamma_deg = src.gamma_deg;
  e->gamma_pts = src.gamma_pts;
  e->neg_h = src.neg_h;
  e->ck_gamma_pts = src.ck_gamma_pts;
  e->ck_gamma_g = src.ck_gamma_g;
  e->has_ck_gamma_g = src.has_ck_gamma_g;
  MsmBases& b = e->bases;
  b.n = src.bases.n;
  b.c = src.bases.c;
  b.windows = src.bases.windows;
  b.c2 = src.bases.c2;
  b.windows2 = src.bases.windows2;
  b.c3 = src.bases.c3;
  b.windows3 = src.bases.windows3;
  b.top_shift3 = src.bases.top_shift3;
  const size_t n1 = b.n ? b.n : 1;
  CAP_HIP(hipMalloc(&b.ext, sizeof(g1_affine) * n1 * b.windows));
  CAP_HIP(copy_between(b.ext, c.device, src.bases.ext, src.device, sizeof(g1_affine) * n1 * b.windows, c.stream));
  if (src.bases.ext2) {
    CAP_HIP(hipMalloc(&b.ext2, sizeof(g1_affine) * n1 * b.windows2));
    CAP_HIP(copy_between(b.ext2, c.device, src.bases.ext2, src.device, sizeof(g1_affine) * n1 * b.windows2, c.stream));
  }
  if (src.bases.ext3) {
    CAP_HIP(hipMalloc(&b.ext3, sizeof(g1_affine) * n1 * b.windows3));
    CAP_HIP(copy_between(b.ext3, c.device, src.bases.ext3, src.device, sizeof(g1_affine) * n1 * b.windows3, c.stream));
  }
  if (src.bases.ext0) {
    b.c0 = src.bases.c0;
    b.windows0 = src.bases.windows0;
    CAP_HIP(hipMalloc(&b.ext0, sizeof(g1_affine) * n1 * b.windows0));
    CAP_HIP(copy_between(b.ext0, c.device, src.bases.ext0, src.device, sizeof(g1_affine) * n1 * b.windows0, c.stream));
  }
  CAP_HIP(hipStreamSynchronize(c.stream));
  *out = e;
  return CAPGPU_OK;
}

}  // namespace

int register_srs(g1_affine* d_bases, size_t n, uint64_t* handle_out) {
  Context& c = ctx();
  std::shared_ptr<SrsEntry> e;
  int rc = make_srs_entry(d_bases, n, 0, &e);
  if (rc) return rc;
  Runtime& R = rt();
  const uint64_t h = R.next_handle.fetch_add(1);
  {
    std::lock_guard<std::mutex> lk(R.mu);
    SrsRecord rec;
    rec.full = e;
    rec.total_n = n;
    R.srs[h] = rec;
  }
  c.srs[h] = e;
  *handle_out = h;
  return CAPGPU_OK;
}

int srs_record(uint64_t h, SrsRecord* out) {
  Runtime& R = rt();
  std::lock_guard<std::mutex> lk(R.mu);
  auto it = R.srs.find(h);
  if (it == R.srs.end()) {
    set_error("capgpu: unknown SRS handle %llu", (unsigned long long)h);
    return CAPGPU_ERR_BAD_HANDLE;
  }
  *out = it->second;
  return CAPGPU_OK;
}

SrsEntry* find_srs_entry(uint64_t h) {
  Context& c = ctx();
  auto it = c.srs.find(h);
  return it == c.srs.end() ? nullptr : it->second.get();
}

int find_srs(uint64_t h, const MsmBases** out) {
  Context& c = ctx();
  auto it = c.srs.find(h);
  if (it == c.srs.end()) {
    // first use on this context: replicate from the home context (tables are immutable, the record keeps them alive)
    SrsRecord rec;
    int rc = srs_record(h, &rec);
    if (rc) return rc;
    if (rec.sharded()) {
      set_error("capgpu: SRS %llu is sharded by point range over the devices of this process (MSM only)",
                (unsigned long long)h);
      return CAPGPU_ERR_INVALID_ARG;
    }
    std::shared_ptr<SrsEntry> rep;
    if (rec.full->device == c.device && !force_replicate()) rep = rec.full;
    else if ((rc = clone_srs_to_current(*rec.full, &rep))) return rc;
    else rt().replications++;
    {
      Runtime& R = rt();
      std::lock_guard<std::mutex> lk(R.mu);
      if (R.srs.find(h) == R.srs.end()) {  // freed while it was being copied
        set_error("capgpu: unknown SRS handle %llu", (unsigned long long)h);
        return CAPGPU_ERR_BAD_HANDLE;
      }
    }
    it = c.srs.emplace(h, rep).first;
  }
  if (it->second->is_shard) {  // entries of a sharded handle are only ever used through the sharded MSM path
    set_error("capgpu: SRS %llu is sharded by point range over the devices of this process (MSM only)",
              (unsigned long long)h);
    return CAPGPU_ERR_INVALID_ARG;
  }
  *out = &it->second->bases;
  return CAPGPU_OK;
}

int find_lagrange(uint64_t h, uint32_t log_n, const MsmBases** out) {
  const MsmBases* B = nullptr;
  int rc = find_srs(h, &B);
  if (rc) return rc;
  Context& c = ctx();
  SrsEntry* e = c.srs.find(h)->second.get();
  std::lock_guard<std::mutex> lk(e->lag_mu);
  auto it = e->lagrange.find(log_n);
  if (it == e->lagrange.end()) {
    if (e->lagrange_failed.count(log_n)) {  // (not retried for every proof: the callers commit from coefficients instead)
      set_error("capgpu: the Lagrange-form commit key of the 2^%u domain could not be built earlier (out of device memory?)",
                log_n);
      return CAPGPU_ERR_OOM;
    }
    if (B->n < ((size_t)1 << log_n) + 3) {
      set_error("capgpu: SRS %llu has %zu points, the Lagrange-form commit key of a 2^%u domain needs %zu",
                (unsigned long long)h, B->n, log_n, ((size_t)1 << log_n) + 3);
      return CAPGPU_ERR_INVALID_ARG;
    }
    if (c.capturing) {
      set_error("capgpu: Lagrange-form commit key requested inside a stream capture");
      return CAPGPU_ERR_INVALID_ARG;
    }
    if (const char* t = getenv("CAPGPU_TEST_FAIL_LAGRANGE"); t && atoi(t)) {  // test hook: as if the device were full
      e->lagrange_failed.insert(log_n);
      set_error("capgpu: Lagrange-form commit key: out of device memory (CAPGPU_TEST_FAIL_LAGRANGE)");
      return CAPGPU_ERR_OOM;
    }
    std::unique_ptr<MsmBases> L(new MsmBases);
    rc = lagrange_build(*B, log_n, L.get(), c.stream);
    if (rc) {
      e->lagrange_failed.insert(log_n);
      msm_free_bases(L.get());
      return hip_fail((hipError_t)rc, "lagrange_build");
    }
    if ((rc = take_launch_error())) {
      e->lagrange_failed.insert(log_n);
      msm_free_bases(L.get());
      return rc;
    }
    it = e->lagrange.emplace(log_n, std::move(L)).first;
  }
  *out = it->second.get();
  return CAPGPU_OK;
}

uint64_t register_key(const std::shared_ptr<ProvingKey>& K) {
  Runtime& R = rt();
  const uint64_t h = R.next_handle.fetch_add(1);
  {
    std::lock_guard<std::mutex> lk(R.mu);
    R.keys[h] = K;
  }
  ctx().keys[h] = K;
  return h;
}


}  // namespace cap

using namespace cap;

namespace {

// ---- sharded SRS (one process, several devices: SURVEY 8e inside the process) ------------------------------------
// An SRS of at least this many points is cut by point range over the contexts when it is created.
size_t shard_min_points() {
  const char* e = getenv("CAPGPU_SHARD_MIN_POINTS");
  long long x = e ? atoll(e) : (1ll << 20);
  return (size_t)(x >= 2 ? x : 2);
}
// the contexts a sharded SRS is cut over: one per capgpu_init entry (extra contexts of a device share its shard's device)
std::vector<size_t> shard_slots() {
  std::vector<size_t> v;
  for (auto& c : rt().ctxs)
    if (c->primary) v.push_back((size_t)c->slot);
  return v;
}
bool should_shard(size_t n) {
  const size_t S = shard_slots().size();
  return S > 1 && n >= shard_min_points() && n >= S;
}
// range of shard r of S over n points
void shard_range(size_t n, size_t S, size_t r, size_t* lo, size_t* len) {
  const size_t base = n / S, rem = n % S;
  *lo = r * base + std::min(r, rem);
  *len = base + (r < rem ? 1 : 0);
}

// runs job(r) for the r-th shard context, each on its own host thread placed on that context (the caller holds all
// context locks: AllEntries); returns the first failure and makes its message the caller's
int for_each_context(const std::function<int(size_t)>& job) {
  const std::vector<size_t> slots = shard_slots();
  const size_t S = slots.size();
  std::vector<int> rcs(S, CAPGPU_OK);
  std::vector<std::string> errs(S);
  auto body = [&](size_t r) {
    ScopedCtx sc(*rt().ctxs[slots[r]]);
    rcs[r] = job(r);
    if (rcs[r]) errs[r] = last_error();
  };
  std::vector<std::thread> th;
  for (size_t r = 1; r < S; r++) th.emplace_back(body, r);
  body(0);
  for (auto& t : th) t.join();
  for (size_t r = 0; r < S; r++)
    if (rcs[r]) {
      set_error("%s", errs[r].c_str());
      return rcs[r];
    }
  return CAPGPU_OK;
}

// publishes the shards built by for_each_context as one logical handle
uint64_t register_shards(std::vector<std::shared_ptr<SrsEntry>>& shards, size_t n) {
  Runtime& R = rt();
  const uint64_t h = R.next_handle.fetch_add(1);
  const std::vector<size_t> slots = shard_slots();
  std::vector<std::shared_ptr<SrsEntry>> by_slot(R.ctxs.size());
  for (size_t r = 0; r < shards.size(); r++) {
    shards[r]->is_shard = true;
    R.ctxs[slots[r]]->srs[h] = shards[r];
    by_slot[slots[r]] = shards[r];
  }
  std::lock_guard<std::mutex> lk(R.mu);
  SrsRecord rec;
  rec.shards = by_slot;
  rec.total_n = n;
  R.srs[h] = rec;
  return h;
}

// `count` MSMs over logical points [offset, offset + n) of a sharded SRS.  scalars: host memory (h_scalars) or device
// memory of context `home` (d_scalars), array k at + k * stride elements.  Every context runs the part of the range it
// holds down to one point per MSM; the partials travel to `home` (peer copies of count * 96 bytes: the exchange step of
// SURVEY 8e) and one wavefront per MSM adds them up into d_out (on home).  The caller holds AllEntries and is on home.
// resident (optional): the scalar slices already sit on the shard contexts (capgpu_msm_scalars_*): nothing but the
// partials moves.
int msm_sharded(const SrsRecord& rec, Context& home, size_t offset, const uint64_t* h_scalars, const fe* d_scalars,
                size_t stride, size_t n, int count, int montgomery, g1_jac* d_out, const ScalarSet* resident = nullptr) {
  const std::vector<size_t> slots = shard_slots();
  rt().shard_calls++;
  const size_t S = slots.size();
  int rc = scratch_reserve(home.gather, sizeof(g1_jac) * S * (size_t)count);
  if (rc) return rc;
  g1_jac* gather = (g1_jac*)home.gather.p;
  CAP_HIP(hipMemsetAsync(gather, 0, sizeof(g1_jac) * S * (size_t)count, home.stream));  // Z = 0: infinity
  hipEvent_t ready = nullptr;  // home's stream up to here: the device scalars exist, the gather buffer is cleared
  CAP_HIP(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
  CAP_HIP(hipEventRecord(ready, home.stream));
  std::vector<hipEvent_t> done(S, nullptr);
  rc = for_each_context([&](size_t r) -> int {
    Context& c = ctx();
    if (!rec.shards[slots[r]]) return CAPGPU_OK;
    const SrsEntry& sh = *rec.shards[slots[r]];
    const size_t lo = std::max(offset, sh.range_lo), hi = std::min(offset + n, sh.range_lo + sh.bases.n);
    if (lo >= hi) return CAPGPU_OK;
    const size_t len = hi - lo;
    const ScalarSlice* res = nullptr;
    if (resident) {
      for (const ScalarSlice& sl : resident->slices)
        if (sl.slot == c.slot) res = &sl;
      if (!res || res->lo != lo || res->len != len) {
        set_error("capgpu_msm_g1_resident: the scalar set was not cut for this SRS's point ranges");
        return CAPGPU_ERR_INVALID_ARG;
      }
    }
    int r2 = scratch_reserve(c.stage_a, (res ? 0 : sizeof(fe) * len * (size_t)count) + sizeof(g1_jac) * (size_t)count);
    if (r2) return r2;
    fe* d_sc = res ? res->d : (fe*)c.stage_a.p;
    g1_jac* d_part = (g1_jac*)((char*)c.stage_a.p + (res ? 0 : sizeof(fe) * len * (size_t)count));
    CAP_HIP(hipStreamWaitEvent(c.stream, ready, 0));
    for (int k = 0; k < count && !res; k++) {
      // fallback form: the scalars live on the caller's side and every call scatters them again (448 MB leaving one GPU
      // per 2^24-point MSM) - capgpu_msm_scalars_scatter_dev / _upload make them resident once instead
      if (h_scalars)
        CAP_HIP(hipMemcpyAsync(d_sc + (size_t)k * len, h_scalars + 4 * ((size_t)k * stride + (lo - offset)),
                               sizeof(fe) * len, hipMemcpyHostToDevice, c.stream));
      else
        CAP_HIP(copy_between(d_sc + (size_t)k * len, c.device, d_scalars + (size_t)k * stride + (lo - offset),
                             home.device, sizeof(fe) * len, c.stream));
      if (h_scalars || c.slot != home.slot) rt().shard_scalar_bytes += sizeof(fe) * len;
    }
    if ((r2 = scratch_reserve(c.msm_ws, msm_workspace_bytes(sh.bases, len, (uint32_t)count)))) return r2;
    r2 = msm_run(sh.bases, lo - sh.range_lo, d_sc, len, 1, 0, len, (uint32_t)count, montgomery, d_part, c.msm_ws.p,
                 c.msm_ws.cap, c.stream);
    if (r2) return hip_fail((hipError_t)r2, "msm_run");
    CAP_HIP(copy_between(gather + r * (size_t)count, home.device, d_part, c.device, sizeof(g1_jac) * (size_t)count,
                         c.stream));
    if (c.slot != home.slot) rt().shard_partial_bytes += sizeof(g1_jac) * (size_t)count;
    CAP_HIP(hipEventCreateWithFlags(&done[r], hipEventDisableTiming));
    CAP_HIP(hipEventRecord(done[r], c.stream));
    return take_launch_error();
  });
  (void)hipEventDestroy(ready);
  for (size_t r = 0; r < S; r++) {
    if (!done[r]) continue;
    if (rc == CAPGPU_OK && hipStreamWaitEvent(home.stream, done[r], 0) != hipSuccess) rc = CAPGPU_ERR_HIP;
    (void)hipEventDestroy(done[r]);
  }
  if (rc) return rc;
  g1_sum_ranks(gather, (uint32_t)S, (uint32_t)count, (uint32_t)count, d_out, home.stream);
  return take_launch_error();
}

}  // namespace

extern "C" {

const char* capgpu_last_error(void) { return last_error(); }
const char* capgpu_version(void) { return "capgpu 0.2.0 (gfx950)"; }

// CAPGPU_SEGV_BACKTRACE=1 (diagnostics): a SIGSEGV / SIGABRT in any thread of the process prints that thread's native
// stack (symbol + offset: resolve with addr2line on libcapgpu.so) before the previous handler - Python's faulthandler
// under pytest - gets its turn.  A fault in a pool or dealer thread is otherwise invisible to the Python-side report.
static struct sigaction g_prev_segv, g_prev_abrt;
static int g_segv_fd = -1;  // a file of its own (CAPGPU_SEGV_BACKTRACE=<path>): a test runner may have redirected fd 2
static char g_segv_path[512];  // "<path>.<pid>", opened in the handler (open is async-signal-safe): no empty files otherwise
static void segv_backtrace(int sig, siginfo_t* info, void* uctx) {
  void* frames[64];
  const int n = backtrace(frames, 64);
  const char msg[] = "\ncapgpu: fatal signal, native stack of the faulting thread:\n";
  if (g_segv_path[0] && g_segv_fd < 0) g_segv_fd = open(g_segv_path, O_WRONLY | O_CREAT | O_APPEND, 0644);
  for (int fd : {2, g_segv_fd}) {
    if (fd < 0) continue;
    (void)!write(fd, msg, sizeof(msg) - 1);
    backtrace_symbols_fd(frames, n, fd);
    if (info) {
      char line[96];
      const int len = snprintf(line, sizeof(line), "signal %d, fault address %p, thread %ld\n", sig, info->si_addr, (long)gettid());
      if (len > 0) (void)!write(fd, line, (size_t)len);
    }
  }
  // the loaded objects' base addresses (offsets above are relative to them)
  if (g_segv_fd >= 0) {
    const int maps = open("/proc/self/maps", O_RDONLY);
    if (maps >= 0) {
      char buf[4096];
      ssize_t k;
      while ((k = read(maps, buf, sizeof(buf))) > 0) (void)!write(g_segv_fd, buf, (size_t)k);
      close(maps);
    }
  }
  struct sigaction* prev = sig == SIGSEGV ? &g_prev_segv : &g_prev_abrt;
  if (prev->sa_flags & SA_SIGINFO) {
    if (prev->sa_sigaction) prev->sa_sigaction(sig, info, uctx);
  } else if (prev->sa_handler && prev->sa_handler != SIG_DFL && prev->sa_handler != SIG_IGN) {
    prev->sa_handler(sig);
  }
  signal(sig, SIG_DFL);
  raise(sig);
}
static void install_segv_backtrace() {
  const char* e = getenv("CAPGPU_SEGV_BACKTRACE");
  if (!e || !*e || !strcmp(e, "0")) return;
  static bool done = false;
  if (done) return;
  done = true;
  if (strcmp(e, "1") != 0) {  // a path: <path>.<pid>
    snprintf(g_segv_path, sizeof(g_segv_path), "%s.%d", e, (int)getpid());
  }
  struct sigaction sa;
  memset(&sa, 0, sizeof(sa));
  sa.sa_sigaction = segv_backtrace;
  sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
  sigemptyset(&sa.sa_mask);
  sigaction(SIGSEGV, &sa, &g_prev_segv);
  sigaction(SIGABRT, &sa, &g_prev_abrt);
}

int capgpu_init(const int* device_ids, int n_devices) {
  static std::mutex init_mu;
  std::lock_guard<std::mutex> ilk(init_mu);
  install_segv_backtrace();
  Runtime& R = rt();
  if (R.initialised.load()) return CAPGPU_OK;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0) {
    set_error("capgpu: no HIP device visible (%s); this library has no CPU fallback",
              e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    (void)hipGetLastError();
    return CAPGPU_ERR_NO_DEVICE;
  }
  std::vector<int> ids;
  if (device_ids && n_devices > 0) ids.assign(device_ids, device_ids + n_devices);
  else ids.push_back(0);
  // one context per listed device.  A device listed twice is refused unless CAPGPU_ALLOW_DUPLICATE_DEVICES=1 (tests
  // drive the multi-device paths on one GPU that way); CAPGPU_CONTEXTS_PER_DEVICE=k gives every listed device k
  // contexts, whose batches then overlap on the device (a batch's latency-bound launches and host transcript phases
  // run under the other's issue-bound kernels).
  const char* dup_env = getenv("CAPGPU_ALLOW_DUPLICATE_DEVICES");
  const bool allow_dup = dup_env && atoi(dup_env) != 0;
  for (size_t i = 0; i < ids.size(); i++) {
    if (ids[i] < 0 || ids[i] >= count) {
      set_error("capgpu: device id %d out of range (0..%d)", ids[i], count - 1);
      return CAPGPU_ERR_INVALID_ARG;
    }
    for (size_t j = 0; j < i && !allow_dup; j++)
      if (ids[j] == ids[i]) {
        set_error("capgpu_init: device %d listed twice", ids[i]);
        return CAPGPU_ERR_INVALID_ARG;
      }
  }
  std::vector<bool> secondary;  // the extra contexts of CAPGPU_CONTEXTS_PER_DEVICE: they share their device's tables
  {
    const char* pe = getenv("CAPGPU_CONTEXTS_PER_DEVICE");
    // default: ONE bound device gets four contexts - batches overlap on it, one's copies, latency-bound launches and
    // host transcript phases under the others' issue-bound kernels.  A dealt host batch is cut in two (+1.3 % .. +3.8 %
    // proofs/s at batch 256, +5 % at batch 64; plonk.hip: deal_max_parts); the gathered batches of coalesced single-proof
    // calls use all four (64 closed-loop callers: 1040-1090 proofs/s on two contexts, 1130-1186 on four, 1100-1110 on six or
    // eight).  Secondary contexts share their device's tables; several devices get one context each
    const int per = pe ? std::min(std::max(atoi(pe), 1), 8) : (ids.size() == 1 ? 4 : 1);
    std::vector<int> x;
    for (int d : ids)
      for (int k = 0; k < per; k++) {
        x.push_back(d);
        secondary.push_back(k > 0);
      }
    ids.swap(x);
  }
  if (ids.size() > 64) {
    set_error("capgpu_init: %zu contexts requested, 64 at most", ids.size());
    return CAPGPU_ERR_INVALID_ARG;
  }
  std::vector<std::unique_ptr<Context>> made;
  auto undo = [&] {
    for (auto& c : made) {
      (void)hipSetDevice(c->device);
      ntt_free_small_tables(&c->small);
      if (c->own_stream) hipStreamDestroy(c->own_stream);
    }
  };
  for (size_t i = 0; i < ids.size(); i++) {
    const int dev = ids[i];
    hipDeviceProp_t prop;
    hipError_t he = hipSetDevice(dev);
    if (he == hipSuccess) he = hipGetDeviceProperties(&prop, dev);
    if (he != hipSuccess) {
      undo();
      return hip_fail(he, "hipSetDevice / hipGetDeviceProperties");
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
      set_error("capgpu: device %d is %s; kernels are built for gfx950 (MI355X) only", dev, prop.gcnArchName);
      undo();
      return CAPGPU_ERR_NO_DEVICE;
    }
    std::unique_ptr<Context> c(new Context);
    c->slot = (int)i;
    c->device = dev;
    c->primary = !secondary[i];
    he = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (he != hipSuccess) {
      undo();
      return hip_fail(he, "hipStreamCreateWithFlags");
    }
    c->stream = c->own_stream;
    made.push_back(std::move(c));
    int rc = ntt_build_small_tables(&made.back()->small, made.back()->stream);
    if (rc) {
      undo();
      return hip_fail((hipError_t)rc, "ntt_build_small_tables");
    }
    made.back()->initialised = true;
  }
  // Peer access between every pair of bound devices: without it hipMemcpyPeerAsync - key / SRS replication, the scalar
  // slices and the 96-byte partials of a sharded MSM - is staged through host memory by the runtime.  A pair without
  // a peer path keeps working that way; capgpu_device_peer_info reports which pairs got the direct path.
  R.peer.clear();
  for (auto& a : made)
    for (auto& b : made) {
      if (a->device == b->device || R.peer.count({a->device, b->device})) continue;
      int can = 0, state = 0;
      if (hipSetDevice(a->device) == hipSuccess && hipDeviceCanAccessPeer(&can, a->device, b->device) == hipSuccess && can) {
        const hipError_t pe = hipDeviceEnablePeerAccess(b->device, 0);
        if (pe == hipSuccess || pe == hipErrorPeerAccessAlreadyEnabled) state = 1;
        (void)hipGetLastError();
      }
      R.peer[{a->device, b->device}] = state;
    }
  R.ctxs = std::move(made);
  R.rr.store(0);
  R.initialised.store(true, std::memory_order_release);
  (void)hipSetDevice(R.ctxs[0]->device);
  return CAPGPU_OK;
}

void capgpu_shutdown(void) {
  Runtime& R = rt();
  if (!R.initialised.load()) return;
  (void)capgpu_comm_destroy();
  plonk_reset_staging();
  {
    AllEntries all;
    for (auto& cp : R.ctxs) {
      Context& c = *cp;
      (void)hipSetDevice(c.device);
      (void)hipStreamSynchronize(c.stream);
      if (c.copy_stream) (void)hipStreamSynchronize(c.copy_stream);
      if (c.side_stream) (void)hipStreamSynchronize(c.side_stream);
    }
    {
      std::map<uint64_t, std::shared_ptr<ScalarSet>> sets;
      {
        std::lock_guard<std::mutex> lk(R.mu);
        sets.swap(R.scalar_sets);
        R.srs.clear();
        R.keys.clear();
      }
      sets.clear();  // frees the slices on their devices (the contexts and their streams still exist)
    }
    for (auto& cp : R.ctxs) {
      Context& c = *cp;
      (void)hipSetDevice(c.device);
      c.keys.clear();
      c.srs.clear();
      for (auto& kv : c.domains) ntt_free_domain(&kv.second);
      c.domains.clear();
      for (auto& kv : c.domains3) ntt3_free_domain(&kv.second);
      c.domains3.clear();
      ntt_free_small_tables(&c.small);
      for (Scratch* s : {&c.ntt_scratch, &c.msm_ws, &c.stage_a, &c.stage_b, &c.prove_ws, &c.gather}) {
        if (s->p) hipFree(s->p);
        scratch_count(c.device, 0, s->cap);
        s->p = nullptr;
        s->cap = 0;
      }
      if (c.pin_host) hipHostFree(c.pin_host);
      c.pin_host = nullptr;
      c.pin_cap = 0;
      c.pool.reset();
      c.prove_graphs.reset();
      if (c.tm0) hipEventDestroy(c.tm0);
      if (c.tm1) hipEventDestroy(c.tm1);
      c.tm0 = c.tm1 = nullptr;
      if (c.own_stream) hipStreamDestroy(c.own_stream);
      if (c.copy_stream) hipStreamDestroy(c.copy_stream);
      if (c.side_stream) hipStreamDestroy(c.side_stream);
      if (c.ev_fork) hipEventDestroy(c.ev_fork);
      if (c.ev_join) hipEventDestroy(c.ev_join);
      c.ev_fork = c.ev_join = nullptr;
      c.own_stream = c.stream = c.copy_stream = c.side_stream = nullptr;
      c.initialised = false;
    }
    R.initialised.store(false, std::memory_order_release);
  }
  tl_ctx = nullptr;
  tl_bound = -1;
  R.ctxs.clear();
}

int capgpu_device_count(int* count_out) {
  if (!count_out) return CAPGPU_ERR_INVALID_ARG;
  *count_out = rt().initialised.load() ? (int)num_contexts() : 0;
  return CAPGPU_OK;
}
int capgpu_context_count(int* count_out) { return capgpu_device_count(count_out); }
int capgpu_physical_device_count(int* count_out) {
  if (!count_out) return CAPGPU_ERR_INVALID_ARG;
  *count_out = 0;
  if (!rt().initialised.load()) return CAPGPU_OK;
  std::vector<int> seen;
  for (auto& c : rt().ctxs)
    if (std::find(seen.begin(), seen.end(), c->device) == seen.end()) seen.push_back(c->device);
  *count_out = (int)seen.size();
  return CAPGPU_OK;
}
int capgpu_set_device(int slot) {
  CAP_CHECK_INIT();
  if (slot < -1 || slot >= (int)num_contexts()) {
    set_error("capgpu_set_device: slot %d out of range (-1 .. %zu)", slot, num_contexts() - 1);
    return CAPGPU_ERR_INVALID_ARG;
  }
  tl_bound = slot;
  CAP_HIP(hipSetDevice(ctx().device));
  return CAPGPU_OK;
}
int capgpu_get_device(int* slot_out, int* hip_device_out) {
  CAP_CHECK_INIT();
  if (slot_out) *slot_out = tl_bound;
  if (hip_device_out) *hip_device_out = ctx().device;
  return CAPGPU_OK;
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

int capgpu_mem_info(uint64_t* free_bytes_out, uint64_t* total_bytes_out) {
  CAP_CHECK_INIT();
  Entry lk(ctx());
  size_t f = 0, t = 0;
  CAP_HIP(hipMemGetInfo(&f, &t));
  if (free_bytes_out) *free_bytes_out = (uint64_t)f;
  if (total_bytes_out) *total_bytes_out = (uint64_t)t;
  return CAPGPU_OK;
}

int capgpu_trim(uint64_t* bytes_released_out, int* contexts_busy_out) {
  CAP_CHECK_INIT();
  int busy = 0;
  const size_t released = trim_idle_contexts(-1, nullptr, &busy) + plonk_trim_staging();
  if (bytes_released_out) *bytes_released_out = (uint64_t)released;
  if (contexts_busy_out) *contexts_busy_out = busy;
  return CAPGPU_OK;
}
int capgpu_set_memory_limit(uint64_t scratch_bytes_per_device) {
  CAP_CHECK_INIT();
  g_scratch_limit.store((size_t)scratch_bytes_per_device);
  if (scratch_bytes_per_device) {  // already above it: give back what idle contexts hold (a running call keeps what it has)
    std::vector<int> seen;
    for (auto& cp : rt().ctxs) {
      const int d = cp->device;
      if (std::find(seen.begin(), seen.end(), d) != seen.end()) continue;
      seen.push_back(d);
      if (g_scratch_bytes[(size_t)d & 63].load() > scratch_bytes_per_device) (void)trim_idle_contexts(d, nullptr, nullptr);
    }
  }
  return CAPGPU_OK;
}
int capgpu_scratch_info(uint64_t* scratch_bytes_out, uint64_t* limit_out) {
  CAP_CHECK_INIT();
  if (scratch_bytes_out) *scratch_bytes_out = (uint64_t)g_scratch_bytes[(size_t)ctx().device & 63].load();
  if (limit_out) *limit_out = (uint64_t)g_scratch_limit.load();
  return CAPGPU_OK;
}
int capgpu_trace_enable(int on) {
  trace_enable(on != 0);
  return CAPGPU_OK;
}
int capgpu_trace_dump(const char* path, uint64_t* events_out) {
  if (!path) return CAPGPU_ERR_INVALID_ARG;
  const long n = trace_dump(path);
  if (n < 0) {
    set_error("capgpu_trace_dump: cannot write %s", path);
    return CAPGPU_ERR_INVALID_ARG;
  }
  if (events_out) *events_out = (uint64_t)n;
  return CAPGPU_OK;
}

int capgpu_malloc(void** dev_ptr_out, size_t bytes) {
  CAP_CHECK_INIT();
  if (!dev_ptr_out) return CAPGPU_ERR_INVALID_ARG;
  Entry lk(ctx());
  CAP_HIP(hipMalloc(dev_ptr_out, bytes ? bytes : 1));
  return CAPGPU_OK;
}
int capgpu_free(void* dev_ptr) {
  CAP_CHECK_INIT();
  Entry lk(ctx());
  CAP_HIP(hipStreamSynchronize(ctx().stream));
  CAP_HIP(hipFree(dev_ptr));
  return CAPGPU_OK;
}
int capgpu_memcpy_h2d(void* dev_dst, const void* host_src, size_t bytes) {
  CAP_CHECK_INIT();
  Entry lk(ctx());
  CAP_HIP(hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, ctx().stream));
  CAP_HIP(hipStreamSynchronize(ctx().stream));
  return CAPGPU_OK;
}
int capgpu_memcpy_d2h(void* host_dst, const void* dev_src, size_t bytes) {
  CAP_CHECK_INIT();
  Entry lk(ctx());
  CAP_HIP(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, ctx().stream));
  CAP_HIP(hipStreamSynchronize(ctx().stream));
  return CAPGPU_OK;
}
int capgpu_sync(void) {
  CAP_CHECK_INIT();
  Entry lk(ctx());
  CAP_HIP(hipStreamSynchronize(ctx().stream));
  return CAPGPU_OK;
}
int capgpu_sync_all(void) {
  CAP_CHECK_INIT();
  int prev = -1;
  (void)hipGetDevice(&prev);
  std::vector<int> seen;
  hipError_t e = hipSuccess;
  for (auto& cp : rt().ctxs) {
    if (std::find(seen.begin(), seen.end(), cp->device) != seen.end()) continue;
    seen.push_back(cp->device);
    if (e == hipSuccess) e = hipSetDevice(cp->device);
    if (e == hipSuccess) e = hipDeviceSynchronize();
  }
  if (prev >= 0) (void)hipSetDevice(prev);
  CAP_HIP(e);
  return CAPGPU_OK;
}
int capgpu_runtime_info(int* hip_runtime_version_out, int* hip_driver_version_out) {
  int v = 0;
  if (hip_runtime_version_out) {
    CAP_HIP(hipRuntimeGetVersion(&v));
    *hip_runtime_version_out = v;
  }
  if (hip_driver_version_out) {
    CAP_HIP(hipDriverGetVersion(&v));
    *hip_driver_version_out = v;
  }
  return CAPGPU_OK;
}
// device time of whatever the calling thread's context executes between the two calls: HIP events on ITS stream
// ("hipEvent around device section"); a pair per context, created on first use
int capgpu_timer_begin(void) {
  CAP_CHECK_INIT();
  Context& c = ctx();
  Entry lk(c);
  if (!c.tm0) {
    CAP_HIP(hipEventCreate(&c.tm0));
    CAP_HIP(hipEventCreate(&c.tm1));
  }
  // ONE measurement per context at a time: the event pair is shared, and _end waits outside the context lock.  The thread
  // that opened a measurement may restart it; another thread is refused until the owner's _end.
  const std::thread::id me = std::this_thread::get_id();
  if (c.tm_open && c.tm_owner != me) {
    set_error("capgpu_timer_begin: another thread has a measurement open on context %d (one per context at a time; "
              "bind the threads to different contexts with capgpu_set_device)", c.slot);
    return CAPGPU_ERR_INVALID_ARG;
  }
  CAP_HIP(hipEventRecord(c.tm0, c.stream));
  c.tm_open = true;
  c.tm_owner = me;
  return CAPGPU_OK;
}
int capgpu_timer_end(double* ms_out) {
  CAP_CHECK_INIT();
  Context& c = ctx();
  hipEvent_t e0, e1;
  {
    Entry lk(c);
    if (!c.tm0 || !ms_out || !c.tm_open || c.tm_owner != std::this_thread::get_id()) {
      set_error("capgpu_timer_end: no measurement of this thread is open on this context (or null output)");
      return CAPGPU_ERR_INVALID_ARG;
    }
    CAP_HIP(hipEventRecord(c.tm1, c.stream));
    e0 = c.tm0;
    e1 = c.tm1;
  }
  // not under the context lock: other threads may enqueue meanwhile (they cannot re-record the events: tm_open)
  hipError_t e = hipEventSynchronize(e1);
  float ms = 0;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  {
    Entry lk(c);
    c.tm_open = false;
  }
  CAP_HIP(e);
  *ms_out = (double)ms;
  return CAPGPU_OK;
}
int capgpu_set_stream(void* hip_stream) {
  CAP_CHECK_INIT();
  Context& c = ctx();
  Entry lk(c);
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
  const unsigned char* src = (const unsigned char*)bases;
  // packs [lo, lo + len) of the caller's array, uploads it to the current context and builds its window tables
  auto upload_range = [&](size_t lo, size_t len, std::shared_ptr<SrsEntry>* out) -> int {
    Context& c = ctx();
    std::vector<g1_affine> packed(len ? len : 1);
    for (size_t i = 0; i < len; i++) {
      memcpy(&packed[i], src + (lo + i) * stride_bytes, 64);
      if (stride_bytes == 72 && src[(lo + i) * stride_bytes + 64]) memset(&packed[i], 0, 64);  // infinity flag
    }
    DevTmp<g1_affine> d;
    CAP_HIP(d.alloc(len));
    CAP_HIP(hipMemcpyAsync(d, packed.data(), sizeof(g1_affine) * len, hipMemcpyHostToDevice, c.stream));
    if (!coords_montgomery && len) {
      size_t cnt = 2 * len;
      launch("fq_to_mont_kernel", fq_to_mont_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, c.stream,
             reinterpret_cast<fe*>(d.p), cnt);
    }
    return make_srs_entry(d, len, lo, out);
  };
  if (should_shard(n)) {
    AllEntries all;
    const size_t S = shard_slots().size();
    std::vector<std::shared_ptr<SrsEntry>> shards(S);
    int rc = for_each_context([&](size_t r) -> int {
      size_t lo, len;
      shard_range(n, S, r, &lo, &len);
      return upload_range(lo, len, &shards[r]);
    });
    if (rc) return rc;
    *handle_out = register_shards(shards, n);
    return CAPGPU_OK;
  }
  Context& c = ctx();
  Entry lk(c);
  std::shared_ptr<SrsEntry> e;
  int rc = upload_range(0, n, &e);
  if (rc) return rc;
  Runtime& R = rt();
  const uint64_t h = R.next_handle.fetch_add(1);
  {
    std::lock_guard<std::mutex> rlk(R.mu);
    SrsRecord rec;
    rec.full = e;
    rec.total_n = n;
    R.srs[h] = rec;
  }
  c.srs[h] = e;
  *handle_out = h;
  return take_launch_error();
}

static int srs_generate_common(int mode, const uint64_t a[4], const uint64_t b[4], size_t n, uint64_t* handle_out) {
  CAP_CHECK_INIT();
  if (!handle_out || !a || n == 0) {
    set_error("capgpu_srs_generate: bad argument");
    return CAPGPU_ERR_INVALID_ARG;
  }
  const fe am = Fr::to_mont(fe_from_u64x4(a));
  const fe bm = b ? Fr::to_mont(fe_from_u64x4(b)) : Fr::zero();
  // points [lo, lo + len) of the sequence, generated on the current context
  auto generate_range = [&](size_t lo, size_t len, g1_affine* d) -> int {
    Context& c = ctx();
    DevTmp<fe> d_tab, d_pw;
    if (mode == 0) {
      std::vector<fe> pw(64);
      fe x = am;
      for (int i = 0; i < 64; i++) {
        pw[i] = x;
        x = Fr::sqr(x);
      }
      CAP_HIP(d_pw.alloc(64));
      CAP_HIP(d_tab.alloc(len));
      CAP_HIP(hipMemcpyAsync(d_pw, pw.data(), sizeof(fe) * 64, hipMemcpyHostToDevice, c.stream));
      launch("fr_powers_kernel", fr_powers_kernel, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, c.stream, d_tab.p,
             len, (const fe*)d_pw.p, lo);
      CAP_HIP(hipStreamSynchronize(c.stream));
    }
    launch("srs_fixed_base_kernel", srs_fixed_base_kernel, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, c.stream,
           d, len, mode, (const fe*)d_tab.p, am, bm, lo);
    CAP_HIP(hipStreamSynchronize(c.stream));
    return take_launch_error();
  };
  if (should_shard(n)) {
    AllEntries all;
    const size_t S = shard_slots().size();
    std::vector<std::shared_ptr<SrsEntry>> shards(S);
    int rc = for_each_context([&](size_t r) -> int {
      size_t lo, len;
      shard_range(n, S, r, &lo, &len);
      DevTmp<g1_affine> d;
      CAP_HIP(d.alloc(len));
      int r2 = generate_range(lo, len, d);
      return r2 ? r2 : make_srs_entry(d, len, lo, &shards[r]);
    });
    if (rc) return rc;
    *handle_out = register_shards(shards, n);
    return CAPGPU_OK;
  }
  Context& c = ctx();
  Entry lk(c);
  DevTmp<g1_affine> d;
  CAP_HIP(d.alloc(n));
  int rc = generate_range(0, n, d);
  if (rc) return rc;
  return register_srs(d, n, handle_out);
}
int capgpu_srs_generate(const uint64_t tau[4], size_t n, uint64_t* handle_out) {
  return srs_generate_common(0, tau, nullptr, n, handle_out);
}
int capgpu_srs_generate_affine_seq(const uint64_t a[4], const uint64_t b[4], size_t n, uint64_t* handle_out) {
  if (!b) return CAPGPU_ERR_INVALID_ARG;
  return srs_generate_common(1, a, b, n, handle_out);
}

// universal_setup with hiding powers (src/proof/mod.rs:59-69 -> KZG10::setup): besides [tau^i] G the SRS carries
// powers_of_gamma_g = { i: [gamma tau^i] G } for degrees 0 ..= max_degree + 1 (n + 1 entries for n powers), which
// capgpu_srs_serialize writes and jf-plonk's trim reads when a reference-side consumer preprocesses under the stored file.
int capgpu_srs_generate_hiding(const uint64_t tau[4], const uint64_t gamma[4], size_t n, uint64_t* handle_out) {
  CAP_CHECK_INIT();
  if (!tau || !gamma || !handle_out || n == 0) {
    set_error("capgpu_srs_generate_hiding: bad argument");
    return CAPGPU_ERR_INVALID_ARG;
  }
  if (should_shard(n)) {
    set_error("capgpu_srs_generate_hiding: %zu points would be sharded over the devices; hiding powers belong to a commit-key "
              "sized SRS", n);
    return CAPGPU_ERR_INVALID_ARG;
  }
  Context& c = ctx();
  Entry lk(c);
  uint64_t h = 0;
  int rc = capgpu_srs_generate(tau, n, &h);
  if (rc) return rc;
  auto fail = [&](int code) {
    (void)hipStreamSynchronize(c.stream);
    c.srs.erase(h);
    std::lock_guard<std::mutex> rlk(rt().mu);
    rt().srs.erase(h);
    return code;
  };
  const size_t m = n + 1;
  DevTmp<fe> d_pw, d_tab;
  DevTmp<g1_affine> d_pts;
  hipError_t e = d_pw.alloc(64);
  if (e == hipSuccess) e = d_tab.alloc(m);
  if (e == hipSuccess) e = d_pts.alloc(m);
  if (e != hipSuccess) return fail(hip_fail(e, "hipMalloc"));
  std::vector<fe> pw(64);
  const fe tm = Fr::to_mont(fe_from_u64x4(tau)), gm = Fr::to_mont(fe_from_u64x4(gamma));
  fe x = tm;
  for (int i = 0; i < 64; i++) {
    pw[i] = x;
    x = Fr::sqr(x);
  }
  e = hipMemcpyAsync(d_pw, pw.data(), sizeof(fe) * 64, hipMemcpyHostToDevice, c.stream);
  if (e != hipSuccess) return fail(hip_fail(e, "hipMemcpyAsync"));
  const unsigned blocks = (unsigned)((m + 255) / 256);
  launch("fr_powers_kernel", fr_powers_kernel, dim3(blocks), dim3(256), 0, c.stream, d_tab.p, m, (const fe*)d_pw.p, (size_t)0);
  launch("fr_scale_kernel", fr_scale_kernel, dim3(blocks), dim3(256), 0, c.stream, d_tab.p, m, gm);
  launch("srs_fixed_base_kernel", srs_fixed_base_kernel, dim3(blocks), dim3(256), 0, c.stream, d_pts.p, m, 0,
         (const fe*)d_tab.p, tm, Fr::zero(), (size_t)0);
  std::vector<uint8_t> bytes(32 * m);
  if ((rc = params::compress_g1(d_pts, 0, m, bytes.data(), c.stream))) return fail(rc);
  if ((rc = take_launch_error())) return fail(rc);
  SrsEntry* E = find_srs_entry(h);
  E->gamma_deg.resize(m);
  for (size_t i = 0; i < m; i++) E->gamma_deg[i] = i;
  E->gamma_pts = std::move(bytes);
  *handle_out = h;
  return CAPGPU_OK;
}

int capgpu_srs_size(uint64_t handle, size_t* n_out) {
  CAP_CHECK_INIT();
  SrsRecord rec;
  int rc = srs_record(handle, &rec);
  if (rc) return rc;
  if (!n_out) return CAPGPU_ERR_INVALID_ARG;
  *n_out = rec.total_n;
  return CAPGPU_OK;
}
int capgpu_srs_shards(uint64_t handle, int* shards_out) {
  CAP_CHECK_INIT();
  SrsRecord rec;
  int rc = srs_record(handle, &rec);
  if (rc) return rc;
  int cnt = 1;
  if (rec.sharded()) {
    cnt = 0;
    for (auto& sh : rec.shards) cnt += sh ? 1 : 0;
  }
  if (shards_out) *shards_out = cnt;
  return CAPGPU_OK;
}

// copies entries [off, off + n) of a resident table back as arkworks-form points (current context)
static int download_range(const SrsEntry& e, size_t off, size_t n, g1_affine* pts) {
  Context& c = ctx();
  CAP_HIP(hipMemcpyAsync(pts, e.bases.ext + off, sizeof(g1_affine) * n, hipMemcpyDeviceToHost, c.stream));
  CAP_HIP(hipStreamSynchronize(c.stream));
  // the resident table is in the internal Montgomery form (x * 2^261); hand back arkworks' form (x * 2^256)
  for (size_t i = 0; i < n; i++) {
    if (G1::is_inf(pts[i])) continue;
    pts[i].x = Fq29::to_ext(Fq29::load(pts[i].x));
    pts[i].y = Fq29::to_ext(Fq29::load(pts[i].y));
  }
  return CAPGPU_OK;
}
int capgpu_srs_download(uint64_t handle, size_t offset, size_t n, void* out) {
  CAP_CHECK_INIT();
  SrsRecord rec;
  int rc = srs_record(handle, &rec);
  if (rc) return rc;
  if (offset + n > rec.total_n || !out) return CAPGPU_ERR_INVALID_ARG;
  g1_affine* pts = reinterpret_cast<g1_affine*>(out);
  if (rec.sharded()) {
    for (size_t r = 0; r < rec.shards.size(); r++) {
      if (!rec.shards[r]) continue;
      const SrsEntry& sh = *rec.shards[r];
      const size_t lo = std::max(offset, sh.range_lo), hi = std::min(offset + n, sh.range_lo + sh.bases.n);
      if (lo >= hi) continue;
      Context& c = *rt().ctxs[r];
      ScopedCtx sc(c);
      Entry lk(c);
      if ((rc = download_range(sh, lo - sh.range_lo, hi - lo, pts + (lo - offset)))) return rc;
    }
    return CAPGPU_OK;
  }
  Context& c = ctx();
  Entry lk(c);
  const MsmBases* B = nullptr;
  if ((rc = find_srs(handle, &B))) return rc;
  return download_range(*find_srs_entry(handle), offset, n, pts);
}
int capgpu_srs_free(uint64_t handle) {
  CAP_CHECK_INIT();
  Runtime& R = rt();
  {
    std::lock_guard<std::mutex> lk(R.mu);
    auto it = R.srs.find(handle);
    if (it == R.srs.end()) {
      set_error("capgpu: unknown SRS handle %llu", (unsigned long long)handle);
      return CAPGPU_ERR_BAD_HANDLE;
    }
    R.srs.erase(it);
  }
  // every context that holds the tables drains its stream before letting go of them
  for (auto& cp : R.ctxs) {
    Context& c = *cp;
    ScopedCtx sc(c);
    Entry lk(c);
    auto it = c.srs.find(handle);
    if (it == c.srs.end()) continue;
    (void)hipStreamSynchronize(c.stream);
    c.srs.erase(it);
  }
  return CAPGPU_OK;
}

// ---- MSM ------------------------------------------------------------------------------------------
int capgpu_msm_g1_dev(uint64_t srs_handle, size_t offset, const void* d_scalars, size_t scalar_stride, size_t n,
                      int count, int scalars_montgomery, void* d_out_xyz) {
  CAP_CHECK_INIT();
  SrsRecord rec;
  int rc = srs_record(srs_handle, &rec);
  if (rc) return rc;
  if (count < 0 || offset + n > rec.total_n || (!d_scalars && n) || !d_out_xyz) {
    set_error("capgpu_msm_g1: bad argument (offset %zu + n %zu vs SRS size %zu)", offset, n, rec.total_n);
    return CAPGPU_ERR_INVALID_ARG;
  }
  if (count == 0) return CAPGPU_OK;
  Context& c = ctx();
  if (rec.sharded()) {
    if (thread_entry_depth() > 0) {  // (would take the other contexts' locks out of order)
      set_error("capgpu_msm_g1_dev: a sharded SRS cannot be used from inside another entry point");
      return CAPGPU_ERR_INVALID_ARG;
    }
    AllEntries all;
    ScopedCtx sc(c);
    return msm_sharded(rec, c, offset, nullptr, (const fe*)d_scalars, scalar_stride, n, count, scalars_montgomery,
                       (g1_jac*)d_out_xyz);
  }
  Entry lk(c);
  const MsmBases* B = nullptr;
  if ((rc = find_srs(srs_handle, &B))) return rc;
  size_t need = msm_workspace_bytes(*B, n, (uint32_t)count);
  rc = scratch_reserve(c.msm_ws, need);
  if (rc) return rc;
  rc = msm_run(*B, offset, (const fe*)d_scalars, scalar_stride, 1, 0, n, (uint32_t)count, scalars_montgomery,
               (g1_jac*)d_out_xyz, c.msm_ws.p, c.msm_ws.cap, c.stream);
  if (rc) return hip_fail((hipError_t)rc, "msm_run");
  return take_launch_error();
}

// ---- scalars resident with their points ----------------------------------------------------------------
namespace {
// cuts [offset, offset + n) by the point ranges of `rec` and allocates one slice per range on its context; `fill`
// copies a slice's scalars into place (enqueued on the slice's context stream)
int make_scalar_set(uint64_t srs_handle, const SrsRecord& rec, size_t offset, size_t n, int count,
                    const std::function<int(Context&, ScalarSlice&)>& fill, uint64_t* handle_out) {
  auto set = std::make_shared<ScalarSet>();
  set->srs = srs_handle;
  set->offset = offset;
  set->n = n;
  set->count = count;
  if (rec.sharded()) {
    for (size_t slot = 0; slot < rec.shards.size(); slot++) {
      if (!rec.shards[slot]) continue;
      const SrsEntry& sh = *rec.shards[slot];
      const size_t lo = std::max(offset, sh.range_lo), hi = std::min(offset + n, sh.range_lo + sh.bases.n);
      if (lo >= hi) continue;
      ScalarSlice sl;
      sl.slot = (int)slot;
      sl.lo = lo;
      sl.len = hi - lo;
      set->slices.push_back(sl);
    }
  } else {
    ScalarSlice sl;
    sl.slot = ctx().slot;
    sl.lo = offset;
    sl.len = n;
    set->slices.push_back(sl);
  }
  for (ScalarSlice& sl : set->slices) {
    Context& c = *rt().ctxs[(size_t)sl.slot];
    ScopedCtx sc(c);
    CAP_HIP(hipMalloc(&sl.d, sizeof(fe) * (sl.len ? sl.len : 1) * (size_t)count));
    int rc = fill(c, sl);
    if (rc) return rc;
  }
  for (ScalarSlice& sl : set->slices) {  // the caller may reuse its buffer once this returns
    Context& c = *rt().ctxs[(size_t)sl.slot];
    ScopedCtx sc(c);
    CAP_HIP(hipStreamSynchronize(c.stream));
  }
  Runtime& R = rt();
  const uint64_t h = R.next_handle.fetch_add(1);
  std::lock_guard<std::mutex> lk(R.mu);
  R.scalar_sets[h] = set;
  *handle_out = h;
  return CAPGPU_OK;
}
int scalar_set_args(uint64_t srs_handle, size_t offset, const void* scalars, size_t stride, size_t n, int count,
                    uint64_t* handle_out, SrsRecord* rec) {
  int rc = srs_record(srs_handle, rec);
  if (rc) return rc;
  if (!handle_out || !scalars || n == 0 || count < 1 || offset + n > rec->total_n || (count > 1 && stride < n)) {
    set_error("capgpu_msm_scalars: bad argument (offset %zu + n %zu vs SRS size %zu, count %d, stride %zu)", offset, n,
              rec->total_n, count, stride);
    return CAPGPU_ERR_INVALID_ARG;
  }
  if (thread_entry_depth() > 0) {
    set_error("capgpu_msm_scalars: cannot be used from inside another entry point");
    return CAPGPU_ERR_INVALID_ARG;
  }
  return CAPGPU_OK;
}
}  // namespace

int capgpu_msm_scalars_upload(uint64_t srs_handle, size_t offset, const uint64_t* scalars, size_t scalar_stride, size_t n,
                              int count, uint64_t* scalars_handle_out) {
  CAP_CHECK_INIT();
  SrsRecord rec;
  int rc = scalar_set_args(srs_handle, offset, scalars, scalar_stride, n, count, scalars_handle_out, &rec);
  if (rc) return rc;
  AllEntries all;
  ScopedCtx sc(ctx());
  return make_scalar_set(srs_handle, rec, offset, n, count, [&](Context& c, ScalarSlice& sl) -> int {
    for (int k = 0; k < count; k++)  // host -> the slice's own device: no hop through a "home" GPU
      CAP_HIP(hipMemcpyAsync(sl.d + (size_t)k * sl.len, scalars + 4 * ((size_t)k * scalar_stride + (sl.lo - offset)),
                             sizeof(fe) * sl.len, hipMemcpyHostToDevice, c.stream));
    rt().shard_scalar_bytes += sizeof(fe) * sl.len * (size_t)count;
    return CAPGPU_OK;
  }, scalars_handle_out);
}

int capgpu_msm_scalars_scatter_dev(uint64_t srs_handle, size_t offset, const void* d_scalars, size_t scalar_stride,
                                   size_t n, int count, uint64_t* scalars_handle_out) {
  CAP_CHECK_INIT();
  SrsRecord rec;
  int rc = scalar_set_args(srs_handle, offset, d_scalars, scalar_stride, n, count, scalars_handle_out, &rec);
  if (rc) return rc;
  Context& home = ctx();
  AllEntries all;
  ScopedCtx sc(home);
  CAP_HIP(hipStreamSynchronize(home.stream));  // the scalars were written by work on the caller's context
  const fe* src = (const fe*)d_scalars;
  return make_scalar_set(srs_handle, rec, offset, n, count, [&](Context& c, ScalarSlice& sl) -> int {
    for (int k = 0; k < count; k++)
      CAP_HIP(copy_between(sl.d + (size_t)k * sl.len, c.device, src + (size_t)k * scalar_stride + (sl.lo - offset),
                           home.device, sizeof(fe) * sl.len, c.stream));
    if (c.slot != home.slot) rt().shard_scalar_bytes += sizeof(fe) * sl.len * (size_t)count;
    return CAPGPU_OK;
  }, scalars_handle_out);
}

int capgpu_msm_scalars_free(uint64_t scalars_handle) {
  CAP_CHECK_INIT();
  std::shared_ptr<ScalarSet> set;
  {
    Runtime& R = rt();
    std::lock_guard<std::mutex> lk(R.mu);
    auto it = R.scalar_sets.find(scalars_handle);
    if (it == R.scalar_sets.end()) {
      set_error("capgpu: unknown scalar-set handle %llu", (unsigned long long)scalars_handle);
      return CAPGPU_ERR_BAD_HANDLE;
    }
    set = it->second;
    R.scalar_sets.erase(it);
  }
  set.reset();  // frees the slices (each on its device, after its stream drained)
  return CAPGPU_OK;
}

int capgpu_msm_g1_resident(uint64_t srs_handle, uint64_t scalars_handle, int scalars_montgomery, void* d_out_xyz) {
  CAP_CHECK_INIT();
  SrsRecord rec;
  int rc = srs_record(srs_handle, &rec);
  if (rc) return rc;
  std::shared_ptr<ScalarSet> set;
  {
    Runtime& R = rt();
    std::lock_guard<std::mutex> lk(R.mu);
    auto it = R.scalar_sets.find(scalars_handle);
    if (it != R.scalar_sets.end()) set = it->second;
  }
  if (!set) {
    set_error("capgpu: unknown scalar-set handle %llu", (unsigned long long)scalars_handle);
    return CAPGPU_ERR_BAD_HANDLE;
  }
  if (set->srs != srs_handle || !d_out_xyz) {
    set_error("capgpu_msm_g1_resident: the scalar set belongs to SRS %llu", (unsigned long long)set->srs);
    return CAPGPU_ERR_INVALID_ARG;
  }
  Context& c = ctx();
  if (rec.sharded()) {
    if (thread_entry_depth() > 0) {
      set_error("capgpu_msm_g1_resident: a sharded SRS cannot be used from inside another entry point");
      return CAPGPU_ERR_INVALID_ARG;
    }
    AllEntries all;
    ScopedCtx sc(c);
    return msm_sharded(rec, c, set->offset, nullptr, nullptr, 0, set->n, set->count, scalars_montgomery,
                       (g1_jac*)d_out_xyz, set.get());
  }
  const ScalarSlice& sl = set->slices[0];
  if (sl.slot != c.slot && rt().ctxs[(size_t)sl.slot]->device != c.device) {
    set_error("capgpu_msm_g1_resident: the scalars live on context %d, this thread is on context %d", sl.slot, c.slot);
    return CAPGPU_ERR_INVALID_ARG;
  }
  return capgpu_msm_g1_dev(srs_handle, set->offset, sl.d, sl.len, sl.len, set->count, scalars_montgomery, d_out_xyz);
}

int capgpu_msm_shard_stats(uint64_t* scalar_bytes_out, uint64_t* partial_bytes_out, uint64_t* calls_out,
                           uint64_t* replications_out) {
  Runtime& R = rt();
  if (scalar_bytes_out) *scalar_bytes_out = R.shard_scalar_bytes.load();
  if (partial_bytes_out) *partial_bytes_out = R.shard_partial_bytes.load();
  if (calls_out) *calls_out = R.shard_calls.load();
  if (replications_out) *replications_out = R.replications.load();
  return CAPGPU_OK;
}

int capgpu_device_peer_info(int slot_a, int slot_b, int* access_out) {
  CAP_CHECK_INIT();
  const int S = (int)num_contexts();
  if (slot_a < 0 || slot_b < 0 || slot_a >= S || slot_b >= S || !access_out) {
    set_error("capgpu_device_peer_info: slots %d, %d out of range (0 .. %d)", slot_a, slot_b, S - 1);
    return CAPGPU_ERR_INVALID_ARG;
  }
  const int da = rt().ctxs[(size_t)slot_a]->device, db = rt().ctxs[(size_t)slot_b]->device;
  if (da == db) {
    *access_out = 2;
    return CAPGPU_OK;
  }
  auto it = rt().peer.find({da, db});
  *access_out = it != rt().peer.end() ? it->second : 0;
  return CAPGPU_OK;
}

int capgpu_msm_plan(uint64_t srs_handle, size_t n, int count, char* buf, size_t cap) {
  CAP_CHECK_INIT();
  SrsRecord rec;
  int rc = srs_record(srs_handle, &rec);
  if (rc) return rc;
  if (!buf || cap == 0 || count < 1 || n > rec.total_n) return CAPGPU_ERR_INVALID_ARG;
  if (rec.sharded()) {  // the plan of the first shard's part, prefixed with the shard count
    size_t S = 0;
    const SrsEntry* first = nullptr;
    for (auto& sh : rec.shards)
      if (sh) {
        S++;
        if (!first) first = sh.get();
      }
    int len = snprintf(buf, cap, "shards=%zu ", S);
    if (len > 0 && (size_t)len < cap)
      msm_plan_describe(first->bases, std::min(first->bases.n, (n + S - 1) / S), (uint32_t)count, buf + len, cap - len);
    return CAPGPU_OK;
  }
  Context& c = ctx();
  Entry lk(c);
  const MsmBases* B = nullptr;
  if ((rc = find_srs(srs_handle, &B))) return rc;
  msm_plan_describe(*B, n, (uint32_t)count, buf, cap);
  return CAPGPU_OK;
}

int capgpu_msm_g1_batch(uint64_t srs_handle, const size_t* offsets, const uint64_t* const* scalars, const size_t* ns,
                        int count, uint64_t* out_xyz) {
  CAP_CHECK_INIT();
  if (count < 0 || (count && (!offsets || !scalars || !ns || !out_xyz))) return CAPGPU_ERR_INVALID_ARG;
  if (count == 0) return CAPGPU_OK;
  SrsRecord rec;
  int rc = srs_record(srs_handle, &rec);
  if (rc) return rc;
  if (rec.sharded()) {
    Context& c = ctx();
    if (thread_entry_depth() > 0) {
      set_error("capgpu_msm_g1: a sharded SRS cannot be used from inside another entry point");
      return CAPGPU_ERR_INVALID_ARG;
    }
    AllEntries all;
    ScopedCtx sc(c);
    if ((rc = scratch_reserve(c.stage_b, sizeof(g1_jac)))) return rc;
    for (int g = 0; g < count; g++) {
      if (offsets[g] + ns[g] > rec.total_n || (!scalars[g] && ns[g])) {
        set_error("capgpu_msm_g1: bad argument (offset %zu + n %zu vs SRS size %zu)", offsets[g], ns[g], rec.total_n);
        return CAPGPU_ERR_INVALID_ARG;
      }
      if ((rc = msm_sharded(rec, c, offsets[g], scalars[g], nullptr, ns[g], ns[g], 1, 0, (g1_jac*)c.stage_b.p)))
        return rc;
      CAP_HIP(hipMemcpyAsync(out_xyz + 12 * (size_t)g, c.stage_b.p, sizeof(g1_jac), hipMemcpyDeviceToHost, c.stream));
      CAP_HIP(hipStreamSynchronize(c.stream));
    }
    return CAPGPU_OK;
  }
  // host buffers, logical handle: an unbound thread's call goes to whichever context is free
  Context& c = pick_context();
  ScopedCtx sc(c);
  Entry lk(c);
  // equal (offset, n) entries run as one batched launch; otherwise one launch each
  bool uniform = true;
  for (int i = 1; i < count; i++) uniform = uniform && offsets[i] == offsets[0] && ns[i] == ns[0];
  int groups = uniform ? 1 : count;
  for (int g = 0; g < groups; g++) {
    int first = uniform ? 0 : g, cnt = uniform ? count : 1;
    size_t n = ns[first];
    rc = scratch_reserve(c.stage_a, sizeof(fe) * (n ? n : 1) * cnt + 96 * cnt);
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

// KZG commitment of a polynomial given by its VALUES on the 2^log_n-th roots of unity (+ the two blinders of jf-plonk's
// wire polynomials): an MSM on the Lagrange-form commit key (lagrange.hip), the form round 1 of the prover uses
int capgpu_msm_g1_lagrange(uint64_t srs_handle, uint32_t log_n, const uint64_t* scalars, size_t count,
                           int scalars_montgomery, uint64_t out_xyz[12]) {
  CAP_CHECK_INIT();
  const size_t n = (size_t)1 << (log_n & 31);
  if (!scalars || !out_xyz || log_n > 26 || count == 0 || count > n + 3) {
    set_error("capgpu_msm_g1_lagrange: bad argument (1 .. 2^log_n + 3 scalars, log_n <= 26)");
    return CAPGPU_ERR_INVALID_ARG;
  }
  Context& c = pick_context();
  ScopedCtx sc(c);
  Entry lk(c);
  const MsmBases* L = nullptr;
  int rc = find_lagrange(srs_handle, log_n, &L);
  if (rc) return rc;
  if ((rc = scratch_reserve(c.stage_a, sizeof(fe) * count + sizeof(g1_jac) + 256))) return rc;
  if ((rc = scratch_reserve(c.msm_ws, msm_workspace_bytes(*L, count, 1)))) return rc;
  g1_jac* d_out = (g1_jac*)c.stage_a.p;
  fe* d_sc = (fe*)((char*)c.stage_a.p + 256);
  CAP_HIP(hipMemcpyAsync(d_sc, scalars, sizeof(fe) * count, hipMemcpyHostToDevice, c.stream));
  rc = msm_run(*L, 0, d_sc, 0, 1, 0, count, 1, scalars_montgomery ? 1 : 0, d_out, c.msm_ws.p, c.msm_ws.cap, c.stream);
  if (rc) return hip_fail((hipError_t)rc, "msm_run");
  CAP_HIP(hipMemcpyAsync(out_xyz, d_out, sizeof(g1_jac), hipMemcpyDeviceToHost, c.stream));
  CAP_HIP(hipStreamSynchronize(c.stream));
  return take_launch_error();
}

int capgpu_g1_sum(const uint64_t* points_xyz, size_t n, uint64_t out_xyz[12]) {
  CAP_CHECK_INIT();
  if ((!points_xyz && n) || !out_xyz) return CAPGPU_ERR_INVALID_ARG;
  Context& c = ctx();
  Entry lk(c);
  int rc = scratch_reserve(c.stage_a, sizeof(g1_jac) * (n + 1));
  if (rc) return rc;
  g1_jac* d = (g1_jac*)c.stage_a.p;
  if (n) CAP_HIP(hipMemcpyAsync(d + 1, points_xyz, sizeof(g1_jac) * n, hipMemcpyHostToDevice, c.stream));
  launch("g1_sum_kernel", g1_sum_kernel, dim3(1), dim3(64), 0, c.stream, (const g1_jac*)(d + 1), n, d);
  CAP_HIP(hipMemcpyAsync(out_xyz, d, sizeof(g1_jac), hipMemcpyDeviceToHost, c.stream));
  CAP_HIP(hipStreamSynchronize(c.stream));
  return take_launch_error();
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
  Entry lk(c);
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
  Context& c = pick_context();
  ScopedCtx sc(c);
  Entry lk(c);
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
  Entry lk(c);
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

int capgpu_ubench_issue_rates(double* rates_out, int count) {
  CAP_CHECK_INIT();
  if (!rates_out || count < 1) return CAPGPU_ERR_INVALID_ARG;
  Context& c = ctx();
  Entry lk(c);
  hipDeviceProp_t prop;
  CAP_HIP(hipGetDeviceProperties(&prop, c.device));
  // CAPGPU_UBENCH_ITERS: length of one measurement launch (1000 iterations of 64 instructions ~ 0.4-0.7 ms)
  const char* ie = getenv("CAPGPU_UBENCH_ITERS");
  const int blocks = prop.multiProcessorCount * 8, iters = ie && atoi(ie) > 0 ? atoi(ie) : 1000;
  DevTmp<uint64_t> d;
  CAP_HIP(d.alloc((size_t)blocks * 256));
  CAP_HIP(hipMemsetAsync(d, 0x5a, sizeof(uint64_t) * blocks * 256, c.stream));
  hipEvent_t e0, e1;
  CAP_HIP(hipEventCreate(&e0));
  CAP_HIP(hipEventCreate(&e1));
  using Kern = void (*)(uint64_t*, int);
  const Kern kerns[10] = {ubench_issue_kernel<0>, ubench_issue_kernel<1>, ubench_issue_kernel<2>, ubench_issue_kernel<3>,
                          ubench_issue_kernel<4>, ubench_issue_kernel<5>, ubench_issue_kernel<6>, ubench_issue_kernel<7>,
                          ubench_issue_kernel<8>, ubench_issue_kernel<8>};
  for (int k = 0; k < count && k < 10; k++) {
    // 52 KiB of dynamic LDS per 4-wave workgroup: three workgroups per CU = three waves per SIMD
    const size_t lds = k == 9 ? 52 * 1024 : 0;
    hipLaunchKernelGGL(kerns[k], dim3(blocks), dim3(256), lds, c.stream, d.p, 10);  // warm-up
    double best = 0;
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(e0, c.stream);
      hipLaunchKernelGGL(kerns[k], dim3(blocks), dim3(256), lds, c.stream, d.p, iters);
      hipEventRecord(e1, c.stream);
      CAP_HIP(hipEventSynchronize(e1));
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      const double rate = (double)blocks * 256 * iters * 64 / (ms * 1e-3);
      if (rate > best) best = rate;
    }
    rates_out[k] = best;
  }
  for (int k = 10; k < count; k++) rates_out[k] = 0;
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  (void)hipGetLastError();
  return CAPGPU_OK;
}

// the profiler is per context (contexts launch concurrently); the ABI reports the sums over all of them
int capgpu_profile_enable(int on) {
  for (auto& cp : rt().ctxs) {
    Entry lk(*cp);
    cp->prof.on = on != 0;
  }
  return CAPGPU_OK;
}
int capgpu_profile_reset(void) {
  for (auto& cp : rt().ctxs) {
    ScopedCtx sc(*cp);
    Entry lk(*cp);
    cp->prof.reset();
  }
  return CAPGPU_OK;
}
static std::map<std::string, Profiler::Stat> profile_totals() {
  std::map<std::string, Profiler::Stat> tot;
  for (auto& cp : rt().ctxs) {
    ScopedCtx sc(*cp);
    Entry lk(*cp);
    for (const auto& kv : cp->prof.stats()) {
      tot[kv.first].ms += kv.second.ms;
      tot[kv.first].launches += kv.second.launches;
    }
  }
  return tot;
}
int capgpu_profile_get(const char* name, double* total_ms_out, uint64_t* launches_out) {
  if (!name) return CAPGPU_ERR_INVALID_ARG;
  const auto st = profile_totals();
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
  if (!buf || cap == 0) return CAPGPU_ERR_INVALID_ARG;
  size_t o = 0;
  buf[0] = 0;
  for (const auto& kv : profile_totals()) {
    int w = snprintf(buf + o, cap - o, "%s %.6f %llu\n", kv.first.c_str(), kv.second.ms,
                     (unsigned long long)kv.second.launches);
    if (w < 0 || (size_t)w >= cap - o) break;
    o += (size_t)w;
  }
  return CAPGPU_OK;
}

}  // extern "C"
