// Multi-GPU exchange step of the point-range-sharded MSM, inside the C ABI, for jobs that run ONE
// PROCESS PER GPU.  (A single process that drives several GPUs needs no communicator: capgpu_init binds them all and the
// sharded MSM of capgpu.hip moves the partials with peer copies.)
//
// Rank g keeps the window table of its point range resident, receives the matching scalar slice, and runs the whole
// Pippenger locally down to ONE point per MSM.  The single exchange step is an all-gather of those 96-byte Jacobian
// points over RCCL (xGMI), enqueued on the library stream straight from device memory, followed by G - 1 group
// additions on the device.  RCCL has no elliptic-curve reduction operator, so there is no all-reduce; exchanging bucket
// arrays would move 2^c * 96 B per window for nothing.  The message is latency-bound (96 B x count per rank): xGMI
// bandwidth plays no role.
//
// Failure handling: every rank's payload carries a status word behind its points.  A rank whose local step failed still
// takes part in the all-gather (points at infinity, status = its error code), so nobody is left waiting, and every rank
// returns an error afterwards.  A rank that never arrives is caught by a deadline (CAPGPU_COMM_TIMEOUT_MS, default
// 60 s): on the communicator's creation (which runs on a helper thread, RCCL blocks inside it) and on the wait behind
// each exchange (the communicator is aborted) - the call fails instead of hanging with the context lock held.
//
// RCCL is loaded at capgpu_comm_init time (dlopen), so the library itself loads - and the single-GPU path runs - on
// machines without it.  If the process already holds an RCCL (PyTorch ships one), that copy is used.
#include <dlfcn.h>

#include <string>
#include <rccl/rccl.h>  // types only: every function is looked up at run time
#include <stdio.h>
#include <string.h>

#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>

#include "context.hpp"
#include "curve29.hpp"
#include "launch.hpp"

namespace cap {
namespace {

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;  // optional
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

struct Comm {
  Rccl api;
  ncclComm_t comm = nullptr;
  bool loopback = false;  // test communicator: `world` ranks played one after the other on this device
  int rank = 0, world = 1;
  int slot = 0;               // the context the communicator belongs to
  g1_jac* d_gather = nullptr;  // world * (count + 1) slots: every rank's points + its status slot
  size_t gather_cap = 0;
  g1_jac* d_send = nullptr;  // count + 1 slots
  size_t send_cap = 0;
  uint32_t* h_status = nullptr;  // pinned, `world` words
  size_t status_cap = 0;
  bool shard_prover = false;
};
Comm& comm() {
  static Comm c;
  return c;
}

bool comm_debug() {
  const char* e = getenv("CAPGPU_COMM_DEBUG");
  return e && atoi(e) != 0;
}
#define COMM_DBG(...)                                 \
  do {                                                \
    if (comm_debug()) {                               \
      fprintf(stderr, "[capgpu_comm] " __VA_ARGS__);  \
      fputc('\n', stderr);                            \
      fflush(stderr);                                 \
    }                                                 \
  } while (0)

long timeout_ms() {
  const char* e = getenv("CAPGPU_COMM_TIMEOUT_MS");
  long x = e ? atol(e) : 60000;
  return x > 0 ? x : 60000;
}

int load_rccl(Rccl& r) {
  if (r.lib) return CAPGPU_OK;
  const char* override_path = getenv("CAPGPU_RCCL_LIBRARY");
  void* h = nullptr;
  if (override_path) h = dlopen(override_path, RTLD_NOW | RTLD_LOCAL);
  // The RCCL that sits BESIDE the HIP runtime serving this process first (round 6): a process may hold two ROCm stacks -
  // PyTorch's wheel bundles libamdhip64 + librccl of ROCm 7.0.2 next to /opt/rocm's 7.2 - and which libamdhip64.so.7 serves
  // everything is decided by load order; an RCCL from the other stack fails in ncclCommInitRank ("unhandled cuda error":
  // seen with the library loaded before torch).  dladdr on a runtime entry point names the file, its directory the stack.
  if (!h) {
    Dl_info info;
    if (dladdr(reinterpret_cast<void*>(&hipGetDeviceCount), &info) && info.dli_fname) {
      std::string dir(info.dli_fname);
      const size_t slash = dir.rfind('/');
      if (slash != std::string::npos) {
        dir.resize(slash + 1);
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
          if (h) break;
          h = dlopen((dir + name).c_str(), RTLD_NOW | RTLD_LOCAL);
        }
      }
    }
  }
  // a copy already in the process (same soname) next: two RCCLs in one process would each set up their own IPC state
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!h) {
    set_error("capgpu_comm: cannot load librccl.so.1 (%s)", dlerror());
    return CAPGPU_ERR_NO_DEVICE;
  }
  r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
  r.CommAbort = (decltype(r.CommAbort))dlsym(h, "ncclCommAbort");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
  r.AllGather = (decltype(r.AllGather))dlsym(h, "ncclAllGather");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
  if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.GetErrorString) {
    set_error("capgpu_comm: librccl.so.1 lacks an expected entry point");
    dlclose(h);
    return CAPGPU_ERR_NO_DEVICE;
  }
  r.lib = h;
  return CAPGPU_OK;
}
int rccl_fail(ncclResult_t e, const char* what) {
  Comm& c = comm();
  set_error("capgpu_comm: RCCL error %d (%s) in %s", (int)e, c.api.GetErrorString ? c.api.GetErrorString(e) : "?", what);
  return CAPGPU_ERR_HIP;
}

// gives up the communicator after a deadline was missed: abort (never the collective ncclCommDestroy) and forget it
void abandon_comm() {
  Comm& c = comm();
  if (c.comm) {
    if (c.api.CommAbort) c.api.CommAbort(c.comm);
    c.comm = nullptr;
  }
  c.rank = 0;
  c.world = 1;
  c.shard_prover = false;
}

// waits for the stream with a deadline instead of hipStreamSynchronize: a peer that never enters the collective must
// not freeze this process with its context lock held
int wait_stream(hipStream_t s, const char* what) {
  const auto t0 = std::chrono::steady_clock::now();
  for (int spin = 0;; spin++) {
    hipError_t e = hipStreamQuery(s);
    if (e == hipSuccess) return CAPGPU_OK;
    if (e != hipErrorNotReady) return hip_fail(e, "hipStreamQuery");
    if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(timeout_ms())) {
      abandon_comm();
      set_error("capgpu_comm: %s did not complete within %ld ms (a rank is missing?); communicator aborted", what,
                timeout_ms());
      return CAPGPU_ERR_COMM;
    }
    if (spin < 2000) std::this_thread::yield();
    else std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
}

// out[k] = sum over ranks r of all[r * stride + k]   (one wavefront per MSM; lanes stride over the ranks)
__global__ __launch_bounds__(64) void g1_sum_ranks_kernel(const g1_jac* __restrict__ all, uint32_t world, uint32_t stride,
                                                          g1_jac* __restrict__ out) {
  const uint32_t k = blockIdx.x;
  g1x acc = G1L::inf();
  for (uint32_t r = threadIdx.x; r < world; r += 64) {
    g1_jac p = all[(size_t)r * stride + k];
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
  if (threadIdx.x == 0) out[k] = G1L::to_jac_ext(acc);
}

}  // namespace

void g1_sum_ranks(const g1_jac* d_all, uint32_t world, uint32_t stride, uint32_t count, g1_jac* d_out, hipStream_t s) {
  if (count == 0) return;
  launch("g1_sum_ranks", g1_sum_ranks_kernel, dim3(count), dim3(64), 0, s, d_all, world, stride, d_out);
}

static bool comm_exists() { return comm().comm != nullptr || comm().loopback; }
bool comm_active() { return comm_exists() && comm().world > 1; }
bool comm_shard_prover() { return comm_exists() && comm().shard_prover && comm().slot == ctx().slot; }
int comm_shard_slot() { return comm_exists() && comm().shard_prover ? comm().slot : -1; }
bool comm_loopback() { return comm().loopback; }
void comm_loopback_rank(int r) {
  if (comm().loopback && r >= 0 && r < comm().world) comm().rank = r;
}
int comm_rank() { return comm().rank; }
int comm_world() { return comm().world; }

// d_points: this rank's `count` partial results (device, written by work already enqueued on `s`).  On return
// d_points[k] holds the sum over all ranks of their k-th point, on every rank, and every rank knows whether any of them
// failed.  (Loopback: the ranks 0 .. world - 2 only deposit their payload; the call of the last rank finishes the job.)
int comm_allgather_sum(g1_jac* d_points, uint32_t count, hipStream_t s, int local_rc) {
  Comm& c = comm();
  if (!comm_exists()) {
    set_error("capgpu_comm: no communicator (call capgpu_comm_init first)");
    return CAPGPU_ERR_NOT_INITIALISED;
  }
  if (c.slot != ctx().slot) {
    set_error("capgpu_comm: the communicator belongs to context %d, this call runs on context %d", c.slot, ctx().slot);
    return CAPGPU_ERR_INVALID_ARG;
  }
  const std::string local_msg = local_rc ? last_error() : "";
  const size_t stride = (size_t)count + 1;  // points + status slot
  if (stride > c.send_cap || (size_t)c.world * stride > c.gather_cap || (size_t)c.world > c.status_cap) {
    CAP_HIP(hipStreamSynchronize(s));
    if (c.d_gather) CAP_HIP(hipFree(c.d_gather));
    if (c.d_send) CAP_HIP(hipFree(c.d_send));
    if (c.h_status) CAP_HIP(hipHostFree(c.h_status));
    c.d_gather = c.d_send = nullptr;
    c.h_status = nullptr;
    c.gather_cap = c.send_cap = c.status_cap = 0;
    CAP_HIP(hipMalloc(&c.d_send, sizeof(g1_jac) * stride));
    c.send_cap = stride;
    CAP_HIP(hipMalloc(&c.d_gather, sizeof(g1_jac) * (size_t)c.world * stride));
    c.gather_cap = (size_t)c.world * stride;
    CAP_HIP(hipHostMalloc((void**)&c.h_status, sizeof(uint32_t) * (size_t)c.world, hipHostMallocDefault));
    c.status_cap = (size_t)c.world;
  }
  // payload: the points (infinity, Z = 0, when the local step failed) and the status word
  if (local_rc) CAP_HIP(hipMemsetAsync(c.d_send, 0, sizeof(g1_jac) * stride, s));
  else {
    if (count) CAP_HIP(hipMemcpyAsync(c.d_send, d_points, sizeof(g1_jac) * count, hipMemcpyDeviceToDevice, s));
    CAP_HIP(hipMemsetAsync(c.d_send + count, 0, sizeof(g1_jac), s));
  }
  if (local_rc) CAP_HIP(hipMemsetD32Async((hipDeviceptr_t)(c.d_send + count), (int)(uint32_t)local_rc, 1, s));
  if (c.loopback) {
    CAP_HIP(hipMemcpyAsync(c.d_gather + (size_t)c.rank * stride, c.d_send, sizeof(g1_jac) * stride,
                           hipMemcpyDeviceToDevice, s));
    if (c.rank != c.world - 1) return local_rc;
  } else {
    ncclResult_t e = c.api.AllGather(c.d_send, c.d_gather, sizeof(g1_jac) * stride, ncclUint8, c.comm, s);
    if (e != ncclSuccess) return rccl_fail(e, "ncclAllGather");
  }
  g1_sum_ranks(c.d_gather, (uint32_t)c.world, (uint32_t)stride, count, d_points, s);
  CAP_HIP(hipMemcpy2DAsync(c.h_status, sizeof(uint32_t), c.d_gather + count, sizeof(g1_jac) * stride, sizeof(uint32_t),
                           (size_t)c.world, hipMemcpyDeviceToHost, s));
  int rc = c.loopback ? (hipStreamSynchronize(s) == hipSuccess ? CAPGPU_OK : CAPGPU_ERR_HIP)
                      : wait_stream(s, "the all-gather of MSM partials");
  if (rc) return rc;
  if (local_rc) {
    set_error("%s", local_msg.c_str());
    return local_rc;
  }
  for (int r = 0; r < c.world; r++)
    if (c.h_status[r]) {
      set_error("capgpu_comm: rank %d of %d failed its part of the sharded MSM (code %d)", r, c.world,
                (int)c.h_status[r]);
      return CAPGPU_ERR_COMM;
    }
  return CAPGPU_OK;
}

}  // namespace cap

using namespace cap;

extern "C" {

int capgpu_comm_unique_id(uint8_t id_out[128]) {
  CAP_CHECK_INIT();
  if (!id_out) return CAPGPU_ERR_INVALID_ARG;
  Entry lk(ctx());
  Comm& c = comm();
  int rc = load_rccl(c.api);
  if (rc) return rc;
  ncclUniqueId id;
  ncclResult_t e = c.api.GetUniqueId(&id);
  if (e) return rccl_fail(e, "ncclGetUniqueId");
  static_assert(sizeof(id.internal) == 128, "ncclUniqueId is 128 bytes");
  memcpy(id_out, id.internal, 128);
  return CAPGPU_OK;
}

int capgpu_comm_init(int rank, int world, const uint8_t id[128]) {
  CAP_CHECK_INIT();
  if (!id || world < 1 || rank < 0 || rank >= world) {
    set_error("capgpu_comm_init: bad argument (rank %d of %d)", rank, world);
    return CAPGPU_ERR_INVALID_ARG;
  }
  Comm& c = comm();
  {
    Entry lk(ctx());
    if (comm_exists()) {
      set_error("capgpu_comm_init: a communicator already exists (capgpu_comm_destroy first)");
      return CAPGPU_ERR_INVALID_ARG;
    }
    int rc = load_rccl(c.api);
    if (rc) return rc;
  }
  // Creating the communicator is collective and RCCL blocks inside it until every rank has arrived (its non-blocking
  // mode blocks in the bootstrap all the same - probed on this image, tools/gpu_comm_timeout_probe.py).  So the call
  // runs on a helper thread and this one waits for it with a deadline, the context lock NOT held: a rank that never
  // shows up costs the helper thread (it stays parked inside RCCL; should the world complete later it destroys the
  // communicator it got), not the process - every other entry point keeps working.
  struct Pending {
    std::mutex mu;
    std::condition_variable cv;
    bool finished = false, abandoned = false;
    ncclResult_t result = ncclSuccess;
    ncclComm_t nc = nullptr;
  };
  auto pend = std::make_shared<Pending>();
  {
    const int device = ctx().device;
    ncclUniqueId uid;
    memcpy(uid.internal, id, 128);
    const Rccl api = c.api;
    std::thread([pend, api, uid, world, rank, device] {
      (void)hipSetDevice(device);
      ncclComm_t nc = nullptr;
      COMM_DBG("ncclCommInitRank rank %d of %d", rank, world);
      ncclResult_t e = api.CommInitRank(&nc, world, uid, rank);
      COMM_DBG("... returned %d", (int)e);
      std::lock_guard<std::mutex> lk(pend->mu);
      pend->result = e;
      pend->nc = nc;
      pend->finished = true;
      if (pend->abandoned && e == ncclSuccess && nc) {
        if (api.CommAbort) api.CommAbort(nc);
        else api.CommDestroy(nc);
      }
      pend->cv.notify_all();
    }).detach();
  }
  ncclComm_t nc = nullptr;
  {
    std::unique_lock<std::mutex> plk(pend->mu);
    if (!pend->cv.wait_for(plk, std::chrono::milliseconds(timeout_ms()), [&] { return pend->finished; })) {
      pend->abandoned = true;
      set_error("capgpu_comm_init: rank %d of %d: the other ranks did not arrive within %ld ms", rank, world,
                timeout_ms());
      return CAPGPU_ERR_COMM;
    }
    if (pend->result != ncclSuccess) return rccl_fail(pend->result, "ncclCommInitRank");
    nc = pend->nc;
  }
  Entry lk(ctx());
  if (comm_exists()) {  // a concurrent capgpu_comm_init won the race
    c.api.CommDestroy(nc);
    set_error("capgpu_comm_init: a communicator already exists (capgpu_comm_destroy first)");
    return CAPGPU_ERR_INVALID_ARG;
  }
  c.comm = nc;
  c.loopback = false;
  c.rank = rank;
  c.world = world;
  c.slot = ctx().slot;
  return CAPGPU_OK;
}

int capgpu_comm_init_loopback(int world) {
  CAP_CHECK_INIT();
  if (world < 1 || world > 4096) {
    set_error("capgpu_comm_init_loopback: bad world size %d", world);
    return CAPGPU_ERR_INVALID_ARG;
  }
  Entry lk(ctx());
  Comm& c = comm();
  if (comm_exists()) {
    set_error("capgpu_comm_init_loopback: a communicator already exists (capgpu_comm_destroy first)");
    return CAPGPU_ERR_INVALID_ARG;
  }
  c.comm = nullptr;
  c.loopback = true;
  c.rank = 0;
  c.world = world;
  c.slot = ctx().slot;
  return CAPGPU_OK;
}
int capgpu_comm_loopback_set_rank(int rank) {
  CAP_CHECK_INIT();
  Entry lk(ctx());
  Comm& c = comm();
  if (!c.loopback || rank < 0 || rank >= c.world) {
    set_error("capgpu_comm_loopback_set_rank: no loopback communicator, or rank %d out of range", rank);
    return CAPGPU_ERR_INVALID_ARG;
  }
  c.rank = rank;
  return CAPGPU_OK;
}

int capgpu_comm_destroy(void) {
  CAP_CHECK_INIT();
  Comm& c = comm();
  if (!comm_exists() && !c.d_gather) return CAPGPU_OK;
  Context& owner = *rt().ctxs[(size_t)c.slot < num_contexts() ? c.slot : 0];
  ScopedCtx sc(owner);
  Entry lk(owner);
  (void)hipStreamSynchronize(owner.stream);
  if (c.comm) c.api.CommDestroy(c.comm);
  c.comm = nullptr;
  c.loopback = false;
  c.rank = 0;
  c.world = 1;
  c.shard_prover = false;
  if (c.d_gather) hipFree(c.d_gather);
  if (c.d_send) hipFree(c.d_send);
  if (c.h_status) hipHostFree(c.h_status);
  c.d_gather = c.d_send = nullptr;
  c.h_status = nullptr;
  c.gather_cap = c.send_cap = c.status_cap = 0;
  return CAPGPU_OK;
}

int capgpu_comm_info(int* rank_out, int* world_out) {
  Comm& c = comm();
  if (rank_out) *rank_out = comm_exists() ? c.rank : 0;
  if (world_out) *world_out = comm_exists() ? c.world : 0;
  return CAPGPU_OK;
}

int capgpu_msm_g1_sharded_dev(uint64_t srs_handle, size_t offset, const void* d_scalars, size_t scalar_stride,
                              size_t n_local, int count, int scalars_montgomery, void* d_out_xyz) {
  CAP_CHECK_INIT();
  Context& c = ctx();
  Entry lk(c);
  if (!comm_exists()) {
    set_error("capgpu_msm_g1_sharded: no communicator (call capgpu_comm_init first)");
    return CAPGPU_ERR_NOT_INITIALISED;
  }
  if (count < 0 || !d_out_xyz) {
    set_error("capgpu_msm_g1_sharded: bad argument");
    return CAPGPU_ERR_INVALID_ARG;
  }
  if (count == 0) return CAPGPU_OK;
  // a local failure does not leave the other ranks waiting: this rank still enters the exchange, with its error code
  int rc = capgpu_msm_g1_dev(srs_handle, offset, d_scalars, scalar_stride, n_local, count, scalars_montgomery, d_out_xyz);
  rc = comm_allgather_sum((g1_jac*)d_out_xyz, (uint32_t)count, c.stream, rc);
  if (rc) return rc;
  return take_launch_error();
}

int capgpu_msm_g1_sharded(uint64_t srs_handle, size_t offset, const uint64_t* scalars, size_t n_local,
                          uint64_t out_xyz[12]) {
  CAP_CHECK_INIT();
  if ((!scalars && n_local) || !out_xyz) return CAPGPU_ERR_INVALID_ARG;
  Context& c = ctx();
  Entry lk(c);
  int rc = scratch_reserve(c.stage_a, sizeof(fe) * (n_local ? n_local : 1) + sizeof(g1_jac));
  if (rc) return rc;
  fe* d_sc = (fe*)c.stage_a.p;
  void* d_out = (char*)c.stage_a.p + sizeof(fe) * (n_local ? n_local : 1);
  if (n_local) CAP_HIP(hipMemcpyAsync(d_sc, scalars, sizeof(fe) * n_local, hipMemcpyHostToDevice, c.stream));
  rc = capgpu_msm_g1_sharded_dev(srs_handle, offset, d_sc, n_local, n_local, 1, 0, d_out);
  if (rc) return rc;
  CAP_HIP(hipMemcpyAsync(out_xyz, d_out, sizeof(g1_jac), hipMemcpyDeviceToHost, c.stream));
  CAP_HIP(hipStreamSynchronize(c.stream));
  return CAPGPU_OK;
}

int capgpu_plonk_shard_msm(int on) {
  CAP_CHECK_INIT();
  Entry lk(ctx());
  if (on && !comm_exists()) {
    set_error("capgpu_plonk_shard_msm: no communicator (call capgpu_comm_init first)");
    return CAPGPU_ERR_NOT_INITIALISED;
  }
  comm().shard_prover = on != 0;
  return CAPGPU_OK;
}

}  // extern "C"
