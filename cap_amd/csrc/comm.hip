// Multi-GPU exchange step of the point-range-sharded MSM, inside the C ABI.
//
// One process per GPU.  Rank g keeps the window table of its point range resident, receives the matching scalar
// slice, and runs the whole Pippenger locally down to ONE point per MSM.  The single exchange step is an all-gather of
// those 96-byte Jacobian points over RCCL (xGMI), enqueued on the library stream straight from device memory, followed
// by G - 1 group additions on the device.  RCCL has no elliptic-curve reduction operator, so there is no all-reduce;
// exchanging bucket arrays would move 2^c * 96 B per window for nothing.  The message is latency-bound (96 B x count
// per rank): xGMI bandwidth plays no role.
//
// A caller that is one process with many threads (the reference: rayon, src/utils/params_builder.rs:194-226) starts
// one worker process per GPU and hands each its range; this file is what those workers call.
//
// RCCL is loaded at capgpu_comm_init time (dlopen), so the library itself loads - and the single-GPU path runs - on
// machines without it.  If the process already holds an RCCL (PyTorch ships one), that copy is used.
#include <dlfcn.h>
#include <string.h>

#include "context.hpp"
#include "curve29.hpp"
#include "launch.hpp"

namespace cap {
namespace {

// the slice of rccl.h this file needs (ABI-stable NCCL 2 signatures)
typedef struct ncclComm* ncclComm_t;
typedef struct {
  char internal[128];
} ncclUniqueId;
typedef int ncclResult_t;
constexpr int kNcclUint8 = 1;

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

struct Comm {
  Rccl api;
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  g1_jac* d_gather = nullptr;  // world * count points
  size_t gather_cap = 0;       // points
  bool shard_prover = false;
};
Comm& comm() {
  static Comm c;
  return c;
}

int load_rccl(Rccl& r) {
  if (r.lib) return CAPGPU_OK;
  const char* override_path = getenv("CAPGPU_RCCL_LIBRARY");
  void* h = nullptr;
  if (override_path) h = dlopen(override_path, RTLD_NOW | RTLD_LOCAL);
  // a copy already in the process (same soname) first: two RCCLs in one process would each set up their own IPC state
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!h) {
    set_error("capgpu_comm: cannot load librccl.so.1 (%s)", dlerror());
    return CAPGPU_ERR_NO_DEVICE;
  }
  r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
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
  set_error("capgpu_comm: RCCL error %d (%s) in %s", e, c.api.GetErrorString ? c.api.GetErrorString(e) : "?", what);
  return CAPGPU_ERR_HIP;
}

// out[k] = sum over ranks r of all[r * count + k]   (one wavefront per MSM; lanes stride over the ranks)
__global__ __launch_bounds__(64) void g1_sum_ranks_kernel(const g1_jac* __restrict__ all, uint32_t world, uint32_t count,
                                                          g1_jac* __restrict__ out) {
  const uint32_t k = blockIdx.x;
  g1x acc = G1L::inf();
  for (uint32_t r = threadIdx.x; r < world; r += 64) {
    g1_jac p = all[(size_t)r * count + k];
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

bool comm_active() { return comm().comm != nullptr && comm().world > 1; }
bool comm_shard_prover() { return comm().comm != nullptr && comm().shard_prover; }
int comm_rank() { return comm().rank; }
int comm_world() { return comm().world; }

// d_points: this rank's `count` partial results (device, written by work already enqueued on `s`).  On return (stream
// order) d_points[k] holds the sum over all ranks of their k-th point, on every rank.
int comm_allgather_sum(g1_jac* d_points, uint32_t count, hipStream_t s) {
  Comm& c = comm();
  if (!c.comm) {
    set_error("capgpu_comm: no communicator (call capgpu_comm_init first)");
    return CAPGPU_ERR_NOT_INITIALISED;
  }
  if (count == 0) return CAPGPU_OK;
  const size_t need = (size_t)c.world * count;
  if (need > c.gather_cap) {
    CAP_HIP(hipStreamSynchronize(s));
    if (c.d_gather) CAP_HIP(hipFree(c.d_gather));
    c.d_gather = nullptr;
    c.gather_cap = 0;
    CAP_HIP(hipMalloc(&c.d_gather, sizeof(g1_jac) * need));
    c.gather_cap = need;
  }
  ncclResult_t e = c.api.AllGather(d_points, c.d_gather, sizeof(g1_jac) * count, kNcclUint8, c.comm, s);
  if (e) return rccl_fail(e, "ncclAllGather");
  launch("g1_sum_ranks", g1_sum_ranks_kernel, dim3(count), dim3(64), 0, s, (const g1_jac*)c.d_gather, (uint32_t)c.world,
         count, d_points);
  return CAPGPU_OK;
}

}  // namespace cap

using namespace cap;

extern "C" {

int capgpu_comm_unique_id(uint8_t id_out[128]) {
  CAP_CHECK_INIT();
  if (!id_out) return CAPGPU_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  Comm& c = comm();
  int rc = load_rccl(c.api);
  if (rc) return rc;
  ncclUniqueId id;
  ncclResult_t e = c.api.GetUniqueId(&id);
  if (e) return rccl_fail(e, "ncclGetUniqueId");
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
    std::lock_guard<std::recursive_mutex> lk(ctx().mu);
    if (c.comm) {
      set_error("capgpu_comm_init: a communicator already exists (capgpu_comm_destroy first)");
      return CAPGPU_ERR_INVALID_ARG;
    }
    int rc = load_rccl(c.api);
    if (rc) return rc;
  }
  // ncclCommInitRank is collective and blocks until every rank has arrived: the process lock is NOT held across it,
  // so a rank that never shows up cannot freeze the single-GPU entry points of this process (bench.py relies on it)
  ncclUniqueId uid;
  memcpy(uid.internal, id, 128);
  ncclComm_t nc = nullptr;
  ncclResult_t e = c.api.CommInitRank(&nc, world, uid, rank);
  if (e) return rccl_fail(e, "ncclCommInitRank");
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  if (c.comm) {  // a concurrent capgpu_comm_init won the race
    c.api.CommDestroy(nc);
    set_error("capgpu_comm_init: a communicator already exists (capgpu_comm_destroy first)");
    return CAPGPU_ERR_INVALID_ARG;
  }
  c.comm = nc;
  c.rank = rank;
  c.world = world;
  return CAPGPU_OK;
}

int capgpu_comm_destroy(void) {
  CAP_CHECK_INIT();
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  Comm& c = comm();
  if (!c.comm) return CAPGPU_OK;
  (void)hipStreamSynchronize(ctx().stream);
  c.api.CommDestroy(c.comm);
  c.comm = nullptr;
  c.rank = 0;
  c.world = 1;
  c.shard_prover = false;
  if (c.d_gather) hipFree(c.d_gather);
  c.d_gather = nullptr;
  c.gather_cap = 0;
  return CAPGPU_OK;
}

int capgpu_comm_info(int* rank_out, int* world_out) {
  Comm& c = comm();
  if (rank_out) *rank_out = c.comm ? c.rank : 0;
  if (world_out) *world_out = c.comm ? c.world : 0;
  return CAPGPU_OK;
}

int capgpu_msm_g1_sharded_dev(uint64_t srs_handle, size_t offset, const void* d_scalars, size_t scalar_stride,
                              size_t n_local, int count, int scalars_montgomery, void* d_out_xyz) {
  CAP_CHECK_INIT();
  Context& c = ctx();
  std::lock_guard<std::recursive_mutex> lk(c.mu);
  if (!comm().comm) {
    set_error("capgpu_msm_g1_sharded: no communicator (call capgpu_comm_init first)");
    return CAPGPU_ERR_NOT_INITIALISED;
  }
  int rc = capgpu_msm_g1_dev(srs_handle, offset, d_scalars, scalar_stride, n_local, count, scalars_montgomery, d_out_xyz);
  if (rc) return rc;
  rc = comm_allgather_sum((g1_jac*)d_out_xyz, (uint32_t)count, c.stream);
  if (rc) return rc;
  return take_launch_error();
}

int capgpu_msm_g1_sharded(uint64_t srs_handle, size_t offset, const uint64_t* scalars, size_t n_local,
                          uint64_t out_xyz[12]) {
  CAP_CHECK_INIT();
  if ((!scalars && n_local) || !out_xyz) return CAPGPU_ERR_INVALID_ARG;
  Context& c = ctx();
  std::lock_guard<std::recursive_mutex> lk(c.mu);
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
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  if (on && !comm().comm) {
    set_error("capgpu_plonk_shard_msm: no communicator (call capgpu_comm_init first)");
    return CAPGPU_ERR_NOT_INITIALISED;
  }
  comm().shard_prover = on != 0;
  return CAPGPU_OK;
}

}  // extern "C"
