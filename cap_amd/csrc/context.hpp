// Process-wide state of libcapgpu.so: the bound device, the library stream,
// resident tables (NTT domains, SRS window tables, proving keys) and scratch.
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/capgpu.h"
#include "msm.hpp"
#include "ntt.hpp"

namespace cap {

struct Scratch {
  void* p = nullptr;
  size_t cap = 0;
};

struct SrsEntry {
  MsmBases bases;
  // Parts of a loaded parameter blob the prover never reads but a store -> load round trip must not lose
  // (jf-plonk's `trim` indexes powers_of_gamma_g; ark-serialize bytes kept verbatim, validated at load):
  std::vector<uint64_t> gamma_deg;  // BTreeMap keys of UniversalParams::powers_of_gamma_g (empty for a Vec)
  std::vector<uint8_t> gamma_pts;   // 32 B compressed G1 each
  std::vector<uint8_t> neg_h;       // UniversalParams::neg_powers_of_h entries, 72 B each (u64 key + compressed G2)
};

struct ProvingKey;  // plonk.hip

struct Context {
  bool initialised = false;
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t copy_stream = nullptr;  // H2D of host-resident witnesses (plonk.hip), created on first use
  hipStream_t stream = nullptr;  // the stream work is enqueued on (own_stream unless capgpu_set_stream)
  NttSmallTables small;
  std::map<uint32_t, NttDomain> domains;
  std::map<uint32_t, Ntt3Domain> domains3;  // N = 3 * 2^log_m (the prover's quotient domain)
  std::map<uint64_t, SrsEntry> srs;
  std::map<uint64_t, std::shared_ptr<ProvingKey>> keys;
  uint64_t next_handle = 1;
  Scratch ntt_scratch, msm_ws, stage_a, stage_b;
  std::recursive_mutex mu;
};

Context& ctx();
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);  // records message, returns CAPGPU_ERR_HIP / _OOM
int take_launch_error();                       // CAPGPU_OK, or CAPGPU_ERR_HIP naming the first kernel whose launch failed

// grows (never shrinks) a scratch buffer; synchronises the stream before freeing the old one
int scratch_reserve(Scratch& s, size_t bytes);
// cached domain tables for 2^log_n
int get_domain(uint32_t log_n, const NttDomain** out);
int get_domain3(uint32_t log_m, const Ntt3Domain** out);
// builds the window table of `n` device-resident affine bases (arkworks form, (0,0) = infinity) and registers it
int register_srs(g1_affine* d_bases, size_t n, uint64_t* handle_out);
int find_srs(uint64_t h, const MsmBases** out);
SrsEntry* find_srs_entry(uint64_t h);  // nullptr when unknown

// comm.hip: the RCCL communicator of a multi-process job (one process per GPU)
bool comm_active();        // a communicator of more than one rank exists
bool comm_shard_prover();  // the prover's commitment MSMs are sharded by point range over the ranks
int comm_rank();
int comm_world();
struct g1_jac;
// all-gather of `count` points per rank on stream s + their per-index sums, in place (same result on every rank)
int comm_allgather_sum(g1_jac* d_points, uint32_t count, hipStream_t s);

// Owning device pointer for temporaries of an entry point: freed on every return path (the OOM paths included).
template <class T>
struct DevTmp {
  T* p = nullptr;
  DevTmp() = default;
  DevTmp(const DevTmp&) = delete;
  DevTmp& operator=(const DevTmp&) = delete;
  ~DevTmp() {
    if (p) hipFree(p);
  }
  hipError_t alloc(size_t count) { return hipMalloc(&p, sizeof(T) * (count ? count : 1)); }
  operator T*() const { return p; }
};

// The HIP current device is per host thread; entry points may arrive on any thread (rayon workers in the reference),
// so each one binds the library's device before it touches HIP.
#define CAP_CHECK_INIT()                                                  \
  do {                                                                    \
    if (!cap::ctx().initialised) {                                        \
      cap::set_error("capgpu: not initialised (call capgpu_init first)"); \
      return CAPGPU_ERR_NOT_INITIALISED;                                  \
    }                                                                     \
    CAP_HIP(hipSetDevice(cap::ctx().device));                             \
  } while (0)

#define CAP_HIP(expr)                                      \
  do {                                                     \
    hipError_t _e = (expr);                                \
    if (_e != hipSuccess) return cap::hip_fail(_e, #expr); \
  } while (0)

}  // namespace cap
