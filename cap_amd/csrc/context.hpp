// Process-wide state of libcapgpu.so: the bound device, the library stream,
// resident tables (NTT domains, SRS window tables, proving keys) and scratch.
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <memory>
#include <mutex>
#include <string>

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
};

struct ProvingKey;  // plonk.hip

struct Context {
  bool initialised = false;
  int device = 0;
  hipStream_t own_stream = nullptr;
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

// grows (never shrinks) a scratch buffer; synchronises the stream before freeing the old one
int scratch_reserve(Scratch& s, size_t bytes);
// cached domain tables for 2^log_n
int get_domain(uint32_t log_n, const NttDomain** out);
int get_domain3(uint32_t log_m, const Ntt3Domain** out);
// builds the window table of `n` device-resident affine bases (arkworks form, (0,0) = infinity) and registers it
int register_srs(g1_affine* d_bases, size_t n, uint64_t* handle_out);
int find_srs(uint64_t h, const MsmBases** out);

#define CAP_CHECK_INIT()                                                  \
  do {                                                                    \
    if (!cap::ctx().initialised) {                                        \
      cap::set_error("capgpu: not initialised (call capgpu_init first)"); \
      return CAPGPU_ERR_NOT_INITIALISED;                                  \
    }                                                                     \
  } while (0)

#define CAP_HIP(expr)                                      \
  do {                                                     \
    hipError_t _e = (expr);                                \
    if (_e != hipSuccess) return cap::hip_fail(_e, #expr); \
  } while (0)

}  // namespace cap
