// State of libcapgpu.so.
//
// One process drives N GPUs (capgpu_init(device_ids, n)): the reference is ONE process whose rayon threads each call
// prove() (src/utils/params_builder.rs:194-226), so the library - not the caller - spreads that work over the devices.
//   Runtime   process-wide: the device contexts, the table of logical handles, the round-robin cursor.
//   Context   one per bound device: stream, resident tables (NTT domains, SRS window tables, proving keys), scratch,
//             host thread pool, launch profiler, and the lock that serialises work on that device.
// Handles are LOGICAL: an SRS or a proving key is created on one context (its home) and replicated on the first use
// on another one (tables are immutable after creation, so a replica is a peer copy - or, for a second context on the
// same device, the same memory).  An SRS of >= 2^20 points is instead SHARDED by point range over the contexts at
// upload (SURVEY 8e inside one process): MSMs on it run on all devices and meet in one exchange of 96-byte partials.
// Lock order: context locks in ascending slot order, the registry lock (Runtime::mu) last and never held while a
// context lock is being acquired.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "../../include/capgpu.h"
#include "host_pool.hpp"
#include "launch.hpp"
#include "msm.hpp"
#include "ntt.hpp"
#include "trace.hpp"

namespace cap {

struct Scratch {
  void* p = nullptr;
  size_t cap = 0;
};

struct SrsEntry {
  MsmBases bases;
  int device = 0;       // HIP device the tables live on
  bool is_shard = false;
  size_t range_lo = 0;  // a shard of a sharded SRS: holds logical points [range_lo, range_lo + bases.n)
  // Parts of a loaded parameter blob the prover never reads but a store -> load round trip must not lose
  // (jf-plonk's `trim` indexes powers_of_gamma_g; ark-serialize bytes kept verbatim, validated at load):
  std::vector<uint64_t> gamma_deg;  // BTreeMap keys of UniversalParams::powers_of_gamma_g
  std::vector<uint8_t> gamma_pts;   // 32 B compressed G1 each
  std::vector<uint8_t> neg_h;       // UniversalParams::neg_powers_of_h entries, 72 B each (u64 key + compressed G2)
  // CommitKey::powers_of_gamma_g of a ProvingKey blob (a Vec<G1>, degrees 0, 1, ...): only capgpu_plonk_key_deserialize
  // fills it, only capgpu_plonk_key_serialize reads it
  std::vector<uint8_t> ck_gamma_pts;
  // the open key's gamma_g of that blob (affine, Montgomery words; valid when has_ck_gamma_g)
  g1_affine ck_gamma_g{};
  bool has_ck_gamma_g = false;
  // Lagrange-form commit keys derived from `bases` (lagrange.hpp), one per domain size a proving key under this SRS
  // uses; built on first use (find_lagrange), shared by the contexts of this device, never serialized
  std::mutex lag_mu;
  std::map<uint32_t, std::unique_ptr<MsmBases>> lagrange;  // log_n -> table of 2^log_n + 2 points
  std::set<uint32_t> lagrange_failed;  // domains whose table could not be built (no memory): not retried per proof
  SrsEntry() = default;
  SrsEntry(const SrsEntry&) = delete;
  SrsEntry& operator=(const SrsEntry&) = delete;
  ~SrsEntry() {
    for (auto& kv : lagrange) msm_free_bases(kv.second.get());
    msm_free_bases(&bases);
  }
};

// a logical SRS handle: one full table (replicated on demand), or one shard per context
struct SrsRecord {
  std::shared_ptr<SrsEntry> full;                 // null when sharded
  std::vector<std::shared_ptr<SrsEntry>> shards;  // [slot], null for contexts that hold none; empty when not sharded
  size_t total_n = 0;
  bool sharded() const { return !shards.empty(); }
};

struct ProvingKey;  // plonk.hip
struct ProveGraphCache;  // plonk.hip: instantiated hipGraphs of small-batch prover schedules

// Scalars of MSMs on a (possibly sharded) SRS, resident where their points are ("GPU g holds its bases
// resident and receives the matching scalar slice"): `count` arrays over logical points [offset, offset + n), cut by the
// SRS's point ranges at upload / scatter time.  An MSM on a ScalarSet exchanges nothing but the 96-byte partials.
struct ScalarSlice {
  int slot = -1;         // context the slice lives on
  fe* d = nullptr;       // [count][len], array k at + k * len
  size_t lo = 0, len = 0;  // logical points [lo, lo + len)
};
struct ScalarSet {
  uint64_t srs = 0;
  size_t offset = 0, n = 0;
  int count = 0;
  std::vector<ScalarSlice> slices;
  ScalarSet() = default;
  ScalarSet(const ScalarSet&) = delete;
  ScalarSet& operator=(const ScalarSet&) = delete;
  ~ScalarSet();  // capgpu.hip: frees every slice on its device
};

struct Context {
  bool initialised = false;
  int slot = 0;    // index in Runtime::ctxs
  int device = 0;  // HIP device id (two contexts may share a device: CAPGPU_CONTEXTS_PER_DEVICE)
  bool primary = true;  // the first context of its capgpu_init entry: the ones a sharded SRS is cut over
  hipStream_t own_stream = nullptr;
  hipStream_t copy_stream = nullptr;  // H2D of host-resident witnesses (plonk.hip), created on first use
  // small batches: round 1's interpolations and coset transforms beside its commitment MSMs (plonk.hip: side_stream)
  hipStream_t side_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipStream_t stream = nullptr;  // the stream work is enqueued on (own_stream unless capgpu_set_stream)
  hipEvent_t tm0 = nullptr, tm1 = nullptr;  // capgpu_timer_begin / _end
  bool tm_open = false;                      // a measurement is open (one per context at a time) ...
  std::thread::id tm_owner;                  // ... by this thread
  NttSmallTables small;
  std::map<uint32_t, NttDomain> domains;
  std::map<uint32_t, Ntt3Domain> domains3;  // N = 3 * 2^log_m (the prover's quotient domain)
  std::map<uint64_t, std::shared_ptr<SrsEntry>> srs;  // logical handle -> the table resident HERE (full or shard)
  std::map<uint64_t, std::shared_ptr<ProvingKey>> keys;
  Scratch ntt_scratch, msm_ws, stage_a, stage_b, prove_ws, gather;
  void* pin_host = nullptr;  // pinned host memory for the small results a prover round hands back (pinned_reserve)
  size_t pin_cap = 0;
  std::shared_ptr<ProveGraphCache> prove_graphs;  // created on first use; dropped before the stream at shutdown
  bool capturing = false;  // a stream capture is open on `stream`: scratch buffers must not grow (scratch_reserve refuses)
  std::unique_ptr<HostPool> pool;  // created on first use
  Profiler prof;
  LaunchError lerr;
  int depth = 0;  // nesting of entry points on this context (Entry)
  // a call on this context failed against capgpu_set_memory_limit and left buffers sized for it behind: the next outermost
  // entry releases the context's scratch first, so that a smaller call finds the room the failed one was refused
  bool trim_pending = false;
  std::recursive_mutex mu;
};

struct Runtime {
  std::mutex mu;  // registry lock: the handle tables below
  std::atomic<bool> initialised{false};
  std::vector<std::unique_ptr<Context>> ctxs;
  std::map<uint64_t, SrsRecord> srs;
  std::map<uint64_t, std::shared_ptr<ProvingKey>> keys;  // home copies
  std::map<uint64_t, std::shared_ptr<ScalarSet>> scalar_sets;
  std::atomic<uint64_t> next_handle{1};
  std::atomic<uint32_t> rr{0};  // round-robin cursor of the dealers
  // peer[a][b] for HIP devices a, b of the bound contexts: 1 = a reads / writes b's memory directly
  // (hipDeviceEnablePeerAccess done at init), 0 = no peer path (copies are staged by the runtime)
  std::map<std::pair<int, int>, int> peer;
  // bytes moved BETWEEN contexts (or from the host) by sharded MSMs since init: scalar slices / 96-byte partials
  std::atomic<uint64_t> shard_scalar_bytes{0}, shard_partial_bytes{0}, shard_calls{0};
  // tables replicated onto another context (SRS window tables, proving keys): count since init
  std::atomic<uint64_t> replications{0};
};
// CAPGPU_FORCE_REPLICATE=1 (tests): a context replicates SRS / key tables even from a context on the SAME device, so a
// one-GPU box executes clone_srs_to_current / clone_key_to_current and the proofs made from replicas
bool force_replicate();
Runtime& rt();

// The context of the calling thread: the one a dispatcher put it on (ScopedCtx), else the one it bound with
// capgpu_set_device, else context 0.  Before capgpu_init: an inert context (initialised == false).
Context& ctx();
int thread_bound_slot();  // -1: not bound (the library may deal this thread's host-buffer calls to any context)
inline size_t num_contexts() { return rt().ctxs.size(); }

// puts the calling thread on context c for a scope (and makes c's device the thread's HIP device)
struct ScopedCtx {
  Context* prev;
  int prev_device = -1;  // the thread's HIP device before the outermost scope (restored on exit)
  explicit ScopedCtx(Context& c);
  ~ScopedCtx();
  ScopedCtx(const ScopedCtx&) = delete;
  ScopedCtx& operator=(const ScopedCtx&) = delete;
};

// Lock of an entry point on its context.  The outermost entry also clears the launch-error latch, so an error left
// behind by a call that returned early is never reported by the next one (entry points call each other: the latch
// survives the inner calls of one outer call).
int& thread_entry_depth();  // context locks the calling thread holds through Entry
void context_trim_if_pending(Context& c);  // capgpu.hip; called with c.mu held and no call in progress on c
struct Entry {
  Context& c;
  explicit Entry(Context& c_) : c(c_) {
    c.mu.lock();
    thread_entry_depth()++;
    if (c.depth++ == 0) {
      c.lerr = LaunchError{};
      if (c.trim_pending) context_trim_if_pending(c);
    }
  }
  ~Entry() {
    c.depth--;
    thread_entry_depth()--;
    c.mu.unlock();
  }
  Entry(const Entry&) = delete;
  Entry& operator=(const Entry&) = delete;
};
// every context of the process, locked in slot order (sharded MSMs, shutdown); puts nothing on the thread
struct AllEntries {
  std::vector<std::unique_ptr<Entry>> held;
  AllEntries() {
    for (auto& c : rt().ctxs) held.emplace_back(new Entry(*c));
  }
};

void set_error(const char* fmt, ...);
const char* last_error();
int hip_fail(hipError_t e, const char* what);  // records message, returns CAPGPU_ERR_HIP / _OOM
int take_launch_error();                       // CAPGPU_OK, or CAPGPU_ERR_HIP naming the first kernel whose launch failed

// grows (never shrinks) a scratch buffer of the current context; drains its streams before freeing the old one.  Scratch is
// what capgpu_trim releases and capgpu_set_memory_limit caps (per device); device memory held outside a Scratch that is
// scratch by nature (plonk.hip's staging slots) is entered with scratch_account and asks scratch_room_for first.
int scratch_reserve(Scratch& s, size_t bytes);
// per thread: a buffer that has to grow grows to `bytes` x this factor (>= 1) instead of bytes + 25 % - set by callers whose
// next request is likely larger by more than that (gathered batches)
double& scratch_growth_scale();
void scratch_account(int device, size_t add, size_t sub);
bool scratch_room_for(int device, size_t bytes);
size_t plonk_trim_staging();  // plonk.hip: frees the unused staging slots of coalesced callers; returns the bytes
void plonk_reset_staging();   // ... and the pool's streams (capgpu_shutdown)
// grows (never shrinks) context c's pinned host buffer; synchronises its stream before freeing the old one
int pinned_reserve(Context& c, size_t bytes);
// cached domain tables for 2^log_n
int get_domain(uint32_t log_n, const NttDomain** out);
int get_domain3(uint32_t log_m, const Ntt3Domain** out);
// builds the window table of `n` device-resident affine bases (arkworks form, (0,0) = infinity) on the current
// context and registers it under a new logical handle
int register_srs(g1_affine* d_bases, size_t n, uint64_t* handle_out);
// the full SRS `h` resident on the current context (replicated from its home on first use); sharded handles are refused
int find_srs(uint64_t h, const MsmBases** out);
SrsEntry* find_srs_entry(uint64_t h);  // the current context's entry; nullptr when unknown there
// the Lagrange-form commit key of SRS `h` for a domain of 2^log_n points, resident on the current context's device:
// built on first use (on the context's stream, synchronously) and kept with the SRS (lagrange.hpp)
int find_lagrange(uint64_t h, uint32_t log_n, const MsmBases** out);
// a copy of the registry record (shared ownership of the tables); CAPGPU_ERR_BAD_HANDLE when unknown
int srs_record(uint64_t h, SrsRecord* out);
// proving keys: register a key created on the current context / find (replicate) one / drop everywhere
uint64_t register_key(const std::shared_ptr<ProvingKey>& K);
int lookup_key(uint64_t h, std::shared_ptr<ProvingKey>* out);
int clone_key_to_current(const ProvingKey& src, int src_device, std::shared_ptr<ProvingKey>* out);  // plonk.hip
// a context nobody is using right now (locked; the caller unlocks c->mu), else nullptr
Context* try_acquire_context();
// the context a host-buffer call of an unbound thread is dealt to: a free one if there is one, else round-robin
Context& pick_context();
// device-to-device copy between contexts (same device: plain copy), asynchronous on `s`
hipError_t copy_between(void* dst, int dst_device, const void* src, int src_device, size_t bytes, hipStream_t s);

// comm.hip: the RCCL communicator of a multi-process job (one process per GPU)
bool comm_active();        // a communicator of more than one rank exists
bool comm_shard_prover();  // the prover's commitment MSMs are sharded by point range over the ranks
// slot of the context the prover's sharded commitment MSMs run on while capgpu_plonk_shard_msm is on, else -1: every
// rank must then prove the same batch in the same order, so the dealers keep such batches whole and on that context
int comm_shard_slot();
bool comm_loopback();      // test communicator: the ranks are played one after the other on this device
void comm_loopback_rank(int r);
int comm_rank();
int comm_world();
struct g1_jac;
// all-gather of `count` points per rank on stream s + their per-index sums, in place (same result on every rank).
// local_rc: this rank's status so far; a rank that failed still takes part (with points at infinity) and EVERY rank
// returns an error afterwards, so that no rank is left waiting in the collective.
int comm_allgather_sum(g1_jac* d_points, uint32_t count, hipStream_t s, int local_rc = CAPGPU_OK);
// out[k] = sum_r all[r * stride + k], k < count, on stream s (one wavefront per k)
void g1_sum_ranks(const g1_jac* d_all, uint32_t world, uint32_t stride, uint32_t count, g1_jac* d_out, hipStream_t s);

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
// so each one binds its context's device before it touches HIP.
#define CAP_CHECK_INIT()                                                  \
  do {                                                                    \
    if (!cap::rt().initialised.load(std::memory_order_acquire)) {         \
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
