// Device-resident TurboPlonk prover, batched over P proofs that share a proving key.
//
// Replaces `jf_plonk::proof_system::PlonkKzgSnark::{preprocess, prove}` for CAP's
// circuits - call sites src/proof/transfer.rs:133 and :181-186, src/proof/mint.rs:76
// and :113, src/proof/freeze.rs:102 and :151 (algorithm: SURVEY.md §3.2 / Appendix A).
// The host keeps only what is O(1) per proof: the Keccak transcript, the challenge
// arithmetic and Jacobian -> affine of the 13 commitments.  All O(n) work - 7 iNTT(n),
// the coset transforms of the quotient step (jf-plonk: 26 of size 8n; here 8 of size 6n per
// proof, the 18 key columns being cached), 13 MSM, grand product, quotient, evaluations,
// linearisation, openings - runs on the GPU without leaving HBM between rounds.
//
// MI355X-first choices: a batch of P proofs is proved in lockstep so that every launch
// is P times larger (13 MSMs become 5 launches of 5P / P / 5P / 2P MSMs; NTTs are
// batched the same way); with 288 GB of HBM the proving key also keeps the coset
// evaluations of its 18 fixed polynomials resident (set CAPGPU_RECOMPUTE_PK_COSET=1 to
// re-transform them for every proof exactly as the reference schedule does).
// k_quotient is one long chain of products per coset point: column-wise multiplication schedule (field29.hpp), measured
// 16.75 -> 16.04 ms per step
#define CAP_FL_SCHED 1
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "coalescer.hpp"
#include "context.hpp"
#include "host_util.hpp"
#include "launch.hpp"
#include "params.hpp"
#include "plonk_kernels.hpp"

namespace cap {

using namespace pk;

struct ProvingKey {
  size_t n = 0, m = 0, ps = 0;  // domain, quotient domain, polynomial stride (n + 8)
  uint32_t log_n = 0, log_m = 0;
  size_t num_inputs = 0;
  uint64_t srs_handle = 0;
  fe* coef = nullptr;      // [18][ps]: 13 selector + 5 sigma polynomials (coefficients)
  fe* sig_eval = nullptr;  // [5][n]
  fe* pk_coset = nullptr;  // [18][m]
  fe* inv_nx1 = nullptr;   // [m]
  QuotConst qc;    // arkworks form (host arithmetic, k_perm_numden)
  QuotConst qc29;  // internal form of the lazy field (k_quotient)
  capgpu_verifying_key vk;
  std::vector<uint8_t> vk_bytes;
  bool recompute = false;
  int device = 0;  // HIP device the tables live on
  uint64_t uid = next_uid();  // never reused: identifies the key's tables in the signature of a captured graph
  static uint64_t next_uid() {
    static std::atomic<uint64_t> n{1};
    return n.fetch_add(1);
  }
  // Immutable once published: contexts on the same device share the object, other devices get a peer copy
  // (clone_key_to_current).  The batch workspace lives in the context (Context::prove_ws).
  ProvingKey() = default;
  ProvingKey(const ProvingKey&) = delete;
  ProvingKey& operator=(const ProvingKey&) = delete;
  ~ProvingKey() {
    for (void* p : {(void*)coef, (void*)sig_eval, (void*)pk_coset, (void*)inv_nx1})
      if (p) hipFree(p);
  }
};

// ---- hipGraph replay of the small-batch schedule -------------------------------------------------------------------
// One proof is ~100 kernel launches in five Fiat-Shamir rounds, most of them a few microseconds long: at batch 1 .. 16 the
// gaps between dependent launches and the host's launch calls are a visible part of the proof's latency (the reference's
// criterion bench times ONE note per iteration, benches/transfer.rs:103-105; rayon callers arrive one note at a time).
// The kernels between two host synchronisations (the transcript needs the commitments of a round before it can hand out
// the next challenge) form a SEGMENT whose launches depend on nothing but the call's signature - key, batch size, buffers:
// every per-call value (challenges, descriptors, blinders) travels through device memory written by copies that stay
// outside the segments.  The first call of a signature runs directly (it sizes the scratch buffers), the second is
// captured segment by segment (hipStreamBeginCapture ... hipGraphInstantiate), later ones replay seven hipGraphLaunch
// calls.  CAPGPU_GRAPH_MAX_BATCH (default 16; 0 = off) bounds the batch sizes that take this path.
constexpr int kGraphSegs = 8;
struct ProveGraphSig {
  uint64_t key_uid = 0, srs = 0;
  uint32_t P = 0;
  size_t num_inputs = 0;
  int form = 0;
  bool multi = false;
  const void *d_wires = nullptr, *ws = nullptr, *msm_ws = nullptr, *ntt_scratch = nullptr, *bases = nullptr;
  const void* lagrange = nullptr;  // the Lagrange-form commit key round 1 commits on (null: from coefficients)
  hipStream_t stream = nullptr;
  // rounds 1-2 with the side stream (segments 1 and 3 are then empty: their transforms were captured inside segments 0
  // and 7): a set captured one way must never be replayed the other way - round 3 would read stale coset evaluations
  bool overlap = false;
  bool operator==(const ProveGraphSig& o) const {
    return key_uid == o.key_uid && srs == o.srs && P == o.P && num_inputs == o.num_inputs && form == o.form &&
           multi == o.multi && d_wires == o.d_wires && ws == o.ws && msm_ws == o.msm_ws && ntt_scratch == o.ntt_scratch &&
           bases == o.bases && lagrange == o.lagrange && stream == o.stream && overlap == o.overlap;
  }
};
struct ProveGraphSet {
  ProveGraphSig sig;
  hipGraphExec_t exec[kGraphSegs] = {};
  uint32_t seen = 0;    // calls with this signature so far
  bool broken = false;  // a capture failed: this signature stays on direct launches
  uint64_t stamp = 0;
  ProveGraphSet() = default;
  ProveGraphSet(const ProveGraphSet&) = delete;
  ProveGraphSet& operator=(const ProveGraphSet&) = delete;
  ~ProveGraphSet() { drop(); }
  void drop() {
    for (auto& e : exec)
      if (e) {
        (void)hipGraphExecDestroy(e);
        e = nullptr;
      }
  }
};
struct ProveGraphCache {
  std::vector<std::unique_ptr<ProveGraphSet>> sets;
  uint64_t clock = 0;
};
static std::atomic<uint64_t> g_graph_captured{0}, g_graph_replayed{0};

namespace {

// hipGraph replay is used only on a HIP runtime at least as new as the one this library was built with.  A process that
// loaded ANOTHER libamdhip64.so.7 first runs the library on that copy (same SONAME: the loader keeps the first) - PyTorch's
// wheel bundles ROCm 7.0.2's - and round 6 caught three crashes INSIDE that runtime, under stream capture / graph launch
// (tests/test_gpu_graphs.py, test_gpu_input_forms.py, test_gpu_lagrange.py; one in ~12 suite runs; never on /opt/rocm's
// 7.2 in thousands of captures: tools/gpu_capture_stress.py, the fuzz campaigns).  On an older runtime small batches launch
// directly - a few per cent of latency - unless CAPGPU_GRAPH_FORCE=1.
bool graph_runtime_ok() {
  static const bool ok = [] {
    const char* f = getenv("CAPGPU_GRAPH_FORCE");
    if (f && atoi(f) != 0) return true;
    int v = 0;
    if (hipRuntimeGetVersion(&v) != hipSuccess) {
      (void)hipGetLastError();
      return false;
    }
    return v >= HIP_VERSION;
  }();
  return ok;
}
uint32_t graph_max_batch() {
  if (!graph_runtime_ok()) return 0;
  const char* e = getenv("CAPGPU_GRAPH_MAX_BATCH");
  const int x = e ? atoi(e) : 16;
  return (uint32_t)(x < 0 ? 0 : (x > 64 ? 64 : x));
}
// the graph set of this call's signature, or nullptr when the call launches directly (first sighting, failed capture)
ProveGraphSet* graph_set_for(Context& c, const ProveGraphSig& sig) {
  if (!c.prove_graphs) c.prove_graphs = std::make_shared<ProveGraphCache>();
  ProveGraphCache& gc = *c.prove_graphs;
  gc.clock++;
  for (auto& sp : gc.sets)
    if (sp->sig == sig) {
      sp->stamp = gc.clock;
      sp->seen++;
      return sp->broken ? nullptr : sp.get();
    }
  // a new signature: it replaces a stale one of the same (key, batch, form) - a scratch buffer moved - or the least
  // recently used of 8
  ProveGraphSet* slot = nullptr;
  for (auto& sp : gc.sets)
    if (sp->sig.key_uid == sig.key_uid && sp->sig.P == sig.P && sp->sig.form == sig.form && sp->sig.multi == sig.multi &&
        sp->sig.d_wires == sig.d_wires)
      slot = sp.get();
  if (!slot && gc.sets.size() >= 8) {
    slot = gc.sets[0].get();
    for (auto& sp : gc.sets)
      if (sp->stamp < slot->stamp) slot = sp.get();
  }
  if (!slot) {
    gc.sets.emplace_back(new ProveGraphSet);
    slot = gc.sets.back().get();
  }
  if (slot->exec[0] || slot->seen) (void)hipStreamSynchronize(c.stream);  // an instantiated graph may still be running
  slot->drop();
  slot->sig = sig;
  slot->seen = 1;
  slot->broken = false;
  slot->stamp = gc.clock;
  return nullptr;  // first call with this signature: direct launches (they size every scratch buffer)
}
// runs one segment: replay, or capture + instantiate + launch, or - without a graph set - the launches themselves
template <class F>
int run_segment(Context& c, ProveGraphSet* g, int id, F&& enqueue) {
  if (!g || g->broken) return enqueue();
  hipStream_t s = c.stream;
  if (g->exec[id]) {
    CAP_HIP(hipGraphLaunch(g->exec[id], s));
    g_graph_replayed++;
    return CAPGPU_OK;
  }
  if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) {
    (void)hipGetLastError();
    g->broken = true;
    return enqueue();
  }
  const LaunchError before = launch_error();  // an error latched by an earlier segment of this call must survive the attempt
  c.capturing = true;
  int rc = enqueue();
  c.capturing = false;
  hipGraph_t graph = nullptr;
  hipError_t e = hipStreamEndCapture(s, &graph);
  if (rc == CAPGPU_OK && e == hipSuccess && graph && launch_error().code == before.code) {
    hipGraphExec_t ex = nullptr;
    e = hipGraphInstantiate(&ex, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e == hipSuccess && ex) {
      g->exec[id] = ex;
      g_graph_captured++;
      CAP_HIP(hipGraphLaunch(ex, s));
      return CAPGPU_OK;
    }
  } else if (graph) {
    (void)hipGraphDestroy(graph);
  }
  // nothing of the captured attempt has run: abandon graphs for this signature and enqueue the segment directly
  (void)hipGetLastError();
  launch_error() = before;
  g->broken = true;
  return enqueue();
}

// The per-proof host work between rounds runs on the context's own pool (host_pool.hpp).
HostPool& host_pool() {
  Context& c = ctx();
  if (!c.pool) c.pool.reset(new HostPool(HostPool::default_threads((unsigned)std::max<size_t>(num_contexts(), 1))));
  return *c.pool;
}

template <class F>
void parallel_for(uint32_t count, F&& fn) {
  HostPool& pool = host_pool();
  if (pool.size() <= 1 || count <= 1) {
    for (uint32_t i = 0; i < count; i++) fn(i);
    return;
  }
  std::function<void(uint32_t)> job = [&](uint32_t i) { fn(i); };
  pool.run(count, job);
}

// ---- launch helpers -----------------------------------------------------------------------------------------
inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

void pad_copy(hipStream_t s, fe* dst, size_t dst_outer, size_t dst_inner, const fe* src, size_t src_outer,
              size_t src_inner, uint32_t inner, uint32_t count, size_t len, size_t total) {
  if (count == 0 || total == 0) return;
  launch("k_pad_copy", k_pad_copy, dim3(cdiv(total, kThreads), count), dim3(kThreads), 0, s, dst, dst_outer, dst_inner,
         src, src_outer, src_inner, inner, len, total);
}

template <int OP, int REV>
void scan_exclusive(hipStream_t s, const fe* in, fe* out, size_t len, size_t stride, uint32_t batch, fe* tot) {
  uint32_t nblocks = cdiv(len, kScanBlock);
  launch(OP == 0 ? "k_scan_local_mul" : "k_scan_local_add", k_scan_local<OP, REV>, dim3(nblocks, batch), dim3(kThreads),
         0, s, in, out, len, stride, tot, nblocks);
  if (nblocks > 1) {
    launch("k_scan_totals", k_scan_totals<OP>, dim3(batch), dim3(64), 0, s, tot, nblocks);
    launch("k_scan_apply", k_scan_apply<OP, REV>, dim3(nblocks, batch), dim3(kThreads), 0, s, out, len, stride,
           (const fe*)tot, nblocks);
  }
}

int run_ntt(hipStream_t s, uint32_t log_n, fe* data, size_t stride, uint32_t count, int dir, int coset) {
  Context& c = ctx();
  const NttDomain* dom = nullptr;
  int rc = get_domain(log_n, &dom);
  if (rc) return rc;
  rc = scratch_reserve(c.ntt_scratch, sizeof(fe) * stride * count);
  if (rc) return rc;
  rc = ntt_run(*dom, c.small, data, (fe*)c.ntt_scratch.p, stride, count, dir, coset, s);
  if (rc) return hip_fail((hipError_t)rc, "ntt_run");
  return CAPGPU_OK;
}

// out-of-place form: `count` arrays of src_len elements at src (src_stride apart; zero-extended to the transform
// size inside the kernel) -> transforms at dst (dst_stride apart).  Saves the padded copy an in-place call needs.
int run_ntt_from(hipStream_t s, uint32_t log_n, const fe* src, size_t src_stride, size_t src_len, fe* dst,
                 size_t dst_stride, uint32_t count, int dir, int coset) {
  Context& c = ctx();
  const NttDomain* dom = nullptr;
  int rc = get_domain(log_n, &dom);
  if (rc) return rc;
  rc = scratch_reserve(c.ntt_scratch, (sizeof(fe) << log_n) * count);
  if (rc) return rc;
  NttIo io{};
  io.src = src;
  io.src_outer = src_stride;
  io.src_inner = 0;
  io.src_group = 1;
  io.src_len = src_len;
  io.dst_outer = dst_stride;
  io.dst_inner = 0;
  io.dst_group = 1;
  rc = ntt_run(*dom, c.small, dst, (fe*)c.ntt_scratch.p, dst_stride, count, dir, coset, s, &io);
  if (rc) return hip_fail((hipError_t)rc, "ntt_run");
  return CAPGPU_OK;
}

// the quotient domain: N = 6n = 3 * 2^(log n + 1) points (ntt.hpp)
int quot_domains(uint32_t log_mm, const Ntt3Domain** d3, const NttDomain** dm) {
  int rc = get_domain3(log_mm, d3);
  if (rc) return rc;
  return get_domain(log_mm, dm);
}
// coset evaluations on the 6n domain of `count` polynomials read through io (one polynomial family per call)
int run_ntt3_fwd(hipStream_t s, uint32_t log_mm, fe* data, uint32_t count, const NttIo& io) {
  Context& c = ctx();
  const Ntt3Domain* d3 = nullptr;
  const NttDomain* dm = nullptr;
  int rc = quot_domains(log_mm, &d3, &dm);
  if (rc) return rc;
  if ((rc = scratch_reserve(c.ntt_scratch, (sizeof(fe) << log_mm) * 6 * count))) return rc;
  rc = ntt3_forward(*d3, *dm, c.small, data, io, count, (fe*)c.ntt_scratch.p, s);
  if (rc) return hip_fail((hipError_t)rc, "ntt3_forward");
  return CAPGPU_OK;
}
int run_ntt3_inv(hipStream_t s, uint32_t log_mm, fe* data, uint32_t count) {
  Context& c = ctx();
  const Ntt3Domain* d3 = nullptr;
  const NttDomain* dm = nullptr;
  int rc = quot_domains(log_mm, &d3, &dm);
  if (rc) return rc;
  if ((rc = scratch_reserve(c.ntt_scratch, (sizeof(fe) << log_mm) * 6 * count))) return rc;
  rc = ntt3_inverse(*d3, *dm, c.small, data, count, (fe*)c.ntt_scratch.p, s);
  if (rc) return hip_fail((hipError_t)rc, "ntt3_inverse");
  return CAPGPU_OK;
}

int run_msm(hipStream_t s, const MsmBases& B, const fe* scalars, size_t outer_stride, uint32_t inner,
            size_t inner_stride, size_t n, uint32_t batch, g1_jac* d_out) {
  Context& c = ctx();
  auto local = [&](size_t lo, size_t len) -> int {
    int rc = scratch_reserve(c.msm_ws, msm_workspace_bytes(B, len, batch));
    if (rc) return rc;
    rc = msm_run(B, lo, scalars + lo, outer_stride, inner, inner_stride, len, batch, 1, d_out, c.msm_ws.p, c.msm_ws.cap, s);
    return rc ? hip_fail((hipError_t)rc, "msm_run") : CAPGPU_OK;
  };
  if (!comm_shard_prover()) return local(0, n);
  // Mode A of BASELINE config 4 (capgpu_plonk_shard_msm): every rank proves the same batch, each commitment MSM is cut
  // by point range over the ranks and finished by one all-gather of 96-byte partials + G - 1 additions.
  // All ranks then hold the same commitments, derive the same challenges and stay in lock step.  A rank whose local
  // part fails still enters the exchange (comm_allgather_sum), so every rank leaves with an error.  Under the loopback
  // communicator this process plays the ranks one after the other.
  const size_t world = (size_t)comm_world();
  const bool loop = comm_loopback();
  for (size_t rank = loop ? 0 : (size_t)comm_rank(), last = loop ? world - 1 : rank; rank <= last; rank++) {
    if (loop) comm_loopback_rank((int)rank);
    const size_t base = n / world, rem = n % world;
    const size_t lo = rank * base + std::min(rank, rem), len = base + (rank < rem ? 1 : 0);
    int rc = comm_allgather_sum(d_out, batch, s, local(lo, len));
    if (rc) return rc;
  }
  return CAPGPU_OK;
}

struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(void* b) : base((char*)b) {}
  template <class T>
  T* take(size_t count) {
    off = (off + 255) / 256 * 256;
    T* p = base ? (T*)(base + off) : nullptr;
    off += sizeof(T) * count;
    return p;
  }
};

struct BatchWs {
  fe *wpoly, *wev, *pi, *num, *den, *pre, *sfx, *scan_tot, *inv_total, *zpoly, *coset, *pkc, *t, *pows, *pows_small, *pw, *batchpoly,
      *hbuf, *quot, *evals, *eval_partial, *d_pub, *d_blind;
  Chal* chal;
  Chal* chal29;  // the same challenges in the internal form
  uint32_t* flags;
  g1_jac* comms;
  EvalDesc* edesc;
  LinTerm* terms;
  const fe** key_ptrs;  // [2][P]: sigma evaluations / coset columns of every proof's key (mixed-key batches)
  size_t total;
};
constexpr uint32_t kEvalChunks = 16;
constexpr uint32_t kLinTerms = 29;

BatchWs carve(void* base, const ProvingKey& K, uint32_t P, size_t num_inputs, bool coeffs) {
  Carver c(base);
  BatchWs w{};
  size_t n = K.n, m = K.m, ps = K.ps;
  w.wpoly = c.take<fe>((size_t)P * NW * ps);
  // coefficient-form input: the witness VALUES round 2 reads are one forward transform of the caller's polynomials
  w.wev = coeffs ? c.take<fe>((size_t)P * NW * n) : nullptr;
  w.pi = c.take<fe>((size_t)P * n);
  w.num = c.take<fe>((size_t)P * n);
  w.den = c.take<fe>((size_t)P * n);
  w.pre = c.take<fe>((size_t)P * n);
  w.sfx = c.take<fe>((size_t)P * n);
  w.scan_tot = c.take<fe>((size_t)P * 2 * (cdiv(ps, kScanBlock) + 1));
  w.inv_total = c.take<fe>(P);
  w.zpoly = c.take<fe>((size_t)P * ps);
  w.coset = c.take<fe>((size_t)P * 7 * m);
  w.pkc = K.recompute ? c.take<fe>((size_t)18 * m) : nullptr;
  w.t = c.take<fe>((size_t)P * m);
  w.pows = c.take<fe>((size_t)P * 4 * ps);
  w.pw = c.take<fe>((size_t)P * 4 * 24);
  w.pows_small = c.take<fe>((size_t)P * 4 * (kPowLow + (ps + kPowLow - 1) / kPowLow));
  w.batchpoly = c.take<fe>((size_t)P * 2 * ps);
  w.hbuf = c.take<fe>((size_t)P * 2 * ps);
  w.quot = c.take<fe>((size_t)P * 2 * ps);
  w.evals = c.take<fe>((size_t)P * 10);
  w.eval_partial = c.take<fe>((size_t)P * 10 * kEvalChunks);
  w.d_pub = c.take<fe>((size_t)P * (num_inputs ? num_inputs : 1));
  w.d_blind = c.take<fe>((size_t)P * 13);
  w.chal = c.take<Chal>(P);
  w.chal29 = c.take<Chal>(P);
  w.flags = c.take<uint32_t>(P);
  w.comms = c.take<g1_jac>((size_t)P * 5);
  w.edesc = c.take<EvalDesc>((size_t)P * 10);
  w.terms = c.take<LinTerm>((size_t)P * kLinTerms);
  w.key_ptrs = c.take<const fe*>((size_t)P * 2);
  w.total = c.off + 256;
  return w;
}

// the 18 fixed polynomials -> coset evaluations on the 6n quotient domain
int compute_pk_coset(hipStream_t s, const ProvingKey& K, fe* dst) {
  return run_ntt3_fwd(s, K.log_m, dst, 18, NttIo{K.coef, K.ps, 0, K.n, 1, K.m, 0, 1});
}

// Round 1's wire commitments: from the wire polynomials' coefficients (jf-plonk's way: KZG10::commit under
// src/proof/transfer.rs:181-186) or from the witness VALUES on the Lagrange-form commit key of the domain (lagrange.hip) -
// the same group elements, the same proof bytes; a CAP witness is mostly zeros, booleans and small limbs, whose MSM scalars
// have one non-zero digit or none.  capgpu_plonk_set_wire_commit: 0 coefficients, 1 evaluations, -1 the default
// (evaluations; CAPGPU_WIRE_COMMIT=coeffs turns it off for the process).
std::atomic<int> g_wire_commit{-1};
bool wire_commit_from_evals() {
  static const int env_default = [] {
    const char* e = getenv("CAPGPU_WIRE_COMMIT");
    return (e && (!strcmp(e, "coeffs") || !strcmp(e, "0"))) ? 0 : 1;
  }();
  const int m = g_wire_commit.load(std::memory_order_relaxed);
  return (m < 0 ? env_default : m) != 0;
}

// The stream the chunks of host-resident wire columns are copied on: a copy on the launch stream itself would queue up
// behind the kernels of the chunk before it (calls are serialised by the process lock; created on first use).
hipStream_t h2d_stream() {
  Context& c = ctx();
  if (!c.copy_stream && hipStreamCreateWithFlags(&c.copy_stream, hipStreamNonBlocking) != hipSuccess)
    c.copy_stream = nullptr;
  return c.copy_stream;
}
// Small batches leave most of the chip idle during round 1's commitment MSMs (a chain of a dozen short launches per MSM
// launch): the wire polynomials' interpolation, blinding and coset transforms - which round 3 needs, not the commitments
// when those are taken from evaluations - run beside them on a side stream of the context (fork / join by events inside
// segment 0, so a captured graph gets two branches).  CAPGPU_R1_OVERLAP_MAX: largest batch that does so (default 3, 0 =
// off).  Measured, same box (profiles/small_launch_ab_r05.txt): batch 1 2.59 -> 2.38 ms, batch 2 3.8 -> 3.7 ms, batch 4
// even, batches of 8 and 16 1-2 % SLOWER - there the transforms no longer fit beside the MSMs, they only slow them down.
uint32_t r1_overlap_max() {
  static const uint32_t v = [] {
    const char* e = getenv("CAPGPU_R1_OVERLAP_MAX");
    const int x = e ? atoi(e) : 3;
    return (uint32_t)(x < 0 ? 0 : (x > 4096 ? 4096 : x));
  }();
  return v;
}
hipStream_t side_stream(Context& c) {
  if (!c.side_stream) {
    if (hipStreamCreateWithFlags(&c.side_stream, hipStreamNonBlocking) != hipSuccess) c.side_stream = nullptr;
    if (c.side_stream && (hipEventCreateWithFlags(&c.ev_fork, hipEventDisableTiming) != hipSuccess ||
                          hipEventCreateWithFlags(&c.ev_join, hipEventDisableTiming) != hipSuccess)) {
      (void)hipStreamDestroy(c.side_stream);
      c.side_stream = nullptr;
    }
    (void)hipGetLastError();
  }
  return c.side_stream;
}
// The parts of ONE dealt host batch (capgpu_plonk_prove_batch cuts it over two contexts of a device) copy their witnesses
// in PART ORDER, not side by side: two copies at once share the link, both parts' first chunks land late and the GPU idles
// for both; in part order the first part's first chunk lands after half that time and its kernels run while the second
// part's witnesses arrive (round-5 VERDICT item 2: pcie_inclusive 0.92 of the resident rate).  A part takes its turn
// before its first copy - for a BOUNDED time: it holds its context's lock while it waits, and a concurrent batch whose
// parts picked the contexts in the other order would otherwise deadlock with it (tests/test_gpu_multidev.py found that);
// after 100 ms it copies anyway, side by side as before round 6 - and passes it on when its last copy has landed; the
// dealer passes a part's turn on when the part returns, whatever happened inside (an error path never holds the others
// up).  Only batches of 64 proofs and more take turns: below that the copies are too short to matter.
struct H2dTurn {
  std::mutex mu;
  std::condition_variable cv;
  uint32_t next = 0;
  bool wait_for(uint32_t idx, uint32_t timeout_ms) {  // false: timed out (the caller goes ahead regardless)
    std::unique_lock<std::mutex> lk(mu);
    return cv.wait_for(lk, std::chrono::milliseconds(timeout_ms), [&] { return next >= idx; });
  }
  void pass(uint32_t idx) {  // part idx is done copying (idempotent)
    std::lock_guard<std::mutex> lk(mu);
    if (next < idx + 1) next = idx + 1;
    cv.notify_all();
  }
};
thread_local H2dTurn* tl_h2d_turn = nullptr;
thread_local uint32_t tl_h2d_index = 0;
static bool h2d_in_part_order() {
  static const bool on = [] {
    const char* e = getenv("CAPGPU_H2D_PART_ORDER");
    return !e || atoi(e) != 0;
  }();
  return on;
}

// chunks of proofs the host-resident wire columns of a batch are copied and committed in (round 1 of prove_batch)
uint32_t h2d_chunks(uint32_t P) {
  static const int forced = [] {
    const char* e = getenv("CAPGPU_PROVE_CHUNKS");  // tests: any batch in 1..16 chunks
    const int x = e ? atoi(e) : 0;
    return x >= 1 && x <= 16 ? x : 0;
  }();
  if (forced) return std::min<uint32_t>((uint32_t)forced, P);
  return P >= 64 ? 4u : (P >= 32 ? 2u : 1u);  // a chunk's commitments should still fill the chip (>= 80 MSMs)
}
// first proof of chunk ck (ck = chunks: P).  Equal chunks - except that the FIRST chunk of a batch of >= 64 proofs that
// starts on an idle device is kept short (CAPGPU_PROVE_FIRST_CHUNK proofs, default 16; 0 = equal chunks): nothing runs
// until it has landed.
uint32_t h2d_chunk_start(uint32_t P, uint32_t chunks, uint32_t ck, bool short_first) {
  static const uint32_t first = [] {
    const char* e = getenv("CAPGPU_PROVE_FIRST_CHUNK");
    const int x = e ? atoi(e) : 16;
    return (uint32_t)(x >= 0 && x <= 4096 ? x : 16);
  }();
  if (ck == 0) return 0;
  if (ck >= chunks) return P;
  if (!short_first || first == 0 || chunks < 3 || first * chunks >= P) return (uint32_t)((uint64_t)P * ck / chunks);
  return first + (uint32_t)((uint64_t)(P - first) * (ck - 1) / (chunks - 1));
}

// msgs / msg_lens (optional): one transcript init message per proof; otherwise ext_msg is shared by the batch.
// keys (optional): the proving key of every proof - keys of ONE domain size under ONE SRS (the reference proves transfer,
// mint and freeze notes side by side, src/utils/params_builder.rs:194-226; proofs of different circuits on the same
// domain share every MSM and NTT launch, only k_perm_numden / k_quotient and the descriptors of rounds 4-5 read key
// data).  K is then keys[0] - it lends the domain-level tables and the workspace - and pub_inputs holds P rows of
// `num_inputs` = the largest count among the keys, a key with fewer inputs using the first of its row.
int prove_batch(const ProvingKey& K, uint32_t P, const fe* d_wires, const uint64_t* pub_inputs, size_t num_inputs,
                const uint8_t* ext_msg, size_t ext_len, const uint64_t* blinders, capgpu_proof* proofs,
                const uint8_t* const* msgs = nullptr, const size_t* msg_lens = nullptr,
                const std::vector<const ProvingKey*>* keys = nullptr, const uint64_t* const* h_wires = nullptr,
                int form = CAPGPU_INPUT_EVALS) {
  // form: CAPGPU_INPUT_EVALS - d_wires / h_wires hold the wire assignment, 5 columns of n values per proof (round 1
  // interpolates them); CAPGPU_INPUT_COEFFS - they hold the 5 wire POLYNOMIALS, n coefficients each, as jf-relation's
  // compute_wire_polynomials returns them (src/proof/transfer.rs:181-186 holds that circuit): round 1 takes them as they
  // are and round 2's witness values come from ONE forward transform on the device.  The proofs are the same bytes.
  // h_wires (optional): the wire columns are still in host memory - h_wires[p] points to the 5 n elements of proof p -
  // and d_wires is an empty device buffer for them.
  // Round 1 then runs in chunks of proofs - copy, interpolate, blind, commit - so that the copy of a chunk (pageable
  // memory: the call blocks the host, not the device) overlaps the commitments of the one before.
  Context& c = ctx();
  hipStream_t s = c.stream;
  trace("pb_begin", c.slot, P);
  struct TraceEnd {
    int slot;
    ~TraceEnd() { trace("pb_end", slot); }
  } trace_end{c.slot};
  // host-resident witnesses are copied on the context's copy stream straight from the callers' buffers: no exit path -
  // an error return in particular - may leave such a copy in flight (the caller frees or reuses its buffer, and the
  // next call writes the same staging area)
  struct CopyDrain {
    Context& c;
    bool armed;
    ~CopyDrain() {
      if (armed && c.copy_stream) (void)hipStreamSynchronize(c.copy_stream);
    }
  } drain{c, h_wires != nullptr};
  const size_t n = K.n, m = K.m, ps = K.ps;
  auto key_of = [&](uint32_t p) -> const ProvingKey& { return keys ? *(*keys)[p] : K; };
  if (keys) {
    size_t max_ni = 0;
    for (uint32_t p = 0; p < P; p++) {
      const ProvingKey& Kp = key_of(p);
      if (Kp.n != K.n || Kp.srs_handle != K.srs_handle || Kp.recompute || K.recompute) {
        set_error("capgpu_plonk_prove_multi: the keys of one batch must share the domain size and the SRS");
        return CAPGPU_ERR_INVALID_ARG;
      }
      max_ni = std::max(max_ni, Kp.num_inputs);
    }
    if (num_inputs != max_ni) {
      set_error("capgpu_plonk_prove_multi: rows of %zu public inputs given, the keys need %zu", num_inputs, max_ni);
      return CAPGPU_ERR_INVALID_ARG;
    }
  } else if (num_inputs != K.num_inputs) {
    set_error("capgpu_plonk_prove: %zu public inputs given, key expects %zu", num_inputs, K.num_inputs);
    return CAPGPU_ERR_INVALID_ARG;
  }
  if (const int ss = comm_shard_slot(); ss >= 0 && ss != c.slot) {
    set_error("capgpu_plonk_prove: commitment MSMs are sharded over the communicator of context %d (capgpu_plonk_shard_msm); "
              "this call runs on context %d and would prove unsharded while its peers wait", ss, c.slot);
    return CAPGPU_ERR_INVALID_ARG;
  }
  const MsmBases* B = nullptr;
  int rc = find_srs(K.srs_handle, &B);
  if (rc) return rc;
  // the Lagrange-form commit key (built on the first proof of this domain size under this SRS if preprocess did not);
  // sharded commitment MSMs (mode A of config 4) cut the monomial key by point range: they keep the coefficient form
  // The Lagrange-form key is an OPTIMISATION (the same commitments come from the coefficients): when it cannot be had -
  // its table is as large as the SRS's window table and is built on first use with temporaries of its own - the proof is
  // made the coefficient way instead of failing.  Only a bad handle is an error of the call.
  const MsmBases* Lag = nullptr;
  if (wire_commit_from_evals() && !comm_shard_prover() && (rc = find_lagrange(K.srs_handle, K.log_n, &Lag))) {
    if (rc == CAPGPU_ERR_BAD_HANDLE) return rc;
    Lag = nullptr;
    (void)hipGetLastError();
    trace("pb_lagrange_fallback", c.slot, rc);
  }
  // workspace
  const bool coeffs = form == CAPGPU_INPUT_COEFFS;
  if ((rc = scratch_reserve(c.prove_ws, carve(nullptr, K, P, num_inputs, coeffs).total))) return rc;
  BatchWs w = carve(c.prove_ws.p, K, P, num_inputs, coeffs);
  const NttDomain* dom_n = nullptr;
  const Ntt3Domain* dom_q = nullptr;
  if ((rc = get_domain(K.log_n, &dom_n))) return rc;
  if ((rc = get_domain3(K.log_m, &dom_q))) return rc;
  // small batches replay their kernel segments as hipGraphs (see ProveGraphSet)
  const uint32_t chunks = h_wires ? h2d_chunks(P) : 1;
  // small batches run rounds 1-2 on two streams (r1_overlap_max); decided here, once, because it shapes the captured graphs
  hipStream_t s2 = nullptr;
  const bool overlap = chunks == 1 && P <= r1_overlap_max() && !c.prof.on && !comm_shard_prover() && s == c.own_stream &&
                       (s2 = side_stream(c)) != nullptr;
  // no exit path - an error between a fork and its join in particular - may leave the side stream running kernels on this
  // context's workspace: the next call (or a scratch growth) would pull it from under them
  struct SideDrain {
    Context& c;
    bool armed;
    ~SideDrain() {
      if (armed && c.side_stream) (void)hipStreamSynchronize(c.side_stream);
    }
  } side_drain{c, overlap};
  ProveGraphSet* gs = nullptr;
  // (only on the library's own stream: a caller's stream - capgpu_set_stream - may carry work of its own)
  if (P <= graph_max_batch() && chunks == 1 && !c.prof.on && comm_shard_slot() < 0 && s == c.own_stream) {
    ProveGraphSig sig;
    sig.key_uid = K.uid;
    sig.srs = K.srs_handle;
    sig.P = P;
    sig.num_inputs = num_inputs;
    sig.form = form;
    sig.multi = keys != nullptr;
    sig.d_wires = d_wires;
    sig.ws = c.prove_ws.p;
    sig.msm_ws = c.msm_ws.p;
    sig.ntt_scratch = c.ntt_scratch.p;
    sig.bases = B->ext;
    sig.lagrange = Lag ? Lag->ext : nullptr;
    sig.stream = s;
    sig.overlap = overlap;
    gs = graph_set_for(c, sig);
  }
  auto seg = [&](int id, const std::function<int()>& enqueue) -> int { return run_segment(c, gs, id, enqueue); };
  static const bool inv_on_device = [] {  // the round-2 inversion as a device kernel (the pre-round-4 schedule)
    const char* e = getenv("CAPGPU_PERM_INV_ON_DEVICE");
    return e && atoi(e) != 0;
  }();

  // ---- transcripts (host) --------------------------------------------------------------------------------
  std::vector<SolidityTranscript> tr(P);
  parallel_for(P, [&](uint32_t p) {
    if (msgs) {
      if (msgs[p] && msg_lens[p]) tr[p].append(msgs[p], msg_lens[p]);
    } else if (ext_msg && ext_len) {
      tr[p].append(ext_msg, ext_len);
    }
    const ProvingKey& Kp = key_of(p);
    tr[p].append(Kp.vk_bytes.data(), Kp.vk_bytes.size());
    for (size_t i = 0; i < Kp.num_inputs; i++) append_fr(tr[p], fe_from_words(pub_inputs + 4 * (p * num_inputs + i)));
  });
  std::vector<uint64_t> pub_rows;  // mixed keys: the unused tail of a shorter key's row must be zero on the device
  if (keys && num_inputs) {
    pub_rows.assign(pub_inputs, pub_inputs + (size_t)4 * P * num_inputs);
    for (uint32_t p = 0; p < P; p++)
      for (size_t i = key_of(p).num_inputs; i < num_inputs; i++)
        for (int k = 0; k < 4; k++) pub_rows[4 * (p * num_inputs + i) + k] = 0;
    pub_inputs = pub_rows.data();
  }
  if (num_inputs)
    CAP_HIP(hipMemcpyAsync(w.d_pub, pub_inputs, sizeof(fe) * P * num_inputs, hipMemcpyHostToDevice, s));
  std::vector<const fe*> key_ptrs;
  const fe* const* sig_of = nullptr;
  const fe* const* pkc_of = nullptr;
  if (keys) {
    key_ptrs.resize((size_t)2 * P);
    for (uint32_t p = 0; p < P; p++) {
      key_ptrs[p] = key_of(p).sig_eval;
      key_ptrs[P + p] = key_of(p).pk_coset;
    }
    CAP_HIP(hipMemcpyAsync(w.key_ptrs, key_ptrs.data(), sizeof(const fe*) * key_ptrs.size(), hipMemcpyHostToDevice, s));
    sig_of = w.key_ptrs;
    pkc_of = w.key_ptrs + P;
  }
  CAP_HIP(hipMemcpyAsync(w.d_blind, blinders, sizeof(fe) * P * 13, hipMemcpyHostToDevice, s));
  CAP_HIP(hipMemsetAsync(w.flags, 0, sizeof(uint32_t) * P, s));

  // Results the host needs between the rounds come back into PINNED memory of the context: a device-to-host copy into
  // pageable memory makes the runtime wait for the stream on the host first and copy through a staging buffer of its own
  // - two host round trips where one is needed, seven times per proof.
  g1_jac* hj = nullptr;      // [5 P] commitments of a round
  fe* h_tot = nullptr;       // [P] grand-product totals out, their inverses back
  fe* h_evals = nullptr;     // [10 P]
  uint32_t* h_flags = nullptr;  // [P]
  {
    const size_t need = sizeof(g1_jac) * P * NW + sizeof(fe) * P * 11 + sizeof(uint32_t) * P + 1024;
    if ((rc = pinned_reserve(c, need))) return rc;
    char* b = (char*)c.pin_host;
    hj = (g1_jac*)b;
    b += (sizeof(g1_jac) * P * NW + 255) / 256 * 256;
    h_tot = (fe*)b;
    b += (sizeof(fe) * P + 255) / 256 * 256;
    h_evals = (fe*)b;
    b += (sizeof(fe) * P * 10 + 255) / 256 * 256;
    h_flags = (uint32_t*)b;
  }
  std::vector<g1_affine> ha;
  // `keep_busy` enqueues work that does not depend on the next challenge: it runs on the GPU while the host turns the
  // commitments into challenges
  auto fetch_comms = [&](uint32_t count, const std::function<int()>& keep_busy = nullptr) -> int {
    CAP_HIP(hipMemcpyAsync(hj, w.comms, sizeof(g1_jac) * count, hipMemcpyDeviceToHost, s));
    CAP_HIP(hipStreamSynchronize(s));
    if (keep_busy) {
      int brc = keep_busy();
      if (brc) return brc;
    }
    // Jacobian -> affine on the host while the GPU waits for the next challenge: chunks of 64 points (one shared
    // inversion each) spread over the pool instead of one serial pass over up to 5P points
    ha.resize(count);
    const uint32_t chunk = 64, nchunks = (count + chunk - 1) / chunk;
    parallel_for(nchunks, [&](uint32_t ci) {
      const uint32_t lo = ci * chunk, hi = std::min(count, lo + chunk);
      std::vector<g1_jac> in(hj + lo, hj + hi);
      std::vector<g1_affine> out;
      batch_to_affine(in, out);
      std::copy(out.begin(), out.end(), ha.begin() + lo);
    });
    return CAPGPU_OK;
  };

  // ---- round 1: wire polynomials, public-input polynomial, 5 commitments ------------------------------
  // the interpolations read the witness columns / public inputs where they are and write the coefficient arrays (no
  // padded copies); k_blind sets the 8-element tail of every wire polynomial (two blinders, six zeros)
  // kernels of one chunk of proofs [p0, p0 + cnt): interpolation (or, from coefficient-form input, the copy into place
  // and the forward transform round 2 reads), blinding, and - when the batch is chunked - the chunk's commitments
  // the five commitments of proofs [p0, p0 + cnt): MSMs of the blinded polynomials' n + 2 coefficients on the monomial key,
  // or - same group elements - of each column's n VALUES followed by its two blinders on the Lagrange-form key.  The
  // scalars of the second form are staged in the quotient's array, which round 3 writes long after these MSMs.
  auto commit_wires = [&](uint32_t p0, uint32_t cnt) -> int {
    g1_jac* out = w.comms + (size_t)p0 * NW;
    if (!Lag) return run_msm(s, *B, w.wpoly + (size_t)p0 * NW * ps, ps, 1, 0, n + 2, cnt * NW, out);
    fe* stage = w.t + (size_t)p0 * NW * (n + 2);
    const fe* ev = (coeffs ? (const fe*)w.wev : d_wires) + (size_t)p0 * NW * n;
    launch("k_stage_evals", k_stage_evals, dim3(cdiv(n + 2, kThreads), cnt * NW), dim3(kThreads), 0, s, ev, n, n,
           (const fe*)(w.d_blind + (size_t)p0 * 13), (uint32_t)NW, 0u, 2u, stage);
    return run_msm(s, *Lag, stage, n + 2, 1, 0, n + 2, cnt * NW, out);
  };
  auto r1_chunk_kernels = [&](uint32_t p0, uint32_t cnt, bool commit) -> int {
    const size_t wo = (size_t)p0 * NW * n;
    fe* wp = w.wpoly + (size_t)p0 * NW * ps;
    int r;
    if (coeffs) {
      pad_copy(s, wp, ps, 0, d_wires + wo, n, 0, 1, cnt * NW, n, n);
      if ((r = run_ntt_from(s, K.log_n, d_wires + wo, n, n, w.wev + wo, n, cnt * NW, 0, 0))) return r;
    } else if ((r = run_ntt_from(s, K.log_n, d_wires + wo, n, n, wp, ps, cnt * NW, 1, 0))) {
      return r;
    }
    launch("k_blind", k_blind<1>, dim3(cnt * NW), dim3(64), 0, s, wp, ps, n, (const fe*)(w.d_blind + (size_t)p0 * 13),
           (uint32_t)NW, 0u, 2u, cnt * NW);
    if (commit && (r = commit_wires(p0, cnt))) return r;
    return CAPGPU_OK;
  };
  auto r1_tail_kernels = [&]() -> int {
    int r;
    if (num_inputs) {
      if ((r = run_ntt_from(s, K.log_n, w.d_pub, num_inputs, num_inputs, w.pi, n, P, 1, 0))) return r;
    } else {
      CAP_HIP(hipMemsetAsync(w.pi, 0, sizeof(fe) * (size_t)P * n, s));
    }
    if (chunks == 1 && (r = commit_wires(0, P))) return r;
    return CAPGPU_OK;
  };
  H2dTurn* const turn = h_wires ? tl_h2d_turn : nullptr;
  const uint32_t turn_idx = tl_h2d_index;
  if (turn) {
    const bool in_time = turn->wait_for(turn_idx, 100);
    trace("pb_h2d_turn", c.slot, in_time ? (int64_t)turn_idx : -1);
  }
  const bool short_first = h_wires && (!turn || turn_idx == 0);  // (a later part's copies run under the first part's kernels)
  for (uint32_t ck = 0; ck < chunks; ck++) {
    const uint32_t p0 = h2d_chunk_start(P, chunks, ck, short_first), p1 = h2d_chunk_start(P, chunks, ck + 1, short_first);
    if (h_wires) {
      hipStream_t cs = chunks > 1 ? h2d_stream() : nullptr;
      if (!cs) cs = s;
      if (cs != s && ck == 0) {  // the staging area may still be read by kernels of a call that returned early
        hipEvent_t ev;
        CAP_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        hipError_t e1 = hipEventRecord(ev, s), e2 = e1 == hipSuccess ? hipStreamWaitEvent(cs, ev, 0) : e1;
        (void)hipEventDestroy(ev);
        CAP_HIP(e2);
      }
      // one copy per run of proofs that are contiguous in host memory (a plain batch: one per chunk)
      trace("pb_h2d_issue", c.slot, ck);
      for (uint32_t p = p0; p < p1;) {
        uint32_t q = p + 1;
        while (q < p1 && h_wires[q] == h_wires[q - 1] + (size_t)4 * NW * n) q++;
        CAP_HIP(hipMemcpyAsync(const_cast<fe*>(d_wires) + (size_t)p * NW * n, h_wires[p],
                               sizeof(fe) * (size_t)(q - p) * NW * n, hipMemcpyHostToDevice, cs));
        p = q;
      }
      trace("pb_h2d_issued", c.slot, ck);
      if (cs != s) {  // the chunk's kernels wait for its copy, not for the copies after it
        hipEvent_t ev;
        CAP_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        hipError_t e1 = hipEventRecord(ev, cs), e2 = e1 == hipSuccess ? hipStreamWaitEvent(s, ev, 0) : e1;
        (void)hipEventDestroy(ev);  // released once the recorded work is done
        CAP_HIP(e2);
      }
    }
    if (chunks > 1 && (rc = r1_chunk_kernels(p0, p1 - p0, true))) return rc;
  }
  if (turn) {
    // (a pageable source has landed when hipMemcpyAsync returns; a pinned one when its stream has drained: the next part
    // gets the link to itself either way.  This part's kernels are already enqueued behind their chunks.)
    if (chunks > 1 && c.copy_stream) CAP_HIP(hipStreamSynchronize(c.copy_stream));
    turn->pass(turn_idx);
  }
  // round 3's coset evaluations of the wire and public-input polynomials (on the 6n quotient domain, straight from
  // their coefficient arrays: the transform zero-extends them) depend on nothing the transcript still has to produce
  auto r3_wire_cosets = [&](hipStream_t st) -> int {
    int r = run_ntt3_fwd(st, K.log_m, w.coset, P * NW, NttIo{w.wpoly, NW * ps, ps, n + 2, NW, 7 * m, m, NW});
    if (r) return r;
    return run_ntt3_fwd(st, K.log_m, w.coset + 6 * m, P, NttIo{w.pi, n, 0, n, 1, 7 * m, 0, 1});
  };
  auto r1_overlapped = [&]() -> int {
    int r;
    // every transform below shares c.ntt_scratch: those of the side stream run in its order, the one in front of the
    // fork before them.  The largest size is reserved now - a growth later would free a buffer still in use.
    if ((r = scratch_reserve(c.ntt_scratch, (sizeof(fe) << K.log_m) * 6 * (size_t)P * NW))) return r;
    auto interpolate_and_blind = [&](hipStream_t st) -> int {
      int q = CAPGPU_OK;
      if (coeffs) pad_copy(st, w.wpoly, ps, 0, d_wires, n, 0, 1, P * NW, n, n);
      else q = run_ntt_from(st, K.log_n, d_wires, n, n, w.wpoly, ps, P * NW, 1, 0);
      if (q) return q;
      launch("k_blind", k_blind<1>, dim3(P * NW), dim3(64), 0, st, w.wpoly, ps, n, (const fe*)w.d_blind, (uint32_t)NW, 0u,
             2u, P * NW);
      return CAPGPU_OK;
    };
    // in front of the fork: what the commitments read
    if (coeffs && (r = run_ntt_from(s, K.log_n, d_wires, n, n, w.wev, n, P * NW, 0, 0))) return r;
    if (!Lag && (r = interpolate_and_blind(s))) return r;
    CAP_HIP(hipEventRecord(c.ev_fork, s));
    CAP_HIP(hipStreamWaitEvent(s2, c.ev_fork, 0));
    // side stream: the polynomials and their coset evaluations
    if (Lag && (r = interpolate_and_blind(s2))) return r;
    if (num_inputs) {
      if ((r = run_ntt_from(s2, K.log_n, w.d_pub, num_inputs, num_inputs, w.pi, n, P, 1, 0))) return r;
    } else {
      CAP_HIP(hipMemsetAsync(w.pi, 0, sizeof(fe) * (size_t)P * n, s2));
    }
    if ((r = r3_wire_cosets(s2))) return r;
    // main stream: the five commitments
    if ((r = commit_wires(0, P))) return r;
    CAP_HIP(hipEventRecord(c.ev_join, s2));
    CAP_HIP(hipStreamWaitEvent(s, c.ev_join, 0));
    return CAPGPU_OK;
  };
  if ((rc = seg(0, [&]() -> int {
         if (overlap) return r1_overlapped();
         int r = chunks == 1 ? r1_chunk_kernels(0, P, false) : CAPGPU_OK;
         return r ? r : r1_tail_kernels();
       })))
    return rc;
  // (enqueued behind the commitments while the host hashes - unless the side stream already ran them beside the MSMs)
  if ((rc = fetch_comms(P * NW, [&]() -> int {
         if (overlap) return CAPGPU_OK;
         return seg(1, [&]() -> int { return r3_wire_cosets(s); });
       })))
    return rc;
  trace("pb_r1_done", c.slot);
  std::vector<Chal> chal(P);
  parallel_for(P, [&](uint32_t p) {
    for (int i = 0; i < NW; i++) {
      append_g1(tr[p], ha[p * NW + i]);
      affine_to_words(ha[p * NW + i], proofs[p].wires_poly_comms[i]);
    }
    (void)get_challenge(tr[p]);  // plookup's tau: drawn by jf-plonk even when the circuit has no lookups
    chal[p].beta = get_challenge(tr[p]);
    chal[p].gamma = get_challenge(tr[p]);
    chal[p].alpha = Fr::zero();
    chal[p].alpha2 = Fr::zero();
  });
  CAP_HIP(hipMemcpyAsync(w.chal, chal.data(), sizeof(Chal) * P, hipMemcpyHostToDevice, s));

  // the commitment to z and - independent of it - z's coset evaluations for round 3: one after the other (the second while
  // the host hashes), or, for the small batches of `overlap`, side by side on the two streams
  auto z_cosets = [&](hipStream_t st) -> int {
    return run_ntt3_fwd(st, K.log_m, w.coset + 5 * m, P, NttIo{w.zpoly, ps, 0, n + 3, 1, 7 * m, 0, 1});
  };
  // z's values -> its commitment.  From coefficients (jf-plonk's way): interpolate, blind, MSM on the monomial key.  From
  // evaluations (the Lagrange-form key): the MSM of the n values and the three blinders needs neither - it starts at
  // once, and the small batches of `overlap` interpolate, blind and transform z to the cosets beside it.
  auto finish_and_commit_z = [&]() -> int {
    int r;
    launch("k_perm_finish", k_perm_finish, dim3(cdiv(ps, kThreads), P), dim3(kThreads), 0, s, (const fe*)w.pre,
           (const fe*)w.sfx, (const fe*)w.den, (const fe*)w.inv_total, n, w.zpoly, ps);
    auto interpolate_and_blind = [&](hipStream_t st) -> int {
      int q = run_ntt(st, K.log_n, w.zpoly, ps, P, 1, 0);
      if (q) return q;
      launch("k_blind", k_blind<0>, dim3(P), dim3(64), 0, st, w.zpoly, ps, n, (const fe*)w.d_blind, 1u, 10u, 3u, P);
      return CAPGPU_OK;
    };
    if (Lag) {  // (the values are staged before the in-place interpolation overwrites them)
      launch("k_stage_evals", k_stage_evals, dim3(cdiv(n + 3, kThreads), P), dim3(kThreads), 0, s, (const fe*)w.zpoly, ps, n,
             (const fe*)w.d_blind, 1u, 10u, 3u, w.t);
    }
    if (!overlap) {
      if ((r = interpolate_and_blind(s))) return r;
      return Lag ? run_msm(s, *Lag, w.t, n + 3, 1, 0, n + 3, P, w.comms) : run_msm(s, *B, w.zpoly, ps, 1, 0, n + 3, P, w.comms);
    }
    if (!Lag && (r = interpolate_and_blind(s))) return r;  // the monomial MSM reads the blinded coefficients
    CAP_HIP(hipEventRecord(c.ev_fork, s));
    CAP_HIP(hipStreamWaitEvent(s2, c.ev_fork, 0));
    if (Lag && (r = interpolate_and_blind(s2))) return r;
    if ((r = z_cosets(s2))) return r;
    r = Lag ? run_msm(s, *Lag, w.t, n + 3, 1, 0, n + 3, P, w.comms) : run_msm(s, *B, w.zpoly, ps, 1, 0, n + 3, P, w.comms);
    CAP_HIP(hipEventRecord(c.ev_join, s2));
    CAP_HIP(hipStreamWaitEvent(s, c.ev_join, 0));
    return r;
  };
  // ---- round 2: permutation grand product --------------------------------------------------------------
  if ((rc = seg(2, [&]() -> int {
         launch("k_perm_numden", k_perm_numden, dim3(cdiv(n, kThreads), P), dim3(kThreads), 0, s,
                coeffs ? (const fe*)w.wev : d_wires, (const fe*)K.sig_eval, sig_of, (const fe*)dom_n->tw_fwd,
                (const Chal*)w.chal, K.qc29, n, w.num, w.den);
         {
           uint32_t nb = cdiv(n, kScanBlock);
           scan_exclusive<0, 0>(s, w.num, w.pre, n, n, P, w.scan_tot);
           scan_exclusive<0, 1>(s, w.den, w.sfx, n, n, P, w.scan_tot + (size_t)P * nb);
         }
         if (inv_on_device) {
           launch("k_perm_inv_total", k_perm_inv_total, dim3(cdiv(P, 64)), dim3(64), 0, s, (const fe*)w.sfx,
                  (const fe*)w.den, n, w.inv_total, P);
         } else {
           launch("k_perm_total", k_perm_total, dim3(cdiv(P, 64)), dim3(64), 0, s, (const fe*)w.sfx, (const fe*)w.den, n,
                  w.inv_total, P);
           return CAPGPU_OK;  // the segment ends here: the host inverts the totals (below)
         }
         return finish_and_commit_z();
       })))
    return rc;
  if (!inv_on_device) {
    // 1 / prod(den) per proof on the host: P products come back (32 B each), one shared inversion (Montgomery's trick),
    // P inverses go out - a round trip of tens of microseconds against 0.17 ms of a single device thread
    std::vector<fe> pref(P);
    fe* tot = h_tot;
    CAP_HIP(hipMemcpyAsync(tot, w.inv_total, sizeof(fe) * P, hipMemcpyDeviceToHost, s));
    CAP_HIP(hipStreamSynchronize(s));
    // (a proof whose product is zero - one of its denominators vanished, probability ~ 2^-236 - must not poison the
    // shared inversion: it is left out of the chain and gets the inverse 0, as the per-proof device inversion gave it)
    fe acc = Fr::one();
    for (uint32_t p = 0; p < P; p++) {
      pref[p] = acc;
      if (!Fr::is_zero(tot[p])) acc = Fr::mul(acc, tot[p]);
    }
    // (the product is a function of the secret witness: a fixed-length Fermat chain - 254 squarings whatever the value,
    // ~20 us once per batch - instead of the variable-time Euclidean Fr::inv the public scalars of the other rounds use)
    acc = Fr::inv_fermat(acc);
    for (uint32_t p = P; p-- > 0;) {
      if (Fr::is_zero(tot[p])) continue;
      const fe inv_p = Fr::mul(acc, pref[p]);
      acc = Fr::mul(acc, tot[p]);
      tot[p] = inv_p;
    }
    CAP_HIP(hipMemcpyAsync(w.inv_total, tot, sizeof(fe) * P, hipMemcpyHostToDevice, s));
    if ((rc = seg(7, [&]() -> int {
           return finish_and_commit_z();
         })))
      return rc;
  }
  if ((rc = fetch_comms(P, [&]() -> int {  // likewise the coset evaluations of z
         if (overlap) return CAPGPU_OK;
         return seg(3, [&]() -> int { return z_cosets(s); });
       })))
    return rc;
  trace("pb_r2_done", c.slot);
  parallel_for(P, [&](uint32_t p) {
    append_g1(tr[p], ha[p]);
    affine_to_words(ha[p], proofs[p].prod_perm_poly_comm);
    chal[p].alpha = get_challenge(tr[p]);
    chal[p].alpha2 = Fr::sqr(chal[p].alpha);
  });
  CAP_HIP(hipMemcpyAsync(w.chal, chal.data(), sizeof(Chal) * P, hipMemcpyHostToDevice, s));
  std::vector<Chal> chal29(P);
  parallel_for(P, [&](uint32_t p) {
    auto conv = [](const fe& a) { return Fr29::pack(Fr29::canonical(Fr29::from_ext(a))); };
    chal29[p].beta = conv(chal[p].beta);
    chal29[p].gamma = conv(chal[p].gamma);
    chal29[p].alpha = conv(chal[p].alpha);
    chal29[p].alpha2 = conv(chal[p].alpha2);
  });
  CAP_HIP(hipMemcpyAsync(w.chal29, chal29.data(), sizeof(Chal) * P, hipMemcpyHostToDevice, s));

  // ---- round 3: quotient polynomial (its seven coset transforms were enqueued behind the round 1 and 2 MSMs) ------
  if ((rc = seg(4, [&]() -> int {
         int r;
         const fe* pkc = K.pk_coset;
         if (K.recompute) {
           // reference schedule: the 18 selector / sigma polynomials are re-transformed for every proof
           for (uint32_t p = 0; p < P; p++)
             if ((r = compute_pk_coset(s, K, w.pkc))) return r;
           pkc = w.pkc;
         }
         launch("k_quotient", k_quotient, dim3(P, cdiv(m, kThreads)), dim3(kThreads), 0, s, pkc, pkc_of,
                (const fe*)w.coset, (const fe*)dom_q->xs29, (const fe*)K.inv_nx1, (const Chal*)w.chal29, K.qc29, m, w.t);
         if ((r = run_ntt3_inv(s, K.log_m, w.t, P))) return r;
         {
           size_t lo = NW * (n + 1) + 3;  // first index that must be zero: degree is exactly 5(n+1)+2
           launch("k_check_degree", k_check_degree, dim3(cdiv(m - (lo - 1), kThreads), P), dim3(kThreads), 0, s,
                  (const fe*)w.t, m, lo, w.flags);
         }
         return run_msm(s, *B, w.t, m, NW, n + 2, n + 2, P * NW, w.comms);
       })))
    return rc;
  uint32_t* flags = h_flags;
  CAP_HIP(hipMemcpyAsync(flags, w.flags, sizeof(uint32_t) * P, hipMemcpyDeviceToHost, s));
  if ((rc = fetch_comms(P * NW))) return rc;
  trace("pb_r3_done", c.slot);
  for (uint32_t p = 0; p < P; p++) {
    if (flags[p]) {
      set_error("capgpu_plonk_prove: proof %u: quotient polynomial has the wrong degree (flags %u): "
                "the circuit is not satisfied by this witness",
                p, flags[p]);
      return CAPGPU_ERR_PROOF;
    }
  }
  std::vector<fe> zeta(P), zeta_w(P);
  std::vector<fe> pw((size_t)P * 4 * 24);
  const fe omega = ntt_root_of_unity(K.log_n);
  parallel_for(P, [&](uint32_t p) {
    for (int i = 0; i < NW; i++) {
      append_g1(tr[p], ha[p * NW + i]);
      affine_to_words(ha[p * NW + i], proofs[p].split_quot_poly_comms[i]);
    }
    zeta[p] = get_challenge(tr[p]);
    zeta_w[p] = Fr::mul(zeta[p], omega);
    fe zi = Fr::inv(Fr::mul(zeta[p], zeta_w[p]));  // one inversion for both: 1/z = zw * zi, 1/zw = z * zi
    fe bases4[4] = {zeta[p], zeta_w[p], Fr::mul(zeta_w[p], zi), Fr::mul(zeta[p], zi)};
    for (int q = 0; q < 4; q++) {
      fe x = bases4[q];
      for (int b = 0; b < 24; b++) {
        pw[((size_t)p * 4 + q) * 24 + b] = x;
        x = Fr::sqr(x);
      }
    }
  });

  // ---- round 4: evaluations -------------------------------------------------------------------------------
  CAP_HIP(hipMemcpyAsync(w.pw, pw.data(), sizeof(fe) * pw.size(), hipMemcpyHostToDevice, s));
  std::vector<EvalDesc> ed((size_t)P * 10);
  for (uint32_t p = 0; p < P; p++) {
    const fe* pz = w.pows + ((size_t)p * 4 + 0) * ps;
    const fe* pzw = w.pows + ((size_t)p * 4 + 1) * ps;
    for (int i = 0; i < NW; i++) ed[p * 10 + i] = EvalDesc{w.wpoly + ((size_t)p * NW + i) * ps, pz, (uint32_t)(n + 2), 0};
    for (int i = 0; i < NW - 1; i++)
      ed[p * 10 + NW + i] = EvalDesc{key_of(p).coef + (size_t)(NS + i) * ps, pz, (uint32_t)n, 0};
    ed[p * 10 + 9] = EvalDesc{w.zpoly + (size_t)p * ps, pzw, (uint32_t)(n + 3), 0};
  }
  CAP_HIP(hipMemcpyAsync(w.edesc, ed.data(), sizeof(EvalDesc) * ed.size(), hipMemcpyHostToDevice, s));
  if ((rc = seg(5, [&]() -> int {
         const uint32_t small_len = kPowLow + cdiv(ps, kPowLow);
         launch("k_powers_small", k_powers_small, dim3(cdiv(small_len, kThreads), P * 4), dim3(kThreads), 0, s,
                w.pows_small, small_len, (const fe*)w.pw);
         launch("k_powers", k_powers, dim3(cdiv(ps, kThreads), P * 4), dim3(kThreads), 0, s, w.pows, ps, ps,
                (const fe*)w.pows_small, small_len);
         uint32_t per_chunk = cdiv(n + 3, kEvalChunks);
         launch("k_eval_partial", k_eval_partial, dim3(kEvalChunks, P * 10), dim3(kThreads), 0, s,
                (const EvalDesc*)w.edesc, w.eval_partial, kEvalChunks, per_chunk);
         launch("k_eval_final", k_eval_final, dim3(P * 10), dim3(64), 0, s, (const fe*)w.eval_partial, kEvalChunks,
                w.evals);
         return CAPGPU_OK;
       })))
    return rc;
  fe* evals = h_evals;
  CAP_HIP(hipMemcpyAsync(evals, w.evals, sizeof(fe) * (size_t)P * 10, hipMemcpyDeviceToHost, s));
  CAP_HIP(hipStreamSynchronize(s));
  trace("pb_r4_done", c.slot);

  // ---- round 5: linearisation + opening proofs ---------------------------------------------------------
  std::vector<LinTerm> terms((size_t)P * kLinTerms);
  const fe n_mont = fr_from_u64((uint64_t)n);
  std::atomic<int> lin_err{0};
  parallel_for(P, [&](uint32_t p) {
    const fe* ev = &evals[(size_t)p * 10];
    const fe *we = ev, *se = ev + NW;
    const fe znext = ev[9];
    for (int i = 0; i < 10; i++) append_fr(tr[p], ev[i]);
    for (int i = 0; i < NW; i++) fe_to_words(we[i], proofs[p].wires_evals[i]);
    for (int i = 0; i < NW - 1; i++) fe_to_words(se[i], proofs[p].wire_sigma_evals[i]);
    fe_to_words(znext, proofs[p].perm_next_eval);
    const fe v = get_challenge(tr[p]);
    const Chal& ch = chal[p];
    // scalars
    uint32_t e_n[8] = {(uint32_t)n, (uint32_t)((uint64_t)n >> 32), 0, 0, 0, 0, 0, 0};
    fe zeta_n = Fr::pow(zeta[p], e_n);
    fe zh = Fr::sub(zeta_n, Fr::one());
    fe l1 = Fr::mul(zh, Fr::inv(Fr::mul(n_mont, Fr::sub(zeta[p], Fr::one()))));
    LinTerm* T = &terms[(size_t)p * kLinTerms];
    int t = 0;
    auto add_term = [&](const fe* poly, const fe& sc, size_t len) {
      T[t].poly = poly;
      T[t].scalar = Fr29::pack(Fr29::canonical(Fr29::from_ext(sc)));  // internal form: k_lincomb is on the lazy field
      T[t].len = (uint32_t)len;
      t++;
    };
    const fe* const coef = key_of(p).coef;
    auto sel = [&](int i) { return coef + (size_t)i * ps; };
    for (int j = 0; j < 4; j++) add_term(sel(j), we[j], n);
    fe w01 = Fr::mul(we[0], we[1]), w23 = Fr::mul(we[2], we[3]);
    add_term(sel(4), w01, n);
    add_term(sel(5), w23, n);
    for (int j = 0; j < 4; j++) {
      fe w2 = Fr::sqr(we[j]);
      add_term(sel(6 + j), Fr::mul(Fr::sqr(w2), we[j]), n);
    }
    add_term(sel(10), Fr::neg(we[4]), n);
    add_term(sel(11), Fr::one(), n);
    add_term(sel(12), Fr::mul(Fr::mul(w01, w23), we[4]), n);
    // z(X) coefficient: alpha * prod(w_i + beta k_i zeta + gamma) + alpha^2 L1(zeta)
    fe bz = Fr::mul(ch.beta, zeta[p]);
    fe cz = ch.alpha;
    for (int j = 0; j < NW; j++)
      cz = Fr::mul(cz, Fr::add(Fr::add(we[j], ch.gamma), j == 0 ? bz : Fr::mul(K.qc.k[j], bz)));
    cz = Fr::add(cz, Fr::mul(ch.alpha2, l1));
    add_term(w.zpoly + (size_t)p * ps, cz, n + 3);
    // last sigma polynomial: - alpha beta z(zeta w) prod_{i<4}(w_i + beta sigma_i + gamma)
    fe cs = Fr::mul(Fr::mul(ch.alpha, ch.beta), znext);
    for (int j = 0; j < NW - 1; j++) cs = Fr::mul(cs, Fr::add(Fr::add(we[j], ch.gamma), Fr::mul(ch.beta, se[j])));
    add_term(coef + (size_t)(NS + NW - 1) * ps, Fr::neg(cs), n);
    // quotient part: - Z_H(zeta) * sum zeta^(i(n+2)) t_i(X)
    uint32_t e_n2[8] = {(uint32_t)(n + 2), (uint32_t)((uint64_t)(n + 2) >> 32), 0, 0, 0, 0, 0, 0};
    fe zp = Fr::pow(zeta[p], e_n2);
    fe cq = Fr::neg(zh);
    for (int j = 0; j < NW; j++) {
      add_term(w.t + (size_t)p * m + (size_t)j * (n + 2), cq, n + 2);
      cq = Fr::mul(cq, zp);
    }
    // batched opening at zeta: + v^(j+1) * {wire polys, first 4 sigma polys}
    fe cf = v;
    for (int j = 0; j < NW; j++) {
      add_term(w.wpoly + ((size_t)p * NW + j) * ps, cf, n + 2);
      cf = Fr::mul(cf, v);
    }
    for (int j = 0; j < NW - 1; j++) {
      add_term(coef + (size_t)(NS + j) * ps, cf, n);
      cf = Fr::mul(cf, v);
    }
    if (t != (int)kLinTerms) lin_err = t;
  });
  if (lin_err) {
    set_error("capgpu: internal error: %d linear terms", lin_err.load());
    return CAPGPU_ERR_PROOF;
  }
  CAP_HIP(hipMemcpyAsync(w.terms, terms.data(), sizeof(LinTerm) * terms.size(), hipMemcpyHostToDevice, s));
  // batchpoly[p][0] = linear combination, batchpoly[p][1] = z polynomial
  if ((rc = seg(6, [&]() -> int {
         launch("k_lincomb", k_lincomb, dim3(cdiv(ps, kThreads), P), dim3(kThreads), 0, s, (const LinTerm*)w.terms,
                kLinTerms, w.batchpoly, 2 * ps, ps);
         pad_copy(s, w.batchpoly + ps, 2 * ps, 0, w.zpoly, ps, 0, 1, P, n + 3, ps);
         launch("k_div_prepare", k_div_prepare, dim3(cdiv(n + 3, kThreads), P * 2), dim3(kThreads), 0, s,
                (const fe*)w.batchpoly, (const fe*)w.pows, ps, n + 3, w.hbuf);
         // the suffix sums go to batchpoly (its contents are dead once h is formed)
         scan_exclusive<1, 1>(s, w.hbuf, w.batchpoly, n + 3, ps, P * 2, w.scan_tot);
         launch("k_div_finish", k_div_finish, dim3(cdiv(ps, kThreads), P * 2), dim3(kThreads), 0, s,
                (const fe*)w.batchpoly, (const fe*)w.pows, ps, n + 3, w.quot);
         return run_msm(s, *B, w.quot, ps, 1, 0, n + 2, P * 2, w.comms);
       })))
    return rc;
  if ((rc = fetch_comms(P * 2))) return rc;
  trace("pb_r5_done", c.slot);
  for (uint32_t p = 0; p < P; p++) {
    affine_to_words(ha[p * 2], proofs[p].opening_proof);
    affine_to_words(ha[p * 2 + 1], proofs[p].shifted_opening_proof);
  }
  side_drain.armed = false;  // (every join was waited for in stream order and the stream has drained)
  return take_launch_error();
}

// ---- proving-key construction shared by preprocess and the blob loader -------------------------------------
int key_init(ProvingKey& K, size_t n, size_t num_inputs, uint64_t srs_handle) {
  K.n = n;
  K.m = 6 * n;  // quotient domain: 3 * 2^(log n + 1) points
  K.ps = n + 8;
  while (((size_t)1 << K.log_n) < n) K.log_n++;
  K.log_m = K.log_n + 1;  // log2 of the power-of-two factor of m
  if (K.log_m > 25) {
    set_error("capgpu_plonk_preprocess: domain too large");
    return CAPGPU_ERR_INVALID_ARG;
  }
  K.num_inputs = num_inputs;
  K.srs_handle = srs_handle;
  K.device = ctx().device;
  const char* env = getenv("CAPGPU_RECOMPUTE_PK_COSET");
  K.recompute = env && atoi(env) != 0;
  CAP_HIP(hipMalloc(&K.coef, sizeof(fe) * 18 * K.ps));
  CAP_HIP(hipMalloc(&K.sig_eval, sizeof(fe) * NW * n));
  CAP_HIP(hipMalloc(&K.inv_nx1, sizeof(fe) * K.m));
  return CAPGPU_OK;
}

// everything derived from the coefficient table: constants of the quotient kernel, 1 / (n (x - 1)) on the coset,
// and the cached coset evaluations of the 18 fixed polynomials
int key_finish_tables(hipStream_t s, ProvingKey& K) {
  const size_t n = K.n, m = K.m;
  int rc;
  const uint64_t five[4] = {5, 0, 0, 0};
  K.qc.g = Fr::to_mont(fe_from_words(five));
  for (int i = 0; i < NW; i++) K.qc.k[i] = Fr::to_mont(fe_from_words(K_CANON[i]));
  {
    uint32_t e_n[8] = {(uint32_t)n, (uint32_t)((uint64_t)n >> 32), 0, 0, 0, 0, 0, 0};
    fe gn = Fr::pow(K.qc.g, e_n);
    const Ntt3Domain* dq = nullptr;
    if ((rc = get_domain3(K.log_m, &dq))) return rc;
    fe w6 = Fr::pow(dq->omega, e_n);  // omega_m^n: a primitive 6th root of unity (m = 6n)
    fe x = gn;
    for (int i = 0; i < 8; i++) K.qc.zh_inv[i] = Fr::zero();
    for (int i = 0; i < 6; i++) {
      K.qc.zh_inv[i] = Fr::inv(Fr::sub(x, Fr::one()));
      x = Fr::mul(x, w6);
    }
  }
  const Ntt3Domain* dom_m = nullptr;
  if ((rc = get_domain3(K.log_m, &dom_m))) return rc;
  launch("k_inv_nx1", k_inv_nx1, dim3(cdiv(m, kThreads)), dim3(kThreads), 0, s, K.inv_nx1, (const fe*)dom_m->xs_ext,
         fr_from_u64((uint64_t)n), m);
  ntt_table_to_internal(K.inv_nx1, K.inv_nx1, m, s);
  {
    auto conv = [](const fe& a) { return Fr29::pack(Fr29::canonical(Fr29::from_ext(a))); };
    K.qc29.g = conv(K.qc.g);
    for (int i = 0; i < NW; i++) K.qc29.k[i] = conv(K.qc.k[i]);
    for (int i = 0; i < 8; i++) K.qc29.zh_inv[i] = conv(K.qc.zh_inv[i]);
  }
  if (!K.recompute) {
    CAP_HIP(hipMalloc(&K.pk_coset, sizeof(fe) * 18 * m));
    if ((rc = compute_pk_coset(s, K, K.pk_coset))) return rc;
  }
  return CAPGPU_OK;
}

// verifying key and transcript prefix from the 18 affine commitments (13 selectors, then 5 sigmas)
void key_set_vk(ProvingKey& K, const std::vector<g1_affine>& ha) {
  memset(&K.vk, 0, sizeof(K.vk));
  K.vk.domain_size = K.n;
  K.vk.num_inputs = K.num_inputs;
  for (int i = 0; i < NW; i++) fe_to_words(K.qc.k[i], K.vk.k[i]);
  for (int i = 0; i < NS; i++) affine_to_words(ha[i], K.vk.selector_comms[i]);
  for (int i = 0; i < NW; i++) affine_to_words(ha[NS + i], K.vk.sigma_comms[i]);
  // transcript prefix: field bits, domain size, #inputs, k_i, selector and sigma commitments
  SolidityTranscript t;
  t.append_u64_le(254);
  t.append_u64_le((uint64_t)K.n);
  t.append_u64_le((uint64_t)K.num_inputs);
  for (int i = 0; i < NW; i++) append_fr(t, K.qc.k[i]);
  for (int i = 0; i < 18; i++) append_g1(t, ha[i]);
  K.vk_bytes = t.buf;
}

// ---- coalescing of concurrent single-proof calls ----------------------------------------------------------------
// The reference proves notes under rayon (`into_par_iter()`, src/utils/params_builder.rs:194-226): many host threads
// each calling prove() for ONE note.  Behind one device and one process lock those calls would run one after the
// other at single-proof latency (4 ms each, the chip mostly idle).  With coalescing switched on
// (capgpu_plonk_set_coalescing) the calls that arrive for the same proving key while the device is busy - or within a
// short window - are gathered and proved as ONE device batch; every caller gets its own proof and its own error code.
struct ProveReq {
  uint64_t pk;
  const uint64_t* wires;
  const uint64_t* pubs;
  size_t num_inputs;
  const uint8_t* msg;
  size_t msg_len;
  const uint64_t* blinders;
  capgpu_proof* out;
  int form = CAPGPU_INPUT_EVALS;  // the requests of one gathered batch share it (it is part of the group id)
  int rc = CAPGPU_OK;
  std::string err;
  bool done = false;
  // the caller's own copy of its witness on the device, started when the call arrived (StagePool); null: not staged
  const void* d_wires = nullptr;
  hipEvent_t staged = nullptr;  // recorded behind that copy
};

// OPTIONAL (CAPGPU_COALESCE_PRESTAGE=1; off by default): witnesses of coalesced single-proof calls copied to the device BY
// THEIR CALLERS, when the call arrives - into a slot of this pool, on the pool's streams - instead of by the leader once
// the batch has been gathered.  The copy of a 24 .. 40 proof batch (3 - 4.5 ms at 5.2 MB per proof) sits at the head of
// every batch with nothing of that batch running, a tenth of its lifetime; staged, the batch starts from device memory
// (one device-to-device copy per request into its contiguous input array) and a caller's latency drops by those
// milliseconds (50 -> 45 ms with 64 closed-loop callers).  THROUGHPUT does not move (profiles/phase_trace_r06.md: 0.875 of
// the resident rate either way): the link needs the same 4 ms for the batch's 200 MB wherever the copy is issued, the
// callers now come back spread over those milliseconds, and the batches that form are smaller.  One physical device only
// (the slot must live where the batch will run); slots are scratch for capgpu_trim / capgpu_set_memory_limit.
struct StageSlot {
  void* d = nullptr;
  hipEvent_t ev = nullptr;
  size_t bytes = 0;
  hipStream_t stream = nullptr;  // one of the pool's copy streams (callers copy concurrently: several, round-robin)
};
struct StagePool {
  std::mutex mu;
  static constexpr int kStreams = 4;
  hipStream_t streams[kStreams] = {};
  int device = -1;
  std::vector<StageSlot> idle;
  size_t live = 0, made = 0;
  static constexpr size_t kMaxSlots = 320;
  static bool enabled() {  // (read per call: a process may switch it)
    const char* e = getenv("CAPGPU_COALESCE_PRESTAGE");
    return e && atoi(e) != 0;
  }
  // a slot of `bytes` on `dev`, or an empty one (the caller's witness then travels with the batch, as before)
  StageSlot acquire(int dev, size_t bytes) {
    std::lock_guard<std::mutex> lk(mu);
    if (device >= 0 && device != dev) return {};
    if (!streams[0]) {
      for (int i = 0; i < kStreams; i++)
        if (hipStreamCreateWithFlags(&streams[i], hipStreamNonBlocking) != hipSuccess) {
          (void)hipGetLastError();
          for (int j = 0; j < i; j++) (void)hipStreamDestroy(streams[j]);
          for (int j = 0; j < kStreams; j++) streams[j] = nullptr;
          return {};
        }
      device = dev;
    }
    for (size_t i = 0; i < idle.size(); i++)
      if (idle[i].bytes == bytes) {
        StageSlot s = idle[i];
        idle.erase(idle.begin() + (long)i);
        return s;
      }
    if (live >= kMaxSlots || !scratch_room_for(dev, bytes)) return {};
    StageSlot s;
    if (hipMalloc(&s.d, bytes) != hipSuccess || hipEventCreateWithFlags(&s.ev, hipEventDisableTiming) != hipSuccess) {
      if (s.d) (void)hipFree(s.d);
      (void)hipGetLastError();
      return {};
    }
    s.bytes = bytes;
    s.stream = streams[made++ % kStreams];
    live++;
    scratch_account(dev, bytes, 0);
    return s;
  }
  void release(const StageSlot& s) {
    if (!s.d) return;
    std::lock_guard<std::mutex> lk(mu);
    idle.push_back(s);
  }
  size_t trim() {  // frees the idle slots (the ones in use stay with their callers)
    std::lock_guard<std::mutex> lk(mu);
    size_t freed = 0;
    for (int i = 0; i < kStreams; i++)
      if (streams[i]) (void)hipStreamSynchronize(streams[i]);
    for (StageSlot& s : idle) {
      (void)hipFree(s.d);
      (void)hipEventDestroy(s.ev);
      freed += s.bytes;
      scratch_account(device, 0, s.bytes);
      live--;
    }
    idle.clear();
    (void)hipGetLastError();
    return freed;
  }
  void reset() {  // capgpu_shutdown: nothing is in flight any more
    (void)trim();
    std::lock_guard<std::mutex> lk(mu);
    for (int i = 0; i < kStreams; i++) {
      if (streams[i]) (void)hipStreamDestroy(streams[i]);
      streams[i] = nullptr;
    }
    device = -1;
    (void)hipGetLastError();
  }
};
StagePool& stage_pool() {
  static StagePool p;
  return p;
}
// (the gathering protocol - queues, leaders, windows, the cut over two contexts - is coalescer.hpp, which also builds for
// the host alone and runs under ThreadSanitizer there: tests/cpp/coalescer_tsan.cpp)
struct Coalescer : CoalescerCore<ProveReq> {
  // (proving-key handle, input form) -> (group id, domain size of the key); handles are never reused
  std::map<uint64_t, std::pair<uint64_t, size_t>> group_of;
};
Coalescer& coalescer() {
  static Coalescer c;
  return c;
}

}  // namespace

size_t plonk_trim_staging() { return stage_pool().trim(); }
void plonk_reset_staging() { stage_pool().reset(); }

// the registry's (home) copy of a key, without replicating it
static int home_key(uint64_t h, std::shared_ptr<ProvingKey>* out) {
  Runtime& R = rt();
  std::lock_guard<std::mutex> lk(R.mu);
  auto it = R.keys.find(h);
  if (it == R.keys.end()) {
    set_error("capgpu: unknown proving key handle %llu", (unsigned long long)h);
    return CAPGPU_ERR_BAD_HANDLE;
  }
  *out = it->second;
  return CAPGPU_OK;
}

int clone_key_to_current(const ProvingKey& src, int src_device, std::shared_ptr<ProvingKey>* out) {
  Context& c = ctx();
  auto K = std::make_shared<ProvingKey>();
  K->n = src.n;
  K->m = src.m;
  K->ps = src.ps;
  K->log_n = src.log_n;
  K->log_m = src.log_m;
  K->num_inputs = src.num_inputs;
  K->srs_handle = src.srs_handle;
  K->qc = src.qc;
  K->qc29 = src.qc29;
  K->vk = src.vk;
  K->vk_bytes = src.vk_bytes;
  K->recompute = src.recompute;
  K->device = c.device;
  auto dup = [&](fe** dst, const fe* from, size_t count) -> int {
    if (!from) return CAPGPU_OK;
    CAP_HIP(hipMalloc(dst, sizeof(fe) * count));
    CAP_HIP(copy_between(*dst, c.device, from, src_device, sizeof(fe) * count, c.stream));
    return CAPGPU_OK;
  };
  int rc;
  if ((rc = dup(&K->coef, src.coef, 18 * src.ps))) return rc;
  if ((rc = dup(&K->sig_eval, src.sig_eval, (size_t)NW * src.n))) return rc;
  if ((rc = dup(&K->pk_coset, src.pk_coset, 18 * src.m))) return rc;
  if ((rc = dup(&K->inv_nx1, src.inv_nx1, src.m))) return rc;
  CAP_HIP(hipStreamSynchronize(c.stream));
  *out = K;
  return CAPGPU_OK;
}

// the key resident on the current context: replicated from its home on first use (same device: the same object)
int lookup_key(uint64_t h, std::shared_ptr<ProvingKey>* out) {
  Context& c = ctx();
  auto it = c.keys.find(h);
  if (it != c.keys.end()) {
    *out = it->second;
    return CAPGPU_OK;
  }
  std::shared_ptr<ProvingKey> home, rep;
  int rc = home_key(h, &home);
  if (rc) return rc;
  if (home->device == c.device && !force_replicate()) rep = home;
  else if ((rc = clone_key_to_current(*home, home->device, &rep))) return rc;
  else rt().replications++;
  if ((rc = home_key(h, &home))) return rc;  // freed while it was being copied
  c.keys[h] = rep;
  *out = rep;
  return CAPGPU_OK;
}

}  // namespace cap

using namespace cap;

extern "C" {

static bool bad_form(int form) {
  if (form == CAPGPU_INPUT_EVALS || form == CAPGPU_INPUT_COEFFS) return false;
  set_error("capgpu_plonk: input_form %d is neither CAPGPU_INPUT_EVALS (0) nor CAPGPU_INPUT_COEFFS (1)", form);
  return true;
}

int capgpu_plonk_preprocess_ex(uint64_t srs_handle, size_t n, size_t num_inputs, const uint64_t* selectors,
                               const uint64_t* sigma_evals, int input_form, uint64_t* pk_handle_out,
                               capgpu_verifying_key* vk_out) {
  CAP_CHECK_INIT();
  if (bad_form(input_form)) return CAPGPU_ERR_INVALID_ARG;
  Context& c = ctx();
  Entry lk(c);
  // n >= 16: the five split-quotient commitments read 5 (n + 2) coefficients of the 6n-point quotient array
  if (!selectors || !sigma_evals || !pk_handle_out || n < 16 || (n & (n - 1)) || num_inputs >= n) {
    set_error("capgpu_plonk_preprocess: bad argument (n must be a power of two >= 16, num_inputs < n)");
    return CAPGPU_ERR_INVALID_ARG;
  }
  const MsmBases* B = nullptr;
  int rc = find_srs(srs_handle, &B);
  if (rc) return rc;
  if (B->n < n + 3) {
    set_error("capgpu_plonk_preprocess: SRS holds %zu powers, the circuit needs %zu (n + 3)", B->n, n + 3);
    return CAPGPU_ERR_INVALID_ARG;
  }
  hipStream_t s = c.stream;
  auto K = std::make_shared<ProvingKey>();
  if ((rc = key_init(*K, n, num_inputs, srs_handle))) return rc;
  const size_t ps = K->ps;
  // stage the evaluation columns, then interpolate
  DevTmp<fe> stage;
  DevTmp<g1_jac> d_comms;
  CAP_HIP(stage.alloc(18 * n));
  CAP_HIP(hipMemcpyAsync(stage, selectors, sizeof(fe) * NS * n, hipMemcpyHostToDevice, s));
  CAP_HIP(hipMemcpyAsync(stage.p + (size_t)NS * n, sigma_evals, sizeof(fe) * NW * n, hipMemcpyHostToDevice, s));
  CAP_HIP(hipMemcpyAsync(K->sig_eval, stage.p + (size_t)NS * n, sizeof(fe) * NW * n, hipMemcpyDeviceToDevice, s));
  pad_copy(s, K->coef, ps, 0, stage.p, n, 0, 1, 18, n, ps);
  if (input_form == CAPGPU_INPUT_COEFFS) {
    // the 18 polynomials arrive as jf-relation computes them (compute_selector_polynomials /
    // compute_extended_permutation_polynomials): nothing to interpolate; round 2 reads sigma's VALUES on the domain
    if ((rc = run_ntt(s, K->log_n, K->sig_eval, n, NW, 0, 0))) return rc;
  } else if ((rc = run_ntt(s, K->log_n, K->coef, ps, 18, 1, 0))) {
    return rc;
  }
  if ((rc = key_finish_tables(s, *K))) return rc;
  // verifying key: commitments of the 18 polynomials
  CAP_HIP(d_comms.alloc(18));
  if ((rc = run_msm(s, *B, K->coef, ps, 1, 0, n, 18, d_comms))) return rc;
  std::vector<g1_jac> hj(18);
  std::vector<g1_affine> ha;
  CAP_HIP(hipMemcpyAsync(hj.data(), d_comms, sizeof(g1_jac) * 18, hipMemcpyDeviceToHost, s));
  CAP_HIP(hipStreamSynchronize(s));
  batch_to_affine(hj, ha);
  key_set_vk(*K, ha);
  if (vk_out) *vk_out = K->vk;
  // the Lagrange-form commit key of this domain under this SRS (round 1's wire commitments; built once per pair and kept
  // with the SRS): made here so that the first proof does not pay for it
  if (wire_commit_from_evals() && !comm_shard_prover()) {
    const MsmBases* Lag = nullptr;
    // (an optimisation: a key whose Lagrange-form table cannot be built proves from coefficients - see prove_batch)
    if ((rc = find_lagrange(srs_handle, K->log_n, &Lag)) == CAPGPU_ERR_BAD_HANDLE) return rc;
    (void)hipGetLastError();
  }
  *pk_handle_out = register_key(K);
  return take_launch_error();
}

int capgpu_plonk_preprocess(uint64_t srs_handle, size_t n, size_t num_inputs, const uint64_t* selectors,
                            const uint64_t* sigma_evals, uint64_t* pk_handle_out, capgpu_verifying_key* vk_out) {
  return capgpu_plonk_preprocess_ex(srs_handle, n, num_inputs, selectors, sigma_evals, CAPGPU_INPUT_EVALS, pk_handle_out,
                                    vk_out);
}

// ---- ProvingKey blob (layout in include/capgpu.h) ---------------------------------------------
int capgpu_plonk_key_serialize(uint64_t pk_handle, const uint64_t gamma_g[8], const uint64_t h[16],
                               const uint64_t beta_h[16], uint8_t* out, size_t cap, size_t* len_out) {
  CAP_CHECK_INIT();
  Context& c = ctx();
  Entry lk(c);
  if (!h || !beta_h || !len_out) {
    set_error("capgpu_plonk_key_serialize: bad argument");
    return CAPGPU_ERR_INVALID_ARG;
  }
  std::shared_ptr<ProvingKey> K;
  int rc = lookup_key(pk_handle, &K);
  if (rc) return rc;
  const MsmBases* B = nullptr;
  if ((rc = find_srs(K->srs_handle, &B))) return rc;
  const size_t n = K->n, ps = K->ps, n_ck = n + 3;
  // upper bound: every polynomial at full length
  const SrsEntry* E0 = find_srs_entry(K->srs_handle);
  const size_t bound = 2 * 8 + 18 * (8 + 32 * n) + 8 + 32 * n_ck + 8 +
                       (E0 ? std::max(E0->ck_gamma_pts.size(), (size_t)32 * n_ck) : 0) + 1024;
  if (!out) {
    *len_out = bound;
    return CAPGPU_OK;
  }
  hipStream_t s = c.stream;
  std::vector<uint8_t> coef(32 * 18 * ps), ck(32 * n_ck);
  if ((rc = params::fr_mont_to_bytes(K->coef, 18 * ps, coef.data(), s))) return rc;
  if ((rc = params::compress_g1(B->ext, 1, n_ck, ck.data(), s))) return rc;
  params::Writer w;
  auto poly = [&](int idx) {  // DensePolynomial: no trailing zero coefficients
    const uint8_t* p = &coef[32 * (size_t)idx * ps];
    size_t len = n;
    auto zero = [&](size_t i) {
      for (int b = 0; b < 32; b++)
        if (p[32 * i + b]) return false;
      return true;
    };
    while (len && zero(len - 1)) len--;
    w.u64(len);
    w.put(p, 32 * len);
  };
  w.u64(NW);
  for (int i = 0; i < NW; i++) poly(NS + i);
  w.u64(NS);
  for (int i = 0; i < NS; i++) poly(i);
  w.u64(n_ck);
  w.put(ck.data(), ck.size());
  {
    // CommitKey::powers_of_gamma_g (Vec<G1>, degrees 0 .. n_ck - 1 - what jf-plonk's trim emits).  A key loaded from a
    // ProvingKey blob re-emits the blob's vector.  A key preprocessed under a loaded UniversalSrs takes the degrees
    // 0 .. n_ck - 1 from that SRS's BTreeMap, in order - all of them or, when one is missing, none (a sparse map must
    // not turn into a shorter vector with the degrees lost).  A synthetic SRS has no hiding powers: empty vector.
    const SrsEntry* E = find_srs_entry(K->srs_handle);
    std::vector<uint8_t> gp;
    if (E && !E->ck_gamma_pts.empty()) {
      gp = E->ck_gamma_pts;
    } else if (E && !E->gamma_deg.empty()) {
      std::map<uint64_t, size_t> at;
      for (size_t i = 0; i < E->gamma_deg.size(); i++) at[E->gamma_deg[i]] = i;
      bool all = true;
      for (uint64_t d = 0; d < n_ck && all; d++) all = at.count(d) != 0;
      if (all)
        for (uint64_t d = 0; d < n_ck; d++)
          gp.insert(gp.end(), E->gamma_pts.begin() + 32 * at[d], E->gamma_pts.begin() + 32 * at[d] + 32);
    }
    w.u64(gp.size() / 32);
    if (!gp.empty()) w.put(gp.data(), gp.size());
  }
  params::OpenKey ok;
  {
    g1_affine g0;
    if (!params::g1_decompress_host(ck.data(), &g0)) return CAPGPU_ERR_SERIALIZATION;
    ok.g = g0;
  }
  // open key's gamma_g: the caller's, else what the blob this key (or its SRS) was loaded from held - the key blob's own
  // value, or degree 0 of a UniversalSrs's hiding powers - else infinity (a synthetic SRS has none)
  ok.gamma_g.x = ok.gamma_g.y = Fq::zero();
  if (gamma_g) {
    ok.gamma_g = params::g1_from_words(gamma_g);
  } else if (const SrsEntry* Eg = find_srs_entry(K->srs_handle)) {
    if (Eg->has_ck_gamma_g) {
      ok.gamma_g = Eg->ck_gamma_g;
    } else {
      for (size_t i = 0; i < Eg->gamma_deg.size(); i++)
        if (Eg->gamma_deg[i] == 0) {
          g1_affine g0;
          if (params::g1_decompress_host(&Eg->gamma_pts[32 * i], &g0)) ok.gamma_g = g0;
          break;
        }
    }
  }
  ok.h = params::g2_from_words(h);
  ok.beta_h = params::g2_from_words(beta_h);
  params::write_vk(w, K->vk, ok);
  w.u8(0);  // plookup_pk = None
  *len_out = w.buf.size();
  if (cap < w.buf.size()) {
    set_error("capgpu_plonk_key_serialize: buffer of %zu bytes, %zu needed", cap, w.buf.size());
    return CAPGPU_ERR_INVALID_ARG;
  }
  memcpy(out, w.buf.data(), w.buf.size());
  return CAPGPU_OK;
}

int capgpu_plonk_key_deserialize(const uint8_t* bytes, size_t len, uint64_t* srs_handle_out, uint64_t* pk_handle_out,
                                 capgpu_verifying_key* vk_out, uint64_t h_out[16], uint64_t beta_h_out[16],
                                 size_t* consumed_out) {
  CAP_CHECK_INIT();
  Context& c = ctx();
  Entry lk(c);
  if (!bytes || !srs_handle_out || !pk_handle_out) {
    set_error("capgpu_plonk_key_deserialize: bad argument");
    return CAPGPU_ERR_INVALID_ARG;
  }
  params::Reader rd(bytes, len);
  auto fail = [&](const char* why) {
    set_error("capgpu_plonk_key_deserialize: %s (byte %zu of %zu)", why, rd.pos, len);
    return CAPGPU_ERR_SERIALIZATION;
  };
  struct Span {
    const uint8_t* p;
    uint64_t len;
  };
  Span polys[18];  // internal order: 13 selectors, then 5 sigmas
  uint64_t cnt = 0;
  if (!rd.count(8, &cnt)) return fail("unexpected end of input");
  if (cnt != NW) return fail("sigmas: a TurboPlonk key has 5 of them");
  for (int i = 0; i < NW; i++) {
    if (!rd.count(32, &polys[NS + i].len)) return fail("unexpected end of input");
    polys[NS + i].p = rd.take(32 * polys[NS + i].len);
  }
  if (!rd.count(8, &cnt)) return fail("unexpected end of input");
  if (cnt != NS) return fail("selectors: a TurboPlonk key has 13 of them");
  for (int i = 0; i < NS; i++) {
    if (!rd.count(32, &polys[i].len)) return fail("unexpected end of input");
    polys[i].p = rd.take(32 * polys[i].len);
  }
  uint64_t n_ck = 0, n_gamma = 0;
  if (!rd.count(32, &n_ck)) return fail("unexpected end of input");
  const uint8_t* ck = rd.take(32 * n_ck);
  if (!rd.count(32, &n_gamma)) return fail("unexpected end of input");
  const uint8_t* gamma = rd.take(32 * n_gamma);
  capgpu_verifying_key vk;
  params::OpenKey ok;
  if (const char* why = params::read_vk(rd, &vk, &ok)) return fail(why);
  const uint8_t* tag = rd.take(1);
  if (!tag) return fail("unexpected end of input");
  if (*tag) return fail("plookup proving keys are not supported");
  const size_t n = vk.domain_size;
  if (n < 16 || (n & (n - 1)) || vk.num_inputs >= n)
    return fail("domain_size must be a power of two >= 16 above num_inputs");
  if (n_ck < n + 3) return fail("commit key shorter than domain_size + 3");
  for (int i = 0; i < 18; i++)
    if (polys[i].len > n) return fail("polynomial longer than the domain");
  // the prover's quotient kernel is specialised to the k_i of jf-plonk; refuse anything else
  for (int i = 0; i < NW; i++) {
    uint64_t want[4];
    fe_to_words(Fr::to_mont(fe_from_words(K_CANON[i])), want);
    if (memcmp(want, vk.k[i], 32) != 0) return fail("coset representatives k_i differ from jf-plonk's");
  }

  hipStream_t s = c.stream;
  int rc;
  // commit key -> device -> window table
  uint64_t srs_handle = 0;
  {
    DevTmp<g1_affine> d_ck;
    CAP_HIP(d_ck.alloc(std::max<uint64_t>(n_ck, n_gamma)));
    rc = n_gamma ? params::decompress_g1(gamma, n_gamma, d_ck, s) : CAPGPU_OK;  // validated; bytes kept below
    if (rc == CAPGPU_OK) rc = params::decompress_g1(ck, n_ck, d_ck, s);
    if (rc == CAPGPU_OK) rc = register_srs(d_ck, n_ck, &srs_handle);
    if (rc) return rc;
    if (SrsEntry* E = find_srs_entry(srs_handle)) {
      E->ck_gamma_pts.assign(gamma, gamma + 32 * n_gamma);
      E->ck_gamma_g = ok.gamma_g;
      E->has_ck_gamma_g = true;
    }
  }
  auto K = std::make_shared<ProvingKey>();
  auto bail = [&](int code) {
    // the SRS was registered a moment ago on THIS context and nobody else knows its handle: drop it here (going through
    // capgpu_srs_free would take the other contexts' locks while this one is held - out of lock order)
    (void)hipStreamSynchronize(c.stream);
    c.srs.erase(srs_handle);
    std::lock_guard<std::mutex> rlk(rt().mu);
    rt().srs.erase(srs_handle);
    return code;
  };
  if ((rc = key_init(*K, n, vk.num_inputs, srs_handle))) return bail(rc);
  const size_t ps = K->ps;
  {
    std::vector<uint8_t> coef(32 * 18 * ps, 0);
    for (int i = 0; i < 18; i++)
      if (polys[i].len) memcpy(&coef[32 * (size_t)i * ps], polys[i].p, 32 * polys[i].len);
    if ((rc = params::fr_bytes_to_mont(coef.data(), 18 * ps, K->coef, s))) return bail(rc);
  }
  // sigma evaluations on the domain (round 2 reads them)
  pad_copy(s, K->sig_eval, n, 0, K->coef + (size_t)NS * ps, ps, 0, 1, NW, n, n);
  if ((rc = run_ntt(s, K->log_n, K->sig_eval, n, NW, 0, 0))) return bail(rc);
  if ((rc = key_finish_tables(s, *K))) return bail(rc);
  std::vector<g1_affine> ha(18);
  for (int i = 0; i < NS; i++) ha[i] = params::g1_from_words(vk.selector_comms[i]);
  for (int i = 0; i < NW; i++) ha[NS + i] = params::g1_from_words(vk.sigma_comms[i]);
  key_set_vk(*K, ha);
  CAP_HIP(hipStreamSynchronize(s));
  if (vk_out) *vk_out = K->vk;
  if (h_out) params::g2_to_words(ok.h, h_out);
  if (beta_h_out) params::g2_to_words(ok.beta_h, beta_h_out);
  if (consumed_out) *consumed_out = rd.pos;
  *pk_handle_out = register_key(K);
  *srs_handle_out = srs_handle;
  return CAPGPU_OK;
}

int capgpu_plonk_key_info(uint64_t pk_handle, size_t* domain_size_out, size_t* num_inputs_out,
                          uint64_t* srs_handle_out) {
  CAP_CHECK_INIT();
  std::shared_ptr<ProvingKey> K;
  int rc = home_key(pk_handle, &K);
  if (rc) return rc;
  if (domain_size_out) *domain_size_out = K->n;
  if (num_inputs_out) *num_inputs_out = K->num_inputs;
  if (srs_handle_out) *srs_handle_out = K->srs_handle;
  return CAPGPU_OK;
}

int capgpu_plonk_free_key(uint64_t pk_handle) {
  CAP_CHECK_INIT();
  Runtime& R = rt();
  {
    std::lock_guard<std::mutex> lk(R.mu);
    auto it = R.keys.find(pk_handle);
    if (it == R.keys.end()) {
      set_error("capgpu: unknown proving key handle %llu", (unsigned long long)pk_handle);
      return CAPGPU_ERR_BAD_HANDLE;
    }
    R.keys.erase(it);
  }
  // every context that holds the key (or a replica) drains its stream before letting go of the tables
  for (auto& cp : R.ctxs) {
    Context& c = *cp;
    ScopedCtx sc(c);
    Entry lk(c);
    auto it = c.keys.find(pk_handle);
    if (it == c.keys.end()) continue;
    (void)hipStreamSynchronize(c.stream);
    c.keys.erase(it);
  }
  return CAPGPU_OK;
}

int capgpu_plonk_prove_batch_dev_ex(uint64_t pk_handle, int count, const void* d_wires, const uint64_t* pub_inputs,
                                    size_t num_inputs, const uint8_t* ext_msg, size_t ext_msg_len,
                                    const uint64_t* blinders, int input_form, capgpu_proof* proofs_out) {
  CAP_CHECK_INIT();
  if (bad_form(input_form)) return CAPGPU_ERR_INVALID_ARG;
  Context& c = ctx();
  Entry lk(c);
  if (count < 0 || (count && (!d_wires || !blinders || !proofs_out || (num_inputs && !pub_inputs)))) {
    set_error("capgpu_plonk_prove: bad argument");
    return CAPGPU_ERR_INVALID_ARG;
  }
  if (count == 0) return CAPGPU_OK;
  std::shared_ptr<ProvingKey> K;
  int rc = lookup_key(pk_handle, &K);
  if (rc) return rc;
  return prove_batch(*K, (uint32_t)count, (const fe*)d_wires, pub_inputs, num_inputs, ext_msg, ext_msg_len, blinders,
                     proofs_out, nullptr, nullptr, nullptr, nullptr, input_form);
}
int capgpu_plonk_prove_batch_dev(uint64_t pk_handle, int count, const void* d_wires, const uint64_t* pub_inputs,
                                 size_t num_inputs, const uint8_t* ext_msg, size_t ext_msg_len,
                                 const uint64_t* blinders, capgpu_proof* proofs_out) {
  return capgpu_plonk_prove_batch_dev_ex(pk_handle, count, d_wires, pub_inputs, num_inputs, ext_msg, ext_msg_len,
                                         blinders, CAPGPU_INPUT_EVALS, proofs_out);
}

// ---- dealing host-buffer batches over the device contexts -------------------------------------------------------
// A batch that arrives with its witnesses in host memory names no device, so a process that drives several (capgpu_init
// with more than one id, or CAPGPU_CONTEXTS_PER_DEVICE) cuts it into contiguous parts, one per context, each proved
// from a thread of its own: proofs are independent, so the parts' proofs are bit for bit those of the undivided batch.
// A thread that bound itself to a device (capgpu_set_device) keeps its batches there.
static int deal_min() {  // proofs a part must hold at least (smaller batches stay on one context)
  const char* e = getenv("CAPGPU_DEAL_MIN");
  const int x = e ? atoi(e) : 8;
  return x >= 1 ? x : 8;
}
// ... and at most two parts per DEVICE (CAPGPU_DEAL_PARTS_PER_DEVICE): two halves of a batch overlap on a GPU - one's
// copies, latency-bound launches and transcript steps under the other's issue-bound kernels - but four quarters are smaller
// launches for nothing (256 host-resident proofs on a device with four contexts: 1274 proofs/s as two parts, 1200 as four).
// The contexts beyond two are there for the coalescer's gathered batches (capgpu_init).
static size_t deal_max_parts(size_t contexts) {
  static const size_t per_device = [] {
    const char* e = getenv("CAPGPU_DEAL_PARTS_PER_DEVICE");
    const int x = e ? atoi(e) : 2;
    return (size_t)(x >= 1 ? x : 2);
  }();
  std::vector<int> seen;
  for (size_t i = 0; i < contexts; i++) {
    const int d = rt().ctxs[i]->device;
    if (std::find(seen.begin(), seen.end(), d) == seen.end()) seen.push_back(d);
  }
  const size_t devices = std::max<size_t>(seen.size(), 1);
  return devices * std::min<size_t>(per_device, std::max<size_t>(contexts / devices, 1));
}
static int deal(int count, const std::function<int(int first, int cnt)>& part) {
  const size_t S = num_contexts();
  const size_t parts = std::min<size_t>(std::min<size_t>(S, deal_max_parts(S)), (size_t)std::max(count / deal_min(), 1));
  // Mode A (capgpu_plonk_shard_msm): the ranks of the communicator prove the same batch in lock step, so the batch stays
  // whole and on the communicator's context - cut over contexts, one part would run unsharded and which proofs meet in
  // an exchange would depend on a cursor the ranks do not share (pick_context sends it there)
  if (comm_shard_slot() >= 0 || thread_bound_slot() >= 0 || S <= 1 || parts <= 1 || thread_entry_depth() > 0) {
    Context& c = pick_context();
    ScopedCtx sc(c);
    return part(0, count);
  }
  std::vector<int> rcs(parts, CAPGPU_OK);
  std::vector<std::string> errs(parts);
  // which contexts: the idle ones first, in slot order (a lone caller keeps to the same - warm - contexts call after call;
  // concurrent callers find different ones), the round-robin cursor for the rest
  const uint32_t start = rt().rr.fetch_add((uint32_t)parts, std::memory_order_relaxed);
  std::vector<size_t> pick;
  for (size_t i = 0; i < S && pick.size() < parts; i++) {
    std::recursive_mutex& m = rt().ctxs[i]->mu;
    if (m.try_lock()) {
      m.unlock();
      pick.push_back(i);
    }
  }
  for (size_t k = 0; pick.size() < parts && k < S; k++) {
    const size_t i = (start + k) % S;
    if (std::find(pick.begin(), pick.end(), i) == pick.end()) pick.push_back(i);
  }
  H2dTurn turn;
  // (parts on different devices have a link each: only the parts of ONE device take turns)
  bool one_device = true;
  for (size_t i = 1; i < parts; i++) one_device = one_device && rt().ctxs[pick[i]]->device == rt().ctxs[pick[0]]->device;
  const bool ordered = h2d_in_part_order() && one_device && count >= 64;
  // Two parts whose copies go in part order do not start together: the first has the device to itself while the second's
  // witnesses arrive, stays ahead through every round and would end well before it, leaving the second part's last rounds
  // alone on the device (its host steps uncovered).  The first part is therefore the larger: CAPGPU_DEAL_FIRST_SIXTEENTHS
  // (default 9: 144 + 112 of 256; 8 = equal halves).
  static const uint64_t first16 = [] {
    const char* e = getenv("CAPGPU_DEAL_FIRST_SIXTEENTHS");
    const int x = e ? atoi(e) : 9;
    return (uint64_t)(x >= 4 && x <= 12 ? x : 9);
  }();
  auto cut = [&](size_t i) -> int {  // first proof of part i
    if (i == 0) return 0;
    if (i >= parts) return count;
    if (ordered && parts == 2) return (int)((uint64_t)count * first16 / 16);
    return (int)((uint64_t)count * i / parts);
  };
  auto body = [&](size_t i) {
    const int first = cut(i), last = cut(i + 1);
    ScopedCtx sc(*rt().ctxs[pick[i]]);
    if (ordered) {
      tl_h2d_turn = &turn;
      tl_h2d_index = (uint32_t)i;
    }
    rcs[i] = part(first, last - first);
    if (rcs[i]) errs[i] = last_error();
    tl_h2d_turn = nullptr;
    turn.pass((uint32_t)i);  // (whatever happened inside: a part that failed before its copies must not hold the others up)
  };
  std::vector<std::thread> th;
  for (size_t i = 1; i < parts; i++) th.emplace_back(body, i);
  body(0);
  for (auto& t : th) t.join();
  for (size_t i = 0; i < parts; i++)
    if (rcs[i]) {
      set_error("%s", errs[i].c_str());
      return rcs[i];
    }
  return CAPGPU_OK;
}

int capgpu_plonk_prove_batch_ex(uint64_t pk_handle, int count, const uint64_t* wires, const uint64_t* pub_inputs,
                                size_t num_inputs, const uint8_t* ext_msg, size_t ext_msg_len, const uint64_t* blinders,
                                int input_form, capgpu_proof* proofs_out) {
  CAP_CHECK_INIT();
  if (bad_form(input_form)) return CAPGPU_ERR_INVALID_ARG;
  if (count < 0 || (count && !wires)) {
    set_error("capgpu_plonk_prove: bad argument");
    return CAPGPU_ERR_INVALID_ARG;
  }
  if (count == 0) return CAPGPU_OK;
  std::shared_ptr<ProvingKey> K0;
  int rc0 = home_key(pk_handle, &K0);
  if (rc0) return rc0;
  if (!blinders || !proofs_out || (num_inputs && !pub_inputs)) {
    set_error("capgpu_plonk_prove: bad argument");
    return CAPGPU_ERR_INVALID_ARG;
  }
  const size_t n = K0->n;
  return deal(count, [&](int first, int cnt) -> int {
    Context& c = ctx();
    Entry lk(c);
    std::shared_ptr<ProvingKey> K;
    int rc = lookup_key(pk_handle, &K);
    if (rc) return rc;
    rc = scratch_reserve(c.stage_b, sizeof(fe) * (size_t)cnt * NW * n);
    if (rc) return rc;
    // the columns are copied inside round 1, chunk by chunk, behind the commitments of the chunk before
    std::vector<const uint64_t*> rows(cnt);
    for (int i = 0; i < cnt; i++) rows[i] = wires + (size_t)4 * (first + i) * NW * n;
    return prove_batch(*K, (uint32_t)cnt, (const fe*)c.stage_b.p, pub_inputs ? pub_inputs + (size_t)4 * first * num_inputs : nullptr,
                       num_inputs, ext_msg, ext_msg_len, blinders + (size_t)4 * 13 * first, proofs_out + first, nullptr,
                       nullptr, nullptr, rows.data(), input_form);
  });
}
int capgpu_plonk_prove_batch(uint64_t pk_handle, int count, const uint64_t* wires, const uint64_t* pub_inputs,
                             size_t num_inputs, const uint8_t* ext_msg, size_t ext_msg_len, const uint64_t* blinders,
                             capgpu_proof* proofs_out) {
  return capgpu_plonk_prove_batch_ex(pk_handle, count, wires, pub_inputs, num_inputs, ext_msg, ext_msg_len, blinders,
                                     CAPGPU_INPUT_EVALS, proofs_out);
}

// Proofs of several proving keys in one device batch (see prove_batch): pk_handles[i] is the key of proof i.
int capgpu_plonk_prove_multi_dev_ex(const uint64_t* pk_handles, int count, const void* d_wires,
                                    const uint64_t* pub_inputs, size_t num_inputs, const uint8_t* const* ext_msgs,
                                    const size_t* ext_msg_lens, const uint64_t* blinders, int input_form,
                                    capgpu_proof* proofs_out) {
  CAP_CHECK_INIT();
  if (bad_form(input_form)) return CAPGPU_ERR_INVALID_ARG;
  Context& c = ctx();
  Entry lk(c);
  if (count < 0 || (count && (!pk_handles || !d_wires || !blinders || !proofs_out || (num_inputs && !pub_inputs) ||
                              (ext_msgs && !ext_msg_lens)))) {
    set_error("capgpu_plonk_prove_multi: bad argument");
    return CAPGPU_ERR_INVALID_ARG;
  }
  if (count == 0) return CAPGPU_OK;
  std::vector<std::shared_ptr<ProvingKey>> hold(count);
  std::vector<const ProvingKey*> keys(count);
  for (int i = 0; i < count; i++) {
    int rc = lookup_key(pk_handles[i], &hold[i]);
    if (rc) return rc;
    keys[i] = hold[i].get();
  }
  return prove_batch(*keys[0], (uint32_t)count, (const fe*)d_wires, pub_inputs, num_inputs, nullptr, 0, blinders,
                     proofs_out, ext_msgs, ext_msg_lens, &keys, nullptr, input_form);
}
int capgpu_plonk_prove_multi_dev(const uint64_t* pk_handles, int count, const void* d_wires, const uint64_t* pub_inputs,
                                 size_t num_inputs, const uint8_t* const* ext_msgs, const size_t* ext_msg_lens,
                                 const uint64_t* blinders, capgpu_proof* proofs_out) {
  return capgpu_plonk_prove_multi_dev_ex(pk_handles, count, d_wires, pub_inputs, num_inputs, ext_msgs, ext_msg_lens,
                                         blinders, CAPGPU_INPUT_EVALS, proofs_out);
}

int capgpu_plonk_prove_multi_ex(const uint64_t* pk_handles, int count, const uint64_t* wires,
                                const uint64_t* pub_inputs, size_t num_inputs, const uint8_t* const* ext_msgs,
                                const size_t* ext_msg_lens, const uint64_t* blinders, int input_form,
                                capgpu_proof* proofs_out) {
  CAP_CHECK_INIT();
  if (bad_form(input_form)) return CAPGPU_ERR_INVALID_ARG;
  if (count < 0 || (count && (!wires || !pk_handles))) {
    set_error("capgpu_plonk_prove_multi: bad argument");
    return CAPGPU_ERR_INVALID_ARG;
  }
  if (count == 0) return CAPGPU_OK;
  std::shared_ptr<ProvingKey> K0;
  int rc0 = home_key(pk_handles[0], &K0);
  if (rc0) return rc0;
  if (!blinders || !proofs_out || (num_inputs && !pub_inputs) || (ext_msgs && !ext_msg_lens)) {
    set_error("capgpu_plonk_prove_multi: bad argument");
    return CAPGPU_ERR_INVALID_ARG;
  }
  const size_t n = K0->n;
  return deal(count, [&](int first, int cnt) -> int {
    Context& c = ctx();
    Entry lk(c);
    std::vector<std::shared_ptr<ProvingKey>> hold(cnt);
    std::vector<const ProvingKey*> keys(cnt);
    int rc;
    for (int i = 0; i < cnt; i++) {
      if ((rc = lookup_key(pk_handles[first + i], &hold[i]))) return rc;
      keys[i] = hold[i].get();
    }
    // a part's rows carry `num_inputs` = the largest count among the keys of the WHOLE call; prove_batch wants the
    // largest among its own keys: re-pack when the part's maximum is smaller
    size_t ni = 0;
    for (int i = 0; i < cnt; i++) ni = std::max(ni, keys[i]->num_inputs);
    std::vector<uint64_t> pubs;
    const uint64_t* pp = pub_inputs ? pub_inputs + (size_t)4 * first * num_inputs : nullptr;
    if (ni != num_inputs) {
      if (ni > num_inputs) {
        set_error("capgpu_plonk_prove_multi: rows of %zu public inputs given, the keys need %zu", num_inputs, ni);
        return CAPGPU_ERR_INVALID_ARG;
      }
      pubs.assign((size_t)4 * ni * cnt + 4, 0);
      for (int i = 0; i < cnt; i++)
        if (ni) memcpy(&pubs[(size_t)4 * ni * i], pp + (size_t)4 * num_inputs * i, 32 * ni);
      pp = pubs.data();
    }
    if ((rc = scratch_reserve(c.stage_b, sizeof(fe) * (size_t)cnt * NW * n))) return rc;
    std::vector<const uint64_t*> rows(cnt);
    for (int i = 0; i < cnt; i++) rows[i] = wires + (size_t)4 * (first + i) * NW * n;
    return prove_batch(*keys[0], (uint32_t)cnt, (const fe*)c.stage_b.p, pp, ni, nullptr, 0,
                       blinders + (size_t)4 * 13 * first, proofs_out + first, ext_msgs ? ext_msgs + first : nullptr,
                       ext_msg_lens ? ext_msg_lens + first : nullptr, &keys, rows.data(), input_form);
  });
}
int capgpu_plonk_prove_multi(const uint64_t* pk_handles, int count, const uint64_t* wires, const uint64_t* pub_inputs,
                             size_t num_inputs, const uint8_t* const* ext_msgs, const size_t* ext_msg_lens,
                             const uint64_t* blinders, capgpu_proof* proofs_out) {
  return capgpu_plonk_prove_multi_ex(pk_handles, count, wires, pub_inputs, num_inputs, ext_msgs, ext_msg_lens, blinders,
                                     CAPGPU_INPUT_EVALS, proofs_out);
}

// one gathered batch: device staging of every request's wires, per-proof messages and keys; a batch that fails because
// ONE witness does not satisfy its circuit is re-run request by request so that only its owner sees the failure.
// Runs on the calling thread's context (the leader took its lock).
static void run_coalesced(std::vector<ProveReq*>& reqs) {
  Context& c = ctx();
  Entry lk(c);
  auto fail_all = [&](int rc) {
    for (ProveReq* r : reqs) {
      if (r->rc != CAPGPU_OK) continue;
      r->rc = rc;
      r->err = capgpu_last_error();
    }
  };
  // one request on this context (the recursive lock is held)
  auto prove_one = [&](ProveReq* r) {
    std::shared_ptr<ProvingKey> K;
    int rc = lookup_key(r->pk, &K);
    if (rc == CAPGPU_OK) rc = scratch_reserve(c.stage_b, sizeof(fe) * NW * K->n);
    if (rc == CAPGPU_OK) {
      const uint64_t* row = r->wires;
      rc = prove_batch(*K, 1, (const fe*)c.stage_b.p, r->pubs, r->num_inputs, r->msg, r->msg_len, r->blinders, r->out,
                       nullptr, nullptr, nullptr, &row, r->form);
    }
    r->rc = rc;
    if (rc) r->err = capgpu_last_error();
  };
  std::vector<ProveReq*> good;
  std::vector<std::shared_ptr<ProvingKey>> hold;
  for (ProveReq* r : reqs) {
    std::shared_ptr<ProvingKey> K;
    int rc = lookup_key(r->pk, &K);
    if (rc == CAPGPU_OK && r->num_inputs != K->num_inputs) {
      set_error("capgpu_plonk_prove: %zu public inputs given, key expects %zu", r->num_inputs, K->num_inputs);
      rc = CAPGPU_ERR_INVALID_ARG;
    }
    if (rc != CAPGPU_OK) {
      r->rc = rc;
      r->err = capgpu_last_error();
      continue;
    }
    good.push_back(r);
    hold.push_back(K);
  }
  if (good.empty()) return;
  const size_t g = good.size(), n = hold[0]->n;
  size_t ni = 0;  // row length of the public inputs: the largest count among the batch's keys
  bool mixed = false;
  for (size_t i = 0; i < g; i++) {
    ni = std::max(ni, hold[i]->num_inputs);
    mixed = mixed || hold[i].get() != hold[0].get();
  }
  bool recompute = false;
  for (size_t i = 0; i < g; i++) recompute = recompute || hold[i]->recompute;
  if (mixed && recompute) {  // the reference-schedule test mode keeps one key per batch: prove these one by one
    for (size_t i = 0; i < g; i++) prove_one(good[i]);
    coalescer().batches += g;
    coalescer().proofs += g;
    return;
  }
  const size_t per = sizeof(fe) * NW * n;
  // gathered batches differ in size from one to the next: scratch that has to grow for one grows to the next multiple of
  // 32 proofs (64 at least) at once - a context otherwise re-allocates gigabytes (0.1 - 0.6 s each time) whenever a batch
  // is a few proofs larger than every batch it has seen (round 6: it cost the bench's coalesced leg a third)
  struct GrowthScale {
    double prev;
    explicit GrowthScale(double f) : prev(scratch_growth_scale()) { scratch_growth_scale() = f; }
    ~GrowthScale() { scratch_growth_scale() = prev; }
  } growth((double)std::max<size_t>(64, (g + 31) / 32 * 32) / (double)g);
  int rc = scratch_reserve(c.stage_b, per * g);
  if (rc) return fail_all(rc);
  std::vector<uint64_t> pubs(4 * ni * g + 4, 0), blind(4 * 13 * g);
  std::vector<const uint8_t*> msgs(g);
  std::vector<size_t> lens(g);
  std::vector<capgpu_proof> out(g);
  std::vector<const ProvingKey*> keys(g);
  std::vector<const uint64_t*> rows(g);  // every caller's own buffer: copied inside round 1, chunk by chunk
  size_t staged = 0;
  for (size_t i = 0; i < g; i++) staged += good[i]->d_wires != nullptr;
  for (size_t i = 0; i < g; i++) {
    rows[i] = good[i]->wires;
    if (good[i]->num_inputs) memcpy(&pubs[4 * ni * i], good[i]->pubs, 32 * good[i]->num_inputs);
    memcpy(&blind[4 * 13 * i], good[i]->blinders, 32 * 13);
    msgs[i] = good[i]->msg;
    lens[i] = good[i]->msg_len;
    keys[i] = hold[i].get();
  }
  bool resident = false;
  if (staged) {
    // the callers copied their witnesses when they arrived (StagePool): gather them into the batch's input array behind
    // their copies; a request that got no slot goes host -> device here, on the batch's stream
    trace("co_gather_staged", (int64_t)staged, (int64_t)g);
    hipError_t e = hipSuccess;
    for (size_t i = 0; i < g && e == hipSuccess; i++) {
      char* dst = (char*)c.stage_b.p + per * i;
      if (good[i]->d_wires) {
        e = hipStreamWaitEvent(c.stream, good[i]->staged, 0);
        if (e == hipSuccess) e = hipMemcpyAsync(dst, good[i]->d_wires, per, hipMemcpyDeviceToDevice, c.stream);
      } else {
        e = hipMemcpyAsync(dst, good[i]->wires, per, hipMemcpyHostToDevice, c.stream);
      }
    }
    if (e != hipSuccess) return fail_all(hip_fail(e, "gathering the staged witnesses"));
    resident = true;
  }
  rc = prove_batch(*keys[0], (uint32_t)g, (const fe*)c.stage_b.p, pubs.data(), ni, nullptr, 0, blind.data(), out.data(),
                   msgs.data(), lens.data(), mixed ? &keys : nullptr, resident ? nullptr : rows.data(), good[0]->form);
  if (rc == CAPGPU_OK) rc = take_launch_error();
  if (rc == CAPGPU_OK) {
    for (size_t i = 0; i < g; i++) *good[i]->out = out[i];
  } else if (rc == CAPGPU_ERR_PROOF && g > 1) {
    for (size_t i = 0; i < g; i++) prove_one(good[i]);  // find the owner(s) of the unsatisfied witness
  } else {
    for (ProveReq* r : good) {
      r->rc = rc;
      r->err = capgpu_last_error();
    }
  }
  coalescer().batches++;
  coalescer().proofs += g;
}

int capgpu_plonk_prove_ex(uint64_t pk_handle, const uint64_t* wires, const uint64_t* pub_inputs, size_t num_inputs,
                          const uint8_t* ext_msg, size_t ext_msg_len, const uint64_t* blinders, int input_form,
                          capgpu_proof* proof_out) {
  Coalescer& co = coalescer();
  if (co.window_us == 0)
    return capgpu_plonk_prove_batch_ex(pk_handle, 1, wires, pub_inputs, num_inputs, ext_msg, ext_msg_len, blinders,
                                       input_form, proof_out);
  CAP_CHECK_INIT();
  if (bad_form(input_form)) return CAPGPU_ERR_INVALID_ARG;
  if (!wires || !blinders || !proof_out || (num_inputs && !pub_inputs)) {
    set_error("capgpu_plonk_prove: bad argument");
    return CAPGPU_ERR_INVALID_ARG;
  }
  ProveReq req{pk_handle, wires, pub_inputs, num_inputs, ext_msg, ext_msg_len, blinders, proof_out, input_form};
  std::unique_lock<std::mutex> lk(co.mu);
  uint64_t group = 0;
  size_t key_n = 0;
  {
    // (key, form) -> group: bit 7 of the group id is the input form, the bits below it the domain size
    const uint64_t gkey = (pk_handle << 1) | (uint64_t)(input_form == CAPGPU_INPUT_COEFFS);
    auto it = co.group_of.find(gkey);
    if (it == co.group_of.end()) {
      // first call for this key: its domain size and SRS make the group
      lk.unlock();
      size_t kn = 0;
      uint64_t ksrs = 0;
      int rc = capgpu_plonk_key_info(pk_handle, &kn, nullptr, &ksrs);
      if (rc) return rc;
      lk.lock();
      group = (ksrs << 8) ^ (uint64_t)__builtin_ctzll(kn | (1ull << 63)) ^
              ((uint64_t)(input_form == CAPGPU_INPUT_COEFFS) << 7);
      co.group_of[gkey] = {group, kn};
      key_n = kn;
    } else {
      group = it->second.first;
      key_n = it->second.second;
    }
  }
  // this caller's witness starts its way to the device now, on the staging pool's stream, while the batch it will be
  // part of is still being gathered (one bound device only: the slot must live where the batch runs)
  StageSlot slot;
  {
    int devices = 0;
    if (StagePool::enabled() && key_n && capgpu_physical_device_count(&devices) == CAPGPU_OK && devices == 1 &&
        comm_shard_slot() < 0) {
      co.arriving++;  // (a leader's window stays open for callers that are on their way)
      lk.unlock();
      const int dev = ctx().device;
      const size_t per = sizeof(fe) * NW * key_n;
      slot = stage_pool().acquire(dev, per);
      if (slot.d) {
        hipError_t e = hipMemcpyAsync(slot.d, wires, per, hipMemcpyHostToDevice, slot.stream);
        if (e == hipSuccess) e = hipEventRecord(slot.ev, slot.stream);
        if (e != hipSuccess) {  // not fatal: the witness travels with the batch instead
          (void)hipGetLastError();
          (void)hipStreamSynchronize(slot.stream);
          stage_pool().release(slot);
          slot = StageSlot{};
        }
      }
      req.d_wires = slot.d;
      req.staged = slot.ev;
      trace("co_prestaged", slot.d != nullptr);
      lk.lock();
      co.arriving--;
    }
  }
  struct SlotReturn {  // (the batch that read the slot is done when submit returns; error paths included)
    StageSlot& s;
    ~SlotReturn() { stage_pool().release(s); }
  } slot_return{slot};
  // what the protocol needs from the runtime: free device contexts and the prover
  struct Hooks {
    // a free device context: any of the process's (several batches are then in flight, one per context), or the one
    // this thread bound itself to; mode A of config 4: gathered batches go to the communicator's context, whole
    void* acquire() {
      const int ss = comm_shard_slot();
      const int bound = ss >= 0 && (size_t)ss < num_contexts() ? ss : thread_bound_slot();
      if (bound >= 0) {
        Context* b = rt().ctxs[(size_t)bound].get();
        return b->mu.try_lock() ? b : nullptr;
      }
      return try_acquire_context();
    }
    void* acquire_second() {
      if (thread_bound_slot() >= 0 || comm_shard_slot() >= 0 || num_contexts() <= 1) return nullptr;
      Context* c2 = try_acquire_context();
      if (c2) c2->mu.unlock();  // the helper thread locks it itself (the lock belongs to the thread that takes it)
      return c2;
    }
    void run(void* ctx, std::vector<ProveReq*>& reqs, bool on_helper) {
      Context& c = *static_cast<Context*>(ctx);
      ScopedCtx sc(c);
      if (on_helper) {
        Entry elk(c);
        run_coalesced(reqs);
      } else {
        run_coalesced(reqs);  // re-enters the (recursive) context lock this thread holds
      }
    }
    void release(void* ctx) { static_cast<Context*>(ctx)->mu.unlock(); }
    size_t deal_min() { return (size_t)::deal_min(); }
    size_t split_eighths() {  // CAPGPU_COALESCE_SPLIT = eighths of the first part
      static const size_t eighths = [] {
        const char* e = getenv("CAPGPU_COALESCE_SPLIT");
        const int x = e ? atoi(e) : 3;
        return (size_t)(x >= 1 && x <= 4 ? x : 3);
      }();
      return eighths;
    }
    // batch parts running at once before a leader keeps collecting instead of taking another free context:
    // CAPGPU_COALESCE_INFLIGHT (default 2) per bound DEVICE - two large batches fill a GPU, a third only fragments them
    size_t max_in_flight() {
      static const size_t per_device = [] {
        const char* e = getenv("CAPGPU_COALESCE_INFLIGHT");
        const int x = e ? atoi(e) : 2;
        return (size_t)(x >= 1 && x <= 64 ? x : 2);
      }();
      int devices = 1;
      (void)capgpu_physical_device_count(&devices);
      return per_device * (size_t)std::max(devices, 1);
    }
    bool early_release() {  // CAPGPU_COALESCE_EARLY=0: a cut batch's callers return when both parts are done, as before
      static const bool early = [] {
        const char* e = getenv("CAPGPU_COALESCE_EARLY");
        return !e || atoi(e) != 0;
      }();
      return early;
    }
  } hooks;
  co.submit(lk, req, group, hooks);
  if (req.rc != CAPGPU_OK) set_error("%s", req.err.c_str());
  return req.rc;
}

int capgpu_plonk_prove(uint64_t pk_handle, const uint64_t* wires, const uint64_t* pub_inputs, size_t num_inputs,
                       const uint8_t* ext_msg, size_t ext_msg_len, const uint64_t* blinders, capgpu_proof* proof_out) {
  return capgpu_plonk_prove_ex(pk_handle, wires, pub_inputs, num_inputs, ext_msg, ext_msg_len, blinders,
                               CAPGPU_INPUT_EVALS, proof_out);
}

int capgpu_plonk_set_coalescing(uint32_t window_us, uint32_t max_batch) {
  Coalescer& co = coalescer();
  std::lock_guard<std::mutex> lk(co.mu);
  co.window_us = window_us;
  co.max_batch = max_batch ? max_batch : 256;
  return CAPGPU_OK;
}

int capgpu_plonk_set_wire_commit(int mode) {
  if (mode < -1 || mode > 1) {
    set_error("capgpu_plonk_set_wire_commit: mode must be -1 (default), 0 (coefficients) or 1 (evaluations)");
    return CAPGPU_ERR_INVALID_ARG;
  }
  g_wire_commit.store(mode);
  return CAPGPU_OK;
}

int capgpu_plonk_graph_stats(uint64_t* segments_captured_out, uint64_t* segments_replayed_out) {
  if (segments_captured_out) *segments_captured_out = g_graph_captured.load();
  if (segments_replayed_out) *segments_replayed_out = g_graph_replayed.load();
  return CAPGPU_OK;
}

int capgpu_plonk_coalescing_stats(uint64_t* batches_out, uint64_t* proofs_out) {
  Coalescer& co = coalescer();
  if (batches_out) *batches_out = co.batches.load();
  if (proofs_out) *proofs_out = co.proofs.load();
  return CAPGPU_OK;
}

}  // extern "C"
