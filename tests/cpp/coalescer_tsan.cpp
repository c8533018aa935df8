// Host-only test of the call-coalescing protocol (cap_amd/csrc/coalescer.hpp) - the code that runs inside
// capgpu_plonk_prove_ex when capgpu_plonk_set_coalescing is on - with stub device contexts (a recursive mutex each, as
// cap::Context has) and a stub prover.  Built with -fsanitize=thread by tests/test_coalescer_host.py: the reference's calling
// pattern (many rayon threads, one prove() per note each, src/utils/params_builder.rs:194-226) in closed loop, several
// groups, batches cut over two contexts.  Checks: every request is served exactly once, by a batch of its own group, no
// batch exceeds max_batch, a context never runs two batches at once, every caller gets ITS result, nobody is left
// waiting; ThreadSanitizer checks the rest.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <vector>

#include "../../cap_amd/csrc/coalescer.hpp"

struct Req {
  uint64_t group = 0;
  uint64_t input = 0;
  uint64_t result = 0;
  int served = 0;
  bool done = false;
};

struct StubCtx {
  std::recursive_mutex mu;
  std::atomic<int> running{0};
};

static std::atomic<int> failures{0};
#define CHECK(x)                                                        \
  do {                                                                  \
    if (!(x)) {                                                         \
      failures++;                                                       \
      fprintf(stderr, "CHECK failed line %d: %s\n", __LINE__, #x);      \
    }                                                                   \
  } while (0)

struct Hooks {
  std::vector<StubCtx>& ctxs;
  cap::CoalescerCore<Req>& co;
  std::atomic<uint32_t> rr{0};
  std::atomic<uint64_t> max_seen{0}, cut_batches{0};
  uint32_t max_batch;
  Hooks(std::vector<StubCtx>& c, cap::CoalescerCore<Req>& k, uint32_t mb) : ctxs(c), co(k), max_batch(mb) {}
  StubCtx* try_acquire() {
    const uint32_t start = rr.load(std::memory_order_relaxed);
    for (size_t i = 0; i < ctxs.size(); i++) {
      StubCtx* c = &ctxs[(start + i) % ctxs.size()];
      if (c->mu.try_lock()) {
        rr.store((uint32_t)((start + i + 1) % ctxs.size()), std::memory_order_relaxed);
        return c;
      }
    }
    return nullptr;
  }
  size_t in_flight_limit = 2;
  size_t max_in_flight() { return in_flight_limit; }
  void* acquire() { return try_acquire(); }
  void* acquire_second() {
    if (ctxs.size() <= 1) return nullptr;
    StubCtx* c2 = try_acquire();
    if (c2) c2->mu.unlock();
    return c2;
  }
  void run(void* ctx, std::vector<Req*>& reqs, bool on_helper) {
    StubCtx& c = *static_cast<StubCtx*>(ctx);
    if (on_helper) c.mu.lock();  // the leader's own context is already locked by acquire()
    CHECK(c.running.fetch_add(1) == 0);
    CHECK(!reqs.empty() && reqs.size() <= max_batch);
    if (on_helper) cut_batches++;
    uint64_t seen = max_seen.load();
    while (reqs.size() > seen && !max_seen.compare_exchange_weak(seen, reqs.size())) {
    }
    for (Req* r : reqs) {
      CHECK(r->group == reqs[0]->group);
      CHECK(!r->done);
      r->served++;
      r->result = r->input * 3 + 1;
    }
    std::this_thread::sleep_for(std::chrono::microseconds(150 + 10 * reqs.size()));  // "the device is busy"
    c.running.fetch_sub(1);
    co.batches++;
    co.proofs += reqs.size();
    if (on_helper) c.mu.unlock();
  }
  void release(void* ctx) { static_cast<StubCtx*>(ctx)->mu.unlock(); }
  size_t deal_min() { return 4; }
  size_t split_eighths() { return 3; }
  bool early = true;
  bool early_release() { return early; }
};

static void scenario(int n_ctx, int threads, int calls, int groups, uint32_t window_us, uint32_t max_batch, bool early) {
  cap::CoalescerCore<Req> co;
  co.window_us = window_us;
  co.max_batch = max_batch;
  std::vector<StubCtx> ctxs(n_ctx);
  Hooks h(ctxs, co, max_batch);
  h.early = early;
  std::atomic<uint64_t> served{0};
  std::vector<std::thread> th;
  for (int t = 0; t < threads; t++)
    th.emplace_back([&, t] {
      for (int k = 0; k < calls; k++) {
        Req r;  // a stack object, as in capgpu_plonk_prove_ex: must not be touched once done
        r.group = (uint64_t)((t + k) % groups);
        r.input = (uint64_t)t * 1000 + k;
        std::unique_lock<std::mutex> lk(co.mu);
        co.submit(lk, r, r.group, h);
        lk.unlock();
        CHECK(r.done && r.served == 1 && r.result == r.input * 3 + 1);
        served++;
      }
    });
  for (auto& x : th) x.join();
  CHECK(served.load() == (uint64_t)threads * calls);
  CHECK(co.proofs.load() == (uint64_t)threads * calls);
  CHECK(co.batches.load() >= 1 && co.batches.load() <= (uint64_t)threads * calls);
  for (auto& kv : co.pending) CHECK(kv.second.empty());
  for (auto& kv : co.leader) CHECK(!kv.second);
  for (auto& c : ctxs) {
    CHECK(c.mu.try_lock());
    c.mu.unlock();
  }
  printf("ctx %d threads %d x %d calls, %d groups, window %u us, max %u, early %d: %llu batches (largest %llu, %llu on a helper)\n",
         n_ctx, threads, calls, groups, window_us, max_batch, (int)early, (unsigned long long)co.batches.load(),
         (unsigned long long)h.max_seen.load(), (unsigned long long)h.cut_batches.load());
}

int main() {
  scenario(1, 16, 6, 1, 200, 256, true);    // one context: batches queue up behind each other
  scenario(4, 32, 6, 1, 200, 256, true);    // four contexts (one bound GPU's default): cut batches, several in flight
  scenario(4, 32, 6, 3, 100, 8, true);      // three groups (domain sizes), small max_batch: full batches wake the leader
  scenario(2, 24, 5, 2, 300, 256, false);   // callers of a cut batch released together (CAPGPU_COALESCE_EARLY=0)
  scenario(3, 1, 10, 1, 50, 256, true);     // a lone caller leads every batch itself
  if (failures.load()) {
    fprintf(stderr, "%d check(s) failed\n", failures.load());
    return 1;
  }
  printf("OK\n");
  return 0;
}
