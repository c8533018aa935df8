// Persistent host thread pool for the per-proof work between prover rounds (Keccak transcripts, challenge
// arithmetic, Jacobian -> affine).  That work is independent across the proofs of a batch; the reference runs it
// under rayon (src/utils/params_builder.rs:194-226).  A serial loop leaves the GPU idle for ~20 % of a 64-proof step,
// and spawning threads per call costs about a millisecond, seven times per step - so workers are created once and
// parked on a condition variable.  One pool per device context: contexts prove concurrently (one per GPU of a
// single-process job, context.hpp) and must not queue up behind each other's host phases.
#pragma once
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace cap {

class HostPool {
 public:
  explicit HostPool(unsigned threads) : nt_(std::max(threads, 1u)) {
    for (unsigned t = 1; t < nt_; t++) workers_.emplace_back([this] { loop(); });
  }
  ~HostPool() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
      epoch_++;
    }
    cv_.notify_all();
    for (auto& w : workers_) w.join();
  }
  HostPool(const HostPool&) = delete;
  HostPool& operator=(const HostPool&) = delete;
  unsigned size() const { return nt_; }
  // threads a context's pool gets when `contexts` of them share the host: CAPGPU_HOST_THREADS (per context) or the
  // hardware threads divided among the contexts - of ALL the processes a launcher put on this host (one process per
  // GPU: LOCAL_WORLD_SIZE as torchrun and its kin export it; eight ranks must not each size their pools for the whole
  // machine) -, 32 at most (more does not shorten the 256-transcript phases)
  static unsigned default_threads(unsigned contexts) {
    const char* e = getenv("CAPGPU_HOST_THREADS");
    const char* lw = getenv("LOCAL_WORLD_SIZE");
    const unsigned procs = lw && atoi(lw) > 0 ? (unsigned)atoi(lw) : 1u;
    const unsigned sharers = std::max(contexts, 1u) * procs;
    unsigned hw = std::thread::hardware_concurrency();
    unsigned v = e ? (unsigned)atoi(e) : std::max((hw + sharers - 1) / sharers, 2u);
    return std::min(std::max(v, 1u), 32u);
  }
  // runs job(i) for i in [0, count); the caller takes part
  void run(uint32_t count, const std::function<void(uint32_t)>& job) {
    std::lock_guard<std::mutex> serial(run_mu_);  // one parallel region at a time
    {
      std::lock_guard<std::mutex> lk(mu_);
      job_ = &job;
      count_ = count;
      next_.store(0, std::memory_order_relaxed);
      pending_ = (unsigned)workers_.size();
      epoch_++;
    }
    cv_.notify_all();
    drain();
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [&] { return pending_ == 0; });
    job_ = nullptr;
  }

 private:
  void drain() {
    for (;;) {
      uint32_t i = next_.fetch_add(1, std::memory_order_relaxed);
      if (i >= count_) break;
      (*job_)(i);
    }
  }
  void loop() {
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return epoch_ != seen; });
        seen = epoch_;
        if (stop_) return;
      }
      drain();
      std::lock_guard<std::mutex> lk(mu_);
      if (--pending_ == 0) done_cv_.notify_one();
    }
  }
  unsigned nt_ = 1;
  std::vector<std::thread> workers_;
  std::mutex mu_, run_mu_;
  std::condition_variable cv_, done_cv_;
  const std::function<void(uint32_t)>* job_ = nullptr;
  uint32_t count_ = 0;
  std::atomic<uint32_t> next_{0};
  unsigned pending_ = 0;
  uint64_t epoch_ = 0;
  bool stop_ = false;
};

}  // namespace cap
