// Kernel launch wrapper with optional per-kernel HIP-event timing (the numbers
// bench.py reports as roofline.achieved come from here: events are recorded on
// the stream the kernel is launched on, not on torch's current stream).
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace cap {

class Profiler {
 public:
  bool on = false;
  struct Pending {
    const char* name;
    hipEvent_t a, b;
  };
  struct Stat {
    double ms = 0;
    uint64_t launches = 0;
  };
  void begin(const char* name, hipStream_t s) {
    Pending p;
    p.name = name;
    p.a = get_event();
    p.b = get_event();
    hipEventRecord(p.a, s);
    cur_ = p;
  }
  void end(hipStream_t s) {
    hipEventRecord(cur_.b, s);
    pending_.push_back(cur_);
  }
  // resolve all pending pairs (synchronises on their end events)
  void fold() {
    for (auto& p : pending_) {
      hipEventSynchronize(p.b);
      float ms = 0;
      hipEventElapsedTime(&ms, p.a, p.b);
      auto& st = stats_[p.name];
      st.ms += ms;
      st.launches += 1;
      pool_.push_back(p.a);
      pool_.push_back(p.b);
    }
    pending_.clear();
  }
  void reset() {
    fold();
    stats_.clear();
  }
  const std::map<std::string, Stat>& stats() {
    fold();
    return stats_;
  }

 private:
  hipEvent_t get_event() {
    if (!pool_.empty()) {
      hipEvent_t e = pool_.back();
      pool_.pop_back();
      return e;
    }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
  }
  Pending cur_{};
  std::vector<Pending> pending_;
  std::vector<hipEvent_t> pool_;
  std::map<std::string, Stat> stats_;
};

Profiler& profiler();

// First failed kernel launch since the last take_launch_error(): a bad launch configuration (grid too large, too much
// LDS) is reported by the entry point that issued it instead of surfacing at some later synchronisation.
struct LaunchError {
  hipError_t code = hipSuccess;
  const char* kernel = nullptr;
};
LaunchError& launch_error();
inline void note_launch(const char* name) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    LaunchError& le = launch_error();
    if (le.code == hipSuccess) {
      le.code = e;
      le.kernel = name;
    }
  }
}

template <class K, class... A>
inline void launch(const char* name, K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t stream, A... args) {
  Profiler& p = profiler();
  if (p.on) {
    p.begin(name, stream);
    hipLaunchKernelGGL(kernel, grid, block, lds, stream, args...);
    p.end(stream);
  } else {
    hipLaunchKernelGGL(kernel, grid, block, lds, stream, args...);
  }
  note_launch(name);
}

}  // namespace cap
