// Host-only check of the library's phase trace (cap_amd/csrc/trace.hpp): off means nothing is recorded; on, eight threads
// emit concurrently and every event comes out of the dump once, oldest first, with its tag and arguments; a new enable
// starts a fresh trace.  tests/test_coalescer_host.py builds and runs it (also under ThreadSanitizer).
#include "../../cap_amd/csrc/trace.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

#define CHECK(x)                                                     \
  do {                                                               \
    if (!(x)) {                                                      \
      fprintf(stderr, "CHECK failed: %s (line %d)\n", #x, __LINE__); \
      exit(1);                                                       \
    }                                                                \
  } while (0)

int main(int argc, char** argv) {
  const char* path = argc > 1 ? argv[1] : "/tmp/capgpu_trace_check.txt";
  cap::trace("never", 1, 2);  // off: dropped
  cap::trace_enable(true);
  const int T = 8, N = 5000;
  std::vector<std::thread> th;
  for (int t = 0; t < T; t++)
    th.emplace_back([t] {
      for (int i = 0; i < N; i++) cap::trace(t % 2 ? "odd" : "even", t, i);
    });
  for (auto& x : th) x.join();
  cap::trace_enable(false);
  cap::trace("after", 0, 0);  // off again: dropped
  CHECK(cap::trace_dump(path) == (long)T * N);
  FILE* f = fopen(path, "r");
  CHECK(f != nullptr);
  std::map<long long, long long> next;  // per emitting thread: the next index expected (events of one thread stay ordered)
  double t_us, last = -1;
  char tid[64], tag[64];
  long long a, b, count = 0;
  while (fscanf(f, "%lf %63s %63s %lld %lld", &t_us, tid, tag, &a, &b) == 5) {
    CHECK(!strcmp(tag, a % 2 ? "odd" : "even"));
    CHECK(b == next[a]);
    next[a] = b + 1;
    CHECK(t_us >= 0);
    (void)last;
    count++;
  }
  fclose(f);
  CHECK(count == (long long)T * N);
  for (int t = 0; t < T; t++) CHECK(next[t] == N);
  // a fresh trace forgets the old one
  cap::trace_enable(true);
  cap::trace("fresh", 7, 7);
  cap::trace_enable(false);
  CHECK(cap::trace_dump(path) == 1);
  CHECK(cap::trace_dump("/nonexistent-dir/x") == -1);
  printf("OK\n");
  return 0;
}
