// Probe: host-to-device rate of a caller's PAGEABLE wire buffers (what capgpu_plonk_prove_batch is handed) - the runtime's own
// staged copy against the library staging them itself: T host threads memcpy pieces into a pinned buffer, one DMA per
// piece.  tools/probe_h2d.bin [MB per chunk] ; prints GB/s.  (round 5: is pcie_inclusive's -8 % the copy path?)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const size_t chunk = (size_t)(argc > 1 ? atoi(argv[1]) : 168) << 20;
  const int chunks = 8;
  char* host = (char*)malloc(chunk * chunks);
  for (size_t i = 0; i < chunk * chunks; i += 4096) host[i] = (char)i;  // touch every page
  char* dev = nullptr;
  hipMalloc(&dev, chunk * chunks);
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  // (a) the runtime's pageable path
  for (int rep = 0; rep < 2; rep++) {
    double t0 = now();
    for (int c = 0; c < chunks; c++) hipMemcpyAsync(dev + c * chunk, host + c * chunk, chunk, hipMemcpyHostToDevice, s);
    hipStreamSynchronize(s);
    double dt = now() - t0;
    printf("pageable hipMemcpyAsync: %.1f GB/s (%.1f ms per %zu MB chunk)\n", chunk * chunks / dt / 1e9, dt / chunks * 1e3, chunk >> 20);
  }
  // (b) own staging: T threads copy pieces into pinned slots, DMA from there
  const size_t piece = 8 << 20;
  for (int T : {1, 2, 4, 8, 16}) {
    const int slots = 2 * T;
    char* pin = nullptr;
    hipHostMalloc((void**)&pin, piece * slots, hipHostMallocDefault);
    std::vector<hipEvent_t> ev(slots);
    for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    std::vector<hipStream_t> st(T);
    for (auto& x : st) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
    const size_t total = chunk * chunks, pieces = total / piece;
    double t0 = now();
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
      th.emplace_back([&, t] {
        hipSetDevice(0);
        int k = 0;
        for (size_t p = t; p < pieces; p += T, k ^= 1) {
          const int slot = 2 * t + k;
          hipEventSynchronize(ev[slot]);  // the slot's previous DMA is done (a fresh event is complete)
          memcpy(pin + (size_t)slot * piece, host + p * piece, piece);
          hipMemcpyAsync(dev + p * piece, pin + (size_t)slot * piece, piece, hipMemcpyHostToDevice, st[t]);
          hipEventRecord(ev[slot], st[t]);
        }
        hipStreamSynchronize(st[t]);
      });
    for (auto& x : th) x.join();
    double dt = now() - t0;
    printf("own staging, %2d threads x 8 MB pieces: %.1f GB/s (%.1f ms per %zu MB chunk)\n", T, total / dt / 1e9, dt / chunks * 1e3, chunk >> 20);
    hipHostFree(pin);
  }
  // (c) everything pinned (the ceiling)
  char* pin_all = nullptr;
  hipHostMalloc((void**)&pin_all, chunk, hipHostMallocDefault);
  memcpy(pin_all, host, chunk);
  double t0 = now();
  for (int c = 0; c < chunks; c++) hipMemcpyAsync(dev + c * chunk, pin_all, chunk, hipMemcpyHostToDevice, s);
  hipStreamSynchronize(s);
  double dt = now() - t0;
  printf("pinned source (ceiling): %.1f GB/s\n", chunk * chunks / dt / 1e9);
  return 0;
}
