// Gathering of concurrent single-proof calls into device batches (capgpu_plonk_set_coalescing): the THREADING logic,
// free of HIP so that it also builds for the host alone - tests/cpp/coalescer_tsan.cpp runs it under ThreadSanitizer with
// stub contexts and a stub prover (round-4 VERDICT: "the threading logic has no test that runs without a GPU").
//
// The reference proves notes under rayon (`into_par_iter()`, src/utils/params_builder.rs:194-226): many host threads each
// calling prove() for ONE note.  Behind one device those calls would run one after the other at single-proof latency.
// Here the calls of one GROUP (proving keys of one domain size under one SRS: a device batch can mix them) that arrive
// while the device is busy - or within a short window - are gathered and proved as ONE batch; every caller gets its own
// result.  Protocol, all under `mu`:
//   * a caller appends its request to the group's queue and waits on `cv`;
//   * when the group has no leader, a queued caller becomes the leader: it waits one window (restarted while requests
//     keep arriving, 16 windows at most), takes a free context (Hooks::acquire: a context nobody holds, LOCKED), takes
//     up to max_batch requests off the queue, gives up the leadership and runs the batch outside `mu`;
//   * while max_in_flight batch parts are already running (round 6: two by default) the leader keeps COLLECTING instead
//     of taking a third free context: closed-loop callers (a rayon loop over notes) otherwise fragment into a dozen
//     small batches on every context of the device - 64 callers ran as 48 batches of 11 proofs on four contexts, each
//     paying its own latency-bound launches (profiles/phase_trace_r06.md) - where two or three large ones fill the chip;
//     a part that ends wakes the leader, a full queue or 16 windows without progress end the wait as well;
//   * a batch large enough is cut in two UNEVEN parts when a second context is free (Hooks::acquire_second) and nothing
//     else is in flight: the second part runs on a helper thread; each part's callers are released when THEIR part is
//     done (`early`);
//   * a request is never touched again once `done` is set: its caller returns and the request - a stack object - dies.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

#include "trace.hpp"

namespace cap {

// Req: any type with a `bool done` member.
// Hooks (all called WITHOUT `mu` held, except acquire, which is called with it held and must not block):
//   void* acquire();                      a context this thread may run a batch on, locked for it; nullptr: none free now
//   void* acquire_second();               a second free context for the other part of a cut batch (NOT left locked: the
//                                         helper thread locks it itself through run), or nullptr (also: cutting not allowed)
//   void run(void* ctx, std::vector<Req*>& reqs, bool on_helper);   prove the batch on ctx, filling every request's result
//   void release(void* ctx);              unlock what acquire locked
//   size_t deal_min();                    fewest requests a part of a cut batch holds
//   size_t split_eighths();               size of the first part of a cut batch, in eighths (3: 3/8 : 5/8)
//   bool early_release();                 release a part's callers when that part is done (else when both are)
//   size_t max_in_flight();               batch parts running at once before a leader keeps collecting instead (>= 1)
template <class Req>
struct CoalescerCore {
  std::mutex mu;
  std::condition_variable cv;
  std::map<uint64_t, std::vector<Req*>> pending;  // per group
  std::map<uint64_t, bool> leader;
  uint32_t window_us = 0;  // 0 = off
  uint32_t max_batch = 256;
  std::atomic<uint64_t> batches{0}, proofs{0};
  uint32_t in_flight = 0;  // batch parts running right now (under mu)
  // callers that have entered the library but are not queued yet (they are copying their witness to the device,
  // plonk.hip: StagePool): a leader's window does not close while some are on their way
  std::atomic<uint32_t> arriving{0};

  // cv.wait_for, except under ThreadSanitizer: libstdc++ waits on the steady clock through pthread_cond_clockwait, which
  // gcc's libtsan (<= 11) does not intercept - it misses the unlock inside the wait and reports a "double lock" at the
  // next lock().  The system-clock form goes through pthread_cond_timedwait, which it knows.
  template <class Dur, class Pred>
  bool timed_wait(std::unique_lock<std::mutex>& lk, Dur d, Pred pred) {
#if defined(__SANITIZE_THREAD__)
    return cv.wait_until(lk, std::chrono::system_clock::now() + d, pred);
#else
    return cv.wait_for(lk, d, pred);
#endif
  }

  template <class Dur>
  void timed_wait(std::unique_lock<std::mutex>& lk, Dur d) {  // (no predicate: any notification ends the wait)
#if defined(__SANITIZE_THREAD__)
    cv.wait_until(lk, std::chrono::system_clock::now() + d);
#else
    cv.wait_for(lk, d);
#endif
  }

  // Appends `req` to its group and returns when it is done (req.done).  Called with `lk` (on mu) HELD; returns with it
  // held.
  template <class Hooks>
  void submit(std::unique_lock<std::mutex>& lk, Req& req, uint64_t group, Hooks& h) {
    std::vector<Req*>& q = pending[group];
    q.push_back(&req);
    trace("co_submit", (int64_t)q.size());
    if (q.size() >= max_batch) cv.notify_all();
    bool waited_window = false;
    while (!req.done) {
      const bool queued = std::find(q.begin(), q.end(), &req) != q.end();
      if (leader[group] || !queued) {  // somebody is gathering / proving a batch that holds (or will hold) this request
        timed_wait(lk, std::chrono::milliseconds(1), [&] { return req.done; });
        continue;
      }
      // this thread leads the group's next batch: collect for the window, and for as long as every context is busy
      leader[group] = true;
      trace("co_lead", (int64_t)q.size());
      if (!waited_window) {
        // the window restarts while calls keep arriving (threads released by the previous batch come back one by one),
        // up to 16 windows in all
        const auto cap = std::chrono::steady_clock::now() + std::chrono::microseconds(16ull * window_us);
        for (size_t seen = q.size();; seen = q.size()) {
          const bool full = timed_wait(lk, std::chrono::microseconds(window_us), [&] { return q.size() >= max_batch; });
          if (full || (q.size() == seen && arriving.load() == 0) || std::chrono::steady_clock::now() >= cap) break;
        }
        waited_window = true;
        trace("co_window_end", (int64_t)q.size());
      }
      // enough parts in flight to keep the device busy: keep collecting until one of them ends (its release notifies
      // cv), the queue is full, or 64 windows have passed (a part that never ends must not hold the queue for ever)
      {
        const size_t limit = std::max<size_t>(h.max_in_flight(), 1);
        const auto cap = std::chrono::steady_clock::now() + std::chrono::microseconds(64ull * std::max(window_us, 100u));
        while (in_flight >= limit && q.size() < max_batch && std::chrono::steady_clock::now() < cap)
          timed_wait(lk, std::chrono::microseconds(std::max(window_us, 100u)));
      }
      // a free context (several batches are then in flight, one per context); later arrivals join the queue meanwhile
      void* c = nullptr;
      for (;;) {
        c = h.acquire();
        if (c) break;
        timed_wait(lk, std::chrono::microseconds(100));
      }
      trace("co_acquired", (int64_t)q.size(), (int64_t)in_flight);
      const size_t take = std::min<size_t>(q.size(), max_batch);
      std::vector<Req*> reqs(q.begin(), q.begin() + take);
      q.erase(q.begin(), q.begin() + take);
      leader[group] = false;
      const bool alone = in_flight == 0;  // nothing else is running: a large batch may be cut over two contexts
      in_flight++;
      lk.unlock();
      // a second free context takes part of a batch large enough to cut: the parts overlap on the device, or run on two
      // devices.  The cut is UNEVEN (3/8 : 5/8 by default): callers that come straight back for their next proof (a rayon
      // loop over notes) would otherwise return together, queue together and leave the device idle while every next batch
      // is gathered and copied.  Two parts of different size end at different times; from then on one batch is in flight
      // while the other is being gathered.
      std::vector<Req*> second;
      void* c2 = nullptr;
      if (alone && reqs.size() >= 2 * h.deal_min() && (c2 = h.acquire_second()) != nullptr) {
        const size_t first = std::max<size_t>(h.deal_min(), reqs.size() * h.split_eighths() / 8);
        second.assign(reqs.begin() + first, reqs.end());
        reqs.resize(first);
      }
      const bool early = h.early_release();
      trace("co_run", (int64_t)reqs.size(), (int64_t)second.size());
      std::thread helper;
      if (c2) {
        {
          std::lock_guard<std::mutex> g(mu);
          in_flight++;
        }
        helper = std::thread([this, &second, c2, &h, early] {
          h.run(c2, second, true);
          trace("co_run2_end", (int64_t)second.size());
          std::lock_guard<std::mutex> g(mu);
          in_flight--;
          if (early)
            for (Req* r : second) r->done = true;  // (a request is not touched again once it is marked: its caller returns)
          cv.notify_all();
        });
      }
      h.run(c, reqs, false);  // (run counts what it proved in `batches` / `proofs`)
      h.release(c);
      trace("co_run_end", (int64_t)reqs.size());
      lk.lock();
      in_flight--;
      if (early)
        for (Req* r : reqs) r->done = true;
      cv.notify_all();
      lk.unlock();
      if (helper.joinable()) helper.join();
      lk.lock();
      if (!early) {
        for (Req* r : reqs) r->done = true;
        for (Req* r : second) r->done = true;
        cv.notify_all();
      }
    }
    trace("co_return");
  }
};

}  // namespace cap
