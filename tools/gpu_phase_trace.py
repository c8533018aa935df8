#!/usr/bin/env python3
"""Where the wall time of the reference's two calling patterns goes (round-5 VERDICT items 2 and 3), from the library's
host-side phase trace (capgpu_trace_enable / _dump, cap_amd/csrc/trace.hpp):

  host     capgpu_plonk_prove_batch on 256 HOST-resident witnesses (bench.py's `pcie_inclusive`): per part of the dealt
           batch - when its copy turn came, when each chunk had landed, when each round's results were back
  coalesce 64 closed-loop callers of capgpu_plonk_prove_ex (bench.py's `coalesced_single_calls`): batch sizes, window
           and context waits, how many batches are in flight over time, caller latency
  resident the same batch from device-resident witnesses on two contexts (the headline's timed region), for the ratio

One JSON line per mode on stdout; raw traces under gpurun_out/.  Environment knobs (CAPGPU_H2D_PART_ORDER,
CAPGPU_PROVE_CHUNKS, CAPGPU_COALESCE_SPLIT, ...) are read once per process: run one process per configuration.
    python tools/gpu_phase_trace.py host|coalesce|resident [TAG] [--batch 256] [--reps 4] [--threads 64] [--calls 8]
"""
import argparse
import collections
import ctypes
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def read_trace(path):
    ev = []
    with open(path) as f:
        for ln in f:
            t, tid, tag, a, b = ln.split()
            ev.append((float(t), tid, tag, int(a), int(b)))
    return ev


def pb_calls(ev):
    """prove_batch calls: slot -> list of dicts {begin, end, P, phases{tag: t}, chunks[(issue, issued)]}"""
    open_, calls = {}, []
    for t, tid, tag, a, b in ev:
        if tag == "pb_begin":
            open_[(tid, a)] = {"slot": a, "P": b, "begin": t, "phases": {}, "chunks": [], "tid": tid}
        elif tag.startswith("pb_"):
            cur = open_.get((tid, a))
            if cur is None:
                continue
            if tag == "pb_end":
                cur["end"] = t
                calls.append(open_.pop((tid, a)))
            elif tag == "pb_h2d_issue":
                cur["chunks"].append([t, None])
            elif tag == "pb_h2d_issued":
                cur["chunks"][-1][1] = t
            else:
                cur["phases"][tag] = t
    return calls


def in_flight_profile(intervals, t0, t1):
    """share of [t0, t1] during which k intervals are open, k = 0, 1, 2, 3+"""
    pts = []
    for a, b in intervals:
        pts.append((max(a, t0), 1))
        pts.append((min(b, t1), -1))
    pts.sort()
    share = collections.Counter()
    k, last = 0, t0
    for t, d in pts:
        if t > last:
            share[min(k, 3)] += t - last
            last = t
        k += d
    if t1 > last:
        share[min(k, 3)] += t1 - last
    tot = max(t1 - t0, 1e-9)
    return {str(i) + ("+" if i == 3 else ""): round(share[i] / tot, 4) for i in range(4)}


def setup(cg, bu, np, P):
    log_n, ni = 15, 27
    n = 1 << log_n
    tau = bu.SplitMix64(0xCA9).field()
    srs = cg.srs_generate(tau, n + 3)
    sc = bu.synthetic_circuit(log_n, ni, seed=2 + log_n + ni)
    pk, _ = cg.plonk_preprocess(srs, n, ni, sc.selectors_mont(), sc.sigma_mont())
    wires, pubs = sc.witnesses_mont([3 + i for i in range(P)])
    blind = np.stack([bu.to_mont_array(bu.blinders(7000 + i)) for i in range(P)])
    return pk, wires, pubs, blind, bytes(range(32)), n, ni


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["host", "host2", "coalesce", "resident"])
    ap.add_argument("--pin", action="store_true", help="host / host2: the witnesses in page-locked memory (hipHostRegister)")
    ap.add_argument("tag", nargs="?", default="run")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--threads", type=int, default=64)
    ap.add_argument("--calls", type=int, default=8)
    ap.add_argument("--window-us", type=int, default=500)
    ap.add_argument("--python-threads", action="store_true", help="callers are Python threads (rounds 3-5) instead of native ones")
    args = ap.parse_args()
    import numpy as np
    from cap_amd import bench_utils as bu
    from cap_amd import lib as cg
    cg.init(0)
    P = args.batch
    pk, wires, pubs, blind, msg, n, ni = setup(cg, bu, np, P)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    raw = os.path.join(ROOT, "gpurun_out", f"phase_trace_{args.mode}_{args.tag}.txt")
    knobs = {k: v for k, v in os.environ.items() if k.startswith("CAPGPU_")}
    out = {"mode": args.mode, "tag": args.tag, "batch": P, "knobs": knobs}

    if args.mode == "resident":
        per = 5 * n * 32
        d = cg.DevBuf.from_numpy(wires)
        halves = [(0, P // 2), (P // 2, P)]
        bar = threading.Barrier(3)

        def run(i):
            cg.set_device(i)
            lo, hi = halves[i]
            buf = d.view(lo * per, (hi - lo) * per)
            cg.plonk_prove_batch_dev(pk, buf, pubs[lo:hi], blind[lo:hi], msg, hi - lo)
            bar.wait()
            for _ in range(args.reps):
                cg.plonk_prove_batch_dev(pk, buf, pubs[lo:hi], blind[lo:hi], msg, hi - lo)
            bar.wait()

        th = [threading.Thread(target=run, args=(i,)) for i in range(2)]
        for t in th:
            t.start()
        bar.wait()
        cg.trace_enable(True)
        t0 = time.perf_counter()
        bar.wait()
        dt = time.perf_counter() - t0
        cg.trace_enable(False)
        for t in th:
            t.join()
        out["proofs_per_s"] = P * args.reps / dt
        out["ms_per_batch"] = dt / args.reps * 1e3
        cg.trace_dump(raw)

    elif args.mode == "host2":
        # two host threads, each bound to a context, each proving ITS half of the host-resident batch call after call: the
        # headline's timed region with the witnesses in host memory
        halves = [(0, P // 2), (P // 2, P)]
        bar = threading.Barrier(3)
        if args.pin:
            import torch
            rc = torch.cuda.cudart().cudaHostRegister(wires.ctypes.data, wires.nbytes, 0)
            out["host_register_rc"] = int(rc)

        def run2(i):
            cg.set_device(i)
            lo, hi = halves[i]
            cg.plonk_prove_batch(pk, wires[lo:hi], pubs[lo:hi], blind[lo:hi], msg, hi - lo)
            bar.wait()
            for _ in range(args.reps):
                cg.plonk_prove_batch(pk, wires[lo:hi], pubs[lo:hi], blind[lo:hi], msg, hi - lo)
            bar.wait()

        th = [threading.Thread(target=run2, args=(i,)) for i in range(2)]
        for t in th:
            t.start()
        bar.wait()
        cg.trace_enable(True)
        t0 = time.perf_counter()
        bar.wait()
        dt = time.perf_counter() - t0
        cg.trace_enable(False)
        for t in th:
            t.join()
        out["proofs_per_s"] = P * args.reps / dt
        out["ms_per_batch"] = dt / args.reps * 1e3
        cg.trace_dump(raw)

    elif args.mode == "host":
        if args.pin:
            import torch
            rc = torch.cuda.cudart().cudaHostRegister(wires.ctypes.data, wires.nbytes, 0)
            out["host_register_rc"] = int(rc)
        cg.set_device(-1)
        cg.plonk_prove_batch(pk, wires, pubs, blind, msg, P)       # warm-up: scratch, pinned pages of `wires`
        cg.trace_enable(True)
        t0 = time.perf_counter()
        for _ in range(args.reps):
            cg.plonk_prove_batch(pk, wires, pubs, blind, msg, P)
        dt = time.perf_counter() - t0
        cg.trace_enable(False)
        cg.trace_dump(raw)
        out["proofs_per_s"] = P * args.reps / dt
        out["ms_per_batch"] = dt / args.reps * 1e3
        calls = sorted(pb_calls(read_trace(raw)), key=lambda c_: c_["begin"])
        # group the parts of one dealt batch: calls that overlap in time
        groups, cur = [], []
        for c_ in calls:
            if cur and c_["begin"] > max(x["end"] for x in cur):
                groups.append(cur)
                cur = []
            cur.append(c_)
        if cur:
            groups.append(cur)
        rows = []
        for g in groups:
            g0 = min(x["begin"] for x in g)
            row = {"wall_ms": round((max(x["end"] for x in g) - g0) / 1e3, 2), "parts": []}
            for x in sorted(g, key=lambda x: x["slot"]):
                ph = x["phases"]
                row["parts"].append({
                    "slot": x["slot"], "proofs": x["P"],
                    "turn_ms": round((ph.get("pb_h2d_turn", x["begin"]) - g0) / 1e3, 2),
                    "chunks_landed_ms": [round((ck[1] - g0) / 1e3, 2) for ck in x["chunks"] if ck[1]],
                    **{k[3:] + "_ms": round((v - g0) / 1e3, 2) for k, v in ph.items() if k.endswith("_done")},
                    "end_ms": round((x["end"] - g0) / 1e3, 2)})
            rows.append(row)
        out["batches"] = rows

    else:
        T, per_thread = args.threads, args.calls
        cg.plonk_set_coalescing(args.window_us, 256)
        L = cg.load()
        mbuf = (ctypes.c_uint8 * len(msg)).from_buffer_copy(msg)
        u64p = ctypes.POINTER(ctypes.c_uint64)
        proofs_c = [[cg.Proof() for _ in range(per_thread)] for _ in range(T)]

        def _args(i, pr):
            return (ctypes.c_uint64(pk), wires[i].ctypes.data_as(u64p), pubs[i].ctypes.data_as(u64p),
                    ctypes.c_size_t(ni), mbuf, ctypes.c_size_t(len(msg)), blind[i].ctypes.data_as(u64p),
                    ctypes.c_int(0), ctypes.byref(pr))

        idx = [(t * per_thread + k) % P for t in range(T) for k in range(per_thread)]
        flat = [proofs_c[t][k] for t in range(T) for k in range(per_thread)]
        if args.python_threads:
            calls = [[_args((t * per_thread + k) % P, proofs_c[t][k]) for k in range(per_thread)] for t in range(T)]
        for rep in range(2):               # the first repetition warms scratch and the callers' pages up
            errs = []
            if rep == 1:
                cg.trace_enable(True)
            if not args.python_threads:    # native callers (cap_amd/csrc/witgen.c): what rayon workers are
                failed, dt = bu.closed_loop_callers(L.capgpu_plonk_prove_ex, pk, [wires[i] for i in idx], [pubs[i] for i in idx],
                                                    ni, msg, [blind[i] for i in idx], flat, T, per_thread)
                errs = [failed] if failed else []
                continue
            bar = threading.Barrier(T + 1)

            def _worker(t):
                bar.wait()
                for a in calls[t]:
                    rc = L.capgpu_plonk_prove_ex(*a)
                    if rc:
                        errs.append(rc)
                        return

            ths = [threading.Thread(target=_worker, args=(t,)) for t in range(T)]
            for th in ths:
                th.start()
            bar.wait()
            t0 = time.perf_counter()
            for th in ths:
                th.join()
            dt = time.perf_counter() - t0
        cg.trace_enable(False)
        cg.plonk_set_coalescing(0)
        cg.trace_dump(raw)
        out.update({"proofs_per_s": T * per_thread / dt, "threads": T, "callers": "python" if args.python_threads else "native", "calls_per_thread": per_thread, "errors": errs[:3],
                    "wall_ms": dt * 1e3})
        ev = read_trace(raw)
        calls_pb = pb_calls(ev)
        t_first = min(c_["begin"] for c_ in calls_pb)
        t_last = max(c_["end"] for c_ in calls_pb)
        sizes = [c_["P"] for c_ in calls_pb]
        out["device_batches"] = len(sizes)
        out["batch_size"] = {"mean": round(sum(sizes) / len(sizes), 1), "min": min(sizes), "max": max(sizes),
                             "histogram": dict(sorted(collections.Counter((s // 8) * 8 for s in sizes).items()))}
        out["batches_in_flight_share_of_wall"] = in_flight_profile([(c_["begin"], c_["end"]) for c_ in calls_pb], t_first, t_last)
        # per batch: the H2D inside it (begin -> last chunk landed), round 1 .. 5, and its whole length
        def med(xs):
            xs = sorted(xs)
            return round(xs[len(xs) // 2] / 1e3, 3) if xs else None
        out["per_batch_ms_median"] = {
            "h2d_until_last_chunk": med([c_["chunks"][-1][1] - c_["begin"] for c_ in calls_pb if c_["chunks"] and c_["chunks"][-1][1]]),
            "round1_done": med([c_["phases"]["pb_r1_done"] - c_["begin"] for c_ in calls_pb if "pb_r1_done" in c_["phases"]]),
            "whole_batch": med([c_["end"] - c_["begin"] for c_ in calls_pb])}
        # the leader's path: lead -> window_end -> acquired -> run
        lead, win, acq, lat = {}, [], [], []
        sub = {}
        for t, tid, tag, a, b in ev:
            if tag == "co_submit":
                sub[tid] = t
            elif tag == "co_return" and tid in sub:
                lat.append(t - sub.pop(tid))
            elif tag == "co_lead":
                lead[tid] = t
            elif tag == "co_window_end" and tid in lead:
                win.append(t - lead[tid])
                lead[tid] = t
            elif tag == "co_acquired" and tid in lead:
                acq.append(t - lead.pop(tid))
        out["leader_ms_median"] = {"window": med(win), "wait_for_a_free_context": med(acq)}
        out["leader_ms_mean"] = {"window": round(sum(win) / max(len(win), 1) / 1e3, 3),
                                 "wait_for_a_free_context": round(sum(acq) / max(len(acq), 1) / 1e3, 3)}
        out["caller_latency_ms"] = {"median": med(lat), "mean": round(sum(lat) / max(len(lat), 1) / 1e3, 2)}
    free_b, total_b = cg.mem_info()
    out["device_memory_in_use_GB"] = round((total_b - free_b) / 1e9, 1)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
