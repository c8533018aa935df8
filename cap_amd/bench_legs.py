"""The measurement legs of bench.py that are functions of the library alone (no process-group state): single and batched
MSM legs (BASELINE configs 2 and 5), the NTT leg, SURVEY 8(d)'s algorithmic byte counts and the ordering of the JSON line.
Moved out of bench.py in round 6 (round-5 VERDICT item 7: "bench.py is 1453 lines"); bench.py remains the entry point and
holds the timed region, the process-group logic and the legs that share its state."""
from __future__ import annotations

import time

import numpy as np

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MADS_PER_MUL = 171       # lazy 9 x 29-bit Montgomery multiplication: 81 product + 81 reduction + 9 digit multiply-adds
# VALU wave-instructions msm_accumulate executes per mixed addition: SQ_INSTS_VALU of the kernel / (additions / 64),
# profiles/inst_counters_r02.json (static PMC pass; the loop body's ISA counts 2190, of which 1550 are multiply-adds)
INSTR_PER_MIXED_ADD = 2160

def algorithmic_bytes_per_proof(n: int) -> dict:
    """reference schedule, primitives only."""
    msm_pairs = 5 * (n + 2) + (n + 3) + 5 * (n + 2) + 2 * (n + 2)
    ntt_elems = 7 * n + 26 * 8 * n
    return {"msm_pairs": msm_pairs, "msm_bytes": 96 * msm_pairs, "ntt_elems": ntt_elems, "ntt_bytes": 64 * ntt_elems,
            "total_bytes": 96 * msm_pairs + 64 * ntt_elems}


P_FQ = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
A_SEQ, B_SEQ = 0x1234567890ABCDEF1234567890ABCDEF, 0xFEDCBA0987654321FEDCBA


def _words_to_ints(words):
    w = np.asarray(words, dtype=np.uint64).reshape(-1, 4)
    return [int(r[0]) | int(r[1]) << 64 | int(r[2]) << 128 | int(r[3]) << 192 for r in w]


def _pcts(samples_ms):
    """median / p10 / p90 / min / max of per-iteration times (>= 10 warm-up + >= 50 timed iterations)"""
    xs = sorted(samples_ms)
    q = lambda f: xs[min(len(xs) - 1, int(f * len(xs)))]       # noqa: E731
    return {"median": q(0.5), "p10": q(0.10), "p90": q(0.90), "min": xs[0], "max": xs[-1], "iterations": len(xs)}


WARMUP_ITERS, TIMED_ITERS = 10, 50


def ntt_leg(cg, bu, log_n=17, iters=TIMED_ITERS, warmup=WARMUP_ITERS):
    """The north star's NTT size on one GPU (ark-poly's in-place radix-2 transforms, natural order in and out):
    one device-resident 2^log_n transform at a time (latency: what a caller of capgpu_ntt_fr_dev sees) and 64 of them in
    one call (throughput); forward then inverse must give the input back.  64 B per element per transform.
    Every iteration (one forward call, then one inverse call) is bracketed by HIP events on the library's stream
    (capgpu_timer_begin / _end): device time per call = pair / 2; median, p10, p90 over `iters` iterations after `warmup`."""
    import numpy as np
    n = 1 << log_n
    out = []
    for count in (1, 64):
        a = bu.random_canonical_scalars(1234 + count, count * n)   # canonical residues: the round trip returns them
        d = cg.DevBuf.from_numpy(a)
        for _ in range(warmup):
            cg.ntt_fr_dev(d, log_n, count=count)
            cg.ntt_fr_dev(d, log_n, count=count, inverse=True)
        cg.sync()
        dev_ms = []
        t0 = time.perf_counter()
        for _ in range(iters):
            cg.timer_begin()
            cg.ntt_fr_dev(d, log_n, count=count)
            cg.ntt_fr_dev(d, log_n, count=count, inverse=True)
            dev_ms.append(cg.timer_end() / 2)
        wall_ms = (time.perf_counter() - t0) * 1e3 / (2 * iters)
        back = d.to_numpy().reshape(a.shape)
        d.free()
        st = _pcts(dev_ms)
        ms = st["median"]
        out.append({"log_n": log_n, "transforms_per_call": count, "ms_per_call": ms, "device_ms_per_call": st,
                    "host_wall_ms_per_call": wall_ms, "warmup_iterations": warmup,
                    "timing": "HIP events on the library stream around a forward + inverse pair, / 2; median",
                    "GBps_algorithmic": 64.0 * n * count / ms / 1e6, "frac_of_hbm_peak": 64.0 * n * count / ms / 1e6 / 8000.0,
                    "round_trip_identity": bool(np.array_equal(back, a))})
    return out


def msm_leg(cg, bu, torch, dist, rank, world, log_n, iters=TIMED_ITERS, warmup=WARMUP_ITERS, coll_dev="cuda",
            use_lib_comm=False):
    """Point-range-sharded MSM: bases P_i = [a + i b]G generated on each rank's GPU for its range,
    scalars resident, local Pippenger, ONE exchange step (all-gather of a 96-byte Jacobian point per rank) and
    G-1 group additions.  With use_lib_comm the exchange is the library's own (RCCL all-gather on its stream from
    device memory + sum on the device: capgpu_msm_g1_sharded_dev); otherwise (gloo test runs) it hops through
    torch.distributed.  Checked against [sum k_i (a + i b)] G.
    Timing: `warmup` untimed calls, then `iters` calls each bracketed by HIP events on the library's stream
    (the whole sharded MSM - local Pippenger, exchange, sum - is enqueued there); `ms` is the MEDIAN device time, p10 / p90
    beside it; the gloo test path, whose exchange leaves the stream, is timed per call on the host clock instead.  With
    N > 1 the reported figure is the max over ranks of each rank's median."""
    from cap_amd import parallel as par
    n_total = 1 << log_n
    lo, hi = par.shard_range(n_total, rank, world)
    n = hi - lo
    a, b = A_SEQ % bu.R, B_SEQ % bu.R
    srs = cg.srs_generate_affine_seq((a + lo * b) % bu.R, b, n)
    plan = cg.msm_plan(srs, n, 1)
    sc = bu.random_canonical_scalars(5, n_total)
    d_sc = cg.DevBuf.from_numpy(np.ascontiguousarray(sc[lo:hi]))
    d_out = cg.DevBuf(96)
    on_stream = dist is None or use_lib_comm

    def one():
        if dist is None:
            cg.msm_g1_dev(srs, d_sc, n, d_out=d_out)
            return None
        if use_lib_comm:
            cg.msm_g1_sharded_dev(srs, d_sc, n, d_out=d_out)       # local MSM + all-gather + sum: all enqueued
            return None
        cg.msm_g1_dev(srs, d_sc, n, d_out=d_out)
        return cg.g1_sum(par.all_gather_points(d_out.to_numpy(), device=coll_dev))

    for _ in range(warmup):
        one()
    cg.sync()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    samples, total = [], None
    t0 = time.perf_counter()
    for _ in range(iters):
        if on_stream:
            cg.timer_begin()
            total = one()
            samples.append(cg.timer_end())
        else:
            ts = time.perf_counter()
            total = one()
            cg.sync()
            samples.append((time.perf_counter() - ts) * 1e3)
    cg.sync()
    torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t0) / iters * 1e3
    st = _pcts(samples)
    dt = st["median"] * 1e-3
    if total is None:
        total = d_out.to_numpy()
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ok = None
    if rank == 0:
        s0, s1 = bu.weighted_scalar_sums(sc, 0)
        h1 = cg.srs_generate_affine_seq((a * s0 + b * s1) % bu.R, 0, 1)
        rinv = pow(1 << 256, -1, P_FQ)
        ex, ey = [v * rinv % P_FQ for v in _words_to_ints(cg.srs_download(h1, 0, 1))]
        cg.srs_free(h1)
        X, Y, Z = [v * rinv % P_FQ for v in _words_to_ints(total)]
        if Z == 0:
            ok = False
        else:
            zi = pow(Z, -1, P_FQ)
            ok = bool((X * zi * zi % P_FQ, Y * zi * zi * zi % P_FQ) == (ex, ey))
    cg.srs_free(srs)
    gbps = 96.0 * n_total / dt / 1e9
    if world == 1:
        how = "single GPU"
    elif use_lib_comm:
        how = f"point range x{world}: local Pippenger + ncclAllGather(96 B) on the library stream + on-device sum (capgpu_msm_g1_sharded_dev)"
    else:
        how = f"point range x{world} + torch.distributed all_gather(96 B) + capgpu_g1_sum (gloo test path)"
    return {"log_n": log_n, "points": n_total, "ms": dt * 1e3, "device_ms_this_rank": st, "host_wall_ms_per_call": wall_ms,
            "warmup_iterations": warmup,
            "timing": ("HIP events on the library stream around each call; median" if on_stream
                       else "host clock around each call + sync (gloo test path); median") +
                      ("; max over ranks of the rank medians" if world > 1 else ""),
            "GBps_algorithmic": gbps, "frac_of_hbm_peak": gbps / HBM_PEAK_GBPS,
            "sharding": how, "plan_per_rank": plan, "identity_check": ok}


def msm_batch_leg(cg, bu, log_n=17, counts=(5, 64), iters=TIMED_ITERS, warmup=WARMUP_ITERS):
    """BASELINE config 2's size as a THROUGHPUT figure (round-5 VERDICT item 5): `count` independent MSMs of 2^log_n points
    on one SRS in ONE call (capgpu_msm_g1_dev with count > 1 - what a round of an n = 2^17 circuit's prover commits: five
    wire or five quotient polynomials; 64 = a batch of proofs' worth).  Bases P_i = [a + i b]G, every MSM its own uniform
    scalars, resident; each call bracketed by HIP events on the library's stream; median, p10 / p90; 96 B per (point,
    scalar) pair; every result checked against [sum k_i (a + i b)]G."""
    n = 1 << log_n
    a, b = A_SEQ % bu.R, B_SEQ % bu.R
    srs = cg.srs_generate_affine_seq(a, b, n)
    rinv = pow(1 << 256, -1, P_FQ)
    out = []
    for count in counts:
        sc = bu.random_canonical_scalars(500 + count, count * n).reshape(count, n, 4)
        d_sc = cg.DevBuf.from_numpy(sc)
        d_out = cg.DevBuf(96 * count)
        for _ in range(warmup):
            cg.msm_g1_dev(srs, d_sc, n, count=count, d_out=d_out)
        cg.sync()
        samples = []
        t0 = time.perf_counter()
        for _ in range(iters):
            cg.timer_begin()
            cg.msm_g1_dev(srs, d_sc, n, count=count, d_out=d_out)
            samples.append(cg.timer_end())
        cg.sync()
        wall_ms = (time.perf_counter() - t0) / iters * 1e3
        res = d_out.to_numpy().reshape(count, 12)
        ok = True
        for q in range(count):
            s0, s1 = bu.weighted_scalar_sums(sc[q], 0)
            h1 = cg.srs_generate_affine_seq((a * s0 + b * s1) % bu.R, 0, 1)
            ex, ey = [v * rinv % P_FQ for v in _words_to_ints(cg.srs_download(h1, 0, 1))]
            cg.srs_free(h1)
            X, Y, Z = [v * rinv % P_FQ for v in _words_to_ints(res[q])]
            zi = pow(Z, -1, P_FQ) if Z else 0
            ok = ok and bool(Z and (X * zi * zi % P_FQ, Y * zi * zi * zi % P_FQ) == (ex, ey))
        st = _pcts(samples)
        gbps = 96.0 * n * count / st["median"] / 1e6
        out.append({"log_n": log_n, "points": n, "msms_per_call": count, "ms": st["median"], "device_ms_per_call": st,
                    "ms_per_msm": st["median"] / count, "host_wall_ms_per_call": wall_ms, "warmup_iterations": warmup,
                    "timing": "HIP events on the library stream around each call; median",
                    "GBps_algorithmic": gbps, "frac_of_hbm_peak": gbps / HBM_PEAK_GBPS,
                    "plan": cg.msm_plan(srs, n, count), "identity_check": ok})
        d_sc.free()
        d_out.free()
    cg.srs_free(srs)
    return out


def headline_last(out: dict) -> dict:
    """Order the keys of the line so that what the contract names - metric, value, roofline, cpu_baseline, the workload - and a
    compact `summary` of the secondary legs come LAST: a log that keeps only the tail of the line (the driver's keeps 2000
    characters) still shows them.  Long explanatory strings of `roofline` / `cpu_baseline` move to `*_notes` further up;
    nothing is dropped from the line."""
    o = dict(out)
    cfg = o.get("config", {})
    legs = cfg.get("legs", {})
    val = o.get("value") or 0.0

    def rate(key, field="proofs_per_s"):
        v = o.get(key)
        return v.get(field) if isinstance(v, dict) else None

    def r3(x):
        return None if x is None else float(f"{x:.4g}")

    summary = {"workload": cfg.get("workload")}
    for name, key in (("pcie_inclusive", "pcie_inclusive"), ("pcie_inclusive_coeffs", "pcie_inclusive_coeffs"),
                      ("coalesced_single_calls", "coalesced_single_calls"), ("reference_schedule", "reference_schedule"),
                      ("realistic_witness", "realistic_witness"), ("n2p16", "n2p16"), ("n1_same_run", "n1_same_run")):
        v = rate(key)
        if v is not None:
            summary[name + "_proofs_per_s"] = r3(v)
            if name in ("pcie_inclusive", "coalesced_single_calls") and val:
                summary[name + "_over_value"] = r3(v / val)
                if o[key].get("over_resident_same_minute"):
                    summary[name + "_over_resident_same_minute"] = r3(o[key]["over_resident_same_minute"])
                if o[key].get("callers_128_over_resident_same_minute"):
                    summary[name + "_128_callers_over_resident_same_minute"] = r3(o[key]["callers_128_over_resident_same_minute"])
    if isinstance(o.get("n1_same_run"), dict):
        summary["value_over_n_times_n1_same_run"] = r3(o["n1_same_run"].get("value_over_n_times_this"))
    m64 = o.get("mixed64") or o.get("mixed64_multi_gpu")
    if isinstance(m64, dict):
        summary["mixed64_proofs_per_s"] = {k.replace("_proofs_per_s", ""): r3(v) for k, v in m64.items()
                                           if k.endswith("_proofs_per_s") and isinstance(v, (int, float))}
    if legs:
        summary["legs_median_ms"] = {k[:-3] if k.endswith("_ms") else k: r3(v.get("median")) for k, v in legs.items()}
        summary["msm_GBps"] = {k[:-3]: r3(v.get("GBps_algorithmic")) for k, v in legs.items() if k.startswith("msm_")}
    for k in ("device_memory_in_use_GB", "device_memory_after_trim_GB"):
        if k in cfg:
            summary[k] = cfg[k]
    for key, long_fields in (("roofline", ("traffic_source", "measured_in")), ("cpu_baseline", ("clock", "timed_on", "cpu_model"))):
        v = o.get(key)
        if isinstance(v, dict):
            v = dict(v)
            notes = {f: v.pop(f) for f in long_fields if f in v}
            if notes:
                o[key + "_notes"] = notes
            o[key] = v
    tail = ["config", "summary", "roofline", "cpu_baseline", "metric", "value", "unit", "n_gpus", "steps", "warmup",
            "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"]
    o["summary"] = summary
    return {**{k: v for k, v in o.items() if k not in tail}, **{k: o[k] for k in tail if k in o}}
