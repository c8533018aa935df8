#!/usr/bin/env python3
"""bench.py - transfer-note proofs/sec (2-in/2-out) on N MI355X, one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input: `--batch` independent 2-in/2-out
transfer-note proofs (circuit shape n = 2^15, 27 public inputs - src/utils/mod.rs:149-153; random satisfiable
TurboPlonk instance, SRS = powers of a known tau) proved by the device prover behind the capgpu C ABI.  The
witnesses, the proving key and the SRS are resident in HBM when the timed region starts.  With N > 1 every rank
proves its own batch (proofs are independent: replicas, no data-path collective) -> weak scaling;
value = all proofs of all ranks / max-over-ranks time.

Rank 0 prints ONE JSON line; see DESIGN.md "Measurement" for the definition of every field.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
TRAFFIC_BATCH = 256      # batch size of the committed PMC pass (profiles/traffic_r01.json)


def algorithmic_bytes_per_proof(n: int) -> dict:
    """reference schedule, primitives only."""
    msm_pairs = 4 * 0 + 5 * (n + 2) + (n + 3) + 5 * (n + 2) + 2 * (n + 2)
    ntt_elems = 7 * n + 26 * 8 * n
    return {"msm_pairs": msm_pairs, "msm_bytes": 96 * msm_pairs, "ntt_elems": ntt_elems, "ntt_bytes": 64 * ntt_elems,
            "total_bytes": 96 * msm_pairs + 64 * ntt_elems}


P_FQ = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47


def _weighted_sums(sc: np.ndarray, lo: int):
    """sum k_i and sum (lo + i) k_i for canonical scalars (n, 4) uint64, exactly, with numpy on 16-bit pieces."""
    n = sc.shape[0]
    pieces = sc.view(np.uint16).reshape(n, 16).astype(np.uint64)
    idx = (np.arange(n, dtype=np.uint64) + np.uint64(lo))
    s0 = s1 = 0
    for start in range(0, n, 1 << 18):
        blk = pieces[start:start + (1 << 18)]
        col = blk.sum(axis=0)
        wcol = (blk * idx[start:start + (1 << 18), None]).sum(axis=0)
        s0 += sum(int(col[j]) << (16 * j) for j in range(16))
        s1 += sum(int(wcol[j]) << (16 * j) for j in range(16))
    return s0, s1


def msm_leg(cg, bu, torch, dist, rank, world, log_n, iters=5, coll_dev="cuda"):
    """Point-range-sharded MSM: bases P_i = [a + i b]G generated on each rank's GPU for its range,
    scalars resident, local Pippenger, ONE exchange step (all-gather of a 96-byte Jacobian point per rank) and
    G-1 group additions.  Checked against [sum k_i (a + i b)] G."""
    from cap_amd import parallel as par
    n_total = 1 << log_n
    lo, hi = par.shard_range(n_total, rank, world)
    n = hi - lo
    a, b = 0x1234567890ABCDEF1234567890ABCDEF % bu.R, 0xFEDCBA0987654321FEDCBA % bu.R
    srs = cg.srs_generate_affine_seq((a + lo * b) % bu.R, b, n)
    rng = np.random.default_rng(5)
    sc = rng.integers(0, 1 << 63, size=(n_total, 4), dtype=np.uint64)
    sc[:, 3] &= np.uint64((1 << 61) - 1)          # < 2^253 < r : canonical
    mine = np.ascontiguousarray(sc[lo:hi])
    d_sc = cg.DevBuf.from_numpy(mine)
    d_out = cg.DevBuf(96)
    cg.msm_g1_dev(srs, d_sc, n, d_out=d_out)       # warm-up
    cg.sync()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    # the timed region is the whole sharded MSM: local Pippenger, the one exchange step (all-gather of a 96-byte
    # partial per rank) and the G - 1 additions of the partials
    t0 = time.perf_counter()
    for _ in range(iters):
        cg.msm_g1_dev(srs, d_sc, n, d_out=d_out)
        part = d_out.to_numpy()                    # synchronises: the partial sum leaves the device here
        if dist is not None:
            parts = par.all_gather_points(part, device=coll_dev)
            total = cg.g1_sum(parts)
        else:
            total = part
    cg.sync()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ok = None
    if rank == 0:
        s0, s1 = _weighted_sums(sc, 0)
        expect_scalar = (a * s0 + b * s1) % bu.R
        h1 = cg.srs_generate_affine_seq(expect_scalar, 0, 1)
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
    return {"log_n": log_n, "points": n_total, "ms": dt * 1e3, "GBps_algorithmic": gbps, "frac_of_hbm_peak": gbps / HBM_PEAK_GBPS,
            "sharding": f"point range x{world} + all_gather(96 B) + g1_sum" if world > 1 else "single GPU",
            "identity_check": ok}


def _words_to_ints(words):
    w = np.asarray(words, dtype=np.uint64).reshape(-1, 4)
    return [int(r[0]) | int(r[1]) << 64 | int(r[2]) << 128 | int(r[3]) << 192 for r in w]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="proofs per step per GPU")
    ap.add_argument("--log-n", type=int, default=15, help="evaluation domain (15: pinned for depth 10; 16: upper bound)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-schedule", action="store_true")
    ap.add_argument("--workload", default="transfer", choices=["transfer", "mixed64"],
                    help="transfer: P identical-shape 2-in/2-out transfer proofs per GPU (weak scaling, the headline); "
                         "mixed64: BASELINE config 4 - 32 transfer(2x3) + 13 mint + 19 freeze(3) proofs in total, "
                         "proof i on rank i mod N (strong scaling)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, the real multi-GPU path); gloo only to exercise the N>1 logic on a 1-GPU box "
                         "together with CAPGPU_BENCH_DEVICE=0")
    ap.add_argument("--no-msm", action="store_true")
    ap.add_argument("--msm-log-n", type=int, default=22, help="size of the sharded MSM leg (24 = BASELINE config 5)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "CAPGPU_BENCH_DEVICE" in os.environ:          # test hook: all ranks on one device (gloo only)
        local_rank = int(os.environ["CAPGPU_BENCH_DEVICE"])
    import torch
    dist = None
    coll_dev = "cuda" if args.dist_backend == "nccl" else "cpu"
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from cap_amd import bench_utils as bu
    from cap_amd import lib as cg

    cg.init(local_rank)                       # raises (no fallback) when the HIP library / a gfx950 device is missing
    torch.cuda.set_device(local_rank)
    P, log_n = args.batch, args.log_n
    n = 1 << log_n
    num_inputs = 27
    tau = bu.SplitMix64(0xCA9).field()

    # ---- setup (untimed): SRS, circuit(s), proving key(s), resident witnesses -----------------------------
    t_setup = time.time()
    ext_msg = bytes(range(32))                # stands for the serialised txn-memo verification key
    if args.workload == "transfer":
        plan = [("transfer_2x2", log_n, num_inputs, P)]
    else:
        from cap_amd import parallel as par
        mix = [("transfer_2x3", 32), ("mint", 13), ("freeze_3", 19)]       # src/lib.rs:734-736 ratio 5:2:3
        plan, gi = [], 0
        for kind, cnt in mix:
            mine = [i for i in range(gi, gi + cnt) if i % world == rank]
            gi += cnt
            ln, ni = bu.NOTE_SHAPES[kind]
            plan.append((kind, ln, ni, len(mine)))
        log_n = max(ln for _, ln, _, _ in plan)
        n = 1 << log_n
    srs = cg.srs_generate(tau, n + 3)
    groups = []
    for kind, ln, ni, cnt in plan:
        sc = bu.synthetic_circuit(ln, ni, seed=2 + ln + ni)
        sel, sig = sc.selectors_mont(), sc.sigma_mont()
        pk, _vk = cg.plonk_preprocess(srs, 1 << ln, ni, sel, sig)
        if not groups:
            _vk0 = _vk
        n_wit = max(1, min(cnt, 4))
        wit = [sc.witness(1000 * rank + 3 + i) for i in range(n_wit)]
        g = {"kind": kind, "n": 1 << ln, "num_inputs": ni, "count": cnt, "pk": pk, "sel": sel, "sig": sig}
        if cnt:
            g["wires"] = np.stack([sc.wires_mont(wit[i % n_wit][0]) for i in range(cnt)])
            g["pubs"] = np.stack([bu.to_mont_array(wit[i % n_wit][1]) for i in range(cnt)])
            g["blind"] = np.stack([bu.to_mont_array(bu.blinders(7000 + 100 * rank + i)) for i in range(cnt)])
            g["d_wires"] = cg.DevBuf.from_numpy(g["wires"])
        groups.append(g)
    g0 = groups[0]
    pk, sel, sig = g0["pk"], g0["sel"], g0["sig"]
    wires, pubs, blind = g0.get("wires"), g0.get("pubs"), g0.get("blind")
    t_setup = time.time() - t_setup

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def step(key):
        out_proofs = []
        for g in groups:
            if g["count"]:
                k = key if (key is not None and g is g0) else g["pk"]
                out_proofs += cg.plonk_prove_batch_dev(k, g["d_wires"], g["pubs"], g["blind"], ext_msg, g["count"])
        return out_proofs

    def timed(key, steps, warmup, profile):
        for _ in range(warmup):
            step(key)
        if profile:
            cg.profile_reset()
            cg.profile_enable(True)
        sync_all()
        t0 = time.perf_counter()
        for _ in range(steps):
            proofs = step(key)
        sync_all()
        dt = time.perf_counter() - t0
        stats = cg.profile_stats() if profile else {}
        cg.profile_enable(False)
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, stats, proofs

    dt, stats, proofs = timed(pk, args.steps, args.warmup, profile=True)
    per_step_all_ranks = P * world if args.workload == "transfer" else 64
    total_proofs = per_step_all_ranks * args.steps
    value = total_proofs / dt

    # ---- roofline of the dominant kernel (HIP events on the launch stream, timed region itself) -------------
    ab = algorithmic_bytes_per_proof(n)
    kern_ms = {k: v[0] for k, v in stats.items()}
    dom = max(kern_ms, key=kern_ms.get) if kern_ms else None
    per_step_bytes = {
        # K5: 96 B per (point, scalar) pair handled by the launch (64 B affine base + 32 B scalar)
        "msm_accumulate": ab["msm_bytes"] * P,
        # K2: 64 B per element per transform; a transform of 2^11..2^20 elements is one column + one row pass
        "ntt_col_pass": 0.5 * (64 * (7 * n + 8 * 6 * n)) * P,
        "ntt_row_pass": 0.5 * (64 * (7 * n + 8 * 6 * n)) * P,
        # K8: 26 arrays of 6n elements (25 in, 1 out) x 32 B (jf-plonk's 8n domain would be a third more)
        "k_quotient": 26 * 6 * n * 32 * P,
    }
    traffic_tab = {}
    try:
        with open(os.path.join(ROOT, "profiles", "traffic_r01.json")) as f:
            traffic_tab = json.load(f).get("per_launch_bytes", {})
    except OSError:
        pass
    roofline = None
    if dom is not None:
        launches = stats[dom][1]
        avg_ms = stats[dom][0] / max(launches, 1)
        if dom in per_step_bytes and args.workload == "transfer":
            bytes_per_launch = per_step_bytes[dom] * args.steps / max(launches, 1)
            achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        else:
            bytes_per_launch, achieved = None, None
        roofline = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": (achieved / HBM_PEAK_GBPS) if achieved else None,
                    # HBM bytes per launch from the committed PMC pass (profiles/traffic_r01.json, batch 16) or null
                    "traffic": traffic_tab.get(dom) if (P == TRAFFIC_BATCH and log_n == 15 and args.workload == "transfer") else None,
                    "avg_launch_ms": avg_ms, "launches": launches, "algorithmic_bytes_per_launch": bytes_per_launch,
                    "share_of_kernel_time": kern_ms[dom] / sum(kern_ms.values())}
    # SURVEY 8d asks for an integer-ALU fraction beside the HBM one: the path is multiplication-bound.  Mixed additions of
    # a step = non-zero digits minus one per non-empty bucket (the first entry of a bucket is a copy); one mixed
    # addition = 8 products + 2 squarings + 9 reductions ~ 10.5 Montgomery multiplications of the lazy 29-bit field,
    # whose measured ceiling in isolation is 158 G/s (tools/ubench_mlo.hip, profiles/ubench_mlo_r01.txt).
    alu = None
    if dom == "msm_accumulate" and args.workload == "transfer" and P >= 32 and n >= 4096:
        digits, buckets = 17, 1 << 14                       # c = 15 table, 254-bit scalars
        adds_per_step = P * (13 * ((n + 2) * digits - buckets) + (n + 3 - (n + 2)) * digits)
        mul_eq = adds_per_step * args.steps * 10.5 / (kern_ms[dom] * 1e-3) / 1e9
        alu = {"kernel": dom, "achieved": mul_eq, "peak": 158.0, "unit": "G field multiplications/s (equivalent)",
               "frac": mul_eq / 158.0, "mixed_adds_per_step": adds_per_step}
    top = sorted(kern_ms.items(), key=lambda kv: -kv[1])[:8]
    whole_gbps = ab["total_bytes"] * total_proofs / dt / 1e9

    out = {
        "metric": "transfer-note proofs/sec (2-in/2-out)", "value": value, "unit": "proofs/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak" if args.workload == "transfer" else "strong", "vs_baseline": None, "dtype": "u32 limbs (254-bit Montgomery integers; 9 x 29-bit lazy limbs in the hot kernels)", "data": "synthetic",
        "config": {"workload": (f"full 2-in/2-out transfer-note PLONK proof (13 MSM + 33 NTT), n=2^{log_n}, 27 public inputs, "
                                f"batch {P} proofs/step/GPU, device-resident witness + key + SRS") if args.workload == "transfer"
                   else "BASELINE config 4: 64 mixed proofs per step = 32 transfer(2-in/3-out, n=2^15) + 13 mint (n=2^14) + "
                        "19 freeze(3 inputs, n=2^15), one SRS, three keys, proof i on rank i mod N",
                   "domain_size": n, "batch_per_gpu": P, "parallelism": f"replicas x{world} (independent proofs)",
                   "pk_coset_cache": "18 fixed selector/sigma coset NTTs cached in the proving key (see "
                                     "reference_schedule for the per-proof recompute number)"},
        "roofline": roofline,
        "alu_roofline": alu,
        "proof_hbm_roofline": {"algorithmic_bytes_per_proof": ab["total_bytes"], "achieved_GBps": whole_gbps,
                               "frac_of_peak": whole_gbps / HBM_PEAK_GBPS},
        "top_kernels_ms": {k: round(v, 3) for k, v in top},
        "setup_s": round(t_setup, 2),
    }

    if world == 1 and not args.no_reference_schedule and args.workload == "transfer":
        os.environ["CAPGPU_RECOMPUTE_PK_COSET"] = "1"
        pk_ref, _ = cg.plonk_preprocess(srs, n, num_inputs, sel, sig)
        os.environ.pop("CAPGPU_RECOMPUTE_PK_COSET")
    if not args.no_reference_schedule and args.workload == "transfer":
        # every rank must take part (barriers); only rank 0 built the key when world == 1
        if world == 1:
            rs = max(2, args.steps // 2)
            dt_ref, _, proofs_ref = timed(pk_ref, rs, 1, profile=False)
            a, b = cg.proof_to_arrays(proofs_ref[0]), cg.proof_to_arrays(proofs[0])
            same = all(np.array_equal(a[k], b[k]) for k in a)
            out["reference_schedule"] = {"proofs_per_s": P * rs / dt_ref, "steps": rs,
                                         "note": "all 25 coset NTTs re-run per proof as jf-plonk does",
                                         "proof_identical_to_cached_mode": bool(same)}
            cg.plonk_free_key(pk_ref)

    # ---- CPU baseline: the C restatement of the arkworks/jf-plonk algorithm, 1 thread, rank 0, N = 1 ----------
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload == "transfer":
        from oracle import capref as cr        # cpu_baseline leg: the only place bench.py touches oracle/
        key = cr.PlonkKey(cg.srs_download(srs, 0, n + 3), n, num_inputs, sel, sig)
        n_cpu = 2 if log_n <= 15 else 1          # bounded sample: ~10-12 s of single-thread work
        parity = True
        t0 = time.perf_counter()
        cpu_out = [key.prove(wires[i], pubs[i], blind[i], ext_msg) for i in range(n_cpu)]
        t_cpu = (time.perf_counter() - t0) / n_cpu
        for i, (rc, comms, evals) in enumerate(cpu_out):
            a = cg.proof_to_arrays(proofs[i])
            got_comms = np.concatenate([a["wires_poly_comms"], a["prod_perm_poly_comm"][None],
                                        a["split_quot_poly_comms"], a["opening_proof"][None],
                                        a["shifted_opening_proof"][None]])
            got_evals = np.concatenate([a["wires_evals"], a["wire_sigma_evals"], a["perm_next_eval"][None]])
            parity = parity and bool(rc == 0 and np.array_equal(got_comms, comms) and np.array_equal(got_evals, evals))
        h2 = cg.g2_generator()
        t0 = time.perf_counter()
        accepted = cg.plonk_verify(_vk0, h2, cg.g2_mul(h2, tau), pubs[0], proofs[0], ext_msg)
        out["verify"] = {"accepted_by_product_verifier": bool(accepted), "ms": (time.perf_counter() - t0) * 1e3,
                         "note": "host-side pairing check (capgpu_plonk_verify), outside the timed region"}
        out["cpu_baseline"] = {"value": 1.0 / t_cpu, "unit": "proofs/s", "cores": 1, "kind": "port",
                               "sample": f"{n_cpu} proofs of the same workload (n=2^{log_n}), {t_cpu * n_cpu:.1f} s, single-thread C "
                                         "restatement of the arkworks/jf-plonk algorithm (reference schedule, no asm)",
                               "gpu_proof_bit_exact_vs_cpu": parity}
        out["speedup_vs_cpu_1core"] = value / (1.0 / t_cpu)
    # ---- MSM leg: BASELINE config 2 (2^17 points) on one GPU, and a point-range-sharded MSM over all ranks ------
    if not args.no_msm:
        legs = []
        if world == 1:
            legs.append(msm_leg(cg, bu, torch, dist, rank, world, 17, coll_dev=coll_dev))
        legs.append(msm_leg(cg, bu, torch, dist, rank, world, args.msm_log_n, coll_dev=coll_dev))
        out["msm"] = legs
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
