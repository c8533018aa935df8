"""The JSON line bench.py prints is ordered so that a log keeping only its tail still shows what the contract names
(cap_amd/bench_legs.py: headline_last; round-5 VERDICT item 7) - checked here on a synthetic line, without a GPU."""
import json

from cap_amd import bench_legs as bl


def fake_line():
    legs = {"msm_2^17_ms": {"median": 0.477, "GBps_algorithmic": 26.4}, "msm_2^17_x64_ms": {"median": 11.2, "GBps_algorithmic": 71.6},
            "ntt_2^17_x1_ms": {"median": 0.0427, "GBps_algorithmic": 196.0}, "single_proof_ms": {"median": 2.46}}
    return {
        "metric": "transfer-note proofs/sec (2-in/2-out)", "value": 1330.0, "unit": "proofs/s", "n_gpus": 2, "steps": 20,
        "warmup": 5, "ms_per_step": 192.5, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32 limbs",
        "data": "synthetic",
        "config": {"workload": "full 2-in/2-out transfer-note PLONK proof", "legs": legs, "device_memory_in_use_GB": 120.5,
                   "device_memory_after_trim_GB": 5.9, "blob": "x" * 5000},
        "top_kernels_ms": {"msm_accumulate": 1000.0}, "alu_roofline": {"long": "y" * 3000},
        "roofline": {"bound": "hbm", "kernel": "msm_accumulate", "achieved": 95.5, "peak": 8000.0, "unit": "GB/s", "frac": 0.0119,
                     "traffic": 3.7e10, "traffic_source": "z" * 400, "measured_in": "w" * 200, "avg_launch_ms": 27.4},
        "cpu_baseline": {"value": 0.178, "unit": "proofs/s", "cores": 1, "kind": "port", "sample": "2 proofs", "clock": "c" * 100,
                         "cpu_model": "EPYC", "timed_on": "rank 0 of 2"},
        "pcie_inclusive": {"proofs_per_s": 1255.0, "over_resident_same_minute": 0.948},
        "coalesced_single_calls": {"proofs_per_s": 1168.0, "over_resident_same_minute": 0.882},
        "n1_same_run": {"proofs_per_s": 1318.0, "value_over_n_times_this": 0.5045},
        "mixed64_multi_gpu": {"mode_B_replicas_proofs_per_s": 1290.0, "note": "n" * 300},
    }


def test_contract_keys_come_last_and_the_tail_is_short():
    out = bl.headline_last(fake_line())
    keys = list(out)
    assert keys[-12:] == ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                          "vs_baseline", "dtype", "data"]
    assert keys[-16:-12] == ["config", "summary", "roofline", "cpu_baseline"]
    tail = json.dumps({k: out[k] for k in keys[-15:]})
    assert len(tail) < 1900, len(tail)
    # nothing is dropped: the long strings moved to *_notes, every original key is still there
    assert set(fake_line()) <= set(out)
    assert out["roofline_notes"]["traffic_source"].startswith("z") and "traffic_source" not in out["roofline"]
    assert out["cpu_baseline_notes"]["cpu_model"] == "EPYC" and out["cpu_baseline"]["kind"] == "port"
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in out["roofline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in out["cpu_baseline"]


def test_summary_holds_the_legs_and_the_ratios():
    s = bl.headline_last(fake_line())["summary"]
    assert s["workload"].startswith("full 2-in/2-out")
    assert s["pcie_inclusive_proofs_per_s"] == 1255.0 and abs(s["pcie_inclusive_over_value"] - 1255 / 1330) < 1e-3
    assert s["pcie_inclusive_over_resident_same_minute"] == 0.948
    assert s["coalesced_single_calls_over_resident_same_minute"] == 0.882
    assert s["n1_same_run_proofs_per_s"] == 1318.0 and s["value_over_n_times_n1_same_run"] == 0.5045
    assert s["mixed64_proofs_per_s"] == {"mode_B_replicas": 1290.0}
    assert s["legs_median_ms"]["msm_2^17_x64"] == 11.2 and s["legs_median_ms"]["single_proof"] == 2.46
    assert s["msm_GBps"] == {"msm_2^17": 26.4, "msm_2^17_x64": 71.6}
    assert s["device_memory_after_trim_GB"] == 5.9


def test_algorithmic_bytes_are_the_survey_figures():
    """13 (n + 2) 96 B + (7 n + 26 x 8 n) 64 B = 0.492 GB at n = 2^15 (z has n + 3 coefficients: + 96 B)"""
    ab = bl.algorithmic_bytes_per_proof(1 << 15)
    assert ab["msm_pairs"] == 13 * ((1 << 15) + 2) + 1
    assert ab["ntt_elems"] == 7 * (1 << 15) + 26 * 8 * (1 << 15)
    assert abs(ab["total_bytes"] / 1e9 - 0.4918) < 5e-4
