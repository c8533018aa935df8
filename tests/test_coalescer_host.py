"""The threading logic behind capgpu_plonk_set_coalescing (cap_amd/csrc/coalescer.hpp: queues, leaders, windows, batches cut
over two contexts, early release) under ThreadSanitizer on the host, with stub contexts and a stub prover - no GPU
(round-4 VERDICT item 8; the GPU suite exercises the same code with the real prover: tests/test_gpu_plonk.py,
tools/gpu_thread_stress.py).  (`-m "not gpu"`)"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "coalescer_tsan.cpp")


def build(tmp_path, flags, src=SRC, name="coalescer_test"):
    cxx = shutil.which("g++") or shutil.which("clang++")
    if not cxx:
        pytest.skip("no C++ compiler")
    exe = str(tmp_path / name)
    r = subprocess.run([cxx, "-std=c++17", "-O1", "-g", "-Wall", "-Wextra", "-Werror", "-pthread"] + flags + [src, "-o", exe],
                       capture_output=True, text=True)
    return exe, r


def test_coalescer_protocol_plain(tmp_path):
    exe, r = build(tmp_path, [])
    assert r.returncode == 0, r.stderr[-2000:]
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout[-1500:] + out.stderr[-1500:]


def test_coalescer_protocol_under_thread_sanitizer(tmp_path):
    exe, r = build(tmp_path, ["-fsanitize=thread"])
    if r.returncode != 0 and ("tsan" in r.stderr.lower() or "sanitize" in r.stderr.lower()):
        pytest.skip("this toolchain has no ThreadSanitizer runtime")
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    if out.returncode != 0 and "FATAL: ThreadSanitizer" in out.stderr and "WARNING: ThreadSanitizer" not in out.stderr:
        pytest.skip("ThreadSanitizer cannot start in this container: " + out.stderr.strip().splitlines()[0])
    assert "WARNING: ThreadSanitizer" not in out.stderr, out.stderr[-3000:]
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout[-1500:] + out.stderr[-1500:]


@pytest.mark.parametrize("flags", [[], ["-fsanitize=thread"]])
def test_phase_trace_ring(tmp_path, flags):
    """cap_amd/csrc/trace.hpp (capgpu_trace_enable / _dump): eight threads emit 40 000 events; the dump holds each once, a
    thread's events in order; off records nothing; a new enable starts afresh - plain and under ThreadSanitizer"""
    exe, r = build(tmp_path, flags, src=os.path.join(ROOT, "tests", "cpp", "trace_check.cpp"), name="trace_check")
    if r.returncode != 0 and flags and ("tsan" in r.stderr.lower() or "sanitize" in r.stderr.lower()):
        pytest.skip("this toolchain has no ThreadSanitizer runtime")
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66")
    out = subprocess.run([exe, str(tmp_path / "trace.txt")], capture_output=True, text=True, timeout=300, env=env)
    if flags and out.returncode != 0 and "FATAL: ThreadSanitizer" in out.stderr and "WARNING: ThreadSanitizer" not in out.stderr:
        pytest.skip("ThreadSanitizer cannot start in this container: " + out.stderr.strip().splitlines()[0])
    assert "WARNING: ThreadSanitizer" not in out.stderr, out.stderr[-3000:]
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout[-800:] + out.stderr[-1500:]
