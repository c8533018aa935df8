"""The lazy 29-bit field / curve code is host+device; these CPU tests compile it for the host with its bound
assertions enabled (CAP_FL_CHECK) and check it against Python integers and against the 32-bit field code.
(`-m "not gpu"`)"""
import os
import random
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CPP = os.path.join(HERE, "cpp")
P = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
RR = 1 << 261


def _cxx():
    for c in ("g++", "/opt/rocm/lib/llvm/bin/clang++", "clang++"):
        if shutil.which(c) or os.path.exists(c):
            return c
    pytest.skip("no host C++ compiler")


@pytest.fixture(scope="module", params=["rowwise", "colwise"])
def binaries(tmp_path_factory, request):
    """Both schedules of the Montgomery multiplication (field29.hpp: row-wise column accumulators / column-wise with
    the carry as the multiply-add's addend) are built and must give the same results within the same bounds."""
    out = tmp_path_factory.mktemp("f29" + request.param)
    cxx = _cxx()
    bins = {}
    for name in ("field29_check", "curve29_check", "quad_check"):
        exe = str(out / name)
        flags = ["-DCAP_FL_COLWISE"] if request.param == "colwise" else ["-DCAP_FL_ROWWISE"]
        subprocess.check_call([cxx, "-O1", "-std=c++17"] + flags + [os.path.join(CPP, name + ".cpp"), "-o", exe])
        bins[name] = exe
    return bins


def test_field29_against_python_integers(binaries):
    random.seed(5)
    lines, exp = [], []
    for w, m in (("q", P), ("r", R)):
        inv = pow(RR, -1, m)
        for it in range(300):
            ka, kb = random.choice([1, 2, 4, 17, 40]), random.choice([1, 2, 4, 15])
            a = random.randrange(int(ka * m)) if it > 20 else random.choice([0, 1, m - 1, m, 2 * m - 1, int(16.9 * m)])
            b = random.randrange(int(kb * m)) if it > 20 else random.choice([0, 1, m - 1, m, 2 * m, int(15.8 * m)])

            def add(op, e):
                lines.append(f"{w} {op} {a:x} {b:x}")
                exp.append((op, e, m))
            add("m", a * b * inv % m)
            add("q", a * a * inv % m)
            add("w", a % m)
            add("c", a % m)
            add("z", 1 if a % m == 0 else 0)
            if b < 15.9 * m:
                add("s", (a - b) % m)
            add("a", (a + b) % m)
            if a < (1 << 256):
                add("p", a)
                add("e", a * 32 % m)
                add("t", a * RR % m)
            add("x", a * pow(32, -1, m) % m)
            add("f", a * inv % m)
            if a < 8 * m and b < 8 * m:
                add("M", 2 * a * b * inv % m)
            add("E", 1 if (a - b) % m == 0 else 0)
            if it % 10 == 0:    # inverse in the Montgomery domain: (a / R')^-1 * R' = R'^2 / a
                add("i", pow(a, -1, m) * RR * RR % m if a % m else 0)
        # the constant-multiplicand product (mul_shoup): any lazy a with limbs < 2^30 (here: up to 2^262 - 1 in value,
        # written as a normalized 261-bit part the parser spreads over the limbs), every canonical constant w
        for it in range(300):
            a = random.randrange(1 << 261) if it > 12 else random.choice([0, 1, m - 1, m, (1 << 261) - 1, 168 * m, 5 * m - 1])
            b = random.randrange(m) if it > 12 else random.choice([0, 1, 2, m - 1, m - 2, m // 2, (1 << 253), 3])
            lines.append(f"{w} S {a:x} {b:x}")
            exp.append(("S", a * b % m, m))
            lines.append(f"{w} Q {a:x} {b:x}")
            exp.append(("Q", (b << 261) // m, m))
            b8 = random.randrange(int(7.9 * m))
            lines.append(f"{w} 8 {a % (1 << 260):x} {b8:x}")
            exp.append(("8", (a % (1 << 260) - b8) % m, m))
        for k in range(0, 160, 7):       # every multiple of p is recognised as zero
            lines.append(f"{w} z {k * m:x} 0")
            exp.append(("z", 1, m))
    out = subprocess.run([binaries["field29_check"]], input="\n".join(lines) + "\n", capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-500:]
    res = out.stdout.split()
    assert len(res) == len(exp)
    for line, o, (op, e, m) in zip(lines, res, exp):
        v = int(o, 16)
        if op in "zEpcxfQ":
            assert v == e, line
        elif op == "S":
            assert v % m == e and v < 4 * m, line        # (a < 2^261 here: the 4p bound)
        elif op == "w":
            assert v % m == e and v < 2 * m, line
        else:
            assert v % m == e and v < (1 << 261), line


CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.fixture(scope="module")
def sanitized(tmp_path_factory):
    """The same host checks built with clang's unsigned-integer-overflow sanitizer: every 64-bit column sum and 32-bit
    limb sum of field29.hpp / curve29.hpp traps on wrap-around (the saturated reference headers, which wrap by design,
    are on the ignore list; the few intentional wraps of field29.hpp carry CAP_WRAPS)."""
    if not os.path.exists(CLANG):
        pytest.skip("no clang++ for the sanitizer build")
    out = tmp_path_factory.mktemp("f29san")
    ign = out / "ignore.txt"
    ign.write_text("src:*/field.hpp\nsrc:*/curve.hpp\n")
    bins = {}
    for name in ("curve29_check", "madd_check", "quad_check"):
        exe = str(out / name)
        subprocess.check_call([CLANG, "-O1", "-std=c++17", "-fsanitize=unsigned-integer-overflow",
                               f"-fsanitize-ignorelist={ign}", "-fno-sanitize-recover=all",
                               os.path.join(CPP, name + ".cpp"), "-o", exe])
        bins[name] = exe
    return bins


def test_accumulation_loop_addition(tmp_path):
    """G1L::madd_acc (the loop body of msm_accumulate, with its un-carried operands) against the saturated curve code:
    long signed chains through doubling, cancellation, infinity and the 32-byte memory image, both multiplication
    schedules, bound assertions on."""
    exe = str(tmp_path / "madd_check")
    subprocess.check_call([_cxx(), "-O1", "-std=c++17", os.path.join(CPP, "madd_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "bad=0" in out.stdout, out.stdout[-500:] + out.stderr[-500:]


def test_no_integer_wraps_under_the_sanitizer(sanitized):
    for name, exe in sanitized.items():
        out = subprocess.run([exe], capture_output=True, text=True)
        assert out.returncode == 0 and "bad=0" in out.stdout, name + ": " + out.stdout[-300:] + out.stderr[-800:]


def test_curve29_against_saturated_curve_code(binaries):
    out = subprocess.run([binaries["curve29_check"]], capture_output=True, text=True)
    assert out.returncode == 0 and "bad=0" in out.stdout, out.stdout[-500:] + out.stderr[-500:]


def test_quad_point_arithmetic_on_simulated_lanes(binaries):
    """quad29.hpp - one XYZZ point spread over four lanes, additions four multiplications deep (the tails of the small MSM
    launches) - on four simulated lanes with the bound assertions on, against the one-lane G1LT::add / dbl: single
    operations, chains fed back through the memory image, a tree, infinity, P + P and P - P."""
    out = subprocess.run([binaries["quad_check"]], capture_output=True, text=True)
    assert out.returncode == 0 and "bad=0" in out.stdout, out.stdout[-500:] + out.stderr[-500:]


def test_host_multiplication_and_inversion_match_the_device_forms(tmp_path):
    """field.hpp on the host: 64-bit-limb CIOS and the Euclidean inversion (the prover's scalars between the rounds, the
    verifier) against the 32-bit-limb CIOS and the Fermat inversion the device code runs - the same bits for any 256-bit
    inputs, both fields, edge values included."""
    exe = str(tmp_path / "field_host_check")
    subprocess.check_call([_cxx(), "-O2", "-std=c++17", os.path.join(CPP, "field_host_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "bad=0" in out.stdout, out.stdout[-500:] + out.stderr[-500:]
