"""Static instruction mix of the issue-bound kernels, by issue class, from the compiler's own assembly:
    python tools/isa_mix.py [profiles/isa_mix_r04.json]
Compiles the library's sources for gfx950 with its flags (-S, device only; no GPU needed) and counts instructions per
class - the classes capgpu_ubench_issue_rates measures on the device.

  msm_accumulate        the common path of ONE mixed addition exactly: the loop header block and the block G1L::madd_acc
                        marks with an asm comment (`per_class` at the top level, as in round 3).  bench.py multiplies it
                        by the step's additions: `alu_roofline.issue_frac`.
  ntt_col_pass, ntt_row_pass, k_quotient, msm_reduce_segments
                        these kernels run several loops whose trip counts the assembly does not show, so only the
                        SHARES of the classes are taken from it - over the kernel's arithmetic blocks (every basic block
                        with at least 8 multiply-adds: field multiplications; the address arithmetic and loop control
                        around them are left out, which prices the kernels slightly too dear) - and bench.py applies
                        them to the VALU instructions the kernel really executed (SQ_INSTS_VALU, static PMC pass):
                        `issue_frac_other_kernels`.
bench.py prices the mixes against the issue rates measured in the same run."""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dest = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "isa_mix_r04.json")
CLASS_OF = [
    (r"^v_mad_u64_u32|^v_mad_i64_i32", "v_mad_u64_u32"),
    (r"^v_mul_lo_u32|^v_mul_hi_u32", "v_mul_lo_u32"),
    (r"^v_lshrrev_b64|^v_ashrrev_i64|^v_lshlrev_b64", "v_lshrrev_b64"),
    (r"^v_lshl_add_u64", "v_lshl_add_u64"),
    (r"^v_alignbit_b32", "v_alignbit_b32"),
    (r"^v_mov_b32|^v_accvgpr", "v_mov_b32"),
    (r"^v_and_b32|^v_or_b32|^v_xor_b32|^v_and_or_b32|^v_or3_b32", "v_and_b32"),
    (r"^v_", "v_add_u32"),                    # every other VALU instruction: priced like a 32-bit add
]


def classify(mn):
    for pat, cls in CLASS_OF:
        if re.match(pat, mn):
            return cls
    return "non_valu"


def assembly(src):
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "out.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-S", "--cuda-device-only", "-O3", "-std=c++17", "--offload-arch=gfx950",
                               "-ffp-contract=off", "-w", os.path.join(ROOT, "cap_amd", "csrc", src), "-o", asm])
        return open(asm).read().split("\n")


def kernel_blocks(lines, mangled_re):
    """basic blocks of the (first) kernel whose mangled name matches: branch targets (.LBBn_m:) and fall-through blocks"""
    start = [i for i, ln in enumerate(lines) if re.match(mangled_re, ln)][0]
    end = [i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end")][0]
    blocks, cur = [], None
    cur = {"label": "entry", "header": False, "marked": False, "ops": [], "loop": None}
    blocks.append(cur)
    for ln in lines[start + 1:end]:
        s = ln.strip()
        if re.match(r"^\.LBB\d+_\d+:", s) or s.startswith("; %bb."):
            inside = re.search(r"in Loop: Header=(BB\d+_\d+)", s)
            cur = {"label": s.split(":")[0].replace("; ", ""), "header": "Loop Header: Depth=" in s, "marked": False, "ops": [],
                   "loop": ".L" + inside.group(1) if inside else None}
            blocks.append(cur)
        elif "madd_acc: common path" in s:
            cur["marked"] = True
        elif s.startswith(";") and "Loop Header: Depth=" in s:   # (a nested loop's header comment follows its label line)
            cur["header"] = True
        elif s and not s.startswith((".", ";", "//")):
            cur["ops"].append(s.split()[0])
    return blocks


def mix_of(blocks):
    mix = collections.Counter()
    for b in blocks:
        for op in b["ops"]:
            mix[classify(op)] += 1
    return mix


# ---- msm_accumulate: one mixed addition, exactly ---------------------------------------------------------------------
msm = assembly("msm.hip")
blocks = kernel_blocks(msm, r"^_ZN3cap12_GLOBAL__N_114msm_accumulate.*:\s*")
# The common path of one mixed addition, in layout order: the loop header and the small blocks after it (list entry,
# 64-byte gather, unpack, sign), the block with the two products that feed the x-difference test (u2, s2), and the block
# G1L::madd_acc marks with an asm comment - everything after the test.  (Round 3 first took "the largest block" here, which
# is the general addition G1L::add_mixed - the fallback for a bucket still at infinity - with 64 x 32-bit products that the
# common path does not have.)
im = [i for i, b in enumerate(blocks) if b["marked"]]
assert len(im) == 1, "expected exactly one marked block in msm_accumulate"
im = im[0]
# the header of the loop over a work item's entries: the innermost loop the marked block belongs to (since the kernel became
# a persistent launch that loop sits inside the loop over chunks of items, and it has small loops of its own inside - the
# exact zero test's - which the common path does not enter)
ih = [i for i, b in enumerate(blocks) if b["label"] == blocks[im]["loop"]][0]
ip = max(i for i in range(ih, im) if sum(op.startswith("v_mad_u64") for op in blocks[i]["ops"]) >= 300)
path = [b for b in blocks[ih:ip] if len(b["ops"]) < 150 and "v_mad_i64_i32" not in b["ops"]] + [blocks[ip], blocks[im]]
mix = mix_of(path)
valu = sum(v for k, v in mix.items() if k != "non_valu")
out = {"kernel": "msm_accumulate", "source": "hipcc -S --cuda-device-only -O3 --offload-arch=gfx950 cap_amd/csrc/*.hip (tools/isa_mix.py)",
       "blocks": {b["label"]: len(b["ops"]) for b in path},
       "what": "loop header (list entry, 64-byte gather, unpack, sign) + the inlined G1L::madd_acc: one mixed addition on "
               "the common path",
       "valu_instructions_per_mixed_addition": valu, "per_class": dict(mix)}


# ---- the other issue-bound kernels: class shares over their arithmetic blocks ---------------------------------------
def shares(lines, mangled_re, name):
    bl = kernel_blocks(lines, mangled_re)
    hot = [b for b in bl if sum(op.startswith("v_mad_u64") for op in b["ops"]) >= 8]
    m = mix_of(hot)
    v = {k: c for k, c in m.items() if k != "non_valu"}
    tot = sum(v.values())
    allv = sum(c for k, c in mix_of(bl).items() if k != "non_valu")
    return {"kernel": name, "arithmetic_blocks": len(hot), "blocks": len(bl), "valu_in_arithmetic_blocks": tot,
            "valu_in_all_blocks": allv, "class_share": {k: c / tot for k, c in sorted(v.items())}}


ntt = assembly("ntt.hip")
plonk = assembly("plonk.hip")
others = {}
# (the one-tile-per-workgroup instantiations of the NTT passes: template argument false)
for lines, pat, name in ((ntt, r"^_ZN3cap\S*12ntt_col_passILb0E\S*:\s*", "ntt_col_pass"),
                         (ntt, r"^_ZN3cap\S*12ntt_row_passILb0E\S*:\s*", "ntt_row_pass"),
                         (plonk, r"^_ZN3cap\S*10k_quotient\S*:\s*", "k_quotient"),
                         (msm, r"^_ZN3cap12_GLOBAL__N_119msm_reduce_segments\S*:\s*", "msm_reduce_segments")):
    try:
        others[name] = shares(lines, pat, name)
    except IndexError:
        others[name] = {"kernel": name, "error": "kernel symbol not found in the assembly"}
out["other_kernels"] = others
out["other_kernels_what"] = ("class shares over the basic blocks that hold field arithmetic (>= 8 multiply-adds); applied by "
                             "bench.py to the executed VALU count of the static PMC pass (SQ_INSTS_VALU)")
json.dump(out, open(dest, "w"), indent=1)
print(json.dumps(out))
