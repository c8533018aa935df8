"""Static instruction mix of msm_accumulate's hot path, by issue class, from the compiler's own assembly:
    python tools/isa_mix.py [profiles/isa_mix_r03.json]
Compiles cap_amd/csrc/msm.hip for gfx950 with the library's flags (-S, device only; no GPU needed), finds the kernel, takes
its loop header block and its largest block (the inlined G1L::madd_acc: the common path of every mixed addition) and
counts the instructions per class.  bench.py prices this mix against the issue rates capgpu_ubench_issue_rates measures
on the device (`alu_roofline.issue_frac`)."""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dest = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "isa_mix_r03.json")
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


with tempfile.TemporaryDirectory() as tmp:
    asm = os.path.join(tmp, "msm.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-S", "--cuda-device-only", "-O3", "-std=c++17", "--offload-arch=gfx950",
                           "-ffp-contract=off", "-w", os.path.join(ROOT, "cap_amd", "csrc", "msm.hip"), "-o", asm])
    lines = open(asm).read().split("\n")
start = [i for i, ln in enumerate(lines) if re.match(r"^_ZN3cap12_GLOBAL__N_114msm_accumulate.*:\s*", ln)][0]
end = [i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end")][0]
# basic blocks: branch targets (.LBBn_m:) and fall-through blocks ("; %bb.k:")
blocks, cur = [], None
for ln in lines[start:end]:
    s = ln.strip()
    if re.match(r"^\.LBB\d+_\d+:", s) or s.startswith("; %bb."):
        cur = {"label": s.split(":")[0].replace("; ", ""), "header": "Loop Header: Depth=1" in s, "marked": False, "ops": []}
        blocks.append(cur)
    elif cur is not None and "madd_acc: common path" in s:
        cur["marked"] = True
    elif cur is not None and s and not s.startswith((".", ";", "//")):
        cur["ops"].append(s.split()[0])
# The common path of one mixed addition, in layout order: the loop header and the small blocks after it (list entry,
# 64-byte gather, unpack, sign), the block with the two products that feed the x-difference test (u2, s2), and the block
# G1L::madd_acc marks with an asm comment - everything after the test.  (Round 3 took "the largest block" here, which is
# the general addition G1L::add_mixed - the fallback for a bucket still at infinity - with 64 x 32-bit products that the
# common path does not have.)
ih = [i for i, b in enumerate(blocks) if b["header"]][0]
im = [i for i, b in enumerate(blocks) if b["marked"]]
assert len(im) == 1, "expected exactly one marked block in msm_accumulate"
im = im[0]
ip = max(i for i in range(ih, im) if sum(op.startswith("v_mad_u64") for op in blocks[i]["ops"]) >= 300)
path = [b for b in blocks[ih:ip] if len(b["ops"]) < 150 and "v_mad_i64_i32" not in b["ops"]] + [blocks[ip], blocks[im]]
mix = collections.Counter()
for b in path:
    for op in b["ops"]:
        mix[classify(op)] += 1
valu = sum(v for k, v in mix.items() if k != "non_valu")
out = {"kernel": "msm_accumulate", "source": "hipcc -S --cuda-device-only -O3 --offload-arch=gfx950 cap_amd/csrc/msm.hip (tools/isa_mix.py)",
       "blocks": {b["label"]: len(b["ops"]) for b in path},
       "what": "loop header (list entry, 64-byte gather, unpack, sign) + the inlined G1L::madd_acc: one mixed addition on "
               "the common path",
       "valu_instructions_per_mixed_addition": valu, "per_class": dict(mix)}
json.dump(out, open(dest, "w"), indent=1)
print(json.dumps(out))
