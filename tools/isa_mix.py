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
blocks, cur = [], None
for ln in lines[start:end]:
    s = ln.strip()
    if re.match(r"^\.LBB\d+_\d+:", s):
        cur = {"label": s.split(":")[0], "header": "Loop Header" in s, "ops": []}
        blocks.append(cur)
    elif cur is not None and s and not s.startswith((".", ";", "//")):
        cur["ops"].append(s.split()[0])
hot = max(blocks, key=lambda b: len(b["ops"]))
header = [b for b in blocks if b["header"] and "Depth=1" in lines[start:end][0] or b["header"]][0]
mix = collections.Counter()
for b in (header, hot):
    for op in b["ops"]:
        mix[classify(op)] += 1
valu = sum(v for k, v in mix.items() if k != "non_valu")
out = {"kernel": "msm_accumulate", "source": "hipcc -S --cuda-device-only -O3 --offload-arch=gfx950 cap_amd/csrc/msm.hip (tools/isa_mix.py)",
       "blocks": {header["label"]: len(header["ops"]), hot["label"]: len(hot["ops"])},
       "what": "loop header (list entry, 64-byte gather, unpack, sign) + the inlined G1L::madd_acc: one mixed addition on "
               "the common path",
       "valu_instructions_per_mixed_addition": valu, "per_class": dict(mix)}
json.dump(out, open(dest, "w"), indent=1)
print(json.dumps(out))
