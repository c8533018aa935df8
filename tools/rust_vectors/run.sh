#!/usr/bin/env bash
# ONE command that pins this repository's parity against the reference itself (arkworks 0.3 / jf-plonk @ bcd92b2).
# Needs what this repository's build box does not have: cargo + rustc (>= 1.56), network access for the crates, and a
# checkout of EspressoSystems/cap (for its Cargo.lock).  Nothing else - no GPU for steps 1-3.
#
#     tools/rust_vectors/run.sh /path/to/EspressoSystems-cap [--gpu]
#
#   1. copies the reference's Cargo.lock next to this crate (same transitive versions as the reference builds with)
#   2. cargo run --release: arkworks / jellyfish compute the vectors and write tests/golden/ref_{msm,ntt,proof,params}.json
#   3. runs the consumers on the CPU: Python oracle, C restatement, host verifier, parameter (de)serialisers
#      (tests/test_ref_vectors.py; until step 2 has run they skip with "parity unpinned")
#   4. with --gpu (on an MI355X box): the same vectors through the HIP path behind the C ABI
# Exit code 0 = parity pinned.  A mismatch fails the test that names the FIRST differing field (transcript layout,
# selector order, blinder order and blob field order are recollection until then: DESIGN.md section 2 and 7).
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
repo="$(cd "$here/../.." && pwd)"
ref="${1:-}"
if [ -z "$ref" ] || [ ! -f "$ref/Cargo.lock" ]; then
  echo "usage: $0 /path/to/EspressoSystems-cap [--gpu]   (the directory that holds the reference's Cargo.lock)" >&2
  exit 2
fi
command -v cargo >/dev/null || { echo "cargo not found: this step needs a Rust toolchain" >&2; exit 2; }
expected=(ref_msm.json ref_ntt.json ref_proof.json ref_proof_sparse.json ref_params.json)

echo "[1/4] Cargo.lock of the reference -> $here"
cp "$ref/Cargo.lock" "$here/Cargo.lock"

echo "[2/4] cargo run --release -- $repo/tests/golden"
(cd "$here" && cargo run --release -- "$repo/tests/golden")
for f in "${expected[@]}"; do
  [ -s "$repo/tests/golden/$f" ] || { echo "missing $repo/tests/golden/$f" >&2; exit 1; }
done

echo "[3/4] CPU consumers"
(cd "$repo" && make -C oracle -s && python3 -m pytest tests/test_ref_vectors.py -q -m "not gpu" -rs)

if [ "${2:-}" = "--gpu" ]; then
  echo "[4/4] HIP path (needs an MI355X)"
  (cd "$repo" && python3 -c "import __graft_entry__ as g; g.build()" && python3 -m pytest tests/test_ref_vectors.py -q -m gpu -rs)
else
  echo "[4/4] skipped (pass --gpu on an MI355X box):  python3 -m pytest tests/test_ref_vectors.py -q -m gpu"
fi
echo "parity pinned: commit tests/golden/ref_*.json   (git add ${expected[*]/#/tests/golden/})"
