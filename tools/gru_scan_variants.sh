#!/bin/bash
# The persistent scans at every benchmark shape, one process per shape (the library reads PBSED_* once): tools/gru_scan_prof.py.
# Round 4 used this to A / B switches that were removed with the slower forms they selected (XCD-local exchange, 4 contraction
# waves, bf16x3 at H = 512); what remains is the per-shape run, plus any `VAR=value` given on the command line as a variant.
# Usage (GPU box): tools/gru_scan_variants.sh [VAR=value ...] > gpurun_out/gru_variants.txt
cd "$(dirname "$0")/.."
run() { echo "=== $*"; env "$@" python tools/gru_scan_prof.py $EXTRA 2>&1 | grep -v "Warning\|amdgpu.ids"; }
for shape in c2 c3 deep; do
  EXTRA="--shape $shape --no-prof" run X=0
  for v in "$@"; do EXTRA="--shape $shape --no-prof" run "$v"; done
done
EXTRA="--shape c3 --no-prof --precision bf16" run X=0
EXTRA="--shape c2 --block 2" run X=0
