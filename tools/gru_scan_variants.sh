#!/bin/bash
# A / B of the persistent scans' switches, one process per variant (the library reads PBSED_* once): tools/gru_scan_prof.py
# Usage (GPU box): tools/gru_scan_variants.sh [shape] > gpurun_out/gru_variants.txt
shape=${1:-c2}
cd "$(dirname "$0")/.."
run() { echo "=== $*"; env "$@" python tools/gru_scan_prof.py --shape $shape $EXTRA 2>&1 | grep -v "Warning\|amdgpu.ids"; }
run PBSED_GRU_LOCAL=0
run PBSED_GRU_LOCAL=1
run PBSED_GRU_LOCAL=0 PBSED_GRU_DBG=1
run PBSED_GRU_LOCAL=1 PBSED_GRU_DBG=1
EXTRA=--no-prof run PBSED_GRU_LOCAL=0
EXTRA=--no-prof run PBSED_GRU_LOCAL=1
