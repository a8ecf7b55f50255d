#!/bin/bash
# A / B of the persistent scans' switches, one process per variant (the library reads PBSED_* once): tools/gru_scan_prof.py
# Usage (GPU box): tools/gru_scan_variants.sh > gpurun_out/gru_variants.txt
cd "$(dirname "$0")/.."
run() { echo "=== $*"; env "$@" python tools/gru_scan_prof.py $EXTRA 2>&1 | grep -v "Warning\|amdgpu.ids"; }
export PBSED_GRU_LOCAL=0
EXTRA="--shape c2 --no-prof" run X=0
EXTRA="--shape c2 --no-prof" run PBSED_GRU_NW4=3
EXTRA="--shape c2 --no-prof" run X=0
EXTRA="--shape c2 --block 2" run PBSED_GRU_NW4=3
EXTRA="--shape c3 --no-prof" run X=0
EXTRA="--shape c3 --no-prof" run PBSED_GRU_NW4=3
EXTRA="--shape deep --no-prof" run X=0
EXTRA="--shape deep --no-prof" run PBSED_GRU_X3_H512=3
EXTRA="--shape c3 --no-prof --precision bf16" run X=0
EXTRA="--shape c3 --no-prof --precision bf16" run PBSED_GRU_NW4=3
