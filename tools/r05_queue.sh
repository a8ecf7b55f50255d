#!/bin/bash
# What waits for a GPU (round 5, pool closed to this build): ONE gpurun call measures the parked patches and runs the parity tests on them.
#   gpurun --timeout 1500 -- tools/r05_queue.sh
mkdir -p gpurun_out
V=tools/variants
tools/ab_lib.sh "c2 c5 c3" base=- scalar=$V/libpbsed_scalar.so hoist=$V/libpbsed_scalar_hoist.so all=$V/libpbsed_all.so 2>&1 | tee gpurun_out/r05_ab_scalar_hoist.txt
PBSED_LIB=$(realpath $V/libpbsed_all.so) timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r05_variant_gpu_tests.txt
