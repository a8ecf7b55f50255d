#!/bin/bash
# What waits for a GPU (round 5, pool closed to this build): ONE gpurun call measures the parked patches and runs the parity tests on them.
#   gpurun --timeout 2700 -- tools/r05_queue.sh      (~62 bench runs of 15 - 30 s + the GPU suite: ~35 min)
mkdir -p gpurun_out
V=tools/variants
tools/ab_lib.sh "c2 c5 c3" base=- scalar=$V/libpbsed_scalar.so hoist=$V/libpbsed_scalar_hoist.so s16=$V/libpbsed_s16.so s16c=$V/libpbsed_s16c.so lm=$V/libpbsed_lm.so wxe=$V/libpbsed_wxe.so era=$V/libpbsed_era.so all=$V/libpbsed_all.so 2>&1 | tee gpurun_out/r05_ab_scalar_hoist.txt
tools/ab_lib.sh "deep" base=- s16=$V/libpbsed_s16.so res=$V/libpbsed_res.so all=$V/libpbsed_all.so 2>&1 | tee gpurun_out/r05_ab_deep.txt
PBSED_TEST_UNMEASURED=1 timeout 300 python -m pytest tests/test_gpu_model.py -m gpu -q -k beside_the_bptt_scans 2>&1 | tail -3 | tee gpurun_out/r05_side_wgrad_test.txt
tools/ab_bench.sh PBSED_SIDE_WGRAD c2 2>&1 | tee gpurun_out/r05_ab_side_wgrad.txt; tools/ab_bench.sh PBSED_SIDE_WGRAD c3 2>&1 | tee -a gpurun_out/r05_ab_side_wgrad.txt        # host-side switch (engine.SIDE_WGRAD): the heads' weight gradients beside the BPTT scan
PBSED_LIB=$(realpath $V/libpbsed_all.so) timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r05_variant_gpu_tests.txt
