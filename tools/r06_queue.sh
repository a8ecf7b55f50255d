#!/bin/bash
# What waits for a GPU (round 6: the pool was closed to this build again).  Three gpurun calls, most important first:
#   gpurun --timeout 1500 -- tools/r06_queue.sh 1     HEAD on the hardware record: the GPU suite, the default bench line, smoke()
#   gpurun --timeout 2400 -- tools/r06_queue.sh 2     the rocprofv3 set of profiles/ (tools/run_profiles.sh) for HEAD
#   gpurun --timeout 2700 -- tools/r06_queue.sh 3     A/B of everything parked: the patch stack (tools/micro/attic), SIDE_WGRAD,
#                                                     (now with its CU budget), PBSED_WGRAD_XCD_COLS, PBSED_FUSE_BN_BWD
# Everything lands under gpurun_out/r06_*; copy what is to be judged into profiles/.
mkdir -p gpurun_out
stage=${1:-1}
if [ "$stage" = 1 ]; then
  (timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -15) | tee gpurun_out/r06_gputest_head.txt
  cp gpurun_out/parity.jsonl gpurun_out/r06_parity.jsonl 2>/dev/null
  timeout 900 python bench.py > gpurun_out/r06_bench_head.json 2> gpurun_out/r06_bench_head.err; echo "bench rc=$?"
  tail -c 2500 gpurun_out/r06_bench_head.json
  timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 | tee gpurun_out/r06_smoke.txt
  # the placement probe's verdict on this device and the BPTT scan with / without the XCD-local exchange
  tools/ab_bench.sh PBSED_GRU_XCD_LOCAL c2 2>&1 | tee gpurun_out/r06_ab_xcd_local.txt
elif [ "$stage" = 2 ]; then
  bash tools/run_profiles.sh 2>&1 | tail -20
elif [ "$stage" = 3 ]; then
  V=tools/variants
  PBSED_TEST_UNMEASURED=1 timeout 300 python -m pytest tests/test_gpu_model.py -m gpu -q -k beside_the_bptt_scans 2>&1 | tail -3 | tee gpurun_out/r06_side_wgrad_test.txt
  tools/ab_bench.sh PBSED_WGRAD_XCD_COLS c2 2>&1 | tee gpurun_out/r06_ab_wgrad_xcd_cols.txt
  tools/ab_bench.sh PBSED_SIDE_WGRAD c2 2>&1 | tee gpurun_out/r06_ab_side_wgrad.txt; tools/ab_bench.sh PBSED_SIDE_WGRAD c3 2>&1 | tee -a gpurun_out/r06_ab_side_wgrad.txt
  tools/ab_bench.sh PBSED_FUSE_BN_BWD c2 2>&1 | tee gpurun_out/r06_ab_fuse_bn_bwd.txt
  tools/ab_lib.sh "c2 c5 c3" base=- scalar=$V/libpbsed_scalar.so hoist=$V/libpbsed_scalar_hoist.so s16=$V/libpbsed_s16.so s16c=$V/libpbsed_s16c.so lm=$V/libpbsed_lm.so wxe=$V/libpbsed_wxe.so era=$V/libpbsed_era.so all=$V/libpbsed_all.so 2>&1 | tee gpurun_out/r06_ab_patch_stack.txt
  tools/ab_lib.sh "deep" base=- s16=$V/libpbsed_s16.so res=$V/libpbsed_res.so all=$V/libpbsed_all.so 2>&1 | tee gpurun_out/r06_ab_deep.txt
  PBSED_LIB=$(realpath $V/libpbsed_all.so) timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r06_variant_gpu_tests.txt
  # PMC: HBM bytes of the 128->128 weight gradient with / without the XCD-contiguous column walk (own passes, counters only)
  export TMPDIR=/tmp; REPO=$(pwd); cd /tmp
  for v in 0 1; do
    PBSED_WGRAD_XCD_COLS=$v PBSED_GRU_PERSIST=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $REPO/gpurun_out/r06_fetch_xcd_cols_$v -- \
        python $REPO/bench.py --headline-only --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2> $REPO/gpurun_out/r06_fetch_xcd_cols_$v.log
  done
fi
