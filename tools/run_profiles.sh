#!/bin/bash
# Collect the rocprofv3 evidence for profiles/ on the GPU box (run through gpurun from the repo root):
#   1. kernel trace + stats of the default bench command (c2) and of --config c3 / c5 (per-kernel durations)
#   2./3. PMC passes (FETCH_SIZE, WRITE_SIZE) of c2 in their own runs, counters only (MI355X_MICROARCH.md HBM section);
#         these use the launch-per-step GRU scans (the HBM numbers are for the conv kernels).
#   4. SQ / TCC counters WITH the persistent scans on (what the scan kernels themselves do): bounded by `timeout`, the
#      scans' own bounded spin turns a scan that cannot make progress under counter collection into an error, not a hang.
set -x
REPO=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_stats -- \
    python $REPO/bench.py --headline-only --no-cpu-baseline --steps 5 --warmup 3 > $REPO/gpurun_out/prof_stats_bench.json 2> $REPO/gpurun_out/prof_stats.log
for cfg in c3 c5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_stats_$cfg -- \
      python $REPO/bench.py --config $cfg --no-cpu-baseline --steps 5 --warmup 3 > $REPO/gpurun_out/prof_stats_bench_$cfg.json 2> $REPO/gpurun_out/prof_stats_$cfg.log
done
PBSED_GRU_PERSIST=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $REPO/gpurun_out/prof_fetch -- \
    python $REPO/bench.py --headline-only --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2> $REPO/gpurun_out/prof_fetch.log
PBSED_GRU_PERSIST=0 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $REPO/gpurun_out/prof_write -- \
    python $REPO/bench.py --headline-only --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2> $REPO/gpurun_out/prof_write.log
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv \
    -d $REPO/gpurun_out/prof_sq -- python $REPO/bench.py --headline-only --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2> $REPO/gpurun_out/prof_sq.log
echo "sq pass rc=$?" >> $REPO/gpurun_out/prof_sq.log
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv \
    -d $REPO/gpurun_out/prof_tcc -- python $REPO/bench.py --headline-only --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2> $REPO/gpurun_out/prof_tcc.log
echo "tcc pass rc=$?" >> $REPO/gpurun_out/prof_tcc.log
# 5. the bf16 pipe: part-product count, MFMA-busy cycles and total cycles of every kernel that forms its products from bf16
#    (or bf16x3) operands - the persistent scans, the GRU weight gradients, Conv1d, the time-major projections, conv_winox3,
#    conv_wgrad_pc (SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD, GRBM_GUI_ACTIVE the kernel's clocks x 8 XCDs)
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU --output-format csv \
    -d $REPO/gpurun_out/prof_mfma -- python $REPO/bench.py --headline-only --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2> $REPO/gpurun_out/prof_mfma.log
echo "mfma pass rc=$?" >> $REPO/gpurun_out/prof_mfma.log
# 6. the 'deep' net_config (f4): kernel trace + stats
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_stats_deep -- \
    python $REPO/bench.py --config deep --no-cpu-baseline --steps 3 --warmup 2 > $REPO/gpurun_out/prof_stats_bench_deep.json 2> $REPO/gpurun_out/prof_stats_deep.log
cd $REPO
ls gpurun_out/prof_stats/* | head; tail -2 gpurun_out/prof_stats.log; tail -3 gpurun_out/prof_sq.log; tail -3 gpurun_out/prof_tcc.log
