#!/bin/bash
# Collect the rocprofv3 evidence for profiles/ on the GPU box (run through gpurun from the repo root):
#   1. kernel trace + stats of the default bench command (per-kernel durations)
#   2./3. PMC passes (FETCH_SIZE, WRITE_SIZE) in their own runs, counters only (MI355X_MICROARCH.md HBM section).
# The PMC passes use the launch-per-step GRU scans: counter collection serialises dispatches and the persistent
# scans are not what the HBM-traffic numbers are for.
set -x
REPO=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_stats -- \
    python $REPO/bench.py --no-cpu-baseline --steps 5 --warmup 3 > $REPO/gpurun_out/prof_stats_bench.json 2> $REPO/gpurun_out/prof_stats.log
PBSED_GRU_PERSIST=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $REPO/gpurun_out/prof_fetch -- \
    python $REPO/bench.py --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2> $REPO/gpurun_out/prof_fetch.log
PBSED_GRU_PERSIST=0 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $REPO/gpurun_out/prof_write -- \
    python $REPO/bench.py --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2> $REPO/gpurun_out/prof_write.log
cd $REPO
ls gpurun_out/prof_stats/* | head; tail -2 gpurun_out/prof_stats.log
