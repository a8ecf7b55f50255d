#!/bin/bash
# A race check for the DEVICE code's use of LDS / global memory inside a block (no GPU needed): the emulated kernel tests and the
# emulated training steps (tests/emu) under other fiber schedules - descending thread order, and two seeded shuffles re-drawn on every
# scheduler pass.  Every such order is a legal execution of the block; a missing barrier between a write and another wave's read
# shows up as a failed comparison under some order.  (Between BLOCKS: tests/test_scan_protocol.py and the concurrent mode of the
# scans' tests.)      tools/emu_schedules.sh [pytest args]        exit 0 = every schedule passes
set -e
cd "$(dirname "$0")/.."
for s in 1 2 3; do
  echo "== HIPEMU_SCHEDULE=$s"
  HIPEMU_SCHEDULE=$s python -m pytest ${@:-tests/test_emulated_kernels.py tests/test_emulated_model.py} -q -p no:cacheprovider 2>&1 | tail -3
  [ ${PIPESTATUS[0]} -eq 0 ] || exit 1
done
