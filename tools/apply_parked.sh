#!/bin/bash
# Apply the parked kernel patches of tools/micro/attic to the TREE, in the stack order of tools/build_variants.sh, up to and including
# the patch named (or all of them), and rebuild:   tools/apply_parked.sh [last_patch_name|all]
# (they are textually stacked - a later one needs the earlier ones; each is an equivalent re-ordering / re-writing of loads, so
# applying one that does not pay costs nothing but its diff).  Afterwards: pytest -m gpu, tools/run_profiles.sh, then delete the
# applied patches from the attic and from build_variants.sh (tests/test_host_logic.py checks that the two agree).
set -e
cd "$(dirname "$0")/.."
LAST=${1:-all}
PATCHES=$(grep -o 'attic/[a-z0-9_]*\.patch' tools/build_variants.sh | sed 's#attic/##' | awk '!seen[$0]++')
for p in $PATCHES; do
  echo "applying $p"
  patch -s -p1 < tools/micro/attic/$p
  [ "${p%.patch}" = "${LAST%.patch}" ] && break
done
(cd pb_sed_amd/csrc && bash build.sh 2>&1 | grep "^built\|error:")
