#!/bin/bash
# Builds of the library with the parked patches of tools/micro/attic applied, for tools/ab_lib.sh (PBSED_LIB):
#   tools/variants/libpbsed_base.so          the tree as it is
#   tools/variants/libpbsed_scalar.so        + scalar_tile_loads.patch
#   tools/variants/libpbsed_scalar_hoist.so  + epilogue_hoist.patch on top
#   tools/variants/libpbsed_s16.so           + s16_branch_free_loads.patch on top
#   tools/variants/libpbsed_res.so           + residual_group_loads.patch on top
#   tools/variants/libpbsed_s16c.so          + s16_constants_once.patch on top
#   tools/variants/libpbsed_lm.so            + logmel_setup_one_round_trip.patch on top
#   tools/variants/libpbsed_wxe.so           + winox3_dgrad_epilogue_pipeline.patch on top
#   tools/variants/libpbsed_era.so           + epilogue_requests_ahead.patch on top
#   tools/variants/libpbsed_all.so           + winox3_seq_len_once_per_tile.patch on top
# (the .so files are git-ignored and travel to the GPU box with the snapshot)
set -e
cd "$(dirname "$0")/.."
ROOT=$(pwd)
mkdir -p tools/variants
W=$(mktemp -d)
mkdir -p $W/pb_sed_amd
cp -r pb_sed_amd/csrc $W/pb_sed_amd/csrc
rm -rf $W/pb_sed_amd/csrc/build
build() { (cd $W/pb_sed_amd/csrc && bash build.sh 2>&1 | grep "^built\|error:" || true); cp $W/pb_sed_amd/libpbsed_mi355.so $ROOT/tools/variants/$1; }
build libpbsed_base.so
(cd $W && patch -s -p1 < $ROOT/tools/micro/attic/scalar_tile_loads.patch)
build libpbsed_scalar.so
(cd $W && patch -s -p1 < $ROOT/tools/micro/attic/epilogue_hoist.patch)
build libpbsed_scalar_hoist.so
(cd $W && patch -s -p1 < $ROOT/tools/micro/attic/s16_branch_free_loads.patch)
build libpbsed_s16.so
(cd $W && patch -s -p1 < $ROOT/tools/micro/attic/residual_group_loads.patch)
build libpbsed_res.so
(cd $W && patch -s -p1 < $ROOT/tools/micro/attic/s16_constants_once.patch)
build libpbsed_s16c.so
(cd $W && patch -s -p1 < $ROOT/tools/micro/attic/logmel_setup_one_round_trip.patch)
build libpbsed_lm.so
(cd $W && patch -s -p1 < $ROOT/tools/micro/attic/winox3_dgrad_epilogue_pipeline.patch)
build libpbsed_wxe.so
(cd $W && patch -s -p1 < $ROOT/tools/micro/attic/epilogue_requests_ahead.patch)
build libpbsed_era.so
(cd $W && patch -s -p1 < $ROOT/tools/micro/attic/winox3_seq_len_once_per_tile.patch)
(cd $W && patch -s -p1 < $ROOT/tools/micro/attic/comment_wording.patch)      # comments only
build libpbsed_all.so
rm -rf $W
md5sum tools/variants/*.so
