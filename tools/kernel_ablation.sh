#!/bin/bash
# Where does a producer / consumer kernel's time go?  Builds variants of the library with parts of ONE kernel compiled out
# (macro bits: 1 no producer global loads, 2 no producer transform / LDS stores, 4 no streamed operand (winox3: U),
# 8 no consumer LDS reads / MFMAs) into gpurun_out/abl/ and times them.  GPU box only:
#   bash tools/kernel_ablation.sh conv_winox3 WX_DBG   'NOERR=1 PRECS=winox3 ONLY=128x128 python tools/wino_bench.py'
#   bash tools/kernel_ablation.sh conv_wgrad  WGPC_DBG 'ONLY=128x128 python tools/gpu_conv_bench.py'
set -e
cd "$(dirname "$0")/.."
src=$1; macro=$2; cmd=$3
mkdir -p gpurun_out/abl
for d in ${DBGS:-0 1 2 3 8 11}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value -D$macro=$d \
    -c pb_sed_amd/csrc/$src.hip -o gpurun_out/abl/v$d.o 2>/dev/null &
done
wait
for d in ${DBGS:-0 1 2 3 8 11}; do
  objs=$(ls pb_sed_amd/csrc/build/*.o | grep -v "/$src.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs gpurun_out/abl/v$d.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib \
    -o gpurun_out/abl/lib$d.so
  echo "$macro=$d"
  PBSED_LIB=$PWD/gpurun_out/abl/lib$d.so bash -c "$cmd" 2>&1 | grep -E -- "->" | sed -e "s/fwd.*| wgrad/wgrad/"
done
