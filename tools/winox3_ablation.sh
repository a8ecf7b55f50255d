#!/bin/bash
# Where does conv_winox3's time go?  Builds variants of the library with parts of the kernel compiled out (WX_DBG bits:
# 1 no producer global loads, 2 no producer transform / LDS stores, 4 no U stream, 8 no consumer LDS reads / MFMAs) into
# gpurun_out/wx/ and times them with tools/wino_bench.py.  Usage (GPU box): bash tools/winox3_ablation.sh [build|run]
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/wx
if [ "${1:-build}" = build ]; then
  for d in ${DBGS:-0 1 2 3 4 5 8 12}; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value -DWX_DBG=$d \
      -c pb_sed_amd/csrc/conv_winox3.hip -o gpurun_out/wx/w$d.o &
  done
  wait
  for d in ${DBGS:-0 1 2 3 4 5 8 12}; do
    objs=$(ls pb_sed_amd/csrc/build/*.o | grep -v conv_winox3.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs gpurun_out/wx/w$d.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib \
      -o gpurun_out/wx/lib$d.so
  done
else
  for d in ${DBGS:-0 1 2 3 4 5 8 12}; do
    echo "WX_DBG=$d"
    PBSED_LIB=$PWD/gpurun_out/wx/lib$d.so NOERR=1 PRECS=winox3 ONLY=${ONLY:-128x128,64x64} python tools/wino_bench.py 2>&1 | grep -v amdgpu.ids
  done
fi
