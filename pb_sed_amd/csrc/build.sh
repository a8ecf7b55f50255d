#!/bin/bash
# Build libpbsed_mi355.so for gfx950 (cross-compiles without a GPU).  Usage: build.sh [-j]
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value"
mkdir -p build
pids=()
for f in api conv conv_bf16 conv_wino conv_winox3 conv_s16 conv1d_pc conv_wgrad gru gru_stack gru_wgrad tm_gemm misc logmel postproc collective; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ common.h -nt build/$f.o ] || \
     [ pbsed_internal.h -nt build/$f.o ] || [ fft512.h -nt build/$f.o ] || [ conv_epilogue.h -nt build/$f.o ] || [ pack_elems.h -nt build/$f.o ] || [ conv_wgrad_s16.h -nt build/$f.o ] || \
     [ gru_granule_map.h -nt build/$f.o ] || [ gru_granule_role.inc -nt build/$f.o ]; then
    $HIPCC $FLAGS -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC build/*.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib -o ../libpbsed_mi355.so
echo "built $(cd .. && pwd)/libpbsed_mi355.so"
