#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer pass over the HOST side of the C-ABI library (no GPU needed): builds the library with -fsanitize=address
# (device code untouched: -fno-gpu-sanitize) into a scratch directory and drives every entry point through its argument checks,
# argument structs, pointer tables and error strings (tools/asan_host_drive.py).  Exit 0 = no report.
#   tools/asan_host_check.sh [scratch_dir] [asan|tsan]      (tsan: -fsanitize=thread, four driver threads)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=${1:-$(mktemp -d)}
MODE=${2:-asan}
if [ $MODE = tsan ]; then SAN="-fsanitize=thread"; RT=tsan; export PBSED_DRIVE_THREADS=4; export TSAN_OPTIONS="report_signal_unsafe=0 halt_on_error=0"
else SAN="-fsanitize=address,undefined -fno-sanitize=vptr"; RT=asan; export ASAN_OPTIONS=detect_leaks=0; fi
ASAN=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.$RT-x86_64.so 2>/dev/null | head -1)
[ -n "$ASAN" ] || { echo "no libclang_rt.asan in this image"; exit 77; }
mkdir -p $W/build
cd $ROOT/pb_sed_amd/csrc
FLAGS="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC $SAN -fno-gpu-sanitize -fno-omit-frame-pointer -Wno-unused-result -Wno-unused-value"
pids=()
for f in api conv conv_bf16 conv_wino conv_winox3 conv_s16 conv1d_pc conv_wgrad gru gru_stack gru_wgrad tm_gemm misc logmel postproc collective; do
  /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o $W/build/$f.o 2> $W/build/$f.err &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC ${SAN%% -fno-sanitize=vptr} -fno-gpu-sanitize $W/build/*.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib -o $W/libpbsed_asan.so
cd $ROOT
LD_PRELOAD=$ASAN PBSED_LIB=$W/libpbsed_asan.so python tools/asan_host_drive.py > $W/drive.log 2>&1 || true
tail -1 $W/drive.log
if grep -q "ERROR: AddressSanitizer\|runtime error:\|WARNING: ThreadSanitizer" $W/drive.log || ! grep -q "^DONE" $W/drive.log; then
  grep -B2 -A25 "ERROR: AddressSanitizer\|WARNING: ThreadSanitizer\|runtime error:" $W/drive.log | head -80
  grep "^CALL" $W/drive.log | tail -1
  exit 1
fi
