#!/bin/bash
# AddressSanitizer over the DEVICE code (no GPU needed): the kernels of csrc/*.hip, executed on the CPU by the emulator of tests/emu
# (tests/emu/shim/hip/hip_runtime.h), are compiled with -fsanitize=address; every global load / store of a kernel that leaves its
# (host-allocated) tensor is then reported with the kernel's source line.  Raw-buffer accesses are range-checked by the shim itself (out of range
# = 0 / dropped, as the hardware does); this covers the plain pointer accesses.  Runs the emulated kernel tests and the emulated
# training steps.   tools/emu_asan.sh [pytest args]        exit 0 = no report
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
ASAN=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so 2>/dev/null | head -1)
[ -n "$ASAN" ] || { echo "no libclang_rt.asan in this image"; exit 77; }
cd $ROOT
LOG=$(mktemp)
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$ASAN PBSED_EMU_OPT="-O0 -g -fsanitize=address -fno-omit-frame-pointer" \
  python -m pytest ${@:-tests/test_emulated_kernels.py tests/test_emulated_model.py} -q -p no:cacheprovider 2>&1 | tee $LOG | tail -5
if grep -q "ERROR: AddressSanitizer" $LOG; then grep -A12 "ERROR: AddressSanitizer" $LOG | head -60; exit 1; fi
grep -q " passed" $LOG && ! grep -q " failed\| error" $LOG
