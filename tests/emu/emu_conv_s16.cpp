// TEST INFRASTRUCTURE: pb_sed_amd/csrc/conv_s16.hip compiled for the HOST against tests/emu/shim (every HIP thread a fiber) - the same
// extern "C" entry points as the library's, taking HOST pointers.  Built twice by tests/test_emulated_kernels.py: from the tree
// and from the tree + the parked patches of tools/micro/attic.
#include <hip/hip_runtime.h>

#include <cstdarg>

namespace pbsed {
alignas(16) thread_local unsigned char smem_raw[160 * 1024];          // the kernels' `extern __shared__ unsigned char smem_raw[]` (one block at a time)
}
#include "conv_s16.hip"

namespace pbsed {
static char g_err[512];
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int check_launch(const char*) { return 0; }
int device_cus() { return 2; }                            // the persistent kernels then walk several tiles per block
float* scratch_for(hipStream_t, size_t) { return nullptr; }
float* scratch_zeroed_front(hipStream_t, size_t, size_t) { return nullptr; }
}  // namespace pbsed
extern "C" const char* emu_last_error() { return pbsed::g_err; }
