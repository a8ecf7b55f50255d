// TEST INFRASTRUCTURE: pb_sed_amd/csrc/misc.hip (losses, batch-norm backward, optimiser, layout kernels) compiled for the HOST against tests/emu/shim (see emu_conv_s16.cpp).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdlib>

namespace pbsed {
alignas(16) thread_local float sh[16 * 1024];
}
#include "misc.hip"

namespace pbsed {
static char g_err[512];
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int check_launch(const char*) { return 0; }
int device_cus() { return 4; }
float* scratch_for(hipStream_t, size_t) { return nullptr; }
float* scratch_zeroed_front(hipStream_t, size_t, size_t) { return nullptr; }
// (conv.hip's tile selection, used by the weight packers of misc.hip: not part of this unit's tests)
void conv_fwd_tile_dims(int, int, int, int, int*, int*) { abort(); }
}  // namespace pbsed
extern "C" const char* emu_last_error() { return pbsed::g_err; }
