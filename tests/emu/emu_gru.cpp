// TEST INFRASTRUCTURE: pb_sed_amd/csrc/gru.hip (layout kernels, the per-layer scans) compiled for the HOST against tests/emu/shim
// (see emu_conv_s16.cpp).
#include <hip/hip_runtime.h>

#include <cstdarg>

#include "gru.hip"

namespace pbsed {
static char g_err[512];
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int check_launch(const char*) { return 0; }
int device_cus() { return 2; }
float* scratch_for(hipStream_t, size_t) { return nullptr; }
float* scratch_zeroed_front(hipStream_t, size_t, size_t) { return nullptr; }
}  // namespace pbsed
extern "C" const char* emu_last_error() { return pbsed::g_err; }
