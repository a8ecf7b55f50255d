// TEST INFRASTRUCTURE: pb_sed_amd/csrc/tm_gemm.hip and gru_wgrad.hip (the time-major products around the scans) compiled for the HOST
// against tests/emu/shim (see emu_conv_s16.cpp).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdlib>

#include "common.h"
namespace pbsed {
alignas(16) thread_local unsigned short tg_smem[80 * 1024];
alignas(16) thread_local u32x4_t smem_b16[10 * 1024];
}
#include "tm_gemm.hip"
#include "gru_wgrad.hip"

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
int launch_cus() { return device_cus(); }
static float* g_scratch = nullptr;
static size_t g_scratch_n = 0;
float* scratch_for(hipStream_t, size_t floats) {
    if (floats > g_scratch_n) {
        free(g_scratch);
        g_scratch = (float*)calloc(floats, sizeof(float));
        g_scratch_n = floats;
    }
    return g_scratch;
}
float* scratch_zeroed_front(hipStream_t s, size_t front, size_t total) { return scratch_for(s, total > front ? total : front); }
}  // namespace pbsed
extern "C" const char* emu_last_error() { return pbsed::g_err; }
