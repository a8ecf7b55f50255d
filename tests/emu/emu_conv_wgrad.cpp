// TEST INFRASTRUCTURE: pb_sed_amd/csrc/conv_wgrad.hip (every conv weight-gradient kernel) compiled for the HOST against tests/emu/shim
// (see emu_conv_s16.cpp); the library's entry points for it live in api.hip, so this unit exports one of its own.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdlib>
#include <cstring>

namespace pbsed {
alignas(16) thread_local float smem[40 * 1024];
}
#include "conv_wgrad.hip"

namespace pbsed {
static char g_err[512];
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int check_launch(const char*) { return 0; }
static int g_cus = 2;
int device_cus() { return g_cus; }
int launch_cus() { return g_cus; }                              // emu_set_cus: a larger device = a wider split of the launch
static float* g_scratch = nullptr;
static size_t g_scratch_n = 0;
float* scratch_for(hipStream_t, size_t floats) {                  // zeroed on every growth; the slotted launches keep their front zero
    if (floats > g_scratch_n) {
        free(g_scratch);
        g_scratch = (float*)calloc(floats, sizeof(float));
        g_scratch_n = floats;
    }
    return g_scratch;
}
float* scratch_zeroed_front(hipStream_t s, size_t front, size_t total) {
    float* p = scratch_for(s, total > front ? total : front);
    return p;
}
}  // namespace pbsed
extern "C" const char* emu_last_error() { return pbsed::g_err; }
extern "C" void emu_set_cus(int n) { pbsed::g_cus = n; }
// api.hip::pbsed_conv_bwd_weight / _bf16 in one: dw (+=), db (+=)
extern "C" int emu_conv_bwd_weight(const float* x, const float* scale, const float* shift, int relu, const int* seq_len, const float* g,
                                   const unsigned char* unpool_idx, float* dw, float* db, int B, int Cin, int Cout, int F, int T, int KH,
                                   int KW, int bf16) {
    pbsed::ConvWgradArgs a{};
    a.x = x; a.scale = scale; a.shift = shift; a.seq_len = seq_len; a.g = g; a.unpool_idx = unpool_idx; a.dw = dw; a.db = db;
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.F = F; a.T = T; a.relu = relu; a.bf16 = bf16;
    return pbsed::conv_wgrad_launch(a, KH, KW, nullptr);
}
