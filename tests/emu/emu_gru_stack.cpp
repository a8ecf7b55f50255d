// TEST INFRASTRUCTURE: pb_sed_amd/csrc/gru_stack.hip (the persistent GRU scans) compiled for the HOST against tests/emu/shim.  The scans'
// workgroups hand their states to each other through memory while they run, so this unit is driven in the shim's CONCURRENT mode
// (hipemu_set_concurrent(1): every workgroup an OS thread of its own, its threads fibers): the tagged-word protocol - publish,
// poll, parity, pacing, time-out - runs for real, under the host's memory model (x86 TSO is stronger than the GPU's; what this
// checks is the protocol's LOGIC and the arithmetic, section 6a of DESIGN.md checks the index maps exhaustively).
#include <hip/hip_runtime.h>

#include <cstdarg>

namespace pbsed {
alignas(16) thread_local float red_dyn[64 * 1024];       // the forward kernels' `extern __shared__ float red_dyn[]`
}
#include "gru_stack.hip"

namespace pbsed {
static char g_err[512];
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int check_launch(const char*) { return 0; }
int device_cus() { return 256; }                          // the launchers' co-residency checks are those of an MI355X
float* scratch_for(hipStream_t, size_t) { return nullptr; }
float* scratch_zeroed_front(hipStream_t, size_t, size_t) { return nullptr; }
}  // namespace pbsed
extern "C" const char* emu_last_error() { return pbsed::g_err; }
