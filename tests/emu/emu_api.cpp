// TEST INFRASTRUCTURE: pb_sed_amd/csrc/api.hip (error reporting, scratch registry, the conv entry points) with conv.hip (fp32 tiles),
// conv_wgrad.hip and misc.hip compiled for the HOST against tests/emu/shim (see emu_conv_s16.cpp).  The library's own set_error /
// scratch_for / device_cus run here (over the shim's malloc-backed hipMalloc); tests/emu/cpu_device.py serves the whole C-ABI from this
// unit and the other emu_*.cpp units.
#include <hip/hip_runtime.h>

namespace pbsed {
alignas(16) static thread_local float smem[40 * 1024];
alignas(16) static thread_local float sh[16 * 1024];
}
#include "api.hip"
#include "conv.hip"
#include "conv_wgrad.hip"
#include "misc.hip"
