// C-ABI glue: error reporting + convolution entry points (see include/pbsed.h for the contract).
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <pthread.h>

#include "common.h"
#include "pbsed_internal.h"

namespace pbsed {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return PBSED_E_HIP;
    }
    return PBSED_OK;
}

// ---- per-DEVICE facts and per-(device, stream) scratch: nothing process-wide that a second device or stream could trip over
int device_cus() {
    static int cus[64] = {0};                        // indexed by device ordinal; written once with the same value by whoever is first
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    int& c = cus[dev & 63];
    if (c == 0) {
        hipDeviceProp_t prop;
        const int n = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
        c = n < 8 ? 8 : n;
    }
    return c;
}

// Grid-sizing budget of the persistent weight-gradient launchers (pbsed_set_launch_cus): a launch that is meant to run BESIDE a
// persistent scan must not be cut for the whole device - its grid would be static shares of a device it does not get, and the
// blocks that find no CU wait for the scan to end (the launch then takes scan + its own time, as if it had not been moved).
static std::atomic<int> g_launch_cus{0};
int launch_cus() {
    const int n = device_cus(), budget = g_launch_cus.load(std::memory_order_relaxed);
    return budget > 0 && budget < n ? budget : n;
}

namespace {
struct ScratchSlot {
    int dev;
    hipStream_t stream;
    float* ptr;
    size_t floats;
    bool owned;            // library-owned fallback (per device, stream = any)
    size_t clean;          // floats at the front that are known to be zero (scratch_zeroed_front)
};
constexpr int kScratchSlots = 64;
ScratchSlot g_scratch[kScratchSlots] = {};
int g_scratch_n = 0;
pthread_mutex_t g_scratch_mu = PTHREAD_MUTEX_INITIALIZER;
}  // namespace

// Scratch for the partial-sum slots of the weight-gradient kernels on (current device, stream): the caller's registered
// buffer (pbsed_set_scratch) when it is large enough, otherwise a library-owned buffer per DEVICE, grown on demand (then
// calls for that device must come from one stream at a time - the documented default).
float* scratch_for(hipStream_t stream, size_t floats) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    pthread_mutex_lock(&g_scratch_mu);
    ScratchSlot* own = nullptr;
    float* out = nullptr;
    for (int i = 0; i < g_scratch_n; ++i) {
        ScratchSlot& e = g_scratch[i];
        if (e.dev != dev) continue;
        if (!e.owned && e.stream == stream && e.floats >= floats) { out = e.ptr; break; }
        if (e.owned) own = &e;
    }
    if (!out) {
        if (!own && g_scratch_n < kScratchSlots) { own = &g_scratch[g_scratch_n++]; *own = ScratchSlot{dev, nullptr, nullptr, 0, true, 0}; }
        if (own) {
            if (own->floats < floats) {
                if (own->ptr) { (void)hipDeviceSynchronize(); (void)hipFree(own->ptr); own->ptr = nullptr; own->floats = 0; own->clean = 0; }
                if (hipMalloc(&own->ptr, floats * sizeof(float)) == hipSuccess) own->floats = floats;
            }
            out = own->floats >= floats ? own->ptr : nullptr;
        }
    }
    pthread_mutex_unlock(&g_scratch_mu);
    return out;
}

// The first `front` floats of that scratch, ZERO on return and kept zero by their users (the slotted weight-gradient launches add
// into them and wgrad_slot_reduce_kernel writes zeros back behind its reads): filled once per registration / allocation, in
// stream order, instead of once per launch.  Everything behind `front` is free-for-all (scratch_for(..) + front).
float* scratch_zeroed_front(hipStream_t stream, size_t front, size_t total) {
    float* base = scratch_for(stream, total);
    if (!base) return nullptr;
    int dev = 0;
    (void)hipGetDevice(&dev);
    bool fill = false;
    pthread_mutex_lock(&g_scratch_mu);
    for (int i = 0; i < g_scratch_n; ++i) {
        ScratchSlot& e = g_scratch[i];
        if (e.dev == dev && e.ptr == base && e.clean < front) { e.clean = front; fill = true; }
    }
    pthread_mutex_unlock(&g_scratch_mu);
    if (fill && hipMemsetAsync(base, 0, front * sizeof(float), stream) != hipSuccess) return nullptr;
    return base;
}

}  // namespace pbsed

using namespace pbsed;

extern "C" {

const char* pbsed_last_error(void) { return g_err; }

// Caller-owned scratch for (current device, `stream`): the weight-gradient kernels (conv and GRU) put their partial-sum
// slots there instead of into the library's per-device buffer, so several streams of one device can run them concurrently.
// scratch = NULL removes the registration.  pbsed_scratch_bytes() is large enough for every launch of the reference nets.
size_t pbsed_scratch_bytes(void) { return (size_t)160 << 20; }

// CU budget of the weight-gradient launchers (launch_cus above); 0 = the whole device.  Returns the previous budget.
int pbsed_set_launch_cus(int cus) {
    return g_launch_cus.exchange(cus > 0 ? cus : 0, std::memory_order_relaxed);
}

int pbsed_set_scratch(void* scratch, size_t bytes, void* stream) {
    int dev = 0;
    PBSED_HIP_TRY(hipGetDevice(&dev), "hipGetDevice");
    pthread_mutex_lock(&g_scratch_mu);
    int hit = -1;
    for (int i = 0; i < g_scratch_n; ++i)
        if (!g_scratch[i].owned && g_scratch[i].dev == dev && g_scratch[i].stream == (hipStream_t)stream) hit = i;
    int rc = PBSED_OK;
    if (hit < 0 && scratch) {
        for (int i = 0; i < g_scratch_n && hit < 0; ++i)          // a withdrawn registration (scratch = NULL) frees its entry
            if (!g_scratch[i].owned && g_scratch[i].ptr == nullptr) hit = i;
        if (hit >= 0) {}
        else if (g_scratch_n < kScratchSlots) hit = g_scratch_n++;
        else { set_error("pbsed_set_scratch: registration table full (%d)", kScratchSlots); rc = PBSED_E_ARG; }
    }
    if (hit >= 0) g_scratch[hit] = ScratchSlot{dev, (hipStream_t)stream, scratch ? (float*)scratch : nullptr, scratch ? bytes / sizeof(float) : 0, false, 0};
    pthread_mutex_unlock(&g_scratch_mu);
    return rc;
}

int pbsed_version(void) { return 1; }

int pbsed_conv_fwd(const float* x, const float* w_packed, const float* bias, const float* scale,
                   const float* shift, int relu, const int* seq_len, float* y, unsigned char* pool_idx,
                   double* stats, int stats_per_cf, int B, int Cin, int Cout, int F, int T, int KH, int KW,
                   int pool, void* stream) {
    ConvFwdArgs a{};
    a.x = x; a.wp = w_packed; a.bias = bias; a.scale = scale; a.shift = shift; a.seq_len = seq_len;
    a.y = y; a.pool_idx = pool_idx; a.stats = stats; a.stats_cf = stats_per_cf; a.relu = relu;
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.F = F; a.T = T;
    int ck, ct;
    conv_fwd_tile_dims(KH, KW, Cin, Cout, &ck, &ct);
    a.CinP = (Cin + ck - 1) / ck * ck;
    a.CoutP = (Cout + ct - 1) / ct * ct;
    return conv_fwd_launch(a, KH, KW, pool, 0, (hipStream_t)stream);
}

// pbsed_conv_fwd + a residual connection ending at this layer: `residual` [B, Cout, Fo, T] is added to the (biased,
// pooled) output before it is stored and before the statistics are taken, so the next layer's batch norm sees the sum.
int pbsed_conv_fwd_res(const float* x, const float* w_packed, const float* bias, const float* scale,
                       const float* shift, int relu, const int* seq_len, float* y, unsigned char* pool_idx,
                       double* stats, int stats_per_cf, int B, int Cin, int Cout, int F, int T, int KH, int KW,
                       int pool, const float* residual, void* stream) {
    ConvFwdArgs a{};
    a.x = x; a.wp = w_packed; a.bias = bias; a.scale = scale; a.shift = shift; a.seq_len = seq_len;
    a.y = y; a.pool_idx = pool_idx; a.stats = stats; a.stats_cf = stats_per_cf; a.relu = relu; a.res = residual;
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.F = F; a.T = T;
    int ck, ct;
    conv_fwd_tile_dims(KH, KW, Cin, Cout, &ck, &ct);
    a.CinP = (Cin + ck - 1) / ck * ck;
    a.CoutP = (Cout + ct - 1) / ct * ct;
    return conv_fwd_launch(a, KH, KW, pool, 0, (hipStream_t)stream);
}

// Data gradient.  g: grad wrt the forward conv's output [B,Cout,Fo,T] (pooled if unpool_idx);
// wd_packed: pack_conv_weights(dgrad=1).  Output dz [B,Cin,F,T]: if bx != null the result is already
// pushed back through mask -> ReLU -> BN-apply of the layer's prologue (dz wrt BN output) and
// stats [Cin][2] accumulates (sum dz, sum dz*xhat); otherwise plain grad wrt the conv input.
int pbsed_conv_bwd_data(const float* g, const float* wd_packed, const unsigned char* unpool_idx,
                        const int* seq_len, float* dz, const float* bx, const float* bmean,
                        const float* binvstd, const float* bscale, const float* bshift, int relu,
                        double* stats, int B, int Cin, int Cout, int F, int T, int KH, int KW,
                        void* stream) {
    ConvFwdArgs a{};
    a.x = g; a.wp = wd_packed; a.seq_len = seq_len; a.y = dz; a.unpool_idx = unpool_idx;
    a.bx = bx; a.bmean = bmean; a.binvstd = binvstd; a.bscale = bscale; a.bshift = bshift;
    a.relu = relu; a.stats = bx ? stats : nullptr;
    a.B = B; a.Cin = Cout; a.Cout = Cin; a.F = F; a.T = T;      // roles swapped
    int ck, ct;
    conv_fwd_tile_dims(KH, KW, Cout, Cin, &ck, &ct);
    a.CinP = (Cout + ck - 1) / ck * ck;
    a.CoutP = (Cin + ct - 1) / ct * ct;
    return conv_fwd_launch(a, KH, KW, 0, 1, (hipStream_t)stream);
}

int pbsed_conv_bwd_weight(const float* x, const float* scale, const float* shift, int relu,
                          const int* seq_len, const float* g, const unsigned char* unpool_idx, float* dw,
                          float* db, int B, int Cin, int Cout, int F, int T, int KH, int KW,
                          void* stream) {
    ConvWgradArgs a{};
    a.x = x; a.scale = scale; a.shift = shift; a.relu = relu; a.seq_len = seq_len; a.g = g;
    a.unpool_idx = unpool_idx; a.dw = dw; a.db = db;
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.F = F; a.T = T;
    return conv_wgrad_launch(a, KH, KW, (hipStream_t)stream);
}

// pbsed_conv_bwd_weight with the BN backward of the NEXT layer's input norm folded into the dY loader: `dz` is that norm's
// masked ReLU-backward gradient (what pbsed_conv_bwd_data wrote), `gx` the raw output of this conv (the norm's input), `coef`
// [3][Cout * (per_cf ? Fo : 1)] from pbsed_bn_bwd_coef, `gseq` the sequence lengths the norm masks with (null = T).  The
// kernel multiplies with dY = k1 dz + k2 gx + k3 (0 beyond the sequence) and writes it to `gout` (same shape as dz) for the
// layer's data gradient - the stand-alone pbsed_bn_bwd pass over the tensor is not needed.  Shapes without such a kernel
// return PBSED_E_UNSUPPORTED (ask pbsed_conv_bwd_weight_bng_supported first).
int pbsed_conv_bwd_weight_bng(const float* x, const float* scale, const float* shift, int relu, const int* seq_len,
                              const float* dz, const float* gx, const float* coef, int per_cf, const int* gseq, float* gout,
                              const unsigned char* unpool_idx, float* dw, float* db, int B, int Cin, int Cout, int F, int T,
                              int KH, int KW, void* stream) {
    if (!dz || !gx || !coef || !gout) { set_error("conv_bwd_weight_bng: dz, gx, coef and gout are required"); return PBSED_E_ARG; }
    ConvWgradArgs a{};
    a.x = x; a.scale = scale; a.shift = shift; a.relu = relu; a.seq_len = seq_len; a.g = dz;
    a.unpool_idx = unpool_idx; a.dw = dw; a.db = db;
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.F = F; a.T = T;
    a.gx = gx; a.gcoef = coef; a.gseq = gseq; a.gout = gout; a.g_cf = per_cf;
    return conv_wgrad_launch(a, KH, KW, (hipStream_t)stream);
}

int pbsed_conv_bwd_weight_bng_supported(int KH, int KW, int Cin, int Cout, int F, int T, int per_cf) {
    return conv_wgrad_bng_supported(KH, KW, Cin, Cout, F, T, 0, per_cf) ? 1 : 0;
}

// Same contract with bf16-MFMA operands (x after its prologue and dY are rounded to bf16 while staged; products are
// accumulated, reduced and returned in fp32).  Layers with fewer than 32 input or output channels run the fp32 kernels.
int pbsed_conv_bwd_weight_bf16(const float* x, const float* scale, const float* shift, int relu,
                               const int* seq_len, const float* g, const unsigned char* unpool_idx, float* dw,
                               float* db, int B, int Cin, int Cout, int F, int T, int KH, int KW, void* stream) {
    ConvWgradArgs a{};
    a.x = x; a.scale = scale; a.shift = shift; a.relu = relu; a.seq_len = seq_len; a.g = g;
    a.unpool_idx = unpool_idx; a.dw = dw; a.db = db;
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.F = F; a.T = T; a.bf16 = 1;
    return conv_wgrad_launch(a, KH, KW, (hipStream_t)stream);
}

int pbsed_memset_async(void* p, int value, size_t bytes, void* stream) {
    hipError_t e = hipMemsetAsync(p, value, bytes, (hipStream_t)stream);
    if (e != hipSuccess) { set_error("memset: %s", hipGetErrorString(e)); return PBSED_E_HIP; }
    return PBSED_OK;
}

}  // extern "C"
