// TEST INFRASTRUCTURE: a minimal HOST stand-in for <hip/hip_runtime.h> that lets a kernel translation unit of pb_sed_amd/csrc be compiled
// for x86 and EXECUTED on the CPU, lane by lane (tests/test_emulated_kernels.py).  Every HIP thread of a block is a fiber
// (a stack each, switched by hipemu_switch) of one OS thread; cross-lane instructions (MFMA, DPP, shuffles, readfirstlane) and __syncthreads are rendezvous
// points of the wave's / the block's fibers, so the semantics are those of the hardware as long as cross-lane operations sit in
// wave-uniform control flow (they do in these kernels).  Blocks run one after the other.  The amdgcn builtins the kernels call
// are ordinary functions here.  What this is for: functional equivalence of kernel CHANGES without a GPU (same emulator, two
// versions of a kernel, bit-identical outputs expected) and a sanity check against a plain float64 restatement - not timing, and
// not the last bits of MFMA accumulation (the emulator sums a product row in double and rounds once).
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sched.h>
#include <time.h>

#include <algorithm>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __shared__ thread_local      /* one instance per OS thread = per block (the fibers of a block share their thread) */
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define HIP_EMU 1

// ------------------------------------------------------------------------------------------------ types
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "emulated"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s_, size_t n, int, hipStream_t) { memcpy(d, s_, n); return hipSuccess; }
#ifndef HIPEMU_CUS
#define HIPEMU_CUS 8
#endif
struct hipDeviceProp_t { int multiProcessorCount = HIPEMU_CUS; };
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { *p = hipDeviceProp_t{}; return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
enum { hipMemcpyDeviceToHost = 2 };
template <class T> static inline hipError_t hipMalloc(T** p, size_t n) { *p = (T*)malloc(n); return *p ? hipSuccess : 2; }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
template <class K> static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, K, int, size_t) { *n = 1; return hipSuccess; }

// ------------------------------------------------------------------------------------------------ the fiber runtime
namespace hipemu {
constexpr int MAX_THREADS = 1024, WAVE = 64;
struct Wave {
    int arrived = 0;
    unsigned gen = 0;
    alignas(16) unsigned char slot[WAVE][64];       // exchange area: up to 64 bytes per lane and rendezvous
};
struct Runtime {
    void* sched_sp = nullptr;                        // saved stack pointers (hipemu_switch): the scheduler's and one per fiber
    void* sp[MAX_THREADS];
    // a fiber parked at a rendezvous names the generation word it waits on: the scheduler passes it over until that word moves
    // (instead of switching to it only for it to look and yield again)
    const volatile unsigned* wait_word[MAX_THREADS] = {nullptr};
    unsigned wait_val[MAX_THREADS];
    char* stack[MAX_THREADS] = {nullptr};
    bool done[MAX_THREADS];
    int cur = 0, nthreads = 0, active = 0;   // active: fibers of the block that have not returned (s_barrier counts live waves only)
    int blk_arrived = 0;
    unsigned blk_gen = 0;
    Wave wave[MAX_THREADS / WAVE];
    std::function<void()> body;
};
extern thread_local Runtime* tls_rt;
inline Runtime& rt() {
    if (!tls_rt) tls_rt = new Runtime();
    return *tls_rt;
}
struct Idx { unsigned x, y, z; };
}  // namespace hipemu
extern thread_local hipemu::Idx threadIdx, blockIdx, blockDim, gridDim;

// the context switch (hipemu_runtime.cpp): callee-saved registers onto the current stack, its pointer to *save, continue on `load`.
// (swapcontext costs a sigprocmask system call per switch - the emulator spent nearly all of its time there.)
extern "C" void hipemu_switch(void** save, void* load);
namespace hipemu {
inline void yield() { Runtime& r = rt(); hipemu_switch(&r.sp[r.cur], r.sched_sp); }
inline void park(const volatile unsigned* word, unsigned val) {
    Runtime& r = rt();
    const int me = r.cur;
    r.wait_word[me] = word;
    r.wait_val[me] = val;
    while (*word == val) hipemu_switch(&r.sp[me], r.sched_sp);
    r.wait_word[me] = nullptr;
}
// rendezvous of the 64 lanes of the calling fiber's wave (every lane of a wave must call it: wave-uniform control flow)
inline void wave_sync() {
    Runtime& r = rt();
    Wave& w = r.wave[r.cur / WAVE];
    const int lanes = std::min(WAVE, r.nthreads - (r.cur / WAVE) * WAVE);
    const unsigned g = w.gen;
    if (++w.arrived == lanes) { w.arrived = 0; ++w.gen; return; }
    park(&w.gen, g);
}
inline void block_sync() {
    Runtime& r = rt();
    const unsigned g = r.blk_gen;
    if (++r.blk_arrived >= r.active) { r.blk_arrived = 0; ++r.blk_gen; return; }
    park(&r.blk_gen, g);
}
inline int lane() { return rt().cur % WAVE; }
inline Wave& my_wave() { Runtime& r = rt(); return r.wave[r.cur / WAVE]; }
void trampoline();
// run `body` for every thread of every block of the grid: threads as fibers; blocks one after the other on the calling thread, or -
// hipemu_set_concurrent(1), for PERSISTENT kernels whose blocks talk to each other through memory - every block on an OS thread of
// its own, all at once (grids of at most 256 blocks)
void launch(dim3 grid, dim3 block, const std::function<void()>& body);
extern int g_concurrent;
}  // namespace hipemu

#define hipLaunchKernelGGL(kern, grid, block, lds, stream, ...) hipemu::launch((grid), (block), [=]() { kern(__VA_ARGS__); })
static inline void __syncthreads() { hipemu::block_sync(); }

// ------------------------------------------------------------------------------------------------ scalar helpers
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
static inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
#define __expf(x) expf(x)
static inline float rsqrtf(float x) { return 1.f / sqrtf(x); }
static inline double rsqrt(double x) { return 1. / sqrt(x); }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
static inline long long min(long long a, long long b) { return a < b ? a : b; }
static inline long long max(long long a, long long b) { return a > b ? a : b; }
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }
static inline double atomicAdd(double* p, double v) { const double o = *p; *p = o + v; return o; }      // (one OS thread: fibers)
static inline float atomicAdd(float* p, float v) { const float o = *p; *p = o + v; return o; }
static inline float unsafeAtomicAdd(float* p, float v) { return atomicAdd(p, v); }

// ------------------------------------------------------------------------------------------------ cross-lane operations
template <class T> static inline T __shfl(T v, int src) {
    hipemu::Wave& w = hipemu::my_wave();
    memcpy(w.slot[hipemu::lane()], &v, sizeof(T));
    hipemu::wave_sync();
    T r;
    memcpy(&r, w.slot[src & 63], sizeof(T));
    hipemu::wave_sync();
    return r;
}
template <class T> static inline T __shfl_xor(T v, int m) { return __shfl(v, hipemu::lane() ^ m); }
static inline bool __all(bool p) {
    hipemu::Wave& w = hipemu::my_wave();
    w.slot[hipemu::lane()][0] = p;
    hipemu::wave_sync();
    bool a = true;
    for (int l = 0; l < 64; ++l) a = a && w.slot[l][0];
    hipemu::wave_sync();
    return a;
}
static inline int __builtin_amdgcn_readfirstlane(int v) { return __shfl(v, 0); }
static inline int __builtin_amdgcn_readlane(int v, int l) { return __shfl(v, l); }
// v_mov_b32_dpp: the controls the kernels use (quad_perm, row_shl / row_shr, row_mirror, row_half_mirror), bound_ctrl -> 0 / old
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    hipemu::Wave& w = hipemu::my_wave();
    const int l = hipemu::lane(), row = l & ~15, p = l & 15;
    memcpy(w.slot[l], &src, 4);
    hipemu::wave_sync();
    int from = -1;
    if (ctrl < 0x100) from = (l & ~3) + ((ctrl >> (2 * (l & 3))) & 3);                  // quad_perm
    else if (ctrl >= 0x101 && ctrl <= 0x10F) from = p + (ctrl - 0x100) <= 15 ? l + (ctrl - 0x100) : -1;     // row_shl:n reads lane + n
    else if (ctrl >= 0x111 && ctrl <= 0x11F) from = p - (ctrl - 0x110) >= 0 ? l - (ctrl - 0x110) : -1;      // row_shr:n reads lane - n
    else if (ctrl == 0x140) from = row + 15 - p;                                         // row_mirror
    else if (ctrl == 0x141) from = row + (p & 8) + 7 - (p & 7);                          // row_half_mirror
    else { fprintf(stderr, "hipemu: dpp control 0x%x not emulated\n", ctrl); abort(); }
    int r = from >= 0 ? 0 : (bound_ctrl ? 0 : old);
    if (from >= 0) memcpy(&r, w.slot[from], 4);
    hipemu::wave_sync();
    return r;
}
static inline unsigned __builtin_amdgcn_perm(unsigned a, unsigned b, unsigned sel) {       // v_perm_b32: bytes 0..3 of b, 4..7 of a
    const unsigned char src[8] = {(unsigned char)b, (unsigned char)(b >> 8), (unsigned char)(b >> 16), (unsigned char)(b >> 24),
                                  (unsigned char)a, (unsigned char)(a >> 8), (unsigned char)(a >> 16), (unsigned char)(a >> 24)};
    unsigned r = 0;
    for (int i = 0; i < 4; ++i) {
        const unsigned s = (sel >> (8 * i)) & 0xff;
        if (s > 7) { fprintf(stderr, "hipemu: v_perm selector %u not emulated\n", s); abort(); }
        r |= (unsigned)src[s] << (8 * i);
    }
    return r;
}
static inline unsigned __builtin_amdgcn_alignbit(unsigned hi, unsigned lo, unsigned sh) {   // v_alignbit_b32: ({hi, lo} >> sh)[31:0]
    return (unsigned)(((((unsigned long long)hi) << 32) | lo) >> (sh & 31));
}
typedef float hipemu_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 hipemu_bf16x8 __attribute__((ext_vector_type(8)));
// v_mfma_f32_16x16x32_bf16: A[i = lane & 15][k = 8 (lane >> 4) + e], B[k = 8 (lane >> 4) + e][j = lane & 15],
// D[i = 4 (lane >> 4) + r][j = lane & 15] (products exact; the 32 of them and C summed in double and rounded ONCE here -
// the matrix core keeps more than fp32 inside one instruction; a chain of 32 fp32 additions would be the worse model)
static inline hipemu_f32x4 __builtin_amdgcn_mfma_f32_16x16x32_bf16(hipemu_bf16x8 a, hipemu_bf16x8 b, hipemu_f32x4 c, int, int, int) {
    hipemu::Wave& w = hipemu::my_wave();
    const int l = hipemu::lane();
    memcpy(w.slot[l], &a, 16);
    memcpy(w.slot[l] + 16, &b, 16);
    hipemu::wave_sync();
    hipemu_f32x4 d = c;
    const int j = l & 15;
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * (l >> 4) + r;
        double acc = (double)c[r];
        for (int k = 0; k < 32; ++k) {
            hipemu_bf16x8 av, bv;
            memcpy(&av, w.slot[i + 16 * (k >> 3)], 16);
            memcpy(&bv, w.slot[j + 16 * (k >> 3)] + 16, 16);
            acc += (double)(float)av[k & 7] * (double)(float)bv[k & 7];
        }
        d[r] = (float)acc;
    }
    hipemu::wave_sync();
    return d;
}
// v_mfma_f32_16x16x4_f32: A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15], D[4 (lane >> 4) + r][lane & 15]
static inline hipemu_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, hipemu_f32x4 c, int, int, int) {
    hipemu::Wave& w = hipemu::my_wave();
    const int l = hipemu::lane();
    memcpy(w.slot[l], &a, 4);
    memcpy(w.slot[l] + 4, &b, 4);
    hipemu::wave_sync();
    hipemu_f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * (l >> 4) + r, j = l & 15;
        double acc = (double)c[r];
        for (int k = 0; k < 4; ++k) {
            float av, bv;
            memcpy(&av, w.slot[i + 16 * k], 4);
            memcpy(&bv, w.slot[j + 16 * k] + 4, 4);
            acc += (double)av * (double)bv;
        }
        d[r] = (float)acc;
    }
    hipemu::wave_sync();
    return d;
}

// ------------------------------------------------------------------------------------------------ buffer resources
typedef struct { char* base; unsigned size; } __amdgpu_buffer_rsrc_t;
static inline __amdgpu_buffer_rsrc_t __builtin_amdgcn_make_buffer_rsrc(const volatile void* p, short, unsigned num_bytes, unsigned) {
    return __amdgpu_buffer_rsrc_t{(char*)p, p ? num_bytes : 0u};
}
typedef unsigned hipemu_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned hipemu_u32x2 __attribute__((ext_vector_type(2)));
template <class T> static inline T hipemu_buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    T v;
    memset(&v, 0, sizeof(T));
    const unsigned long o = (unsigned long)voff + soff;            // (32-bit wrap of voff + soff is not modelled: the kernels keep offsets below 2^32)
    // range check per dword, as the hardware does for raw buffers
    for (unsigned d = 0; d < sizeof(T); d += (sizeof(T) >= 4 ? 4 : sizeof(T)))
        if (o + d + (sizeof(T) >= 4 ? 4 : sizeof(T)) <= r.size) memcpy((char*)&v + d, r.base + o + d, sizeof(T) >= 4 ? 4 : sizeof(T));
    return v;
}
template <class T> static inline void hipemu_buf_store(T v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const unsigned long o = (unsigned long)voff + soff;
    for (unsigned d = 0; d < sizeof(T); d += (sizeof(T) >= 4 ? 4 : sizeof(T)))
        if (o + d + (sizeof(T) >= 4 ? 4 : sizeof(T)) <= r.size) memcpy(r.base + o + d, (char*)&v + d, sizeof(T) >= 4 ? 4 : sizeof(T));
}
static inline hipemu_u32x4 __builtin_amdgcn_raw_buffer_load_b128(__amdgpu_buffer_rsrc_t r, unsigned v, unsigned s, int) { return hipemu_buf_load<hipemu_u32x4>(r, v, s); }
static inline hipemu_u32x2 __builtin_amdgcn_raw_buffer_load_b64(__amdgpu_buffer_rsrc_t r, unsigned v, unsigned s, int) { return hipemu_buf_load<hipemu_u32x2>(r, v, s); }
static inline unsigned __builtin_amdgcn_raw_buffer_load_b32(__amdgpu_buffer_rsrc_t r, unsigned v, unsigned s, int) { return hipemu_buf_load<unsigned>(r, v, s); }
static inline unsigned char __builtin_amdgcn_raw_buffer_load_b8(__amdgpu_buffer_rsrc_t r, unsigned v, unsigned s, int) { return hipemu_buf_load<unsigned char>(r, v, s); }
static inline void __builtin_amdgcn_raw_buffer_store_b128(hipemu_u32x4 x, __amdgpu_buffer_rsrc_t r, unsigned v, unsigned s, int) { hipemu_buf_store(x, r, v, s); }
static inline void __builtin_amdgcn_raw_buffer_store_b32(unsigned x, __amdgpu_buffer_rsrc_t r, unsigned v, unsigned s, int) { hipemu_buf_store(x, r, v, s); }
static inline void __builtin_amdgcn_raw_buffer_store_b8(unsigned char x, __amdgpu_buffer_rsrc_t r, unsigned v, unsigned s, int) { hipemu_buf_store(x, r, v, s); }
static inline void __builtin_amdgcn_fence(int, const char*) {}
static inline void __builtin_amdgcn_wave_barrier() { hipemu::wave_sync(); }     // (the lanes of a wave run one after the other here: the kernels'
                                                                                 // "all lanes read, then all lanes write" points are real rendezvous)
// a spin loop must let the producers run: the fiber yields; in concurrent mode the OS thread also gives its core away - once per
// 64 polls (a wave's worth) with a real sleep, so that a hundred polling workgroups do not starve the few that have work
static inline void __builtin_amdgcn_s_sleep(int) {
    hipemu::yield();
    if (hipemu::g_concurrent) {
        static thread_local unsigned polls = 0;
        if ((++polls & 63u) == 0) { struct timespec ts = {0, 100000}; nanosleep(&ts, nullptr); }
    }
}
static inline void __builtin_amdgcn_s_setprio(int) {}
// s_getreg_b32 is only used for HW_REG_XCC_ID (the XCD a wave runs on).  Mode 0: the dispatcher of an 8-XCD device in SPX mode
// (block id mod 8); 1: one XCD for everybody (a CPX partition); 2: pairs of consecutive blocks share an XCD (a placement that is
// NOT a function of id mod 8) - hipemu_set_xcc_mode, for the tests of the placement probe and of the scans' own check.
// 3: round-robin, but the dispatcher's XCD pointer carries over from launch to launch (g_xcc_start: blocks launched so far).
namespace hipemu { extern int g_xcc_mode; extern unsigned g_xcc_start; }
static inline unsigned __builtin_amdgcn_s_getreg(int) {
    const unsigned id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    return hipemu::g_xcc_mode == 1 ? 0u : hipemu::g_xcc_mode == 2 ? (id >> 1) & 7u : hipemu::g_xcc_mode == 3 ? (id + hipemu::g_xcc_start) & 7u : id & 7u;
}
#ifndef __HIP_MEMORY_SCOPE_AGENT          /* (__hip_atomic_load / _store are clang builtins on every target; the scope names are HIP-mode macros) */
#define __HIP_MEMORY_SCOPE_SINGLETHREAD 1
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#endif
static inline float __builtin_amdgcn_exp2f(float x) { return exp2f(x); }
static inline float __builtin_amdgcn_rcpf(float x) { return 1.f / x; }
static inline void __builtin_amdgcn_sched_barrier(int) {}
static inline void __builtin_amdgcn_sched_group_barrier(int, int, int) {}
