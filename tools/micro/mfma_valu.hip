// Microbenchmark: does VALU work hide behind fp32 MFMAs on gfx950?
//   mode 0: every wave runs {4 independent MFMA 16x16x4 f32 + K independent v_fma} per iteration
//   mode 1: even waves run MFMAs only, odd waves run the VALU work only (2 waves per SIMD)
// Build: hipcc -O3 --offload-arch=gfx950 mfma_valu.hip -o mfma_valu ; run: ./mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int K, int MODE>
__global__ __launch_bounds__(512) void kern(float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
    const float a = out[0], b = out[1];
    const bool do_m = MODE == 0 || (wave & 4) == 0;      // waves 0-3 -> SIMD 0-3 first wave; 4-7 second wave per SIMD
    const bool do_v = MODE == 0 || (wave & 4) != 0;
    for (int it = 0; it < iters; ++it) {
        if (do_m) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
        if (do_v) {
#pragma unroll
            for (int k = 0; k < K; ++k) v[k & 7] = __builtin_fmaf(v[k & 7], a, b);
        }
        if (MODE == 0) {
            // interleave: 1 MFMA, K/4 VALU
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (K > 0) __builtin_amdgcn_sched_group_barrier(0x002, K / 4, 0);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.678f) out[2] = s;
}

template <int K, int MODE>
void run(float* d, int nthreads) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((kern<K, MODE>), dim3(256), dim3(nthreads), 0, 0, d, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL((kern<K, MODE>), dim3(256), dim3(nthreads), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // cycles per iteration at 2.4 GHz
    printf("mode %d threads %d K=%2d: %.3f ms  %.1f cycles/iter (4 MFMA = 128 pipe cycles)\n", MODE, nthreads, K, ms,
           ms * 1e-3 * 2.4e9 / iters);
}

int main() {
    float* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
    run<0, 0>(d, 256); run<8, 0>(d, 256); run<16, 0>(d, 256); run<24, 0>(d, 256); run<32, 0>(d, 256); run<48, 0>(d, 256);
    run<0, 0>(d, 512); run<16, 0>(d, 512); run<32, 0>(d, 512);
    run<0, 1>(d, 512); run<8, 1>(d, 512); run<16, 1>(d, 512); run<32, 1>(d, 512); run<64, 1>(d, 512);
    return 0;
}
