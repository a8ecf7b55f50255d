// Microbenchmark: issue cost of v_pk_fma_f32 against v_fma_f32 on gfx950 (one wave per SIMD, 8 independent chains).
// Build: hipcc -O3 --offload-arch=gfx950 pk_fma.hip -o pk_fma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>      // 0: 256 v_fma_f32 per iteration, 1: 256 v_pk_fma_f32 per iteration (512 fma's worth)
__global__ __launch_bounds__(256) void kern(float* out, int iters) {
    const float x = out[1], y = out[2];
    if (MODE == 0) {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 256; ++k) v[k & 7] = __builtin_fmaf(v[k & 7], x, y);
        }
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += v[i];
        if (s == 12345.678f) out[0] = s;
    } else {
        f32x2 v[8];
        const f32x2 x2 = {x, x + 1.f}, y2 = {y, y - 1.f};
        for (int i = 0; i < 8; ++i) v[i] = f32x2{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f - i};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 256; ++k) v[k & 7] = __builtin_elementwise_fma(v[k & 7], x2, y2);
        }
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
        if (s == 12345.678f) out[0] = s;
    }
}

template <int MODE>
double run(float* d) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((kern<MODE>), dim3(256), dim3(256), 0, 0, d, 50);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((kern<MODE>), dim3(256), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3 * 2.4e9 / iters;
}

int main() {
    float* d;
    (void)hipMalloc(&d, 64);
    (void)hipMemset(d, 0, 64);
    printf("256 v_fma_f32 per wave:    %.0f clocks (at 2.4 GHz)\n", run<0>(d));
    printf("256 v_pk_fma_f32 per wave: %.0f clocks\n", run<1>(d));
    return 0;
}
