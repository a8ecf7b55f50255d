// Microbenchmark: how do bf16 MFMAs of one wave and VALU work of ANOTHER wave on the same SIMD overlap on gfx950?
// Block = 8 waves: waves 0..3 (one per SIMD) issue 72 independent-chain v_mfma_f32_16x16x32_bf16 per iteration, waves 4..7
// (their SIMD partners) issue PV v_fma_f32 per iteration (8 independent chains).  Both loop `iters` times without barriers:
// the kernel takes max(consumer, producer) if the two pipes overlap, their sum if they exclude each other.
// Build: hipcc -O3 --offload-arch=gfx950 mfma_valu_bf16.hip -o mfma_valu_bf16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int PV, int MODE>      // MODE 1: consumers only, 2: producers only, 3: both
__global__ __launch_bounds__(512) void kern(float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    const u32x4 a = {0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = {0x3f803f80u, 0x3f813f80u, 0x3f803f80u, 0x3f803f80u};
    if (wave < 4) {
        if (!(MODE & 1)) return;
        f32x4 acc[12];
        for (int i = 0; i < 12; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int i = 0; i < 12; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[i], 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        if (s == 12345.678f) out[0] = s;
    } else {
        if (!(MODE & 2)) return;
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
        const float x = out[1], y = out[2];
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < PV; ++k) v[k & 7] = __builtin_fmaf(v[k & 7], x, y);
        }
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += v[i];
        if (s == 12345.678f) out[0] = s;
    }
}

template <int PV, int MODE>
double run(float* d) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((kern<PV, MODE>), dim3(256), dim3(512), 0, 0, d, 50);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((kern<PV, MODE>), dim3(256), dim3(512), 0, 0, d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3 * 2.4e9 / iters;
}

template <int PV>
void row(float* d) {
    const double c = run<PV, 1>(d), p = run<PV, 2>(d), both = run<PV, 3>(d);
    printf("PV %4d: 72 MFMA alone %6.0f clk, %4d v_fma alone %6.0f clk (%.1f clk each), together %6.0f clk   (max %6.0f, sum %6.0f)\n", PV, c, PV, p,
           p / PV, both, c > p ? c : p, c + p);
}

int main() {
    float* d; (void)hipMalloc(&d, 64); (void)hipMemset(d, 0, 64);
    row<64>(d); row<128>(d); row<256>(d); row<512>(d); row<1024>(d);
    return 0;
}
