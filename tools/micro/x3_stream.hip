// Microbenchmark for the bf16x3 Winograd convolution design (csrc/conv_winox3.hip): can ONE consumer wave per SIMD keep the
// bf16 MFMA pipe busy when its A operands (pre-split transformed weights, 9 KB per 72 MFMAs) stream from L2 straight into
// registers and its B operands come from LDS, while a second wave per SIMD does the staging work (VALU + ds_write)?
//   iteration of a consumer wave = one transform point: 9 A fragments (3 kh x 3 parts, 16 B per lane each) from global,
//   6 halo rows x 3 parts from LDS, 12 (row, kh) sets x 6 part products = 72 v_mfma_f32_16x16x32_bf16 (ideal 72 x 16 clk)
//   mode bit 0: A fragments are re-loaded from an L2-resident buffer every iteration (else: loaded once)
//   mode bit 1: producer waves 4..7 run a VALU + ds_write loop beside the consumers
//   mode bit 2: B fragments are re-read from LDS every set group (else: once per iteration)
// Build: hipcc -O3 --offload-arch=gfx950 x3_stream.hip -o x3_stream ; run: ./x3_stream
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x4 mf(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int MODE>
__global__ __launch_bounds__(512) void kern(const u32x4* __restrict__ u, float* out, int iters, unsigned umask) {
    extern __shared__ u32x4 lds[];                       // [3 parts][6 rows][64 lanes] + producer scratch
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 3 * 6 * 64 + 4096; i += 512) lds[i] = u32x4{0x3f803f80u + i, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    __syncthreads();
    if (wave < 4) {
        f32x4 acc[4][3];
        for (int f = 0; f < 4; ++f)
            for (int k = 0; k < 3; ++k) acc[f][k] = f32x4{0.f, 0.f, 0.f, 0.f};
        // a wave's stream: 9 KB per iteration, contiguous; different waves / blocks start at different places
        unsigned pos = ((blockIdx.x * 4 + wave) * 9 * 64 * 7) & umask;
        u32x4 A[2][9];
        for (int j = 0; j < 9; ++j) A[0][j] = u[((pos + j * 64) & umask) + lane];
        auto body = [&](u32x4 (&Ac)[9], u32x4 (&An)[9]) __attribute__((always_inline)) {
            pos = (pos + 9 * 64) & umask;
            if (MODE & 1) {
#pragma unroll
                for (int j = 0; j < 9; ++j) An[j] = u[((pos + j * 64) & umask) + lane];
            }
#pragma unroll
            for (int h = 0; h < 6; ++h) {
                u32x4 B[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) B[p] = lds[(p * 6 + ((MODE & 4) ? h : 0)) * 64 + lane];
                // part products, smallest first, round-robin over the (row, kh) sets of this halo row
#pragma unroll
                for (int pp = 0; pp < 6; ++pp) {
                    const int pa = pp == 0 ? 2 : pp == 1 ? 0 : pp == 2 ? 1 : pp == 3 ? 1 : 0;     // lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi
                    const int pb = pp == 0 ? 0 : pp == 1 ? 2 : pp == 2 ? 1 : pp == 3 ? 0 : pp == 4 ? 1 : 0;
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh) {
                        const int f = h - kh;
                        if (f >= 0 && f < 4) acc[f][kh] = mf(Ac[kh * 3 + pa], B[pb], acc[f][kh]);
                    }
                }
            }
        };
        for (int it = 0; it < iters; it += 2) {
            body(A[0], A[1]);
            body(A[1], A[0]);
        }
        float s = 0.f;
        for (int f = 0; f < 4; ++f)
            for (int k = 0; k < 3; ++k) s += acc[f][k][0] + acc[f][k][1] + acc[f][k][2] + acc[f][k][3];
        if (s == 12345.678f) out[0] = s;
    } else if (MODE & 2) {
        float v[12];
        for (int i = 0; i < 12; ++i) v[i] = tid * 0.001f + i;
        const float a = out[1], b = out[2];
        unsigned short* hs = reinterpret_cast<unsigned short*>(lds + 3 * 6 * 64);
        for (int it = 0; it < iters; ++it) {
            // per consumer iteration (one point): 3 items x 4 tiles: ~5 transform + 4 split VALU, 3 ds_write_b16 each
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                float x = v[i];
                x = __builtin_fmaf(x, a, b); x = __builtin_fmaf(x, a, b); x = __builtin_fmaf(x, a, b); x = __builtin_fmaf(x, a, b);
                x = __builtin_fmaf(x, a, b);
                const float h = __uint_as_float(__float_as_uint(x) & 0xffff0000u), r = x - h;
                const float m = __uint_as_float(__float_as_uint(r) & 0xffff0000u), l = r - m;
                hs[(tid - 256) * 2 + i * 1024] = __float_as_uint(h) >> 16;
                hs[(tid - 256) * 2 + i * 1024 + 8192] = __float_as_uint(m) >> 16;
                hs[(tid - 256) * 2 + i * 1024 + 16384] = __float_as_uint(l) >> 16;
                v[i] = l;
            }
        }
        float s = 0.f;
        for (int i = 0; i < 12; ++i) s += v[i];
        if (s == 12345.678f) out[0] = s;
    }
}

template <int MODE>
void run(const u32x4* u, float* d, unsigned umask, const char* what) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = (3 * 6 * 64 + 4096) * 16;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((kern<MODE>), dim3(256), dim3(512), lds, 0, u, d, 50, umask);
    hipEventRecord(e0);
    hipLaunchKernelGGL((kern<MODE>), dim3(256), dim3(512), lds, 0, u, d, iters, umask);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double clk = ms * 1e-3 * 2.4e9 / iters;
    printf("mode %d (%s): %.3f ms  %.0f clk/iter at 2.4 GHz (ideal 1152 = 72 MFMA x 16)  MFMA rate %.0f%%  U stream %.1f B/clk/CU\n", MODE,
           what, ms, clk, 100. * 1152 / clk, (MODE & 1) ? 4 * 9 * 1024 / clk : 0.);
}

int main() {
    const size_t n16 = (size_t)1 << 22;                  // 64 MB of 16-byte words; the mask selects the footprint
    u32x4* u; hipMalloc(&u, n16 * 16); hipMemset(u, 0x3f, n16 * 16);
    float* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
    const unsigned m2 = (2u << 20) / 16 - 1, m32 = (32u << 20) / 16 - 1;
    run<0>(u, d, m2, "registers only");
    run<4>(u, d, m2, "B from LDS per halo row");
    run<5>(u, d, m2, "+ A stream, 2 MB footprint");
    run<5>(u, d, m32, "+ A stream, 32 MB footprint");
    run<7>(u, d, m2, "+ producer waves, 2 MB");
    run<7>(u, d, m32, "+ producer waves, 32 MB");
    run<6>(u, d, m2, "producer waves, no A stream");
    return 0;
}
