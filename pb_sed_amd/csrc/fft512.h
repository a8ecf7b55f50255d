// Complex helpers and the radix-8 butterfly of the 512-point Stockham FFT in logmel.hip (3 radix-8 passes, one butterfly
// per lane of a 64-wide wave; the passes and the real-FFT-1024 unpacking live next to their LDS indexing in logmel.hip).
#pragma once
#ifdef __HIPCC__
#define PBSED_HD __host__ __device__ __forceinline__
#else
#define PBSED_HD inline
#endif

namespace pbsed {

struct cpx { float x, y; };
PBSED_HD cpx cadd(cpx a, cpx b) { return cpx{a.x + b.x, a.y + b.y}; }
PBSED_HD cpx csub(cpx a, cpx b) { return cpx{a.x - b.x, a.y - b.y}; }
PBSED_HD cpx cmul(cpx a, cpx b) { return cpx{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
PBSED_HD cpx mul_mi(cpx a) { return cpx{a.y, -a.x}; }   // * (-i)

PBSED_HD void dft4(cpx c0, cpx c1, cpx c2, cpx c3, cpx& y0, cpx& y1, cpx& y2, cpx& y3) {
    const cpx e0 = cadd(c0, c2), e1 = csub(c0, c2), o0 = cadd(c1, c3), o1 = mul_mi(csub(c1, c3));
    y0 = cadd(e0, o0); y1 = cadd(e1, o1); y2 = csub(e0, o0); y3 = csub(e1, o1);
}

PBSED_HD void dft8(cpx* v) {
    const float h = 0.70710678118654752440f;
    cpx a0 = cadd(v[0], v[4]), a1 = cadd(v[1], v[5]), a2 = cadd(v[2], v[6]), a3 = cadd(v[3], v[7]);
    cpx b0 = csub(v[0], v[4]), b1 = csub(v[1], v[5]), b2 = csub(v[2], v[6]), b3 = csub(v[3], v[7]);
    b1 = cpx{(b1.x + b1.y) * h, (b1.y - b1.x) * h};     // * (1-i)/sqrt2
    b2 = mul_mi(b2);                                      // * (-i)
    b3 = cpx{(b3.y - b3.x) * h, (-b3.x - b3.y) * h};    // * (-1-i)/sqrt2
    dft4(a0, a1, a2, a3, v[0], v[2], v[4], v[6]);
    dft4(b0, b1, b2, b3, v[1], v[3], v[5], v[7]);
}

}  // namespace pbsed
