// fft_butterflies.hpp -- register-level butterflies shared by the own FFT kernels (fft64k.hip: 65536 = 256 x 256 for the overlap-add filter;
// fastddc_mfma.hip: the 512-point inverse transforms and the 65536 = 512 x 128 forward transform of the channelizer).  Natural order in and out.
#pragma once
#include <hip/hip_runtime.h>

namespace {

// (the transforms carry no bit-exact contract -- float FFT paths are gated at 1e-5 relative RMS -- so their products may contract into FMAs even though the
// library is built -ffp-contract=off for the phase recurrences)
__host__ __device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
#pragma clang fp contract(fast)
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__host__ __device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__host__ __device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// multiplication by -j (forward) / +j (inverse)
template <bool INV> __host__ __device__ __forceinline__ float2 rot90(float2 a) { return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x); }

template <bool INV>
__host__ __device__ __forceinline__ void dft4(float2 &x0, float2 &x1, float2 &x2, float2 &x3)
{
    const float2 s02 = cadd(x0, x2), d02 = csub(x0, x2), s13 = cadd(x1, x3), d13 = rot90<INV>(csub(x1, x3));
    x0 = cadd(s02, s13); x2 = csub(s02, s13); x1 = cadd(d02, d13); x3 = csub(d02, d13);
}

// 16-point DFT in registers, natural order in and out:  n = 4 n1 + n2, k = k1 + 4 k2
template <bool INV>
__host__ __device__ __forceinline__ void dft16(float2 (&v)[16])
{
    const float c1 = 0.92387953251128673848f, s1 = 0.38268343236508978178f, r2 = 0.70710678118654752440f;   // cos, sin of pi/8; sqrt(1/2)
#pragma unroll
    for (int n2 = 0; n2 < 4; n2++) dft4<INV>(v[n2], v[4 + n2], v[8 + n2], v[12 + n2]);       // over n1: v[4 k1 + n2]
    // twiddles W16^(n2 k1), W16 = exp(-+ 2 pi i / 16)
    const float sg = INV ? 1.f : -1.f;
    const float2 w1 = make_float2(c1, sg * s1), w2 = make_float2(r2, sg * r2), w3 = make_float2(s1, sg * c1);
    const float2 w4 = make_float2(0.f, sg), w6 = make_float2(-r2, sg * r2), w9 = make_float2(-c1, -sg * s1);
    v[4 + 1] = cmul(v[4 + 1], w1); v[4 + 2] = cmul(v[4 + 2], w2); v[4 + 3] = cmul(v[4 + 3], w3);          // k1 = 1
    v[8 + 1] = cmul(v[8 + 1], w2); v[8 + 2] = cmul(v[8 + 2], w4); v[8 + 3] = cmul(v[8 + 3], w6);          // k1 = 2
    v[12 + 1] = cmul(v[12 + 1], w3); v[12 + 2] = cmul(v[12 + 2], w6); v[12 + 3] = cmul(v[12 + 3], w9);    // k1 = 3
#pragma unroll
    for (int k1 = 0; k1 < 4; k1++) dft4<INV>(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);    // over n2: v[4 k1 + k2]
    // v[4 k1 + k2] holds X[k1 + 4 k2]: transpose to natural order
    float2 t;
    t = v[1]; v[1] = v[4]; v[4] = t;   t = v[2]; v[2] = v[8]; v[8] = t;   t = v[3]; v[3] = v[12]; v[12] = t;
    t = v[6]; v[6] = v[9]; v[9] = t;   t = v[7]; v[7] = v[13]; v[13] = t; t = v[11]; v[11] = v[14]; v[14] = t;
}


// 8-point DFT in registers, natural order in and out: even / odd split, X[k] = E[k] + W8^k O[k], X[k+4] = E[k] - W8^k O[k]
template <bool INV>
__host__ __device__ __forceinline__ void dft8(float2 (&v)[8])
{
    const float r2 = 0.70710678118654752440f, sg = INV ? 1.f : -1.f;
    dft4<INV>(v[0], v[2], v[4], v[6]);
    dft4<INV>(v[1], v[3], v[5], v[7]);
    const float2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
    const float2 o0 = v[1], o1 = cmul(v[3], make_float2(r2, sg * r2)), o2 = rot90<INV>(v[5]), o3 = cmul(v[7], make_float2(-r2, sg * r2));
    v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
    v[1] = cadd(e1, o1); v[5] = csub(e1, o1);
    v[2] = cadd(e2, o2); v[6] = csub(e2, o2);
    v[3] = cadd(e3, o3); v[7] = csub(e3, o3);
}

} // namespace
