// probe_mfma_rate.hip -- issue rate of v_mfma_i32_16x16x64_i8 from ONE wave per SIMD (the regime of k_wfm_mfma),
// with 3 and 6 independent accumulators, plus the same with an XOR + 3 cvt/fma VALU ops between MFMAs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v4i __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
template <int NACC, bool VALU>
__global__ __launch_bounds__(64) void k_rate(const v4i *src, v4i *dst, int iters)
{
    v4i a = src[threadIdx.x], b = src[64 + threadIdx.x];
    v4i acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = v4i{0, 0, 0, 0};
    float f = 0.f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 4; rep++) {
#pragma unroll
            for (int i = 0; i < NACC; i++) {
                acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[i], 0, 0, 0);
                if (VALU) { b[i & 3] ^= 0x01010101; f = fmaf((float)b[0], 1.0001f, f); f = fmaf(f, 0.999f, 1.0f); }
            }
        }
    }
    v4i r = acc[0];
#pragma unroll
    for (int i = 1; i < NACC; i++) r += acc[i];
    r[0] += (int)f;
    dst[blockIdx.x * 64 + threadIdx.x] = r;
}
template <int NACC, bool VALU> void run(const char *name, v4i *src, v4i *dst, int blocks)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    hipLaunchKernelGGL((k_rate<NACC, VALU>), dim3(blocks), dim3(64), 0, 0, src, dst, 10);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_rate<NACC, VALU>), dim3(blocks), dim3(64), 0, 0, src, dst, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double mfma_per_wave = (double)iters * 4 * NACC;
    const double waves_per_simd = blocks / 1024.0;
    printf("%-28s blocks=%5d: %.3f ms -> %.1f ns per MFMA per SIMD (%.1f cycles @2.4GHz), %.0f TOPS\n", name, blocks, ms,
           ms * 1e6 / (mfma_per_wave * waves_per_simd), ms * 1e6 / (mfma_per_wave * waves_per_simd) * 2.4, blocks * mfma_per_wave * 32768.0 / (ms * 1e-3) / 1e12);
}
int main()
{
    v4i *src, *dst; CK(hipMalloc(&src, 2048)); CK(hipMalloc(&dst, 8192 * 1024)); CK(hipMemset(src, 1, 2048));
    run<3, false>("3 acc, MFMA only", src, dst, 1024);
    run<6, false>("6 acc, MFMA only", src, dst, 1024);
    run<3, true>("3 acc, + xor/cvt/2fma each", src, dst, 1024);
    run<6, true>("6 acc, + xor/cvt/2fma each", src, dst, 1024);
    run<3, false>("3 acc, MFMA only, 2 waves/SIMD", src, dst, 2048);
    run<3, true>("3 acc, +VALU, 2 waves/SIMD", src, dst, 2048);
    return 0;
}
