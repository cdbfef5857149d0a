// tools/probes/dft64_rate.hip -- the 64-point register transform of fftfilt_wave.hpp alone (no memory traffic): ns per transform and wave, scalar against packed form,
// one and two waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -I csdr_amd/csrc tools/probes/dft64_rate.hip -o tools/probes/dft64_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#pragma clang fp contract(fast)
#define FFL_HD __host__ __device__ __forceinline__
#include "fft_butterflies.hpp"
#include "fftfilt_wave.hpp"
template <int KIND> __global__ __launch_bounds__(512, 2) void k(float2 *io, int iters)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (KIND == 0) {
        float2 v[64];
#pragma unroll
        for (int j = 0; j < 64; j++) v[j] = io[t + j];
        for (int it = 0; it < iters; it++) { dft64<false>(v); dft64<true>(v);
#pragma unroll
            for (int j = 0; j < 64; j++) { v[j].x *= 0.015625f; v[j].y *= 0.015625f; } }
#pragma unroll
        for (int j = 0; j < 64; j++) io[t + j] = v[j];
    } else {
        fw_pk2 v[64];
#pragma unroll
        for (int j = 0; j < 64; j++) v[j] = fw_pk2{io[t + j].x, io[t + j].y};
        for (int it = 0; it < iters; it++) { fw_pk_dft64<false>(v); fw_pk_dft64<true>(v);
#pragma unroll
            for (int j = 0; j < 64; j++) v[j] *= 0.015625f; }
#pragma unroll
        for (int j = 0; j < 64; j++) io[t + j] = make_float2(v[j].x, v[j].y);
    }
}
template <int KIND> void run(const char *name, float2 *io)
{
    for (int threads : {256, 512}) {
        const int iters = 2000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<KIND><<<256, threads>>>(io, 10); hipDeviceSynchronize();
        hipEventRecord(e0); k<KIND><<<256, threads>>>(io, iters); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-8s %d wave(s) per SIMD: %.3f us per 64-point transform and wave (%.3f per SIMD)\n", name, threads / 256, ms * 1e3 / (2.0 * iters), ms * 1e3 / (2.0 * iters) / (threads / 256));
    }
}
int main()
{
    float2 *io; hipMalloc(&io, 64 << 20); hipMemset(io, 0, 64 << 20);
    run<0>("scalar", io); run<1>("packed", io);
    return 0;
}
