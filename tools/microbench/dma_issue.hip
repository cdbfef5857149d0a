// dma_issue.hip -- how fast NW fetching waves of a workgroup can stream R = 16 / NW rows each by LDS-DMA (global_load_lds_dwordx4, 1 KiB pieces), continuously:
// per round a wave issues P pieces for each of its rows (straight-line code, M0 stepped by s_add), then waits until at most KEEP pieces are still in flight.  No
// barrier, no consumer: the per-wave / per-CU limits of the fetch path alone.  1024 rows x 4.8 MB, 64 x 4 workgroups (one per CU), every byte read once.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/dma_issue.hip -o /tmp/dma_issue && /tmp/dma_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__device__ unsigned long long g_cyc, g_cnt;

template <int NW, int P, int KEEP, bool NT>
__global__ __launch_bounds__(64 * NW) void k(const uint8_t *in, unsigned pitch, int rounds, unsigned *sink)
{
    constexpr int R = 16 / NW;
    extern __shared__ float4 raw[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) uint8_t *)raw + wv * R * 8192;      // 8 KiB of ring per row
    const uint8_t *base = in + (size_t)(blockIdx.x * 16 + wv * R) * pitch + (size_t)blockIdx.y * ((size_t)rounds * P * 1024);
    uint32_t vo[R];
#pragma unroll
    for (int r = 0; r < R; r++) vo[r] = r * pitch + 16 * lane;
    long long t = 0;
    int slot = 0;
    for (int rd = 0; rd < rounds; rd++) {
        const long long t0 = __builtin_readcyclecounter();
#pragma unroll
        for (int pc = 0; pc < P; pc++) {
            const uint8_t *sb = base + ((size_t)rd * P + pc) * 1024;
            const uint32_t la = lds0 + slot * 1024;
            uint32_t keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1" : "=&s"(keep) : "s"(la) : "memory");
#pragma unroll
            for (int r = 0; r < R; r++) {
                if (NT) asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt\n\ts_add_u32 m0, m0, 8192" :: "v"(vo[r]), "s"(sb) : "memory", "scc");
                else asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_add_u32 m0, m0, 8192" :: "v"(vo[r]), "s"(sb) : "memory", "scc");
            }
            asm volatile("s_mov_b32 m0, %0" :: "s"(keep) : "memory");
            slot = (slot + 1) & 7;
        }
        t += __builtin_readcyclecounter() - t0;
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(KEEP) : "memory");
    }
    if (lane == 0) { atomicAdd(&g_cyc, (unsigned long long)t); atomicAdd(&g_cnt, (unsigned long long)rounds * P * R); }
    if (rounds < 0) sink[0] = 1;
}
template <int NW, int P, int KEEP, bool NT = true> void run(const uint8_t *d, unsigned pitch, unsigned *sink)
{
    const int rounds = (int)(pitch / 4 / (P * 1024));
    const size_t lds = 16 * 8192;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void *)k<NW, P, KEEP, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL((k<NW, P, KEEP, NT>), dim3(64, 4), dim3(64 * NW), lds, 0, d, pitch, rounds, sink);
    hipDeviceSynchronize();
    unsigned long long z = 0; hipMemcpyToSymbol(HIP_SYMBOL(g_cyc), &z, 8); hipMemcpyToSymbol(HIP_SYMBOL(g_cnt), &z, 8);
    hipEventRecord(e0);
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL((k<NW, P, KEEP, NT>), dim3(64, 4), dim3(64 * NW), lds, 0, d, pitch, rounds, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    unsigned long long cyc, cnt; hipMemcpyFromSymbol(&cyc, HIP_SYMBOL(g_cyc), 8); hipMemcpyFromSymbol(&cnt, HIP_SYMBOL(g_cnt), 8);
    printf("%d fetching wave(s), %2d rows each, %d KiB per row and round, <= %2d pieces left in flight%s: %6.1f cycles per piece in the issue loop, %.3f ms, %.2f TB/s\n",
           NW, 16 / NW, P, KEEP, NT ? "" : " (default policy)", (double)cyc / cnt, ms, (double)cnt / 10 * 1024 / (ms * 1e-3) / 1e12);
}
int main()
{
    const unsigned pitch = 4800512; uint8_t *d; unsigned *sink;
    if (hipMalloc(&d, (size_t)1024 * pitch + (4 << 20)) != hipSuccess) return 1;
    hipMalloc(&sink, 64); hipMemset(d, 1, (size_t)1024 * pitch);
    run<1, 1, 0>(d, pitch, sink);  run<1, 1, 16>(d, pitch, sink); run<1, 1, 32>(d, pitch, sink); run<1, 1, 48>(d, pitch, sink);
    run<1, 2, 0>(d, pitch, sink);  run<1, 2, 16>(d, pitch, sink); run<1, 2, 32>(d, pitch, sink);
    run<2, 1, 0>(d, pitch, sink);  run<2, 1, 8>(d, pitch, sink);  run<2, 1, 16>(d, pitch, sink); run<2, 1, 24>(d, pitch, sink); run<2, 1, 32>(d, pitch, sink); run<2, 1, 48>(d, pitch, sink);
    run<2, 2, 0>(d, pitch, sink);  run<2, 2, 16>(d, pitch, sink); run<2, 2, 32>(d, pitch, sink); run<2, 2, 40>(d, pitch, sink);
    run<2, 3, 0>(d, pitch, sink);  run<2, 3, 24>(d, pitch, sink); run<2, 3, 36>(d, pitch, sink);
    run<2, 4, 0>(d, pitch, sink);  run<2, 4, 24>(d, pitch, sink);
    run<4, 1, 0>(d, pitch, sink);  run<4, 1, 8>(d, pitch, sink);  run<4, 1, 16>(d, pitch, sink); run<4, 2, 8>(d, pitch, sink); run<4, 2, 16>(d, pitch, sink);
    run<8, 1, 4>(d, pitch, sink);  run<8, 1, 8>(d, pitch, sink);  run<8, 2, 8>(d, pitch, sink);
    run<2, 2, 16, false>(d, pitch, sink);
    return 0;
}
