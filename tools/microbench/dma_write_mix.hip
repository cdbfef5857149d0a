// dma_write_mix.hip -- what a thin stream of stores (2 % of the bytes) costs a saturated LDS-DMA read stream, depending on how many contiguous bytes per row
// one store event covers.  Per workgroup (one per CU): wave 0 streams 16 rows by LDS-DMA (1 KiB per row and round, <= 16 pieces left in flight: 7.0 TB/s alone,
// tools/microbench/dma_issue.hip); wave 1 writes, every PER rounds, LINES x 128 bytes to each of 16 output rows (16 bytes per lane), paced by a counter in LDS.
// The chain kernels' shape: 1024 input rows x 4.8 MB, output 2 % of that.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/dma_write_mix.hip -o /tmp/dma_write_mix && /tmp/dma_write_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int LINES, int RP = 0>     // 128-byte lines per row and store event; 0 = no stores.  RP: cache policy of the reads (0 nt, 1 default, 2 sc1, 3 sc0 sc1, 4 nt sc1, 5 nt sc0 sc1)
__global__ __launch_bounds__(128) void k(const uint8_t *in, unsigned pitch, int rounds, uint8_t *out, unsigned out_pitch, int same_line)
{
    extern __shared__ float4 raw[];
    volatile int *flag = reinterpret_cast<volatile int *>(reinterpret_cast<uint8_t *>(raw) + 16 * 8192);
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x == 0) *flag = 0;
    __syncthreads();
    if (wv == 0) {
        const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) uint8_t *)raw;
        const uint8_t *base = in + (size_t)(blockIdx.x * 16) * pitch + (size_t)blockIdx.y * ((size_t)rounds * 1024);
        uint32_t vo[16];
#pragma unroll
        for (int r = 0; r < 16; r++) vo[r] = r * pitch + 16 * lane;
        int slot = 0;
        for (int rd = 0; rd < rounds; rd++) {
            const uint8_t *sb = base + (size_t)rd * 1024;
            const uint32_t la = lds0 + slot * 1024;
            uint32_t keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1" : "=&s"(keep) : "s"(la) : "memory");
#pragma unroll
            for (int r = 0; r < 16; r++) {
                if (RP == 0) asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt\n\ts_add_u32 m0, m0, 8192" :: "v"(vo[r]), "s"(sb) : "memory", "scc");
                if (RP == 1) asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_add_u32 m0, m0, 8192" :: "v"(vo[r]), "s"(sb) : "memory", "scc");
                if (RP == 2) asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 sc1\n\ts_add_u32 m0, m0, 8192" :: "v"(vo[r]), "s"(sb) : "memory", "scc");
                if (RP == 3) asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 sc0 sc1\n\ts_add_u32 m0, m0, 8192" :: "v"(vo[r]), "s"(sb) : "memory", "scc");
                if (RP == 4) asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 sc1 nt\n\ts_add_u32 m0, m0, 8192" :: "v"(vo[r]), "s"(sb) : "memory", "scc");
                if (RP == 5) asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 sc0 sc1 nt\n\ts_add_u32 m0, m0, 8192" :: "v"(vo[r]), "s"(sb) : "memory", "scc");
            }
            asm volatile("s_mov_b32 m0, %0" :: "s"(keep) : "memory");
            slot = (slot + 1) & 7;
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            if (lane == 0) *flag = rd + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (LINES > 0) {
        // 2 % of the bytes: 16 KiB read per round -> 328 bytes written per round -> one event of 16 rows x LINES x 128 bytes every LINES x 6.25 rounds
        const int per4 = LINES * 25;                                     // rounds x 4 between events
        uint8_t *ob = out + (size_t)(blockIdx.x * 16) * out_pitch + (size_t)blockIdx.y * (out_pitch / 4);
        size_t pos = 0;
        const uint4 v = {1u, 2u, 3u, (unsigned)lane};
        if (same_line >= 2) {
            // staging: events of LINES x 128 bytes per row go to a per-workgroup scratch (16 rows x SCR bytes, meant to stay in L2); once SCR bytes per row are
            // there they are copied to the final rows in one piece
            const unsigned SCR = (unsigned)same_line;                    // bytes per row staged before the flush (a multiple of LINES * 128)
            uint8_t *scr = out + (size_t)1024 * out_pitch + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 * SCR;
            unsigned fill = 0;
            for (int ev = 1; ev * per4 / 4 <= rounds; ev++) {
                const int target = ev * per4 / 4;
                while (*flag < target) __builtin_amdgcn_s_sleep(8);
                for (int r = 0; r < 16; r++)
                    for (int l0 = 0; l0 < LINES * 8; l0 += 64)
                        if (l0 + lane < LINES * 8) *reinterpret_cast<uint4 *>(scr + (size_t)r * SCR + fill + (size_t)(l0 + lane) * 16) = v;
                fill += LINES * 128;
                if (fill == SCR) {
                    for (int r = 0; r < 16; r++)
                        for (unsigned b = 0; b < SCR; b += 1024) {
                            const uint4 t = *reinterpret_cast<const uint4 *>(scr + (size_t)r * SCR + b + lane * 16);
                            *reinterpret_cast<uint4 *>(ob + (size_t)r * out_pitch + pos + b + lane * 16) = t;
                        }
                    pos += SCR; fill = 0;
                }
            }
        } else if (same_line == -1) {
            // line major: one store instruction = one 128-byte line of 8 rows (lane = 8 row + piece), the LINES lines of an event back to back
            for (int ev = 1; ev * per4 / 4 <= rounds; ev++) {
                const int target = ev * per4 / 4;
                while (*flag < target) __builtin_amdgcn_s_sleep(8);
                for (int l = 0; l < LINES; l++)
                    for (int half = 0; half < 2; half++)
                        *reinterpret_cast<uint4 *>(ob + (size_t)(8 * half + (lane >> 3)) * out_pitch + pos + (size_t)l * 128 + (lane & 7) * 16) = v;
                pos += LINES * 128;
            }
        } else
        for (int ev = 1; ev * per4 / 4 <= rounds; ev++) {
            const int target = ev * per4 / 4;
            while (*flag < target) __builtin_amdgcn_s_sleep(8);
            for (int r = 0; r < 16; r++)
                for (int l0 = 0; l0 < LINES * 8; l0 += 64)
                    if (l0 + lane < LINES * 8) *reinterpret_cast<uint4 *>(ob + (size_t)r * out_pitch + (same_line ? 0 : pos) + (size_t)(l0 + lane) * 16) = v;
            pos += LINES * 128;
        }
    }
}
template <int LINES, int RP = 0> void run(const uint8_t *d, unsigned pitch, uint8_t *out, unsigned out_pitch, int same_line)
{
    const int rounds = (int)(pitch / 4 / 1024);
    const size_t lds = 16 * 8192 + 64;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void *)k<LINES, RP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL((k<LINES, RP>), dim3(64, 4), dim3(128), lds, 0, d, pitch, rounds, out, out_pitch, same_line);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL((k<LINES, RP>), dim3(64, 4), dim3(128), lds, 0, d, pitch, rounds, out, out_pitch, same_line);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("reads policy %d, stores: %2d line(s) = %4d bytes per row and event%s: %.3f ms per pass over the 4.92 GB (%.2f TB/s of reads) [%d]\n", RP, LINES, LINES * 128, same_line == 1 ? " (always the same lines: never leave L2)" : same_line == -1 ? " (line major: each instruction one line of 8 rows)" : same_line ? " staged in a scratch, flushed in pieces of" : "",
           ms, 1024.0 * rounds * 4 * 1024 / (ms * 1e-3) / 1e12, same_line);
}
int main()
{
    const unsigned pitch = 4800512, out_pitch = 98304; uint8_t *d, *o;
    if (hipMalloc(&d, (size_t)1024 * pitch + (4 << 20)) != hipSuccess) return 1;
    if (hipMalloc(&o, (size_t)1024 * out_pitch + (80 << 20)) != hipSuccess) return 1;
    hipMemset(d, 1, (size_t)1024 * pitch);
    for (int rep = 0; rep < 1; rep++) {
        run<0>(d, pitch, o, out_pitch, 0);
        run<1>(d, pitch, o, out_pitch, 0); run<1>(d, pitch, o, out_pitch, 1);
        run<8>(d, pitch, o, out_pitch, 0); run<32>(d, pitch, o, out_pitch, 0); run<128>(d, pitch, o, out_pitch, 0);
        run<8>(d, pitch, o, out_pitch, -1); run<32>(d, pitch, o, out_pitch, -1); run<64>(d, pitch, o, out_pitch, -1); run<128>(d, pitch, o, out_pitch, -1);
        run<0, 1>(d, pitch, o, out_pitch, 0); run<1, 1>(d, pitch, o, out_pitch, 0);

    }
    return 0;
}
