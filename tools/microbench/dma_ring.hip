// dma_ring.hip -- what the memory system gives the chain kernels' access pattern: R rows (streams) per workgroup, every row read front to back in runs
// of RUN KiB by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction) into a per-row LDS ring DEPTH runs deep, one barrier per step -- the skeleton of
// k_wfm_mfma_seq / k_ddc_mfma without any arithmetic.  Sweeps rows per workgroup, run length, ring depth, workgroups per CU, segments per row, cache policy,
// and a contiguous-region stream as the reference point.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/dma_ring.hip -o /tmp/dma_ring && /tmp/dma_ring
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
#include <algorithm>

struct P {
    const uint8_t *in; unsigned long long pitch; long long seg_bytes; int R, run_kib, depth, barrier, nt, contig, consume;
};

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__device__ __forceinline__ void wait_vm_n(int n)
{
#define C(k) case k: wait_vm<k>(); break;
    switch (n) {
        C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15) C(16) C(17) C(18) C(19) C(20) C(21) C(22) C(23) C(24) C(25) C(26) C(27)
        C(28) C(29) C(30) C(31) C(32) C(33) C(34) C(35) C(36) C(37) C(38) C(39) C(40) C(41) C(42) C(43) C(44) C(45) C(46) C(47) C(48) C(49) C(50) C(51) C(52) C(53)
        C(54) C(55) C(56) C(57) C(58) C(59) C(60) C(61) C(62)
        default: wait_vm<63>(); break;
    }
#undef C
}

template <bool NT>
__device__ __forceinline__ void dma_1k(const uint8_t *sbase, uint32_t voff, uint32_t lds_addr)
{
    uint32_t keep;
    if (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}

__device__ unsigned long long g_issue_cycles, g_issue_count;

template <int W, bool NT>
__global__ __launch_bounds__(64 * W) void k_ring(P p, unsigned *sink)
{
    long long t_issue = 0, n_issue = 0;
    extern __shared__ float4 raw[];
    uint8_t *lds = reinterpret_cast<uint8_t *>(raw);
    const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int rpw = p.R / W, run = p.run_kib * 1024, ring = run * p.depth;
    const uint32_t lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) uint8_t *)lds;
    const long long n_steps = p.seg_bytes / run;
    // row r of this workgroup: stream block blockIdx.x, segment blockIdx.y
    const uint8_t *base;
    unsigned long long row_pitch, step_adv;
    if (p.contig) {            // the workgroup streams ONE contiguous region: step s = R * run bytes, row r = the r-th run of it
        base = p.in + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (size_t)p.R * p.seg_bytes;
        row_pitch = run; step_adv = (unsigned long long)p.R * run;
    } else {
        base = p.in + (size_t)blockIdx.x * p.R * p.pitch + (size_t)blockIdx.y * p.seg_bytes;
        row_pitch = p.pitch; step_adv = run;
    }
    const int per_step = rpw * p.run_kib;                       // loads per wave per step
    const int ahead = p.depth - 1;                              // steps in flight behind the one waited for
    unsigned acc = 0;
    long long issued = 0;
    auto issue = [&](long long s) {
        const long long t_in = __builtin_readcyclecounter();
        const uint8_t *sb = base + s * step_adv;
        const int slot = (int)(s % p.depth) * run;
        for (int r = 0; r < rpw; r++) {
            const int row = wv * rpw + r;
            for (int k = 0; k < p.run_kib; k++) {
                const uint32_t la = __builtin_amdgcn_readfirstlane((int)(lds_base + row * ring + slot + k * 1024));
                dma_1k<NT>(sb, (uint32_t)(row * row_pitch + k * 1024 + 16 * lane), la);
            }
        }
        t_issue += __builtin_readcyclecounter() - t_in; n_issue += per_step;
    };
    for (; issued < n_steps && issued < ahead; issued++) issue(issued);
    for (long long s = 0; s < n_steps; s++) {
        if (issued < n_steps) { issue(issued); issued++; }
        const long long newer = issued - s - 1;                 // steps issued after step s
        wait_vm_n((int)(newer * per_step));
        if (p.barrier) __syncthreads();
        if (p.consume) {                                        // every wave reads its rows' landed run (what a consumer at least does)
            const int slot = (int)(s % p.depth) * run;
            for (int r = 0; r < rpw; r++)
                for (int k = 0; k < p.run_kib; k++) {
                    const uint4 v = *reinterpret_cast<const uint4 *>(lds + (wv * rpw + r) * ring + slot + k * 1024 + 16 * lane);
                    acc ^= v.x ^ v.y ^ v.z ^ v.w;
                }
            if (p.barrier) __syncthreads();                     // (the slot is refilled by the next issue)
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
    if (lane == 0) { atomicAdd(&g_issue_cycles, (unsigned long long)t_issue); atomicAdd(&g_issue_count, (unsigned long long)n_issue); }
}

struct Cfg { const char *name; int W, R, run_kib, depth, barrier, nt, contig, consume, gx, gy; };

int main(int argc, char **argv)
{
    const long long row_bytes = 4800512;                       // 2 x 2400256: a 1-s block of a 2.4 MS/s u8 IQ stream (bench.py)
    const int n_rows = 1024;
    uint8_t *d; unsigned *sink;
    if (hipMalloc(&d, (size_t)n_rows * row_bytes + (1 << 20)) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&sink, 64);
    hipMemset(d, 1, (size_t)n_rows * row_bytes);
    std::vector<Cfg> cfgs = {
        // name                                   W  R  run depth bar nt contig consume gx gy
        {"wfm-like: 16 rows x 1 KiB, depth 8",    8, 16, 1, 8, 1, 1, 0, 0, 64, 4},
        {"  + consumer reads",                    8, 16, 1, 8, 1, 1, 0, 1, 64, 4},
        {"  no barrier",                          8, 16, 1, 8, 0, 1, 0, 0, 64, 4},
        {"  default cache policy",                8, 16, 1, 8, 1, 0, 0, 0, 64, 4},
        {"  depth 6",                             8, 16, 1, 6, 1, 1, 0, 0, 64, 4},
        {"  depth 4",                             8, 16, 1, 4, 1, 1, 0, 0, 64, 4},
        {"  runs of 2 KiB, depth 4",              8, 16, 2, 4, 1, 1, 0, 0, 64, 4},
        {"  runs of 4 KiB, depth 2",              8, 16, 4, 2, 1, 1, 0, 0, 64, 4},
        {"  8 segments (2 wg/cu would need lds)", 8, 16, 1, 8, 1, 1, 0, 0, 64, 8},
        {"  2 segments (half the CUs)",           8, 16, 1, 8, 1, 1, 0, 0, 64, 2},
        {"8 rows x 1 KiB depth 8, 2 wg/cu",       8, 8,  1, 8, 1, 1, 0, 0, 128, 4},
        {"8 rows x 2 KiB depth 4, 2 wg/cu",       8, 8,  2, 4, 1, 1, 0, 0, 128, 4},
        {"8 rows x 2 KiB depth 8, 1 wg/cu",       8, 8,  2, 8, 1, 1, 0, 0, 128, 2},
        {"4 rows x 4 KiB depth 8, 1 wg/cu (4 w)", 4, 4,  4, 8, 1, 1, 0, 0, 256, 1},
        {"4 rows x 4 KiB depth 4, 2 wg/cu (4 w)", 4, 4,  4, 4, 1, 1, 0, 0, 256, 2},
        {"32 rows x 1 KiB depth 4",               8, 32, 1, 4, 1, 1, 0, 0, 32, 8},
        {"contiguous 16 KiB per step, depth 8",   8, 16, 1, 8, 1, 1, 1, 0, 64, 4},
        {"contiguous, no barrier",                8, 16, 1, 8, 0, 1, 1, 0, 64, 4},
        {"contiguous, default policy",            8, 16, 1, 8, 1, 0, 1, 0, 64, 4},
        {"contiguous 4 waves 16 KiB depth 4 x2",  4, 16, 1, 4, 1, 1, 1, 0, 128, 4},
        {"2 waves fetch 16 rows x 1 KiB, depth 6",2, 16, 1, 6, 1, 1, 0, 0, 64, 4},
        {"  default policy",                      2, 16, 1, 6, 1, 0, 0, 0, 64, 4},
        {"  runs of 2 KiB, depth 3",              2, 16, 2, 3, 1, 1, 0, 0, 64, 4},
        {"  runs of 4 KiB, depth 2",              2, 16, 4, 2, 1, 1, 0, 0, 64, 4},
        {"  contiguous",                          2, 16, 1, 6, 1, 1, 1, 0, 64, 4},
        {"4 waves fetch 16 rows x 1 KiB, depth 6",4, 16, 1, 6, 1, 1, 0, 0, 64, 4},
        {"1 wave fetches 16 rows x 1 KiB, depth 4",1, 16, 1, 4, 1, 1, 0, 0, 64, 4},
        {"1 wave, contiguous",                    1, 16, 1, 4, 1, 1, 1, 0, 64, 4},
        {"2 waves, 16 rows x 1 KiB, depth 2",     2, 16, 1, 2, 1, 1, 0, 0, 64, 4},
        {"2 waves, 16 rows x 1 KiB, depth 3",     2, 16, 1, 3, 1, 1, 0, 0, 64, 4},
        {"2 waves, 16 rows x 1 KiB, depth 4",     2, 16, 1, 4, 1, 1, 0, 0, 64, 4},
        {"2 waves, 16 rows x 1 KiB, depth 5",     2, 16, 1, 5, 1, 1, 0, 0, 64, 4},
        {"2 waves, 16 rows x 1 KiB, depth 8",     2, 16, 1, 8, 1, 1, 0, 0, 64, 4},
        {"2 waves, 16 rows x 2 KiB, depth 2",     2, 16, 2, 2, 1, 1, 0, 0, 64, 4},
        {"2 waves, 16 rows x 2 KiB, depth 4",     2, 16, 2, 4, 1, 1, 0, 0, 64, 4},
        {"2 waves, 16 rows x 3 KiB, depth 2",     2, 16, 3, 2, 1, 1, 0, 0, 64, 4},
        {"2 waves, 16 rows x 3 KiB, depth 3",     2, 16, 3, 3, 1, 1, 0, 0, 64, 4},
        {"1 wave, 16 rows x 1 KiB, depth 2",      1, 16, 1, 2, 1, 1, 0, 0, 64, 4},
        {"1 wave, 16 rows x 1 KiB, depth 3",      1, 16, 1, 3, 1, 1, 0, 0, 64, 4},
        {"1 wave, 16 rows x 2 KiB, depth 2",      1, 16, 2, 2, 1, 1, 0, 0, 64, 4},
        {"1 wave, 16 rows x 2 KiB, depth 3",      1, 16, 2, 3, 1, 1, 0, 0, 64, 4},
        {"4 waves, 16 rows x 1 KiB, depth 3",     4, 16, 1, 3, 1, 1, 0, 0, 64, 4},
        {"4 waves, 16 rows x 2 KiB, depth 2",     4, 16, 2, 2, 1, 1, 0, 0, 64, 4},
    };
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    printf("%-44s %8s %8s %8s\n", "pattern (1024 rows x 4.8 MB = 4.92 GB)", "best ms", "med ms", "TB/s(med)");
    for (int pass = 0; pass < 2; pass++)
        for (const Cfg &c : cfgs) {
            P p{d, (unsigned long long)row_bytes, 0, c.R, c.run_kib, c.depth, c.barrier, c.nt, c.contig, c.consume};
            p.seg_bytes = (c.contig ? (long long)n_rows * row_bytes / ((long long)c.gx * c.gy * c.R) : row_bytes / c.gy) / (c.run_kib * 1024) * (c.run_kib * 1024);
            if (!c.contig && c.gx * c.R != n_rows) { printf("%-44s bad geometry\n", c.name); continue; }
            const size_t lds = (size_t)c.R * c.run_kib * 1024 * c.depth;
            const double bytes = (double)c.gx * c.gy * c.R * (double)(p.seg_bytes / (c.run_kib * 1024)) * c.run_kib * 1024;
            auto launch = [&]() {
#define L(WW, NTT) { hipFuncSetAttribute((const void *)k_ring<WW, NTT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                     hipLaunchKernelGGL((k_ring<WW, NTT>), dim3(c.gx, c.gy), dim3(64 * WW), lds, 0, p, sink); }
                if (c.W == 8) { if (c.nt) L(8, true) else L(8, false) } else if (c.W == 4) { if (c.nt) L(4, true) else L(4, false) }
                else if (c.W == 2) { if (c.nt) L(2, true) else L(2, false) } else { if (c.nt) L(1, true) else L(1, false) }
#undef L
            };
            for (int i = 0; i < 30; i++) launch();             // clocks
            hipDeviceSynchronize();
            std::vector<float> ts;
            for (int i = 0; i < reps; i++) {
                hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms);
            }
            if (hipGetLastError() != hipSuccess) { printf("%-44s launch error\n", c.name); continue; }
            std::sort(ts.begin(), ts.end());
            unsigned long long cyc = 0, cnt = 0, zero = 0;
            hipMemcpyFromSymbol(&cyc, HIP_SYMBOL(g_issue_cycles), 8); hipMemcpyFromSymbol(&cnt, HIP_SYMBOL(g_issue_count), 8);
            hipMemcpyToSymbol(HIP_SYMBOL(g_issue_cycles), &zero, 8); hipMemcpyToSymbol(HIP_SYMBOL(g_issue_count), &zero, 8);
            if (pass == 1) printf("%-44s %8.4f %8.4f %8.2f   (lds %zu KiB, %.2f GB)  issue: %.0f cycles per 1-KiB piece and wave\n", c.name, ts[0], ts[ts.size() / 2],
                                  bytes / (ts[ts.size() / 2] * 1e-3) / 1e12, lds / 1024, bytes / 1e9, cnt ? (double)cyc / cnt : 0.0);
        }
    return 0;
}
