// tools/probes/icache.hip -- does a long straight-line loop body run from the instruction cache?  Waves (two per SIMD, staggered so that they are at different places of
// the body) execute a loop whose body is N x 8 bytes of independent v_pk_add_f32; ns per instruction and SIMD against the body's size.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int KB> __global__ __launch_bounds__(512) void k(float *out, int iters, int stagger)
{
    f2 a[8]; f2 b = {1.0001f, 0.9999f};
    for (int j = 0; j < 8; j++) a[j] = f2{(float)threadIdx.x + j, (float)j};
    const int wave = (blockIdx.x * 8 + (threadIdx.x >> 6));
    const int pre = __builtin_amdgcn_readfirstlane((wave * 37) % 61) * stagger;
    for (int i = 0; i < pre; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a[j]) : "v"(b));
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < KB * 16; r++) {                             // 16 x 8 instructions x 8 bytes = 1 KiB
#pragma unroll
            for (int j = 0; j < 8; j++) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a[j]) : "v"(b));
        }
    }
    float s = 0; for (int j = 0; j < 8; j++) s += a[j].x + a[j].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KB> void run(float *out, int stagger)
{
    const int iters = 4096 / KB;                                        // the same instruction count for every size
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KB><<<256, 512>>>(out, 2, stagger); hipDeviceSynchronize();
    hipEventRecord(e0); k<KB><<<256, 512>>>(out, iters, stagger); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)iters * KB * 128 * 2;
    printf("body %3d KiB, stagger %d: %.3f ms, %.2f ns per packed instruction and SIMD\n", KB, stagger, ms, ms * 1e6 / insts_per_simd);
}
int main()
{
    float *out; hipMalloc(&out, 4 << 20);
    for (int stagger : {0, 40}) { run<4>(out, stagger); run<16>(out, stagger); run<24>(out, stagger); run<32>(out, stagger); run<48>(out, stagger); run<64>(out, stagger); run<96>(out, stagger); run<128>(out, stagger); }
    return 0;
}
