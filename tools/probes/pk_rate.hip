// tools/probes/pk_rate.hip -- issue rate of packed f32 VALU on gfx950: cycles per instruction of v_add_f32 / v_fma_f32 / v_pk_add_f32 / v_pk_fma_f32 / v_pk_mul_f32
// for 1, 2 and 4 waves per SIMD (independent accumulators, no memory traffic).   hipcc --offload-arch=gfx950 -O3 pk_rate.hip -o pk_rate && ./pk_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP 64
template <int KIND> __global__ __launch_bounds__(1024) void k(float *out, long long *cyc, int iters)
{
    f2 a[8]; f2 b = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
    for (int j = 0; j < 8; j++) a[j] = f2{(float)threadIdx.x + j, (float)j};
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP; r++) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if (KIND == 0) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[j].x) : "v"(b.x));
                if (KIND == 1) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(a[j].x) : "v"(b.x), "v"(c.x));
                if (KIND == 2) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a[j]) : "v"(b));
                if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(a[j]) : "v"(b), "v"(c));
                if (KIND == 4) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(a[j]) : "v"(b));
                if (KIND == 5) asm volatile("v_pk_add_f32 %0, %1, %0 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0]" : "+v"(a[j]) : "v"(b));
                if (KIND == 6) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[j].x) : "v"(b.x));
            }
        }
    }
    const long long t1 = clock64();
    float s = 0; for (int j = 0; j < 8; j++) s += a[j].x + a[j].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int KIND> void run(const char *name)
{
    float *out; long long *cyc; hipMalloc(&out, 16 * 1024 * 1024); hipMalloc(&cyc, 8 * 1024);
    for (int cfg = 0; cfg < 6; cfg++) {
        const int threads = cfg < 3 ? (256 << cfg) : 256, blocks = cfg < 3 ? 256 : (256 << (cfg - 2));
        const int iters = 200;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<KIND><<<blocks, threads>>>(out, cyc, 10); hipDeviceSynchronize();
        hipEventRecord(e0); k<KIND><<<blocks, threads>>>(out, cyc, iters); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c[1]; hipMemcpy(c, cyc, 8, hipMemcpyDeviceToHost);
        const int wps = threads / 256 * (blocks / 256);
        const double insts_per_simd = (double)iters * REP * 8 * wps, own = (double)iters * REP * 8;
        printf("%-14s %4d blocks x %4d threads = %d waves/SIMD: %.3f ms, %.2f ns per wave-instruction per SIMD, wave 0's clock64 ticks per own instruction %.3f\n", name, blocks, threads, wps, ms, ms * 1e6 / insts_per_simd, (double)c[0] / own);
    }
}
int main() { run<0>("v_add_f32"); run<1>("v_fma_f32"); run<2>("v_pk_add_f32"); run<3>("v_pk_fma_f32"); return 0; }
