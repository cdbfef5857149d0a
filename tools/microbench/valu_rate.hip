// valu_rate.hip -- issue rate of plain vs packed f32 VALU instructions on gfx950 (what the FFT kernels' cost model rests on).
// hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float c2 __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, m = 1.0001f, b = 0.5f;
    c2 p0 = {a0, 1}, p1 = {1, 2}, p2 = {2, 3}, p3 = {3, 4}, p4 = {4, 5}, p5 = {5, 6}, p6 = {6, 7}, p7 = {7, 8}, pm = {1.0001f, 0.9999f}, pb = {0.5f, 0.25f};
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) { REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(b));) }
        if (KIND == 1) { REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm), "v"(pb));) }
        if (KIND == 2) { REP8(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(b));) }
        if (KIND == 3) { REP8(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm), "v"(pb));) }
        if (KIND == 4) { REP8(asm volatile("v_pk_add_f32 %0, %0, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %1, %1, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %2, %2, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %3, %3, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %4, %4, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %5, %5, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %6, %6, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %7, %7, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm), "v"(pb));) }
        if (KIND == 5) { REP8(asm volatile("v_pk_mul_f32 %0, %0, %8 op_sel:[0,0] op_sel_hi:[0,1]\n v_pk_mul_f32 %1, %1, %8 op_sel:[0,0] op_sel_hi:[0,1]\n v_pk_mul_f32 %2, %2, %8 op_sel:[0,0] op_sel_hi:[0,1]\n v_pk_mul_f32 %3, %3, %8 op_sel:[0,0] op_sel_hi:[0,1]\n v_pk_mul_f32 %4, %4, %8 op_sel:[0,0] op_sel_hi:[0,1]\n v_pk_mul_f32 %5, %5, %8 op_sel:[0,0] op_sel_hi:[0,1]\n v_pk_mul_f32 %6, %6, %8 op_sel:[0,0] op_sel_hi:[0,1]\n v_pk_mul_f32 %7, %7, %8 op_sel:[0,0] op_sel_hi:[0,1]" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm), "v"(pb));) }
        if (KIND == 6) { REP8(asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(b));) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}
template <int KIND> void run(const char *name, float *d, int wg_per_cu)
{
    const int iters = 2000, cus = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<cus * wg_per_cu, 256>>>(d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<KIND><<<cus * wg_per_cu, 256>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // each workgroup = 4 waves, one per SIMD; wg_per_cu waves per SIMD; instructions per wave = iters * 64
    const double inst_per_simd = (double)iters * 64 * wg_per_cu;
    printf("%-34s waves/SIMD %d: %.3f ms, %.2f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz)\n", name, wg_per_cu, ms, ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4);
}
int main()
{
    float *d; hipMalloc(&d, 256 * 256 * 8 * 4);
    for (int w : {1, 2, 4}) {
        if (w == 1) { run<0>("v_fma_f32", d, 1); run<1>("v_pk_fma_f32", d, 1); run<2>("v_add_f32", d, 1); run<3>("v_pk_add_f32", d, 1); run<4>("v_pk_add_f32 op_sel/neg", d, 1); run<5>("v_pk_mul_f32 op_sel", d, 1); run<6>("v_mov_b32", d, 1); }
        if (w == 2) { run<0>("v_fma_f32", d, 2); run<1>("v_pk_fma_f32", d, 2); run<3>("v_pk_add_f32", d, 2); }
        if (w == 4) { run<0>("v_fma_f32", d, 4); run<1>("v_pk_fma_f32", d, 4); run<3>("v_pk_add_f32", d, 4); run<6>("v_mov_b32", d, 4); }
    }
    return 0;
}
