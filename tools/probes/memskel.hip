// tools/probes/memskel.hip -- the memory skeleton of a wave-per-window overlap-save filter: every wave reads a 32-KiB window (stride 24 KiB) and writes 24 KiB, nothing else.
// Variants: bytes per lane and instruction (8 / 16), instructions per batch between waits, waves per workgroup.   hipcc --offload-arch=gfx950 -O3 memskel.hip -o memskel
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));
__device__ f2 bl2(i4 rsrc, int voff, int soff, int aux) __asm("llvm.amdgcn.raw.buffer.load.v2f32");
__device__ void bs2(f2 v, i4 rsrc, int voff, int soff, int aux) __asm("llvm.amdgcn.raw.buffer.store.v2f32");
__device__ f4 bl4(i4 rsrc, int voff, int soff, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ void bs4(f4 v, i4 rsrc, int voff, int soff, int aux) __asm("llvm.amdgcn.raw.buffer.store.v4f32");

// MODE 0: 64 x b64 loads, wait, 64 x b64 stores (48 useful);  1: b128 (32 loads, 24 stores);  2: b64 in four batches of 16 loads / 16 stores;  3: b64, loads only;  4: b64, stores only
template <int MODE, int WAVES> __global__ __launch_bounds__(64 * WAVES, 8 / WAVES) void k(const float *__restrict__ in, float *__restrict__ out, int n_windows, long in_bytes)
{
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned long long bi = (unsigned long long)in, bo = (unsigned long long)out;
#ifdef INTERLEAVED
    const int w_end = n_windows, stride = gridDim.x * WAVES;            // consecutive windows on consecutive workgroups (= round robin over the XCDs)
    for (int w = blockIdx.x * WAVES + wv; w < w_end; w += stride) {
#else
    const int per_xcd = (n_windows + 7) >> 3, xcd = blockIdx.x & 7, stride = (gridDim.x >> 3) * WAVES;
    const int w_end = min(n_windows, (xcd + 1) * per_xcd);
    for (int w = xcd * per_xcd + (blockIdx.x >> 3) * WAVES + wv; w < w_end; w += stride) {
#endif
        const unsigned long long a = bi + (unsigned long long)w * 24576, b = bo + (unsigned long long)w * 24576;
        const i4 rx = {(int)(unsigned)a, (int)((a >> 32) & 0xffffu), 32768, 0x00020000};
        const i4 ry = {(int)(unsigned)b, (int)((b >> 32) & 0xffffu), 24576, 0x00020000};
        if (MODE == 1) {
            f4 v[32];
#pragma unroll
            for (int j = 0; j < 32; j++) v[j] = bl4(rx, lane * 16 + 1024 * j, 0, 0);
#pragma unroll
            for (int j = 0; j < 32; j++) bs4(v[j], ry, lane * 16 + 1024 * j - 8192, 0, 0);
        } else if (MODE == 2) {
#pragma unroll
            for (int g = 0; g < 4; g++) {
                f2 v[16];
#pragma unroll
                for (int j = 0; j < 16; j++) v[j] = bl2(rx, lane * 8 + 512 * (16 * g + j), 0, 0);
#pragma unroll
                for (int j = 0; j < 16; j++) bs2(v[j], ry, lane * 8 + 512 * (16 * g + j) - 8192, 0, 0);
            }
        } else {
            f2 v[64];
#pragma unroll
            for (int j = 0; j < 64; j++) v[j] = MODE == 4 ? f2{(float)j, (float)w} : bl2(rx, lane * 8 + 512 * j, 0, 0);
            if (MODE == 3) {
                f2 s = {0, 0};
#pragma unroll
                for (int j = 0; j < 64; j++) s += v[j];
                if (s.x == 12345.f) bs2(s, ry, lane * 8, 0, 0);
            } else {
#pragma unroll
                for (int j = 0; j < 64; j++) bs2(v[j], ry, lane * 8 + 512 * j - 8192, 0, 0);
            }
        }
    }
}
template <int MODE, int WAVES> void run(const char *name, float *in, float *out, int n_windows, long in_bytes)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * (8 / WAVES);
    for (int i = 0; i < 3; i++) k<MODE, WAVES><<<grid, 64 * WAVES>>>(in, out, n_windows, in_bytes);
    hipEventRecord(e0);
    for (int i = 0; i < 20; i++) k<MODE, WAVES><<<grid, 64 * WAVES>>>(in, out, n_windows, in_bytes);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    const double alg = (double)n_windows * 24576 * 2;
    printf("%-52s %d waves/wg: %.4f ms, %.2f TB/s algorithmic (%.2f of 8)\n", name, WAVES, ms, alg / ms / 1e9, alg / ms / 1e9 / 8);
}
int main()
{
    const int n_windows = 64 * 336;                                    // bench_fftfilt's 64 streams x 16 blocks at 1023 taps
    const long in_bytes = (long)n_windows * 24576 + 65536;
    float *in, *out; hipMalloc(&in, in_bytes); hipMalloc(&out, in_bytes); hipMemset(in, 0, in_bytes);
    run<0, 4>("b64: 64 loads, wait, 64 stores", in, out, n_windows, in_bytes);
    run<0, 8>("b64: 64 loads, wait, 64 stores", in, out, n_windows, in_bytes);
    run<0, 1>("b64: 64 loads, wait, 64 stores", in, out, n_windows, in_bytes);
    run<1, 4>("b128: 32 loads, wait, 32 stores", in, out, n_windows, in_bytes);
    run<2, 4>("b64: four batches of 16 loads + 16 stores", in, out, n_windows, in_bytes);
    run<3, 4>("b64: loads only", in, out, n_windows, in_bytes);
    run<4, 4>("b64: stores only", in, out, n_windows, in_bytes);
    return 0;
}
