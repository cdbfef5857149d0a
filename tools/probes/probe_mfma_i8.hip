// probe_mfma_i8.hip -- (1) determine the A/B/C lane layouts of v_mfma_i32_16x16x64_i8 on gfx950 empirically,
// (2) measure the streaming rate of the B-operand access pattern planned for the MFMA WFM kernel
//     (lane l reads 16 B of stream (l%16) at byte offset base + 64*kstep + 16*(l/16)).
// build: hipcc --offload-arch=gfx950 -O3 -o probe_mfma_i8 probe_mfma_i8.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
#include <string.h>
typedef int v4i __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void k_mfma(const v4i *a, const v4i *b, v4i *d)
{
    v4i acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
    d[threadIdx.x] = acc;
}

// pattern benchmark: each wave owns `groups` groups of 16 streams and walks a time segment
template <int KSTEPS>
__global__ __launch_bounds__(256) void k_pattern(const uint8_t *__restrict__ in, size_t pitch, int n_stream_blocks, long seg_bytes, long tile_stride_bytes, unsigned *__restrict__ sink)
{
    const int wave = (blockIdx.x * 256 + threadIdx.x) / 64;
    const int lane = threadIdx.x & 63;
    const int sb = wave % n_stream_blocks;           // stream block of 64 streams
    const long seg = wave / n_stream_blocks;
    unsigned acc = 0;
    const long t0 = seg * seg_bytes, t1 = t0 + seg_bytes;
    for (long t = t0; t + 64 * KSTEPS <= t1; t += tile_stride_bytes) {
        for (int g = 0; g < 4; g++) {
            const uint8_t *row = in + (size_t)(sb * 64 + g * 16 + (lane & 15)) * pitch + t + 16 * (lane >> 4);
            uint4 v[KSTEPS];
#pragma unroll
            for (int k = 0; k < KSTEPS; k++) v[k] = *reinterpret_cast<const uint4 *>(row + 64 * k);
#pragma unroll
            for (int k = 0; k < KSTEPS; k++) acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main()
{
    // ---------------- layout probe
    std::vector<int8_t> A(16 * 64), B(64 * 16);
    srand(1);
    for (auto &x : A) x = (int8_t)(rand() % 17 - 8);
    for (auto &x : B) x = (int8_t)(rand() % 13 - 6);
    std::vector<int> ref(16 * 16, 0);
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { int s = 0; for (int k = 0; k < 64; k++) s += A[i * 64 + k] * B[k * 16 + j]; ref[i * 16 + j] = s; }
    v4i *da, *db, *dd; CK(hipMalloc(&da, 64 * 16)); CK(hipMalloc(&db, 64 * 16)); CK(hipMalloc(&dd, 64 * 16));
    for (int hyp = 0; hyp < 2; hyp++) {
        // hyp 0: k = 16*(l/16) + byte ; hyp 1: k = 8*(l/16) + (byte%8) + 32*(byte/8)
        std::vector<int8_t> la(64 * 16), lb(64 * 16);
        for (int l = 0; l < 64; l++) for (int b = 0; b < 16; b++) {
            const int k = hyp == 0 ? 16 * (l / 16) + b : 8 * (l / 16) + (b % 8) + 32 * (b / 8);
            la[l * 16 + b] = A[(l % 16) * 64 + k];
            lb[l * 16 + b] = B[k * 16 + (l % 16)];
        }
        CK(hipMemcpy(da, la.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(db, lb.data(), 1024, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, da, db, dd); CK(hipDeviceSynchronize());
        std::vector<int> out(256); CK(hipMemcpy(out.data(), dd, 1024, hipMemcpyDeviceToHost));
        int bad_std = 0, bad_t = 0;
        for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) {
            const int row = 4 * (l / 16) + r, col = l % 16;
            if (out[l * 4 + r] != ref[row * 16 + col]) bad_std++;
            if (out[l * 4 + r] != ref[col * 16 + row]) bad_t++;
        }
        printf("LAYOUT hyp%d: mismatches with C[row=4*(l/16)+r][col=l%%16]: %d ; with transposed: %d\n", hyp, bad_std, bad_t);
    }
    // ---------------- access-pattern bandwidth
    const int S = 1024; const size_t pitch = 2 * 2400256ul;
    uint8_t *din; unsigned *sink; CK(hipMalloc(&din, (size_t)S * pitch)); CK(hipMalloc(&sink, 64)); CK(hipMemset(din, 1, (size_t)S * pitch));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int nseg : {16, 32, 64, 128}) {
        for (int variant = 0; variant < 2; variant++) {
            const int nsb = S / 64; const long seg_bytes = (long)(pitch / nseg) & ~63L;
            const int waves = nsb * nseg; const int blocks = waves / 4;
            const long stride = variant == 0 ? 512 : 400;     // 0: non-overlapping 8 k-steps ; 1: 400-byte tile stride with 512-byte windows (overlap re-reads)
            float best = 1e9;
            for (int rep = 0; rep < 3; rep++) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL((k_pattern<8>), dim3(blocks), dim3(256), 0, 0, din, pitch, nsb, seg_bytes, stride, sink);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            printf("PATTERN nseg=%3d waves=%5d stride=%ld: %.3f ms -> %.1f GB/s of unique input\n", nseg, waves, stride, best, (double)S * pitch / best / 1e6);
        }
    }
    return 0;
}
