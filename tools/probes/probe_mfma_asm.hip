// probe_mfma_asm.hip -- does an inline-asm v_mfma_i32_16x16x64_i8 with the A operand in AGPRs ("a" constraint) and a chain of
// accumulations give the same result as the builtin?  (debugging aid for csdr_amd/csrc/wfm_mfma.hip)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void k_builtin(const v4i *a, const v4i *b, v4i *d, int n)
{
    v4i acc = {0, 0, 0, 0};
    for (int k = 0; k < n; k++) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[k * 64 + threadIdx.x], b[k * 64 + threadIdx.x] ^ (int)0x80808080, acc, 0, 0, 0);
    d[threadIdx.x] = acc;
}
template <int N>
__global__ void k_asm(const v4i *a, const v4i *b, v4i *d)
{
    v4i A[N], B[N];
#pragma unroll
    for (int k = 0; k < N; k++) { A[k] = a[k * 64 + threadIdx.x]; B[k] = b[k * 64 + threadIdx.x]; }
#pragma unroll
    for (int k = 0; k < N; k++) B[k] ^= (int)0x80808080;
    v4i acc;
    asm volatile("s_nop 1" : "+v"(B[0]), "+v"(B[1]), "+v"(B[2]), "+v"(B[3]));
    asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, 0" : "=&v"(acc) : "a"(A[0]), "v"(B[0]));
#pragma unroll
    for (int k = 1; k < N; k++) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc) : "a"(A[k]), "v"(B[k]));
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc));
    d[threadIdx.x] = acc;
}
int main()
{
    const int N = 4;
    std::vector<int> ha(N * 64 * 4), hb(N * 64 * 4);
    srand(3);
    for (auto &x : ha) x = rand() ^ (rand() << 16);
    for (auto &x : hb) x = rand() ^ (rand() << 16);
    v4i *da, *db, *d1, *d2; CK(hipMalloc(&da, ha.size() * 4)); CK(hipMalloc(&db, hb.size() * 4)); CK(hipMalloc(&d1, 1024)); CK(hipMalloc(&d2, 1024));
    CK(hipMemcpy(da, ha.data(), ha.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_builtin, dim3(1), dim3(64), 0, 0, da, db, d1, N);
    hipLaunchKernelGGL((k_asm<N>), dim3(1), dim3(64), 0, 0, da, db, d2);
    CK(hipDeviceSynchronize());
    std::vector<int> o1(256), o2(256); CK(hipMemcpy(o1.data(), d1, 1024, hipMemcpyDeviceToHost)); CK(hipMemcpy(o2.data(), d2, 1024, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 256; i++) bad += o1[i] != o2[i];
    printf("ASM-vs-builtin mismatches: %d of 256 (sample %d vs %d)\n", bad, o1[5], o2[5]);
    return 0;
}
