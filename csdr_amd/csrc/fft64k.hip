// fft64k.hip -- the FFT stages of bandpass_fir_fft_cc / apply_fir_fft_cc (libcsdr.c:814-849) at fft_size = 65536 (BASELINE config 3)
// as THREE passes over the data instead of hipFFT's multi-kernel plans with separate framing and product kernels:
//
//   four-step decomposition N = 256 x 256, n = 256 n1 + n2, k = k1 + 256 k2:
//     K1  columns : T[k1][n2] = W_N^(n2 k1) * sum_n1 x[256 n1 + n2] W_256^(n1 k1)        (reads the input block directly: the zero padding of
//                                                                                         csdr.c:1864 is a predicate, not a copied frame)
//     K2  rows    : X[k1 + 256 k2] = sum_n2 T[k1][n2] W_256^(n2 k2);   Y = X * taps_fft   (libcsdr.c:826-830; taps_fft kept in the same
//                   [k1][k2] order);   U[k1][n2] = W_N^(-n2 k1) * sum_k2 Y[k1 + 256 k2] W_256^(-n2 k2)
//                   -- forward row transform, bin product and inverse row transform of the SAME 256 elements in one kernel: the spectrum
//                   never goes to memory;
//     K3  columns : y[256 n1 + n2] = sum_k1 U[k1][n2] W_256^(-n1 k1)                       (unnormalised, natural order: what the overlap-add
//                                                                                         stitch of fftpath.hip expects)
//   HBM traffic per FFT point: 8 B in + 3 x 16 B, against ~150 B for frame + 2 x (2-3 hipFFT kernels) + product.
//
// A 256-point transform is done by 16 threads: two radix-16 butterflies in registers (dft16: 4 x 4) with one exchange through LDS
// ([k1][n2] layout, pitch 17, conflict free); a workgroup of 256 threads transforms 16 columns (K1, K3: 128-byte global segments per
// row) or 16 rows (K2) at once.
#include "common.hpp"
#include "fft_butterflies.hpp"
#include <math.h>
#include <vector>
using namespace csdr_amd;

namespace {

constexpr int F64_N = 65536;
constexpr int F64_P = 273;                 // LDS pitch of one 256-point transform (>= 17 * 16, odd: column-major fills are conflict free)

// 256-point transform of transform f (of 16 in the workgroup) by its 16 threads j = 0..15, in place in buf ([16][F64_P], natural order in
// and out; the exchange between the two radix-16 stages uses the same rows in [k1][n2] layout with pitch 17).  tw256: exp(-2 pi i m / 256) in
// LDS.  All 256 threads of the workgroup must call it (barriers).  One buffer instead of two: 39 KB of LDS per workgroup = 4 workgroups per CU
// (with a separate exchange buffer: 2 per CU, and the global-memory latency of a workgroup's load / store phases was not hidden: 2.9 TB/s).
template <bool INV>
__device__ __forceinline__ void fft256(float2 *buf, int f, int j, const float2 *tw256)
{
    float2 v[16];
#pragma unroll
    for (int n1 = 0; n1 < 16; n1++) v[n1] = buf[f * F64_P + 16 * n1 + j];                     // n = 16 n1 + n2, thread j = n2
    dft16<INV>(v);
    __syncthreads();
#pragma unroll
    for (int k1 = 0; k1 < 16; k1++) {
        float2 w = tw256[(j * k1) & 255]; if (INV) w.y = -w.y;
        buf[f * F64_P + 17 * k1 + j] = cmul(v[k1], w);
    }
    __syncthreads();
#pragma unroll
    for (int n2 = 0; n2 < 16; n2++) v[n2] = buf[f * F64_P + 17 * j + n2];                     // thread j = k1
    dft16<INV>(v);
    __syncthreads();
#pragma unroll
    for (int k2 = 0; k2 < 16; k2++) buf[f * F64_P + j + 16 * k2] = v[k2];                     // k = k1 + 16 k2
    __syncthreads();
}

// W_65536^m = tw256[m >> 8] * twlo[m & 255]
__device__ __forceinline__ float2 twiddle_n(const float2 *tw256, const float2 *twlo, int m, bool inv)
{
    float2 w = cmul(tw256[(m >> 8) & 255], twlo[m & 255]);
    if (inv) w.y = -w.y;
    return w;
}

// dynamic LDS: buf, the two twiddle tables
#define F64_LDS extern __shared__ float4 f64_lds_raw[]; float2 *buf = reinterpret_cast<float2 *>(f64_lds_raw), *tw256 = buf + 16 * F64_P, *twlo = tw256 + 256
constexpr size_t F64_LDS_BYTES = (size_t)(16 * F64_P + 512) * sizeof(float2);

__device__ __forceinline__ void load_tables(float2 *tw256, float2 *twlo, const float2 *g_tw)
{
    tw256[threadIdx.x] = g_tw[threadIdx.x]; twlo[threadIdx.x] = g_tw[256 + threadIdx.x];
}

// K1: grid (16 column blocks, n_blocks, n_streams); block 256
__global__ __launch_bounds__(256) void k_f64_cols_fwd(const cf32 *__restrict__ in, size_t in_pitch, int inp, int n_blocks, float2 *__restrict__ T, const float2 *__restrict__ g_tw)
{
    F64_LDS;
    load_tables(tw256, twlo, g_tw);
    const int t = threadIdx.x, c = t & 15, r0 = t >> 4;
    const int cb = blockIdx.x; const size_t s = blockIdx.z, b = blockIdx.y;
    const float2 *x = reinterpret_cast<const float2 *>(in) + s * in_pitch + b * (size_t)inp;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int r = r0 + 16 * i, n = 256 * r + 16 * cb + c;
        buf[c * F64_P + r] = n < inp ? x[n] : make_float2(0.f, 0.f);                           // csdr.c:1864: the block is zero padded to fft_size
    }
    __syncthreads();
    fft256<false>(buf, t >> 4, t & 15, tw256);
    float2 *dst = T + (s * n_blocks + b) * (size_t)F64_N;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int k1 = r0 + 16 * i, n2 = 16 * cb + c;
        dst[256 * k1 + n2] = cmul(buf[c * F64_P + k1], twiddle_n(tw256, twlo, n2 * k1, false));
    }
}

// K2: grid (16 row blocks, batch); rows k1 = 16 rb .. +15
__global__ __launch_bounds__(256) void k_f64_rows(float2 *__restrict__ T, const float2 *__restrict__ Ht, const float2 *__restrict__ g_tw)
{
    F64_LDS;
    load_tables(tw256, twlo, g_tw);
    const int t = threadIdx.x, rb = blockIdx.x;
    float2 *base = T + (size_t)blockIdx.y * F64_N + (size_t)(16 * rb) * 256;
#pragma unroll
    for (int i = 0; i < 16; i++) buf[i * F64_P + t] = base[256 * i + t];
    __syncthreads();
    fft256<false>(buf, t >> 4, t & 15, tw256);
    const float2 *h = Ht + (size_t)(16 * rb) * 256;
#pragma unroll
    for (int i = 0; i < 16; i++) buf[i * F64_P + t] = cmul(buf[i * F64_P + t], h[256 * i + t]);   // X[k1 + 256 k2] * taps_fft[k1 + 256 k2], libcsdr.c:826-830
    __syncthreads();
    fft256<true>(buf, t >> 4, t & 15, tw256);
#pragma unroll
    for (int i = 0; i < 16; i++) base[256 * i + t] = cmul(buf[i * F64_P + t], twiddle_n(tw256, twlo, t * (16 * rb + i), true));
}

// K3 with the overlap-add fused (overlap <= input_size: every output position has at most two contributions): the block's own samples go
// straight to `out` (scaled by 1/N, libcsdr.c:836-839), its tail (the last taps-1 samples) to tails[batch][ovl]; k_f64_tail_add then adds
// the previous block's tail (or the carry from the previous call) to the first `ovl` outputs of every block (libcsdr.c:843-847).
__global__ __launch_bounds__(256) void k_f64_cols_inv_oa(const float2 *__restrict__ U, float2 *__restrict__ out, size_t out_pitch, float2 *__restrict__ tails,
                                                         int inp, int ovl, int n_blocks, float inv_n, const float2 *__restrict__ g_tw)
{
    F64_LDS;
    load_tables(tw256, twlo, g_tw);
    const int t = threadIdx.x, c = t & 15, r0 = t >> 4, cb = blockIdx.x;
    const size_t batch = blockIdx.y, s = batch / n_blocks, b = batch % n_blocks;
    const float2 *src = U + batch * F64_N;
#pragma unroll
    for (int i = 0; i < 16; i++) { const int k1 = r0 + 16 * i; buf[c * F64_P + k1] = src[256 * k1 + 16 * cb + c]; }
    __syncthreads();
    fft256<true>(buf, t >> 4, t & 15, tw256);
    float2 *o = out + s * out_pitch + b * (size_t)inp;
    float2 *tl = tails + batch * (size_t)ovl;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int n1 = r0 + 16 * i, n = 256 * n1 + 16 * cb + c;
        const float2 v = buf[c * F64_P + n1];
        const float2 w = make_float2(v.x * inv_n, v.y * inv_n);
        if (n < inp) o[n] = w; else tl[n - inp] = w;
    }
}

// out[s][b*inp + i] += (b ? tails[s][b-1][i] : carry_in[s][i]);  carry_out[s][i] = tails[s][n_blocks-1][i]     (i < ovl)
__global__ __launch_bounds__(256) void k_f64_tail_add(float2 *__restrict__ out, size_t out_pitch, const float2 *__restrict__ tails, const float2 *__restrict__ carry_in,
                                                      float2 *__restrict__ carry_out, int inp, int ovl, int n_blocks)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ovl) return;
    const size_t s = blockIdx.z; const int b = blockIdx.y;
    const float2 prev = b ? tails[(s * n_blocks + (b - 1)) * (size_t)ovl + i] : carry_in[s * ovl + i];
    float2 *o = out + s * out_pitch + (size_t)b * inp + i;
    const float2 cur = *o;
    *o = make_float2(prev.x + cur.x, prev.y + cur.y);                    // the reference starts from the overlap and adds the new block (two terms: order free)
    if (b == n_blocks - 1) carry_out[s * ovl + i] = tails[(s * n_blocks + b) * (size_t)ovl + i];
}

// K3: grid (16 column blocks, batch)
__global__ __launch_bounds__(256) void k_f64_cols_inv(const float2 *__restrict__ U, float2 *__restrict__ y, const float2 *__restrict__ g_tw)
{
    F64_LDS;
    load_tables(tw256, twlo, g_tw);
    const int t = threadIdx.x, c = t & 15, r0 = t >> 4, cb = blockIdx.x;
    const float2 *src = U + (size_t)blockIdx.y * F64_N;
#pragma unroll
    for (int i = 0; i < 16; i++) { const int k1 = r0 + 16 * i; buf[c * F64_P + k1] = src[256 * k1 + 16 * cb + c]; }
    __syncthreads();
    fft256<true>(buf, t >> 4, t & 15, tw256);
    float2 *dst = y + (size_t)blockIdx.y * F64_N;
#pragma unroll
    for (int i = 0; i < 16; i++) { const int n1 = r0 + 16 * i; dst[256 * n1 + 16 * cb + c] = buf[c * F64_P + n1]; }
}

// Ht[k1][k2] = H[k1 + 256 k2]
__global__ __launch_bounds__(256) void k_f64_transpose_taps(const float2 *__restrict__ H, float2 *__restrict__ Ht)
{
    const int k1 = blockIdx.x, k2 = threadIdx.x;
    Ht[256 * k1 + k2] = H[k1 + 256 * k2];
}

} // namespace

namespace csdr_amd {

// tables: [0,256) exp(-2 pi i m / 256), [256,512) exp(-2 pi i m / 65536)
int fft64k_upload_tables(float2 *d_tw)
{
    std::vector<float2> h(512);
    for (int m = 0; m < 256; m++) {
        const double a = -2.0 * M_PI * m / 256.0, b = -2.0 * M_PI * m / 65536.0;
        h[m] = make_float2((float)cos(a), (float)sin(a)); h[256 + m] = make_float2((float)cos(b), (float)sin(b));
    }
    CSDR_HIP(hipMemcpy(d_tw, h.data(), sizeof(float2) * 512, hipMemcpyHostToDevice));
    return 0;
}

int fft64k_transpose_taps(hipStream_t st, const cf32 *d_taps_fft, cf32 *d_taps_fft_t)
{
    hipLaunchKernelGGL(k_f64_transpose_taps, dim3(256), dim3(256), 0, st, reinterpret_cast<const float2 *>(d_taps_fft), reinterpret_cast<float2 *>(d_taps_fft_t));
    CSDR_LAUNCH_CHECK();
    return 0;
}

// forward transform of n_streams x n_blocks zero-padded input blocks, bin product with the taps' spectrum, inverse transform (unnormalised) into d_td
int fft64k_filter(hipStream_t st, const cf32 *in, size_t in_pitch, int inp, int n_blocks, int n_streams, cf32 *d_work, const cf32 *d_taps_fft_t,
                  const float2 *d_tw, cf32 *d_td)
{
    const int batch = n_blocks * n_streams;
    if (lds_attr_once((const void *)k_f64_cols_fwd, F64_LDS_BYTES) || lds_attr_once((const void *)k_f64_rows, F64_LDS_BYTES) || lds_attr_once((const void *)k_f64_cols_inv, F64_LDS_BYTES)) return -1;
    hipLaunchKernelGGL(k_f64_cols_fwd, dim3(16, n_blocks, n_streams), dim3(256), F64_LDS_BYTES, st, in, in_pitch, inp, n_blocks, reinterpret_cast<float2 *>(d_work), d_tw);
    CSDR_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_f64_rows, dim3(16, batch), dim3(256), F64_LDS_BYTES, st, reinterpret_cast<float2 *>(d_work), reinterpret_cast<const float2 *>(d_taps_fft_t), d_tw);
    CSDR_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_f64_cols_inv, dim3(16, batch), dim3(256), F64_LDS_BYTES, st, reinterpret_cast<const float2 *>(d_work), reinterpret_cast<float2 *>(d_td), d_tw);
    CSDR_LAUNCH_CHECK();
    return 0;
}

// the same with the overlap-add fused into the last pass (needs overlap <= input_size); d_tails: [n_streams * n_blocks][ovl]
int fft64k_filter_oa(hipStream_t st, const cf32 *in, size_t in_pitch, int inp, int ovl, int n_blocks, int n_streams, cf32 *d_work, const cf32 *d_taps_fft_t,
                     const float2 *d_tw, cf32 *d_tails, const cf32 *d_carry_in, cf32 *d_carry_out, cf32 *out, size_t out_pitch)
{
    const int batch = n_blocks * n_streams;
    if (lds_attr_once((const void *)k_f64_cols_fwd, F64_LDS_BYTES) || lds_attr_once((const void *)k_f64_rows, F64_LDS_BYTES) || lds_attr_once((const void *)k_f64_cols_inv_oa, F64_LDS_BYTES)) return -1;
    hipLaunchKernelGGL(k_f64_cols_fwd, dim3(16, n_blocks, n_streams), dim3(256), F64_LDS_BYTES, st, in, in_pitch, inp, n_blocks, reinterpret_cast<float2 *>(d_work), d_tw);
    CSDR_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_f64_rows, dim3(16, batch), dim3(256), F64_LDS_BYTES, st, reinterpret_cast<float2 *>(d_work), reinterpret_cast<const float2 *>(d_taps_fft_t), d_tw);
    CSDR_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_f64_cols_inv_oa, dim3(16, batch), dim3(256), F64_LDS_BYTES, st, reinterpret_cast<const float2 *>(d_work), reinterpret_cast<float2 *>(out), out_pitch,
                       reinterpret_cast<float2 *>(d_tails), inp, ovl, n_blocks, 1.0f / (float)F64_N, d_tw);
    CSDR_LAUNCH_CHECK();
    if (ovl > 0) {
        hipLaunchKernelGGL(k_f64_tail_add, dim3((ovl + 255) / 256, n_blocks, n_streams), dim3(256), 0, st, reinterpret_cast<float2 *>(out), out_pitch,
                           reinterpret_cast<const float2 *>(d_tails), reinterpret_cast<const float2 *>(d_carry_in), reinterpret_cast<float2 *>(d_carry_out), inp, ovl, n_blocks);
        CSDR_LAUNCH_CHECK();
    }
    return 0;
}

} // namespace csdr_amd

// Test hook (tests/test_abi_cpu.py): the register-level 16-point butterfly on the CPU; in/out: 16 interleaved complex floats
extern "C" void csdr_amd_debug_dft16(const float *in32, float *out32, int inverse)
{
    float2 v[16];
    for (int k = 0; k < 16; k++) v[k] = make_float2(in32[2 * k], in32[2 * k + 1]);
    if (inverse) dft16<true>(v); else dft16<false>(v);
    for (int k = 0; k < 16; k++) { out32[2 * k] = v[k].x; out32[2 * k + 1] = v[k].y; }
}
