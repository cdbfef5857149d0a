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
#include <stdlib.h>
#include <vector>
using namespace csdr_amd;

namespace {

constexpr int F64_N = 65536;
constexpr int F64_P = 273;                 // LDS pitch of one 256-point transform (>= 17 * 16, odd: column-major fills are conflict free)

// 256-point transform of transform f (of 16 in the workgroup) by its 16 threads j = 0..15: n = 16 n1 + n2, k = k1 + 16 k2.
// The caller hands thread j the 16 values x[16 n1 + j] (n1 = 0..15) IN REGISTERS -- every kernel below loads exactly that set from global memory, 16 lanes
// (j) of a load instruction reading one 128-byte run -- and gets back X[j + 16 k2] (k2 = 0..15), again the set its stores want: the only trip through LDS
// is the exchange between the two radix-16 stages ([k1][n2] layout, pitch 17; one write and one read per element instead of three each when the tile
// was first parked in LDS, transformed in place and read back).  tw256: exp(-2 pi i m / 256) in LDS.  All 256 threads of the workgroup must call it.
template <bool INV>
__device__ __forceinline__ void fft256_regs(float2 (&v)[16], float2 *buf, int f, int j, const float2 *tw256)
{
    dft16<INV>(v);                                                                             // over n1: v[k1], this thread's n2 = j
#pragma unroll
    for (int k1 = 0; k1 < 16; k1++) {
        float2 w = tw256[(j * k1) & 255]; if (INV) w.y = -w.y;
        buf[f * F64_P + 17 * k1 + j] = cmul(v[k1], w);
    }
    __syncthreads();
#pragma unroll
    for (int n2 = 0; n2 < 16; n2++) v[n2] = buf[f * F64_P + 17 * j + n2];                      // thread j = k1
    dft16<INV>(v);                                                                             // v[k2] = X[j + 16 k2]
}

// W_65536^m = tw256[m >> 8] * twlo[m & 255]
__device__ __forceinline__ float2 twiddle_n(const float2 *tw256, const float2 *twlo, int m, bool inv)
{
    float2 w = cmul(tw256[(m >> 8) & 255], twlo[m & 255]);
    if (inv) w.y = -w.y;
    return w;
}

// dynamic LDS: the exchange buffer, the two twiddle tables
#define F64_LDS extern __shared__ float4 f64_lds_raw[]; float2 *buf = reinterpret_cast<float2 *>(f64_lds_raw), *tw256 = buf + 16 * F64_P, *twlo = tw256 + 256
constexpr size_t F64_LDS_BYTES = (size_t)(16 * F64_P + 512) * sizeof(float2);

__device__ __forceinline__ void load_tables(float2 *tw256, float2 *twlo, const float2 *g_tw)
{
    tw256[threadIdx.x] = g_tw[threadIdx.x]; twlo[threadIdx.x] = g_tw[256 + threadIdx.x];
}

// K1: grid (16 column blocks, n_blocks, n_streams); block 256: thread (c = column of the block, j): rows 16 n1 + j of column 16 cb + c
__global__ __launch_bounds__(256) void k_f64_cols_fwd(const cf32 *__restrict__ in, size_t in_pitch, int inp, int n_blocks, float2 *__restrict__ T, const float2 *__restrict__ g_tw)
{
    F64_LDS;
    load_tables(tw256, twlo, g_tw);
    const int t = threadIdx.x, c = t & 15, j = t >> 4;
    const int cb = blockIdx.x; const size_t s = blockIdx.z, b = blockIdx.y;
    const float2 *x = reinterpret_cast<const float2 *>(in) + s * in_pitch + b * (size_t)inp;
    const int n2 = 16 * cb + c;
    float2 v[16];
#pragma unroll
    for (int n1 = 0; n1 < 16; n1++) {
        const int n = 256 * (16 * n1 + j) + n2;
        v[n1] = n < inp ? x[n] : make_float2(0.f, 0.f);                                        // csdr.c:1864: the block is zero padded to fft_size
    }
    __syncthreads();                                                                           // twiddle tables
    fft256_regs<false>(v, buf, c, j, tw256);
    float2 *dst = T + (s * n_blocks + b) * (size_t)F64_N + n2;
#pragma unroll
    for (int k2 = 0; k2 < 16; k2++) {
        const int k1 = j + 16 * k2;
        dst[256 * k1] = cmul(v[k2], twiddle_n(tw256, twlo, n2 * k1, false));
    }
}

// K2: grid (16 row blocks, batch); rows k1 = 16 rb .. +15: thread (j, f = row of the block): elements 16 n1 + j of row f (128-byte runs per 16 lanes)
__global__ __launch_bounds__(256) void k_f64_rows(float2 *__restrict__ T, const float2 *__restrict__ Ht, const float2 *__restrict__ g_tw)
{
    F64_LDS;
    load_tables(tw256, twlo, g_tw);
    const int t = threadIdx.x, j = t & 15, f = t >> 4, rb = blockIdx.x;
    float2 *row = T + (size_t)blockIdx.y * F64_N + (size_t)(16 * rb + f) * 256 + j;
    const float2 *h = Ht + (size_t)(16 * rb + f) * 256 + j;
    float2 v[16], hv[16];
#pragma unroll
    for (int n1 = 0; n1 < 16; n1++) v[n1] = row[16 * n1];
#pragma unroll
    for (int k2 = 0; k2 < 16; k2++) hv[k2] = h[16 * k2];                                       // taps_fft[k1 + 256 (j + 16 k2)], needed after the forward transform
    __syncthreads();
    fft256_regs<false>(v, buf, f, j, tw256);                                                   // v[k2] = X[k1 + 256 (j + 16 k2)]
#pragma unroll
    for (int k2 = 0; k2 < 16; k2++) v[k2] = cmul(v[k2], hv[k2]);                               // libcsdr.c:826-830
    __syncthreads();                                                                           // the exchange buffer is reused
    // the inverse transform's input index 16 n1 + n2 with n2 = j, n1 = k2: exactly what this thread holds
    fft256_regs<true>(v, buf, f, j, tw256);
    const int k1 = 16 * rb + f;
#pragma unroll
    for (int k2 = 0; k2 < 16; k2++) { const int n2 = j + 16 * k2; row[16 * k2] = cmul(v[k2], twiddle_n(tw256, twlo, n2 * k1, true)); }
}

// K3 with the overlap-add fused (overlap <= input_size: every output position has at most two contributions): the block's own samples go
// straight to `out` (scaled by 1/N, libcsdr.c:836-839), its tail (the last taps-1 samples) to tails[batch][ovl]; k_f64_tail_add then adds
// the previous block's tail (or the carry from the previous call) to the first `ovl` outputs of every block (libcsdr.c:843-847).
__global__ __launch_bounds__(256) void k_f64_cols_inv_oa(const float2 *__restrict__ U, float2 *__restrict__ out, size_t out_pitch, float2 *__restrict__ tails,
                                                         int inp, int ovl, int n_blocks, float inv_n, const float2 *__restrict__ g_tw)
{
    F64_LDS;
    load_tables(tw256, twlo, g_tw);
    const int t = threadIdx.x, c = t & 15, j = t >> 4, cb = blockIdx.x;
    const size_t batch = blockIdx.y, s = batch / n_blocks, b = batch % n_blocks;
    const float2 *src = U + batch * F64_N + 16 * cb + c;
    float2 v[16];
#pragma unroll
    for (int n1 = 0; n1 < 16; n1++) v[n1] = src[256 * (16 * n1 + j)];
    __syncthreads();
    fft256_regs<true>(v, buf, c, j, tw256);
    float2 *o = out + s * out_pitch + b * (size_t)inp;
    float2 *tl = tails + batch * (size_t)ovl;
#pragma unroll
    for (int k2 = 0; k2 < 16; k2++) {
        const int n = 256 * (j + 16 * k2) + 16 * cb + c;
        const float2 w = make_float2(v[k2].x * inv_n, v[k2].y * inv_n);
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
    const int t = threadIdx.x, c = t & 15, j = t >> 4, cb = blockIdx.x;
    const float2 *src = U + (size_t)blockIdx.y * F64_N + 16 * cb + c;
    float2 v[16];
#pragma unroll
    for (int n1 = 0; n1 < 16; n1++) v[n1] = src[256 * (16 * n1 + j)];
    __syncthreads();
    fft256_regs<true>(v, buf, c, j, tw256);
    float2 *dst = y + (size_t)blockIdx.y * F64_N + 16 * cb + c;
#pragma unroll
    for (int k2 = 0; k2 < 16; k2++) dst[256 * (j + 16 * k2)] = v[k2];
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
    if (lds_attr_once((const void *)k_f64_cols_fwd, F64_LDS_BYTES) || lds_attr_once((const void *)k_f64_rows, F64_LDS_BYTES) || lds_attr_once((const void *)k_f64_cols_inv_oa, F64_LDS_BYTES)) return -1;
    // The three passes of a group of streams run back to back, one group after the other: the [256][256] intermediates of a group (0.5 MiB per
    // transform, written by one pass and read by the next) then stay inside the 256 MiB Infinity Cache instead of making two round trips to HBM.
    // CSDR_AMD_FFT64K_GROUP = transforms per group (default 256 = 128 MiB of intermediates: measured best, profiles/r2h; 0 = the whole call at once).
    static const long group_env = getenv("CSDR_AMD_FFT64K_GROUP") ? atol(getenv("CSDR_AMD_FFT64K_GROUP")) : 256;
    int sg = n_streams;
    if (group_env > 0) { sg = (int)(group_env / n_blocks); if (sg < 1) sg = 1; if (sg > n_streams) sg = n_streams; }
    for (int s0 = 0; s0 < n_streams; s0 += sg) {
        const int ns = (n_streams - s0 < sg) ? n_streams - s0 : sg, batch = n_blocks * ns;
        float2 *work = reinterpret_cast<float2 *>(d_work) + (size_t)s0 * n_blocks * F64_N;
        hipLaunchKernelGGL(k_f64_cols_fwd, dim3(16, n_blocks, ns), dim3(256), F64_LDS_BYTES, st, in + (size_t)s0 * in_pitch, in_pitch, inp, n_blocks, work, d_tw);
        CSDR_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_f64_rows, dim3(16, batch), dim3(256), F64_LDS_BYTES, st, work, reinterpret_cast<const float2 *>(d_taps_fft_t), d_tw);
        CSDR_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_f64_cols_inv_oa, dim3(16, batch), dim3(256), F64_LDS_BYTES, st, work, reinterpret_cast<float2 *>(out) + (size_t)s0 * out_pitch, out_pitch,
                           reinterpret_cast<float2 *>(d_tails) + (size_t)s0 * n_blocks * ovl, inp, ovl, n_blocks, 1.0f / (float)F64_N, d_tw);
        CSDR_LAUNCH_CHECK();
    }
    if (ovl > 0) {
        hipLaunchKernelGGL(k_f64_tail_add, dim3((ovl + 255) / 256, n_blocks, n_streams), dim3(256), 0, st, reinterpret_cast<float2 *>(out), out_pitch,
                           reinterpret_cast<const float2 *>(d_tails), reinterpret_cast<const float2 *>(d_carry_in), reinterpret_cast<float2 *>(d_carry_out), inp, ovl, n_blocks);
        CSDR_LAUNCH_CHECK();
    }
    return 0;
}

// the overlap-add's last step on its own (the two-pass block filter of fftfilt_lds.hip leaves out + tails like k_f64_cols_inv_oa does)
int fft64k_tail_add(hipStream_t st, cf32 *out, size_t out_pitch, const cf32 *d_tails, const cf32 *d_carry_in, cf32 *d_carry_out, int inp, int ovl, int n_blocks, int n_streams)
{
    if (ovl <= 0) return 0;
    hipLaunchKernelGGL(k_f64_tail_add, dim3((ovl + 255) / 256, n_blocks, n_streams), dim3(256), 0, st, reinterpret_cast<float2 *>(out), out_pitch,
                       reinterpret_cast<const float2 *>(d_tails), reinterpret_cast<const float2 *>(d_carry_in), reinterpret_cast<float2 *>(d_carry_out), inp, ovl, n_blocks);
    CSDR_LAUNCH_CHECK();
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
