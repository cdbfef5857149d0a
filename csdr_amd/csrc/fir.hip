// fir.hip -- time-domain FIR kernels.
//   fir_decimate_cc   libcsdr.c:528-549  (real taps on interleaved complexf, decimation D)
//   deemphasis_nfm_ff libcsdr.c:1101-1128 (fixed real FIR on floats, no decimation)
//
// Layout / tiling: one workgroup = one (stream, tile of TO outputs).  The input window of
// D*TO + taps - 1 samples is staged once through LDS with coalesced 16-byte global loads (interleaved
// complexf kept as is: a float2 per sample), so every input sample is read from HBM once (+ halo).
// Each lane then produces outputs from LDS; taps are wave-uniform and come through the scalar path (s_load -> SGPR operand).
// Algorithmic traffic: 8 B in + 8/D B out per input sample => HBM bound on paper (3.6 flop/B).
// Measured (profiles/r1_ops.jsonl): 2.9 TB/s algorithmic at D=10/79 taps on long streams (37 % of peak), 1.4 TB/s at D=50/801 taps.
// A register-blocked variant (5 consecutive outputs per lane, zero-padded taps) was tried in round 1 and was SLOWER (2.1 TB/s):
// the kernel is not LDS-read bound; it is the next kernel to restructure (round 2).
#include "common.hpp"
using namespace csdr_amd;

namespace {

// Generic kernel.  blockDim = 256; each thread handles outputs o = tid, tid+256, ... < tile_outputs.
// LDS holds (tile_outputs-1)*D + taps samples.
template <bool COMPLEX>
__global__ __launch_bounds__(256) void k_fir_generic(const float *__restrict__ in, float *__restrict__ out, int n_out, int tile_outputs,
                                                     size_t in_pitch, size_t out_pitch, int D, const float *__restrict__ taps, int ntaps)
{
    extern __shared__ float4 lds_raw[];
    constexpr int W = COMPLEX ? 2 : 1;                       // floats per sample
    float *win = reinterpret_cast<float *>(lds_raw);
    const int o0 = blockIdx.x * tile_outputs;
    const int outs = min(tile_outputs, n_out - o0);
    if (outs <= 0) return;
    const size_t s = blockIdx.y;
    const float *src = in + (s * in_pitch + (size_t)o0 * D) * W;
    const int win_samples = (outs - 1) * D + ntaps;
    const int win_floats = win_samples * W;
    // coalesced staging; 16-byte path when the window start is 16-byte aligned
    if ((((uintptr_t)src) & 15) == 0) {
        const int nv = win_floats / 4;
        for (int v = threadIdx.x; v < nv; v += blockDim.x) reinterpret_cast<float4 *>(win)[v] = reinterpret_cast<const float4 *>(src)[v];
        for (int k = nv * 4 + threadIdx.x; k < win_floats; k += blockDim.x) win[k] = src[k];
    } else {
        for (int k = threadIdx.x; k < win_floats; k += blockDim.x) win[k] = src[k];
    }
    __syncthreads();
    float *dst = out + (s * out_pitch + (size_t)o0) * W;
    for (int o = threadIdx.x; o < outs; o += blockDim.x) {
        const float *x = win + (size_t)o * D * W;
        if (COMPLEX) {
            float ai = 0.f, aq = 0.f;
            for (int t = 0; t < ntaps; t++) {
                const float h = taps[t];                      // uniform address -> scalar load
                const float2 v = reinterpret_cast<const float2 *>(x)[t];
                ai = fmaf(v.x, h, ai); aq = fmaf(v.y, h, aq);
            }
            reinterpret_cast<float2 *>(dst)[o] = make_float2(ai, aq);
        } else {
            float a = 0.f;
            for (int t = 0; t < ntaps; t++) a = fmaf(x[t], taps[t], a);
            dst[o] = a;
        }
    }
}

} // namespace

static int pick_tile(int D, int ntaps, int floats_per_sample, int n_out)
{
    // largest tile (<= 1024 outputs) whose window fits in 64 KiB of LDS, at least 1
    int to = 1024;
    while (to > 1 && ((size_t)(to - 1) * D + ntaps) * floats_per_sample * 4 > 64 * 1024) to /= 2;
    if (to > n_out) to = n_out;
    return to < 1 ? 1 : to;
}

extern "C" {

int csdr_amd_fir_decimate_cc(csdr_amd_ctx *c, const csdr_complexf *in, csdr_complexf *out, int n_streams, int input_size,
                             size_t in_pitch, size_t out_pitch, int decimation, const float *taps, int taps_length)
{
    if (decimation <= 0 || taps_length <= 0) return fail_msg(-3, "fir_decimate_cc: bad decimation/taps_length");
    if (input_size < taps_length || n_streams <= 0) return 0;
    const int n_out = (input_size - taps_length) / decimation + 1;     // libcsdr.c:536-538 loop bound
    const int to = pick_tile(decimation, taps_length, 2, n_out);
    const size_t win_bytes = ((size_t)(to - 1) * decimation + taps_length) * 8;
    if (win_bytes > 160 * 1024 - 256) return fail_msg(-3, "fir_decimate_cc: %d taps exceed the LDS window (use the FFT path)", taps_length);
    if (win_bytes > 64 * 1024) CSDR_HIP(hipFuncSetAttribute((const void *)k_fir_generic<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)win_bytes));
    dim3 grid(cdiv(n_out, to), (unsigned)n_streams);
    hipLaunchKernelGGL((k_fir_generic<true>), grid, dim3(256), win_bytes + 16, c->stream, (const float *)in, (float *)out, n_out, to,
                       in_pitch, out_pitch, decimation, taps, taps_length);
    CSDR_LAUNCH_CHECK();
    return n_out;
}

int csdr_amd_fir_ff(csdr_amd_ctx *c, const float *in, float *out, int n_streams, int input_size, size_t in_pitch, size_t out_pitch,
                    const float *taps, int taps_length)
{
    if (taps_length <= 0) return 0;
    const int n_out = input_size - taps_length;                         // libcsdr.c:1121: i < input_size - taps_length
    if (n_out <= 0 || n_streams <= 0) return 0;
    // window for n outputs of the generic kernel is (outs-1)*1 + ntaps, exactly what those outputs read
    const int to = pick_tile(1, taps_length, 1, n_out);
    const size_t win_bytes = ((size_t)(to - 1) + taps_length) * 4;
    dim3 grid(cdiv(n_out, to), (unsigned)n_streams);
    hipLaunchKernelGGL((k_fir_generic<false>), grid, dim3(256), win_bytes + 16, c->stream, in, out, n_out, to, in_pitch, out_pitch, 1, taps, taps_length);
    CSDR_LAUNCH_CHECK();
    return n_out;
}

} // extern "C"
