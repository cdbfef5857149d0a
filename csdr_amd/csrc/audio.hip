// audio.hip -- demodulator and audio-rate blocks of the receive chains.
//   fmdemod_quadri_cf       libcsdr.c:1021,1040-1071
//   limit_ff / gain_ff      libcsdr.c:1130-1142
//   deemphasis_wfm_ff       libcsdr.c:1081-1097
//   fastagc_ff              libcsdr.c:946-991
//   fractional_decimator_ff libcsdr.c:715-793
// File is built with -ffp-contract=off: products and sums round separately like the reference's SSE code.
#include "common.hpp"
#include <math.h>
#include <vector>
#include <string.h>
using namespace csdr_amd;

namespace {

// ------------------------------------------------------------------ fmdemod_quadri_cf
__global__ __launch_bounds__(256) void k_fmdemod(const cf32 *__restrict__ in, float *__restrict__ out, size_t n,
                                                 size_t in_pitch, size_t out_pitch, const cf32 *__restrict__ last)
{
    const float Kf = 0.340447550238101026565118445432744920253753662109375f;   // libcsdr.c:1021
    const cf32 *src = in + (size_t)blockIdx.y * in_pitch;
    float *dst = out + (size_t)blockIdx.y * out_pitch;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) {
        const cf32 x = src[k];
        const cf32 p = k ? src[k - 1] : last[blockIdx.y];
        const float dq = x.q - p.q, di = x.i - p.i;
        const float num = x.i * dq - x.q * di;
        const float den = x.i * x.i + x.q * x.q;
        // the reference scales and divides in double (:1067) and rounds once; a float product with a Newton-refined reciprocal is within 2 ulp of
        // that (gate: 1e-5 relative RMS) and avoids the ~25-instruction fp64 division sequence that made this kernel arithmetic bound
        float rd = __builtin_amdgcn_rcpf(den);
        rd = fmaf(fmaf(-den, rd, 1.0f), rd, rd);
        dst[k] = (den != 0.f) ? (Kf * num) * rd : 0.f;
    }
}
__global__ void k_store_last(const cf32 *__restrict__ in, size_t n, size_t in_pitch, cf32 *__restrict__ last, int n_streams)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n_streams) last[s] = in[(size_t)s * in_pitch + n - 1];
}

// ------------------------------------------------------------------ limit / gain
__global__ __launch_bounds__(256) void k_limit(const float *__restrict__ in, float *__restrict__ out, size_t n, float m)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t nv = n / 4;
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += stride) {
        float4 x = reinterpret_cast<const float4 *>(in)[v];
        x.x = (m < x.x) ? m : x.x; x.x = (-m > x.x) ? -m : x.x;
        x.y = (m < x.y) ? m : x.y; x.y = (-m > x.y) ? -m : x.y;
        x.z = (m < x.z) ? m : x.z; x.z = (-m > x.z) ? -m : x.z;
        x.w = (m < x.w) ? m : x.w; x.w = (-m > x.w) ? -m : x.w;
        reinterpret_cast<float4 *>(out)[v] = x;
    }
    const size_t t = nv * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) { float x = in[t]; x = (m < x) ? m : x; out[t] = (-m > x) ? -m : x; }
}
__global__ __launch_bounds__(256) void k_gain(const float *__restrict__ in, float *__restrict__ out, size_t n, float g)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t nv = n / 4;
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += stride) {
        float4 x = reinterpret_cast<const float4 *>(in)[v];
        x.x *= g; x.y *= g; x.z *= g; x.w *= g;
        reinterpret_cast<float4 *>(out)[v] = x;
    }
    const size_t t = nv * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = g * in[t];
}

// ------------------------------------------------------------------ deemphasis_wfm_ff
// One wave owns 64 streams.  Time is walked in tiles of 64 samples: the 64x64 tile is loaded with
// coalesced row reads (256 B per stream row) into LDS, each lane then runs ITS stream's recurrence
// over the tile from LDS (row stride 65 floats: conflict free), and the tile is stored back coalesced.
// The recurrence itself is the reference's, operation for operation (bit exact for finite input).
__global__ __launch_bounds__(64) void k_deemph_wfm(const float *__restrict__ in, float *__restrict__ out, int n_streams, size_t n,
                                                   size_t in_pitch, size_t out_pitch, float alpha, float *__restrict__ last_io)
{
    __shared__ float tile[64 * 65];
    const int lane = threadIdx.x;
    const int s0 = blockIdx.x * 64;
    const int rows = min(64, n_streams - s0);
    float y = 0.f;
    if (lane < rows) { y = last_io[s0 + lane]; if (y != y) y = 0.f; }      // NaN state reset, libcsdr.c:1092
    const float one_minus = 1 - alpha;
    for (size_t t0 = 0; t0 < n; t0 += 64) {
        const int cols = (int)((n - t0 < 64) ? (n - t0) : 64);
        for (int r = 0; r < rows; r++) if (lane < cols) tile[r * 65 + lane] = in[(size_t)(s0 + r) * in_pitch + t0 + lane];
        __syncthreads();
        if (lane < rows) for (int k = 0; k < cols; k++) { y = alpha * tile[lane * 65 + k] + one_minus * y; tile[lane * 65 + k] = y; }
        __syncthreads();
        for (int r = 0; r < rows; r++) if (lane < cols) out[(size_t)(s0 + r) * out_pitch + t0 + lane] = tile[r * 65 + lane];
        __syncthreads();
    }
    if (lane < rows) last_io[s0 + lane] = y;
}

// ---- the same recurrence for FEW, LONG streams (a CLI process has one: `csdr deemphasis_wfm_ff` walked its 80 000 samples per block on ONE lane -- 34 M samples/s, the
// slowest stage of the literal README.md:66 pipeline by far).  y[k] = fl(fl(alpha x[k]) + fl(b y[k-1])) is a contraction (b = 1 - alpha < 1): a trajectory started
// from the wrong state forgets it at the rate b^k and, once it agrees with the true one to the last bit, stays identical.  So the stream is cut into chunks of 256
// samples, one lane each; a lane first runs over the DW_M chunks in front of its own from state zero (b^(256 DW_M) < 2^-40: the host picks DW_M, or the serial kernel
// when none fits), notes the state it arrives with, then computes its chunk.  k_deemph_wfm_check compares every lane's arrival state with the END state its
// predecessor computed, bit for bit: if all agree the whole stream is the sequential recurrence's, by induction from chunk 0 (which starts from the carried state).
// Where one does not (in practice never; kept so that the result is the reference's bits, not merely close), k_deemph_wfm_fix recomputes from there on,
// sequentially, until its end state meets the stored one again.  Bit exact like k_deemph_wfm, 2 x the arithmetic, all lanes busy.
constexpr int DW_L = 256, DW_P = DW_L + 1;              // chunk length; LDS row pitch (conflict free for lanes walking their own rows)
template <int M>
__global__ __launch_bounds__(64) void k_deemph_wfm_spec(const float *__restrict__ in, float *__restrict__ out, size_t n, size_t in_pitch, size_t out_pitch, float alpha,
                                                        const float *__restrict__ last_io, float *__restrict__ st_start, float *__restrict__ st_end, size_t n_chunks)
{
    extern __shared__ float dw_tile[];                               // [(M + 64) rows][DW_P]: rows 0 .. M-1 = the chunks in front of this wave's first one
    const int lane = threadIdx.x;
    const size_t s = blockIdx.y, c0 = (size_t)blockIdx.x * 64;
    const float *x = in + s * in_pitch;
    float *y_row = out + s * out_pitch;
    const float one_minus = 1 - alpha;
    for (int r = 0; r < M + 64; r++) {                               // coalesced: one row = 1 KiB
        const long long c = (long long)c0 + r - M;
        if (c < 0 || (size_t)c >= n_chunks) continue;
#pragma unroll
        for (int q = 0; q < DW_L / 64; q++) { const size_t k = (size_t)c * DW_L + lane + 64 * q; dw_tile[r * DW_P + lane + 64 * q] = k < n ? x[k] : 0.f; }
    }
    __syncthreads();
    const size_t c = c0 + lane;
    if (c < n_chunks) {
        // run-in over chunks c - M .. c - 1 from state zero -- or, where the stream's first chunk is among them (c <= M), over chunks 0 .. c - 1 from the carried
        // state (NaN reset, libcsdr.c:1092): those lanes are exact outright
        float y = 0.f;
        if (c <= (size_t)M) { y = last_io[s]; if (y != y) y = 0.f; }
        for (int m = (c < (size_t)M ? (int)(M - c) : 0); m < M; m++) {
            const float *row = dw_tile + (lane + m) * DW_P;
            for (int k = 0; k < DW_L; k++) y = alpha * row[k] + one_minus * y;
        }
        st_start[s * n_chunks + c] = y;
        float *row = dw_tile + (lane + M) * DW_P;
        const int cols = (int)((n - c * DW_L < (size_t)DW_L) ? (n - c * DW_L) : DW_L);
        for (int k = 0; k < cols; k++) { y = alpha * row[k] + one_minus * y; row[k] = y; }
        st_end[s * n_chunks + c] = y;
    }
    __syncthreads();
    for (int r = M; r < M + 64; r++) {
        const size_t cc = c0 + r - M;
        if (cc >= n_chunks) break;
#pragma unroll
        for (int q = 0; q < DW_L / 64; q++) { const size_t k = cc * DW_L + lane + 64 * q; if (k < n) y_row[k] = dw_tile[r * DW_P + lane + 64 * q]; }
    }
}
// flags[s] = number of chunks whose arrival state is not, bit for bit, the end state of the chunk in front
__global__ __launch_bounds__(256) void k_deemph_wfm_check(const float *__restrict__ st_start, const float *__restrict__ st_end, size_t n_chunks, unsigned *__restrict__ flags)
{
    const size_t s = blockIdx.y, c = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (c == 0 || c >= n_chunks) return;
    if (__float_as_uint(st_start[s * n_chunks + c]) != __float_as_uint(st_end[s * n_chunks + c - 1])) atomicAdd(&flags[s], 1u);
}
// one thread per stream: the carried state out; and, where the check found a disagreement (in practice never), the sequential repair: chunk by chunk in order, every
// chunk whose arrival state is not its predecessor's FINAL end state is recomputed from that state
__global__ __launch_bounds__(64) void k_deemph_wfm_fix(const float *__restrict__ in, float *__restrict__ out, size_t n, size_t in_pitch, size_t out_pitch, float alpha,
                                                       float *__restrict__ last_io, const float *__restrict__ st_start, float *__restrict__ st_end, size_t n_chunks,
                                                       unsigned *__restrict__ flags, int n_streams)
{
    const size_t s = threadIdx.x;                                    // (fewer than 32 streams take this path)
    if (s >= (size_t)n_streams) return;
    const float one_minus = 1 - alpha;
    if (flags[s]) {
        const float *x = in + s * in_pitch; float *y_row = out + s * out_pitch;
        const float *ss = st_start + s * n_chunks; float *se = st_end + s * n_chunks;
        for (size_t c = 1; c < n_chunks; c++) {
            if (__float_as_uint(ss[c]) == __float_as_uint(se[c - 1])) continue;      // computed from the right state: exact as it stands
            float y = se[c - 1];
            const int cols = (int)((n - c * DW_L < (size_t)DW_L) ? (n - c * DW_L) : DW_L);
            for (int k = 0; k < cols; k++) { y = alpha * x[c * DW_L + k] + one_minus * y; y_row[c * DW_L + k] = y; }
            se[c] = y;
        }
    }
    last_io[s] = st_end[s * n_chunks + n_chunks - 1];
}

// ------------------------------------------------------------------ fastagc_ff
// State layout per stream: [buffer_1 (block) | buffer_2 (block) | peak_1 peak_2 last_gain pad]
__device__ __forceinline__ const float *agc_seq_block(const float *state, const float *in_row, int block, int j)
{   // the conceptual block sequence: buffer_1, buffer_2, in_0, in_1, ...
    return (j == 0) ? state : (j == 1) ? state + block : in_row + (size_t)(j - 2) * block;
}
__global__ __launch_bounds__(256) void k_agc_peaks(const float *__restrict__ in, size_t in_pitch, int block, int n_blocks, float *__restrict__ peaks)
{   // grid (n_blocks, n_streams): peak |x| of one new block (libcsdr.c:957-962)
    const float *x = in + (size_t)blockIdx.y * in_pitch + (size_t)blockIdx.x * block;
    float m = 0.f;
    for (int k = threadIdx.x; k < block; k += 256) m = fmaxf(m, fabsf(x[k]));
    for (int off = 32; off; off >>= 1) m = fmaxf(m, __shfl_down(m, off, 64));
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) peaks[(size_t)blockIdx.y * (n_blocks + 2) + 2 + blockIdx.x] = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
}
// target gain of output block j (libcsdr.c:964-971): the peak of sequence blocks j, j+1, j+2 (the first two peaks of a call are the state's), capped at 50.
// peaks: [n_streams][n_blocks + 2], entries 2.. = the new blocks' peaks (entries 0, 1 are not used: the state is read directly)
__device__ __forceinline__ float agc_peak_at(const float *st_tail, const float *pk, int i) { return i < 2 ? st_tail[i] : pk[i]; }
__device__ __forceinline__ float agc_gain_after(const float *st_tail, const float *pk, int j, float reference)
{   // g[j + 1] of the reference's chain; g[0] = st_tail[2] (last_gain)
    float peak = agc_peak_at(st_tail, pk, j + 2);
    const float p1 = agc_peak_at(st_tail, pk, j + 1), p0 = agc_peak_at(st_tail, pk, j);
    if (peak < p1) peak = p1;
    if (peak < p0) peak = p0;
    float target = reference / peak;
    if (target > 50.f) target = 50.f;
    return target;
}
__global__ __launch_bounds__(256) void k_agc_apply(const float *__restrict__ in, float *__restrict__ out, size_t in_pitch, size_t out_pitch,
                                                   int block, int n_blocks, const float *__restrict__ state, const float *__restrict__ peaks, float reference,
                                                   int16_t *__restrict__ out_s16, size_t s16_pitch)
{   // grid (n_blocks, n_streams): output block j = sequence block j with the gain ramped from g[j] to g[j+1] (:973-978);
    // optionally convert_f_s16 (libcsdr.c:2397, x86 truncation semantics) of the result in the same pass (chains that end in `| convert_f_s16`)
    const int j = blockIdx.x; const size_t s = blockIdx.y;
    const float *x = agc_seq_block(state + s * (2 * block + 4), in + s * in_pitch, block, j);
    const float *st_tail = state + s * (2 * block + 4) + 2 * block, *pk = peaks + s * (n_blocks + 2);
    const float g0 = j ? agc_gain_after(st_tail, pk, j - 1, reference) : st_tail[2], g1 = agc_gain_after(st_tail, pk, j, reference);      // the short chain of target gains, evaluated where it is used
    float *y = out ? out + s * out_pitch + (size_t)j * block : nullptr;
    int16_t *z = out_s16 ? out_s16 + s * s16_pitch + (size_t)j * block : nullptr;
    auto gain_at = [&](int k) { const float r = (float)k / (float)block; return (float)((double)g0 * (1.0 - (double)r) + (double)(g1 * r)); };
    auto s16_of = [](float v) { const float scaled = v * 32767.0f; return (scaled >= -2147483648.0f && scaled < 2147483648.0f) ? (int)scaled : (int)0x80000000; };
    // four samples per lane (one 16-byte read, one 8-byte s16 store) when the block and all three rows allow it: the same values, a quarter of the instructions
    if ((block & 3) == 0 && ((((size_t)x) | ((size_t)y)) & 15) == 0 && (((size_t)z) & 7) == 0) {
        for (int k = 4 * threadIdx.x; k < block; k += 1024) {
            const float4 xv = *reinterpret_cast<const float4 *>(x + k);
            const float4 v = make_float4(xv.x * gain_at(k), xv.y * gain_at(k + 1), xv.z * gain_at(k + 2), xv.w * gain_at(k + 3));
            if (y) *reinterpret_cast<float4 *>(y + k) = v;
            if (z) {
                const int a = s16_of(v.x), b = s16_of(v.y), c = s16_of(v.z), d = s16_of(v.w);
                *reinterpret_cast<uint2 *>(z + k) = make_uint2((unsigned)(a & 0xffff) | ((unsigned)b << 16), (unsigned)(c & 0xffff) | ((unsigned)d << 16));
            }
        }
        return;
    }
    for (int k = threadIdx.x; k < block; k += 256) {
        const float v = x[k] * gain_at(k);
        if (y) y[k] = v;
        if (z) z[k] = (int16_t)s16_of(v);
    }
}
__global__ __launch_bounds__(256) void k_agc_update(const float *__restrict__ in, size_t in_pitch, int block, int n_blocks,
                                                    float *__restrict__ state, const float *__restrict__ peaks, float reference)
{   // grid (1, n_streams): rotate the two look-ahead buffers (libcsdr.c:983-989)
    const size_t s = blockIdx.y;
    float *st = state + s * (2 * block + 4);
    const float *row = in + s * in_pitch;
    for (int k = threadIdx.x; k < block; k += 256) {
        const float nb1 = *(agc_seq_block(st, row, block, n_blocks) + k);
        const float nb2 = *(agc_seq_block(st, row, block, n_blocks + 1) + k);
        st[k] = nb1; st[block + k] = nb2;
    }
    __syncthreads();                                                  // every thread has read the old look-ahead buffers
    if (threadIdx.x == 0) {
        const float *pk = peaks + s * (n_blocks + 2);
        const float lg = agc_gain_after(st + 2 * block, pk, n_blocks - 1, reference);
        const float p0 = agc_peak_at(st + 2 * block, pk, n_blocks), p1 = agc_peak_at(st + 2 * block, pk, n_blocks + 1);
        st[2 * block + 0] = p0; st[2 * block + 1] = p1; st[2 * block + 2] = lg;
    }
}

// ------------------------------------------------------------------ fractional_decimator_ff
// The plan holds, per output, only what the reference's sequential position bookkeeping yields: the first input sample of its window and the fractional
// position frac = where - lo.  The Lagrange coefficients (libcsdr.c:775-785: P products of P - 1 factors each and a division, the same float operations in the same
// order) are evaluated HERE -- building them on the host cost 4.7 ms per 400 k-sample call and made `csdr fractional_decimator_ff` the slowest stage of the literal
// README.md:66 pipeline (87 M samples/s).
__global__ __launch_bounds__(256) void k_fracdec(const float *__restrict__ in, float *__restrict__ out, int n_out, size_t in_pitch, size_t out_pitch,
                                                 const int *__restrict__ lo_idx, const float *__restrict__ frac, int P, int xifirst, const float *__restrict__ denom,
                                                 const float *__restrict__ taps, int ntaps)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_out) return;
    const float *x = in + (size_t)blockIdx.y * in_pitch + lo_idx[k];
    const float fx = frac[k];
    float acc = 0.f;
    for (int w = 0; w < P; w++) {
        float y;
        if (ntaps) { y = 0.f; for (int t = 0; t < ntaps; t++) y += taps[t] * x[w + t]; }   // fir_one_pass_ff libcsdr.c:675-680
        else y = x[w];
        const int a = xifirst + w;
        float prod = 1;
        for (int b = xifirst; b < xifirst + P; b++) if (a != b) prod *= (fx - b);
        acc += (prod / denom[w]) * y;
    }
    out[(size_t)blockIdx.y * out_pitch + k] = acc;
}

// The plan ON THE DEVICE (round 6).  The reference's position bookkeeping is the float recurrence where <- where + rate (libcsdr.c:763), restarted per call / per CLI
// window with where <- where - input_processed (:790-791).  When where and rate are both multiples of 2^q and every position stays below 2^(q + 24) -- every integer
// and half-integer rate (5, 5.5, 2.5 ...) at the block sizes in use -- each of those adds is EXACT, so position k of a window is where_0 + k rate with no rounding: a
// lane evaluates it directly.  The host then only walks the WINDOWS (closed form per window; one window without the CLI's loop), uploads 16 bytes per window instead of
// 8 bytes per output, and the call no longer waits for the stream.  Rates that are not exact in float (3.3 ...) keep the host walk: their positions are the recurrence's.
struct FdWin { int base; float w0; int first_out; int count; };
__global__ __launch_bounds__(256) void k_fracdec_plan(const FdWin *__restrict__ win, int n_win, int n_out, float rate, int *__restrict__ lo_idx, float *__restrict__ frac)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_out) return;
    int a = 0, b = n_win - 1;                                            // the last window whose first output is <= k
    while (a < b) { const int m = (a + b + 1) >> 1; if (win[m].first_out <= k) a = m; else b = m - 1; }
    const FdWin w = win[a];
    const float where = __fmaf_rn((float)(k - w.first_out), rate, w.w0);      // exact (see above): one rounding of an exactly representable value
    const int lo = (int)ceilf(where) - 1;                                // libcsdr.c:763-766: index_high = ceilf(where), FD_INDEX_LOW = index_high - 1
    lo_idx[k] = w.base + lo;
    frac[k] = where - (float)lo;                                         // :774 float xwhere = d->where - FD_INDEX_LOW
}

} // namespace

struct csdr_amd_fracdec {
    float where, rate; int input_processed, num_poly_points, xifirst, xilast, taps_length;
    std::vector<float> denom, taps;
    // cached plan
    float plan_where; int plan_n; bool plan_valid; int plan_outputs, plan_processed; float plan_where_after;
    int cli_bufsize, plan_bufsize;   // > 0: replay the CLI's loop over the_bufsize-sample windows (csdr.c:1511-1524) instead of one call over the whole array
    std::vector<int> lo; std::vector<float> frac;      // per output: first input sample of its window, fractional position (the coefficients are the kernel's)
    int *d_lo; float *d_frac; float *d_denom; float *d_taps; size_t d_cap;
    FdWin *d_win; size_t win_cap;                       // exact rates: the windows of the plan (the outputs' positions are evaluated on the device)
};

extern "C" {

int csdr_amd_fmdemod_quadri_cf(csdr_amd_ctx *c, const csdr_complexf *in, float *out, int n_streams, size_t n,
                               size_t in_pitch, size_t out_pitch, csdr_complexf *last_io)
{
    if (!n || n_streams <= 0) return 0;
    size_t gx = (n + 255) / 256; if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(k_fmdemod, dim3((unsigned)gx, (unsigned)n_streams), dim3(256), 0, c->stream, in, out, n, in_pitch, out_pitch, last_io);
    CSDR_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_store_last, dim3(cdiv(n_streams, 64)), dim3(64), 0, c->stream, in, n, in_pitch, last_io, n_streams);
    CSDR_LAUNCH_CHECK();
    return 0;
}

int csdr_amd_limit_ff(csdr_amd_ctx *c, const float *in, float *out, size_t n, float m)
{
    if (!n) return 0;
    if (((uintptr_t)in | (uintptr_t)out) & 15) return fail_msg(-3, "limit_ff: pointers must be 16-byte aligned");
    size_t g = (n / 4 + 255) / 256; g = g < 1 ? 1 : (g > 2048 ? 2048 : g);
    hipLaunchKernelGGL(k_limit, dim3((unsigned)g), dim3(256), 0, c->stream, in, out, n, m); CSDR_LAUNCH_CHECK();
    return 0;
}
int csdr_amd_gain_ff(csdr_amd_ctx *c, const float *in, float *out, size_t n, float gval)
{
    if (!n) return 0;
    if (((uintptr_t)in | (uintptr_t)out) & 15) return fail_msg(-3, "gain_ff: pointers must be 16-byte aligned");
    size_t g = (n / 4 + 255) / 256; g = g < 1 ? 1 : (g > 2048 ? 2048 : g);
    hipLaunchKernelGGL(k_gain, dim3((unsigned)g), dim3(256), 0, c->stream, in, out, n, gval); CSDR_LAUNCH_CHECK();
    return 0;
}

int csdr_amd_deemphasis_wfm_ff(csdr_amd_ctx *c, const float *in, float *out, int n_streams, size_t n,
                               size_t in_pitch, size_t out_pitch, float tau, int sample_rate, float *last_io)
{
    if (!n || n_streams <= 0) return 0;
    const float dt = (float)(1.0 / sample_rate);            // libcsdr.c:1090-1091, float after a double division
    const float alpha = dt / (tau + dt);
    // few long streams: chunks of 256 samples on their own lanes (k_deemph_wfm_spec, bit exact); M = chunks of run-in: b^(256 M) < 2^-40
    const double b = 1.0 - (double)alpha;
    int M = 0;
    if (n_streams < 32 && n >= 8 * (size_t)DW_L && b > 0.0 && b < 1.0 && in != out) for (int m : {1, 2, 4, 8}) if (256.0 * m * log2(b) < -40.0) { M = m; break; }
    static const bool serial_env = getenv("CSDR_AMD_DEEMPH_SERIAL") != nullptr;       // (A/B, read once per process)
    if (M && !serial_env) {
        const size_t n_chunks = (n + DW_L - 1) / DW_L;
        float *st = (float *)c->get_scratch(0, sizeof(float) * 2 * n_chunks * (size_t)n_streams + 256);
        if (!st) return -2;
        float *st_start = st, *st_end = st + n_chunks * (size_t)n_streams;
        unsigned *flags = (unsigned *)(st_end + n_chunks * (size_t)n_streams);
        CSDR_HIP(hipMemsetAsync(flags, 0, sizeof(unsigned) * (size_t)n_streams, c->stream));
        const dim3 grid((unsigned)cdiv(n_chunks, 64), (unsigned)n_streams);
        const size_t lds = sizeof(float) * (size_t)(M + 64) * DW_P;
#define DW_LAUNCH(MV) do { const int arc = csdr_amd::lds_attr_once((const void *)k_deemph_wfm_spec<MV>, lds); if (arc) return arc;                                   \
                hipLaunchKernelGGL(k_deemph_wfm_spec<MV>, grid, dim3(64), lds, c->stream, in, out, n, in_pitch, out_pitch, alpha, (const float *)last_io, st_start, st_end, n_chunks); } while (0)
        if (M == 1) DW_LAUNCH(1); else if (M == 2) DW_LAUNCH(2); else if (M == 4) DW_LAUNCH(4); else DW_LAUNCH(8);
#undef DW_LAUNCH
        CSDR_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_deemph_wfm_check, dim3((unsigned)cdiv(n_chunks, 256), (unsigned)n_streams), dim3(256), 0, c->stream, (const float *)st_start, (const float *)st_end, n_chunks, flags);
        CSDR_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_deemph_wfm_fix, dim3(1), dim3(64), 0, c->stream, in, out, n, in_pitch, out_pitch, alpha, last_io, (const float *)st_start, st_end, n_chunks, flags, n_streams);
        CSDR_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(k_deemph_wfm, dim3(cdiv(n_streams, 64)), dim3(64), 0, c->stream, in, out, n_streams, n, in_pitch, out_pitch, alpha, last_io);
    CSDR_LAUNCH_CHECK();
    return 0;
}

int csdr_amd_fastagc_ff(csdr_amd_ctx *c, const float *in, float *out, int n_streams, int n_blocks, int block,
                        size_t in_pitch, size_t out_pitch, float reference, float *state_io)
{
    return csdr_amd::fastagc_ff_s16(c, in, out, nullptr, n_streams, n_blocks, block, in_pitch, out_pitch, 0, reference, state_io, false);
}

} // extern "C"

// fastagc_ff with an optional second output: the same samples through convert_f_s16 (used by the NFM chain object; out may be null)
float *csdr_amd::fastagc_peaks_buffer(csdr_amd_ctx *c, int n_streams, int n_blocks)
{
    return (float *)c->get_scratch(3, sizeof(float) * (size_t)n_streams * (n_blocks + 2));
}
// have_peaks: the producer of `in` already wrote every new block's peak |x| to fastagc_peaks_buffer()[s * (n_blocks + 2) + 2 + j] (the NFM chain's de-emphasis kernel)
int csdr_amd::fastagc_ff_s16(csdr_amd_ctx *c, const float *in, float *out, int16_t *out_s16, int n_streams, int n_blocks, int block,
                             size_t in_pitch, size_t out_pitch, size_t s16_pitch, float reference, float *state_io, bool have_peaks)
{
    if (n_blocks <= 0 || n_streams <= 0) return 0;
    float *peaks = fastagc_peaks_buffer(c, n_streams, n_blocks);
    if (!peaks) return -2;
    if (!have_peaks) { hipLaunchKernelGGL(k_agc_peaks, dim3(n_blocks, n_streams), dim3(256), 0, c->stream, in, in_pitch, block, n_blocks, peaks); CSDR_LAUNCH_CHECK(); }
    hipLaunchKernelGGL(k_agc_apply, dim3(n_blocks, n_streams), dim3(256), 0, c->stream, in, out, in_pitch, out_pitch, block, n_blocks, state_io, peaks, reference, out_s16, s16_pitch); CSDR_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_agc_update, dim3(1, n_streams), dim3(256), 0, c->stream, in, in_pitch, block, n_blocks, state_io, peaks, reference); CSDR_LAUNCH_CHECK();
    return 0;
}

extern "C" {

csdr_amd_fracdec *csdr_amd_fracdec_create(float rate, int num_poly_points, const float *host_taps, int taps_length)
{   // fractional_decimator_ff_init libcsdr.c:715-748
    if (!(rate > 1.0f) || num_poly_points < 2 || num_poly_points > 64) { fail_msg(-3, "fracdec: need rate > 1 and 2 <= num_poly_points <= 64"); return nullptr; }
    csdr_amd_fracdec *d = new csdr_amd_fracdec();
    d->num_poly_points = num_poly_points & ~1;
    d->xifirst = -(num_poly_points / 2) + 1; d->xilast = num_poly_points / 2;
    for (int a = d->xifirst; a <= d->xilast; a++) {
        float prod = 1;
        for (int b = d->xifirst; b <= d->xilast; b++) if (a != b) prod *= (a - b);
        d->denom.push_back(prod);
    }
    d->where = (float)(-d->xifirst); d->rate = rate; d->input_processed = 0;
    d->taps_length = host_taps ? taps_length : 0;
    if (d->taps_length) d->taps.assign(host_taps, host_taps + taps_length);
    d->plan_valid = false; d->d_lo = nullptr; d->d_frac = nullptr; d->d_denom = nullptr; d->d_taps = nullptr; d->d_cap = 0; d->cli_bufsize = 0; d->plan_bufsize = 0; d->d_win = nullptr; d->win_cap = 0;
    return d;
}

void  csdr_amd_fracdec_set_where(csdr_amd_fracdec *d, float where) { d->where = where; }
void  csdr_amd_fracdec_set_cli_bufsize(csdr_amd_fracdec *d, int the_bufsize) { d->cli_bufsize = the_bufsize > 0 ? the_bufsize : 0; d->plan_valid = false; }
float csdr_amd_fracdec_get_where(const csdr_amd_fracdec *d) { return d->where; }

void csdr_amd_fracdec_destroy(csdr_amd_fracdec *d)
{
    if (!d) return;
    if (d->d_lo) (void)hipFree(d->d_lo);
    if (d->d_frac) (void)hipFree(d->d_frac);
    if (d->d_denom) (void)hipFree(d->d_denom);
    if (d->d_taps) (void)hipFree(d->d_taps);
    if (d->d_win) (void)hipFree(d->d_win);
    delete d;
}

int csdr_amd_fractional_decimator_ff(csdr_amd_ctx *c, csdr_amd_fracdec *d, const float *in, float *out, int n_streams, int input_size,
                                     size_t in_pitch, size_t out_pitch, int *input_processed)
{
    const int P = d->num_poly_points;
    if (!(d->plan_valid && d->plan_where == d->where && d->plan_n == input_size && d->plan_bufsize == d->cli_bufsize)) {
        // Replay the reference's float position bookkeeping (libcsdr.c:762-792) on the host: it does not depend
        // on the samples, only on (where, rate, input_size).
        d->lo.clear(); d->frac.clear();
        float where = d->where; int hi = 0;
        // exponent of the lowest set bit of a float (a large number for 0): both where and rate on the lattice 2^q, q <= 0 (input_processed is an integer)
        auto lsb_exp = [](float v) { if (v == 0.f) return 127; uint32_t u; memcpy(&u, &v, 4); const int ef = (int)((u >> 23) & 255); uint32_t m = (u & 0x7fffff) | (ef ? 0x800000u : 0u);
                                     return (ef ? ef : 1) - 127 - 23 + __builtin_ctz(m); };
        int q = lsb_exp(where) < lsb_exp(d->rate) ? lsb_exp(where) : lsb_exp(d->rate); if (q > 0) q = 0;
        const int win_size = (d->cli_bufsize > 0 && input_size >= d->cli_bufsize) ? d->cli_bufsize : input_size;
        const bool exact = q > -40 && ldexp(1.0, q + 24) > (double)win_size + fabs((double)d->rate) + P + d->taps_length + 4 && where >= 0.f;
        std::vector<FdWin> wins;
        int n_out = 0;
        // one call of fractional_decimator_ff over in[base .. base + size), exact rates: K outputs from where, then where <- where_K - processed, in closed form
        auto one_call_exact = [&](int base, int size) {
            const int m = size - P - d->taps_length;                     // the loop runs while ceilf(where) < m, i.e. where <= m - 1
            const double w0 = where, r = d->rate;
            long K = 0;
            if (w0 <= (double)(m - 1)) { K = (long)floor(((double)(m - 1) - w0) / r) + 1; while (K > 0 && w0 + (double)(K - 1) * r > (double)(m - 1)) K--; while (w0 + (double)K * r <= (double)(m - 1)) K++; }
            if (K > 0) { wins.push_back(FdWin{base, where, n_out, (int)K}); n_out += (int)K; }
            const float w_end = (float)(w0 + (double)K * r);              // exact
            hi = (int)ceilf(w_end);
            const int processed = (hi - 1) + d->xifirst;
            where = w_end - (float)processed;
            return processed;
        };
        auto one_call = [&](int base, int size) {                       // fractional_decimator_ff over in[base .. base + size)
            if (exact) return one_call_exact(base, size);
            for (; (hi = (int)ceilf(where)) + P + d->taps_length < size; where += d->rate) {
                const int lo = hi - 1;
                const float x = where - lo;
                d->lo.push_back(base + lo);
                d->frac.push_back(x);
            }
            const int processed = (hi - 1) + d->xifirst;                // libcsdr.c:790-791
            where -= processed;
            return processed;
        };
        if (d->cli_bufsize > 0 && input_size >= d->cli_bufsize) {          // (a shorter input = the stream's tail at EOF: one call, as without the switch)
            // `csdr fractional_decimator_ff` calls the function on the_bufsize-sample windows and re-presents the unprocessed tail
            // (csdr.c:1511-1524): `where` stays small, so for rates that are not exact in float the positions differ from one call over
            // the whole array (whose `where` loses precision as it grows).  Same windows, same float arithmetic, one gather kernel.
            int base = 0;
            while (base + d->cli_bufsize <= input_size) {
                const int processed = one_call(base, d->cli_bufsize);
                if (processed <= 0) break;
                base += processed;
            }
            d->plan_processed = base;
            d->plan_where_after = where;
        } else {
            d->plan_processed = one_call(0, input_size);
            d->plan_where_after = where;
        }
        d->plan_bufsize = d->cli_bufsize;
        d->plan_outputs = exact ? n_out : (int)d->lo.size();
        d->plan_where = d->where; d->plan_n = input_size;
        const size_t need = (size_t)d->plan_outputs + 1;
        if (need > d->d_cap) {
            CSDR_HIP(hipStreamSynchronize(c->stream));
            if (d->d_lo) (void)hipFree(d->d_lo);
            if (d->d_frac) (void)hipFree(d->d_frac);
            d->d_cap = need + need / 2;
            CSDR_HIP(hipMalloc((void **)&d->d_lo, sizeof(int) * d->d_cap));
            CSDR_HIP(hipMalloc((void **)&d->d_frac, sizeof(float) * d->d_cap));
        }
        if (!d->d_denom) {
            CSDR_HIP(hipMalloc((void **)&d->d_denom, sizeof(float) * d->denom.size()));
            CSDR_HIP(hipMemcpy(d->d_denom, d->denom.data(), sizeof(float) * d->denom.size(), hipMemcpyHostToDevice));
        }
        if (d->taps_length && !d->d_taps) {
            CSDR_HIP(hipMalloc((void **)&d->d_taps, sizeof(float) * d->taps_length));
            CSDR_HIP(hipMemcpy(d->d_taps, d->taps.data(), sizeof(float) * d->taps_length, hipMemcpyHostToDevice));
        }
        if (d->plan_outputs && exact) {
            // stream ordered: the window table through the context's pinned staging (an earlier launch that still reads the old plan is in front of it on the stream)
            if (wins.size() > d->win_cap) {
                CSDR_HIP(hipStreamSynchronize(c->stream));
                if (d->d_win) (void)hipFree(d->d_win);
                d->win_cap = wins.size() + wins.size() / 2 + 16;
                CSDR_HIP(hipMalloc((void **)&d->d_win, sizeof(FdWin) * d->win_cap));
            }
            FdWin *hw = (FdWin *)c->pinned_acquire(sizeof(FdWin) * wins.size());
            if (!hw) return -2;
            memcpy(hw, wins.data(), sizeof(FdWin) * wins.size());
            const int urc = c->pinned_upload(d->d_win, sizeof(FdWin) * wins.size()); if (urc) return urc;
            hipLaunchKernelGGL(k_fracdec_plan, dim3(cdiv(d->plan_outputs, 256)), dim3(256), 0, c->stream, d->d_win, (int)wins.size(), d->plan_outputs, d->rate, d->d_lo, d->d_frac);
            CSDR_LAUNCH_CHECK();
        } else if (d->plan_outputs) {
            CSDR_HIP(hipStreamSynchronize(c->stream));     // previous launch may still read the old plan
            CSDR_HIP(hipMemcpy(d->d_lo, d->lo.data(), sizeof(int) * d->lo.size(), hipMemcpyHostToDevice));
            CSDR_HIP(hipMemcpy(d->d_frac, d->frac.data(), sizeof(float) * d->frac.size(), hipMemcpyHostToDevice));
        }
        d->plan_valid = true;
    }
    if (d->plan_outputs && n_streams > 0) {
        hipLaunchKernelGGL(k_fracdec, dim3(cdiv(d->plan_outputs, 256), n_streams), dim3(256), 0, c->stream, in, out, d->plan_outputs,
                           in_pitch, out_pitch, d->d_lo, d->d_frac, P, d->xifirst, d->d_denom, d->d_taps, d->taps_length);
        CSDR_LAUNCH_CHECK();
    }
    d->input_processed = d->plan_processed;
    d->where = d->plan_where_after;
    if (input_processed) *input_processed = d->input_processed;
    return d->plan_outputs;
}

} // extern "C"
