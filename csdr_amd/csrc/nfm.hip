// nfm.hip -- the NFM receive chain (BASELINE config 5, README.md:87) for N independent u8 IQ streams, streaming:
//
//   convert_u8_f | shift_addition_cc r | fir_decimate_cc D tbw HAMMING | fmdemod_quadri_cf | limit_ff | deemphasis_nfm_ff fs | fastagc_ff | convert_f_s16
//
//   front end (2.4 MS/s -> 48 kS/s): csdr_amd_ddc (ddc_mfma.hip), one pass over the input on the matrix cores;
//   back end  (48 kS/s, 1/50 of the input rate): k_nfm_demod_limit (fmdemod_quadri_cf libcsdr.c:1040-1071 + limit_ff :1130-1137 in one
//   pass, appended behind the de-emphasis filter's unconsumed input), the fixed de-emphasis FIR (libcsdr.c:1101-1128, the CLI's re-feed
//   loop csdr.c:1083), fastagc_ff (libcsdr.c:946-991) over whole blocks and convert_f_s16 (:2397).  The de-emphasis filter is run for whole
//   AGC blocks only (its remaining input waits in the carry buffer), so nothing else needs a carry.
//
//   The 201-tap de-emphasis FIR was 80 % of the back end as a float VALU kernel (0.53 ms of 0.66 ms for 512 channels x 1 s: 18 TFLOP/s).  Its
//   input is the LIMITED demodulator output, |x| <= max_amplitude, so 24-bit fixed point represents it to 6e-8 of full scale: the demodulator
//   kernel writes three int8 digit planes instead of floats, and the FIR becomes a banded int8 product on v_mfma_i32_16x16x64_i8 (16 consecutive
//   outputs x 16 channels x 256 inputs per tile, band 93 % dense, one resident weight set): k_nfm_deemph_mfma.  Digit pairs of equal weight
//   share an accumulator; only the (low x low) pair is dropped (2^-31 of full scale).
#include "common.hpp"
#include "nfm_demod.hpp"
#include <math.h>
#include <string.h>
#include <string>
#include <vector>
using namespace csdr_amd;

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
constexpr float NFM_XQ = 8355711.0f;       // 127*65536 + 127*256 + 127: full scale of three balanced base-256 digits
constexpr int NFM_FIR_NK = 4;              // 64-input K-steps per tile of 16 outputs: taps <= 241

// planes: [3][n_streams][dl_pitch] int8 -- digit j of round(v / max_amp * NFM_XQ) = d0 * 65536 + d1 * 256 + d2
__global__ __launch_bounds__(256) void k_nfm_demod_limit(const cf32 *__restrict__ y, size_t y_pitch, int n, const cf32 *__restrict__ last,
                                                         int8_t *__restrict__ planes, size_t plane_bytes, size_t dl_pitch, int dl_fill, float max_amp, float q_per_amp)
{
    const int s = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const cf32 *src = y + (size_t)s * y_pitch;
    const cf32 x = src[k];
    const cf32 p = k ? src[k - 1] : last[s];
    int d[3]; nfm_demod_digits(make_float2(x.i, x.q), make_float2(p.i, p.q), max_amp, q_per_amp, d);
    int8_t *dst = planes + (size_t)s * dl_pitch + dl_fill + k;
    dst[0] = (int8_t)d[0]; dst[plane_bytes] = (int8_t)d[1]; dst[2 * plane_bytes] = (int8_t)d[2];
}

// The same for the outputs the fused front end left over (DdcFuseInfo): the leading outputs [0, n_lead) (plain kernel), the first output of every segment of the
// matrix-core kernel (its predecessor belongs to another workgroup) and the trailing outputs; y holds these samples and their predecessors.
// Thread 0 of a stream also keeps the call's last sample as the next call's predecessor (two `last` buffers: k_nfm_store_last's work).
__global__ __launch_bounds__(64) void k_nfm_demod_boundary(const cf32 *__restrict__ y, size_t y_pitch, const cf32 *__restrict__ last, cf32 *__restrict__ last_out, int n_y,
                                                           DdcFuseInfo info, DdcFuse f)
{
    const int s = blockIdx.y, j = blockIdx.x * 64 + threadIdx.x;
    if (j == 0) last_out[s] = y[(size_t)s * y_pitch + n_y - 1];
    long k;
    if (j < info.n_lead) k = j;
    else if (j < info.n_lead + info.n_seg) { k = info.seg_first + (long)(j - info.n_lead) * info.seg_outputs; if (k < info.n_lead) return; }      // (a partial first tile: the lead covers output 0)
    else if (j < info.n_lead + info.n_seg + info.n_trail) k = info.trail_first + (j - info.n_lead - info.n_seg);
    else return;
    const cf32 *src = y + (size_t)s * y_pitch;
    const cf32 x = src[k];
    const cf32 p = k ? src[k - 1] : last[s];
    int d[3]; nfm_demod_digits(make_float2(x.i, x.q), make_float2(p.i, p.q), f.max_amp, f.q_per_amp, d);
    int8_t *dst = f.planes + (size_t)s * f.dl_pitch + f.dl_fill + k;
    dst[0] = (int8_t)d[0]; dst[f.plane_bytes] = (int8_t)d[1]; dst[2 * f.plane_bytes] = (int8_t)d[2];
}

// out[s][i] = sum_t taps[t] x[s][i + t] for i < n_tiles * 16, x given as digit planes.  One workgroup = 16 channels x NFM_FIR_SPAN consecutive
// tiles: the three planes' bytes of that span (+ the 240-sample window tail) are staged in LDS once (consecutive tiles overlap by 240 of
// their 256 inputs; reading the B operands straight from global memory made this kernel L2-bandwidth bound: 1.2 GB of L2 reads for 74 MB of
// planes, 0.16 ms); the 12 weight fragments (4 K-steps x 3 digits of the Toeplitz band) stay in registers; each wave takes every fourth tile.
// MFMA result layout: lane (col, q) holds outputs 4q .. 4q+3 of channel col.
constexpr int NFM_FIR_SPAN = 64;                                   // tiles per workgroup (1024 outputs)
#ifndef NFM_FIR_SUB_N
#define NFM_FIR_SUB_N 32
#endif
constexpr int NFM_FIR_SUB = NFM_FIR_SUB_N;                         // tiles staged at a time: the span in SPAN / SUB passes (64: 61 KiB of LDS, two workgroups per CU; 32: 37 KiB, four)
constexpr int NFM_FIR_ROW = 16 * NFM_FIR_SUB + 64 * NFM_FIR_NK;    // staged bytes per channel and plane
constexpr int NFM_FIR_RP = NFM_FIR_ROW + 16;                       // LDS pitch: odd multiple of 16 bytes (the 16 channels of a B read hit different banks)

// peaks != nullptr (fastagc block = the workgroup's span of 1024 outputs): the workgroup also leaves max |out| of its span per channel at
// peaks[stream * peak_pitch + 2 + blockIdx.x] -- fastagc_ff's first pass (libcsdr.c:957-962) without reading the audio back.
__global__ __launch_bounds__(256) void k_nfm_deemph_mfma(const int8_t *__restrict__ planes, size_t plane_bytes, size_t dl_pitch, const v4i *__restrict__ frags,
                                                         float scale, float *__restrict__ out, size_t out_pitch, int n_tiles, int n_streams,
                                                         float *__restrict__ peaks, int peak_pitch)
{
    __shared__ __attribute__((aligned(16))) int8_t lds[3 * 16 * NFM_FIR_RP];
    __shared__ float wpeak[4][16];
    const int tid = threadIdx.x, lane = tid & 63, col = lane & 15, q = lane >> 4, wv = tid >> 6;
    v4i A[NFM_FIR_NK * 3];
#pragma unroll
    for (int i = 0; i < NFM_FIR_NK * 3; i++) A[i] = frags[i * 64 + lane];
    const int last = n_streams - 1;
    const int stream = (int)blockIdx.y * 16 + col;
    const int8_t *row = lds + col * NFM_FIR_RP + 16 * q;
    float pmax = 0.f;
    for (int sub = 0; sub < NFM_FIR_SPAN / NFM_FIR_SUB; sub++) {
        const int tile0 = blockIdx.x * NFM_FIR_SPAN + sub * NFM_FIR_SUB;
        const int nt = min(NFM_FIR_SUB, n_tiles - tile0);
        if (nt <= 0) break;                                           // (uniform)
        if (sub) __syncthreads();                                     // everyone has read the previous pass's rows
        // stage [3 planes][16 channels][NFM_FIR_ROW] (rows of channels past the last one re-read it; bytes behind the valid samples meet zero weights)
        for (int i = tid; i < 3 * 16 * (NFM_FIR_ROW / 16); i += 256) {
            const int piece = i % (NFM_FIR_ROW / 16), r = (i / (NFM_FIR_ROW / 16)) % 16, pl = i / (16 * (NFM_FIR_ROW / 16));
            const int st = min((int)blockIdx.y * 16 + r, last);
            *reinterpret_cast<v4i *>(lds + (pl * 16 + r) * NFM_FIR_RP + 16 * piece) =
                *reinterpret_cast<const v4i *>(planes + (size_t)pl * plane_bytes + (size_t)st * dl_pitch + (size_t)tile0 * 16 + 16 * piece);
        }
        __syncthreads();
        for (int t = wv; t < nt; t += 4) {
            const int8_t *src = row + 16 * t;
            v4i acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
            for (int ks = 0; ks < NFM_FIR_NK; ks++) {
                const v4i x0 = *reinterpret_cast<const v4i *>(src + 64 * ks), x1 = *reinterpret_cast<const v4i *>(src + 64 * ks + 16 * NFM_FIR_RP),
                          x2 = *reinterpret_cast<const v4i *>(src + 64 * ks + 32 * NFM_FIR_RP);
                const v4i w0 = A[ks * 3], w1 = A[ks * 3 + 1], w2 = A[ks * 3 + 2];
                acc[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w0, x0, acc[0], 0, 0, 0);          // 2^32
                acc[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w0, x1, acc[1], 0, 0, 0);          // 2^24
                acc[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w1, x0, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w0, x2, acc[2], 0, 0, 0);          // 2^16
                acc[2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w1, x1, acc[2], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w2, x0, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w1, x2, acc[3], 0, 0, 0);          // 2^8
                acc[3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w2, x1, acc[3], 0, 0, 0);
            }
            float4 r;
            float *rv = reinterpret_cast<float *>(&r);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float v = fmaf((float)acc[0][j], 256.0f, (float)acc[1][j]);
                v = fmaf(v, 256.0f, (float)acc[2][j]);
                v = fmaf(v, 256.0f, (float)acc[3][j]);
                rv[j] = v * scale;
            }
            if (stream < n_streams)
                *reinterpret_cast<float4 *>(out + (size_t)stream * out_pitch + 16 * (size_t)(tile0 + t) + 4 * q) = r;
            pmax = fmaxf(fmaxf(pmax, fmaxf(fabsf(r.x), fabsf(r.y))), fmaxf(fabsf(r.z), fabsf(r.w)));
        }
    }
    if (peaks) {                                                      // uniform
        pmax = fmaxf(pmax, __shfl_xor(pmax, 16)); pmax = fmaxf(pmax, __shfl_xor(pmax, 32));      // over q: the channel's outputs of this wave's tiles
        if (q == 0) wpeak[wv][col] = pmax;
        __syncthreads();
        if (tid < 16 && (int)blockIdx.y * 16 + tid < n_streams)
            peaks[(size_t)((int)blockIdx.y * 16 + tid) * peak_pitch + 2 + blockIdx.x] = fmaxf(fmaxf(wpeak[0][tid], wpeak[1][tid]), fmaxf(wpeak[2][tid], wpeak[3][tid]));
    }
}

// Toeplitz band of the de-emphasis taps as three base-256 digits: row o (output), column t (input): taps[t - o]; lane / byte layout of the A
// operand of v_mfma_i32_16x16x64_i8.  Returns the scale of the recombined product:
//   value = (a0 2^24 + a1 2^16 + a2 2^8 + a3) * 2^8 * (gmax / 2^22) * (max_amp / NFM_XQ)
float nfm_fir_table(const float *taps, int Ld, float limit_max, std::vector<int8_t> &fr)
{
    fr.assign((size_t)NFM_FIR_NK * 3 * 64 * 16, 0);
    double gmax = 0;
    for (int k = 0; k < Ld; k++) gmax = fmax(gmax, fabs((double)taps[k]));
    gmax *= 1.0001; if (gmax == 0) gmax = 1;
    const double qscale = 4194304.0 / gmax;
    for (int o = 0; o < 16; o++) for (int tp = 0; tp < Ld; tp++) {
        const int t = o + tp, ks = t / 64, b = t % 64;
        long qv = lrint((double)taps[tp] * qscale);
        const int w2 = (int)(((qv + 128) % 256 + 256) % 256) - 128; qv = (qv - w2) / 256;
        const int w1 = (int)(((qv + 128) % 256 + 256) % 256) - 128; qv = (qv - w1) / 256;
        const int dig[3] = {(int)qv, w1, w2};
        for (int l = 0; l < 3; l++) fr[((size_t)(ks * 3 + l) * 64 + (16 * (b / 16) + o)) * 16 + b % 16] = (int8_t)dig[l];
    }
    return (float)(256.0 * (gmax / 4194304.0) * ((double)limit_max / (double)NFM_XQ));
}

// the digits k_nfm_demod_limit stores for a limited sample v (|v| <= max_amp)
inline void nfm_sample_digits(float v, float q_per_amp, int (&d)[3])
{
    int qv = (int)lrintf(v * q_per_amp);
    d[2] = ((qv + 128) & 255) - 128; qv = (qv - d[2]) >> 8;
    d[1] = ((qv + 128) & 255) - 128; qv = (qv - d[1]) >> 8;
    d[0] = qv;
}

// bytes: buf[s][0 .. count) = buf[s][src_off .. src_off + count) on each of the three planes (ranges may overlap; count <= 2048)
__global__ __launch_bounds__(256) void k_nfm_move_front_planes(int8_t *__restrict__ planes, size_t plane_bytes, size_t pitch, int src_off, int count)
{
    int8_t *row = planes + (size_t)blockIdx.y * plane_bytes + (size_t)blockIdx.x * pitch;
    int8_t v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { const int i = threadIdx.x + 256 * j; v[j] = i < count ? row[src_off + i] : (int8_t)0; }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; j++) { const int i = threadIdx.x + 256 * j; if (i < count) row[i] = v[j]; }
}

__global__ void k_nfm_store_last(const cf32 *__restrict__ y, size_t y_pitch, int n, cf32 *__restrict__ last, int n_streams)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n_streams) last[s] = y[(size_t)s * y_pitch + n - 1];
}

} // namespace

namespace csdr_amd { long ddc_process_fused(csdr_amd_ddc *d, const uint8_t *in, size_t in_pitch, size_t block_samples, csdr_complexf *out, size_t out_pitch, const DdcFuse *fuse, DdcFuseInfo *info); }

struct csdr_amd_nfm {
    csdr_amd_ctx *ctx;
    int n_streams, D, Ld, agc_block, cli_prefix;
    float limit, agc_ref;
    csdr_amd_ddc *ddc;
    cf32 *d_y; size_t y_pitch; cf32 *d_last, *d_last2; int lflip;
    int8_t *d_planes; size_t plane_bytes; size_t dl_pitch; int dl_fill;   // limited demodulator output (three digit planes) waiting for the de-emphasis filter
    void *d_fir_frags; float fir_scale;             // de-emphasis taps as int8 digit fragments; scale of the recombined product
    float *d_de; size_t a_pitch;                    // de-emphasised blocks
    float *d_agc_state;
    size_t max_y;
    bool fuse_off;                                  // CSDR_AMD_NFM_FUSE=0 (A/B: separate demodulator / AGC-peak passes), read when the object is created
};

extern "C" {

static csdr_amd_nfm *nfm_create_impl(csdr_amd_ctx *ctx, int n_streams, const float *rates, bool per_stream, int decimation, const float *host_taps, int taps_length,
                                     int audio_rate, int agc_block, float agc_reference, float limit_max, size_t max_block_samples)
{
    if (!ctx || n_streams < 1 || agc_block < 1 || agc_block > 2048 - 512 || (agc_block % 16) || !(limit_max > 0)) { fail_msg(-3, "nfm_create: bad arguments (agc_block: multiple of 16 up to 1536; limit > 0)"); return nullptr; }
    const float *dt = nullptr;
    const int Ld = csdr_amd_nfm_deemph_taps(audio_rate, &dt);
    if (!Ld) { fail_msg(-3, "nfm_create: no de-emphasis table for sample rate %d (libcsdr.c:1115-1119)", audio_rate); return nullptr; }
    if (Ld + 15 > 64 * NFM_FIR_NK) { fail_msg(-3, "nfm_create: %d de-emphasis taps exceed the matrix-core tile", Ld); return nullptr; }
    if (max_block_samples < 1024) max_block_samples = 1024;
    csdr_amd_nfm *w = new csdr_amd_nfm();
    memset(w, 0, sizeof(*w));
    w->ctx = ctx; w->n_streams = n_streams; w->D = decimation; w->Ld = Ld; w->agc_block = agc_block; w->limit = limit_max; w->agc_ref = agc_reference;
    w->ddc = per_stream ? csdr_amd_ddc_create_rates(ctx, n_streams, rates, decimation, host_taps, taps_length, max_block_samples)
                        : csdr_amd_ddc_create(ctx, n_streams, rates[0], decimation, host_taps, taps_length, max_block_samples);
    if (!w->ddc) { delete w; return nullptr; }
    { const char *fe = getenv("CSDR_AMD_NFM_FUSE"); w->fuse_off = fe && atoi(fe) == 0; }
    w->max_y = max_block_samples / decimation + 2;
    w->y_pitch = (w->max_y + 15) & ~(size_t)15;
    // The reference's `csdr deemphasis_nfm_ff` runs its FIR over its freshly allocated buffer before it reads anything (csdr.c:1076-1081: `processed`
    // starts at 0, the first fread is empty): its output is the FIR of  the_bufsize zeros ++ stream.  The chain object reproduces the pipeline
    // at the default buffer size (1024, csdr.c:189).
    w->cli_prefix = 1024;
    w->dl_pitch = (w->max_y + Ld + agc_block + w->cli_prefix + NFM_FIR_ROW + 63) & ~(size_t)63;        // + the last workgroup's staged span beyond the valid samples (zero weights)
    w->a_pitch = (w->max_y + Ld + agc_block + 1024 + 15) & ~(size_t)15;
    hipError_t e = hipSuccess;
    auto alloc = [&](void **p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes); };
    alloc((void **)&w->d_y, sizeof(cf32) * w->y_pitch * n_streams);
    alloc((void **)&w->d_last, sizeof(cf32) * n_streams);
    alloc((void **)&w->d_last2, sizeof(cf32) * n_streams);
    w->plane_bytes = w->dl_pitch * n_streams;
    alloc((void **)&w->d_planes, 3 * w->plane_bytes);
    alloc(&w->d_fir_frags, (size_t)NFM_FIR_NK * 3 * 64 * 16);
    alloc((void **)&w->d_de, sizeof(float) * w->a_pitch * n_streams);
    alloc((void **)&w->d_agc_state, sizeof(float) * (size_t)n_streams * (2 * agc_block + 4));
    {
        std::vector<int8_t> fr;
        w->fir_scale = nfm_fir_table(dt, Ld, limit_max, fr);
        if (e == hipSuccess) e = hipMemcpy(w->d_fir_frags, fr.data(), fr.size(), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) { fail(e, "hipMalloc/hipMemcpy(nfm state)", __FILE__, __LINE__); csdr_amd_nfm_destroy(w); return nullptr; }
    if (csdr_amd_nfm_reset(w)) { csdr_amd_nfm_destroy(w); return nullptr; }
    return w;
}

csdr_amd_nfm *csdr_amd_nfm_create(csdr_amd_ctx *ctx, int n_streams, float shift_rate, int decimation, const float *host_taps, int taps_length,
                                  int audio_rate, int agc_block, float agc_reference, float limit_max, size_t max_block_samples)
{
    return nfm_create_impl(ctx, n_streams, &shift_rate, false, decimation, host_taps, taps_length, audio_rate, agc_block, agc_reference, limit_max, max_block_samples);
}

csdr_amd_nfm *csdr_amd_nfm_create_rates(csdr_amd_ctx *ctx, int n_streams, const float *shift_rates, int decimation, const float *host_taps, int taps_length,
                                        int audio_rate, int agc_block, float agc_reference, float limit_max, size_t max_block_samples)
{
    if (!shift_rates) { fail_msg(-3, "nfm_create_rates: no rates"); return nullptr; }
    return nfm_create_impl(ctx, n_streams, shift_rates, true, decimation, host_taps, taps_length, audio_rate, agc_block, agc_reference, limit_max, max_block_samples);
}

int csdr_amd_nfm_set_rate(csdr_amd_nfm *w, int stream, float shift_rate) { return csdr_amd_ddc_set_rate(w->ddc, stream, shift_rate); }

void csdr_amd_nfm_destroy(csdr_amd_nfm *w)
{
    if (!w) return;
    (void)hipStreamSynchronize(w->ctx->stream);
    if (w->ddc) csdr_amd_ddc_destroy(w->ddc);
    (void)hipFree(w->d_y); (void)hipFree(w->d_last); (void)hipFree(w->d_last2); (void)hipFree(w->d_planes); (void)hipFree(w->d_fir_frags); (void)hipFree(w->d_de);
    (void)hipFree(w->d_agc_state);
    delete w;
}

int csdr_amd_nfm_reset(csdr_amd_nfm *w)
{
    hipStream_t st = w->ctx->stream;
    w->dl_fill = w->cli_prefix;                                                                                // zeros (the planes are cleared below): see csdr_amd_nfm_create
    CSDR_HIP(hipMemsetAsync(w->d_planes, 0, 3 * w->plane_bytes, st));                                          // bytes behind the valid samples meet zero weights, but must be initialised
    CSDR_HIP(hipMemsetAsync(w->d_last, 0, sizeof(cf32) * w->n_streams, st));                                   // the CLI starts fmdemod from (0, 0) (csdr.c:1044)
    CSDR_HIP(hipMemsetAsync(w->d_last2, 0, sizeof(cf32) * w->n_streams, st)); w->lflip = 0;
    CSDR_HIP(hipMemsetAsync(w->d_agc_state, 0, sizeof(float) * (size_t)w->n_streams * (2 * w->agc_block + 4), st));   // calloc'ed fastagc state (csdr.c:1393-1394)
    return csdr_amd_ddc_reset(w->ddc);
}

csdr_amd_ddc *csdr_amd_nfm_front_end(csdr_amd_nfm *w) { return w->ddc; }

long csdr_amd_nfm_process(csdr_amd_nfm *w, const uint8_t *in, size_t in_pitch, size_t block_samples, int16_t *audio_s16, float *audio_f, size_t out_pitch)
{
    csdr_amd_ctx *c = w->ctx; hipStream_t st = c->stream;
    const int S = w->n_streams;
    // front end; on its matrix-core path the reducer epilogue demodulates, limits and writes the digit planes itself (the decimated complex stream never goes
    // to HBM: only the samples at workgroup / kernel boundaries do, for k_nfm_demod_boundary).  CSDR_AMD_NFM_FUSE=0: the separate pass over y.
    const bool fuse_off = w->fuse_off;
    DdcFuse fz; fz.planes = w->d_planes; fz.plane_bytes = w->plane_bytes; fz.dl_pitch = w->dl_pitch; fz.dl_fill = w->dl_fill; fz.max_amp = w->limit; fz.q_per_amp = NFM_XQ / w->limit;
    DdcFuseInfo fi; memset(&fi, 0, sizeof fi);
    const long n_y = ddc_process_fused(w->ddc, in, in_pitch, block_samples, w->d_y, w->y_pitch, fuse_off ? nullptr : &fz, &fi);
    if (n_y < 0) return n_y;
    if (n_y == 0) return 0;
    if ((size_t)n_y > w->max_y) return fail_msg(-3, "nfm: front end produced more than the planned %zu samples", w->max_y);
    cf32 *last_in = w->lflip ? w->d_last2 : w->d_last, *last_out = w->lflip ? w->d_last : w->d_last2;
    if (fi.fused) {
        const long nb_out = fi.n_lead + fi.n_seg + fi.n_trail;
        hipLaunchKernelGGL(k_nfm_demod_boundary, dim3(cdiv(nb_out > 0 ? nb_out : 1, 64), S), dim3(64), 0, st, w->d_y, w->y_pitch, last_in, last_out, (int)n_y, fi, fz);
        CSDR_LAUNCH_CHECK();
    } else {
        // fmdemod_quadri_cf | limit_ff, appended behind the filter's unconsumed input
        hipLaunchKernelGGL(k_nfm_demod_limit, dim3(cdiv(n_y, 256), S), dim3(256), 0, st, w->d_y, w->y_pitch, (int)n_y, last_in, w->d_planes, w->plane_bytes, w->dl_pitch, w->dl_fill, w->limit, NFM_XQ / w->limit);
        CSDR_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_nfm_store_last, dim3(cdiv(S, 64)), dim3(64), 0, st, w->d_y, w->y_pitch, (int)n_y, last_out, S);
        CSDR_LAUNCH_CHECK();
    }
    w->lflip ^= 1;
    const int n_in = w->dl_fill + (int)n_y;
    // deemphasis_nfm_ff produces input - taps outputs (libcsdr.c:1121); run it for whole AGC blocks only
    const int ne_all = n_in - w->Ld;
    const int nb = ne_all > 0 ? ne_all / w->agc_block : 0;
    const int ne = nb * w->agc_block;
    if (nb > 0) {
        if ((size_t)ne > out_pitch && S > 1) return fail_msg(-3, "nfm: out_pitch %zu smaller than the %d audio samples of this block", out_pitch, ne);
        {
            const int n_tiles = ne / 16, n_sb = (S + 15) / 16;
            const int gx = (n_tiles + NFM_FIR_SPAN - 1) / NFM_FIR_SPAN;
            // fastagc's peak pass rides in the de-emphasis kernel when an AGC block is exactly a workgroup's span
            const bool fuse_peaks = w->agc_block == 16 * NFM_FIR_SPAN && !fuse_off;
            float *peaks = fuse_peaks ? fastagc_peaks_buffer(c, S, nb) : nullptr;
            if (fuse_peaks && !peaks) return -2;
            hipLaunchKernelGGL(k_nfm_deemph_mfma, dim3(gx, n_sb), dim3(256), 0, st, w->d_planes, w->plane_bytes, w->dl_pitch, (const v4i *)w->d_fir_frags, w->fir_scale,
                               w->d_de, w->a_pitch, n_tiles, S, peaks, nb + 2);
            CSDR_LAUNCH_CHECK();
            // fastagc_ff | convert_f_s16 in one pass (the float audio is written only when the caller wants the parity tap)
            int rc = fastagc_ff_s16(c, w->d_de, audio_f, audio_s16, S, nb, w->agc_block, w->a_pitch, out_pitch, out_pitch, w->agc_ref, w->d_agc_state, fuse_peaks);
            if (rc) return rc;
        }
    }
    // keep the unconsumed filter input in front
    const int rem = n_in - ne;
    if (ne > 0 && rem > 0) {
        if (rem > 2048) return fail_msg(-3, "nfm: internal carry of %d samples", rem);
        hipLaunchKernelGGL(k_nfm_move_front_planes, dim3(S, 3), dim3(256), 0, st, w->d_planes, w->plane_bytes, w->dl_pitch, ne, rem);
        CSDR_LAUNCH_CHECK();
    }
    w->dl_fill = rem;
    return ne;
}

} // extern "C"

// Test hook (tests/test_abi_cpu.py): ONE tile of k_nfm_deemph_mfma on the CPU -- the digit planes of 256 limited samples (as k_nfm_demod_limit
// stores them), the Toeplitz digit table and the four accumulator classes, exactly as the kernel combines them.  x: 256 floats, |x| <= max_amp;
// out16: the 16 outputs out[i] = sum_t taps[t] x[i + t].
extern "C" int csdr_amd_debug_nfm_deemph_tile(int audio_rate, float max_amp, const float *x, float *out16)
{
    const float *dt = nullptr;
    const int Ld = csdr_amd_nfm_deemph_taps(audio_rate, &dt);
    if (!Ld || Ld + 15 > 64 * NFM_FIR_NK || !(max_amp > 0)) return -1;
    std::vector<int8_t> fr;
    const float scale = nfm_fir_table(dt, Ld, max_amp, fr);
    const float q_per_amp = NFM_XQ / max_amp;
    int xd[256][3];
    for (int t = 0; t < 256; t++) nfm_sample_digits(x[t], q_per_amp, xd[t]);
    for (int o = 0; o < 16; o++) {
        long acc[4] = {0, 0, 0, 0};
        for (int ks = 0; ks < NFM_FIR_NK; ks++) for (int b = 0; b < 64; b++) {
            const int t = 64 * ks + b;
            int w[3];
            for (int l = 0; l < 3; l++) w[l] = fr[((size_t)(ks * 3 + l) * 64 + (16 * (b / 16) + o)) * 16 + b % 16];
            acc[0] += (long)w[0] * xd[t][0];
            acc[1] += (long)w[0] * xd[t][1] + (long)w[1] * xd[t][0];
            acc[2] += (long)w[0] * xd[t][2] + (long)w[1] * xd[t][1] + (long)w[2] * xd[t][0];
            acc[3] += (long)w[1] * xd[t][2] + (long)w[2] * xd[t][1];
        }
        float v = fmaf((float)acc[0], 256.0f, (float)acc[1]);
        v = fmaf(v, 256.0f, (float)acc[2]);
        v = fmaf(v, 256.0f, (float)acc[3]);
        out16[o] = v * scale;
    }
    return 0;
}

