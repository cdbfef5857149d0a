// nfm.hip -- the NFM receive chain (BASELINE config 5, README.md:87) for N independent u8 IQ streams, streaming:
//
//   convert_u8_f | shift_addition_cc r | fir_decimate_cc D tbw HAMMING | fmdemod_quadri_cf | limit_ff | deemphasis_nfm_ff fs | fastagc_ff | convert_f_s16
//
//   front end (2.4 MS/s -> 48 kS/s): csdr_amd_ddc (ddc_mfma.hip), one pass over the input on the matrix cores;
//   back end  (48 kS/s, 1/50 of the input rate): k_nfm_demod_limit (fmdemod_quadri_cf libcsdr.c:1040-1071 + limit_ff :1130-1137 in one
//   pass, appended behind the de-emphasis filter's unconsumed input), the fixed de-emphasis FIR (libcsdr.c:1101-1128, the CLI's re-feed
//   loop csdr.c:1083), fastagc_ff (libcsdr.c:946-991) over whole blocks and convert_f_s16 (:2397) -- the device-batch operators of this
//   library.  The de-emphasis filter is run for whole AGC blocks only (its remaining input waits in the carry buffer), so nothing else
//   needs a carry.
#include "common.hpp"
#include <math.h>
#include <string.h>
#include <string>
using namespace csdr_amd;

namespace {

__global__ __launch_bounds__(256) void k_nfm_demod_limit(const cf32 *__restrict__ y, size_t y_pitch, int n, const cf32 *__restrict__ last,
                                                         float *__restrict__ dl, size_t dl_pitch, int dl_fill, float max_amp)
{
    const float Kf = 0.340447550238101026565118445432744920253753662109375f;   // libcsdr.c:1021
    const int s = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const cf32 *src = y + (size_t)s * y_pitch;
    const cf32 x = src[k];
    const cf32 p = k ? src[k - 1] : last[s];
    const float dq = x.q - p.q, di = x.i - p.i;
    const float num = x.i * dq - x.q * di, den = x.i * x.i + x.q * x.q;
    float rd = __builtin_amdgcn_rcpf(den);                                     // same evaluation as k_fmdemod (audio.hip)
    rd = fmaf(fmaf(-den, rd, 1.0f), rd, rd);
    float v = (den != 0.f) ? (Kf * num) * rd : 0.f;
    v = (max_amp < v) ? max_amp : v; v = (-max_amp > v) ? -max_amp : v;        // limit_ff libcsdr.c:1133-1136
    dl[(size_t)s * dl_pitch + dl_fill + k] = v;
}

__global__ void k_nfm_store_last(const cf32 *__restrict__ y, size_t y_pitch, int n, cf32 *__restrict__ last, int n_streams)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n_streams) last[s] = y[(size_t)s * y_pitch + n - 1];
}

// buf[s][0 .. count) = buf[s][src_off .. src_off + count)   (ranges may overlap: staged through registers; count <= 256 * 8)
__global__ __launch_bounds__(256) void k_nfm_move_front(float *__restrict__ buf, size_t pitch, int src_off, int count)
{
    float *row = buf + (size_t)blockIdx.x * pitch;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { const int i = threadIdx.x + 256 * j; v[j] = i < count ? row[src_off + i] : 0.f; }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; j++) { const int i = threadIdx.x + 256 * j; if (i < count) row[i] = v[j]; }
}

// convert_f_s16 (libcsdr.c:2397: (short)(int)(x*32767), x86 truncation semantics) of the AGC output into the caller's buffers
__global__ __launch_bounds__(256) void k_nfm_out(const float *__restrict__ agc, size_t agc_pitch, int n, int16_t *__restrict__ s16, float *__restrict__ af, size_t out_pitch)
{
    const int s = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const float x = agc[(size_t)s * agc_pitch + k];
    const float scaled = x * 32767.0f;
    const int iv = (scaled >= -2147483648.0f && scaled < 2147483648.0f) ? (int)scaled : (int)0x80000000;
    s16[(size_t)s * out_pitch + k] = (int16_t)iv;
    if (af) af[(size_t)s * out_pitch + k] = x;
}

} // namespace

struct csdr_amd_nfm {
    csdr_amd_ctx *ctx;
    int n_streams, D, Ld, agc_block;
    float limit, agc_ref;
    csdr_amd_ddc *ddc;
    cf32 *d_y; size_t y_pitch; cf32 *d_last;
    float *d_dl; size_t dl_pitch; int dl_fill;      // limited demodulator output waiting for the de-emphasis filter
    float *d_de, *d_agc; size_t a_pitch;            // de-emphasised blocks, AGC output
    float *d_dtaps, *d_agc_state;
    size_t max_y;
};

extern "C" {

csdr_amd_nfm *csdr_amd_nfm_create(csdr_amd_ctx *ctx, int n_streams, float shift_rate, int decimation, const float *host_taps, int taps_length,
                                  int audio_rate, int agc_block, float agc_reference, float limit_max, size_t max_block_samples)
{
    if (!ctx || n_streams < 1 || agc_block < 1 || agc_block > 2048 - 512) { fail_msg(-3, "nfm_create: bad arguments (agc_block 1..1536)"); return nullptr; }
    const float *dt = nullptr;
    const int Ld = csdr_amd_nfm_deemph_taps(audio_rate, &dt);
    if (!Ld) { fail_msg(-3, "nfm_create: no de-emphasis table for sample rate %d (libcsdr.c:1115-1119)", audio_rate); return nullptr; }
    if (max_block_samples < 1024) max_block_samples = 1024;
    csdr_amd_nfm *w = new csdr_amd_nfm();
    memset(w, 0, sizeof(*w));
    w->ctx = ctx; w->n_streams = n_streams; w->D = decimation; w->Ld = Ld; w->agc_block = agc_block; w->limit = limit_max; w->agc_ref = agc_reference;
    w->ddc = csdr_amd_ddc_create(ctx, n_streams, shift_rate, decimation, host_taps, taps_length, max_block_samples);
    if (!w->ddc) { delete w; return nullptr; }
    w->max_y = max_block_samples / decimation + 2;
    w->y_pitch = (w->max_y + 15) & ~(size_t)15;
    w->dl_pitch = (w->max_y + Ld + agc_block + 15) & ~(size_t)15;
    w->a_pitch = (w->max_y + Ld + agc_block + 15) & ~(size_t)15;
    hipError_t e = hipSuccess;
    auto alloc = [&](void **p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes); };
    alloc((void **)&w->d_y, sizeof(cf32) * w->y_pitch * n_streams);
    alloc((void **)&w->d_last, sizeof(cf32) * n_streams);
    alloc((void **)&w->d_dl, sizeof(float) * w->dl_pitch * n_streams);
    alloc((void **)&w->d_de, sizeof(float) * w->a_pitch * n_streams);
    alloc((void **)&w->d_agc, sizeof(float) * w->a_pitch * n_streams);
    alloc((void **)&w->d_dtaps, sizeof(float) * Ld);
    alloc((void **)&w->d_agc_state, sizeof(float) * (size_t)n_streams * (2 * agc_block + 4));
    if (e == hipSuccess) e = hipMemcpy(w->d_dtaps, dt, sizeof(float) * Ld, hipMemcpyHostToDevice);
    if (e != hipSuccess) { fail(e, "hipMalloc/hipMemcpy(nfm state)", __FILE__, __LINE__); csdr_amd_nfm_destroy(w); return nullptr; }
    if (csdr_amd_nfm_reset(w)) { csdr_amd_nfm_destroy(w); return nullptr; }
    return w;
}

void csdr_amd_nfm_destroy(csdr_amd_nfm *w)
{
    if (!w) return;
    (void)hipStreamSynchronize(w->ctx->stream);
    if (w->ddc) csdr_amd_ddc_destroy(w->ddc);
    (void)hipFree(w->d_y); (void)hipFree(w->d_last); (void)hipFree(w->d_dl); (void)hipFree(w->d_de); (void)hipFree(w->d_agc);
    (void)hipFree(w->d_dtaps); (void)hipFree(w->d_agc_state);
    delete w;
}

int csdr_amd_nfm_reset(csdr_amd_nfm *w)
{
    hipStream_t st = w->ctx->stream;
    w->dl_fill = 0;
    CSDR_HIP(hipMemsetAsync(w->d_last, 0, sizeof(cf32) * w->n_streams, st));                                   // the CLI starts fmdemod from (0, 0) (csdr.c:1044)
    CSDR_HIP(hipMemsetAsync(w->d_agc_state, 0, sizeof(float) * (size_t)w->n_streams * (2 * w->agc_block + 4), st));   // calloc'ed fastagc state (csdr.c:1393-1394)
    return csdr_amd_ddc_reset(w->ddc);
}

csdr_amd_ddc *csdr_amd_nfm_front_end(csdr_amd_nfm *w) { return w->ddc; }

long csdr_amd_nfm_process(csdr_amd_nfm *w, const uint8_t *in, size_t in_pitch, size_t block_samples, int16_t *audio_s16, float *audio_f, size_t out_pitch)
{
    csdr_amd_ctx *c = w->ctx; hipStream_t st = c->stream;
    const int S = w->n_streams;
    const long n_y = csdr_amd_ddc_process(w->ddc, in, in_pitch, block_samples, w->d_y, w->y_pitch);
    if (n_y < 0) return n_y;
    if (n_y == 0) return 0;
    if ((size_t)n_y > w->max_y) return fail_msg(-3, "nfm: front end produced more than the planned %zu samples", w->max_y);
    // fmdemod_quadri_cf | limit_ff, appended behind the filter's unconsumed input
    hipLaunchKernelGGL(k_nfm_demod_limit, dim3(cdiv(n_y, 256), S), dim3(256), 0, st, w->d_y, w->y_pitch, (int)n_y, w->d_last, w->d_dl, w->dl_pitch, w->dl_fill, w->limit);
    CSDR_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_nfm_store_last, dim3(cdiv(S, 64)), dim3(64), 0, st, w->d_y, w->y_pitch, (int)n_y, w->d_last, S);
    CSDR_LAUNCH_CHECK();
    const int n_in = w->dl_fill + (int)n_y;
    // deemphasis_nfm_ff produces input - taps outputs (libcsdr.c:1121); run it for whole AGC blocks only
    const int ne_all = n_in - w->Ld;
    const int nb = ne_all > 0 ? ne_all / w->agc_block : 0;
    const int ne = nb * w->agc_block;
    if (nb > 0) {
        if ((size_t)ne > out_pitch && S > 1) return fail_msg(-3, "nfm: out_pitch %zu smaller than the %d audio samples of this block", out_pitch, ne);
        const int got = csdr_amd_fir_ff(c, w->d_dl, w->d_de, S, ne + w->Ld, w->dl_pitch, w->a_pitch, w->d_dtaps, w->Ld);
        if (got != ne) return got < 0 ? got : fail_msg(-3, "nfm: de-emphasis filter produced %d of %d samples", got, ne);
        int rc = csdr_amd_fastagc_ff(c, w->d_de, w->d_agc, S, nb, w->agc_block, w->a_pitch, w->a_pitch, w->agc_ref, w->d_agc_state);
        if (rc) return rc;
        hipLaunchKernelGGL(k_nfm_out, dim3(cdiv(ne, 256), S), dim3(256), 0, st, w->d_agc, w->a_pitch, ne, audio_s16, audio_f, out_pitch);
        CSDR_LAUNCH_CHECK();
    }
    // keep the unconsumed filter input in front
    const int rem = n_in - ne;
    if (ne > 0 && rem > 0) {
        if (rem > 2048) return fail_msg(-3, "nfm: internal carry of %d samples", rem);
        hipLaunchKernelGGL(k_nfm_move_front, dim3(S), dim3(256), 0, st, w->d_dl, w->dl_pitch, ne, rem);
        CSDR_LAUNCH_CHECK();
    }
    w->dl_fill = rem;
    return ne;
}

} // extern "C"
