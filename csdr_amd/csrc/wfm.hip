// wfm.hip -- fused wide-FM receive chain (BASELINE.json config 2; README.md:66, csdr-fm:41):
//
//   convert_u8_f | shift_addition_cc r | fir_decimate_cc D | fmdemod_quadri_cf |
//   fractional_decimator_ff F | deemphasis_wfm_ff | convert_f_s16
//
// for N independent u8 IQ streams.  Stream model being implemented (verified against the
// compiled reference in tests/test_oracle_vs_ref.py::test_wfm_chain):
//
//   x'[n]  = u8->float(iq[n]) * rot[n]              rot = shift_addition_cc's float32 phasor, 1024-chunks
//   y[k]   = sum_t h[t] x'[D k + t]                 no zero history in front
//   d[m]   = K (I dQ - Q dI)/(I^2+Q^2)  of y[m], y[m-1]
//   a[j]   = d[F j + 10]                            integer-rate Lagrange decimator == pure gather (exact)
//   e[j]   = alpha a[j] + (1-alpha) e[j-1]          de-emphasis
//   s16[j] = trunc(e[j] * 32767)
//
// Only y[Fj+9] and y[Fj+10] feed the audio, so 2 of every F FIR outputs are computed (exact, 2.5x fewer MACs
// at F = 5).  HBM traffic: 2 B in per complex sample + 2/(D F) B out: every input byte is read once.
//
// This file: the chain OBJECT (state, bookkeeping, csdr_amd_wfm_*) and the VALU fallback.  The default path is ONE kernel per call, k_wfm_mfma_seq (wfm_mfma.hip:
// the whole chain on the matrix cores); the two kernels below serve shapes it does not cover (D F odd, long filters) and CSDR_AMD_WFM_PATH=valu:
//   k_wfm_front : (stream, tile of A audio samples) per workgroup.  u8 window -> float -> rotate -> LDS;
//                 FIR pairs from LDS (two lanes per output, 40 taps each); quadrature demod; writes the
//                 pre-de-emphasis audio float (4 B per D*F input samples) to a scratch row.
//   k_wfm_back  : de-emphasis + s16.  The one-pole IIR forgets its state as (1-alpha)^k (0.706^48 = 5.6e-8),
//                 so every 64-sample segment is started 48 samples early from zero state and is then
//                 independent of its predecessor to float precision; the first segment of a call uses the
//                 exact carried state.
#include "common.hpp"
#include "wfm_mfma.hpp"
#include "seeds.hpp"
#include <math.h>
#include <stdlib.h>
#include <vector>
#include <string>
#include <string.h>
using namespace csdr_amd;

namespace {

constexpr int HIST = WFM_HIST;     // complex samples of input history kept per stream (>= D*(F-1+..)+taps needs 88 for 10/79/5)
constexpr int TILE_A = 64;         // audio samples per workgroup
constexpr int WARM = 48;           // de-emphasis warm-up samples

struct WfmParams {
    int D, L, F;                   // decimation, taps, audio decimation
    int T;                         // complex samples in this block
    long long B;                   // global index of the block's first sample
    long long j_first;             // first audio index produced by this call
    int n_audio;                   // audio samples produced by this call
};

__device__ __forceinline__ float u8_to_f(uint32_t v) { return fmaf((float)v, 0x1.010102p-7f, -1.0f); }   // v/127.5 - 1 (<= 1 ulp)

__global__ __launch_bounds__(256) void k_wfm_front(const uint8_t *__restrict__ in, size_t in_pitch, const uint8_t *__restrict__ hist,
                                                   const cf32 *__restrict__ rot, const float *__restrict__ taps,
                                                   float *__restrict__ demod, size_t demod_pitch, WfmParams p)
{
    extern __shared__ float4 lds_raw[];
    float2 *win = reinterpret_cast<float2 *>(lds_raw);
    // grid: x = stream (fastest), y = time tile -- workgroups that run together share the same rotator window in L2
    const int s = blockIdx.x;
    const int a0 = blockIdx.y * TILE_A;
    const int na = min(TILE_A, p.n_audio - a0);
    if (na <= 0) return;
    const long long j0 = p.j_first + a0;
    const int DF = p.D * p.F;
    // window of input needed: FIR outputs F*j+9 .. F*(j0+na-1)+10
    const long long g_first = (long long)p.D * (p.F * j0 + 9);
    const long long g_last = (long long)p.D * (p.F * (j0 + na - 1) + 10) + p.L - 1;
    const long long r_first = (g_first - p.B) & ~7LL;                 // block-relative, aligned down to 8 samples (B is a multiple of 8)
    const int wl = (int)(g_last - p.B - r_first + 1);
    const int nvec = (wl + 7) / 8;
    const uint8_t *row = in + (size_t)s * in_pitch;
    const uint8_t *hrow = hist + (size_t)s * (2 * HIST);
    float2 *ybuf = win + nvec * 8;                                    // 2*TILE_A FIR outputs
    for (int v = threadIdx.x; v < nvec; v += 256) {
        const long long r0 = r_first + 8LL * v;
        uint4 w = make_uint4(0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u);
        if (r0 < 0) w = *reinterpret_cast<const uint4 *>(hrow + 2 * (r0 + HIST));
        else if (r0 + 8 <= p.T) w = *reinterpret_cast<const uint4 *>(row + 2 * r0);
        else { // ragged end of the final block: byte-wise
            uint32_t b[4] = {0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u};
            for (int k = 0; k < 16; k++) if (r0 + k / 2 < p.T) { const uint32_t by = row[2 * r0 + k]; b[k / 4] = (b[k / 4] & ~(0xffu << (8 * (k & 3)))) | (by << (8 * (k & 3))); }
            w = make_uint4(b[0], b[1], b[2], b[3]);
        }
        const cf32 *rr = rot + (r0 + HIST);
        const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float i0 = u8_to_f(ww[k] & 0xff), q0 = u8_to_f((ww[k] >> 8) & 0xff);
            const float i1 = u8_to_f((ww[k] >> 16) & 0xff), q1 = u8_to_f(ww[k] >> 24);
            const float4 r2 = *reinterpret_cast<const float4 *>(rr + 2 * k);
            float4 o;
            o.x = r2.x * i0 - r2.y * q0; o.y = r2.y * i0 + r2.x * q0;
            o.z = r2.z * i1 - r2.w * q1; o.w = r2.w * i1 + r2.z * q1;
            *reinterpret_cast<float4 *>(win + 8 * v + 2 * k) = o;
        }
    }
    __syncthreads();
    // FIR: lane pair (2p, 2p+1) computes output p = 2*a + which  (which = 0: y[Fj+9], 1: y[Fj+10])
    {
        const int pidx = threadIdx.x >> 1, half = threadIdx.x & 1;
        const int a = pidx >> 1, which = pidx & 1;
        float ai = 0.f, aq = 0.f;
        if (a < na) {
            const long long g = (long long)p.D * (p.F * (j0 + a) + 9 + which);
            const float2 *x = win + (int)(g - p.B - r_first);
            const int split = (p.L + 1) / 2;
            const int t0 = half ? split : 0, t1 = half ? p.L : split;
            for (int t = t0; t < t1; t++) { const float h = taps[t]; const float2 v = x[t]; ai = fmaf(v.x, h, ai); aq = fmaf(v.y, h, aq); }
        }
        ai += __shfl_xor(ai, 1, 64); aq += __shfl_xor(aq, 1, 64);
        if (half == 0) ybuf[pidx] = make_float2(ai, aq);
    }
    __syncthreads();
    if (threadIdx.x < na) {
        const float2 pv = ybuf[2 * threadIdx.x], cv = ybuf[2 * threadIdx.x + 1];
        const float dq = cv.y - pv.y, di = cv.x - pv.x;
        const float num = cv.x * dq - cv.y * di, den = cv.x * cv.x + cv.y * cv.y;
        const float K = 0.340447550238101026565118445432744920253753662109375f;
        demod[(size_t)s * demod_pitch + a0 + threadIdx.x] = (den != 0.f) ? (K * num) / den : 0.f;
        (void)DF;
    }
}

// de-emphasis + convert_f_s16.  Block = 256 lanes x 16-sample segments = 4096 audio samples of one stream; every lane starts
// WARM (48) samples early from zero state (0.706^48 = 5.6e-8: the predecessor's state is forgotten to float precision), the very
// first segment of a call starts from the exact carried state.  Input staged once through LDS (lane stride 17 floats: conflict free),
// each lane leaves 16 s16 = 32 contiguous bytes -> fully coalesced 8 KiB per block.
constexpr int BK_SEG = 16, BK_CHUNK = 256 * BK_SEG;
__global__ __launch_bounds__(256) void k_wfm_back(const float *__restrict__ demod_base, size_t demod_pitch, int skip, int n_audio, float alpha,
                                                  const float *__restrict__ last_in, float *__restrict__ last_out,
                                                  int16_t *__restrict__ s16, float *__restrict__ audio_f, size_t out_pitch)
{
    __shared__ float seg[(BK_CHUNK + WARM) + (BK_CHUNK + WARM) / 16 + 16];
    const int s = blockIdx.y;
    const int t0 = blockIdx.x * BK_CHUNK;
    const int cnt = min(BK_CHUNK, n_audio - t0);
    if (cnt <= 0) return;
    const float *row = demod_base + (size_t)s * demod_pitch + skip;   // the matrix-core front end stores whole 4-sample tiles
    const int lead = (t0 >= WARM) ? WARM : t0;                        // samples before t0 available for warm-up
    for (int q = threadIdx.x; q < cnt + lead; q += 256) seg[q + (q >> 4)] = row[t0 - lead + q];
    __syncthreads();
    const int my0 = threadIdx.x * BK_SEG;                             // first sample of my segment, relative to t0
    if (my0 >= cnt) return;
    const float one_minus = 1 - alpha;
    // warm-up window: up to WARM samples before my segment; when it reaches the first sample of this call the exact carried
    // state is the starting point (NaN state reset as libcsdr.c:1092), otherwise zero
    const int back = (my0 + lead >= WARM) ? WARM : (my0 + lead);
    int q = lead + my0 - back;
    float y = 0.f;
    if (t0 + my0 - back == 0) { y = last_in[s]; if (y != y) y = 0.f; }
    for (int k = 0; k < back; k++, q++) y = alpha * seg[q + (q >> 4)] + one_minus * y;
    const int mine = min(BK_SEG, cnt - my0);
    float e[BK_SEG];
#pragma unroll
    for (int k = 0; k < BK_SEG; k++) {
        if (k < mine) { y = alpha * seg[q + (q >> 4)] + one_minus * y; q++; }
        e[k] = y;
    }
    if (t0 + my0 + mine == n_audio) last_out[s] = y;
    uint32_t pk[BK_SEG / 2];
#pragma unroll
    for (int k = 0; k < BK_SEG; k += 2) {
        int iv[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const float scaled = e[k + h] * 32767.0f;                 // convert_f_s16 libcsdr.c:2397 (x86 truncation semantics)
            iv[h] = (scaled >= -2147483648.0f && scaled < 2147483648.0f) ? (int)scaled : (int)0x80000000;
        }
        pk[k / 2] = (uint32_t)(iv[0] & 0xffff) | ((uint32_t)iv[1] << 16);
    }
    int16_t *dst = s16 + (size_t)s * out_pitch + t0 + my0;
    if (mine == BK_SEG && ((((uintptr_t)dst) & 15) == 0)) {
        reinterpret_cast<uint4 *>(dst)[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        reinterpret_cast<uint4 *>(dst)[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
    } else {
#pragma unroll
        for (int k = 0; k < BK_SEG; k++) if (k < mine) dst[k] = (int16_t)((k & 1) ? (pk[k / 2] >> 16) : (pk[k / 2] & 0xffff));
    }
    if (audio_f) {
        float *df = audio_f + (size_t)s * out_pitch + t0 + my0;
#pragma unroll
        for (int k = 0; k < BK_SEG; k++) if (k < mine) df[k] = e[k];
    }
}

__global__ __launch_bounds__(256) void k_wfm_save_hist(const uint8_t *__restrict__ in, size_t in_pitch, int T, uint8_t *__restrict__ hist)
{   // keep the last HIST complex samples of every stream for the next block's FIR windows
    const int s = blockIdx.x;
    const uint8_t *src = in + (size_t)s * in_pitch + 2 * (size_t)(T - HIST);
    uint8_t *dst = hist + (size_t)s * (2 * HIST);
    for (int k = threadIdx.x; k < 2 * HIST / 4; k += 256) reinterpret_cast<uint32_t *>(dst)[k] = reinterpret_cast<const uint32_t *>(src)[k];
}

} // namespace

struct csdr_amd_wfm {
    csdr_amd_ctx *ctx;
    int n_streams, D, L, F, audio_rate;
    float shift_rate, tau, alpha;
    size_t max_block;
    float *d_taps, *d_demod, *d_last[2];
    float phase;                     // shift_addition_cc starting_phase (host float, like the reference's by-value state)
    cf32 *d_rot;
    uint8_t *d_hist;                 // VALU front end: 2 * HIST bytes per stream
    uint8_t *d_head[2]; int hflip;   // matrix-core chain kernel: 1 KiB per stream, second half = the previous block's newest 512 bytes (two buffers: wfm_mfma.hip)
    size_t demod_pitch;
    long long B, next_j;
    int last_T, flip;
    bool ended;
    std::string kernel_name;
    // optional HIP-event timing of the dominant kernel (bench.py roofline leg)
    bool profiling;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    size_t ev_used;
    double prof_ms; long prof_launches;
    // matrix-core front end (wfm_mfma.hip)
    bool use_mfma;
    WfmMfmaDevice mfma;
    float2 *d_ctab; size_t ctab_cap;
    float2 c_prev;                   // phasor seed of the previous block's last chunk (history windows)
    long long tab_first; bool tab_valid;      // the device table of seeds covers chunks [tab_first, tab_first + ctab_cap)
    // a shift rate per stream (csdr_amd_wfm_create_rates): one table set per stream in mfma.*, chunk seeds from a seed table (seeds.hip)
    bool ps; std::vector<float> rates, h_taps; csdr_amd::SeedTables *seeds; float *d_scales;
    float2 *d_dtab_old; int *d_list, *d_lead_n; float *d_lead_d; std::vector<int> retuned;      // streams retuned since the last call: their first samples straddle two rates
};

extern "C" {

// tables of ONE stream of a per-stream object
static int wfm_upload_stream_tables(csdr_amd_wfm *w, int s, float rate)
{
    WfmMfmaTable t;
    wfm_mfma_build_table(w->D, w->L, w->F, rate, w->h_taps.data(), t);
    CSDR_HIP(hipMemcpy((uint8_t *)w->mfma.d_seq_frags + (size_t)s * t.seq_frags.size(), t.seq_frags.data(), t.seq_frags.size(), hipMemcpyHostToDevice));
    CSDR_HIP(hipMemcpy(w->mfma.d_seq_cum + (size_t)s * t.seq_cum.size(), t.seq_cum.data(), t.seq_cum.size() * sizeof(float), hipMemcpyHostToDevice));
    CSDR_HIP(hipMemcpy(w->mfma.d_dtab + (size_t)s * t.dtab.size(), t.dtab.data(), t.dtab.size() * sizeof(float2), hipMemcpyHostToDevice));
    CSDR_HIP(hipMemcpy(w->d_scales + s, &t.seq_scale, sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

static csdr_amd_wfm *wfm_create_impl(csdr_amd_ctx *ctx, int n_streams, const float *rates, bool per_stream, int decimation, const float *host_taps,
                                     int taps_length, int frac_rate, float tau, int audio_rate, size_t max_block_samples)
{
    const float shift_rate = rates ? rates[0] : 0.f;
    if (n_streams <= 0 || decimation <= 0 || taps_length <= 0 || frac_rate <= 1 || !rates) { fail_msg(-3, "wfm: bad parameters"); return nullptr; }
    if (decimation * 1 + taps_length + 8 > HIST) { fail_msg(-3, "wfm: D + taps (%d + %d) exceed the %d-sample history", decimation, taps_length, HIST); return nullptr; }
    if (max_block_samples < 1024) max_block_samples = 1024;
    csdr_amd_wfm *w = new csdr_amd_wfm();
    w->ctx = ctx; w->n_streams = n_streams; w->D = decimation; w->L = taps_length; w->F = frac_rate; w->audio_rate = audio_rate;
    w->shift_rate = shift_rate; w->tau = tau; w->max_block = max_block_samples;
    const float dt = (float)(1.0 / audio_rate); w->alpha = dt / (tau + dt);          // libcsdr.c:1090-1091
    const size_t max_audio = max_block_samples / ((size_t)decimation * frac_rate) + 8;
    w->demod_pitch = (max_audio + 40 + 63) & ~(size_t)63;          // (VALU path's scratch rows; tile padding)
    hipError_t e = hipSuccess;
    auto alloc = [&](void **p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes); };
    w->d_demod = nullptr; w->d_rot = nullptr; w->d_hist = nullptr; w->d_head[0] = w->d_head[1] = nullptr; w->hflip = 0;
    w->use_mfma = false; w->d_ctab = nullptr; w->ctab_cap = 0; w->mfma.d_seq_frags = nullptr; w->mfma.d_seq_cum = nullptr; w->mfma.d_dtab = nullptr;
    w->ps = per_stream; w->seeds = nullptr; w->d_scales = nullptr; w->d_dtab_old = nullptr; w->d_list = nullptr; w->d_lead_n = nullptr; w->d_lead_d = nullptr;
    {
        const char *force = getenv("CSDR_AMD_WFM_PATH");          // "valu" forces the VALU/LDS front end (A/B comparisons); read once, here
        w->use_mfma = !(force && !strcmp(force, "valu")) && wfm_mfma_supported(decimation, taps_length, frac_rate);
    }
    if (per_stream && !w->use_mfma) { fail_msg(-3, "wfm_create_rates: a rate per stream needs the matrix-core chain kernel (decimation x audio decimation even, window <= 256 samples)"); delete w; return nullptr; }
    alloc((void **)&w->d_taps, sizeof(float) * taps_length);
    alloc((void **)&w->d_last[0], sizeof(float) * n_streams);
    alloc((void **)&w->d_last[1], sizeof(float) * n_streams);
    if (w->use_mfma) {
        alloc((void **)&w->d_head[0], wfm_mfma_head_bytes(n_streams));
        alloc((void **)&w->d_head[1], wfm_mfma_head_bytes(n_streams));
    } else {                                                       // the VALU front end's intermediates: demodulated floats, rotator table, history
        alloc((void **)&w->d_demod, sizeof(float) * w->demod_pitch * n_streams);
        alloc((void **)&w->d_rot, sizeof(cf32) * (HIST + max_block_samples + 64));
        alloc((void **)&w->d_hist, (size_t)2 * HIST * n_streams);
    }
    if (e != hipSuccess) { fail(e, "hipMalloc(wfm state)", __FILE__, __LINE__); csdr_amd_wfm_destroy(w); return nullptr; }
    (void)hipMemcpy(w->d_taps, host_taps, sizeof(float) * taps_length, hipMemcpyHostToDevice);
    w->kernel_name = "k_wfm_front";
    if (w->use_mfma && per_stream) {
        WfmMfmaTable t;
        wfm_mfma_build_table(decimation, taps_length, frac_rate, shift_rate, host_taps, t);      // (sizes and geometry; the tables themselves per stream below)
        w->mfma.tile_stride_bytes = t.tile_stride_bytes; w->mfma.win_off_bytes = t.win_off_bytes; w->mfma.seq_scale = t.seq_scale;
        w->rates.assign(rates, rates + n_streams); w->h_taps.assign(host_taps, host_taps + taps_length);
        hipError_t e2 = hipMalloc(&w->mfma.d_seq_frags, t.seq_frags.size() * (size_t)n_streams);
        if (e2 == hipSuccess) e2 = hipMalloc((void **)&w->mfma.d_seq_cum, t.seq_cum.size() * sizeof(float) * (size_t)n_streams);
        if (e2 == hipSuccess) e2 = hipMalloc((void **)&w->mfma.d_dtab, t.dtab.size() * sizeof(float2) * (size_t)n_streams);
        if (e2 == hipSuccess) e2 = hipMalloc((void **)&w->d_scales, sizeof(float) * (size_t)n_streams);
        if (e2 != hipSuccess) { fail(e2, "hipMalloc(wfm per-stream tables)", __FILE__, __LINE__); csdr_amd_wfm_destroy(w); return nullptr; }
        for (int s = 0; s < n_streams; s++) if (wfm_upload_stream_tables(w, s, rates[s])) { csdr_amd_wfm_destroy(w); return nullptr; }
        w->seeds = seeds_create(ctx, n_streams, rates, nullptr, 0, max_block_samples);
        if (!w->seeds) { csdr_amd_wfm_destroy(w); return nullptr; }
        w->kernel_name = "k_wfm_mfma_seq";
    } else if (w->use_mfma) {
        WfmMfmaTable t;
        wfm_mfma_build_table(decimation, taps_length, frac_rate, shift_rate, host_taps, t);
        w->mfma.tile_stride_bytes = t.tile_stride_bytes; w->mfma.win_off_bytes = t.win_off_bytes; w->mfma.seq_scale = t.seq_scale;
        w->ctab_cap = 16 * (max_block_samples / 1024 + 8);                 // 16 calls of the largest block ahead (2.4 M-sample blocks: 300 KB)
        hipError_t e2 = hipMalloc((void **)&w->d_ctab, w->ctab_cap * sizeof(float2));
        if (e2 == hipSuccess) e2 = hipMalloc(&w->mfma.d_seq_frags, t.seq_frags.size());
        if (e2 == hipSuccess) e2 = hipMalloc((void **)&w->mfma.d_seq_cum, t.seq_cum.size() * sizeof(float));
        if (e2 == hipSuccess) e2 = hipMalloc((void **)&w->mfma.d_dtab, t.dtab.size() * sizeof(float2));
        if (e2 == hipSuccess) e2 = hipMemcpy(w->mfma.d_seq_frags, t.seq_frags.data(), t.seq_frags.size(), hipMemcpyHostToDevice);
        if (e2 == hipSuccess) e2 = hipMemcpy(w->mfma.d_seq_cum, t.seq_cum.data(), t.seq_cum.size() * sizeof(float), hipMemcpyHostToDevice);
        if (e2 == hipSuccess) e2 = hipMemcpy(w->mfma.d_dtab, t.dtab.data(), t.dtab.size() * sizeof(float2), hipMemcpyHostToDevice);
        if (e2 != hipSuccess) { fail(e2, "hipMalloc/hipMemcpy(wfm mfma table)", __FILE__, __LINE__); csdr_amd_wfm_destroy(w); return nullptr; }
        w->kernel_name = "k_wfm_mfma_seq";
    }
    w->profiling = false; w->ev_used = 0; w->prof_ms = 0; w->prof_launches = 0;
    if (csdr_amd_wfm_reset(w)) { csdr_amd_wfm_destroy(w); return nullptr; }
    return w;
}

csdr_amd_wfm *csdr_amd_wfm_create(csdr_amd_ctx *ctx, int n_streams, float shift_rate, int decimation, const float *host_taps,
                                  int taps_length, int frac_rate, float tau, int audio_rate, size_t max_block_samples)
{
    return wfm_create_impl(ctx, n_streams, &shift_rate, false, decimation, host_taps, taps_length, frac_rate, tau, audio_rate, max_block_samples);
}

csdr_amd_wfm *csdr_amd_wfm_create_rates(csdr_amd_ctx *ctx, int n_streams, const float *shift_rates, int decimation, const float *host_taps,
                                        int taps_length, int frac_rate, float tau, int audio_rate, size_t max_block_samples)
{
    return wfm_create_impl(ctx, n_streams, shift_rates, true, decimation, host_taps, taps_length, frac_rate, tau, audio_rate, max_block_samples);
}

// Retune of one stream between two calls = `csdr shift_addition_cc --fifo` (csdr.c:881-923): the new rate from the next block's first sample, the phase carried.
int csdr_amd_wfm_set_rate(csdr_amd_wfm *w, int stream, float shift_rate)
{
    if (!w->ps) return fail_msg(-3, "wfm_set_rate: the object shares one rate (create it with csdr_amd_wfm_create_rates)");
    if (stream < 0 || stream >= w->n_streams) return fail_msg(-3, "wfm_set_rate: stream %d out of range", stream);
    if (w->rates[stream] == shift_rate) return 0;
    // The audio samples whose windows reach back across the retune instant are recomputed with both tables (k_wfm_lead): a window spans D + L - 1 samples and audio
    // samples lie D F apart, so d_lead_d holds wfm_lead_max(D, L, F) per stream (10 / 79 / 5: 3; ADVICE r5: sized from the shape, no shape is refused).
    CSDR_HIP(hipStreamSynchronize(w->ctx->stream));                   // calls in flight read this stream's tables
    if (!w->d_dtab_old) CSDR_HIP(hipMalloc((void **)&w->d_dtab_old, sizeof(float2) * 3072 * (size_t)w->n_streams));
    bool listed = false;
    for (int v : w->retuned) listed |= v == stream;
    if (!listed) {
        CSDR_HIP(hipMemcpy(w->d_dtab_old + (size_t)stream * 3072, w->mfma.d_dtab + (size_t)stream * 3072, sizeof(float2) * 3072, hipMemcpyDeviceToDevice));
        w->retuned.push_back(stream);
    }
    const int rc = wfm_upload_stream_tables(w, stream, shift_rate); if (rc) return rc;
    w->rates[stream] = shift_rate;
    return seeds_set_rate(w->seeds, stream, shift_rate, false);
}

float csdr_amd_wfm_get_rate(const csdr_amd_wfm *w, int stream) { return w->ps ? ((stream >= 0 && stream < w->n_streams) ? w->rates[stream] : 0.f) : w->shift_rate; }

void csdr_amd_wfm_destroy(csdr_amd_wfm *w)
{
    if (!w) return;
    (void)hipStreamSynchronize(w->ctx->stream);
    if (w->seeds) seeds_destroy(w->seeds);
    (void)hipFree(w->d_scales); (void)hipFree(w->d_dtab_old); (void)hipFree(w->d_lead_n); (void)hipFree(w->d_lead_d);      // (d_list lives inside d_lead_n)
    (void)hipFree(w->d_taps); (void)hipFree(w->d_demod); (void)hipFree(w->d_last[0]); (void)hipFree(w->d_last[1]);
    (void)hipFree(w->d_rot); (void)hipFree(w->d_hist); (void)hipFree(w->d_head[0]); (void)hipFree(w->d_head[1]);
    for (auto &pr : w->ev_pool) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    (void)hipFree(w->mfma.d_seq_frags); (void)hipFree(w->mfma.d_seq_cum); (void)hipFree(w->mfma.d_dtab); (void)hipFree(w->d_ctab);
    delete w;
}

int csdr_amd_wfm_reset(csdr_amd_wfm *w)
{
    hipStream_t st = w->ctx->stream;
    w->phase = 0.f;
    CSDR_HIP(hipMemsetAsync(w->d_last[0], 0, sizeof(float) * w->n_streams, st));
    CSDR_HIP(hipMemsetAsync(w->d_last[1], 0, sizeof(float) * w->n_streams, st));
    if (w->d_rot) CSDR_HIP(hipMemsetAsync(w->d_rot, 0, sizeof(cf32) * (HIST + w->max_block + 64), st));
    if (w->d_hist) CSDR_HIP(hipMemsetAsync(w->d_hist, 0x80, (size_t)2 * HIST * w->n_streams, st));
    for (int k = 0; k < 2; k++) if (w->d_head[k]) CSDR_HIP(hipMemsetAsync(w->d_head[k], 0x80, wfm_mfma_head_bytes(w->n_streams), st));      // 0x80 = zero samples in front of the stream
    w->hflip = 0;
    w->B = 0; w->next_j = 0; w->last_T = 0; w->flip = 0; w->ended = false;
    w->c_prev = make_float2(1.f, 0.f); w->tab_valid = false; w->tab_first = 0;
    w->retuned.clear();
    if (w->seeds) { const int rc = seeds_reset(w->seeds); if (rc) return rc; }
    return 0;
}

const char *csdr_amd_wfm_kernel_name(const csdr_amd_wfm *w) { return w->kernel_name.c_str(); }
int csdr_amd_wfm_fallback(const csdr_amd_wfm *w) { return w->use_mfma ? 0 : 1; }

int csdr_amd_wfm_set_profiling(csdr_amd_wfm *w, int on)
{
    w->profiling = on != 0; w->ev_used = 0; w->prof_ms = 0; w->prof_launches = 0;
    return 0;
}

int csdr_amd_wfm_kernel_time(csdr_amd_wfm *w, double *total_ms, long *launches)
{   // resolves the recorded event pairs (synchronises the stream)
    CSDR_HIP(hipStreamSynchronize(w->ctx->stream));
    for (size_t k = 0; k < w->ev_used; k++) {
        float ms = 0; CSDR_HIP(hipEventElapsedTime(&ms, w->ev_pool[k].first, w->ev_pool[k].second));
        w->prof_ms += ms; w->prof_launches++;
    }
    w->ev_used = 0;
    if (total_ms) *total_ms = w->prof_ms;
    if (launches) *launches = w->prof_launches;
    return 0;
}

long csdr_amd_wfm_process(csdr_amd_wfm *w, const uint8_t *in, size_t in_pitch, size_t block_samples,
                          int16_t *audio_s16, float *audio_f, size_t out_pitch)
{
    csdr_amd_ctx *c = w->ctx; hipStream_t st = c->stream;
    if (w->ended) return fail_msg(-3, "wfm: stream already ended by a block that was not a multiple of 1024 samples; reset first");
    if (block_samples == 0) return 0;
    if (block_samples > w->max_block) return fail_msg(-3, "wfm: block of %zu samples exceeds max_block_samples %zu", block_samples, w->max_block);
    if (((uintptr_t)in & 15) || (in_pitch & 15)) return fail_msg(-3, "wfm: input pointer and pitch must be 16-byte aligned");
    const int T = (int)block_samples;
    int rc = 0;
    const float2 *ctab_call = nullptr;
    SeedView sv; memset(&sv, 0, sizeof sv);
    if (w->ps) {
        const size_t nch = ((size_t)T + 1023) / 1024;
        rc = seeds_acquire(w->seeds, w->B / 1024 - 1, nch + 3, (T % 1024) ? 0 : nch, &sv); if (rc) return rc;
        ctab_call = sv.ctab;
    } else
    if (w->use_mfma) {
        // 1a. per-chunk phasor seeds C_m = (cos, sin)(starting_phase_m) with the reference's float phase bookkeeping (libcsdr_gpl.c:33-34, 48-51; chunks
        //     of 1024 per csdr.c:911-918).  The sequence depends on nothing but the shift rate, so the device holds a TABLE of it that runs far ahead of the
        //     stream (16 calls' worth) and a call only passes an offset: no upload sits between two calls' kernels (round 2 uploaded every call's seeds in
        //     front of its kernel: a copy-engine hand-off of ~10 us per step).  Entry k of the table = chunk tab_first + k; a call needs the chunk in front
        //     of its block (history windows) up to two behind it.
        const size_t nch = ((size_t)T + 1023) / 1024;
        const long long first = w->B / 1024 - 1;                              // (blocks are multiples of 1024 except a stream's last one)
        const float inc = (w->shift_rate * 2) * PI_F;
        if (nch + 3 > w->ctab_cap) return fail_msg(-3, "wfm: chunk table too small");
        if (!w->tab_valid || first < w->tab_first || first + (long long)nch + 3 > w->tab_first + (long long)w->ctab_cap) {
            float2 *hc = (float2 *)c->pinned_acquire(sizeof(float2) * w->ctab_cap);
            if (!hc) return -2;
            float ph = w->phase;
            hc[0] = w->c_prev;
            for (size_t k = 1; k < w->ctab_cap; k++) {
                hc[k] = make_float2((float)cos((double)ph), (float)sin((double)ph));
                float nx = ph + inc * (float)1024;
                while (nx > PI_F) nx -= 2 * PI_F;
                while (nx < -PI_F) nx += 2 * PI_F;
                ph = nx;
            }
            rc = c->pinned_upload(w->d_ctab, sizeof(float2) * w->ctab_cap); if (rc) return rc;
            w->tab_first = first; w->tab_valid = true;
        }
        ctab_call = w->d_ctab + (first - w->tab_first);
        // the stream's phase behind this block (and the seed of its last chunk, for a table rebuilt at the next call)
        float ph = w->phase;
        for (size_t m = 0; m < nch; m++) {
            if (m + 1 == nch) w->c_prev = make_float2((float)cos((double)ph), (float)sin((double)ph));
            const int len = ((size_t)T - m * 1024 < 1024) ? (int)((size_t)T - m * 1024) : 1024;
            float nx = ph + inc * (float)len;
            while (nx > PI_F) nx -= 2 * PI_F;
            while (nx < -PI_F) nx += 2 * PI_F;
            ph = nx;
        }
        w->phase = ph;
    } else {
        // 1b. rotator table for this block behind the previous block's tail (history positions keep their own phasors)
        if (w->last_T) CSDR_HIP(hipMemcpyAsync(w->d_rot, w->d_rot + w->last_T, sizeof(cf32) * HIST, hipMemcpyDeviceToDevice, st));
        rc = csdr_amd_rotator_generate(c, CSDR_SHIFT_ADDITION, w->shift_rate, &w->phase, w->d_rot + HIST, (size_t)T, 1024, 0);
        if (rc) return rc;
    }
    // 2. audio samples that become computable with this block: F*j+10 is the newest FIR output, needs input up to D*(F*j+10)+L-1
    const long long avail_last = w->B + T - 1;
    long long j_hi = -1;
    if (avail_last - (w->L - 1) >= 0) {
        const long long k_max = (avail_last - (w->L - 1)) / w->D;          // newest complete FIR output index
        if (k_max >= 10) j_hi = (k_max - 10) / w->F;
    }
    const long long n_audio_ll = j_hi - w->next_j + 1;
    const int n_audio = n_audio_ll > 0 ? (int)n_audio_ll : 0;
    if (n_audio > 0 && (size_t)n_audio > out_pitch) return fail_msg(-3, "wfm: out_pitch %zu smaller than the %d audio samples of this block", out_pitch, n_audio);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (w->profiling && n_audio > 0) {
        if (w->ev_used == w->ev_pool.size()) {
            hipEvent_t a, b; CSDR_HIP(hipEventCreate(&a)); CSDR_HIP(hipEventCreate(&b));
            w->ev_pool.emplace_back(a, b);
        }
        e0 = w->ev_pool[w->ev_used].first; e1 = w->ev_pool[w->ev_used].second; w->ev_used++;
    }
    if (w->use_mfma) {
        // the whole chain, the carried de-emphasis state and the next call's history in ONE launch (wfm_mfma.hip)
        WfmBackArgs back;
        back.alpha = w->alpha; back.last_in = w->d_last[w->flip]; back.last_out = w->d_last[w->flip ^ 1];
        back.s16 = audio_s16; back.af = audio_f; back.out_pitch = out_pitch; back.head_in = w->d_head[w->hflip]; back.head_out = w->d_head[w->hflip ^ 1];
        WfmPerStream psa; memset(&psa, 0, sizeof psa);
        if (w->ps) {
            psa.tab_pitch = sv.pitch; psa.tab_len = sv.n_entries; psa.d_scales = w->d_scales;
            // retuned streams: the audio samples whose windows start in front of the block (D (F j + 9) < B) straddle two rates
            long n_lead = 0; const int lead_max = wfm_lead_max(w->D, w->L, w->F);
            psa.lead_stride = lead_max;
            if (!w->retuned.empty() && n_audio > 0) {
                const long long lim = w->B - 1 - 9LL * w->D;
                if (lim >= 0) n_lead = (long)(lim / ((long long)w->D * w->F) - w->next_j + 1);
                if (n_lead < 0) n_lead = 0;
                if (n_lead > lead_max) n_lead = lead_max;
                if (n_lead > n_audio) n_lead = n_audio;
            }
            if (n_lead > 0) {
                const int nr = (int)w->retuned.size(), S = w->n_streams;
                if (!w->d_lead_n) { CSDR_HIP(hipMalloc((void **)&w->d_lead_n, sizeof(int) * 2 * S)); w->d_list = w->d_lead_n + S; CSDR_HIP(hipMalloc((void **)&w->d_lead_d, sizeof(float) * (size_t)lead_max * S)); }
                int *hl = (int *)c->pinned_acquire(sizeof(int) * (size_t)(S + nr)); if (!hl) return -2;      // [S] samples per stream, then the list: one upload
                for (int s = 0; s < S; s++) hl[s] = 0;
                for (int k = 0; k < nr; k++) { hl[w->retuned[k]] = (int)n_lead; hl[S + k] = w->retuned[k]; }
                rc = c->pinned_upload(w->d_lead_n, sizeof(int) * (size_t)(S + nr)); if (rc) return rc;
                rc = wfm_mfma_lead(st, in, in_pitch, back.head_in, w->d_taps, sv.ctab, sv.pitch, w->mfma.d_dtab, w->d_dtab_old, w->d_list, nr, w->d_lead_d, lead_max,
                                   w->D, w->L, w->F, w->B, w->next_j, (int)n_lead);
                if (rc) return rc;
                psa.d_lead_d = w->d_lead_d; psa.d_lead_n = w->d_lead_n;
            }
        }
        rc = wfm_mfma_launch(st, e0, e1, in, in_pitch, w->mfma, ctab_call, w->n_streams, T, w->B, w->next_j, n_audio, back, w->ps ? &psa : nullptr);
        if (rc) return rc;
        if (n_audio > 0) w->retuned.clear();
        w->hflip ^= 1;
        if (n_audio > 0) w->flip ^= 1;
    } else {
        if ((size_t)n_audio + 40 > w->demod_pitch) return fail_msg(-3, "wfm: internal audio buffer too small");
        if (n_audio > 0) {
            WfmParams p; p.D = w->D; p.L = w->L; p.F = w->F; p.T = T; p.B = w->B; p.j_first = w->next_j; p.n_audio = n_audio;
            const int span = w->D * (w->F * (TILE_A - 1) + 1) + w->L + 16;          // samples per tile window (+alignment slack)
            const size_t lds = (size_t)((span + 7) / 8 * 8 + 2 * TILE_A) * sizeof(float2) + 64;
            if (lds > 64 * 1024) { const int arc = lds_attr_once((const void *)k_wfm_front, lds); if (arc) return arc; }
            if (e0) CSDR_HIP(hipEventRecord(e0, st));
            hipLaunchKernelGGL(k_wfm_front, dim3(w->n_streams, cdiv(n_audio, TILE_A)), dim3(256), lds, st,
                               in, in_pitch, w->d_hist, w->d_rot, w->d_taps, w->d_demod, w->demod_pitch, p);
            CSDR_LAUNCH_CHECK();
            if (e1) CSDR_HIP(hipEventRecord(e1, st));
            hipLaunchKernelGGL(k_wfm_back, dim3(cdiv(n_audio, BK_CHUNK), w->n_streams), dim3(256), 0, st,
                               w->d_demod, w->demod_pitch, 0, n_audio, w->alpha, w->d_last[w->flip], w->d_last[w->flip ^ 1], audio_s16, audio_f, out_pitch);
            CSDR_LAUNCH_CHECK();
            w->flip ^= 1;
        }
        // 3. history for the next block
        if (T >= HIST) {
            hipLaunchKernelGGL(k_wfm_save_hist, dim3(w->n_streams), dim3(256), 0, st, in, in_pitch, T, w->d_hist);
            CSDR_LAUNCH_CHECK();
        }
    }
    if (T % 1024) w->ended = true;
    w->B += T; w->next_j += n_audio; w->last_T = T;
    return n_audio;
}

} // extern "C"
