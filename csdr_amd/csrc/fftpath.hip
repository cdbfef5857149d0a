// fftpath.hip -- the FFT-domain filters of the hot path, FFTs by hipFFT/rocFFT (batched), everything
// around them (framing, bin products, alias folding, overlap stitching, residual shift) as HIP kernels:
//
//   bandpass_fir_fft_cc  csdr.c:1810-1886 + apply_fir_fft_cc libcsdr.c:814-849   (overlap-add)
//   fastddc_fwd_cc       csdr.c:2255-2300                                        (overlap-save framing + big FFT)
//   fastddc_inv_cc       csdr.c:2302-2378 + fastddc.c:106-166                    (fold x taps, small IFFT, scrap, shift)
//
// Blocks of one call are transformed together (one batched plan) and stitched on the device; the
// block-to-block dependency of the reference (previous block's tail) becomes a gather from the neighbouring
// block of the batch plus a small carry buffer between calls.
#include "common.hpp"
#include "fastddc.hpp"
#include <hipfft/hipfft.h>

struct csdr_amd_comm;
namespace csdr_amd { const DdcComm *csdr_amd_comm_ddc(csdr_amd_comm *c); }   // comm.cpp
#include <math.h>
#include <vector>
#include <map>
#include <tuple>
#include <mutex>
#include <string.h>
using namespace csdr_amd;

#define CSDR_FFT(expr) do { hipfftResult r__ = (expr); if (r__ != HIPFFT_SUCCESS) return ::csdr_amd::fail_msg(-5, "hipFFT error %d at %s:%d: %s", (int)r__, __FILE__, __LINE__, #expr); } while (0)

namespace {

// ------------------------------------------------------------------ overlap-add filter kernels
__global__ __launch_bounds__(256) void k_oa_frame(const cf32 *__restrict__ in, size_t in_pitch, cf32 *__restrict__ padded,
                                                  int fft, int inp, int n_blocks)
{   // grid (fft/256.., n_blocks, n_streams): zero padded copy of each input block (csdr.c:1864,1873)
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= fft) return;
    const size_t s = blockIdx.z, b = blockIdx.y;
    cf32 v = cf32{0.f, 0.f};
    if (k < inp) v = in[s * in_pitch + b * (size_t)inp + k];
    padded[(s * n_blocks + b) * (size_t)fft + k] = v;
}
__global__ __launch_bounds__(256) void k_bin_product(cf32 *__restrict__ spec, const cf32 *__restrict__ taps_fft, int fft, size_t total)
{   // libcsdr.c:826-830
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const cf32 a = spec[idx], h = taps_fft[idx % fft];
    spec[idx] = cf32{a.i * h.i - a.q * h.q, a.i * h.q + a.q * h.i};
}
__global__ __launch_bounds__(256) void k_oa_stitch(const cf32 *__restrict__ td, const cf32 *__restrict__ carry, cf32 *__restrict__ out, size_t out_pitch,
                                                   int fft, int inp, int ovl, int n_blocks, float inv_n)
{   // out position P = b*inp + i receives every block's inverse-FFT sample that lands on it plus the carried tail
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= inp) return;
    const size_t s = blockIdx.z; const int b = blockIdx.y;
    const cf32 *base = td + s * n_blocks * (size_t)fft;
    float ai = 0.f, aq = 0.f;
    // oldest contribution first (the reference accumulates tails forward in time, libcsdr.c:843-847)
    int mmax = (fft - 1 - i) / inp; if (mmax > b) mmax = b;
    const size_t P = (size_t)b * inp + i;
    if (P < (size_t)ovl) { const cf32 c0 = carry[s * ovl + P]; ai = c0.i; aq = c0.q; }
    for (int m = mmax; m >= 0; m--) {
        const cf32 v = base[(size_t)(b - m) * fft + i + (size_t)m * inp];
        ai += v.i * inv_n; aq += v.q * inv_n;
    }
    out[s * out_pitch + P] = cf32{ai, aq};
}
__global__ __launch_bounds__(256) void k_oa_carry(const cf32 *__restrict__ td, const cf32 *__restrict__ carry_in, cf32 *__restrict__ carry_out,
                                                  int fft, int inp, int ovl, int n_blocks, float inv_n)
{   // pending tail after n_blocks blocks: contributions landing on positions n_blocks*inp + i, i < ovl
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ovl) return;
    const size_t s = blockIdx.y;
    const cf32 *base = td + s * n_blocks * (size_t)fft;
    const size_t P = (size_t)n_blocks * inp + i;
    float ai = 0.f, aq = 0.f;
    if (P < (size_t)ovl) { const cf32 c0 = carry_in[s * ovl + P]; ai = c0.i; aq = c0.q; }
    for (int b = 0; b < n_blocks; b++) {
        const size_t idx = P - (size_t)b * inp;
        if (idx < (size_t)fft) { const cf32 v = base[(size_t)b * fft + idx]; ai += v.i * inv_n; aq += v.q * inv_n; }
    }
    carry_out[s * ovl + i] = cf32{ai, aq};
}

// ------------------------------------------------------------------ fastddc kernels
__global__ __launch_bounds__(256) void k_os_frame(const cf32 *__restrict__ in, const cf32 *__restrict__ tail, cf32 *__restrict__ win,
                                                  int fft, int inp, int ovl)
{   // overlap-save window b = stream[b*inp - ovl, b*inp + inp) (csdr.c:2292-2293); positions < 0 come from the kept tail
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= fft) return;
    const long long pos = (long long)blockIdx.y * inp - ovl + k;
    win[(size_t)blockIdx.y * fft + k] = (pos < 0) ? tail[ovl + pos] : in[pos];
}
__global__ __launch_bounds__(256) void k_os_tail(const cf32 *__restrict__ in, const cf32 *__restrict__ tail_in, cf32 *__restrict__ tail_out, int inp, int ovl, int n_blocks)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= ovl) return;
    const long long pos = (long long)n_blocks * inp - ovl + k;
    tail_out[k] = (pos < 0) ? tail_in[ovl + pos] : in[pos];
}
__global__ __launch_bounds__(256) void k_swap_halves(cf32 *__restrict__ a, int n, size_t total)
{   // fft_swap_sides fastddc.c:91-104 over a batch, out of place not needed: pairwise exchange
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t half = n / 2;
    const size_t row = idx / half, k = idx % half;
    if (row * n >= total) return;
    cf32 *p = a + row * n;
    const cf32 t = p[k]; p[k] = p[k + half]; p[k + half] = t;
}


// alias fold: inv_in[c][b][(m + inv/2) % inv] = (1/pre) * sum_q Xs[i0 + q*inv] * H[c][i0 + q*inv],  i0 = (m + offsetbin - inv/2) mod inv
// (fastddc.c:123-150; Xs = fft_swap_sides(X), the final index shift is the second fft_swap_sides).
// This is the channelizer's traffic: H is fft_size complexf PER CHANNEL (512 KiB at 65536; 128 MiB for 256 channels).  The reference
// streams it once per block; here one lane owns output bin m of one channel for BT blocks at once, so each taps_fft value is
// loaded once per call and reused from a register for all blocks of the call (lanes = consecutive m: coalesced for H and X).
// Channel-tiled fold.  For a residue r = bin mod inv every channel needs the SAME spectrum values X[b][r + q*inv], q = 0..pre-1 (a channel's
// offsetbin only decides which output bin the sum lands in), so a thread owns (residue r, CT channels, BT blocks): per q it loads BT spectrum
// values and CT taps and does CT*BT complex MACs -- 2.7 MACs per load at CT = 8, BT = 4 against 0.94 for the per-channel kernel below, whose
// 9 GB of L2/Infinity-Cache reads per 64-block call were what it ran at (profiles/r1_notes.md).  Index algebra (fastddc.c:106-147 with both
// fft_swap_sides folded in): i = r + q*inv indexes the taps, (i + fft/2) mod fft the spectrum, the sum goes to output bin (r - offsetbin) mod inv.
// Summation order over q is the reference's; products are fused (fmaf), well inside the 1e-5 gate.
template <int CT, int BT>
__global__ __launch_bounds__(256) void k_ddc_fold_ct(const cf32 *__restrict__ spectra, const cf32 *__restrict__ H, cf32 *__restrict__ inv_in,
                                                     const ChanGeom *__restrict__ geom, int fft, int inv, int pre, int n_blocks, int n_channels)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= inv) return;
    const int b0 = blockIdx.y * BT, nb = min(BT, n_blocks - b0);
    const int c0 = blockIdx.z * CT, nc = min(CT, n_channels - c0);
    float ai[CT][BT], aq[CT][BT];
#pragma unroll
    for (int c = 0; c < CT; c++)
#pragma unroll
        for (int k = 0; k < BT; k++) { ai[c][k] = 0.f; aq[c][k] = 0.f; }
    const cf32 *x0 = spectra + (size_t)b0 * fft + r;
    const cf32 *h0 = H + (size_t)c0 * fft + r;
    const int hq = pre / 2;
    // (explicit double buffering of the loads across q was tried: 7.2 -> 6.0 GS/s, the extra registers cost more than the overlap gains)
    for (int q = 0; q < pre; q++) {
        const int xq = (q + hq) & (pre - 1);                        // pre = fft / inv is a power of two
        cf32 xv[BT];
#pragma unroll
        for (int k = 0; k < BT; k++) xv[k] = (k < nb) ? x0[(size_t)k * fft + (size_t)xq * inv] : cf32{0.f, 0.f};
#pragma unroll
        for (int c = 0; c < CT; c++) {
            const cf32 hv = (c < nc) ? h0[(size_t)c * fft + (size_t)q * inv] : cf32{0.f, 0.f};
#pragma unroll
            for (int k = 0; k < BT; k++) {
                ai[c][k] = fmaf(xv[k].i, hv.i, ai[c][k]); ai[c][k] = fmaf(-xv[k].q, hv.q, ai[c][k]);
                aq[c][k] = fmaf(xv[k].i, hv.q, aq[c][k]); aq[c][k] = fmaf(xv[k].q, hv.i, aq[c][k]);
            }
        }
    }
    const float scale = 1.0f / (float)pre;
#pragma unroll
    for (int c = 0; c < CT; c++) {
        if (c >= nc) break;
        int dst = (r - geom[c0 + c].offsetbin) % inv; if (dst < 0) dst += inv;
#pragma unroll
        for (int k = 0; k < BT; k++)
            if (k < nb) inv_in[((size_t)(c0 + c) * n_blocks + b0 + k) * inv + dst] = cf32{ai[c][k] * scale, aq[c][k] * scale};
    }
}

template <int BT>
__global__ __launch_bounds__(256) void k_ddc_fold(const cf32 *__restrict__ spectra, const cf32 *__restrict__ H, cf32 *__restrict__ inv_in,
                                                  const ChanGeom *__restrict__ geom, int fft, int inv, int pre, int n_blocks)
{
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= inv) return;
    const int c = blockIdx.z;
    const int off = geom[c].offsetbin;
    int i0 = (m + off - inv / 2) % inv; if (i0 < 0) i0 += inv;
    const cf32 *h = H + (size_t)c * fft;
    const int dst = (m + inv / 2) % inv;
    const float scale = 1.0f / (float)pre;
    const int b0 = blockIdx.y * BT;
    const int nb = min(BT, n_blocks - b0);
    float ai[BT], aq[BT];
#pragma unroll
    for (int k = 0; k < BT; k++) { ai[k] = 0.f; aq[k] = 0.f; }
    const int half = fft / 2;
    for (int q = 0; q < pre; q++) {
        const int i = i0 + q * inv;
        const cf32 hv = h[i];
        const int xi = (i + half) & (fft - 1);                      // fft_size is a power of two (fastddc.c:50)
        const cf32 *x = spectra + (size_t)b0 * fft + xi;
#pragma unroll
        for (int k = 0; k < BT; k++) {
            if (k < nb) {
                const cf32 xv = x[(size_t)k * fft];
                ai[k] += xv.i * hv.i - xv.q * hv.q;
                aq[k] += xv.i * hv.q + xv.q * hv.i;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < BT; k++)
        if (k < nb) inv_in[((size_t)c * n_blocks + b0 + k) * inv + dst] = cf32{ai[k] * scale, aq[k] * scale};
}

// per channel: the (remain, phase, output offset) chain over the blocks of this call -- data independent
__global__ void k_ddc_chain(DdcChanState *__restrict__ state, const ChanGeom *__restrict__ geom, int n_channels, int n_blocks,
                            int post_in, int post_dec, int *__restrict__ blk_remain, float *__restrict__ blk_phase, int *__restrict__ blk_off, int *__restrict__ counts)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_channels) return;
    DdcChanState s = state[c];
    const float r = geom[c].rate2;
    int off = 0;
    for (int b = 0; b < n_blocks; b++) {
        blk_remain[(size_t)c * n_blocks + b] = s.remain; blk_phase[(size_t)c * n_blocks + b] = s.phase; blk_off[(size_t)c * n_blocks + b] = off;
        int k = 0, pos = s.remain;
        if (pos < post_in) { k = (post_in - 1 - pos) / post_dec + 1; pos += k * post_dec; }
        s.remain = pos - post_in;
        float p = s.phase + r * PI_F * (float)k;                       // libcsdr_gpl.c:155
        while (p > PI_F) p -= 2 * PI_F;
        while (p < -PI_F) p += 2 * PI_F;
        s.phase = p; off += k;
    }
    state[c] = s; counts[c] = off;
}
// one lane per (channel, block): scrap + /inv + decimating_shift_addition_cc replay (fastddc.c:153-162, libcsdr_gpl.c:131-160)
__global__ __launch_bounds__(64) void k_ddc_post(const cf32 *__restrict__ td, cf32 *__restrict__ out, size_t out_pitch, const ChanGeom *__restrict__ geom,
                                                 int n_channels, int n_blocks, int inv, int scrap, int post_in, int post_dec,
                                                 const int *__restrict__ blk_remain, const float *__restrict__ blk_phase, const int *__restrict__ blk_off)
{
    const size_t idx = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (idx >= (size_t)n_channels * n_blocks) return;
    const int c = (int)(idx / n_blocks);
    const cf32 *x = td + idx * inv + scrap;
    cf32 *dst = out + (size_t)c * out_pitch + blk_off[idx];
    const float sd = geom[c].sindelta, cd = geom[c].cosdelta;
    const float ph = blk_phase[idx];
    float co = (float)cos((double)ph), sn = (float)sin((double)ph);
    const float inv_n = 1.0f / (float)inv;
    int k = 0;
    for (int pos = blk_remain[idx]; pos < post_in; pos += post_dec) {
        const cf32 v = cf32{x[pos].i * inv_n, x[pos].q * inv_n};
        dst[k++] = cf32{co * v.i - sn * v.q, sn * v.i + co * v.q};
        const float c1 = co * cd - sn * sd, s1 = sn * cd + co * sd;
        co = c1; sn = s1;
    }
}

__global__ __launch_bounds__(256) void k_cmul(const cf32 *__restrict__ a, const cf32 *__restrict__ b, cf32 *__restrict__ out, size_t n)
{
    const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const cf32 x = a[k], h = b[k];
    out[k] = cf32{x.i * h.i - x.q * h.q, x.i * h.q + x.q * h.i};
}
__global__ __launch_bounds__(256) void k_scale_add(cf32 *__restrict__ io, size_t n, float scale, const cf32 *__restrict__ add, size_t n_add)
{
    const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    cf32 v = io[k]; v.i *= scale; v.q *= scale;
    if (k < n_add) { v.i += add[k].i; v.q += add[k].q; }
    io[k] = v;
}

} // namespace

// cached single-transform plans for the drop-in FFT layer
static std::map<std::pair<int, long>, hipfftHandle> g_c2c_plans;
static std::mutex g_c2c_mu;      // the map is shared by every context (the drop-in layer has one context per host thread)

namespace csdr_amd {
// called by csdr_amd_ctx_destroy: the cached single-transform plans of a context die with its stream
void drop_fft_plans(hipStream_t st)
{
    std::lock_guard<std::mutex> lk(g_c2c_mu);
    for (auto it = g_c2c_plans.begin(); it != g_c2c_plans.end();) {
        if (it->first.second == (long)(uintptr_t)st) { hipfftDestroy(it->second); it = g_c2c_plans.erase(it); } else ++it;
    }
}
}

extern "C" {

int csdr_amd_fft_c2c(csdr_amd_ctx *c, const csdr_complexf *in, csdr_complexf *out, int n, int forward)
{
    auto key = std::make_pair(n, (long)(uintptr_t)c->stream);
    hipfftHandle plan;
    {
        std::lock_guard<std::mutex> lk(g_c2c_mu);
        if (!g_c2c_plans.count(key)) {
            hipfftHandle h;
            CSDR_FFT(hipfftPlan1d(&h, n, HIPFFT_C2C, 1));
            CSDR_FFT(hipfftSetStream(h, c->stream));
            g_c2c_plans[key] = h;
        }
        plan = g_c2c_plans[key];
    }
    CSDR_FFT(hipfftExecC2C(plan, (hipfftComplex *)in, (hipfftComplex *)out, forward ? HIPFFT_FORWARD : HIPFFT_BACKWARD));
    return 0;
}
int csdr_amd_bin_product(csdr_amd_ctx *c, const csdr_complexf *a, const csdr_complexf *b, csdr_complexf *out, size_t n)
{
    if (!n) return 0;
    hipLaunchKernelGGL(k_cmul, dim3(cdiv(n, 256)), dim3(256), 0, c->stream, a, b, out, n); CSDR_LAUNCH_CHECK();
    return 0;
}
int csdr_amd_scale_add(csdr_amd_ctx *c, csdr_complexf *io, size_t n, float scale, const csdr_complexf *add, size_t n_add)
{
    if (!n) return 0;
    hipLaunchKernelGGL(k_scale_add, dim3(cdiv(n, 256)), dim3(256), 0, c->stream, io, n, scale, add, n_add); CSDR_LAUNCH_CHECK();
    return 0;
}

} // extern "C"

// ====================================================================================== overlap-add object
namespace csdr_amd {
struct FftfiltLds;
int fftfilt_lds_pick(int taps_len);
FftfiltLds *fftfilt_lds_create(hipStream_t st, int n, const cf32 *taps, int taps_len, int n_streams);
void fftfilt_lds_destroy(FftfiltLds *p);
int fftfilt_lds_set_taps(FftfiltLds *p, hipStream_t st, const cf32 *taps, int taps_len);
int fftfilt_lds_reset(FftfiltLds *p, hipStream_t st);
int fftfilt_lds_process(FftfiltLds *p, hipStream_t st, const cf32 *in, size_t in_pitch, long m_new, cf32 *out, size_t out_pitch);
const char *fftfilt_lds_kernel_name(const FftfiltLds *p);
int fftfilt_lds_window(const FftfiltLds *p);
int fft64k_tail_add(hipStream_t st, cf32 *out, size_t out_pitch, const cf32 *d_tails, const cf32 *d_carry_in, cf32 *d_carry_out, int inp, int ovl, int n_blocks, int n_streams);
int fft64k_upload_tables(float2 *d_tw);
int fft64k_transpose_taps(hipStream_t st, const cf32 *d_taps_fft, cf32 *d_taps_fft_t);
int fft64k_filter(hipStream_t st, const cf32 *in, size_t in_pitch, int inp, int n_blocks, int n_streams, cf32 *d_work, const cf32 *d_taps_fft_t,
                  const float2 *d_tw, cf32 *d_td);
int fft64k_filter_oa(hipStream_t st, const cf32 *in, size_t in_pitch, int inp, int ovl, int n_blocks, int n_streams, cf32 *d_work, const cf32 *d_taps_fft_t,
                     const float2 *d_tw, cf32 *d_tails, const cf32 *d_carry_in, cf32 *d_carry_out, cf32 *out, size_t out_pitch);
}

struct csdr_amd_fftfilt {
    csdr_amd_ctx *ctx;
    int fft, taps_len, inp, ovl, n_streams, max_blocks;
    cf32 *d_taps_fft, *d_pad, *d_td, *d_carry[2];
    cf32 *d_taps_fft_t; float2 *d_tw;      // fft_size 65536: taps spectrum in [k1][k2] order and the twiddle tables of the three-pass transform (fft64k.hip)
    int flip;
    hipfftHandle plan_one, plan_batch; int plan_batch_n;
    csdr_amd::FftfiltLds *lds;             // taps short enough for LDS-sized windows: the one-pass kernel of fftfilt_lds.hip (every other buffer stays unallocated)
};


static int fftfilt_make_batch_plan(csdr_amd_fftfilt *f, int batch)
{
    if (f->plan_batch_n == batch) return 0;
    if (f->plan_batch_n) { hipfftDestroy(f->plan_batch); f->plan_batch_n = 0; }
    int n[1] = {f->fft};
    CSDR_FFT(hipfftPlanMany(&f->plan_batch, 1, n, nullptr, 1, f->fft, nullptr, 1, f->fft, HIPFFT_C2C, batch));
    CSDR_FFT(hipfftSetStream(f->plan_batch, f->ctx->stream));
    f->plan_batch_n = batch;
    return 0;
}

extern "C" {

int csdr_amd_fftfilt_set_taps(csdr_amd_fftfilt *f, const csdr_complexf *host_taps, int taps_length)
{   // csdr.c:1869-1871: zero padded taps -> forward FFT
    if (taps_length != f->taps_len) return fail_msg(-3, "fftfilt: taps_length changed (%d -> %d); create a new filter", f->taps_len, taps_length);
    if (f->lds) return fftfilt_lds_set_taps(f->lds, f->ctx->stream, host_taps, taps_length);
    std::vector<cf32> pad((size_t)f->fft, cf32{0.f, 0.f});
    for (int k = 0; k < taps_length; k++) pad[k] = host_taps[k];
    CSDR_HIP(hipStreamSynchronize(f->ctx->stream));
    CSDR_HIP(hipMemcpy(f->d_taps_fft, pad.data(), sizeof(cf32) * f->fft, hipMemcpyHostToDevice));
    CSDR_FFT(hipfftExecC2C(f->plan_one, (hipfftComplex *)f->d_taps_fft, (hipfftComplex *)f->d_taps_fft, HIPFFT_FORWARD));
    if (f->d_taps_fft_t) return fft64k_transpose_taps(f->ctx->stream, f->d_taps_fft, f->d_taps_fft_t);
    return 0;
}

csdr_amd_fftfilt *csdr_amd_fftfilt_create(csdr_amd_ctx *ctx, int fft_size, const csdr_complexf *host_taps, int taps_length, int n_streams, int max_blocks)
{
    if (fft_size < 4 || (fft_size & (fft_size - 1)) || taps_length < 1 || taps_length > fft_size || n_streams < 1 || max_blocks < 1) {
        fail_msg(-3, "fftfilt: need power-of-two fft_size >= taps_length"); return nullptr; }
    csdr_amd_fftfilt *f = new csdr_amd_fftfilt();
    f->ctx = ctx; f->fft = fft_size; f->taps_len = taps_length; f->inp = fft_size - taps_length + 1; f->ovl = taps_length - 1;
    f->n_streams = n_streams; f->max_blocks = max_blocks; f->flip = 0; f->plan_batch_n = 0;
    f->lds = nullptr; f->d_taps_fft = f->d_pad = f->d_td = f->d_carry[0] = f->d_carry[1] = f->d_taps_fft_t = nullptr; f->d_tw = nullptr; f->plan_one = 0;
    if (const int win = fftfilt_lds_pick(taps_length)) {
        f->lds = fftfilt_lds_create(ctx->stream, win, host_taps, taps_length, n_streams);
        if (!f->lds) { delete f; return nullptr; }
        return f;
    }
    const size_t tot = (size_t)n_streams * max_blocks * fft_size;
    hipError_t e = hipMalloc((void **)&f->d_taps_fft, sizeof(cf32) * fft_size);
    if (e == hipSuccess) e = hipMalloc((void **)&f->d_pad, sizeof(cf32) * tot);
    if (e == hipSuccess) e = hipMalloc((void **)&f->d_td, sizeof(cf32) * tot);
    if (e == hipSuccess) e = hipMalloc((void **)&f->d_carry[0], sizeof(cf32) * (size_t)n_streams * (f->ovl + 1));
    if (e == hipSuccess) e = hipMalloc((void **)&f->d_carry[1], sizeof(cf32) * (size_t)n_streams * (f->ovl + 1));
    f->d_taps_fft_t = nullptr; f->d_tw = nullptr;
    if (fft_size == 65536 && !getenv("CSDR_AMD_FFT64K_OFF")) {          // config 3: the three-pass transform of fft64k.hip instead of hipFFT
        if (e == hipSuccess) e = hipMalloc((void **)&f->d_taps_fft_t, sizeof(cf32) * fft_size);
        if (e == hipSuccess) e = hipMalloc((void **)&f->d_tw, sizeof(float2) * 512);
        if (e == hipSuccess && fft64k_upload_tables(f->d_tw)) e = hipErrorUnknown;
    }
    if (e != hipSuccess) { fail(e, "hipMalloc(fftfilt)", __FILE__, __LINE__); delete f; return nullptr; }
    if (hipfftPlan1d(&f->plan_one, fft_size, HIPFFT_C2C, 1) != HIPFFT_SUCCESS) { fail_msg(-5, "hipfftPlan1d(%d) failed", fft_size); delete f; return nullptr; }
    hipfftSetStream(f->plan_one, ctx->stream);
    if (csdr_amd_fftfilt_set_taps(f, host_taps, taps_length) || csdr_amd_fftfilt_reset(f)) { delete f; return nullptr; }
    return f;
}

void csdr_amd_fftfilt_destroy(csdr_amd_fftfilt *f)
{
    if (!f) return;
    (void)hipStreamSynchronize(f->ctx->stream);
    if (f->lds) { fftfilt_lds_destroy(f->lds); delete f; return; }
    hipfftDestroy(f->plan_one); if (f->plan_batch_n) hipfftDestroy(f->plan_batch);
    (void)hipFree(f->d_taps_fft); (void)hipFree(f->d_pad); (void)hipFree(f->d_td); (void)hipFree(f->d_carry[0]); (void)hipFree(f->d_carry[1]);
    (void)hipFree(f->d_taps_fft_t); (void)hipFree(f->d_tw);
    delete f;
}

int csdr_amd_fftfilt_input_size(const csdr_amd_fftfilt *f) { return f->inp; }
/* which path serves this filter: the one-pass kernel's name and its window size, or "" / 0 */
const char *csdr_amd_fftfilt_kernel_name(const csdr_amd_fftfilt *f) { return f->lds ? fftfilt_lds_kernel_name(f->lds) : ""; }
int csdr_amd_fftfilt_window(const csdr_amd_fftfilt *f) { return f->lds ? fftfilt_lds_window(f->lds) : 0; }

int csdr_amd_fftfilt_reset(csdr_amd_fftfilt *f)
{   // csdr.c:1862: the first block's overlap source is all zeros
    if (f->lds) return fftfilt_lds_reset(f->lds, f->ctx->stream);
    CSDR_HIP(hipMemsetAsync(f->d_carry[0], 0, sizeof(cf32) * (size_t)f->n_streams * (f->ovl + 1), f->ctx->stream));
    CSDR_HIP(hipMemsetAsync(f->d_carry[1], 0, sizeof(cf32) * (size_t)f->n_streams * (f->ovl + 1), f->ctx->stream));
    f->flip = 0;
    return 0;
}

int csdr_amd_fftfilt_process(csdr_amd_fftfilt *f, const csdr_complexf *in, csdr_complexf *out, int n_blocks, size_t in_pitch, size_t out_pitch)
{
    if (n_blocks <= 0) return 0;
    if (n_blocks > f->max_blocks) return fail_msg(-3, "fftfilt: %d blocks exceed max_blocks %d", n_blocks, f->max_blocks);
    hipStream_t st = f->ctx->stream;
    if (f->lds) return fftfilt_lds_process(f->lds, st, in, in_pitch, (long)n_blocks * f->inp, out, out_pitch);
    const int batch = f->n_streams * n_blocks;
    int rc = 0;
    if (f->d_taps_fft_t && f->ovl <= f->inp) {
        // three passes + a pass over the overlap regions only; the inverse pass writes the output itself (d_td only holds the blocks' tails)
        rc = fft64k_filter_oa(st, in, in_pitch, f->inp, f->ovl, n_blocks, f->n_streams, f->d_pad, f->d_taps_fft_t, f->d_tw, f->d_td, f->d_carry[f->flip], f->d_carry[f->flip ^ 1], out, out_pitch);
        if (rc) return rc;
        if (f->ovl > 0) f->flip ^= 1;
        return 0;
    }
    if (f->d_taps_fft_t) {
        rc = fft64k_filter(st, in, in_pitch, f->inp, n_blocks, f->n_streams, f->d_pad, f->d_taps_fft_t, f->d_tw, f->d_td); if (rc) return rc;
    } else {
    rc = fftfilt_make_batch_plan(f, batch); if (rc) return rc;
    hipLaunchKernelGGL(k_oa_frame, dim3(cdiv(f->fft, 256), n_blocks, f->n_streams), dim3(256), 0, st, in, in_pitch, f->d_pad, f->fft, f->inp, n_blocks); CSDR_LAUNCH_CHECK();
    CSDR_FFT(hipfftExecC2C(f->plan_batch, (hipfftComplex *)f->d_pad, (hipfftComplex *)f->d_pad, HIPFFT_FORWARD));
    const size_t total = (size_t)batch * f->fft;
    hipLaunchKernelGGL(k_bin_product, dim3(cdiv(total, 256)), dim3(256), 0, st, f->d_pad, f->d_taps_fft, f->fft, total); CSDR_LAUNCH_CHECK();
    CSDR_FFT(hipfftExecC2C(f->plan_batch, (hipfftComplex *)f->d_pad, (hipfftComplex *)f->d_td, HIPFFT_BACKWARD));
    }
    const float inv_n = 1.0f / (float)f->fft;
    hipLaunchKernelGGL(k_oa_stitch, dim3(cdiv(f->inp, 256), n_blocks, f->n_streams), dim3(256), 0, st, f->d_td, f->d_carry[f->flip], out, out_pitch,
                       f->fft, f->inp, f->ovl, n_blocks, inv_n); CSDR_LAUNCH_CHECK();
    if (f->ovl > 0) {
        hipLaunchKernelGGL(k_oa_carry, dim3(cdiv(f->ovl, 256), f->n_streams), dim3(256), 0, st, f->d_td, f->d_carry[f->flip], f->d_carry[f->flip ^ 1],
                           f->fft, f->inp, f->ovl, n_blocks, inv_n); CSDR_LAUNCH_CHECK();
        f->flip ^= 1;
    }
    return 0;
}

} // extern "C"

// ====================================================================================== fastddc forward
struct csdr_amd_fastddc_fwd {
    csdr_amd_ctx *ctx; csdr_fastddc_t ddc; int max_blocks;
    cf32 *d_win, *d_tail[2]; int flip;
    std::map<int, hipfftHandle> plans;
};

extern "C" {

csdr_amd_fastddc_fwd *csdr_amd_fastddc_fwd_create(csdr_amd_ctx *ctx, const csdr_fastddc_t *ddc, int max_blocks)
{
    csdr_amd_fastddc_fwd *f = new csdr_amd_fastddc_fwd();
    f->ctx = ctx; f->ddc = *ddc; f->max_blocks = max_blocks; f->flip = 0;
    hipError_t e = hipMalloc((void **)&f->d_win, sizeof(cf32) * (size_t)max_blocks * ddc->fft_size);
    if (e == hipSuccess) e = hipMalloc((void **)&f->d_tail[0], sizeof(cf32) * (size_t)(ddc->overlap_length + 1));
    if (e == hipSuccess) e = hipMalloc((void **)&f->d_tail[1], sizeof(cf32) * (size_t)(ddc->overlap_length + 1));
    if (e != hipSuccess) { fail(e, "hipMalloc(fastddc_fwd)", __FILE__, __LINE__); delete f; return nullptr; }
    (void)hipMemsetAsync(f->d_tail[0], 0, sizeof(cf32) * (size_t)(ddc->overlap_length + 1), ctx->stream);   // csdr.c:2279 null input buffer
    (void)hipMemsetAsync(f->d_tail[1], 0, sizeof(cf32) * (size_t)(ddc->overlap_length + 1), ctx->stream);
    return f;
}

void csdr_amd_fastddc_fwd_destroy(csdr_amd_fastddc_fwd *f)
{
    if (!f) return;
    (void)hipStreamSynchronize(f->ctx->stream);
    for (auto &kv : f->plans) hipfftDestroy(kv.second);
    (void)hipFree(f->d_win); (void)hipFree(f->d_tail[0]); (void)hipFree(f->d_tail[1]);
    delete f;
}

int csdr_amd_fastddc_fwd_process(csdr_amd_fastddc_fwd *f, const csdr_complexf *in, csdr_complexf *spectra, int n_blocks)
{
    if (n_blocks <= 0) return 0;
    if (n_blocks > f->max_blocks) return fail_msg(-3, "fastddc_fwd: %d blocks exceed max_blocks %d", n_blocks, f->max_blocks);
    hipStream_t st = f->ctx->stream;
    const int fft = f->ddc.fft_size, inp = f->ddc.input_size, ovl = f->ddc.overlap_length;
    if (!f->plans.count(n_blocks)) {
        hipfftHandle h; int n[1] = {fft};
        CSDR_FFT(hipfftPlanMany(&h, 1, n, nullptr, 1, fft, nullptr, 1, fft, HIPFFT_C2C, n_blocks));
        CSDR_FFT(hipfftSetStream(h, st));
        f->plans[n_blocks] = h;
    }
    hipLaunchKernelGGL(k_os_frame, dim3(cdiv(fft, 256), n_blocks), dim3(256), 0, st, in, f->d_tail[f->flip], f->d_win, fft, inp, ovl); CSDR_LAUNCH_CHECK();
    CSDR_FFT(hipfftExecC2C(f->plans[n_blocks], (hipfftComplex *)f->d_win, (hipfftComplex *)spectra, HIPFFT_FORWARD));
    hipLaunchKernelGGL(k_os_tail, dim3(cdiv(ovl, 256)), dim3(256), 0, st, in, f->d_tail[f->flip], f->d_tail[f->flip ^ 1], inp, ovl, n_blocks); CSDR_LAUNCH_CHECK();
    f->flip ^= 1;
    return 0;
}

} // extern "C"

// ====================================================================================== fastddc inverse (multi channel)
struct csdr_amd_fastddc_inv {
    csdr_amd_ctx *ctx; int n_channels, max_blocks;
    float tbw; int decimation, window;                             // kept for per-channel retunes
    std::vector<csdr_fastddc_t> geom;
    cf32 *d_H, *d_inv_in, *d_td;
    DdcMfma *mf;                                                    // matrix-core path (fastddc_mfma.hip) when the geometry is config 4's, else nullptr
    ChanGeom *d_geom; DdcChanState *d_state;
    int *d_blk_remain, *d_blk_off, *d_counts; float *d_blk_phase;
    std::map<int, hipfftHandle> plans;
    bool fold_old;                                                  // CSDR_AMD_DDC_FOLD_OLD (A/B: the untiled fold of the general path), read at create
};

extern "C" {

static csdr_amd_fastddc_inv *fastddc_inv_create_comm(csdr_amd_ctx *ctx, float transition_bw, int decimation, const float *shift_rates,
                                                     int n_channels, int window, int max_blocks, const DdcComm *comm);
csdr_amd_fastddc_inv *csdr_amd_fastddc_inv_create(csdr_amd_ctx *ctx, float transition_bw, int decimation, const float *shift_rates,
                                                  int n_channels, int window, int max_blocks)
{
    return fastddc_inv_create_comm(ctx, transition_bw, decimation, shift_rates, n_channels, window, max_blocks, nullptr);
}
static csdr_amd_fastddc_inv *fastddc_inv_create_comm(csdr_amd_ctx *ctx, float transition_bw, int decimation, const float *shift_rates,
                                                     int n_channels, int window, int max_blocks, const DdcComm *comm)
{
    if (n_channels < 1 || max_blocks < 1) { fail_msg(-3, "fastddc_inv: bad sizes"); return nullptr; }
    csdr_amd_fastddc_inv *f = new csdr_amd_fastddc_inv();
    f->ctx = ctx; f->n_channels = n_channels; f->max_blocks = max_blocks; f->fold_old = getenv("CSDR_AMD_DDC_FOLD_OLD") != nullptr;
    f->tbw = transition_bw; f->decimation = decimation; f->window = window;
    f->geom.resize(n_channels);
    std::vector<ChanGeom> cg(n_channels);
    for (int c = 0; c < n_channels; c++) {
        if (csdr_amd_fastddc_init(&f->geom[c], transition_bw, decimation, shift_rates[c])) { fail_msg(-3, "fastddc_init failed (fft_size <= 2)"); delete f; return nullptr; }
        cg[c].offsetbin = f->geom[c].offsetbin; cg[c].sindelta = f->geom[c].dsadata.sindelta; cg[c].cosdelta = f->geom[c].dsadata.cosdelta; cg[c].rate2 = f->geom[c].dsadata.rate;
    }
    const csdr_fastddc_t &g = f->geom[0];
    const int fft = g.fft_size, inv = g.fft_inv_size;
    f->d_inv_in = nullptr; f->d_td = nullptr;
    f->mf = ddc_mfma_create(ctx, fft, inv, g.pre_decimation, n_channels, max_blocks, g.scrap, g.post_input_size, g.post_decimation, g.input_size, g.overlap_length, comm);
    hipError_t e = hipMalloc((void **)&f->d_H, sizeof(cf32) * (size_t)n_channels * fft);
    if (!f->mf) {                                                  // the general path's [channel][block][inv] intermediates
        if (e == hipSuccess) e = hipMalloc((void **)&f->d_inv_in, sizeof(cf32) * (size_t)n_channels * max_blocks * inv);
        if (e == hipSuccess) e = hipMalloc((void **)&f->d_td, sizeof(cf32) * (size_t)n_channels * max_blocks * inv);
    }
    if (e == hipSuccess) e = hipMalloc((void **)&f->d_geom, sizeof(ChanGeom) * n_channels);
    if (e == hipSuccess) e = hipMalloc((void **)&f->d_state, sizeof(DdcChanState) * n_channels);
    if (e == hipSuccess) e = hipMalloc((void **)&f->d_blk_remain, sizeof(int) * (size_t)n_channels * max_blocks);
    if (e == hipSuccess) e = hipMalloc((void **)&f->d_blk_off, sizeof(int) * (size_t)n_channels * max_blocks);
    if (e == hipSuccess) e = hipMalloc((void **)&f->d_blk_phase, sizeof(float) * (size_t)n_channels * max_blocks);
    if (e == hipSuccess) e = hipMalloc((void **)&f->d_counts, sizeof(int) * n_channels);
    if (e != hipSuccess) { fail(e, "hipMalloc(fastddc_inv)", __FILE__, __LINE__); delete f; return nullptr; }
    (void)hipMemcpy(f->d_geom, cg.data(), sizeof(ChanGeom) * n_channels, hipMemcpyHostToDevice);
    (void)hipMemset(f->d_state, 0, sizeof(DdcChanState) * n_channels);                 // csdr.c:2363-2364 bzero(shift_stat)
    // per channel band-pass taps (csdr.c:2345-2349), zero padded, FFT'd in one batch, then fft_swap_sides (csdr.c:2350-2351)
    {
        std::vector<cf32> taps((size_t)n_channels * fft, cf32{0.f, 0.f});
        std::vector<cf32> one(g.taps_length);
        const float half_bw = 0.5f / (float)decimation;
        for (int c = 0; c < n_channels; c++) {
            csdr_amd_firdes_bandpass_c(one.data(), g.taps_length, (-shift_rates[c]) - half_bw, (-shift_rates[c]) + half_bw, window);
            for (int k = 0; k < g.taps_length; k++) taps[(size_t)c * fft + k] = one[k];
        }
        (void)hipMemcpy(f->d_H, taps.data(), sizeof(cf32) * taps.size(), hipMemcpyHostToDevice);
        // The batched plan for the taps is kept for the life of the process, one per (fft, channels), used under a lock: creating AND destroying a plan per
        // object made the create of a SECOND object in a process fault now and then inside this block (GPU memory access fault at addresses of a few MiB, before
        // the new object had processed anything; about one create in five behind a bench run -- profiles/r4_notes.md; not seen once the plan survives).
        // Keyed by the DEVICE too: a plan's twiddles, work buffer and loaded kernels live on the device that was current at its creation; the rank
        // threads of a multi-GPU loopback bank (one context per device, equal n_channels in time-sliced mode) must not share device 0's plan.
        static std::map<std::tuple<int, int, int>, hipfftHandle> taps_plans;
        static std::mutex taps_mu;
        std::lock_guard<std::mutex> lk(taps_mu);
        (void)hipSetDevice(ctx->device);
        const auto key = std::make_tuple(ctx->device, fft, n_channels);
        if (!taps_plans.count(key)) {
            hipfftHandle h; int n[1] = {fft};
            if (hipfftPlanMany(&h, 1, n, nullptr, 1, fft, nullptr, 1, fft, HIPFFT_C2C, n_channels) != HIPFFT_SUCCESS) { fail_msg(-5, "hipfftPlanMany(taps) failed"); delete f; return nullptr; }
            taps_plans[key] = h;
        }
        hipfftHandle h = taps_plans[key];
        hipfftSetStream(h, ctx->stream);
        hipfftExecC2C(h, (hipfftComplex *)f->d_H, (hipfftComplex *)f->d_H, HIPFFT_FORWARD);
        const size_t total = (size_t)n_channels * fft;
        hipLaunchKernelGGL(k_swap_halves, dim3(cdiv(total / 2, 256)), dim3(256), 0, ctx->stream, f->d_H, fft, total);
        if (f->mf && ddc_mfma_set_taps(f->mf, ctx->stream, f->d_H, 0, n_channels)) { delete f; return nullptr; }
        (void)hipStreamSynchronize(ctx->stream);
    }
    return f;
}

// Retune ONE channel: what `csdr fastddc_inv_cc --fd` does when a new rate arrives (csdr.c:2329-2376: geometry, band-pass taps, their FFT and the
// shift status are all rebuilt), applied to one row of the multi-channel object between two process() calls.
int csdr_amd_fastddc_inv_set_rate(csdr_amd_fastddc_inv *f, int channel, float shift_rate)
{
    if (!f || channel < 0 || channel >= f->n_channels) return fail_msg(-3, "fastddc_inv_set_rate: bad channel");
    csdr_amd_ctx *ctx = f->ctx;
    csdr_fastddc_t g;
    if (csdr_amd_fastddc_init(&g, f->tbw, f->decimation, shift_rate)) return fail_msg(-3, "fastddc_init failed");
    f->geom[channel] = g;
    ChanGeom cg; cg.offsetbin = g.offsetbin; cg.sindelta = g.dsadata.sindelta; cg.cosdelta = g.dsadata.cosdelta; cg.rate2 = g.dsadata.rate;
    const int fft = g.fft_size;
    std::vector<cf32> taps((size_t)fft, cf32{0.f, 0.f});
    const float half_bw = 0.5f / (float)f->decimation;
    csdr_amd_firdes_bandpass_c((csdr_complexf *)taps.data(), g.taps_length, (-shift_rate) - half_bw, (-shift_rate) + half_bw, f->window);
    if (f->mf) { const int qrc = ddc_mfma_quiesce(f->mf); if (qrc) return qrc; }   // staged calls read the tables on a side stream
    CSDR_HIP(hipStreamSynchronize(ctx->stream));                     // the previous process() call may still read this channel's row
    CSDR_HIP(hipMemcpy(f->d_geom + channel, &cg, sizeof(ChanGeom), hipMemcpyHostToDevice));
    CSDR_HIP(hipMemset(f->d_state + channel, 0, sizeof(DdcChanState)));
    cf32 *row = f->d_H + (size_t)channel * fft;
    CSDR_HIP(hipMemcpy(row, taps.data(), sizeof(cf32) * (size_t)fft, hipMemcpyHostToDevice));
    int rc = csdr_amd_fft_c2c(ctx, (const csdr_complexf *)row, (csdr_complexf *)row, fft, 1); if (rc) return rc;
    hipLaunchKernelGGL(k_swap_halves, dim3(cdiv((size_t)fft / 2, 256)), dim3(256), 0, ctx->stream, row, fft, (size_t)fft);
    CSDR_LAUNCH_CHECK();
    if (f->mf) { rc = ddc_mfma_set_taps(f->mf, ctx->stream, f->d_H, channel, 1); if (rc) return rc; }
    CSDR_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

void csdr_amd_fastddc_inv_destroy(csdr_amd_fastddc_inv *f)
{
    if (!f) return;
    (void)hipStreamSynchronize(f->ctx->stream);
    for (auto &kv : f->plans) hipfftDestroy(kv.second);
    ddc_mfma_destroy(f->mf);
    (void)hipFree(f->d_H); (void)hipFree(f->d_inv_in); (void)hipFree(f->d_td); (void)hipFree(f->d_geom); (void)hipFree(f->d_state);
    (void)hipFree(f->d_blk_remain); (void)hipFree(f->d_blk_off); (void)hipFree(f->d_blk_phase); (void)hipFree(f->d_counts);
    delete f;
}

const char *csdr_amd_fastddc_inv_kernel_name(const csdr_amd_fastddc_inv *f) { return f->mf ? ddc_mfma_kernel_name(f->mf) : "k_ddc_fold_ct"; }
int csdr_amd_fastddc_inv_set_profiling(csdr_amd_fastddc_inv *f, int on) { return f->mf ? ddc_mfma_set_profiling(f->mf, on) : 0; }
int csdr_amd_fastddc_inv_kernel_time(csdr_amd_fastddc_inv *f, double *total_ms, long *launches)
{
    *total_ms = 0; *launches = 0;
    return f->mf ? ddc_mfma_kernel_time(f->mf, total_ms, launches) : 0;
}
int csdr_amd_fastddc_inv_stage_time(csdr_amd_fastddc_inv *f, int stage, double *total_ms, long *launches)
{
    *total_ms = 0; *launches = 0;
    return f->mf ? ddc_mfma_stage_time(f->mf, stage, total_ms, launches) : 0;
}

int csdr_amd_fastddc_inv_geometry(const csdr_amd_fastddc_inv *f, int channel, csdr_fastddc_t *ddc)
{
    if (channel < 0 || channel >= f->n_channels) return fail_msg(-3, "fastddc_inv: channel out of range");
    *ddc = f->geom[channel]; return 0;
}

int csdr_amd_fastddc_inv_max_output(const csdr_amd_fastddc_inv *f, int n_blocks)
{
    const csdr_fastddc_t &g = f->geom[0];
    return n_blocks * (g.post_input_size / g.post_decimation + 1);
}

int csdr_amd_fastddc_inv_process(csdr_amd_fastddc_inv *f, const csdr_complexf *spectra, int n_blocks, csdr_complexf *out, size_t out_pitch, int *out_counts)
{
    if (n_blocks <= 0) return 0;
    if (n_blocks > f->max_blocks) return fail_msg(-3, "fastddc_inv: %d blocks exceed max_blocks %d", n_blocks, f->max_blocks);
    hipStream_t st = f->ctx->stream;
    const csdr_fastddc_t &g = f->geom[0];
    const int fft = g.fft_size, inv = g.fft_inv_size, pre = g.pre_decimation;
    if ((size_t)csdr_amd_fastddc_inv_max_output(f, n_blocks) > out_pitch) return fail_msg(-3, "fastddc_inv: out_pitch too small");
    if (f->mf) {   // config 4's geometry: fold on the matrix cores, own inverse transforms fused with scrap + residual shift
        int rc = ddc_mfma_submit(f->mf, nullptr, spectra, n_blocks, f->d_state, f->d_geom, true); if (rc) return rc;
        const int *d_cnt = nullptr;
        rc = ddc_mfma_collect(f->mf, f->d_geom, out, out_pitch, &d_cnt); if (rc < 0) return rc;
        if (out_counts) {
            CSDR_HIP(hipMemcpyAsync(out_counts, d_cnt, sizeof(int) * f->n_channels, hipMemcpyDeviceToHost, st));
            CSDR_HIP(hipStreamSynchronize(st));
        }
        return 0;
    }
    const int batch = f->n_channels * n_blocks;
    if (!f->plans.count(batch)) {
        hipfftHandle h; int n[1] = {inv};
        CSDR_FFT(hipfftPlanMany(&h, 1, n, nullptr, 1, inv, nullptr, 1, inv, HIPFFT_C2C, batch));
        CSDR_FFT(hipfftSetStream(h, st));
        f->plans[batch] = h;
    }
    const bool ct_ok = f->n_channels >= 4 && pre >= 2 && (pre & (pre - 1)) == 0 && !f->fold_old;
    // tile choice measured on config 4 (256 channels, 64 blocks per call), GS/s of wideband input: <8,4> 8.6, <8,8> 7.1-7.2, <4,16> 5.0, <6,12> 3.5,
    // <16,4> 3.0, <2,16> 3.0, <16,8> 2.6, <16,2> 1.9, <32,2> 1.0 -- 8 channels share every spectrum value; the 64 accumulators of <8,4> leave room for
    // 4-5 waves per SIMD, which hides the load latency that <8,8> (2 waves per SIMD) exposes
#define FOLD_CT(CTV, BTV) hipLaunchKernelGGL((k_ddc_fold_ct<CTV, BTV>), dim3(cdiv(inv, 256), cdiv(n_blocks, BTV), cdiv(f->n_channels, CTV)), dim3(256), 0, st, spectra, f->d_H, f->d_inv_in, f->d_geom, fft, inv, pre, n_blocks, f->n_channels)
    if (ct_ok && f->n_channels >= 8) { FOLD_CT(8, 4); }
    else if (ct_ok) { FOLD_CT(4, 4); }
#undef FOLD_CT
    else if (n_blocks >= 16) {
        hipLaunchKernelGGL((k_ddc_fold<16>), dim3(cdiv(inv, 256), cdiv(n_blocks, 16), f->n_channels), dim3(256), 0, st, spectra, f->d_H, f->d_inv_in, f->d_geom, fft, inv, pre, n_blocks);
    } else {
        hipLaunchKernelGGL((k_ddc_fold<4>), dim3(cdiv(inv, 256), cdiv(n_blocks, 4), f->n_channels), dim3(256), 0, st, spectra, f->d_H, f->d_inv_in, f->d_geom, fft, inv, pre, n_blocks);
    }
    CSDR_LAUNCH_CHECK();
    CSDR_FFT(hipfftExecC2C(f->plans[batch], (hipfftComplex *)f->d_inv_in, (hipfftComplex *)f->d_td, HIPFFT_BACKWARD));
    hipLaunchKernelGGL(k_ddc_chain, dim3(cdiv(f->n_channels, 64)), dim3(64), 0, st, f->d_state, f->d_geom, f->n_channels, n_blocks, g.post_input_size, g.post_decimation,
                       f->d_blk_remain, f->d_blk_phase, f->d_blk_off, f->d_counts); CSDR_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_ddc_post, dim3(cdiv((size_t)batch, 64)), dim3(64), 0, st, f->d_td, out, out_pitch, f->d_geom, f->n_channels, n_blocks, inv, g.scrap,
                       g.post_input_size, g.post_decimation, f->d_blk_remain, f->d_blk_phase, f->d_blk_off); CSDR_LAUNCH_CHECK();
    if (out_counts) {
        CSDR_HIP(hipMemcpyAsync(out_counts, f->d_counts, sizeof(int) * f->n_channels, hipMemcpyDeviceToHost, st));
        CSDR_HIP(hipStreamSynchronize(st));
    }
    return 0;
}

} // extern "C"

// ====================================================================================== fastddc bank: forward + multi-channel inverse in one object
// The ddcd topology (ddcd_old.cpp:238-252, 474-492: one `csdr fastddc_fwd_cc` feeding N `csdr fastddc_inv_cc --fd` clients) as ONE call per block
// batch: new wideband samples in, every channel's decimated samples out.  At config 4's geometry the forward transform writes the fold's own
// layout directly (fastddc_mfma.hip: no natural-order spectrum, no framing copy); other geometries chain the two halves through a spectrum buffer.
//
// Over several GPUs (csdr_amd_fastddc_bank_create_sharded / _sharded_by) the OUTPUT is always channel-sharded -- rank r delivers a block-distributed slice
// of the channels, the place a client of that channel connects to -- and there are two ways to get there:
//   CSDR_AMD_SHARD_CHANNELS  (default of csdr_amd_fastddc_bank_create_sharded) the compute is channel-sharded too: the forward transform is split by blocks, the transposed spectra are all-gathered
//                            (9.1 B per input sample on every rank's links), every rank folds its channels (fastddc_mfma.hip: ddc_mfma_submit);
//   CSDR_AMD_SHARD_BLOCKS    (opt-in, csdr_amd_fastddc_bank_create_sharded_by) the compute is TIME-sliced: rank r runs the whole single-GPU pipeline -- all channels -- on its run of the batch's
//                            blocks, and only the decimated outputs are exchanged (all-to-all: every rank sends each peer that peer's channels of its run,
//                            1/world of 8 B per input sample per link).  What makes it possible is that the one piece of state that crosses block boundaries,
//                            decimating_shift_addition_cc's (remain, phase) per channel (fastddc.c:152-164), is data independent: every rank walks the chain
//                            over the whole batch itself (ddc_chain_body_seg).  The spectrum never leaves the GPU that computed it.
struct csdr_amd_fastddc_bank {
    csdr_amd_ctx *ctx; csdr_amd_fastddc_inv *inv; csdr_amd_fastddc_fwd *fwd; cf32 *d_spec; int max_blocks; bool fused;
    csdr_amd_comm *comm; int first_channel, n_channels_total; int staged[2], n_staged;      // general path: block counts of the staged calls
    const int *last_counts = nullptr; int last_count_n = 0;            // device counts of the batch collected last (csdr_amd_fastddc_bank_finish)
    // ---- time-sliced mode
    int shard_mode = CSDR_AMD_SHARD_CHANNELS; const DdcComm *dc = nullptr; int world = 1, rank = 0, nbl = 0, out_count = 0;
    csdr_amd_comm *comm_out = nullptr; const DdcComm *dc_out = nullptr;      // time slices: the output exchange (stream xout) on a communicator of its own (csdr_amd_comm_dup), the input exchange (xin) on `comm`
    size_t pitch_loc = 0;
    cf32 *d_in_loc[2] = {nullptr, nullptr}, *d_out_loc[2] = {nullptr, nullptr}, *d_recv[2] = {nullptr, nullptr}, *d_tail_root[2] = {nullptr, nullptr};
    int *d_pref[3] = {nullptr, nullptr, nullptr}; int tail_flip = 0, pref_at = 0;      // run-offset tables: this batch's, the next one's (computed one call ahead), and the one the previous batch's stitch may still read
    hipStream_t xin = nullptr, xout = nullptr;
    hipEvent_t ev_fork = nullptr, ev_in_ready[2] = {nullptr, nullptr}, ev_in_free[2] = {nullptr, nullptr}, ev_out_ready[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr};
    bool in_free_rec[2] = {false, false}, done_rec[2] = {false, false};
    struct Batch { int n_glob; const uint8_t *in; bool local; int fmt; } batch[2]; int fill = 0, drain = 0, n_batches = 0, last_slot = -1;
    // a retune applies from the next SUBMITTED batch on.  Time slices: a batch's chain tables are computed at its collect, so retunes that arrive while batches are
    // staged wait here, tagged with the first batch they concern.  The other modes fix part of a batch's tables at submit and use the rest at collect: there a retune
    // while a batch is staged is refused (seq_submit != seq_collect)
    struct Retune { long first_batch; int channel; float rate; }; std::vector<Retune> deferred; long seq_submit = 0, seq_collect = 0;
};

static void shard_slice(int n, int world, int rank, int *first, int *count)      // block distribution, counts differ by at most one (csdr_amd/dist.py: shard)
{
    const int base = n / world, extra = n % world;
    *count = base + (rank < extra ? 1 : 0); *first = rank * base + (rank < extra ? rank : extra);
}

// out[cc][pref[g][first + cc] + t] = the run of rank g, t < its count: from the received piece (g != me) or this rank's own rows.  pref = the chain's table
// [world + 1][n_channels_total] (fastddc_mfma.hip: ddc_chain_body_seg).  grid (channels of the slice, world, pieces of a run)
__global__ __launch_bounds__(256) void k_bank_stitch(const float2 *__restrict__ own, const float2 *__restrict__ recv, const int *__restrict__ pref, float2 *__restrict__ out,
                                                     size_t pitch_loc, size_t out_pitch, int n_total, int first, int count, int me)
{
    const int cc = blockIdx.x, g = blockIdx.y;
    const int at = pref[(size_t)g * n_total + first + cc], cnt = pref[(size_t)(g + 1) * n_total + first + cc] - at;
    const float2 *src = g == me ? own + (size_t)(first + cc) * pitch_loc : recv + ((size_t)g * count + cc) * pitch_loc;
    float2 *dst = out + (size_t)cc * out_pitch + at;
    const int per = (cnt + (int)gridDim.z - 1) / (int)gridDim.z, t0 = (int)blockIdx.z * per, t1 = min(t0 + per, cnt);
    for (int t = t0 + (int)threadIdx.x; t < t1; t += 256) dst[t] = src[t];
}
// the batch's samples per channel of the slice (the last row of the table)
__global__ __launch_bounds__(64) void k_bank_counts(const int *__restrict__ pref, int *__restrict__ counts, int n_total, int first, int count, int world)
{
    const int cc = blockIdx.x * 64 + threadIdx.x;
    if (cc < count) counts[cc] = pref[(size_t)world * n_total + first + cc];
}

static csdr_amd_fastddc_bank *bank_create(csdr_amd_ctx *ctx, float transition_bw, int decimation, const float *host_shift_rates, int n_channels,
                                          int window, int max_blocks, csdr_amd_comm *comm, int shard_mode)
{
    csdr_amd_fastddc_bank *b = new csdr_amd_fastddc_bank();
    b->ctx = ctx; b->max_blocks = max_blocks; b->fwd = nullptr; b->d_spec = nullptr; b->comm = comm; b->n_staged = 0;
    b->first_channel = 0; b->n_channels_total = n_channels;
    int count = n_channels;
    const DdcComm *dc = comm ? csdr_amd_comm_ddc(comm) : nullptr;
    const bool multi = dc && dc->world > 1;
    if (multi) {
        shard_slice(n_channels, dc->world, dc->rank, &b->first_channel, &count);
        if (count < 1) { fail_msg(-3, "fastddc_bank: fewer channels than ranks"); delete b; return nullptr; }
        b->dc = dc; b->world = dc->world; b->rank = dc->rank; b->out_count = count;
    }
    b->shard_mode = multi ? shard_mode : CSDR_AMD_SHARD_CHANNELS;
    if (b->shard_mode == CSDR_AMD_SHARD_BLOCKS) {                     // every rank: all channels, its run of the blocks
        b->nbl = (max_blocks + b->world - 1) / b->world;
        // two exchanges per batch from two side streams: each on its own communicator (collective -- every rank creates its bank at the same point)
        b->comm_out = csdr_amd_comm_dup(comm);
        if (!b->comm_out) { delete b; return nullptr; }
        b->dc_out = csdr_amd_comm_ddc(b->comm_out);
        b->inv = fastddc_inv_create_comm(ctx, transition_bw, decimation, host_shift_rates, n_channels, window, b->nbl, nullptr);
    } else
        b->inv = fastddc_inv_create_comm(ctx, transition_bw, decimation, host_shift_rates + b->first_channel, count, window, max_blocks, multi ? dc : nullptr);
    if (!b->inv) { csdr_amd_fastddc_bank_destroy(b); return nullptr; }      // (destroy: a BLOCKS-mode bank already holds its second communicator)
    b->fused = ddc_mfma_can_forward(b->inv->mf);
    if (multi && !b->fused) { fail_msg(-3, "fastddc_bank: sharding needs the geometry of the matrix-core path (fft_size 65536, fft_inv_size 512)"); csdr_amd_fastddc_bank_destroy(b); return nullptr; }
    const csdr_fastddc_t g = b->inv->geom[0];
    if (multi && g.input_size < g.overlap_length) { fail_msg(-3, "fastddc_bank: sharding needs input_size >= overlap_length"); csdr_amd_fastddc_bank_destroy(b); return nullptr; }
    if (!b->fused) {
        b->fwd = csdr_amd_fastddc_fwd_create(ctx, &g, max_blocks);
        if (!b->fwd || hipMalloc((void **)&b->d_spec, sizeof(cf32) * (size_t)max_blocks * g.fft_size) != hipSuccess) {
            fail_msg(-2, "fastddc_bank: cannot allocate the spectrum buffer"); csdr_amd_fastddc_bank_destroy(b); return nullptr; }
    }
    if (b->shard_mode == CSDR_AMD_SHARD_BLOCKS) {
        b->pitch_loc = ((size_t)csdr_amd_fastddc_inv_max_output(b->inv, b->nbl) + 8 + 1) / 2 * 2;
        const size_t in_elems = (size_t)b->nbl * g.input_size + g.overlap_length;
        hipError_t e = hipStreamCreateWithFlags(&b->xin, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&b->xout, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&b->ev_fork, hipEventDisableTiming);
        for (int k = 0; k < 2 && e == hipSuccess; k++) {
            e = hipMalloc((void **)&b->d_out_loc[k], sizeof(cf32) * (size_t)n_channels * b->pitch_loc);
            if (e == hipSuccess) e = hipMalloc((void **)&b->d_recv[k], sizeof(cf32) * (size_t)b->world * count * b->pitch_loc);
            for (int t = k; t < 3 && e == hipSuccess; t += 2) e = hipMalloc((void **)&b->d_pref[t], sizeof(int) * ((size_t)(b->world + 1) * n_channels + count));      // the chain's table + this slice's counts
            if (e == hipSuccess && b->rank != 0) e = hipMalloc((void **)&b->d_in_loc[k], sizeof(cf32) * in_elems);
            if (e == hipSuccess && b->rank == 0) { e = hipMalloc((void **)&b->d_tail_root[k], sizeof(cf32) * (size_t)(g.overlap_length + 1));
                if (e == hipSuccess) e = hipMemsetAsync(b->d_tail_root[k], 0, sizeof(cf32) * (size_t)(g.overlap_length + 1), ctx->stream); }      // csdr.c:2279: the stream starts behind zeros
            for (hipEvent_t *ev : {&b->ev_in_ready[k], &b->ev_in_free[k], &b->ev_out_ready[k], &b->ev_done[k]}) if (e == hipSuccess) e = hipEventCreateWithFlags(ev, hipEventDisableTiming);
        }
        if (e != hipSuccess) { fail(e, "fastddc_bank (time-sliced buffers)", __FILE__, __LINE__); csdr_amd_fastddc_bank_destroy(b); return nullptr; }
    }
    return b;
}

// ---- time-sliced mode: submit = the input exchange of a batch (or nothing, when every rank is handed its own run), collect = this rank's pipeline + output exchange
static inline int blk_first_of(const csdr_amd_fastddc_bank *b, int g, int n_glob) { const long long f = (long long)g * b->nbl; return f < n_glob ? (int)f : n_glob; }

static int bank_submit_blocks(csdr_amd_fastddc_bank *b, const void *in_v, int n_glob, bool local, int fmt)
{
    if (b->n_batches == 2) return fail_msg(-3, "fastddc_bank: two batches are already staged; collect one first");
    const uint8_t *in = reinterpret_cast<const uint8_t *>(in_v);
    const size_t es = ddc_in_bytes(fmt);
    const csdr_fastddc_t &g = b->inv->geom[0];
    const int inp = g.input_size, ovl = g.overlap_length, slot = b->fill;
    const int b0 = blk_first_of(b, b->rank, n_glob), b1 = blk_first_of(b, b->rank + 1, n_glob), n_loc = b1 - b0;
    if (!local) {     // the wideband stream lives on rank 0 (ddcd's single fastddc_fwd_cc): every rank is sent the samples of its run, with the overlap in front, over its own link
        const DdcComm *cm = b->dc;
        if (b->rank == 0) { CSDR_HIP(hipEventRecord(b->ev_fork, b->ctx->stream)); CSDR_HIP(hipStreamWaitEvent(b->xin, b->ev_fork, 0)); }      // the producers of `in`
        else if (b->in_free_rec[slot]) CSDR_HIP(hipStreamWaitEvent(b->xin, b->ev_out_ready[slot], 0));                                   // the transforms that read this slot two batches ago
        int rc = cm->group_start(cm); if (rc) return rc;
        if (b->rank == 0) {
            for (int p = 1; p < b->world; p++) {
                const int p0 = blk_first_of(b, p, n_glob), p1 = blk_first_of(b, p + 1, n_glob);
                // (raw samples, 8 / 4 / 2 bytes each: integer ingest halves or quarters the root's egress; counts in 4-byte words)
                if (p1 > p0) { rc = cm->send(cm, in + ((size_t)p0 * inp - ovl) * es, ((size_t)(p1 - p0) * inp + ovl) * es / 4, p, b->xin); if (rc) return rc; }
            }
        } else if (n_loc > 0) { rc = cm->recv(cm, b->d_in_loc[slot], ((size_t)n_loc * inp + ovl) * es / 4, 0, b->xin); if (rc) return rc; }
        rc = cm->group_end(cm); if (rc) return rc;
        CSDR_HIP(hipEventRecord(b->ev_in_ready[slot], b->xin));
    }
    b->batch[slot] = {n_glob, in, local, fmt};
    b->fill ^= 1; b->n_batches++; b->seq_submit++;
    return 0;
}

static int bank_collect_blocks(csdr_amd_fastddc_bank *b, csdr_complexf *out, size_t out_pitch)
{
    if (!b->n_batches) return fail_msg(-3, "fastddc_bank: nothing staged to collect");
    for (size_t k = 0; k < b->deferred.size();) {                     // retunes that concern this batch (and were issued while an earlier one was staged)
        if (b->deferred[k].first_batch <= b->seq_collect) {
            const int rc = csdr_amd_fastddc_inv_set_rate(b->inv, b->deferred[k].channel, b->deferred[k].rate); if (rc) return rc;
            b->deferred.erase(b->deferred.begin() + (long)k);
        } else k++;
    }
    b->seq_collect++;
    csdr_amd_fastddc_inv *f = b->inv;
    const csdr_fastddc_t &g = f->geom[0];
    const int inp = g.input_size, ovl = g.overlap_length, slot = b->drain, W = b->world, me = b->rank;
    const csdr_amd_fastddc_bank::Batch bt = b->batch[slot];
    if ((size_t)csdr_amd_fastddc_inv_max_output(f, bt.n_glob) > out_pitch) return fail_msg(-3, "fastddc_bank: out_pitch too small");
    hipStream_t st = b->ctx->stream;
    const int b0 = blk_first_of(b, me, bt.n_glob), b1 = blk_first_of(b, me + 1, bt.n_glob), n_loc = b1 - b0;
    if (!bt.local && me != 0) CSDR_HIP(hipStreamWaitEvent(st, b->ev_in_ready[slot], 0));
    // the output exchange two batches ago read d_out_loc / the offsets table of this slot: normally long finished -- then no wait packet goes into the stream
    // (a cross-stream wait costs ~10 us of bubble in front of the next kernel even when the event has fired)
    if (b->done_rec[slot] && hipEventQuery(b->ev_done[slot]) != hipSuccess) CSDR_HIP(hipStreamWaitEvent(st, b->ev_done[slot], 0));
    // three tables in rotation: the riders of this batch's inverse-transform kernel write the NEXT batch's offsets while the previous batch's stitch may still
    // read its own table; the one written now was last read by the stitch two batches ago (waited for above)
    int *pref = b->d_pref[b->pref_at], *pref_next = b->d_pref[(b->pref_at + 1) % 3];
    b->pref_at = (b->pref_at + 1) % 3;
    int rc = ddc_mfma_set_segment(f->mf, b->nbl, b0, bt.n_glob, W, pref, pref_next); if (rc) return rc;
    if (n_loc > 0) {
        // the overlap in front of the run: with the run itself (its own format), except on the root of a scattered stream (complexf, carried between batches)
        const size_t es = ddc_in_bytes(bt.fmt);
        const uint8_t *src; const cf32 *tail = nullptr; bool in_front = true;
        if (bt.local) src = bt.in + (size_t)ovl * es;
        else if (me == 0) { tail = b->d_tail_root[b->tail_flip]; src = bt.in; in_front = false; }
        else src = reinterpret_cast<const uint8_t *>(b->d_in_loc[slot]) + (size_t)ovl * es;
        rc = ddc_mfma_submit(f->mf, src, nullptr, n_loc, f->d_state, f->d_geom, true, tail, bt.fmt, in_front); if (rc) return rc;
        rc = ddc_mfma_collect(f->mf, f->d_geom, b->d_out_loc[slot], b->pitch_loc, nullptr, b->ev_out_ready[slot]); if (rc < 0) return rc;
    } else { rc = ddc_mfma_skip_batch(f->mf, f->d_state, f->d_geom); if (rc) return rc; }
    const bool out_ready_recorded = n_loc > 0;                           // (by the call's last kernel itself)
    if (!bt.local && me == 0) {      // the next batch's overlap = the newest ovl samples of the stream (input_size >= overlap_length: checked at create)
        rc = ddc_mfma_convert_samples(st, bt.in, bt.fmt, (long long)bt.n_glob * inp - ovl, ovl, b->d_tail_root[b->tail_flip ^ 1]); if (rc) return rc;
        b->tail_flip ^= 1;
    }
    b->in_free_rec[slot] = true;
    if (!out_ready_recorded) CSDR_HIP(hipEventRecord(b->ev_out_ready[slot], st));
    // output exchange on its own stream (under the next batch's transforms): every rank sends each peer that peer's channels of its run
    const DdcComm *cm = b->dc_out;
    CSDR_HIP(hipStreamWaitEvent(b->xout, b->ev_out_ready[slot], 0));
    rc = cm->group_start(cm); if (rc) return rc;
    for (int p = 0; p < W; p++) {
        if (p == me) continue;
        int pf, pc; shard_slice(b->n_channels_total, W, p, &pf, &pc);
        const int p_loc = blk_first_of(b, p + 1, bt.n_glob) - blk_first_of(b, p, bt.n_glob);
        if (n_loc > 0) { rc = cm->send(cm, b->d_out_loc[slot] + (size_t)pf * b->pitch_loc, 2 * (size_t)pc * b->pitch_loc, p, b->xout); if (rc) return rc; }
        if (p_loc > 0) { rc = cm->recv(cm, b->d_recv[slot] + (size_t)p * b->out_count * b->pitch_loc, 2 * (size_t)b->out_count * b->pitch_loc, p, b->xout); if (rc) return rc; }
    }
    rc = cm->group_end(cm); if (rc) return rc;
    int *d_cnt = pref + (size_t)(W + 1) * b->n_channels_total;
    hipLaunchKernelGGL(k_bank_stitch, dim3(b->out_count, W, 8), dim3(256), 0, b->xout, reinterpret_cast<const float2 *>(b->d_out_loc[slot]), reinterpret_cast<const float2 *>(b->d_recv[slot]),
                       pref, reinterpret_cast<float2 *>(out), b->pitch_loc, out_pitch, b->n_channels_total, b->first_channel, b->out_count, me);
    CSDR_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_bank_counts, dim3(cdiv(b->out_count, 64)), dim3(64), 0, b->xout, pref, d_cnt, b->n_channels_total, b->first_channel, b->out_count, W);
    CSDR_LAUNCH_CHECK();
    CSDR_HIP(hipEventRecord(b->ev_done[slot], b->xout)); b->done_rec[slot] = true;
    b->last_counts = d_cnt; b->last_count_n = b->out_count; b->last_slot = slot;
    b->drain ^= 1; b->n_batches--;
    return 0;
}

extern "C" {

csdr_amd_fastddc_bank *csdr_amd_fastddc_bank_create(csdr_amd_ctx *ctx, float transition_bw, int decimation, const float *host_shift_rates, int n_channels,
                                                    int window, int max_blocks)
{ return bank_create(ctx, transition_bw, decimation, host_shift_rates, n_channels, window, max_blocks, nullptr, CSDR_AMD_SHARD_CHANNELS); }

csdr_amd_fastddc_bank *csdr_amd_fastddc_bank_create_sharded_by(csdr_amd_ctx *ctx, float transition_bw, int decimation, const float *host_shift_rates_all, int n_channels_total,
                                                               int window, int max_blocks, csdr_amd_comm *comm, int shard_mode)
{
    if (shard_mode != CSDR_AMD_SHARD_CHANNELS && shard_mode != CSDR_AMD_SHARD_BLOCKS) { fail_msg(-3, "fastddc_bank: unknown shard mode %d", shard_mode); return nullptr; }
    return bank_create(ctx, transition_bw, decimation, host_shift_rates_all, n_channels_total, window, max_blocks, comm, shard_mode);
}
csdr_amd_fastddc_bank *csdr_amd_fastddc_bank_create_sharded(csdr_amd_ctx *ctx, float transition_bw, int decimation, const float *host_shift_rates_all, int n_channels_total,
                                                            int window, int max_blocks, csdr_amd_comm *comm)
{   // The schedule follows the world size (csdr_amd_fastddc_bank_default_shard_mode).
    const DdcComm *dc = comm ? csdr_amd_comm_ddc(comm) : nullptr;
    return bank_create(ctx, transition_bw, decimation, host_shift_rates_all, n_channels_total, window, max_blocks, comm, csdr_amd_fastddc_bank_default_shard_mode(dc ? dc->world : 1));
}

// Which schedule csdr_amd_fastddc_bank_create_sharded picks for a world of `world` ranks.  Up to two GPUs: BASELINE north_star's partitioning -- the channels sharded,
// the forward transform split by blocks, the transposed spectra all-gathered: at two ranks every link carries half a spectrum per batch each way and the exchange hides
// under the fold.  Beyond two: the all-gather puts (world - 1) / world of 9.1 B per input sample on EVERY rank's links and bounds the bank at ~2.5 x whatever the GPU count
// (per-rank emulation and link budget: DESIGN.md section 6, profiles/r5_fastddc_chanshard_*), while time slices move 8 B per input sample in total: 6.5 x at an emulated
// world of 8 against 5.4 x compute-only for channel shards -- so time slices are the default there.  Either mode stays selectable (csdr_amd_fastddc_bank_create_sharded_by;
// bench_fastddc.py --gpus N measures both with every ingest format and names the best verified one).
int csdr_amd_fastddc_bank_default_shard_mode(int world) { return world > 2 ? CSDR_AMD_SHARD_BLOCKS : CSDR_AMD_SHARD_CHANNELS; }

void csdr_amd_fastddc_bank_destroy(csdr_amd_fastddc_bank *b)
{
    if (!b) return;
    if (b->xin) { (void)hipStreamSynchronize(b->xin); (void)hipStreamDestroy(b->xin); }
    if (b->xout) { (void)hipStreamSynchronize(b->xout); (void)hipStreamDestroy(b->xout); }
    if (b->fwd) csdr_amd_fastddc_fwd_destroy(b->fwd);
    if (b->inv) csdr_amd_fastddc_inv_destroy(b->inv);
    (void)hipFree(b->d_spec);
    for (int k = 0; k < 2; k++) {
        (void)hipFree(b->d_in_loc[k]); (void)hipFree(b->d_out_loc[k]); (void)hipFree(b->d_recv[k]); (void)hipFree(b->d_tail_root[k]); (void)hipFree(b->d_pref[k]); if (k == 0) (void)hipFree(b->d_pref[2]);
        for (hipEvent_t ev : {b->ev_in_ready[k], b->ev_in_free[k], b->ev_out_ready[k], b->ev_done[k]}) if (ev) (void)hipEventDestroy(ev);
    }
    if (b->ev_fork) (void)hipEventDestroy(b->ev_fork);
    if (b->comm_out) csdr_amd_comm_destroy(b->comm_out);
    delete b;
}

int csdr_amd_fastddc_bank_channel_slice(const csdr_amd_fastddc_bank *b, int *first, int *count)
{ *first = b->first_channel; *count = b->shard_mode == CSDR_AMD_SHARD_BLOCKS ? b->out_count : b->inv->n_channels; return 0; }
int csdr_amd_fastddc_bank_shard_mode(const csdr_amd_fastddc_bank *b) { return b->world > 1 ? b->shard_mode : -1; }
// Nothing staged: every retune still queued concerns batches that are not submitted yet, so it takes effect now, in issue order -- BEFORE a newer retune of the
// same channel is applied directly (left in the queue it would be replayed at the next collect on top of the newer rate: ADVICE r4).
static int bank_flush_deferred(csdr_amd_fastddc_bank *b)
{
    for (const auto &d : b->deferred) { const int rc = csdr_amd_fastddc_inv_set_rate(b->inv, d.channel, d.rate); if (rc) return rc; }
    b->deferred.clear();
    return 0;
}
int csdr_amd_fastddc_bank_set_rate(csdr_amd_fastddc_bank *b, int channel, float shift_rate)
{   // channel = index into this rank's OUTPUT slice (an unsharded bank: the channel) in both sharding modes.  A time-sliced bank computes every channel on every
    // rank: there the slice index is mapped to the global channel -- but the other ranks have to hear of the retune too (csdr_amd_fastddc_bank_set_rate_global on
    // every rank); a call on one rank only changes what THIS rank contributes to the channel's output.
    if (b->shard_mode == CSDR_AMD_SHARD_BLOCKS) {
        int first = 0, count = 0;
        if (csdr_amd_fastddc_bank_channel_slice(b, &first, &count) || channel < 0 || channel >= count) return fail_msg(-3, "fastddc_bank: channel %d outside this rank's slice of %d", channel, count);
        if (b->n_batches) { b->deferred.push_back({b->seq_submit, first + channel, shift_rate}); return 0; }      // (staged batches keep the old rate)
        if (const int rc = bank_flush_deferred(b)) return rc;
        return csdr_amd_fastddc_inv_set_rate(b->inv, first + channel, shift_rate);
    }
    if (b->seq_submit != b->seq_collect) return fail_msg(-3, "fastddc_bank: a batch is staged (submitted, not collected): its tables are partly fixed already -- collect it before retuning (a time-sliced bank holds such a retune back itself)");
    return csdr_amd_fastddc_inv_set_rate(b->inv, channel, shift_rate);
}
int csdr_amd_fastddc_bank_set_rate_global(csdr_amd_fastddc_bank *b, int channel, float shift_rate)
{   // every rank makes the same call; a rank applies it to what it computes
    if (channel < 0 || channel >= b->n_channels_total) return fail_msg(-3, "fastddc_bank: channel %d out of range", channel);
    if (b->shard_mode == CSDR_AMD_SHARD_BLOCKS) {
        if (b->n_batches) { b->deferred.push_back({b->seq_submit, channel, shift_rate}); return 0; }                 // (staged batches keep the old rate)
        if (const int rc = bank_flush_deferred(b)) return rc;
        return csdr_amd_fastddc_inv_set_rate(b->inv, channel, shift_rate);
    }
    if (b->seq_submit != b->seq_collect) return fail_msg(-3, "fastddc_bank: a batch is staged (submitted, not collected): its tables are partly fixed already -- collect it before retuning (a time-sliced bank holds such a retune back itself)");
    if (channel < b->first_channel || channel >= b->first_channel + b->inv->n_channels) return 0;
    return csdr_amd_fastddc_inv_set_rate(b->inv, channel - b->first_channel, shift_rate);
}
int csdr_amd_fastddc_bank_input_size(const csdr_amd_fastddc_bank *b) { return b->inv->geom[0].input_size; }
int csdr_amd_fastddc_bank_overlap(const csdr_amd_fastddc_bank *b) { return b->inv->geom[0].overlap_length; }
int csdr_amd_fastddc_bank_max_output(const csdr_amd_fastddc_bank *b, int n_blocks) { return csdr_amd_fastddc_inv_max_output(b->inv, n_blocks); }
csdr_amd_fastddc_inv *csdr_amd_fastddc_bank_inverse(csdr_amd_fastddc_bank *b) { return b->inv; }
int csdr_amd_fastddc_bank_local_blocks(const csdr_amd_fastddc_bank *b, int n_blocks, int *first, int *count)
{
    if (b->shard_mode != CSDR_AMD_SHARD_BLOCKS) { *first = 0; *count = n_blocks; return 0; }
    *first = blk_first_of(b, b->rank, n_blocks); *count = blk_first_of(b, b->rank + 1, n_blocks) - *first;
    return 0;
}

static int bank_submit(csdr_amd_fastddc_bank *b, const void *in, int n_blocks, bool inline_call, int fmt = DDC_IN_CF32)
{
    if (n_blocks <= 0 || n_blocks > b->max_blocks) return fail_msg(-3, "fastddc_bank: %d blocks (max_blocks %d)", n_blocks, b->max_blocks);
    if (fmt < 0 || fmt > 2) return fail_msg(-3, "fastddc_bank: unknown input format %d", fmt);
    if (b->shard_mode == CSDR_AMD_SHARD_BLOCKS) return bank_submit_blocks(b, in, n_blocks, false, fmt);
    if (b->fused) {
        const int rc = ddc_mfma_submit(b->inv->mf, in, nullptr, n_blocks, b->inv->d_state, b->inv->d_geom, inline_call, nullptr, fmt, false);
        if (!rc) b->seq_submit++;
        return rc;
    }
    if (fmt != DDC_IN_CF32) return fail_msg(-3, "fastddc_bank: integer input needs the matrix-core geometry (fft 65536 / inverse 512); convert first (csdr_amd_convert_s16_f / _u8_f)");
    if (b->n_staged) return fail_msg(-3, "fastddc_bank: this geometry stages one call at a time");
    int rc = csdr_amd_fastddc_fwd_process(b->fwd, reinterpret_cast<const csdr_complexf *>(in), b->d_spec, n_blocks); if (rc) return rc;
    b->staged[0] = n_blocks; b->n_staged = 1; b->seq_submit++;
    return 0;
}
int csdr_amd_fastddc_bank_submit(csdr_amd_fastddc_bank *b, const csdr_complexf *in, int n_blocks) { return bank_submit(b, in, n_blocks, false); }
// integer ingest: s16 / u8 IQ pairs, converted inside the forward transform exactly as convert_s16_f / convert_u8_f would (README.md:66-87: the reference feeds
// fastddc_fwd_cc from these converters); a sharded bank scatters the raw integers
int csdr_amd_fastddc_bank_submit_s16(csdr_amd_fastddc_bank *b, const int16_t *in_iq, int n_blocks) { return bank_submit(b, in_iq, n_blocks, false, DDC_IN_S16); }
int csdr_amd_fastddc_bank_submit_u8(csdr_amd_fastddc_bank *b, const uint8_t *in_iq, int n_blocks) { return bank_submit(b, in_iq, n_blocks, false, DDC_IN_U8); }
static int bank_submit_local_fmt(csdr_amd_fastddc_bank *b, const void *in_run, int n_blocks, int fmt)
{
    if (n_blocks <= 0 || n_blocks > b->max_blocks) return fail_msg(-3, "fastddc_bank: %d blocks (max_blocks %d)", n_blocks, b->max_blocks);
    if (b->shard_mode != CSDR_AMD_SHARD_BLOCKS) return fail_msg(-3, "fastddc_bank: submit_local needs a time-sliced bank");
    return bank_submit_blocks(b, in_run, n_blocks, true, fmt);
}
int csdr_amd_fastddc_bank_submit_local(csdr_amd_fastddc_bank *b, const csdr_complexf *in_run, int n_blocks) { return bank_submit_local_fmt(b, in_run, n_blocks, DDC_IN_CF32); }
int csdr_amd_fastddc_bank_submit_local_s16(csdr_amd_fastddc_bank *b, const int16_t *in_run_iq, int n_blocks) { return bank_submit_local_fmt(b, in_run_iq, n_blocks, DDC_IN_S16); }
int csdr_amd_fastddc_bank_submit_local_u8(csdr_amd_fastddc_bank *b, const uint8_t *in_run_iq, int n_blocks) { return bank_submit_local_fmt(b, in_run_iq, n_blocks, DDC_IN_U8); }

// the context's stream (and, with out_counts, the host) behind everything the batch collected last still has in flight on the exchange stream
int csdr_amd_fastddc_bank_finish(csdr_amd_fastddc_bank *b, int *out_counts)
{
    hipStream_t st = b->ctx->stream;
    if (b->shard_mode == CSDR_AMD_SHARD_BLOCKS && b->last_slot >= 0) CSDR_HIP(hipStreamWaitEvent(st, b->ev_done[b->last_slot], 0));
    if (out_counts) {
        if (!b->last_counts) return fail_msg(-3, "fastddc_bank: nothing collected yet");
        CSDR_HIP(hipMemcpyAsync(out_counts, b->last_counts, sizeof(int) * (size_t)b->last_count_n, hipMemcpyDeviceToHost, st));
        CSDR_HIP(hipStreamSynchronize(st));
    }
    return 0;
}

int csdr_amd_fastddc_bank_collect(csdr_amd_fastddc_bank *b, csdr_complexf *out, size_t out_pitch, int *out_counts)
{
    csdr_amd_fastddc_inv *f = b->inv;
    if (b->shard_mode == CSDR_AMD_SHARD_BLOCKS) {
        const int rc = bank_collect_blocks(b, out, out_pitch); if (rc) return rc;
        return out_counts ? csdr_amd_fastddc_bank_finish(b, out_counts) : 0;
    }
    if (!b->fused) {
        if (!b->n_staged) return fail_msg(-3, "fastddc_bank: nothing staged to collect");
        b->n_staged = 0; b->seq_collect++;
        return csdr_amd_fastddc_inv_process(f, b->d_spec, b->staged[0], out, out_pitch, out_counts);
    }
    const int pending = ddc_mfma_pending_blocks(f->mf);                   // validated BEFORE anything is queued that writes `out` with this pitch
    if (pending > 0 && (size_t)csdr_amd_fastddc_inv_max_output(f, pending) > out_pitch) return fail_msg(-3, "fastddc_bank: out_pitch too small");
    const int *d_cnt = nullptr;
    const int rc = ddc_mfma_collect(f->mf, f->d_geom, out, out_pitch, &d_cnt); if (rc < 0) return rc;
    if (b->seq_collect < b->seq_submit) b->seq_collect++;
    b->last_counts = d_cnt; b->last_count_n = f->n_channels;
    return out_counts ? csdr_amd_fastddc_bank_finish(b, out_counts) : 0;
}

static int bank_process_fmt(csdr_amd_fastddc_bank *b, const void *in, int fmt, int n_blocks, csdr_complexf *out, size_t out_pitch, int *out_counts)
{
    if (n_blocks <= 0) return 0;
    if ((size_t)csdr_amd_fastddc_inv_max_output(b->inv, n_blocks) > out_pitch) return fail_msg(-3, "fastddc_bank: out_pitch too small");
    int rc = bank_submit(b, in, n_blocks, true, fmt); if (rc) return rc;
    rc = csdr_amd_fastddc_bank_collect(b, out, out_pitch, out_counts); if (rc) return rc;
    return out_counts ? 0 : csdr_amd_fastddc_bank_finish(b, nullptr);
}
int csdr_amd_fastddc_bank_process(csdr_amd_fastddc_bank *b, const csdr_complexf *in, int n_blocks, csdr_complexf *out, size_t out_pitch, int *out_counts)
{ return bank_process_fmt(b, in, DDC_IN_CF32, n_blocks, out, out_pitch, out_counts); }
int csdr_amd_fastddc_bank_process_s16(csdr_amd_fastddc_bank *b, const int16_t *in_iq, int n_blocks, csdr_complexf *out, size_t out_pitch, int *out_counts)
{ return bank_process_fmt(b, in_iq, DDC_IN_S16, n_blocks, out, out_pitch, out_counts); }
int csdr_amd_fastddc_bank_process_u8(csdr_amd_fastddc_bank *b, const uint8_t *in_iq, int n_blocks, csdr_complexf *out, size_t out_pitch, int *out_counts)
{ return bank_process_fmt(b, in_iq, DDC_IN_U8, n_blocks, out, out_pitch, out_counts); }

} // extern "C"

extern "C" int csdr_amd_fastddc_inv_block(csdr_amd_ctx *c, const csdr_complexf *d_spectrum, const csdr_complexf *d_taps_fft, const csdr_fastddc_t *ddc,
                                          void *status_io, csdr_complexf *d_inv_in, csdr_complexf *d_td, csdr_complexf *d_out)
{
    hipStream_t st = c->stream;
    const int fft = ddc->fft_size, inv = ddc->fft_inv_size;
    ChanGeom g; g.offsetbin = ddc->offsetbin; g.sindelta = ddc->dsadata.sindelta; g.cosdelta = ddc->dsadata.cosdelta; g.rate2 = ddc->dsadata.rate;
    char *scratch = (char *)c->get_scratch(3, 256);
    if (!scratch) return -2;
    ChanGeom *d_g = (ChanGeom *)scratch; DdcChanState *d_s = (DdcChanState *)(scratch + 64);
    int *d_rem = (int *)(scratch + 96), *d_off = (int *)(scratch + 112), *d_cnt = (int *)(scratch + 128); float *d_ph = (float *)(scratch + 144);
    struct { int remain; float phase; int produced; } hs;
    memcpy(&hs, status_io, sizeof(hs));
    DdcChanState s0; s0.remain = hs.remain; s0.phase = hs.phase;
    CSDR_HIP(hipMemcpyAsync(d_g, &g, sizeof(g), hipMemcpyHostToDevice, st));
    CSDR_HIP(hipMemcpyAsync(d_s, &s0, sizeof(s0), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL((k_ddc_fold<4>), dim3(cdiv(inv, 256), 1, 1), dim3(256), 0, st, d_spectrum, d_taps_fft, d_inv_in, d_g, fft, inv, ddc->pre_decimation, 1); CSDR_LAUNCH_CHECK();
    int rc = csdr_amd_fft_c2c(c, d_inv_in, d_td, inv, 0); if (rc) return rc;
    hipLaunchKernelGGL(k_ddc_chain, dim3(1), dim3(64), 0, st, d_s, d_g, 1, 1, ddc->post_input_size, ddc->post_decimation, d_rem, d_ph, d_off, d_cnt); CSDR_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_ddc_post, dim3(1), dim3(64), 0, st, d_td, d_out, (size_t)0, d_g, 1, 1, inv, ddc->scrap, ddc->post_input_size, ddc->post_decimation, d_rem, d_ph, d_off); CSDR_LAUNCH_CHECK();
    DdcChanState s1; int cnt = 0;
    CSDR_HIP(hipMemcpyAsync(&s1, d_s, sizeof(s1), hipMemcpyDeviceToHost, st));
    CSDR_HIP(hipMemcpyAsync(&cnt, d_cnt, sizeof(int), hipMemcpyDeviceToHost, st));
    CSDR_HIP(hipStreamSynchronize(st));
    hs.remain = s1.remain; hs.phase = s1.phase; hs.produced = cnt;
    memcpy(status_io, &hs, sizeof(hs));
    return 0;
}
