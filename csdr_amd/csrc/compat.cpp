// compat.cpp -- the reference's own C API (include/libcsdr_amd_compat.h) on host pointers.
// Every function stages its block into device scratch, runs the batch API's kernels with n_streams = 1
// and copies the result back before returning.  The DSP runs on the device: if the device context cannot be
// created the process is terminated with the reason (there is no CPU fallback).  Host-side code here is limited
// to state bookkeeping, setup-time table generation and format plumbing around the copies.
#include "common.hpp"
#include "../../include/libcsdr_amd_compat.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <unistd.h>
#include <sys/syscall.h>
#include <vector>

using namespace csdr_amd;

namespace {

// One device context PER HOST THREAD (own HIP stream, own scratch / staging buffers): the reference's functions are stateless and
// re-entrant (SURVEY.md section 8b "Threading"), so two threads may call any of them concurrently; with a context each they neither share
// staging buffers nor serialise on one stream.  A thread's context is destroyed when the thread ends; the main thread's lives until the
// process exits (the HIP runtime may already be unloading when its thread-local destructors run).
struct ThreadCtx {
    csdr_amd_ctx *c = nullptr;
    ~ThreadCtx() { if (c && (long)syscall(SYS_gettid) != (long)getpid()) csdr_amd_ctx_destroy(c); }
};
thread_local ThreadCtx t_ctx;

csdr_amd_ctx *ctx()
{
    if (t_ctx.c) return t_ctx.c;
    const char *dev = getenv("CSDR_AMD_DEVICE");
    t_ctx.c = csdr_amd_ctx_create(dev ? atoi(dev) : 0, nullptr);
    if (!t_ctx.c) { fprintf(stderr, "libcsdr_amd: cannot open the MI355X device: %s\n", csdr_amd_last_error()); abort(); }
    return t_ctx.c;
}

void die(const char *where, int rc)
{
    fprintf(stderr, "libcsdr_amd: %s failed (%d): %s\n", where, rc, csdr_amd_last_error());
    abort();
}
#define MUST(call) do { int rc__ = (call); if (rc__ < 0) die(#call, rc__); } while (0)

// staging buffers: slot 4 = input, 5 = output, 6 = aux, 7 = aux2
template <typename T> T *stage_in(int slot, const T *host, size_t n, size_t extra = 0)
{
    csdr_amd_ctx *c = ctx();
    T *d = (T *)c->get_scratch(slot, sizeof(T) * (n + extra) + 64);
    if (!d) die("scratch", -2);
    if (n) MUST(hipMemcpyAsync(d, host, sizeof(T) * n, hipMemcpyHostToDevice, c->stream) == hipSuccess ? 0 : -1);
    return d;
}
template <typename T> T *stage_out(int slot, size_t n)
{
    T *d = (T *)ctx()->get_scratch(slot, sizeof(T) * n + 64);
    if (!d) die("scratch", -2);
    return d;
}
template <typename T> void fetch(T *host, const T *dev, size_t n)
{
    csdr_amd_ctx *c = ctx();
    if (n) MUST(hipMemcpyAsync(host, dev, sizeof(T) * n, hipMemcpyDeviceToHost, c->stream) == hipSuccess ? 0 : -1);
    MUST(hipStreamSynchronize(c->stream) == hipSuccess ? 0 : -1);
}

float run_shifter(int variant, complexf *in, complexf *out, int n, float rate, float phase, int aux)
{
    if (n <= 0) return phase;
    csdr_amd_ctx *c = ctx();
    cf32 *din = stage_in<cf32>(4, (const cf32 *)in, n);
    cf32 *dout = stage_out<cf32>(5, n);
    MUST(csdr_amd_shift_cc(c, variant, rate, &phase, din, dout, 1, (size_t)n, (size_t)n, (size_t)n, n, aux));   // one library call = one chunk
    fetch((cf32 *)out, dout, n);
    return phase;
}

struct FftPlanImpl { int kind; int forward; };   // kind 0: c2c, 1: r2c, 2: c2r

// The device generators take the CLI's `rate` and form the per-sample increment (2*rate)*PI in float themselves.
// Callers of the library API hand over the increment; recover a rate that reproduces it bit for bit.
float rate_from_increment(float inc)
{
    const float guess = inc / (2 * PI_F);
    float cand[3] = { guess, nextafterf(guess, INFINITY), nextafterf(guess, -INFINITY) };
    for (float r : cand) if ((r * 2) * PI_F == inc) return r;
    return guess;
}

} // namespace

extern "C" {

// ------------------------------------------------------------------ design (host side, same code as the batch API)
void firdes_lowpass_f(float *output, int length, float cutoff_rate, window_t window) { csdr_amd_firdes_lowpass_f(output, length, cutoff_rate, (int)window); }
void firdes_bandpass_c(complexf *output, int length, float lowcut, float highcut, window_t window)
{ csdr_amd_firdes_bandpass_c((csdr_complexf *)output, length, lowcut, highcut, (int)window); }
float firdes_wkernel_blackman(float r) { r = (float)(0.5 + r / 2); return (float)(0.42 - 0.5 * cos((double)(2 * PI_F * r)) + 0.08 * cos((double)(4 * PI_F * r))); }  // libcsdr.c:76-82
float firdes_wkernel_hamming(float r) { r = (float)(0.5 + r / 2); return (float)(0.54 - 0.46 * cos((double)(2 * PI_F * r))); }                            // :84-90
float firdes_wkernel_boxcar(float) { return 1.0f; }                                                                                           // :93-96
window_t firdes_get_window_from_string(char *s)
{   // libcsdr.c:57-63
    if (!strcmp(s, "BOXCAR")) return WINDOW_BOXCAR;
    if (!strcmp(s, "BLACKMAN")) return WINDOW_BLACKMAN;
    if (!strcmp(s, "HAMMING")) return WINDOW_HAMMING;
    return WINDOW_DEFAULT;
}
char *firdes_get_string_from_window(window_t w)
{   // libcsdr.c:68-74
    return (char *)(w == WINDOW_BOXCAR ? "BOXCAR" : w == WINDOW_BLACKMAN ? "BLACKMAN" : w == WINDOW_HAMMING ? "HAMMING" : "INVALID");
}
int firdes_filter_len(float tbw) { return csdr_amd_firdes_filter_len(tbw); }
void normalize_fir_f(float *in, float *out, int length)
{   // libcsdr.c:117-125 (setup-time helper on a handful of taps)
    float sum = 0; for (int k = 0; k < length; k++) sum += in[k];
    for (int k = 0; k < length; k++) out[k] = in[k] / sum;
}
int log2n(int x) { return csdr_amd_log2n(x); }
int next_pow2(int x) { return csdr_amd_next_pow2(x); }
float fir_one_pass_ff(float *input, float *taps, int taps_length)
{   // libcsdr.c:675-680: a single dot product; served by the device FIR kernel with one output
    float *din = stage_in<float>(4, input, taps_length, 1), *dt = stage_in<float>(6, taps, taps_length);
    float *dout = stage_out<float>(5, 4);
    MUST(csdr_amd_fir_ff(ctx(), din, dout, 1, taps_length + 1, 0, 0, dt, taps_length));
    float r; fetch(&r, dout, 1); return r;
}

// ------------------------------------------------------------------ converters
#define CONVERT(name, IN_T, OUT_T, DIN_T, DOUT_T, in_count, out_count, call)                                   \
    void name(IN_T *input, OUT_T *output, int input_size) {                                                     \
        if (input_size <= 0) return;                                                                            \
        const size_t n = (size_t)input_size;                                                                    \
        DIN_T *din = stage_in<DIN_T>(4, (const DIN_T *)input, in_count);                                        \
        DOUT_T *dout = stage_out<DOUT_T>(5, out_count);                                                         \
        MUST(call(ctx(), din, dout, n));                                                                        \
        fetch((DOUT_T *)output, dout, out_count);                                                               \
    }
CONVERT(convert_u8_f, unsigned char, float, uint8_t, float, n, n, csdr_amd_convert_u8_f)
CONVERT(convert_s8_f, signed char, float, int8_t, float, n, n, csdr_amd_convert_s8_f)
CONVERT(convert_s16_f, short, float, int16_t, float, n, n, csdr_amd_convert_s16_f)
CONVERT(convert_f_u8, float, unsigned char, float, uint8_t, n, n, csdr_amd_convert_f_u8)
CONVERT(convert_f_s8, float, signed char, float, int8_t, n, n, csdr_amd_convert_f_s8)
CONVERT(convert_f_s16, float, short, float, int16_t, n, n, csdr_amd_convert_f_s16)
void convert_i16_f(short *i, float *o, int n) { convert_s16_f(i, o, n); }     // libcsdr.c:2400
void convert_f_i16(float *i, short *o, int n) { convert_f_s16(i, o, n); }     // libcsdr.c:2401
void convert_f_s24(float *input, unsigned char *output, int input_size, int bigendian)
{
    if (input_size <= 0) return;
    float *din = stage_in<float>(4, input, input_size); uint8_t *dout = stage_out<uint8_t>(5, 3 * (size_t)input_size + 16);
    MUST(csdr_amd_convert_f_s24(ctx(), din, dout, input_size, bigendian));
    fetch((uint8_t *)output, dout, 3 * (size_t)input_size);
}
void convert_s24_f(unsigned char *input, float *output, int input_size, int bigendian)
{
    if (input_size <= 0) return;
    uint8_t *din = stage_in<uint8_t>(4, input, 3 * (size_t)input_size, 16); float *dout = stage_out<float>(5, input_size);
    MUST(csdr_amd_convert_s24_f(ctx(), din, dout, input_size, bigendian));
    fetch(output, dout, input_size);
}

// ------------------------------------------------------------------ shifters
float shift_math_cc(complexf *in, complexf *out, int n, float rate, float phase) { return run_shifter(CSDR_SHIFT_MATH, in, out, n, rate, phase, 0); }

shift_table_data_t shift_table_init(int table_size)
{   // libcsdr.c:211-222; the device regenerates the same table, the host copy is kept for callers that read it
    shift_table_data_t t; t.table = (float *)malloc(sizeof(float) * table_size); t.table_size = table_size;
    for (int k = 0; k < table_size; k++) t.table[k] = (float)sin((double)(((float)k / table_size) * (PI_F / 2)));
    return t;
}
void shift_table_deinit(shift_table_data_t t) { free(t.table); }
float shift_table_cc(complexf *in, complexf *out, int n, float rate, shift_table_data_t t, float phase)
{ return run_shifter(CSDR_SHIFT_TABLE, in, out, n, rate, phase, t.table_size); }

shift_addfast_data_t shift_addfast_init(float rate)
{   // libcsdr.c:307-317
    shift_addfast_data_t d; d.phase_increment = 2 * rate * PI_F;
    for (int j = 0; j < 4; j++) { d.dsin[j] = (float)sin((double)(d.phase_increment * (j + 1))); d.dcos[j] = (float)cos((double)(d.phase_increment * (j + 1))); }
    return d;
}
float shift_addfast_cc(complexf *in, complexf *out, int n, shift_addfast_data_t *d, float phase)
{ return run_shifter(CSDR_SHIFT_ADDFAST, in, out, n, rate_from_increment(d->phase_increment), phase, 0); }

shift_unroll_data_t shift_unroll_init(float rate, int size)
{   // libcsdr.c:268-284
    shift_unroll_data_t d; d.phase_increment = 2 * rate * PI_F; d.size = size;
    d.dsin = (float *)malloc(sizeof(float) * size); d.dcos = (float *)malloc(sizeof(float) * size);
    float a = 0;
    for (int k = 0; k < size; k++) {
        a += d.phase_increment; while (a > PI_F) a -= 2 * PI_F; while (a < -PI_F) a += 2 * PI_F;
        d.dsin[k] = (float)sin((double)a); d.dcos[k] = (float)cos((double)a);
    }
    return d;
}
float shift_unroll_cc(complexf *in, complexf *out, int n, shift_unroll_data_t *d, float phase)
{ return run_shifter(CSDR_SHIFT_UNROLL, in, out, n, rate_from_increment(d->phase_increment), phase, d->size); }

shift_addition_data_t shift_addition_init(float rate)
{   // libcsdr_gpl.c:81-89
    rate *= 2; shift_addition_data_t o;
    o.sindelta = (float)sin((double)(rate * PI_F)); o.cosdelta = (float)cos((double)(rate * PI_F)); o.rate = rate;
    return o;
}
float shift_addition_cc(complexf *in, complexf *out, int n, shift_addition_data_t d, float phase)
{ return run_shifter(CSDR_SHIFT_ADDITION, in, out, n, d.rate / 2, phase, 0); }
float shift_addition_fc(float *in, complexf *out, int n, shift_addition_data_t d, float phase)
{
    if (n <= 0) return phase;
    csdr_amd_ctx *c = ctx();
    float *din = stage_in<float>(4, in, n); cf32 *dout = stage_out<cf32>(5, n);
    cf32 *rot = stage_out<cf32>(7, n);
    MUST(csdr_amd_rotator_generate(c, CSDR_SHIFT_ADDITION, d.rate / 2, &phase, rot, n, n, 0));
    MUST(csdr_amd_mix_fc(c, din, dout, rot, 1, n, n, n));
    fetch((cf32 *)out, dout, n);
    return phase;
}
shift_addition_data_t decimating_shift_addition_init(float rate, int decimation) { return shift_addition_init(rate * decimation); }
decimating_shift_addition_status_t decimating_shift_addition_cc(complexf *in, complexf *out, int n, shift_addition_data_t d, int decimation, decimating_shift_addition_status_t s)
{
    csdr_amd_ctx *c = ctx();
    cf32 *din = stage_in<cf32>(4, (const cf32 *)in, n); cf32 *dout = stage_out<cf32>(5, n / (decimation > 0 ? decimation : 1) + 2);
    shift_addition_data_t *dd = stage_in<shift_addition_data_t>(6, &d, 1);
    int32_t *dst = (int32_t *)stage_in<decimating_shift_addition_status_t>(7, &s, 1);
    MUST(csdr_amd_decimating_shift_addition_cc(c, din, dout, 1, n, n, n, dd, decimation, dst));
    fetch(&s, (decimating_shift_addition_status_t *)dst, 1);
    fetch((cf32 *)out, dout, s.output_size);
    return s;
}

// ------------------------------------------------------------------ filters / demod / audio
int fir_decimate_cc(complexf *in, complexf *out, int n, int decimation, float *taps, int taps_length)
{
    if (n < taps_length) return 0;
    cf32 *din = stage_in<cf32>(4, (const cf32 *)in, n); float *dt = stage_in<float>(6, taps, taps_length);
    cf32 *dout = stage_out<cf32>(5, n / decimation + 2);
    int produced = csdr_amd_fir_decimate_cc(ctx(), din, dout, 1, n, n, n / decimation + 2, decimation, dt, taps_length);
    if (produced < 0) die("fir_decimate_cc", produced);
    fetch((cf32 *)out, dout, produced);
    return produced;
}
int deemphasis_nfm_ff(float *in, float *out, int n, int sample_rate)
{
    const float *taps = nullptr; const int nt = csdr_amd_nfm_deemph_taps(sample_rate, &taps);
    if (!nt) return 0;                                                   // unsupported rate, libcsdr.c:1119
    if (n - nt <= 0) return (n - nt < 0) ? 0 : 0;
    float *din = stage_in<float>(4, in, n); float *dt = stage_in<float>(6, taps, nt); float *dout = stage_out<float>(5, n);
    int produced = csdr_amd_fir_ff(ctx(), din, dout, 1, n, n, n, dt, nt);
    if (produced < 0) die("deemphasis_nfm_ff", produced);
    fetch(out, dout, produced);
    return produced;
}
float deemphasis_wfm_ff(float *in, float *out, int n, float tau, int sample_rate, float last_output)
{
    if (n <= 0) return last_output;
    float *din = stage_in<float>(4, in, n); float *dout = stage_out<float>(5, n); float *dl = stage_in<float>(6, &last_output, 1);
    MUST(csdr_amd_deemphasis_wfm_ff(ctx(), din, dout, 1, n, n, n, tau, sample_rate, dl));
    fetch(out, dout, n);
    float l; fetch(&l, dl, 1); return l;
}
complexf fmdemod_quadri_cf(complexf *in, float *out, int n, float *temp, complexf last)
{
    (void)temp;                                                          // the reference's scratch (libcsdr.c:1042-1043) is not needed
    if (n <= 0) return last;
    cf32 *din = stage_in<cf32>(4, (const cf32 *)in, n); float *dout = stage_out<float>(5, n);
    cf32 *dl = stage_in<cf32>(6, (const cf32 *)&last, 1);
    MUST(csdr_amd_fmdemod_quadri_cf(ctx(), din, dout, 1, n, n, n, dl));
    fetch(out, dout, n);
    return in[n - 1];
}
complexf fmdemod_quadri_novect_cf(complexf *in, float *out, int n, complexf last) { return fmdemod_quadri_cf(in, out, n, nullptr, last); }
void limit_ff(float *in, float *out, int n, float m)
{
    if (n <= 0) return;
    float *din = stage_in<float>(4, in, n); float *dout = stage_out<float>(5, n);
    MUST(csdr_amd_limit_ff(ctx(), din, dout, n, m)); fetch(out, dout, n);
}
void gain_ff(float *in, float *out, int n, float g)
{
    if (n <= 0) return;
    float *din = stage_in<float>(4, in, n); float *dout = stage_out<float>(5, n);
    MUST(csdr_amd_gain_ff(ctx(), din, dout, n, g)); fetch(out, dout, n);
}
void fastagc_ff(fastagc_ff_t *st, float *output)
{   // libcsdr.c:946-991: device state layout [buffer_1 | buffer_2 | peak_1 peak_2 last_gain pad]
    const int n = st->input_size;
    if (n <= 0) return;
    std::vector<float> hs(2 * (size_t)n + 4);
    memcpy(hs.data(), st->buffer_1, sizeof(float) * n); memcpy(hs.data() + n, st->buffer_2, sizeof(float) * n);
    hs[2 * n] = st->peak_1; hs[2 * n + 1] = st->peak_2; hs[2 * n + 2] = st->last_gain; hs[2 * n + 3] = 0;
    float *dstate = stage_in<float>(6, hs.data(), hs.size());
    float *din = stage_in<float>(4, st->buffer_input, n); float *dout = stage_out<float>(5, n);
    MUST(csdr_amd_fastagc_ff(ctx(), din, dout, 1, 1, n, n, n, st->reference, dstate));
    fetch(output, dout, n);
    float tail[4]; fetch(tail, dstate + 2 * n, 4);
    float *recycled = st->buffer_1;                                      // pointer rotation, libcsdr.c:983-988
    st->buffer_1 = st->buffer_2; st->buffer_2 = st->buffer_input; st->buffer_input = recycled;
    st->peak_1 = tail[0]; st->peak_2 = tail[1]; st->last_gain = tail[2];
}

fractional_decimator_ff_t fractional_decimator_ff_init(float rate, int num_poly_points, float *taps, int taps_length)
{   // libcsdr.c:715-748 (host-side state block, same fields and allocations)
    fractional_decimator_ff_t d;
    d.num_poly_points = num_poly_points & ~1;
    d.poly_precalc_denomiator = (float *)malloc(d.num_poly_points * sizeof(float));
    d.xifirst = -(num_poly_points / 2) + 1; d.xilast = num_poly_points / 2;
    int id = 0;
    for (int a = d.xifirst; a <= d.xilast; a++, id++) {
        float prod = 1;
        for (int b = d.xifirst; b <= d.xilast; b++) if (a != b) prod *= (a - b);
        d.poly_precalc_denomiator[id] = prod;
    }
    d.where = -d.xifirst;
    d.coeffs_buf = (float *)malloc(d.num_poly_points * sizeof(float));
    d.filtered_buf = (float *)malloc(d.num_poly_points * sizeof(float));
    d.rate = rate; d.taps = taps; d.taps_length = taps_length; d.input_processed = 0; d.output_size = 0;
    return d;
}
void fractional_decimator_ff(float *in, float *out, int n, fractional_decimator_ff_t *d)
{
    csdr_amd_fracdec *fd = csdr_amd_fracdec_create(d->rate, d->num_poly_points, d->taps, d->taps ? d->taps_length : 0);
    if (!fd) die("fractional_decimator_ff", -3);
    csdr_amd_fracdec_set_where(fd, d->where);
    float *din = stage_in<float>(4, in, n); float *dout = stage_out<float>(5, n);
    int processed = 0;
    int produced = csdr_amd_fractional_decimator_ff(ctx(), fd, din, dout, 1, n, n, n, &processed);
    if (produced < 0) die("fractional_decimator_ff", produced);
    fetch(out, dout, produced);
    d->where = csdr_amd_fracdec_get_where(fd); d->input_processed = processed; d->output_size = produced;
    csdr_amd_fracdec_destroy(fd);
}

// ------------------------------------------------------------------ FFT plan layer (fft_fftw.c:6-45) on hipFFT
void *csdr_fft_malloc(size_t n) { void *p = nullptr; if (posix_memalign(&p, 64, n ? n : 64)) return nullptr; return p; }
void csdr_fft_free(void *p) { free(p); }
// The reference's own header maps fft_malloc / fft_free to fftwf_malloc / fftwf_free (fft_fftw.h:11-12): a client compiled against it asks the dynamic linker for
// these two FFTW symbols and nothing else of FFTW (plans go through make_fft_c2c / fft_execute, served here).  Exported so that such a client needs no -lfftw3f.
// (If the real libfftw3f is loaded as well, whichever comes first in the link order serves both calls: either pair is a plain aligned malloc / free.)
void *fftwf_malloc(size_t n) { return csdr_fft_malloc(n); }
void fftwf_free(void *p) { csdr_fft_free(p); }

static FFT_PLAN_T *make_plan(int size, void *in, void *out, int kind, int forward)
{
    FFT_PLAN_T *p = (FFT_PLAN_T *)malloc(sizeof(FFT_PLAN_T));
    FftPlanImpl *impl = (FftPlanImpl *)malloc(sizeof(FftPlanImpl));
    impl->kind = kind; impl->forward = forward;
    p->size = size; p->input = in; p->output = out; p->plan = impl;
    return p;
}
FFT_PLAN_T *make_fft_c2c(int size, complexf *input, complexf *output, int forward, int) { return make_plan(size, input, output, 0, forward); }
FFT_PLAN_T *make_fft_r2c(int size, float *input, complexf *output, int) { return make_plan(size, input, output, 1, 1); }
FFT_PLAN_T *make_fft_c2r(int size, complexf *input, float *output, int) { return make_plan(size, input, output, 2, 0); }
void fft_destroy(FFT_PLAN_T *p) { if (p) { free(p->plan); free(p); } }
void fft_execute(FFT_PLAN_T *p)
{
    const FftPlanImpl *impl = (const FftPlanImpl *)p->plan;
    const int n = p->size;
    csdr_amd_ctx *c = ctx();
    if (impl->kind == 0) {
        cf32 *din = stage_in<cf32>(4, (const cf32 *)p->input, n); cf32 *dout = stage_out<cf32>(5, n);
        MUST(csdr_amd_fft_c2c(c, din, dout, n, impl->forward));
        fetch((cf32 *)p->output, dout, n);
    } else if (impl->kind == 1) {      // real -> half spectrum: widen to complex on the host side of the copy (setup-rate path)
        std::vector<cf32> tmp(n); const float *x = (const float *)p->input;
        for (int k = 0; k < n; k++) tmp[k] = cf32{x[k], 0.f};
        cf32 *din = stage_in<cf32>(4, tmp.data(), n); cf32 *dout = stage_out<cf32>(5, n);
        MUST(csdr_amd_fft_c2c(c, din, dout, n, 1));
        fetch(tmp.data(), dout, n);
        memcpy(p->output, tmp.data(), sizeof(cf32) * (n / 2 + 1));
    } else {                           // half spectrum -> real
        std::vector<cf32> tmp(n); const cf32 *x = (const cf32 *)p->input;
        for (int k = 0; k <= n / 2; k++) tmp[k] = x[k];
        for (int k = n / 2 + 1; k < n; k++) tmp[k] = cf32{x[n - k].i, -x[n - k].q};
        cf32 *din = stage_in<cf32>(4, tmp.data(), n); cf32 *dout = stage_out<cf32>(5, n);
        MUST(csdr_amd_fft_c2c(c, din, dout, n, 0));
        fetch(tmp.data(), dout, n);
        float *y = (float *)p->output; for (int k = 0; k < n; k++) y[k] = tmp[k].i;
    }
}

void apply_fir_fft_cc(FFT_PLAN_T *plan, FFT_PLAN_T *plan_inverse, complexf *taps_fft, complexf *last_overlap, int overlap_size)
{   // libcsdr.c:814-849, all on the device between one upload and one download per buffer the reference exposes
    const int n = plan->size;
    csdr_amd_ctx *c = ctx();
    cf32 *din = stage_in<cf32>(4, (const cf32 *)plan->input, n);
    cf32 *dspec = stage_out<cf32>(5, n);
    cf32 *dtaps = stage_in<cf32>(6, (const cf32 *)taps_fft, n);
    MUST(csdr_amd_fft_c2c(c, din, dspec, n, 1));
    fetch((cf32 *)plan->output, dspec, n);                               // plan->output holds the spectrum in the reference too
    MUST(csdr_amd_bin_product(c, dspec, dtaps, din, n));
    fetch((cf32 *)plan_inverse->input, din, n);
    MUST(csdr_amd_fft_c2c(c, din, dspec, n, 0));
    cf32 *dov = stage_in<cf32>(7, (const cf32 *)last_overlap, overlap_size);
    MUST(csdr_amd_scale_add(c, dspec, n, 1.0f / (float)n, dov, overlap_size));
    fetch((cf32 *)plan_inverse->output, dspec, n);
}

// ------------------------------------------------------------------ fastddc
int fastddc_init(fastddc_t *ddc, float tbw, int decimation, float shift_rate)
{
    static_assert(sizeof(fastddc_t) == sizeof(csdr_fastddc_t), "fastddc_t layout");
    return csdr_amd_fastddc_init((csdr_fastddc_t *)ddc, tbw, decimation, shift_rate);
}
void fastddc_print(fastddc_t *ddc, char *source)
{   // fastddc.c:75-89 (same text)
    fprintf(stderr,
        "%s: fastddc_print_sizes(): (fft_size = %d) = (taps_length = %d) + (input_size = %d) - 1\n"
        "  overlap     ::  (overlap_length = %d) = taps_length - 1, taps_min_length = %d\n"
        "  decimation  ::  decimation = (pre_decimation = %d) * (post_decimation = %d), fft_inv_size = %d\n"
        "  shift       ::  startbin = %d, offsetbin = %d, v = %d, pre_shift = %g, post_shift = %g\n"
        "  o&s         ::  post_input_size = %d, scrap = %d\n",
        source, ddc->fft_size, ddc->taps_length, ddc->input_size, ddc->overlap_length, ddc->taps_min_length,
        ddc->pre_decimation, ddc->post_decimation, ddc->fft_inv_size, ddc->startbin, ddc->offsetbin, ddc->v,
        ddc->pre_shift, ddc->post_shift, ddc->post_input_size, ddc->scrap);
}
void fft_swap_sides(complexf *io, int fft_size)
{   // fastddc.c:91-104 -- an in-place permutation of a host buffer (setup helper; the batch path folds it into index math)
    const int h = fft_size / 2;
    for (int k = 0; k < h; k++) { complexf t = io[k]; io[k] = io[k + h]; io[k + h] = t; }
}
decimating_shift_addition_status_t fastddc_inv_cc(complexf *input, complexf *output, fastddc_t *ddc, FFT_PLAN_T *plan_inverse, complexf *taps_fft,
                                                  decimating_shift_addition_status_t st)
{
    csdr_amd_ctx *c = ctx();
    const int fft = ddc->fft_size, inv = ddc->fft_inv_size;
    cf32 *dspec = stage_in<cf32>(4, (const cf32 *)input, fft);
    cf32 *dtaps = stage_in<cf32>(6, (const cf32 *)taps_fft, fft);
    cf32 *dwork = stage_out<cf32>(5, 2 * (size_t)inv + ddc->post_input_size + 8);
    cf32 *dinv_in = dwork, *dtd = dwork + inv, *dout = dwork + 2 * inv;
    MUST(csdr_amd_fastddc_inv_block(c, dspec, dtaps, (const csdr_fastddc_t *)ddc, &st, dinv_in, dtd, dout));
    fetch((cf32 *)output, dout, st.output_size);
    if (plan_inverse) {                                                  // the reference leaves its work in the plan's buffers
        if (plan_inverse->input) fetch((cf32 *)plan_inverse->input, dinv_in, inv);
        if (plan_inverse->output) {
            fetch((cf32 *)plan_inverse->output, dtd, inv);
            cf32 *o = (cf32 *)plan_inverse->output; const float s = 1.0f / (float)inv;
            for (int k = 0; k < inv; k++) { o[k].i *= s; o[k].q *= s; }
        }
    }
    fft_swap_sides(input, fft);                                          // the reference swaps its input in place (fastddc.c:123)
    return st;
}


// ------------------------------------------------------------------ f2 blocks (libcsdr.c:861-941, 1004-1019, 1245-1303; libcsdr_gpl.c:163-260)
void amdemod_cf(complexf *in, float *out, int n)
{
    if (n <= 0) return;
    cf32 *din = stage_in<cf32>(4, (const cf32 *)in, n); float *dout = stage_out<float>(5, n);
    MUST(csdr_amd_amdemod_cf(ctx(), (const csdr_complexf *)din, dout, n)); fetch(out, dout, n);
}
void amdemod_estimator_cf(complexf *in, float *out, int n, float alpha, float beta)
{
    if (n <= 0) return;
    cf32 *din = stage_in<cf32>(4, (const cf32 *)in, n); float *dout = stage_out<float>(5, n);
    MUST(csdr_amd_amdemod_estimator_cf(ctx(), (const csdr_complexf *)din, dout, n, alpha, beta)); fetch(out, dout, n);
}
void logpower_cf(complexf *in, float *out, int n, float add_db)
{
    if (n <= 0) return;
    cf32 *din = stage_in<cf32>(4, (const cf32 *)in, n); float *dout = stage_out<float>(5, n);
    MUST(csdr_amd_logpower_cf(ctx(), (const csdr_complexf *)din, dout, n, add_db)); fetch(out, dout, n);
}
float fmdemod_atan_cf(complexf *in, float *out, int n, float last_phase)
{
    if (n <= 0) return last_phase;
    cf32 *din = stage_in<cf32>(4, (const cf32 *)in, n); float *dout = stage_out<float>(5, n); float *dl = stage_in<float>(6, &last_phase, 1);
    MUST(csdr_amd_fmdemod_atan_cf(ctx(), (const csdr_complexf *)din, dout, 1, n, n, n, dl));
    fetch(out, dout, n);
    float l; fetch(&l, dl, 1); return l;
}
dcblock_preserve_t dcblock_ff(float *in, float *out, int n, float a, dcblock_preserve_t preserved)
{
    if (n <= 0) return preserved;
    float *din = stage_in<float>(4, in, n); float *dout = stage_out<float>(5, n); float *ds = stage_in<float>(6, &preserved.last_input, 2);
    MUST(csdr_amd_dcblock_ff(ctx(), din, dout, 1, n, n, n, a, ds));
    fetch(out, dout, n);
    fetch(&preserved.last_input, ds, 2);
    return preserved;
}
float fastdcblock_ff(float *in, float *out, int n, float last_dc_level)
{
    if (n <= 0) return last_dc_level;
    float *din = stage_in<float>(4, in, n); float *dout = stage_out<float>(5, n); float *dl = stage_in<float>(6, &last_dc_level, 1);
    MUST(csdr_amd_fastdcblock_ff(ctx(), din, dout, 1, 1, n, n, n, dl));
    fetch(out, dout, n);                                                 // in == out is allowed by the reference: the device buffers are distinct
    float l; fetch(&l, dl, 1); return l;
}
float agc_ff(float *in, float *out, int n, float reference, float attack_rate, float decay_rate, float max_gain, short hang_time, short attack_wait_time,
             float gain_filter_alpha, float last_gain)
{
    if (n <= 0) return last_gain;
    float *din = stage_in<float>(4, in, n); float *dout = stage_out<float>(5, n); float *dl = stage_in<float>(6, &last_gain, 1);
    MUST(csdr_amd_agc_ff(ctx(), din, dout, 1, n, n, n, n, reference, attack_rate, decay_rate, max_gain, hang_time, attack_wait_time, gain_filter_alpha, dl));
    fetch(out, dout, n);
    float l; fetch(&l, dl, 1); return l;
}
float *precalculate_window(int size, window_t window)
{   // libcsdr.c:1256-1267: caller owns the malloc'ed table
    float *w = (float *)malloc(sizeof(float) * (size > 0 ? size : 1));
    if (w && size > 0) csdr_amd_precalculate_window(w, size, (int)window);
    return w;
}
void apply_precalculated_window_c(complexf *in, complexf *out, int size, float *windowt)
{   // libcsdr.c:1269-1276: one float product per component; a host loop is the whole job at this size, the device path is csdr_amd_fftcc
    for (int k = 0; k < size; k++) { out[k].i = in[k].i * windowt[k]; out[k].q = in[k].q * windowt[k]; }
}
void apply_window_c(complexf *in, complexf *out, int size, window_t window)
{   // libcsdr.c:1245-1254
    float *w = precalculate_window(size, window);
    apply_precalculated_window_c(in, out, size, w);
    free(w);
}


// ------------------------------------------------------------------ f3: IMA ADPCM (ima_adpcm.h:5-11)
ima_adpcm_state_t encode_ima_adpcm_i16_u8(short *in, unsigned char *out, int n, ima_adpcm_state_t state)
{
    if (n < 2) return state;
    int16_t *din = stage_in<int16_t>(4, in, n); uint8_t *dout = stage_out<uint8_t>(5, n / 2 + 16); int *ds = stage_in<int>(6, &state.index, 2);
    MUST(csdr_amd_encode_ima_adpcm_i16_u8(ctx(), din, dout, 1, n, n, n / 2, ds));
    fetch(out, dout, n / 2); fetch(&state.index, ds, 2);
    return state;
}
ima_adpcm_state_t decode_ima_adpcm_u8_i16(unsigned char *in, short *out, int n, ima_adpcm_state_t state)
{
    if (n <= 0) return state;
    uint8_t *din = stage_in<uint8_t>(4, in, n); int16_t *dout = stage_out<int16_t>(5, 2 * (size_t)n + 16); int *ds = stage_in<int>(6, &state.index, 2);
    MUST(csdr_amd_decode_ima_adpcm_u8_i16(ctx(), din, dout, 1, n, n, 2 * (size_t)n, ds));
    fetch(out, dout, 2 * (size_t)n); fetch(&state.index, ds, 2);
    return state;
}

} // extern "C"
